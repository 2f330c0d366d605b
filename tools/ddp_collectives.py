"""Collectives of one data-parallel training step, counted and timed on ONE GPU.

A one-rank RCCL group (`backend="nccl"`) with AFD_FORCE_COLLECTIVES=1: the step issues every collective it would issue
with N ranks -- the gradient arena all-reduce from the end-of-backward hook and the packed SyncBatchNorm sums of every
BatchNorm forward / backward (reference train_classifier.py:319-323 DDP wrap; models.py:260-289 SyncBatchNorm).  What a
one-GPU box can show: how many there are, what they carry, and what they cost the HOST to enqueue (c10d + RCCL launch
path).  With one rank RCCL has nothing to exchange, so the device-side cost of a real exchange over xGMI is NOT
measured here -- the multi-GPU scaling of this project is unmeasured on hardware.

    python3 tools/ddp_collectives.py [out.json]
"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "audiodeepfake-detection_amd"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("LOCAL_RANK", "0")
os.environ["AFD_FORCE_COLLECTIVES"] = "1"
import torch
import torch.distributed as dist
import bench

dev = torch.device("cuda:0")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", device_id=dev)
log = []
real_all_reduce = dist.all_reduce

def counted(t, *a, **kw):
    t0 = time.perf_counter()
    r = real_all_reduce(t, *a, **kw)
    log.append((t.numel() * t.element_size(), str(t.dtype).replace("torch.", ""), bool(kw.get("async_op")), 1e6 * (time.perf_counter() - t0)))
    return r

def counted_async_only(t, *a, **kw):
    # the synchronous calls arrive through ops.all_reduce_sum (counted there); the arena's async call comes straight here
    return counted(t, *a, **kw) if kw.get("async_op") else real_all_reduce(t, *a, **kw)

dist.all_reduce = counted_async_only
from audiofakedetect import ops
real_sum = ops.all_reduce_sum

def counted_sum(t):
    t0 = time.perf_counter()
    real_sum(t)
    log.append((t.numel() * t.element_size(), str(t.dtype).replace("torch.", ""), t.numel() > 100000, 1e6 * (time.perf_counter() - t0)))

ops.all_reduce_sum = counted_sum
out = {"world": 1, "backend": "rccl " + ".".join(str(v) for v in torch.cuda.nccl.version()),
       "note": "one rank, collectives forced (AFD_FORCE_COLLECTIVES=1): counts, payloads and HOST enqueue time; the device "
               "cost of an exchange between GPUs is not measurable on one GPU.  collectives_on = through torch.distributed "
               "(c10d hands each to its own stream and back); collectives_on_direct_rccl = ncclAllReduce on the compute "
               "stream from libafd_hip (ops.enable_direct_rccl, opt-in AFD_RCCL_DIRECT=1)", "workloads": {}}
for w in ("coif4-l8", "coif4-l14"):
    torch.manual_seed(0)
    args, trainer, _ = bench.build(w, 128, True, dev)
    batch = bench.synthetic_batch(128, 0, dev)
    trainer.model.train()
    res = {}
    runs = {"collectives_on": [], "collectives_off": [], "collectives_on_direct_rccl": []}
    for forced in (False, True, "direct", False, True, "direct", False, True, "direct"):  # alternating; the fastest run of each mode is reported
        if forced:
            os.environ["AFD_FORCE_COLLECTIVES"] = "1"
        else:
            os.environ.pop("AFD_FORCE_COLLECTIVES", None)
        if forced == "direct":
            assert ops.enable_direct_rccl()
        else:
            ops.disable_direct_rccl()
        for _ in range(8):
            trainer._run_batch(0, batch)
        torch.cuda.synchronize()
        log.clear()
        n = 20
        t0 = time.perf_counter()
        for _ in range(n):
            trainer._run_batch(0, batch)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        key = "collectives_on_direct_rccl" if forced == "direct" else "collectives_on" if forced else "collectives_off"
        runs[key].append(1e3 * (t2 - t0) / n)
        res[key] = {"ms_per_step": min(runs[key]), "ms_per_step_runs": runs[key]}
        if forced is True:
            per_step = len(log) / n
            kinds2 = {}
            for nbytes, dtype, is_async, us in log:
                k = "gradient arena all-reduce (async, end of backward)" if is_async else f"packed BatchNorm sums ({dtype})"
                d = kinds2.setdefault(k, {"n": 0, "bmin": nbytes, "bmax": nbytes, "us": 0.0})
                d["n"] += 1; d["bmin"] = min(d["bmin"], nbytes); d["bmax"] = max(d["bmax"], nbytes); d["us"] += us
            res["collectives_per_step"] = per_step
            res["by_kind"] = {k: {"per_step": d["n"] / n, "payload_bytes_min": d["bmin"], "payload_bytes_max": d["bmax"],
                                  "host_us_each": d["us"] / d["n"]} for k, d in kinds2.items()}
            res["host_us_per_step_in_collective_calls"] = sum(v[3] for v in log) / n
    res["step_cost_of_issuing_them_ms"] = res["collectives_on"]["ms_per_step"] - res["collectives_off"]["ms_per_step"]
    res["share_of_step"] = res["step_cost_of_issuing_them_ms"] / res["collectives_off"]["ms_per_step"]
    res["step_cost_direct_rccl_ms"] = res["collectives_on_direct_rccl"]["ms_per_step"] - res["collectives_off"]["ms_per_step"]
    out["workloads"][w] = res
    print(w, json.dumps(res), flush=True)
    del trainer, batch
    torch.cuda.empty_cache()
if len(sys.argv) > 1:
    with open(sys.argv[1], "w") as fh:
        json.dump(out, fh, indent=1)
ops.disable_direct_rccl()
dist.destroy_process_group()
