"""Headline benchmark: 1 s @ 22 050 Hz frames/s of (wavelet-packet front end + DCNN train step).

Contract: ``python bench.py --gpus N --steps K --warmup W`` (for N > 1 launched by
``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...``, one rank per
GPU over RCCL).  Rank 0 prints ONE JSON line.

A step is the reference's training step (train_classifier.py:945-995) on one synthetic batch
that is already resident in HBM: zero_grad -> WPT (+log, +normalise) -> DCNN forward -> cross
entropy -> backward -> gradient all-reduce (N > 1) -> Adam, plus the per-step loss/accuracy
read-back the reference performs.  Default workload = BASELINE.json configs[1]:
packets-coif4 level 14 + DCNN, batch 128 per GPU, fp32.

Extra objects on the JSON line:
  roofline      the kernel class with the largest share of the step, timed live with HIP
                events on its launch stream (afd_timing_*), algorithmic flops (conv) or bytes
                (front end) divided by the summed launch durations;
  cpu_baseline  oracle/torch_ref.py (the reference's algorithm on torch CPU, per-node packet
                recursion with Welford left on) timed on this host's cores on a bounded sample.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "audiodeepfake-detection_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

METRIC = "1s@22050Hz frames/sec (WPT-coif4 + DCNN train step) at 1/2/4/8 MI355X"
PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_HBM_GBS = 8000.0         # MI355X_MICROARCH.md: HBM3E spec

WORKLOADS = {
    # name: (transform, wavelet, num_of_scales, time_dim_add, description)
    "coif4-l14": ("packets", "coif4", 16384, 0, "packets-coif4 level-14 + DCNN train step"),
    "coif4-l8": ("packets", "coif4", 256, 0, "packets-coif4 level-8 + DCNN train step"),
    "sym5-l8": ("packets", "sym5", 256, 1, "packets-sym5 level-8 + DCNN train step"),
    "sym5-l14": ("packets", "sym5", 16384, 0, "packets-sym5 level-14 + DCNN train step"),
    "stft": ("stft", "none", 256, 0, "STFT(n_fft 511, hop 220) + DCNN train step"),
    # BASELINE configs[4]: evaluation (forward + argmax) of the LCNN head on STFT features
    "stft-lcnn-eval": ("stft", "none", 256, 0, "STFT(n_fft 511, hop 220) + LCNN eval forward (fp32)"),
    "stft-lcnn": ("stft", "none", 256, 0, "STFT(n_fft 511, hop 220) + LCNN train step (fp32)"),
}


def build(workload: str, batch: int, ddp: bool, device):
    from audiofakedetect import ops
    from audiofakedetect.models import DCNN
    from audiofakedetect.train_classifier import Trainer
    from audiofakedetect.utils import DotDict
    from audiofakedetect.wavelet_math import get_transforms

    transform, wavelet, scales, add, _ = WORKLOADS[workload]
    args = DotDict(
        transform=transform, wavelet=wavelet, num_of_scales=scales, features="none", log_scale=True,
        loss_less="False", power=2.0, hop_length=220, sample_rate=22050, seconds=1, mean=0.0,
        std=1.0, block_norm=False, log_dir="/tmp/afd_bench", data_path=None, only_use=None,
        batch_size=batch, ddp=ddp, ochannels1=64, ochannels2=64, ochannels3=96, ochannels4=128,
        ochannels5=32, kernel1=3, dropout_cnn=0.6, dropout_lstm=0.2, time_dim_add=add,
        learning_rate=4e-4, weight_decay=1e-3, synthetic=True,
    )
    transforms, normalize = get_transforms(args, "none", str(device), False, verbose=False)
    with torch.no_grad():
        probe, _ = transforms(torch.zeros(1, 1, 22050, device=device))
    args.input_dim = [batch] + list(probe.shape[1:])
    # flattened size of the dil_conv output: [time_dim, 64-24=40, P/8-24]
    p8 = args.input_dim[2] // 8
    args.flattend_size = (64 - 24) * (p8 - 24)
    if "lcnn" in workload:
        from audiofakedetect.lcnn import LCNN

        model = LCNN(classes=2, in_channels=1, lstm_channels=scales).to(device)
    else:
        model = DCNN(args).to(device)
    opt = ops.FusedAdam(model.parameters(), lr=args.learning_rate, weight_decay=args.weight_decay)
    trainer = Trainer("/tmp/afd_bench/snap", args, normalize, transforms, None, model, None, None,
                      None, None, opt, ops.CrossEntropyLoss(), None)
    return args, trainer


def synthetic_batch(batch: int, rank: int, device):
    g = torch.Generator().manual_seed(1234 + rank)
    audio = (0.1 * torch.randn(batch, 1, 22050, generator=g)).clamp_(-1.0, 1.0)
    # cross-generator style labels {0: real, 1: B_melgan, 2: C_hifigan}; the step binarises
    # them with `!= 0` as the reference does (train_classifier.py:955-957)
    labels = torch.randint(0, 3, (batch,), generator=g, dtype=torch.int64)
    return {"audio": audio.to(device), "label": labels.to(device)}


def cpu_baseline(workload: str, frames: int, crop_packets: int = 4096):
    """The reference's algorithm on torch CPU (oracle 'port'), on a bounded sample.

    Front end: `frames` full frames (per-node pad+conv1d recursion, Welford on, as the
    reference runs it).  DCNN train step: the same frames; for the level-14 workloads the
    packet axis is cropped to `crop_packets` of P packets and the time scaled by P/crop (the
    convolutions are translation invariant along that axis, cost is linear in it) -- one full
    level-14 frame is 64 GFLOP of fp32 convolutions, minutes of CPU time.
    """
    from oracle import torch_ref, wpt_oracle

    transform, wavelet, scales, add, _ = WORKLOADS[workload]
    cores = min(os.cpu_count() or 1, 16)  # the GPU box's CPU share for one GPU
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(99)
    x = (0.1 * torch.randn(frames, 1, 22050, generator=g)).clamp_(-1, 1)
    labels = torch.randint(0, 2, (frames,), generator=g)
    level = scales.bit_length() - 1
    t0 = time.perf_counter()
    if transform == "packets":
        feats, _ = torch_ref.packets_torch(x, wpt_oracle.TAPS[wavelet], level, log_scale=True,
                                           compute_welford=True, per_node=True)
    else:
        feats = torch_ref.stft_torch(x, 2 * scales - 1, 220, log_scale=True)
    feats = torch_ref.normalize_torch(feats, 0.0, 1.0)
    t_fe = time.perf_counter() - t0
    log(f"cpu baseline: front end {t_fe:.2f} s for {frames} frames ({cores} threads)")
    packets = feats.shape[2]
    crop = packets if packets <= 1024 else crop_packets
    fc = feats[:, :, :crop, :].contiguous()
    net = torch_ref.DCNNRef(fc.shape, time_dim_add=add, flattend_size=40 * (crop // 8 - 24))
    opt = torch.optim.Adam(net.parameters(), lr=4e-4, weight_decay=1e-3)
    net.train()
    t1 = time.perf_counter()
    torch_ref.train_step_torch(net, opt, fc, labels)
    t_step = (time.perf_counter() - t1) * (packets / crop)
    total = t_fe + t_step
    crop_note = "" if crop == packets else f" measured on packets[0:{crop}] of {packets} and scaled x{packets // crop}"
    return {
        "value": frames / total, "unit": "frames/s", "cores": cores, "kind": "port",
        "sample": f"{frames} frame(s) of the same workload, one step: front end {t_fe:.2f} s "
                  f"(per-node pad+conv1d recursion, Welford on) + DCNN fwd/bwd/Adam {t_step:.2f} s"
                  f"{crop_note}; torch CPU, {cores} threads",
    }


def log(msg: str) -> None:
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--batch", type=int, default=128, help="frames per GPU")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="coif4-l14")
    ap.add_argument("--cpu-frames", type=int, default=4, help="CPU baseline sample (0 = skip)")
    ap.add_argument("--cpu-only", action="store_true", help="only run the CPU baseline leg")
    a = ap.parse_args()

    if a.cpu_only:
        print(json.dumps(cpu_baseline(a.workload, max(1, a.cpu_frames))), flush=True)
        return
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus > 1 and world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} needs torch.distributed.run with {a.gpus} ranks (WORLD_SIZE={world})")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # AFD_FORCE_DDP=1: take the data-parallel path (RCCL process group, replica broadcast, SyncBN and
    # gradient all-reduce) even with one rank -- lets a 1-GPU box exercise the collectives
    ddp = world > 1 or bool(os.environ.get("AFD_FORCE_DDP"))
    if ddp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("LOCAL_RANK", "0")
        dist.init_process_group(backend="nccl", device_id=device)

    from audiofakedetect import _native

    _native.load()
    torch.manual_seed(0)
    args, trainer = build(a.workload, a.batch, ddp, device)
    batch = synthetic_batch(a.batch, rank, device)
    trainer.model.train()

    def sync():
        if ddp:
            dist.barrier(device_ids=[local_rank])
        torch.cuda.synchronize()

    log(f"built {a.workload}: features {args.input_dim}, batch/GPU {a.batch}, world {world}")
    eval_only = a.workload.endswith("eval")
    correct = torch.zeros((), dtype=torch.float64, device=device)

    def step():
        if not eval_only:
            trainer._run_batch(0, batch)
            return
        # evaluation step (reference val_test_loop, train_classifier.py:365-497): features,
        # forward, argmax, compare with the binarised label; counts stay on the device
        with torch.no_grad():
            out = trainer.model(trainer._features(batch["audio"]))
            correct.add_((out.argmax(-1) == (batch["label"] != 0)).sum())

    if eval_only:
        trainer.model.eval()
    for i in range(a.warmup):
        step()
        torch.cuda.synchronize()
        log(f"warmup step {i} done")
    sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    if ddp:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()

    log(f"timed {a.steps} steps: {1e3 * elapsed / a.steps:.2f} ms/step")
    # ---- per-kernel-class timing of one more step (HIP events on the launch stream) ----
    kernels = {}
    _native.timing_reset()
    _native.timing_enable(True)
    step()
    torch.cuda.synchronize()
    _native.timing_enable(False)
    for name in ("wpt", "conv_igemm", "conv_wgrad", "stft", "conv_direct", "conv_winograd", "conv_wgrad_1x1"):
        ms, n, work = _native.timing_collect(name)
        if n:
            kernels[name] = {"launches": n, "total_ms": ms, "avg_ms": ms / n, "work": work}
    _native.timing_reset()
    # the same once more with the backward-weight kernels on the main stream: launch durations
    # without another stream's kernels sharing the CUs (reported as roofline["serial"])
    serial = {}
    if not eval_only:
        prev = os.environ.get("AFD_WGRAD_STREAM")
        os.environ["AFD_WGRAD_STREAM"] = "0"
        try:
            _native.timing_enable(True)
            step()
            torch.cuda.synchronize()
            _native.timing_enable(False)
            for name in kernels:
                ms, n, work = _native.timing_collect(name)
                if n:
                    serial[name] = {"launches": n, "total_ms": ms, "avg_ms": ms / n, "work": work}
        finally:
            _native.timing_reset()
            if prev is None:
                os.environ.pop("AFD_WGRAD_STREAM", None)
            else:
                os.environ["AFD_WGRAD_STREAM"] = prev
    step_ms = 1e3 * elapsed / a.steps
    roofline = None
    if kernels:
        # the roofline object is about an HBM- or MFMA-bound class (conv_direct is VALU work)
        # launch durations of the pass without a second stream (a kernel's own time) where there
        # is one; the two-stream step's figures go into roofline["two_streams"]
        own = serial if serial else kernels
        dom = max((k for k in own if k != "conv_direct"), key=lambda k: own[k]["total_ms"])
        k = own[dom]
        if dom in ("wpt", "stft"):
            ach = k["work"] / (k["total_ms"] * 1e-3) / 1e9
            roofline = {"kernel": dom, "bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS,
                        "unit": "GB/s", "frac": ach / PEAK_HBM_GBS, "traffic": None}
        else:
            ach = k["work"] / (k["total_ms"] * 1e-3) / 1e12
            roofline = {"kernel": dom, "bound": "mfma", "achieved": ach,
                        "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                        "frac": ach / PEAK_F32_MFMA_TFLOPS, "traffic": None}
            if dom == "conv_winograd":
                # `achieved` counts the layer's direct-form flops (the algorithmic figure of the
                # contract); the F(2x2,3x3) kernel issues 16 of every 36 of them on the matrix cores
                roofline["mfma_issued"] = ach * 16.0 / 36.0
                roofline["mfma_issued_frac"] = ach * 16.0 / 36.0 / PEAK_F32_MFMA_TFLOPS
                roofline["note"] = ("achieved = direct-form flops / time; Winograd F(2x2,3x3) issues 16/36 "
                                    "of them as exact-fp32 MFMAs (mfma_issued*)")
        # HBM bytes per launch from the PMC passes (rocprofv3 cannot run inside this process):
        # read back from the committed counter summary when it covers this workload
        try:
            with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as fh:
                pmc = json.load(fh)
            if pmc.get("workload") == a.workload and pmc.get("batch") == a.batch:
                # the class's launches are spread over these kernels: launch-weighted mean
                names = {"conv_igemm": ("conv3x3_kernel", "conv1x1_kernel", "conv_igemm_kernel"),
                         "conv_winograd": ("wino_conv_kernel", "wino16_conv_kernel"),
                         "conv_wgrad": ("wgrad3x3_kernel", "conv_wgrad2_kernel", "conv_wgrad_kernel"),
                         "conv_wgrad_1x1": ("conv1x1_wgrad_kernel",),
                         "wpt": ("wpt2_deep_kernel", "wpt2_top_kernel"), "stft": ("stft_mfma_kernel",)}[dom]
                tot = cnt = 0.0
                for nm in names:
                    kk = pmc["kernels"].get(nm)
                    if kk:
                        tot += (kk.get("fetch_bytes_per_launch", 0.0) + kk.get("write_bytes_per_launch", 0.0)) * kk["launches_in_trace"]
                        cnt += kk["launches_in_trace"]
                if cnt:
                    roofline["traffic"] = tot / cnt
                    roofline["traffic_source"] = ("profiles/r01_pmc_traffic.json (FETCH_SIZE x2 for dwordx4 readers "
                                                  "+ WRITE_SIZE, launch-weighted over the class's kernels)")
        except (OSError, ValueError, KeyError):
            pass
        if serial and dom in kernels:
            ko = kernels[dom]
            div = 1e9 if dom in ("wpt", "stft") else 1e12
            oa = ko["work"] / (ko["total_ms"] * 1e-3) / div
            roofline["timing"] = ("launch durations from a step with the backward-weight stream off "
                                  "(AFD_WGRAD_STREAM=0): no other stream's kernels share the CUs")
            roofline["two_streams"] = {"achieved": oa, "frac": oa / roofline["peak"], "avg_launch_ms": ko["avg_ms"],
                                       "note": "the same launches inside the normal step, where backward-weight "
                                               "kernels run on a second stream next to the main stream's kernels"}
        roofline["launches_per_step"] = k["launches"]
        roofline["avg_launch_ms"] = k["avg_ms"]
        roofline["share_of_step"] = k["total_ms"] / step_ms
    frontend = None
    if "wpt" in kernels:
        k = kernels["wpt"]
        gbs = k["work"] / (k["total_ms"] * 1e-3) / 1e9
        frontend = {"kernel": "wpt", "avg_launch_ms": k["avg_ms"], "achieved_GBps": gbs,
                    "frac_of_hbm_peak": gbs / PEAK_HBM_GBS}

    cpu = None
    log(f"kernel classes: { {k: round(v['total_ms'], 3) for k, v in kernels.items()} }")
    if rank == 0 and world == 1 and a.cpu_frames > 0 and not eval_only:
        cpu = cpu_baseline(a.workload, a.cpu_frames)
        log(f"cpu baseline: {cpu['value']:.4f} frames/s")

    if rank == 0:
        loss = trainer.loss_list[-1][2] if trainer.loss_list else None
        line = {
            "metric": METRIC, "value": world * a.batch * a.steps / elapsed, "unit": "frames/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": step_ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": WORKLOADS[a.workload][4], "batch_per_gpu": a.batch,
                       "global_batch": a.batch * world, "frame": "1s@22050Hz mono f32",
                       "features": list(args.input_dim[1:]), "flattend_size": args.flattend_size,
                       "optimizer": "Adam lr 4e-4 wd 1e-3", "parallelism": f"dp{world}"},
            "roofline": roofline, "cpu_baseline": cpu, "frontend": frontend,
            "kernels": kernels, "kernels_serial": serial, "last_loss": loss,
        }
        print(json.dumps(line), flush=True)
    if ddp:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
