"""Step time of the headline workload under different stream arrangements (task 4).

  serial      every kernel on the caller's stream (AFD_WGRAD_STREAM=0)
  two-stream  backward-weight kernels on a second, unmasked stream (round 1-2 default)
  masked a/b  the same two streams created with hipExtStreamCreateWithCUMask: the main chain on `a` CUs, the
              backward-weight stream on `b` (disjoint masks; bit i of the mask = CU i of the agent's enumeration,
              dealt here XCD-interleaved so that both partitions span all eight XCDs)

Usage (GPU box):  python tools/stream_modes.py [--workload coif4-l14] [--steps 8]
Prints one JSON object with ms/step per arrangement.
"""

import argparse
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "audiodeepfake-detection_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

import bench  # noqa: E402


def masked_stream(hip, cus, total=256):
    """A stream confined to the CUs in `cus`."""
    words = (total + 31) // 32
    mask = (ctypes.c_uint32 * words)()
    for c in cus:
        mask[c // 32] |= 1 << (c % 32)
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), words, mask)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask -> {rc}")
    return torch.cuda.ExternalStream(st.value)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="coif4-l14")
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=4)
    a = ap.parse_args()
    from audiofakedetect import _native, ops

    _native.load()
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    args, trainer, _ = bench.build(a.workload, 128, False, dev)
    batch = bench.synthetic_batch(128, 0, dev)
    trainer.model.train()
    hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))

    def timed(stream=None):
        ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream())
        with ctx:
            for _ in range(a.warmup):
                trainer._run_batch(0, batch)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.steps):
                trainer._run_batch(0, batch)
            e1.record()
            torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.steps

    res = {}
    os.environ["AFD_WGRAD_STREAM"] = "0"
    res["serial"] = timed()
    os.environ["AFD_WGRAD_STREAM"] = "1"
    res["two-stream"] = timed()
    plain_side = ops._side_streams.get(0)
    for main_cus in (224, 192, 160, 128):
        # XCD-interleaved split: CU c belongs to the main partition when its slot within the XCD is below the cut
        per_xcd = main_cus // 8
        cu_main = [c for c in range(256) if (c // 8) < per_xcd]   # enumeration A: consecutive CUs share an XCD slot row
        cu_side = [c for c in range(256) if (c // 8) >= per_xcd]
        try:
            ms = masked_stream(hip, cu_main)
            ss = masked_stream(hip, cu_side)
        except Exception as exc:  # noqa: BLE001
            res[f"masked {main_cus}/{256 - main_cus}"] = str(exc)
            continue
        ops._side_streams[0] = ss
        res[f"masked {main_cus}/{256 - main_cus}"] = timed(ms)
        # the same split with the roles' sizes kept but the side stream unmasked (only the main chain confined)
    if plain_side is not None:
        ops._side_streams[0] = plain_side
    # serial once more at the end (clock / thermal drift check)
    os.environ["AFD_WGRAD_STREAM"] = "0"
    res["serial (again)"] = timed()
    print(json.dumps({"workload": a.workload, "steps": a.steps, "ms_per_step": res}))


if __name__ == "__main__":
    main()
