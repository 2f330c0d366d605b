"""Timing of the 3x3 backward-weight entry point on the level-14 block-3 / block-4 shapes (development)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "audiodeepfake-detection_amd"))
import torch
from audiofakedetect import _native
lib = _native.load()
def run(n, cin, h, w, cout, rows, cols, reps=3):
    x = torch.randn(n, cin, h, w, device="cuda"); dy = torch.randn(n, cout, h, w, device="cuda")
    dw = torch.empty(cout, cin, 3, 3, device="cuda"); db = torch.empty(cout, device="cuda")
    ws = torch.empty(lib.afd_conv2d_workspace_bytes(n, cin, h, w, cout, 3, 1, 1), dtype=torch.uint8, device="cuda")
    def go():
        _native.check(lib.afd_conv2d_backward_weight_sums(_native.ptr(x), _native.ptr(dy), _native.ptr(dw), _native.ptr(db), None, n, cin, h, w, cout, 3, 1, 1, rows, cols, _native.ptr(ws), ws.numel(), _native.stream_ptr()), "w")
    go(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); go(); e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1))
    return best
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
for name, shp in (("block3", (n, 64, 13, 8193, 96, 12, 8192)), ("block4", (n, 96, 6, 4096, 128, 6, 4096))):
    for env in ({}, {"AFD_NO_WINO44_WGRAD": "1"}):
        os.environ.update(env)
        print(name, env or "winograd domain", "%.3f ms" % run(*shp), flush=True)
        for k in env:
            os.environ.pop(k)
