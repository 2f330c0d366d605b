"""Timing of the 3x3 backward-data / forward entry points on level-14 block shapes (development)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "audiodeepfake-detection_amd"))
import torch
from audiofakedetect import _native
lib = _native.load()
def run_dgrad(n, cin, h, w, cout, reps=3):
    dy = torch.randn(n, cout, h, w, device="cuda"); wt = torch.randn(cout, cin, 3, 3, device="cuda")
    dx = torch.empty(n, cin, h, w, device="cuda")
    ws = torch.empty(lib.afd_conv2d_workspace_bytes(n, cin, h, w, cout, 3, 1, 1), dtype=torch.uint8, device="cuda")
    def go():
        _native.check(lib.afd_conv2d_backward_data(_native.ptr(dy), _native.ptr(wt), _native.ptr(dx), n, cin, h, w, cout, 3, 1, 1, _native.ptr(ws), ws.numel(), _native.stream_ptr()), "d")
    go(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); go(); e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1))
    return best
def run_fwd(n, cin, h, w, cout, reps=3):
    x = torch.randn(n, cin, h, w, device="cuda"); wt = torch.randn(cout, cin, 3, 3, device="cuda"); b = torch.randn(cout, device="cuda")
    y = torch.empty(n, cout, h, w, device="cuda")
    ws = torch.empty(lib.afd_conv2d_workspace_bytes(n, cin, h, w, cout, 3, 1, 1), dtype=torch.uint8, device="cuda")
    def go():
        _native.check(lib.afd_conv2d_forward(_native.ptr(x), _native.ptr(wt), _native.ptr(b), _native.ptr(y), n, cin, h, w, cout, 3, 1, 1, _native.ptr(ws), ws.numel(), _native.stream_ptr()), "f")
    go(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); go(); e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1))
    return best
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
for env in ({}, {"AFD_WINO44_UALIAS": "1"}, {}):
    os.environ.update(env)
    print(env or "normal", "block3 dgrad (dy 96ch -> dx 64ch, 13x8193): %.3f ms" % run_dgrad(n, 64, 13, 8193, 96),
          "| block4 fwd (96 -> 128, 6x4096): %.3f ms" % run_fwd(n, 96, 6, 4096, 128),
          "| block4 dgrad (128 -> 96): %.3f ms" % run_dgrad(n, 96, 6, 4096, 128), flush=True)
    for k in env: os.environ.pop(k)
