"""Timing of the dilated stack's launches at the level-8 geometry (development): forward, backward-data, backward-weight of
the three layers through the C ABI, B = 128, C = 13.  AFD_LIB selects the build."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "audiodeepfake-detection_amd"))
import torch
from audiofakedetect import _native
lib = _native.load()
P, S = _native.ptr, _native.stream_ptr
n, c = 128, 13
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (h, w, k, pad, dil) in ((64, 32, 3, 1, 1), (64, 32, 5, 2, 2), (60, 28, 7, 2, 4)):
    ho, wo = h + 2 * pad - dil * (k - 1), w + 2 * pad - dil * (k - 1)
    x = torch.randn(n, c, h, w, device="cuda"); wt = torch.randn(c, c, k, k, device="cuda"); b = torch.randn(c, device="cuda")
    y = torch.empty(n, c, ho, wo, device="cuda"); dy = torch.randn_like(y); dx = torch.empty_like(x)
    dw = torch.empty_like(wt); db = torch.empty_like(b)
    ws = torch.empty(lib.afd_conv2d_workspace_bytes(n, c, h, w, c, k, pad, dil), dtype=torch.uint8, device="cuda")
    f = t(lambda: _native.check(lib.afd_conv2d_forward(P(x), P(wt), P(b), P(y), n, c, h, w, c, k, pad, dil, P(ws), ws.numel(), S()), "f"))
    d = t(lambda: _native.check(lib.afd_conv2d_backward_data(P(dy), P(wt), P(dx), n, c, h, w, c, k, pad, dil, P(ws), ws.numel(), S()), "d"))
    g = t(lambda: _native.check(lib.afd_conv2d_backward_weight(P(x), P(dy), P(dw), P(db), n, c, h, w, c, k, pad, dil, P(ws), ws.numel(), S()), "w"))
    print(f"k{k} d{dil}: forward {f:6.1f} us  backward-data {d:6.1f} us  backward-weight {g:6.1f} us", flush=True)
