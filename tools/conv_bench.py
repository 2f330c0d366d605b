"""Conv micro-benchmark (development helper): times fwd / dgrad / wgrad of DCNN layer shapes."""
import sys, time
sys.path.insert(0, "audiodeepfake-detection_amd")
import torch
from audiofakedetect import ops, _native

SHAPES = {
    # name: n, cin, h, w, cout, k, pad, dil
    "conv1": (32, 1, 24, 16384, 64, 3, 2, 1),
    "conv2": (32, 64, 13, 8193, 64, 1, 0, 1),
    "conv3": (32, 64, 13, 8193, 96, 3, 1, 1),
    "conv3p": (32, 64, 12, 8192, 96, 3, 1, 1),
    "conv4": (32, 96, 6, 4096, 128, 3, 1, 1),
    "conv5": (32, 128, 6, 4096, 32, 3, 1, 1),
    "conv6": (32, 32, 6, 4096, 64, 3, 1, 1),
    "l8conv3": (128, 64, 51, 129, 96, 3, 1, 1),
    "l8conv4": (128, 96, 25, 64, 128, 3, 1, 1),
}
which = sys.argv[1:] or list(SHAPES)
iters = 5
for name in which:
    n, cin, h, w, cout, k, pad, dil = SHAPES[name]
    x = torch.randn(n, cin, h, w, device="cuda", requires_grad=True)
    wt = (torch.randn(cout, cin, k, k, device="cuda") / (cin * k * k) ** 0.5).requires_grad_()
    b = torch.zeros(cout, device="cuda", requires_grad=True)
    y = ops.conv2d(x, wt, b, pad, dil)
    dy = torch.randn_like(y)
    flops = 2.0 * n * cout * y.shape[2] * y.shape[3] * cin * k * k
    lib = _native.load()
    geom = (n, cin, h, w, cout, k, pad, dil)
    ws = ops._ws(lib.afd_conv2d_workspace_bytes(*geom), x.device)
    dx = torch.empty_like(x); dw = torch.empty_like(wt); db = torch.empty_like(b)
    def fwd(): lib.afd_conv2d_forward(_native.ptr(x), _native.ptr(wt), _native.ptr(b), _native.ptr(y), *geom, _native.ptr(ws), ws.numel(), _native.stream_ptr())
    def dgrad(): lib.afd_conv2d_backward_data(_native.ptr(dy), _native.ptr(wt), _native.ptr(dx), *geom, _native.ptr(ws), ws.numel(), _native.stream_ptr())
    def wgrad(): lib.afd_conv2d_backward_weight(_native.ptr(x), _native.ptr(dy), _native.ptr(dw), _native.ptr(db), *geom, _native.ptr(ws), ws.numel(), _native.stream_ptr())
    res = []
    for fn in (fwd, dgrad, wgrad):
        if fn is dgrad and cin == 1:
            res.append(float("nan")); continue
        fn(); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / iters)
    print(f"{name:8s} {flops/1e9:8.1f} GFLOP  fwd {res[0]:7.3f} ms {flops/res[0]/1e9:6.1f} TF/s | dgrad {res[1]:7.3f} ms {flops/res[1]/1e9:6.1f} TF/s | wgrad {res[2]:7.3f} ms {flops/res[2]/1e9:6.1f} TF/s", flush=True)
