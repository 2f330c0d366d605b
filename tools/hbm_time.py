"""HIP-event timings of the HBM-bound layers of the level-14 step at their real shapes (B = 128):
    python3 tools/hbm_time.py [conv1|pool|all]
Environment switches of the kernels (e.g. AFD_C1B_WGS) are read once per process: one variant per run."""
import os, sys, torch
sys.path.insert(0, "audiodeepfake-detection_amd")
from audiofakedetect import ops

def timed(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

what = sys.argv[1] if len(sys.argv) > 1 else "all"
dev = "cuda"
if what in ("conv1", "all"):
    x = torch.randn(128, 1, 24, 16384, device=dev)
    w = (torch.randn(64, 1, 3, 3, device=dev) * 0.3).requires_grad_()
    b = torch.zeros(64, device=dev, requires_grad=True)
    a = torch.tensor([0.25], device=dev, requires_grad=True)
    u = ops.conv1_prelu_maxpool(x, w, b, a, 2)
    du = torch.randn_like(u)
    print("conv1 fwd %.3f ms" % timed(lambda: ops.conv1_prelu_maxpool(x, w, b, a, 2)), flush=True)
    print("conv1 bwd %.3f ms (AFD_C1B_WGS=%s)" % (timed(lambda: torch.autograd.grad(u, (w, b, a), du, retain_graph=True)),
                                                  os.environ.get("AFD_C1B_WGS")), flush=True)
    del x, u, du
if what in ("pool", "all"):
    for shape in [(128, 96, 13, 8193), (128, 64, 6, 4096)]:
        z = torch.randn(shape, device=dev, requires_grad=True)
        a = torch.tensor([0.25], device=dev, requires_grad=True)
        u = ops.prelu_maxpool2x2(z, a)
        du = torch.randn_like(u)
        ms = timed(lambda: torch.autograd.grad(u, (z, a), du, retain_graph=True))
        gb = (z.numel() * 4 + u.numel() * 9) / 1e9
        print(shape, "pool bwd %.3f ms %.0f GB/s" % (ms, gb / ms * 1e3), flush=True)
        del z, u, du
if what in ("dil", "all"):
    # the dilated stack of the level-14 model: [128, 3, 64, 2048], (k, pad, dil) = (3,1,1), (5,2,2), (7,2,4)
    h = torch.randn(128, 3, 64, 2048, device=dev)
    for k, pad, d in ((3, 1, 1), (5, 2, 2), (7, 2, 4)):
        x = h.clone().requires_grad_()
        w = (torch.randn(3, 3, k, k, device=dev) / (3 * k * k) ** 0.5).requires_grad_()
        b = torch.zeros(3, device=dev, requires_grad=True)
        y = ops.conv2d(x, w, b, pad, d)
        dy = torch.randn_like(y)
        f = timed(lambda: ops.conv2d(x, w, b, pad, d))
        bw = timed(lambda: torch.autograd.grad(y, (x, w, b), dy, retain_graph=True))
        print("dilated k=%d: fwd %.3f ms, bwd (data + weight) %.3f ms" % (k, f, bw), flush=True)
        h = y.detach()
