import sys, torch
sys.path.insert(0, "audiodeepfake-detection_amd")
from audiofakedetect import _native
lib = _native.load()
def run(cin, cout, n=128, h=13, w=8193, reps=6, stats=True):
    dy = torch.randn(n, cout, h, w, device="cuda"); wt = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
    dx = torch.empty(n, cin, h, w, device="cuda"); xhat = torch.randn(n, cin, h, w, device="cuda")
    sums = torch.empty(2 * cin, dtype=torch.float64, device="cuda")
    ws = torch.empty(lib.afd_conv2d_workspace_bytes(n, cin, h, w, cout, 3, 1, 1), dtype=torch.uint8, device="cuda")
    sws = torch.empty(lib.afd_conv3x3_backward_data_bnstats_workspace_bytes(n, cin, h, w), dtype=torch.uint8, device="cuda")
    def go():
        if stats:
            _native.check(lib.afd_conv3x3_backward_data_bnstats(_native.ptr(dy), _native.ptr(wt), _native.ptr(dx), _native.ptr(xhat), _native.ptr(sums), n, cin, h, w, cout, _native.ptr(ws), ws.numel(), _native.ptr(sws), sws.numel(), _native.stream_ptr()), "d")
        else:
            _native.check(lib.afd_conv2d_backward_data(_native.ptr(dy), _native.ptr(wt), _native.ptr(dx), n, cin, h, w, cout, 3, 1, 1, _native.ptr(ws), ws.numel(), _native.stream_ptr()), "dgrad")
    for _ in range(3): go()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): go()
    e1.record(); torch.cuda.synchronize()
    print(f"dgrad cin {cin} cout {cout} h {h} w {w} stats {stats}: {e0.elapsed_time(e1) / reps:.3f} ms", flush=True)
    ref = torch.nn.grad.conv2d_input((1, cin, h, w), wt.double().cpu(), dy[:1].double().cpu(), padding=1)
    err = (dx[:1].double().cpu() - ref).abs().max().item() / ref.abs().max().item()
    print("   max err / max", err, flush=True)
    if stats:
        r = torch.cat([dx.double().sum((0, 2, 3)), (dx.double() * xhat.double()).sum((0, 2, 3))])
        scale = (dx.double().abs() * xhat.double().abs()).sum((0, 2, 3)).max().item()
        print("   sums err / scale", (sums - r).abs().max().item() / scale, flush=True)
run(64, 96); run(64, 96, stats=False); run(64, 96, n=2, h=7, w=1157); run(64, 96, n=2, h=4, w=259)
