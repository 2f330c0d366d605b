import sys, torch
sys.path.insert(0, "audiodeepfake-detection_amd")
from audiofakedetect import _native
lib = _native.load()
def timeit(go, reps=8):
    for _ in range(3): go()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): go()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
def fwd(cin, cout, n=128, h=6, w=4096):
    x = torch.randn(n, cin, h, w, device="cuda"); wt = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
    b = torch.randn(cout, device="cuda"); y = torch.empty(n, cout, h, w, device="cuda")
    ws = torch.empty(lib.afd_conv2d_workspace_bytes(n, cin, h, w, cout, 3, 1, 1), dtype=torch.uint8, device="cuda")
    ms = timeit(lambda: _native.check(lib.afd_conv2d_forward(_native.ptr(x), _native.ptr(wt), _native.ptr(b), _native.ptr(y), n, cin, h, w, cout, 3, 1, 1, _native.ptr(ws), ws.numel(), _native.stream_ptr()), "fwd"))
    ref = torch.nn.functional.conv2d(x[:1].double().cpu(), wt.double().cpu(), b.double().cpu(), padding=1)
    print(f"fwd {cin}->{cout}: {ms:.3f} ms  err {((y[:1].double().cpu()-ref).abs().max()/ref.abs().max()).item():.2e}", flush=True)
def dgrad(cin, cout, n=128, h=6, w=4096, stats=False):
    dy = torch.randn(n, cout, h, w, device="cuda"); wt = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
    dx = torch.empty(n, cin, h, w, device="cuda"); xhat = torch.randn(n, cin, h, w, device="cuda")
    sums = torch.empty(2 * cin, dtype=torch.float64, device="cuda")
    ws = torch.empty(lib.afd_conv2d_workspace_bytes(n, cin, h, w, cout, 3, 1, 1), dtype=torch.uint8, device="cuda")
    if stats and not lib.afd_conv3x3_backward_data_bnstats_applicable(cin, h, w, cout):
        print("  (stats variant not applicable)"); return
    sws = torch.empty(lib.afd_conv3x3_backward_data_bnstats_workspace_bytes(n, cin, h, w), dtype=torch.uint8, device="cuda")
    def go():
        if stats:
            _native.check(lib.afd_conv3x3_backward_data_bnstats(_native.ptr(dy), _native.ptr(wt), _native.ptr(dx), _native.ptr(xhat), _native.ptr(sums), n, cin, h, w, cout, _native.ptr(ws), ws.numel(), _native.ptr(sws), sws.numel(), _native.stream_ptr()), "d")
        else:
            _native.check(lib.afd_conv2d_backward_data(_native.ptr(dy), _native.ptr(wt), _native.ptr(dx), n, cin, h, w, cout, 3, 1, 1, _native.ptr(ws), ws.numel(), _native.stream_ptr()), "dgrad")
    ms = timeit(go)
    ref = torch.nn.grad.conv2d_input((1, cin, h, w), wt.double().cpu(), dy[:1].double().cpu(), padding=1)
    print(f"dgrad {cout}->{cin} stats {stats}: {ms:.3f} ms  err {((dx[:1].double().cpu()-ref).abs().max()/ref.abs().max()).item():.2e}", flush=True)
fwd(96, 128); dgrad(128, 32); dgrad(128, 32, stats=True)
