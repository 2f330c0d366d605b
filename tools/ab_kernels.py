"""A/B of two builds of the library on single launches of the level-14 step (development helper).

    python tools/ab_kernels.py lib_a.so lib_b.so

Each library runs in its own child process, alternating, three rounds; prints the median per launch in ms.
"""
import json
import os
import subprocess
import sys


def child(lib_path):
    sys.path.insert(0, "audiodeepfake-detection_amd")
    import torch
    from audiofakedetect import _native
    _native.LIB_PATH = os.path.abspath(lib_path)
    lib = _native.load()
    from audiofakedetect import ops
    P, S = _native.ptr, _native.stream_ptr
    n = 128
    res = {}

    def timeit(fn, reps=4):
        fn()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        return best

    # block 3: 64 -> 96 on 13 x 8193, pooled
    cin, h, w, cout = 64, 13, 8193, 96
    x = torch.randn(n, cin, h, w, device="cuda")
    wt = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
    gg = ops._empty_with_slack((n, cout, h // 2, w // 2), torch.float32, "cuda").normal_()
    idx = ops._empty_with_slack((n, cout, h // 2, w // 2), torch.uint8, "cuda").random_(0, 8)
    dx = torch.empty_like(x)
    sums = torch.empty(2 * cin, dtype=torch.float64, device="cuda")
    ws = ops._ws(lib.afd_conv2d_workspace_bytes(n, cin, h, w, cout, 3, 1, 1), x.device)
    sws = ops._ws(lib.afd_conv3x3_backward_data_bnstats_workspace_bytes(n, cin, h, w), x.device, "bnstats")
    res["b3 dgrad pooled"] = timeit(lambda: _native.check(lib.afd_conv3x3_backward_data_bnstats_pooled(
        P(gg), P(idx), P(wt), P(dx), P(sums), n, cin, h, w, cout, P(ws), ws.numel(), P(sws), sws.numel(), S()), "d"))
    dw = torch.empty_like(wt)
    db = torch.empty(cout, device="cuda")
    res["b3 wgrad pooled"] = timeit(lambda: _native.check(lib.afd_conv3x3_backward_weight_pooled(
        P(x), P(gg), P(idx), P(dw), P(db), n, cin, h, w, cout, P(ws), ws.numel(), S()), "w"))
    if hasattr(lib, "afd_conv3x3_backward_weight_fold"):
        aff = torch.stack((torch.zeros(cin, device="cuda"), torch.ones(cin, device="cuda")), 1).contiguous()
        a = torch.full((1,), 0.25, device="cuda")
        res["b3 wgrad pooled fold"] = timeit(lambda: _native.check(lib.afd_conv3x3_backward_weight_fold(
            P(x), P(aff), P(a), P(gg), P(idx), P(dw), P(db), None, n, cin, h, w, cout, h, w, P(ws), ws.numel(), S()), "wf"))
        res["b3 wgrad pooled fold, no PReLU"] = timeit(lambda: _native.check(lib.afd_conv3x3_backward_weight_fold(
            P(x), P(aff), None, P(gg), P(idx), P(dw), P(db), None, n, cin, h, w, cout, h, w, P(ws), ws.numel(), S()), "wf"))
        u = torch.empty(n, cout, h // 2, w // 2, device="cuda")
        b = torch.zeros(cout, device="cuda")
        fs = torch.empty(2 * cout + 1, dtype=torch.float64, device="cuda")
        fws = ops._ws(lib.afd_conv3x3_forward_stats_workspace_bytes(n, h, w, cout), x.device, "fwdstats")
        res["b3 fwd pool stats"] = timeit(lambda: _native.check(lib.afd_conv3x3_forward_stats(
            P(x), P(wt), P(b), P(a), None, P(u), P(idx), P(fs), n, cin, h, w, cout, P(ws), ws.numel(), P(fws), fws.numel(), S()), "f"))
        res["b3 fwd pool stats fold"] = timeit(lambda: _native.check(lib.afd_conv3x3_forward_fold(
            P(x), P(aff), P(a), P(wt), P(b), P(a), None, P(u), P(idx), P(fs), n, cin, h, w, cout, P(ws), ws.numel(), P(fws),
            fws.numel(), S()), "ff"))
    if hasattr(lib, "afd_conv3x3_forward_fold"):
        res["b3 fwd pool stats fold, no PReLU"] = timeit(lambda: _native.check(lib.afd_conv3x3_forward_fold(
            P(x), P(aff), None, P(wt), P(b), P(a), None, P(u), P(idx), P(fs), n, cin, h, w, cout, P(ws), ws.numel(), P(fws),
            fws.numel(), S()), "ff"))
    print(json.dumps(res))


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(sys.argv[2])
        sys.exit(0)
    libs = sys.argv[1:]
    runs = {lib: [] for lib in libs}
    for _ in range(3):
        for lib in libs:
            out = subprocess.check_output([sys.executable, __file__, "--child", lib], text=True)
            runs[lib].append(json.loads(out.strip().splitlines()[-1]))
    keys = []
    for lib in libs:
        for k in runs[lib][0]:
            if k not in keys:
                keys.append(k)
    for key in keys:
        row = []
        for lib in libs:
            v = sorted(r[key] for r in runs[lib] if key in r)
            row.append(f"{v[len(v) // 2]:9.3f} ms" if v else "        -   ")
        print(f"{key:26s} " + "  ".join(row), flush=True)
