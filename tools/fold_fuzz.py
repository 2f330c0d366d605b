"""Fuzz of the input fold (afd_conv3x3_forward_fold / afd_conv3x3_backward_weight_fold) against the two-pass chain
afd_bn_apply_forward -> afd_conv3x3_forward_stats / afd_conv3x3_prelu_pool_forward / backward-weight entry points, through
the C ABI on random geometries: results must be EQUAL (the fold uses bn_apply's arithmetic).  Development / evidence tool:
    python3 tools/fold_fuzz.py [cases] > profiles/rNN_fold_fuzz.txt
"""
import random
import sys

sys.path.insert(0, "audiodeepfake-detection_amd")
import torch
from audiofakedetect import _native, ops

lib = _native.load()
P, S = _native.ptr, _native.stream_ptr
random.seed(11)
torch.manual_seed(11)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
bad = done = 0
FLAVOURS = [  # cin, cout, pooled, statistics
    (64, 96, 1, 1), (96, 128, 0, 1), (128, 32, 0, 1), (32, 64, 1, 0), (64, 64, 1, 0), (32, 96, 1, 1), (64, 128, 0, 1)]
while done < cases:
    cin, cout, pooled, stats = random.choice(FLAVOURS)
    n = random.choice([1, 2, 3])
    h = random.choice([3, 4, 5, 6, 7, 9, 12, 13, 17, 25])
    w = random.choice([48, 64, 65, 128, 129, 256, 257, 320, 513, 1024, 1025, 1100])
    if not lib.afd_conv3x3_input_fold_applicable(cin, h, w, cout, pooled, stats):
        continue
    if pooled and not stats and not lib.afd_conv3x3_prelu_pool_applicable(cin, h, w, cout):
        continue
    done += 1
    slope_v = random.choice([None, 0.25, 0.25, 1.3, -0.4, 0.0, 1.0])
    x = torch.randn(n, cin, h, w, device="cuda") * random.choice([0.5, 1.0, 3.0]) + random.choice([0.0, 0.7, -1.2])
    wt = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
    b = torch.randn(cout, device="cuda")
    a_out = torch.full((1,), 0.25, device="cuda")
    a_in = None if slope_v is None else torch.full((1,), slope_v, device="cuda")
    v = x if a_in is None else torch.where(x > 0, x, slope_v * x)
    mean = v.mean((0, 2, 3)).contiguous()
    invstd = torch.rsqrt(v.var((0, 2, 3), unbiased=False) + 1e-5).contiguous()
    aff = torch.stack((mean, invstd), 1).contiguous()
    xhat = torch.empty_like(x)
    _native.check(lib.afd_bn_apply_forward(P(x), P(a_in), P(mean), P(invstd), None, None, P(xhat), n, cin, h * w, S()), "bn")
    ws = ops._ws(lib.afd_conv2d_workspace_bytes(n, cin, h, w, cout, 3, 1, 1), x.device)
    sws = ops._ws(lib.afd_conv3x3_forward_stats_workspace_bytes(n, h, w, cout), x.device, "fwdstats")
    outs = []
    for fold in (False, True):
        xin = x if fold else xhat
        y = torch.full((n, cout, h, w), float("nan"), device="cuda") if not pooled else None
        u = torch.full((n, cout, h // 2, w // 2), float("nan"), device="cuda") if pooled else None
        idx = ops._empty_with_slack((n, cout, h // 2, w // 2), torch.uint8, x.device).zero_() if pooled else None
        sums = torch.zeros(2 * cout + 1, dtype=torch.float64, device="cuda") if stats else None
        if fold:
            _native.check(lib.afd_conv3x3_forward_fold(P(xin), P(aff), P(a_in), P(wt), P(b), P(a_out), P(y), P(u), P(idx), P(sums),
                                                       n, cin, h, w, cout, P(ws), ws.numel(), P(sws), sws.numel(), S()), "ff")
        elif stats:
            _native.check(lib.afd_conv3x3_forward_stats(P(xin), P(wt), P(b), P(a_out), P(y), P(u), P(idx), P(sums), n, cin, h, w, cout,
                                                        P(ws), ws.numel(), P(sws), sws.numel(), S()), "fs")
        else:
            _native.check(lib.afd_conv3x3_prelu_pool_forward(P(xin), P(wt), P(b), P(a_out), P(u), P(idx), n, cin, h, w, cout, P(ws),
                                                             ws.numel(), S()), "fp")
        # backward-weight: dense dy on the crop, or the pooled gradient with the forward's codes
        torch.manual_seed(done)
        dw = torch.empty_like(wt)
        db = torch.empty(cout, device="cuda")
        if pooled:
            gg = ops._empty_with_slack((n, cout, h // 2, w // 2), torch.float32, x.device).normal_()
            if fold:
                _native.check(lib.afd_conv3x3_backward_weight_fold(P(xin), P(aff), P(a_in), P(gg), P(idx), P(dw), P(db), None, n, cin, h, w,
                                                                   cout, h, w, P(ws), ws.numel(), S()), "wf")
            elif lib.afd_conv3x3_pooled_backward_applicable(cin, h, w, cout):
                _native.check(lib.afd_conv3x3_backward_weight_pooled(P(xin), P(gg), P(idx), P(dw), P(db), n, cin, h, w, cout, P(ws),
                                                                     ws.numel(), S()), "wp")
            else:
                dw = db = None  # no unfolded launch of this form for these channel counts
        else:
            dy = torch.randn(n, cout, h, w, device="cuda")
            if fold:
                _native.check(lib.afd_conv3x3_backward_weight_fold(P(xin), P(aff), P(a_in), P(dy), None, P(dw), P(db), None, n, cin, h, w, cout,
                                                                   h, w, P(ws), ws.numel(), S()), "wf")
            else:
                _native.check(lib.afd_conv2d_backward_weight_sums(P(xin), P(dy), P(dw), P(db), None, n, cin, h, w, cout, 3, 1, 1, h, w,
                                                                  P(ws), ws.numel(), S()), "w")
        outs.append((y, u, idx, None if sums is None else sums[:2 * cout].clone(), dw, db))
    ok = True
    what = []
    for name, a0, a1 in zip(("y", "u", "codes", "sums", "dw", "db"), outs[0], outs[1]):
        if a0 is None or a1 is None:
            continue
        same = torch.equal(a0, a1) and bool(torch.isfinite(a1.double()).all())
        if not same:
            ok = False
            what.append(f"{name} {(a0.double() - a1.double()).abs().max().item():.2e}")
    bad += 0 if ok else 1
    print(f"{'ok ' if ok else 'BAD'} n{n} {cin}->{cout} {h}x{w} pooled={pooled} stats={stats} slope={slope_v} {' '.join(what)}", flush=True)
print("BAD COUNT", bad, "of", done)
