"""Fuzz of afd_conv3x3_input_grad_sums: sum over batch and pixels of g = w^T * dy per input channel, computed from the
weights and the border sums of dy, against the float64 sum of the convolution's input gradient (torch, CPU), on random
geometries -- dense dy with a live crop, and the pooled form (pooled gradient + codes).  Evidence tool:
    python3 tools/gradsum_fuzz.py [cases] > profiles/rNN_gradsum_fuzz.txt
"""
import random
import sys

sys.path.insert(0, "audiodeepfake-detection_amd")
import torch
from audiofakedetect import _native

lib = _native.load()
P, S = _native.ptr, _native.stream_ptr
random.seed(5)
torch.manual_seed(5)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
bad = 0
worst = 0.0
for it in range(cases):
    n = random.choice([1, 2, 3])
    cin, cout = random.choice([(3, 5), (16, 8), (32, 64), (13, 13), (64, 96)])
    h = random.choice([2, 3, 4, 5, 6, 9, 13])
    w = random.choice([2, 3, 8, 17, 64, 65, 129])
    pooled = random.random() < 0.4 and h >= 2 and w >= 2
    wt = torch.randn(cout, cin, 3, 3, device="cuda")
    if pooled:
        hp, wp = h // 2, w // 2
        gg = torch.randn(n, cout, hp, wp, device="cuda")
        codes = torch.randint(0, 8, (n, cout, hp, wp), device="cuda", dtype=torch.uint8)
        dense = torch.zeros(n, cout, h, w, device="cuda")
        yy = 2 * torch.arange(hp, device="cuda").view(1, 1, hp, 1) + ((codes >> 1) & 1)
        xx = 2 * torch.arange(wp, device="cuda").view(1, 1, 1, wp) + (codes & 1)
        nn_ = torch.arange(n, device="cuda").view(n, 1, 1, 1).expand_as(gg)
        cc = torch.arange(cout, device="cuda").view(1, cout, 1, 1).expand_as(gg)
        dense[nn_, cc, yy.expand_as(gg).long(), xx.expand_as(gg).long()] = gg
        dy, cd, rows, cols = gg, codes, h, w
    else:
        rows = random.choice([h, h, max(1, h - 1)])
        cols = random.choice([w, w, max(1, w - 1)])
        dense = torch.randn(n, cout, h, w, device="cuda")
        dense[:, :, rows:, :] = 0
        dense[:, :, :, cols:] = 0
        dy, cd = dense, None
    total = dense.double().sum((0, 2, 3))
    use_double = random.random() < 0.5
    sums = torch.zeros(cin + 8 * cout, dtype=torch.float64, device="cuda")
    _native.check(lib.afd_conv3x3_input_grad_sums(
        P(dy), P(cd), P(wt), P(total) if use_double else None, None if use_double else P(total.float()), P(sums), n, cin, h, w,
        cout, rows, cols, S()), "sums")
    g = torch.nn.grad.conv2d_input((n, cin, h, w), wt.double().cpu(), dense.double().cpu(), padding=1)
    ref = g.sum((0, 2, 3))
    scale = g.abs().sum((0, 2, 3)).max().item() + 1e-30
    err = (sums[:cin].cpu() - ref).abs().max().item() / scale
    worst = max(worst, err)
    ok = err <= 1e-7  # (the border sums are float partials per workgroup, added in double)
    bad += 0 if ok else 1
    print(f"{'ok ' if ok else 'BAD'} n{n} {cin}->{cout} {h}x{w} {'pooled' if pooled else f'crop {rows}x{cols}'} "
          f"{'double' if use_double else 'float'} totals: {err:.2e}", flush=True)
print(f"{cases} cases, {bad} bad, worst error {worst:.2e} of the largest channel's sum of |g|")
