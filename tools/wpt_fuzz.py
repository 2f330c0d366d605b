"""Randomised parity sweep of the packet front end against the float64 oracle (raw coefficients, 5e-6 of the largest
coefficient -- the bar of tests/test_wpt_gpu.py) and, for the log / sign / normalised modes, against the same formulas
applied to the oracle's coefficients.  Batch sizes include non-multiples of 8 (the top kernel's other block placement),
inputs include impulses at both borders, constants, ramps, full-scale noise and very small signals.
    python3 tools/wpt_fuzz.py [cases]"""
import os, random, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "audiodeepfake-detection_amd"))
import torch
from audiofakedetect import wavelets
from audiofakedetect.wavelet_math import Packets
from oracle import wpt_oracle

random.seed(11); rng = np.random.default_rng(11)
N = 22050
def signal(kind, b):
    if kind == "noise": return np.clip(0.1 * rng.standard_normal((b, N)), -1, 1)
    if kind == "full": return rng.uniform(-1, 1, (b, N))
    if kind == "tiny": return 1e-6 * rng.standard_normal((b, N))
    if kind == "const": return np.full((b, N), 0.37) * rng.uniform(-1, 1, (b, 1))
    if kind == "ramp": return np.linspace(-1, 1, N)[None, :] * rng.uniform(0.1, 1, (b, 1))
    x = np.zeros((b, N))
    for i in range(b):
        x[i, random.choice([0, 1, 2, 11024, 11025, N - 3, N - 2, N - 1, random.randrange(N)])] = rng.uniform(-1, 1)
    return x
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
worst = 0.0; bad = 0
for it in range(cases):
    name = random.choice(["coif4", "sym5", "db8"])
    level = random.choice([8, 14, 14])
    b = random.choice([1, 2, 3, 5, 8, 9, 16, 24])
    kind = random.choice(["noise", "full", "tiny", "const", "ramp", "impulse"])
    mode = random.choice(["raw", "raw", "log", "sign"])
    x = signal(kind, b)
    w = wavelets.Wavelet(name)
    ref = wpt_oracle.wpt_nodes(x, w.dec_lo, level)  # [b, P, T]
    xt = torch.tensor(x, dtype=torch.float32)
    ref32 = wpt_oracle.wpt_nodes(xt.double().numpy(), w.dec_lo, level)
    p = Packets(name, max_lev=level, log_scale=mode != "raw", loss_less=mode == "sign")
    got, _ = p(xt.cuda())
    got = got.cpu().double().numpy()  # [b, C, P, T]
    scale = np.abs(ref32).max() + 1e-300
    if mode == "raw":
        err = np.abs(got[:, 0] - ref32).max() / scale
        ok = err <= 5e-6
    else:
        # log(|c|^2 + 1e-12): compare where the coefficient is well above the float32 noise of the transform
        lr = np.log(ref32 ** 2 + 1e-12)
        mask = np.abs(ref32) > 1e-3 * scale
        err = np.abs(got[:, 0] - lr)[mask].max() if mask.any() else 0.0
        ok = err <= 2e-2
        if mode == "sign":
            sg = np.where(ref32 < 0, -1.0, 1.0)
            big = np.abs(ref32) > 1e-4 * scale
            ok = ok and bool((got[:, 1][big] == sg[big]).all())
    worst = max(worst, err if mode == "raw" else 0.0)
    bad += 0 if ok else 1
    print(f"{'ok ' if ok else 'BAD'} {name} level {level} B={b} {kind} {mode}: {err:.2e}", flush=True)
print(f"{cases} cases, {bad} bad, worst raw error {worst:.2e} of the largest coefficient (bar 5e-6)")
sys.exit(1 if bad else 0)
