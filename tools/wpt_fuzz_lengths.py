"""Randomised parity sweep of the packet front end over FILTER LENGTHS and FRAME LENGTHS (round 6: the top kernel takes
every even length up to 64 taps, the generic one up to 128): wavelets from every pywt family incl. the biorthogonal ones,
frame lengths from 300 to 30 000 samples (odd and even), levels 1..8 (as deep as the reflect pad allows), batches that
are and are not multiples of 8; raw coefficients against the float64 oracle at the bar of tests/test_wpt_gpu.py
(5e-6 of the largest coefficient).
    python3 tools/wpt_fuzz_lengths.py [cases]"""
import os, random, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "audiodeepfake-detection_amd"))
import torch
from audiofakedetect import wavelets
from audiofakedetect.wavelet_math import Packets
from oracle import wpt_oracle

random.seed(23); rng = np.random.default_rng(23)
NAMES = (["haar", "db2", "db3", "db5", "db6", "db7", "db9", "db11", "db13", "db16", "db20", "db32", "db38", "sym2", "sym4", "sym7",
          "sym8", "sym9", "sym13", "sym17", "sym20", "coif1", "coif2", "coif3", "coif5", "coif6", "coif7", "coif8", "coif9",
          "coif10", "coif12", "coif17", "dmey", "bior1.3", "bior2.8", "bior3.9", "bior6.8", "rbio2.6", "rbio5.5"])
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
worst = 0.0; bad = 0
for it in range(cases):
    name = random.choice(NAMES)
    w = wavelets.Wavelet(name)
    n = random.choice([300, 511, 1000, 2205, 4097, 8000, 11025, 16000, 22050, 22051, 24000, 26460])  # (the tree of a frame is LDS-resident: up to about 27 000 samples)
    lens = [n]
    for _ in range(8):
        if w.dec_len - 2 + (lens[-1] & 1) >= lens[-1]: break
        lens.append(wavelets.node_length(lens[-1], w.dec_len))
    max_level = len(lens) - 1
    if max_level < 1: continue
    level = random.randint(1, max_level)
    b = random.choice([1, 2, 3, 5, 8, 9, 16])
    x = np.clip(0.2 * rng.standard_normal((b, n)), -1, 1)
    if random.random() < 0.3:
        x[:, 0] = 1.0; x[:, -1] = -1.0   # both borders
    xt = torch.tensor(x, dtype=torch.float32)
    ref = wpt_oracle.wpt_nodes(xt.double().numpy(), w.dec_lo, level, w.dec_hi)
    got, _ = Packets(name, max_lev=level)(xt.cuda())
    got = got[:, 0].cpu().double().numpy()
    err = np.abs(got - ref).max() / (np.abs(ref).max() + 1e-300)
    ok = got.shape == ref.shape and err <= 5e-6
    worst = max(worst, err); bad += (not ok)
    print(f"{'ok ' if ok else 'BAD'} {name:8s} L={w.dec_len:3d} N={n:5d} level {level} B={b:2d}: {err:.2e}", flush=True)
print(f"{cases} cases, {bad} bad, worst error {worst:.2e} of the largest coefficient (bar 5e-6)")
