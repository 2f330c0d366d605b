"""Orthogonal lattice factorisation of the analysis filter banks (coif4, sym5, db8, haar) and what it costs / how
accurate it is in float32 -- the zero-GPU study VERDICT round 3 item 1 asks for.

The two-channel analysis step of the packet transform (reference src/audiofakedetect/wavelet_math.py:182,192 -> ptwt:
cA[i] = sum_m dec_lo[m] xe[2i+1-m], cD likewise, xe = reflect extension) is, on the polyphase pairs
(e_j, o_j) = (xe[2j], xe[2j+1]),

    [cA; cD](z) = H_p(z) [e; o](z),      H_p(z) = sum_q z^-q [[lo[2q+1], lo[2q]], [hi[2q+1], hi[2q]]],   q < K = L/2,

and H_p is paraunitary for an orthogonal wavelet, so it factors into K plane rotations with one delay of the second
channel between them (Vaidyanathan):   H_p(z) = R_{K-1} D(z) R_{K-2} ... D(z) R_0,   D = diag(1, z^-1).
Direct form costs 2 L multiply-adds per output pair (cA[i], cD[i]); the lattice K rotations = 4 K = 2 L multiplies in
normalised form, or -- scaling deferred -- ONE packed FMA (2 multiply-adds) per rotation:

    (A', B') = (A + alpha_s B_prev, A + beta_s B_prev)          B_prev = the neighbour position's B (the delay)

i.e. L multiply-adds per output pair: half of the direct form, plus a halo of K (K-1) / 2 rotations per run of
consecutive outputs (the triangle of positions that only feed later stages).

This script
  1. factors the tap tables in 60-digit arithmetic (mpmath) and checks that the lattice reproduces the taps;
  2. runs the level-8 / level-14 transform through the lattice in float32 (numpy, same operation order as a kernel
     would use: one fused multiply-add per channel and stage, emulated as float32(float64 product-sum) and compares
     with oracle/wpt_oracle.py (float64 direct form) -- bar 5e-6 of the largest coefficient, tests/test_wpt_gpu.py:20;
     the float32 DIRECT form is run beside it for reference;
  3. counts multiply-adds per frame for direct form, the round-3 kernels (composite levels 10-14) and the lattice
     with whole-node runs (deep levels) / runs of R outputs (top levels).

usage: python tools/wpt_lattice.py [--json profiles/r04_lattice_study.json]
"""
from __future__ import annotations

import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import wpt_oracle  # noqa: E402  (a study tool: the oracle is the checker here, as in tests/)

DB8 = [-0.00011747678400228192, 0.0006754494059985568, -0.0003917403729959771, -0.00487035299301066,
       0.008746094047015655, 0.013981027917015516, -0.04408825393106472, -0.01736930100202211,
       0.128747426620186, 0.00047248457399797254, -0.2840155429624281, -0.015829105256023893,
       0.5853546836548691, 0.6756307362980128, 0.3128715909144659, 0.05441584224308161]


def factor(dec_lo, digits: int = 60):
    """Rotations R_0 .. R_{K-1} (2x2 orthogonal matrices, mpmath) with H_p = R_{K-1} D R_{K-2} ... D R_0."""
    import mpmath as mp

    mp.mp.dps = digits
    L = len(dec_lo)
    K = L // 2
    lo = [mp.mpf(v) for v in dec_lo]
    hi = [(-1) ** (k + 1) * lo[L - 1 - k] for k in range(L)]
    P = [mp.matrix([[lo[2 * q + 1], lo[2 * q]], [hi[2 * q + 1], hi[2 * q]]]) for q in range(K)]
    rots = []
    for m in range(K - 1, 0, -1):
        # R^T H^(m) = D H^(m-1): top row of R^T P_m vanishes, bottom row of R^T P_0 vanishes.
        # P_m has rank 1: take the angle from its larger column.
        col = 0 if abs(P[m][0, 0]) + abs(P[m][1, 0]) >= abs(P[m][0, 1]) + abs(P[m][1, 1]) else 1
        x, y = P[m][0, col], P[m][1, col]
        r = mp.sqrt(x * x + y * y)
        # R^T = [[c, s], [-s, c]] with c x + s y = 0  ->  (c, s) = (y, -x) / r
        c, s = y / r, -x / r
        RT = mp.matrix([[c, s], [-s, c]])
        Q = [RT * P[q] for q in range(m + 1)]
        # new coefficients: top rows keep their power, bottom rows shift down by one power
        P = [mp.matrix([[Q[q][0, 0], Q[q][0, 1]], [Q[q + 1][1, 0], Q[q + 1][1, 1]]]) for q in range(m)]
        rots.append(RT.T)
    rots.append(P[0])  # R_0: what is left (orthogonal up to the tap table's own rounding)
    rots.reverse()
    return rots


def polish(rots, dec_lo):
    """The peeling divides by the end taps (coif4: 1.8e-6), which amplifies the tap table's own rounding (the tables
    are orthogonal to 3e-13 only): the peeled lattice reproduces coif4's taps to 5e-7 only.  Least squares over the K
    angles (float64, started from the peeled ones) finds the exactly orthogonal bank nearest the table."""
    import mpmath as mp
    from scipy.optimize import least_squares

    K = len(rots)
    L = 2 * K
    lo = np.asarray(dec_lo, dtype=np.float64)
    hi = np.asarray([(-1) ** (k + 1) * lo[L - 1 - k] for k in range(L)])
    det0 = float(rots[0][0, 0] * rots[0][1, 1] - rots[0][0, 1] * rots[0][1, 0])
    th0 = np.array([float(mp.atan2(r[1, 0], r[0, 0])) for r in rots])

    def mats(th):
        ms = []
        for s in range(K):
            c, sn = np.cos(th[s]), np.sin(th[s])
            m = np.array([[c, -sn], [sn, c]])
            if s == 0 and det0 < 0:
                m = np.array([[c, sn], [sn, -c]])
            ms.append(m)
        return ms

    def taps_of(th):
        ms = mats(th)
        H = [ms[0]]
        for s in range(1, K):
            DH = []
            for q in range(len(H) + 1):
                top = H[q][0] if q < len(H) else np.zeros(2)
                bot = H[q - 1][1] if q >= 1 else np.zeros(2)
                DH.append(np.stack([top, bot]))
            H = [ms[s] @ m for m in DH]
        out_lo = np.zeros(L)
        out_hi = np.zeros(L)
        for q in range(K):
            out_lo[2 * q + 1], out_lo[2 * q] = H[q][0, 0], H[q][0, 1]
            out_hi[2 * q + 1], out_hi[2 * q] = H[q][1, 0], H[q][1, 1]
        return out_lo, out_hi

    def resid(th):
        l2, h2 = taps_of(th)
        return np.concatenate([l2 - lo, h2 - hi])

    # the peeled matrices may differ from plain rotations by signs: check the parametrisation reproduces them
    start = resid(th0)
    sol = least_squares(resid, th0, xtol=1e-15, ftol=1e-15, gtol=1e-15, method="lm")
    ms = mats(sol.x)
    return [mp.matrix(m.tolist()) for m in ms], float(np.abs(start).max()), float(np.abs(sol.fun).max())


def lattice_taps(rots):
    """Polyphase coefficients of R_{K-1} D ... D R_0 -> (lo, hi) tap lists (float)."""
    import mpmath as mp

    K = len(rots)
    H = [rots[0]]  # list over powers of z^-1
    for s in range(1, K):
        # D H: bottom row delayed by one
        z = mp.matrix(2, 2)
        DH = [mp.matrix([[H[q][0, 0] if q < len(H) else 0, H[q][0, 1] if q < len(H) else 0],
                         [H[q - 1][1, 0] if q >= 1 else 0, H[q - 1][1, 1] if q >= 1 else 0]]) for q in range(len(H) + 1)]
        H = [rots[s] * m for m in DH]
        del z
    lo = [0.0] * (2 * K)
    hi = [0.0] * (2 * K)
    for q in range(K):
        lo[2 * q + 1], lo[2 * q] = float(H[q][0, 0]), float(H[q][0, 1])
        hi[2 * q + 1], hi[2 * q] = float(H[q][1, 0]), float(H[q][1, 1])
    return lo, hi


def scaled_form(rots):
    """One packed FMA per stage: (A, B) <- (A + alpha B', A + beta B') with B' the delayed scaled second channel.
    Returns (alpha[K], beta[K], scale_a, scale_b): cA = A / scale_a ... as multipliers out_a, out_b to apply at the end."""
    K = len(rots)
    alpha, beta = [], []
    G, k = 1.0, 1.0  # A = G a, B = G k b
    for s in range(K):
        r00, r01, r10, r11 = (float(rots[s][0, 0]), float(rots[s][0, 1]), float(rots[s][1, 0]), float(rots[s][1, 1]))
        alpha.append(r01 / (r00 * k))
        beta.append(r11 / (r10 * k))
        G = G / r00
        k = r00 / r10
    return alpha, beta, 1.0 / G, 1.0 / (G * k)


def f32(x):
    return np.asarray(x, dtype=np.float32)


def fma32(a, b, c):
    """float32 fused multiply-add emulated exactly: the float64 product of two float32 values is exact, and the
    float64 sum rounds once at 53 bits before the final float32 rounding (double rounding cases are ~2^-29 rare)."""
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


def extended_pairs(x, L):
    """x [..., n] -> (e, o) [..., J] for j = -(K-1) .. n_out-1 of the reflect extension (float32 in, float32 out)."""
    n = x.shape[-1]
    K = L // 2
    n_out = (n + L - 2 + (n & 1)) // 2
    idx = np.arange(-(L - 2), 2 * n_out)  # xe[2j], xe[2j+1] for j = -(K-1) .. n_out-1
    idx = np.abs(idx)
    idx = np.where(idx >= n, 2 * (n - 1) - idx, idx)
    xe = x[..., idx]
    return xe[..., 0::2], xe[..., 1::2], n_out, K


def lattice_step32(x, L, alpha, beta, out_a, out_b, normalise=True):
    """One analysis step of every node x [..., n] through the scaled lattice in float32."""
    e, o, n_out, K = extended_pairs(x, L)
    A = fma32(f32(alpha[0]), o, e)
    B = fma32(f32(beta[0]), o, e)
    for s in range(1, K):
        Bd = B[..., :-1]   # B at position i-1
        Ai = A[..., 1:]
        A, B = fma32(f32(alpha[s]), Bd, Ai), fma32(f32(beta[s]), Bd, Ai)
    assert A.shape[-1] == n_out
    if normalise:
        return A * f32(out_a), B * f32(out_b)
    return A, B


def direct_step32(x, lo, hi):
    """The float32 direct form in the order the round-3 kernels use: FMA chain over the taps."""
    L = len(lo)
    n = x.shape[-1]
    n_out = (n + L - 2 + (n & 1)) // 2
    idx = np.arange(-(L - 2), 2 * n_out)
    idx = np.abs(idx)
    idx = np.where(idx >= n, 2 * (n - 1) - idx, idx)
    xe = x[..., idx]
    ca = np.zeros(x.shape[:-1] + (n_out,), np.float32)
    cd = np.zeros_like(ca)
    for k in range(L):
        seg = xe[..., k:k + 2 * n_out:2]
        ca = fma32(f32(lo[L - 1 - k]), seg, ca)
        cd = fma32(f32(hi[L - 1 - k]), seg, cd)
    return ca, cd


def transform32(x, level, step):
    cur = f32(x)[..., None, :]
    for _ in range(level):
        ca, cd = step(cur)
        cur = np.stack([ca, cd], axis=-2).reshape(ca.shape[:-2] + (-1, ca.shape[-1]))
    f = np.arange(1 << level)
    return cur[..., f ^ (f >> 1), :]


def node_lengths(n, L, levels):
    out = [n]
    for _ in range(levels):
        n = (n + L - 2 + (n & 1)) // 2
        out.append(n)
    return out


def op_counts(L, top_run=None):
    """Multiply-adds per frame (N = 22 050, level 14)."""
    K = L // 2
    n = node_lengths(22050, L, 14)
    direct = sum((1 << k) * n[k] * L for k in range(1, 15))   # L multiply-adds per filter output
    direct_top = sum((1 << k) * n[k] * L for k in range(1, 9))
    # lattice, whole-node runs: per parent node K n_out + K (K-1) / 2 rotations of 2 multiply-adds
    lat = {k: (1 << (k - 1)) * (K * n[k] + K * (K - 1) // 2) * 2 for k in range(1, 15)}
    lat_deep = sum(lat[k] for k in range(9, 15))
    res = {"direct_form": direct, "direct_form_levels_1_8": direct_top, "direct_form_levels_9_14": direct - direct_top,
           "lattice_whole_node_runs_levels_9_14": lat_deep, "lattice_whole_node_runs_all_levels": sum(lat.values())}
    for R in (4, 8, 16, 32):
        res[f"lattice_runs_of_{R}_levels_1_8"] = sum(
            (1 << (k - 1)) * -(-n[k] // R) * (K * R + K * (K - 1) // 2) * 2 for k in range(1, 9))
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", default=None)
    ap.add_argument("--frames", type=int, default=2)
    a = ap.parse_args()
    rng = np.random.default_rng(0)
    t = np.arange(22050) / 22050.0
    noise = np.clip(0.1 * rng.standard_normal((a.frames, 22050)), -1, 1)
    tones = sum(np.sin(2 * np.pi * f * t) for f in (440.0, 3000.0, 7500.0, 10500.0))[None, :] * 0.25
    imp = np.zeros((3, 22050))
    imp[0, 0] = imp[1, 11025] = imp[2, 22049] = 1.0
    inputs = {"noise": noise, "tones": tones, "impulses": imp}
    report = {"bar": "5e-6 of the largest coefficient (tests/test_wpt_gpu.py:20)", "wavelets": {}}
    for name, lo in (("coif4", wpt_oracle.COIF4), ("sym5", wpt_oracle.SYM5), ("db8", DB8)):
        L = len(lo)
        hi = wpt_oracle.dec_hi_from_lo(lo)
        rots = factor(lo)
        import mpmath as mp
        ortho = max(float(abs((r.T * r)[i, j] - (1 if i == j else 0))) for r in rots for i in range(2) for j in range(2))
        peeled_err = None
        rots, peeled_err, _polished = polish(rots, lo)
        lo2, hi2 = lattice_taps(rots)
        tap_err = max(max(abs(x - y) for x, y in zip(lo, lo2)), max(abs(x - y) for x, y in zip(hi, hi2)))
        alpha, beta, out_a, out_b = scaled_form(rots)
        angles = [float(mp.atan2(r[1, 0], r[0, 0])) for r in rots]
        w = {"taps": L, "stages": L // 2, "rotation_angles_rad": angles,
             "rotation_orthogonality_residual": ortho, "peeled_lattice_vs_table_taps_max_abs": peeled_err, "lattice_vs_table_taps_max_abs": tap_err,
             "alpha": alpha, "beta": beta, "out_scale_a": out_a, "out_scale_b": out_b,
             "max_abs_coefficient": max(max(map(abs, alpha)), max(map(abs, beta))), "errors": {}}
        for level in (8, 14):
            for iname, x in inputs.items():
                ref = wpt_oracle.wpt_nodes(x, lo, level)
                scale = np.abs(ref).max()
                lat = transform32(x, level, lambda c: lattice_step32(c, L, alpha, beta, out_a, out_b))
                dire = transform32(x, level, lambda c: direct_step32(c, lo, hi))
                e_lat = float(np.abs(lat - ref).max() / scale)
                e_dir = float(np.abs(dire - ref).max() / scale)
                w["errors"][f"level{level}_{iname}"] = {"lattice_f32": e_lat, "direct_f32": e_dir,
                                                        "passes_5e-6": bool(e_lat <= 5e-6)}
                print(f"{name} level {level} {iname:9s}: lattice f32 {e_lat:.2e}  direct f32 {e_dir:.2e}  (of max |c|)", flush=True)
        w["multiply_adds_per_frame_level14"] = op_counts(L)
        report["wavelets"][name] = w
        print(name, "angles", np.round(angles, 4).tolist())
        print(name, "max |alpha|,|beta|", w["max_abs_coefficient"], "out scales", out_a, out_b, "tap err", tap_err, "ortho", ortho)
        print(name, json.dumps(w["multiply_adds_per_frame_level14"]))
    if a.json:
        with open(a.json, "w") as fh:
            json.dump(report, fh, indent=1)


if __name__ == "__main__":
    main()
