"""The orthogonal lattice the level-9..14 kernel (csrc/wpt4.hip) runs the analysis filter bank in, checked on the CPU:
`afd_wpt_lattice` factors a tap table on the host (no GPU call), and one lattice step in float64 reproduces the
oracle's direct-form analysis step (reference src/audiofakedetect/wavelet_math.py:182,192 -> ptwt's pad + conv1d)."""

import ctypes

import numpy as np
import pytest

from audiofakedetect import _native, wavelets
from oracle import wpt_oracle


def _lattice(lo, hi):
    lib = _native.load()
    L = len(lo)
    K = L // 2
    c_lo = (ctypes.c_float * L)(*lo)
    c_hi = (ctypes.c_float * L)(*hi)
    al, be, sc, res = (ctypes.c_double * K)(), (ctypes.c_double * K)(), (ctypes.c_double * 2)(), ctypes.c_double()
    rc = lib.afd_wpt_lattice(c_lo, c_hi, L, al, be, sc, ctypes.byref(res))
    return rc, np.array(al), np.array(be), np.array(sc), res.value


def _lattice_step(x, L, al, be, sc):
    """One analysis step through the scaled lattice (float64), as include/afd_hip.h states it."""
    n = x.shape[-1]
    K = L // 2
    n_out = (n + L - 2 + (n & 1)) // 2
    idx = np.abs(np.arange(-(L - 2), 2 * n_out))
    idx = np.where(idx >= n, 2 * (n - 1) - idx, idx)
    xe = x[..., idx]
    e, o = xe[..., 0::2], xe[..., 1::2]
    A, B = e + al[0] * o, e + be[0] * o
    for s in range(1, K):
        A, B = A[..., 1:] + al[s] * B[..., :-1], A[..., 1:] + be[s] * B[..., :-1]
    assert A.shape[-1] == n_out
    return sc[0] * A, sc[1] * B


@pytest.mark.parametrize("name", ["coif4", "sym5", "db8", "db2", "db3"])
def test_lattice_step_equals_the_direct_form(name):
    lo = wavelets.Wavelet(name).dec_lo
    hi = wpt_oracle.dec_hi_from_lo(lo)
    rc, al, be, sc, res = _lattice(lo, hi)
    assert rc == 0
    # the float32 tap table is orthogonal to ~1e-8: the nearest lattice reproduces it to that
    assert res <= 2e-7, res
    rng = np.random.default_rng(1)
    for n in (109, 66, 33, 28, 25):
        if len(lo) - 2 + (n & 1) >= n:
            continue
        x = rng.standard_normal((3, n))
        ca, cd = wpt_oracle.analysis_step(x, lo)
        la, ld = _lattice_step(x, len(lo), al, be, sc)
        scale = max(np.abs(ca).max(), np.abs(cd).max())
        assert np.abs(la - ca).max() <= 5e-7 * scale and np.abs(ld - cd).max() <= 5e-7 * scale, (name, n)


def test_a_bank_that_is_not_orthogonal_has_no_lattice():
    lo = list(wavelets.Wavelet("sym5").dec_lo)
    hi = wpt_oracle.dec_hi_from_lo(lo)
    lo[3] += 0.05
    rc, *_ = _lattice(lo, hi)
    assert rc != 0


def test_a_nearly_orthogonal_bank_is_not_run_through_the_nearest_lattice():
    """A bank whose taps are off an orthogonal table by more than a few float32 ulp (a learned or hand-edited filter)
    must keep ITS taps: the lattice is refused and `afd_wpt_forward` falls back to the direct-form kernels."""
    lo = list(wavelets.Wavelet("sym5").dec_lo)
    lo[3] += 1e-5  # fits a lattice to 3e-6 -- good enough to look right, wrong by 1e-5 per level
    rc, *_, res = _lattice(lo, wpt_oracle.dec_hi_from_lo(lo))
    assert rc != 0 and 1e-6 < res < 1e-5, (rc, res)
    lo = list(wavelets.Wavelet("sym5").dec_lo)
    lo[3] += 1e-7  # within the float32 rounding of the table itself
    rc, *_, res = _lattice(lo, wpt_oracle.dec_hi_from_lo(lo))
    assert rc == 0 and res <= 2e-7
