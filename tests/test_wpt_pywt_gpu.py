"""GPU parity of the wavelet-packet kernels (through the C ABI, via the `Packets` plugin) against PyWavelets directly.

The fixtures tests/golden/pywt_*.npz hold the float64 outputs of
``pywt.WaveletPacket(x, w, mode="reflect", maxlevel=l).get_level(l, order="freq")`` -- the reference's own call in
scripts/freq_visual/fingerprints.py:101-106; taps from ``pywt.Wavelet``, as src/audiofakedetect/wavelet_math.py:239 --
written by tests/golden/make_wpt_golden.py.  No oracle in between: fp32 kernel vs pywt,

    max|gpu - pywt| <= 5e-6 * max|pywt|      (the bar of tests/test_wpt_gpu.py; the reference computes in fp32 too).
"""

import os

import numpy as np
import pytest
import torch

from audiofakedetect.wavelet_math import Packets

pytestmark = pytest.mark.gpu
COEF_RTOL = 5e-6
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
START_EXPS = ([f"sym{k}" for k in range(2, 11)] + [f"db{k}" for k in range(2, 11)]
              + [f"coif{k}" for k in range(2, 11)])  # reference scripts/start_exps.sh:3-31
EXTREMES = ["db11", "db20", "db38", "sym11", "sym20", "coif1", "coif11", "coif17", "dmey",
            "bior2.2", "bior4.4", "bior6.8", "rbio3.9"]


def _load(name):
    return np.load(os.path.join(GOLD, name))


def _check(got, ref):
    got = got.cpu().double().numpy()
    assert got.shape == ref.shape
    scale = np.max(np.abs(ref))
    err = np.max(np.abs(got - ref))
    assert err <= COEF_RTOL * scale, f"coef err {err:.3e} vs scale {scale:.3e}"


@pytest.mark.parametrize("name", ["haar", "sym5", "coif4", "db8", "sym8"])
@pytest.mark.parametrize("level", [1, 3, 8])
def test_core_wavelets_match_pywt(name, level):
    core = _load("pywt_wpt_core.npz")
    x = torch.from_numpy(core["x"]).cuda()  # noise, four tones, impulses at 0 / 11 025 / 22 049
    got, _ = Packets(name, max_lev=level)(x)
    _check(got[:, 0], core[f"{name}/{level}"])


@pytest.mark.parametrize("name", ["haar", "sym5", "coif4", "sym8"])
def test_level_14_matches_pywt(name):
    """BASELINE configs[1] (coif4), [2] (sym5), [3] (haar) and the reference's default wavelet at level 14."""
    deep = _load("pywt_wpt_l14.npz")
    x = torch.from_numpy(deep["x"]).cuda()
    got, _ = Packets(name, max_lev=14)(x)
    _check(got[0, 0], deep[name].astype(np.float64))
    # the frame inside a batch of 9 (the batched kernels' frame indexing; B % 8 != 0 takes the other block order)
    xb = torch.cat([torch.zeros(5, 22050), torch.from_numpy(deep["x"]), torch.ones(3, 22050)]).cuda()
    gb, _ = Packets(name, max_lev=14)(xb)
    assert torch.equal(gb[5], got[0])


@pytest.mark.parametrize("name", START_EXPS + EXTREMES)
def test_every_launched_wavelet_matches_pywt(name):
    """Every --wavelet of scripts/start_exps.sh:3-31 at its level 8 (+ the longest of each pywt family, which take the
    generic kernel: 34..102 taps), raw coefficients; then the log-power features the launches ask for."""
    names = _load("pywt_wpt_names.npz")
    x = torch.from_numpy(names["x"]).cuda()
    ref = names[name].astype(np.float64)
    got, _ = Packets(name, max_lev=8)(x)
    _check(got[0, 0], ref)
    logp, _ = Packets(name, max_lev=8, log_scale=True, power=2.0)(x)
    d = COEF_RTOL * np.max(np.abs(ref))
    lref = np.log(ref * ref + 1e-12)
    bound = 1e-5 + 2 * np.abs(ref) * d / (ref * ref + 1e-12) + 2e-6 * np.abs(lref)
    assert np.all(np.abs(logp[0, 0].cpu().double().numpy() - lref) <= bound)


def test_default_wavelet_constructs():
    """`Packets()` with the reference's defaults (wavelet_math.py:226-236: sym8, level 8) on a training-shaped batch."""
    core = _load("pywt_wpt_core.npz")
    x = torch.from_numpy(core["x"]).cuda().unsqueeze(1)
    got, _ = Packets()(x)
    assert got.shape == (3, 1, 256, 101)
    _check(got[:, 0], core["sym8/8"])


def test_the_reference_side_stub_of_integration_md_runs_as_written():
    """INTEGRATION.md shows the ctypes stub a maintainer would put in the reference's wavelet_math.py.  The code block is
    executed verbatim (with the in-repo `Wavelet` standing in for `pywt.Wavelet`: the stub reads dec_lo / dec_hi /
    dec_len only) and its result is held to the pywt fixture, at level 8 and at level 14 (workspace path)."""
    import re
    import sys
    import types

    from audiofakedetect import _native, wavelets

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    block = re.search(r"```python\n(# reference-side stub.*?)```", text, re.S).group(1)
    block = block.replace('ctypes.CDLL("libafd_hip.so")', f'ctypes.CDLL("{_native.LIB_PATH}")')
    fake_pywt = types.ModuleType("pywt")
    fake_pywt.Wavelet = wavelets.Wavelet
    saved = sys.modules.get("pywt")
    sys.modules["pywt"] = fake_pywt
    try:
        ns: dict = {}
        exec(compile(block, "INTEGRATION.md", "exec"), ns)
    finally:
        if saved is None:
            del sys.modules["pywt"]
        else:
            sys.modules["pywt"] = saved
    core, deep = _load("pywt_wpt_core.npz"), _load("pywt_wpt_l14.npz")
    out, _ = ns["packets_mi355x"](torch.from_numpy(core["x"]), wavelets.Wavelet("sym8"), 8, False, False, 2.0)
    _check(out[:, 0], core["sym8/8"])
    out, _ = ns["packets_mi355x"](torch.from_numpy(deep["x"]), wavelets.Wavelet("coif4"), 14, False, False, 2.0)
    _check(out[0, 0], deep["coif4"].astype(np.float64))
