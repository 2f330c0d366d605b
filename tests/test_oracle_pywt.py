"""Pins the three CPU restatements of the wavelet-packet path, and the product's tap tables, on PyWavelets.

Fixtures: tests/golden/pywt_*.npz, written by tests/golden/make_wpt_golden.py from
``pywt.WaveletPacket(x, w, mode="reflect", maxlevel=l).get_level(l, order="freq")`` -- the call of the reference's
scripts/freq_visual/fingerprints.py:101-106 -- and ``pywt.Wavelet(name).dec_lo / dec_hi`` (the reference's tap source,
src/audiofakedetect/wavelet_math.py:239).  pywt 1.1.1, float64.

Tolerances: with pywt's own taps the float64 restatements equal pywt to summation order: 1e-13 absolute on
coefficients <= 8.  The product carries a higher-precision coif4 table than pywt 1.1.1's (they differ by 2.3e-8 per
tap, below half an ulp of the fp32 taps the kernels use): with the product's table coif4 agrees to 1e-7 relative.
"""

import os

import numpy as np
import pytest
import torch

from audiofakedetect import wavelets
from oracle import c_oracle, torch_ref, wpt_oracle

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
START_EXPS = ([f"sym{k}" for k in range(2, 11)] + [f"db{k}" for k in range(2, 11)]
              + [f"coif{k}" for k in range(2, 11)])  # reference scripts/start_exps.sh:3-31
EXTREMES = ["db11", "db20", "db38", "sym11", "sym20", "coif1", "coif11", "coif17", "dmey",
            "bior2.2", "bior4.4", "bior6.8", "rbio3.9"]


@pytest.fixture(scope="module")
def taps():
    return np.load(os.path.join(GOLD, "pywt_taps.npz"))


@pytest.fixture(scope="module")
def core():
    return np.load(os.path.join(GOLD, "pywt_wpt_core.npz"))


@pytest.fixture(scope="module")
def deep():
    return np.load(os.path.join(GOLD, "pywt_wpt_l14.npz"))


@pytest.fixture(scope="module")
def names():
    return np.load(os.path.join(GOLD, "pywt_wpt_names.npz"))


def test_product_tables_equal_pywt(taps):
    """Every discrete wavelet pywt names constructs, with pywt's dec_lo / dec_hi (coif4: the finer table)."""
    seen = sorted({k.split("/")[0] for k in taps.files})
    assert len(seen) == 106 and set(START_EXPS) <= set(seen) and "sym8" in seen
    for name in seen:
        w = wavelets.Wavelet(name)
        tol = 3e-8 if name == "coif4" else 5e-15
        assert w.dec_len == len(taps[name + "/lo"])
        assert np.max(np.abs(np.array(w.dec_lo) - taps[name + "/lo"])) <= tol, name
        assert np.max(np.abs(np.array(w.dec_hi) - taps[name + "/hi"])) <= tol, name


def test_oracle_tables_equal_pywt(taps):
    for name in ("haar", "sym5"):
        assert np.max(np.abs(np.array(wpt_oracle.TAPS[name]) - taps[name + "/lo"])) <= 2.3e-16  # 1/sqrt(2) vs pywt's rounding
    assert np.max(np.abs(np.array(wpt_oracle.TAPS["coif4"]) - taps["coif4/lo"])) <= 3e-8
    for name in ("haar", "sym5", "coif4", "sym8", "db8"):
        assert np.array_equal(np.array(wpt_oracle.dec_hi_from_lo(taps[name + "/lo"])), taps[name + "/hi"])


@pytest.mark.parametrize("name", ["haar", "sym5", "coif4", "db8", "sym8"])
@pytest.mark.parametrize("level", [1, 3, 8])
def test_numpy_and_c_restatements_equal_pywt(core, taps, name, level):
    x = core["x"].astype(np.float64)
    ref = core[f"{name}/{level}"]
    lo = taps[name + "/lo"]
    got = wpt_oracle.wpt_nodes(x, lo, level)
    assert got.shape == ref.shape
    assert np.max(np.abs(got - ref)) <= 1e-13
    assert np.max(np.abs(c_oracle.wpt_nodes_c(x, lo, level) - ref)) <= 1e-13
    # the table the product and the GPU tests use (differs from pywt 1.1.1's for coif4 only)
    own = wpt_oracle.wpt_nodes(x, wavelets.Wavelet(name).dec_lo, level)
    assert np.max(np.abs(own - ref)) <= (1e-7 if name == "coif4" else 1e-13) * np.max(np.abs(ref))


@pytest.mark.parametrize("name", ["haar", "sym5", "coif4", "sym8"])
def test_level_14_equals_pywt(deep, taps, name):
    """Level 14 (BASELINE configs[1], [2], [3]): float32-stored nodes + float64 node sums / energies."""
    x = deep["x"].astype(np.float64)
    lo = taps[name + "/lo"]
    for got in (wpt_oracle.wpt_nodes(x, lo, 14)[0], c_oracle.wpt_nodes_c(x, lo, 14)[0]):
        ref = deep[name]
        assert got.shape == ref.shape == (16384, wavelets.level_lengths(22050, len(lo), 14)[-1])
        assert np.max(np.abs(got - ref)) <= 1e-7 * np.max(np.abs(ref))  # float32 storage of the fixture
        assert np.max(np.abs(got.sum(-1) - deep[name + "/sum"])) <= 1e-12
        assert np.max(np.abs((got * got).sum(-1) - deep[name + "/sumsq"])) <= 1e-11


@pytest.mark.parametrize("name", START_EXPS + EXTREMES)
def test_every_launched_wavelet_equals_pywt(names, taps, name):
    """scripts/start_exps.sh:3-31 at its level 8 (--num-of-scales 256), plus the longest of each family."""
    x = names["x"].astype(np.float64)
    got = wpt_oracle.wpt_nodes(x, taps[name + "/lo"], 8, taps[name + "/hi"])[0]
    ref = names[name]
    assert got.shape == ref.shape
    assert np.max(np.abs(got - ref)) <= 1e-7 * np.max(np.abs(ref))
    assert np.max(np.abs(got.sum(-1) - names[name + "/sum"])) <= 1e-12
    assert np.max(np.abs((got * got).sum(-1) - names[name + "/sumsq"])) <= 1e-11


@pytest.mark.parametrize("name,level", [("sym8", 8), ("coif4", 8), ("sym5", 3)])
def test_torch_restatement_equals_pywt(core, taps, name, level):
    """The F.pad + F.conv1d form (what ptwt launches), float64 here: equal to pywt, node for node."""
    x = torch.from_numpy(core["x"].astype(np.float64))
    got, _ = torch_ref.packets_torch(x, taps[name + "/lo"], level, compute_welford=False, per_node=(level <= 3))
    ref = core[f"{name}/{level}"]
    assert np.max(np.abs(got[:, 0].numpy() - ref)) <= 1e-13
    # and in the reference's precision (float32 taps and data): 1e-5 relative
    got32, _ = torch_ref.packets_torch(x.float(), taps[name + "/lo"], level, compute_welford=False, per_node=False)
    assert np.max(np.abs(got32[:, 0].double().numpy() - ref)) <= 1e-5 * np.max(np.abs(ref))
