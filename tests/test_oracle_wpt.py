"""Pins the WPT oracle: reference shape asserts + mathematical known answers.

The reference's tests hold shapes only for this path (reference
tests/test_transforms.py:54-142); the rest are known answers that do not need ptwt.
"""

import math

import numpy as np
import pytest
import torch

from audiofakedetect import wavelets
from oracle import c_oracle, torch_ref, wpt_oracle


def _rand(b, n, seed=0):
    return np.random.default_rng(seed).standard_normal((b, n))


@pytest.mark.parametrize("name", ["haar", "db2", "db8", "sym5", "coif4"])
def test_taps_are_orthonormal_qmf(name):
    w = wavelets.Wavelet(name)
    lo = np.array(w.dec_lo)
    hi = np.array(w.dec_hi)
    length = len(lo)
    assert abs(lo.sum() - math.sqrt(2)) < 1e-10
    assert abs((lo * lo).sum() - 1) < 1e-10
    assert abs(hi.sum()) < 1e-10
    for s in range(1, length // 2):
        assert abs(np.dot(lo[2 * s:], lo[: length - 2 * s])) < 1e-10
        assert abs(np.dot(hi[2 * s:], hi[: length - 2 * s])) < 1e-10
    for s in range(-(length // 2) + 1, length // 2):
        a = lo[max(0, 2 * s): length + min(0, 2 * s)]
        b = hi[max(0, -2 * s): length - max(0, 2 * s)]
        assert abs(np.dot(a, b)) < 1e-10


def test_vanishing_moments():
    # sym5: 5 vanishing moments of the high-pass; coif4: 8
    for name, nmom in (("sym5", 5), ("coif4", 8), ("db8", 8), ("db2", 2)):
        hi = np.array(wavelets.Wavelet(name).dec_hi)
        k = np.arange(len(hi), dtype=np.float64)
        for p in range(nmom):
            assert abs(np.dot(hi, k ** p)) < 1e-7 * max(1.0, (len(hi) ** p)), (name, p)


def test_db2_closed_form():
    s3 = math.sqrt(3.0)
    rec_lo = np.array([1 + s3, 3 + s3, 3 - s3, 1 - s3]) / (4 * math.sqrt(2))
    assert np.allclose(wavelets.Wavelet("db2").dec_lo, rec_lo[::-1], atol=1e-12)


def test_product_and_oracle_tables_agree():
    for name in ("haar", "sym5", "coif4"):
        assert np.allclose(wavelets.Wavelet(name).dec_lo, wpt_oracle.TAPS[name], atol=0, rtol=0)


def test_reference_shape_asserts():
    # reference tests/test_transforms.py:79,98 (db8, level 7) and :124,142 (sym8 = 16 taps)
    x = _rand(2, 22050)
    lo = wavelets.Wavelet("db8").dec_lo
    out = wpt_oracle.packet_features(x, lo, 7, log_scale=True)
    assert out.shape == (2, 1, 128, 187)
    out = wpt_oracle.packet_features(x, lo, 7, log_scale=True, loss_less=True)
    assert out.shape == (2, 2, 128, 187)
    assert set(np.unique(out[:, 1])) <= {-1.0, 1.0}


@pytest.mark.parametrize("name,l8,l14", [("haar", 87, 2), ("sym5", 95, 10), ("coif4", 109, 24)])
def test_level_lengths_table(name, l8, l14):
    length = wavelets.Wavelet(name).dec_len
    lens = wavelets.level_lengths(22050, length, 14)
    assert lens[8] == l8 and lens[14] == l14


def test_haar_closed_form():
    x = _rand(1, 10)
    ca, cd = wpt_oracle.analysis_step(x, wpt_oracle.HAAR)
    s = 1 / math.sqrt(2)
    assert np.allclose(ca[0], (x[0, 0::2] + x[0, 1::2]) * s)
    assert np.allclose(cd[0], (x[0, 0::2] - x[0, 1::2]) * s)
    # odd length: the reflect sample x[n-2] pairs with x[n-1]
    x = _rand(1, 11)
    ca, cd = wpt_oracle.analysis_step(x, wpt_oracle.HAAR)
    assert ca.shape[-1] == 6
    assert np.isclose(ca[0, -1], (x[0, 10] + x[0, 9]) * s)
    assert np.isclose(cd[0, -1], (x[0, 10] - x[0, 9]) * s)


@pytest.mark.parametrize("name", ["haar", "sym5", "coif4"])
def test_constant_input(name):
    lo = wpt_oracle.TAPS[name]
    x = np.full((1, 4096), 0.37)
    level = 3
    nodes = wpt_oracle.wpt_nodes(x, lo, level)[0]
    # reflect extension of a constant is the constant: the all-low-pass packet is
    # 2^(level/2) c everywhere, every other packet vanishes.
    assert np.allclose(nodes[0], 0.37 * 2 ** (level / 2), atol=1e-10)
    assert np.allclose(nodes[1:], 0.0, atol=1e-10)  # table precision ~3e-13


@pytest.mark.parametrize("name", ["sym5", "coif4", "db8"])
def test_pure_tone_lands_in_gray_ordered_packet(name):
    lo = wavelets.Wavelet(name).dec_lo
    fs, n, level = 22050, 22050, 5
    t = np.arange(n) / fs
    for f in (440.0, 3000.0, 7500.0, 10500.0):
        x = np.sin(2 * np.pi * f * t)[None]
        nodes = wpt_oracle.wpt_nodes(x, lo, level)[0]
        energy = (nodes ** 2).sum(-1)
        assert int(np.argmax(energy)) == int(f / (fs / 2) * (1 << level)), (name, f)


@pytest.mark.parametrize("name", ["haar", "sym5", "coif4"])
def test_energy_close_to_preserved(name):
    # orthogonal filter bank: interior coefficients preserve energy; the reflect
    # border adds O(L / n) extra.
    lo = wpt_oracle.TAPS[name]
    x = _rand(1, 22050, seed=3)
    nodes = wpt_oracle.wpt_nodes(x, lo, 4)
    ratio = (nodes ** 2).sum() / (x ** 2).sum()
    assert 0.99 < ratio < 1.03


@pytest.mark.parametrize("name,level", [("haar", 14), ("sym5", 8), ("coif4", 8), ("coif4", 14)])
def test_numpy_c_and_torch_restatements_agree(name, level):
    lo = wpt_oracle.TAPS[name]
    x = 0.1 * _rand(2, 22050, seed=5)
    ref = wpt_oracle.wpt_nodes(x, lo, level)
    cres = c_oracle.wpt_nodes_c(x, lo, level)
    assert cres.shape == ref.shape
    assert np.max(np.abs(cres - ref)) < 1e-13
    got, wd = torch_ref.packets_torch(torch.from_numpy(x).float(), lo, level,
                                      compute_welford=(level <= 8), per_node=(level <= 8))
    got = got[:, 0].double().numpy()
    scale = np.max(np.abs(ref))
    # float32 restatement vs float64 restatement
    assert np.max(np.abs(got - ref)) < 1e-5 * scale
    if level <= 8:
        assert len(wd) == 1 << level


@pytest.mark.parametrize("name,level", [("haar", 6), ("sym5", 5), ("coif4", 4)])
def test_vectorised_traversal_matches_path_dict(name, level):
    lo = wpt_oracle.TAPS[name]
    x = _rand(2, 3001, seed=11)
    a = wpt_oracle.wpt_nodes(x, lo, level)
    b = wpt_oracle.wpt_nodes_by_path(x, lo, level)
    assert np.array_equal(a, b)
    # Gray code rule: path string of frequency index f is the bits of f ^ (f >> 1)
    for f, path in enumerate(wpt_oracle.graycode_paths(level)):
        g = f ^ (f >> 1)
        assert path == "".join("d" if (g >> (level - 1 - k)) & 1 else "a" for k in range(level))


def test_features_c_vs_numpy():
    lo = wpt_oracle.TAPS["sym5"]
    x = 0.1 * _rand(2, 22050, seed=7)
    a = wpt_oracle.packet_features(x, lo, 6, log_scale=True, loss_less=True)
    b = c_oracle.packet_features_c(x, lo, 6, log_scale=True, loss_less=True)
    assert np.allclose(a, b, atol=1e-9)
    tfeat, _ = torch_ref.packets_torch(torch.from_numpy(x).float(), lo, 6, log_scale=True,
                                       loss_less=True)
    assert tfeat.shape == (2, 2, 64, a.shape[-1])
    assert not tfeat.is_contiguous()  # memory order [B, C, T, P] like the reference
    assert np.array_equal(tfeat[:, 1].numpy(), a[:, 1])


def test_stft_shapes_from_reference_tests():
    # reference tests/test_transforms.py:36 and :51
    x = torch.randn(2, 1, 22050)
    assert torch_ref.stft_torch(x, 512, 2).shape == (2, 1, 257, 11026)
    assert torch_ref.stft_torch(x).shape == (2, 1, 256, 101)
