"""GPU parity of the fused wavelet-packet kernel (through the C ABI) against the oracle.

Tolerances (fp32 kernel vs float64 oracle; the reference itself computes in fp32):
  coefficients : max|gpu - ref| <= 5e-6 * max|ref|                      (COEF_RTOL)
  log features : |gpu - ref| <= 1e-5 + 2|x| d / (x^2 + 1e-12), d = COEF_RTOL * max|x|
                 (first-order bound of log(x^2+eps) under a coefficient error d: the log
                 is ill-conditioned at zero crossings for the reference as well)
  sign channel : bit-exact wherever |x| > d
"""

import numpy as np
import pytest
import torch

from audiofakedetect import wavelets
from audiofakedetect.wavelet_math import Packets, compute_pytorch_packet_representation
from oracle import wpt_oracle

pytestmark = pytest.mark.gpu
COEF_RTOL = 5e-6
N = 22050


def _parity_inputs(seed=0):
    g = torch.Generator().manual_seed(seed)
    x = (0.1 * torch.randn(3, N, generator=g)).clamp_(-1, 1)
    t = torch.arange(N) / 22050.0
    tone = sum(torch.sin(2 * np.pi * f * t) for f in (440.0, 3000.0, 7500.0, 10500.0)) / 4
    imp = torch.zeros(3, N)
    imp[0, 0] = 1.0
    imp[1, 11025] = 1.0
    imp[2, 22049] = 1.0
    return torch.cat([x, tone[None], imp], 0)


def _check_coeffs(got, ref):
    scale = np.max(np.abs(ref))
    err = np.max(np.abs(got - ref))
    assert err <= COEF_RTOL * scale, f"coef err {err:.3e} vs scale {scale:.3e}"


@pytest.mark.parametrize("name,level", [
    ("haar", 1), ("haar", 2), ("haar", 8), ("haar", 14),
    ("sym5", 3), ("sym5", 8), ("sym5", 14),
    ("coif4", 1), ("coif4", 8), ("coif4", 9), ("coif4", 14),
    ("db8", 7), ("db8", 8), ("db8", 14), ("db2", 10), ("db3", 5),
])
def test_coefficients_match_oracle(name, level):
    x = _parity_inputs()
    w = wavelets.Wavelet(name)
    got, _ = Packets(name, max_lev=level)(x.cuda())
    ref = wpt_oracle.packet_features(x.double().numpy(), w.dec_lo, level)
    assert tuple(got.shape) == ref.shape
    _check_coeffs(got.cpu().double().numpy(), ref)


@pytest.mark.parametrize("name,level", [("sym5", 8), ("coif4", 8), ("coif4", 14), ("haar", 14)])
def test_log_power_and_sign_channel(name, level):
    x = _parity_inputs(seed=1)[:4]
    w = wavelets.Wavelet(name)
    got, _ = Packets(name, max_lev=level, log_scale=True, loss_less=True)(x.cuda())
    got = got.cpu().double().numpy()
    coef = wpt_oracle.packet_features(x.double().numpy(), w.dec_lo, level)[:, 0]
    ref = wpt_oracle.packet_features(x.double().numpy(), w.dec_lo, level, log_scale=True,
                                     loss_less=True)
    assert got.shape == ref.shape and got.shape[1] == 2
    d = COEF_RTOL * np.max(np.abs(coef))
    bound = 1e-5 + 2 * np.abs(coef) * d / (coef ** 2 + 1e-12) + 2e-6 * np.abs(ref[:, 0])
    assert np.all(np.abs(got[:, 0] - ref[:, 0]) <= bound)
    sure = np.abs(coef) > d
    assert np.array_equal(got[:, 1][sure], ref[:, 1][sure])
    assert set(np.unique(got[:, 1])) <= {-1.0, 1.0}


def test_input_forms_and_layout():
    x = _parity_inputs()[:2]
    p = Packets("sym5", max_lev=8, log_scale=True)
    a, d = p(x.cuda())
    b, _ = p(x.unsqueeze(1).cuda())
    c, _ = p(x[:1])  # CPU [1, N] as get_input_dims passes it (reference utils.py:606-612)
    assert isinstance(d, dict)
    assert a.shape == (2, 1, 256, 95) and not a.is_contiguous()
    assert a.permute(0, 1, 3, 2).is_contiguous()  # memory order [B, C, T, P]
    assert not c.is_cuda  # results live on the input's device, as in the reference
    assert torch.equal(a, b) and torch.equal(a[:1].cpu(), c)
    raw, _ = compute_pytorch_packet_representation(x.cuda(), wavelets.Wavelet("sym5"), 8)
    assert raw.shape == (2, 1, 95, 256)


def test_reference_test_shapes():
    # reference tests/test_transforms.py:54-142 (db8 / 16-tap, level 7)
    x = torch.randn(2, 22050)
    out, d = compute_pytorch_packet_representation(x, wavelets.Wavelet("db8"), max_lev=7,
                                                   log_scale=True, compute_welford=True)
    assert out.shape == (2, 1, 187, 128) and d is not None
    out, _ = compute_pytorch_packet_representation(x, wavelets.Wavelet("db8"), max_lev=7,
                                                   log_scale=True, loss_less=True)
    assert out.shape == (2, 2, 187, 128)
    out, _ = Packets("db8", max_lev=7, log_scale=True)(x)
    assert out.shape == (2, 1, 128, 187)


def test_loss_less_per_channel_normalisation():
    """--loss-less True: two channels (log magnitude, sign) with their own Welford statistics, applied
    per channel by torchvision Normalize in the reference (wavelet_math.py:380-382, :441)."""
    from audiofakedetect.wavelet_math import Normalize, fuse_normalization
    from oracle import torch_ref

    x = _parity_inputs(seed=2)[:3]
    w = wavelets.Wavelet("sym5")
    mean = torch.tensor([-11.5, 0.03])
    std = torch.tensor([3.75, 0.98])
    ref, _ = torch_ref.packets_torch(x.double(), w.dec_lo, 8, log_scale=True, loss_less=True)
    ref = (ref - mean.double()[None, :, None, None]) / std.double()[None, :, None, None]
    coef = wpt_oracle.packet_features(x.double().numpy(), w.dec_lo, 8)[:, 0]
    d = COEF_RTOL * np.max(np.abs(coef))
    sure = torch.from_numpy(np.abs(coef) > 50 * d)

    def check(got):
        got = got.cpu().double()
        assert got.shape == ref.shape == (3, 2, 256, 95)
        assert (got[:, 0] - ref[:, 0])[sure].abs().max().item() <= 1e-4
        assert (got[:, 1] - ref[:, 1])[sure].abs().max().item() <= 1e-6
        # the sign channel holds exactly the two normalised values
        vals = torch.unique(got[:, 1].float())
        want = torch.sort((torch.tensor([-1.0, 1.0]) - mean[1]) / std[1])[0]
        assert torch.equal(vals, want)

    for name, level in (("sym5", 8),):
        tr = torch.nn.Sequential(Packets(name, max_lev=level, log_scale=True, loss_less=True))
        # (a) unfused: transform, then the Normalize module
        nm = torch.nn.Sequential(Normalize(mean, std))
        feats, _ = tr(x.cuda())
        check(nm(feats))
        # (b) fused into the transform epilogue (what the Trainer does)
        assert fuse_normalization(tr, nm) and nm[0].identity
        feats, _ = tr(x.cuda())
        check(nm(feats))
    # the fused epilogue of every kernel generation carries the sign statistics
    for name, level in (("coif4", 14), ("haar", 14), ("sym5", 14), ("coif4", 8)):
        tr = Packets(name, max_lev=level, log_scale=True, loss_less=True)
        plain, _ = tr(x.cuda())
        tr.fused_norm = ((-11.5, 0.03), (3.75, 0.98))
        fused, _ = tr(x.cuda())
        assert torch.allclose(fused[:, 0], (plain[:, 0] + 11.5) / 3.75, atol=2e-6, rtol=1e-6)
        assert torch.equal(fused[:, 1], (plain[:, 1] - 0.03) / 0.98)
    # block-norm path (afd_packet_block_norm)
    tr = Packets("sym5", max_lev=6, log_scale=True, loss_less=True, block_norm=True)
    plain, _ = tr(x.cuda())
    tr.fused_norm = ((-11.5, 0.03), (3.75, 0.98))
    fused, _ = tr(x.cuda())
    assert torch.allclose(fused[:, 0], (plain[:, 0] + 11.5) / 3.75, atol=2e-6, rtol=1e-6)
    assert torch.equal(fused[:, 1], (plain[:, 1] - 0.03) / 0.98)
    # a scalar statistic still applies to both channels, a wrong channel count raises
    nm1 = Normalize(torch.tensor(0.5), torch.tensor(2.0))
    assert torch.equal(nm1(plain), (plain - 0.5) / 2.0)
    with pytest.raises(ValueError):
        Normalize(torch.zeros(3), torch.ones(3))(plain)


def test_fused_normalisation():
    x = _parity_inputs()[:2].cuda()
    p = Packets("coif4", max_lev=8, log_scale=True)
    plain, _ = p(x)
    p.fused_norm = (-9.5, 3.25)
    fused, _ = p(x)
    assert torch.allclose(fused, (plain - (-9.5)) / 3.25, atol=1e-6, rtol=1e-6)


@pytest.mark.parametrize("n", [22051, 16000, 4097])
def test_other_frame_lengths(n):
    g = torch.Generator().manual_seed(n)
    x = torch.randn(2, n, generator=g)
    for name, level in (("sym5", 6), ("haar", 9)):
        got, _ = Packets(name, max_lev=level)(x.cuda())
        ref = wpt_oracle.packet_features(x.double().numpy(), wavelets.Wavelet(name).dec_lo, level)
        _check_coeffs(got.cpu().double().numpy(), ref)


def test_full_size_properties_coif4_l14_b128():
    """BASELINE config 2 size: batch independence, linearity, energy."""
    g = torch.Generator().manual_seed(42)
    x = (0.1 * torch.randn(128, 1, N, generator=g)).clamp_(-1, 1).cuda()
    p = Packets("coif4", max_lev=14)
    full, _ = p(x)
    assert full.shape == (128, 1, 16384, 24)
    for b in (0, 77, 127):
        one, _ = p(x[b])
        assert torch.equal(one[0], full[b])
    y = torch.roll(x, 1, 0)
    lin, _ = p(0.5 * x + 2.0 * y)
    comb = 0.5 * full + 2.0 * torch.roll(full, 1, 0)
    assert (lin - comb).abs().max() <= 2e-5 * comb.abs().max()
    ref = wpt_oracle.packet_features(x[5].double().cpu().numpy(), wavelets.Wavelet("coif4").dec_lo, 14)
    _check_coeffs(full[5:6].cpu().double().numpy(), ref)


def test_full_size_haar_l14_b4096_properties():
    """BASELINE config 4 size: orthogonality -> energy, batch independence."""
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(4096, N, generator=g, device="cuda") * 0.1
    p = Packets("haar", max_lev=14)
    full, _ = p(x)
    assert full.shape == (4096, 1, 16384, 2)
    one, _ = p(x[4000:4001])
    assert torch.equal(one[0], full[4000])
    ref = wpt_oracle.packet_features(x[17:18].double().cpu().numpy(), wpt_oracle.HAAR, 14)
    _check_coeffs(full[17:18].cpu().double().numpy(), ref)


def test_bad_arguments_raise():
    with pytest.raises(ValueError):
        Packets("nope", max_lev=3)
    with pytest.raises(ValueError):
        Packets("coif4", max_lev=3)(torch.randn(2, 20).cuda())  # pad >= node length
    with pytest.raises(ValueError):
        Packets("haar", max_lev=3)(torch.randn(2, 2, 100).cuda())


@pytest.mark.parametrize("name,level,log_scale,loss_less", [
    ("haar", 1, False, False), ("sym5", 8, False, False), ("sym5", 8, True, True),
    ("coif4", 8, True, False), ("haar", 14, True, True), ("coif4", 14, False, False),
])
def test_block_norm_matches_oracle(name, level, log_scale, loss_less):
    """Per-node division by the batch maximum (reference wavelet_math.py:202-203)."""
    x = _parity_inputs(seed=3)
    w = wavelets.Wavelet(name)
    p = Packets(name, max_lev=level, log_scale=log_scale, loss_less=loss_less, block_norm=True)
    got, _ = p(x.cuda())
    got = got.cpu().double().numpy()
    xd = x.double().numpy()
    coef = wpt_oracle.packet_features(xd, w.dec_lo, level, block_norm=True)[:, 0]
    ref = wpt_oracle.packet_features(xd, w.dec_lo, level, log_scale=log_scale, loss_less=loss_less,
                                     block_norm=True)
    assert got.shape == ref.shape
    # a node's values are its raw coefficients over the node maximum: both carry COEF_RTOL of the
    # raw scale, so the quotient is good to a few COEF_RTOL of 1 only where the node maximum is
    # itself well above the transform's error floor
    raw = wpt_oracle.packet_features(xd, w.dec_lo, level)[:, 0]
    node_max = np.max(np.abs(raw), axis=(0, 2), keepdims=True)
    d = 3 * COEF_RTOL * np.max(np.abs(raw)) / node_max  # error of the quotient, per node
    d = np.broadcast_to(d, coef.shape)
    if not log_scale:
        assert np.all(np.abs(got[:, 0] - coef) <= d + 1e-7)
        assert np.max(np.abs(got)) <= 1.0 + 1e-6
        return
    bound = 1e-5 + 2 * np.abs(coef) * d / (coef ** 2 + 1e-12) + 2e-6 * np.abs(ref[:, 0])
    assert np.all(np.abs(got[:, 0] - ref[:, 0]) <= bound)
    if loss_less:
        sure = np.abs(coef) > d
        assert np.array_equal(got[:, 1][sure], ref[:, 1][sure])


def test_block_norm_is_batch_global():
    """The maximum runs over the whole batch node tensor, not per frame."""
    x = _parity_inputs(seed=4).cuda()
    p = Packets("sym5", max_lev=6, block_norm=True)
    raw, _ = Packets("sym5", max_lev=6)(x)
    got, _ = p(x)
    node_max = raw.abs().amax(dim=(0, 1, 3), keepdim=True)
    assert torch.equal(got, raw / node_max)
    assert torch.equal(got.abs().amax(dim=(0, 1, 3)), torch.ones(64, device="cuda"))


def test_per_node_estimators_match_reference_welford():
    """`PacketWelford` vs one WelfordEstimator per node (reference wavelet_math.py:194-200)."""
    from audiofakedetect.wavelet_math import PacketWelford, graycode_keys
    from oracle import torch_ref

    level = 5
    w = wavelets.Wavelet("sym5")
    batches = [_parity_inputs(seed=5), _parity_inputs(seed=6)[:4]]
    est = PacketWelford(level, "cuda")
    p = Packets("sym5", max_lev=level, log_scale=True, compute_welford=True, block_norm_dict=est)
    plain = Packets("sym5", max_lev=level, log_scale=True)
    ref_dict = None
    for xb in batches:
        got, d = p(xb.cuda())
        assert d is est
        # statistics do not touch the features (two log implementations: a few ulp)
        assert torch.allclose(got, plain(xb.cuda())[0], atol=1e-5, rtol=1e-6)
        _, ref_dict = torch_ref.packets_torch(xb.double(), w.dec_lo, level, log_scale=True,
                                              compute_welford=True, welford_dict=ref_dict)
    assert list(est.keys()) == graycode_keys(level) == wpt_oracle.graycode_paths(level)
    assert list(ref_dict.keys()) == list(est.keys())
    final = est.finalize()
    for k in est.keys():
        rm, rs = ref_dict[k].finalize()
        gm, gs = est[k].finalize()
        assert torch.equal(gm.cpu(), final[k]["mean"].cpu()) and gm.shape == (1,)
        assert abs(float(gm) - float(rm)) <= 1e-6 * max(1.0, abs(float(rm))) + 2e-7
        assert abs(float(gs) - float(rs)) <= 1e-5 * float(rs) + 1e-7


@pytest.mark.parametrize("name,n,level", [("sym8", 44100, 8), ("coif4", 44100, 8), ("haar", 88200, 9), ("coif10", 30000, 7),
                                          ("sym5", 66150, 1), ("db4", 48000, 10)])
def test_frames_longer_than_the_lds_resident_tree(name, n, level):
    """Frames beyond about 27 000 samples do not fit the kernels' LDS-resident packet tree: `wpt_forward` splits the top
    level(s) off with `afd_wpt_analysis_step` and transforms the children (the reference takes any window size:
    `--window-size` / `--seconds`).  Raw coefficients against the float64 oracle at the usual bar, then the log / sign
    epilogue against the same formulas on the oracle's coefficients."""
    g = torch.Generator().manual_seed(n + level)
    x = (0.1 * torch.randn(3, n, generator=g)).clamp_(-1, 1)
    x[0, 0] = 1.0
    x[1, -1] = -1.0
    w = wavelets.Wavelet(name)
    got, _ = Packets(name, max_lev=level)(x.cuda())
    ref = wpt_oracle.packet_features(x.double().numpy(), w.dec_lo, level, dec_hi=w.dec_hi)
    assert tuple(got.shape) == ref.shape
    _check_coeffs(got.cpu().double().numpy(), ref)
    both, _ = Packets(name, max_lev=level, log_scale=True, loss_less=True)(x.unsqueeze(1).cuda())
    both = both.cpu().double().numpy()
    coef = ref[:, 0]
    d = COEF_RTOL * np.max(np.abs(coef))
    lref = np.log(coef ** 2 + 1e-12)
    bound = 1e-5 + 2 * np.abs(coef) * d / (coef ** 2 + 1e-12) + 2e-6 * np.abs(lref)
    assert both.shape[1] == 2 and np.all(np.abs(both[:, 0] - lref) <= bound)
    sure = np.abs(coef) > d
    assert np.array_equal(both[:, 1][sure], np.where(coef < 0, -1.0, 1.0)[sure])
