"""GPU parity of the STFT front end (matrix DFT on the f32 MFMA) against torch.stft.

torch.stft is the ATen op torchaudio's Spectrogram calls (reference wavelet_math.py:47); the
oracle restates the reference's defaults through it in float64.  Tolerances:
  power spectrum : |gpu - ref| <= 2e-5 * max(ref)   (fp32 MFMA chain over 511 terms)
  log spectrum   : |gpu - ref| <= 1e-4 + 2e-5 * max(ref) / (ref_power + 1e-12)
"""

import pytest
import torch

from audiofakedetect.wavelet_math import STFTLayer
from oracle import torch_ref

pytestmark = pytest.mark.gpu


def _x(b=3, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = (0.1 * torch.randn(b, 1, 22050, generator=g)).clamp_(-1, 1)
    t = torch.arange(22050) / 22050.0
    x[0, 0] = sum(torch.sin(2 * torch.pi * f * t) for f in (440.0, 3000.0, 7500.0, 10500.0)) / 4
    return x


def test_reference_test_shapes():
    # reference tests/test_transforms.py:20-51
    x = torch.randn(2, 1, 22050)
    out, none = STFTLayer(n_fft=512, hop_length=2)(x)
    assert out.shape == (2, 1, 257, 11026) and none is None
    out, _ = STFTLayer()(x)
    assert out.shape == (2, 1, 256, 101)


@pytest.mark.parametrize("n_fft,hop", [(511, 220), (512, 200), (255, 100), (511, 1024), (1023, 512), (64, 7), (2048, 3000)])
def test_power_spectrum_matches_torch_stft(n_fft, hop):
    x = _x()
    got, _ = STFTLayer(n_fft=n_fft, hop_length=hop)(x.cuda())
    ref = torch_ref.stft_torch(x.double(), n_fft, hop)
    assert got.shape == ref.shape
    err = (got.cpu().double() - ref).abs().max().item()
    assert err <= 2e-5 * ref.abs().max().item(), err


def test_log_scale_and_fused_norm():
    x = _x(seed=1)
    layer = STFTLayer(log_scale=True)
    got, _ = layer(x.cuda())
    refp = torch_ref.stft_torch(x.double(), 511, 220)
    ref = torch.log(refp + 1e-12)
    bound = 1e-4 + 2e-5 * refp.max() / (refp + 1e-12)
    assert bool(((got.cpu().double() - ref).abs() <= bound).all())
    layer.fused_norm = (-4.0, 3.0)
    fused, _ = layer(x.cuda())
    assert torch.allclose(fused, (got + 4.0) / 3.0, atol=1e-5, rtol=1e-5)


def test_magnitude_power_one():
    x = _x(b=2, seed=2)
    got, _ = STFTLayer(power=1.0)(x.cuda())
    ref = torch_ref.stft_torch(x.double(), 511, 220, power=1.0)
    assert (got.cpu().double() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()


def test_stft_into_dcnn_end_to_end():
    """Config 1 shape: STFT features feed the DCNN plugin (dense [B,1,F,T] -> transposed on GPU)."""
    from audiofakedetect.models import DCNN
    from audiofakedetect.utils import DotDict

    torch.manual_seed(0)
    args = DotDict(input_dim=[4, 1, 256, 101], ochannels1=64, ochannels2=64, ochannels3=96,
                   ochannels4=128, ochannels5=32, kernel1=3, dropout_cnn=0.6, dropout_lstm=0.2,
                   time_dim_add=0, flattend_size=320, ddp=False)
    net = DCNN(args)
    ref = torch_ref.DCNNRef(args.input_dim)
    ref.load_state_dict(net.state_dict())
    net.cuda().eval()
    ref.eval()
    x = _x(b=4, seed=3)
    feats, _ = STFTLayer(log_scale=True)(x.cuda())
    with torch.no_grad():
        out = net(feats)
        out_ref = ref(feats.cpu())
    assert (out.cpu() - out_ref).abs().max().item() <= 1e-4
    assert torch.equal(out.argmax(-1).cpu(), out_ref.argmax(-1))
