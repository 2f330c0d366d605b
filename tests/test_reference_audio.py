"""Plausibility of the packet transform against reference-held data (task 8).

ptwt / pywt cannot be installed here and the reference's tests pin shapes only, so the WPT oracle
stays "parity unpinned" at the ptwt boundary.  What the reference does hold
are numbers that DEPEND on ptwt's output: the weights of the shipped
`models/*packetssym5*...fbmelgan*.pt` checkpoint (trained on ptwt's sym5 level-8 log-packets in
frequency order, wavelet_math.py:182-218) and 7 distinct recordings of one LJSpeech utterance --
the original and six vocoder re-syntheses (tests/new_data/*).  If pad geometry, filter
orientation or packet order differed from what the checkpoint saw in training, it would not
separate them.  It does (48 of 49 one-second frames), and with the packets in natural instead of
Gray-code order it calls almost everything real.  This is evidence, not a pin: the statistics of
the normalisation are taken from the fixture itself, so a global scale error would go unnoticed.

Fixtures: tests/golden/ref_wavs_lj008_0217.pt (make_wav_fixture.py, int16 samples) and
tests/golden/dcnn_shipped_packetssym5.pt (make_golden.py, state_dict tensors).
"""

import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
LEVEL = 8


def _frames():
    fx = torch.load(os.path.join(GOLD, "ref_wavs_lj008_0217.pt"), map_location="cpu")
    frames, fake = [], []
    for folder, pcm in zip(fx["folders"], fx["pcm_int16"]):
        n = pcm.numel() // 22050
        frames.append(pcm[: n * 22050].reshape(n, 22050).to(torch.float32) / 32768.0)  # PCM -> [-1, 1)
        fake += [folder != fx["real_folder"]] * n
    return torch.cat(frames)[:, None, :], torch.tensor(fake)


def _weights():
    g = torch.load(os.path.join(GOLD, "dcnn_shipped_packetssym5.pt"), map_location="cpu")
    return {k.replace("module.", ""): v for k, v in g["state_dict"].items()}, g["time_dim_add"]


def _natural_order(feats):
    """Packets re-ordered so that index g holds filter path g (what get_level(order='natural') gives)."""
    p = feats.shape[2]
    out = torch.empty_like(feats)
    for f in range(p):
        out[:, :, f ^ (f >> 1), :] = feats[:, :, f, :]
    return out


def _check(predict, feats, fake):
    mean, std = feats.mean(), feats.std()
    pred = predict(((feats - mean) / std).contiguous())
    correct = (pred == fake)
    # 49 frames, 7 real / 42 re-synthesised: the checkpoint gets 48 right
    assert correct.sum().item() >= 46, correct.sum().item()
    assert correct[~fake].sum().item() >= 5 and correct[fake].sum().item() >= 40
    wrong = predict(((_natural_order(feats) - mean) / std).contiguous())
    # natural order: the re-synthesised frames are no longer recognised (6 of 42 measured)
    assert (wrong[fake]).sum().item() <= 14, wrong[fake].sum().item()


def test_shipped_sym5_checkpoint_separates_reference_recordings_on_the_oracle():
    from oracle import torch_ref, wpt_oracle

    x, fake = _frames()
    assert x.shape == (49, 1, 22050) and fake.sum().item() == 42
    feats, _ = torch_ref.packets_torch(x, wpt_oracle.TAPS["sym5"], LEVEL, log_scale=True)
    sd, add = _weights()
    net = torch_ref.DCNNRef(feats.shape, time_dim_add=add, flattend_size=320)
    net.load_state_dict(sd, strict=True)
    net.eval()

    def predict(f):
        with torch.no_grad():
            return net(f).argmax(-1) != 0

    _check(predict, feats, fake)


@pytest.mark.gpu
def test_shipped_sym5_checkpoint_separates_reference_recordings_on_the_gpu():
    from audiofakedetect.models import DCNN
    from audiofakedetect.utils import DotDict
    from audiofakedetect.wavelet_math import Packets

    x, fake = _frames()
    tr = Packets("sym5", LEVEL, log_scale=True, loss_less=False, power=2.0, block_norm=False,
                 compute_welford=False, block_norm_dict=None)
    with torch.no_grad():
        feats, _ = tr(x.cuda())
    assert tuple(feats.shape) == (49, 1, 256, 95)
    sd, add = _weights()
    net = DCNN(DotDict(input_dim=list(feats.shape), ochannels1=64, ochannels2=64, ochannels3=96,
                       ochannels4=128, ochannels5=32, kernel1=3, dropout_cnn=0.6, dropout_lstm=0.2,
                       time_dim_add=add, flattend_size=320, ddp=False))
    net.load_state_dict(sd, strict=True)
    net.cuda().eval()

    def predict(f):
        with torch.no_grad():
            return (net(f.cuda()).argmax(-1) != 0).cpu()

    _check(predict, feats.cpu(), fake)

    # and the product's features are the oracle's on this real audio (5e-6 of the largest coefficient,
    # compared before the logarithm as in tests/test_wpt_gpu.py)
    from oracle import wpt_oracle

    lin = Packets("sym5", LEVEL, log_scale=False, loss_less=False, power=2.0, block_norm=False,
                  compute_welford=False, block_norm_dict=None)
    with torch.no_grad():
        got, _ = lin(x[:8].cuda())
    want = wpt_oracle.packet_features(x[:8, 0].double().numpy(), wpt_oracle.TAPS["sym5"], LEVEL)
    want = np.asarray(want).reshape(got.shape)
    assert np.abs(got.cpu().double().numpy() - want).max() <= 5e-6 * np.abs(want).max()
