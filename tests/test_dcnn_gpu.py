"""GPU parity of the DCNN model plugin against vectors produced by the reference classes.

Fixtures tests/golden/dcnn_*.pt come from the reference's DCNN (models.py:240-313) run on
CPU (tests/golden/make_golden.py).  Bars: eval logits within 1e-4, class labels bit-exact
(BASELINE.json north_star); train step (dropout p=0): loss within 1e-5, gradients within
1e-4 of the largest gradient entry, Adam-updated parameters within 2e-6.
"""

import os

import pytest
import torch

from audiofakedetect import ops
from audiofakedetect.models import DCNN, get_model, strip_ddp_prefix
from audiofakedetect.utils import DotDict

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _args(input_dim, time_dim_add=0, flattend_size=320, **kw):
    a = DotDict(input_dim=list(input_dim), ochannels1=64, ochannels2=64, ochannels3=96,
                ochannels4=128, ochannels5=32, kernel1=3, dropout_cnn=0.6, dropout_lstm=0.2,
                time_dim_add=time_dim_add, flattend_size=flattend_size, ddp=False)
    a.update(kw)
    return a


@pytest.mark.parametrize("name", ["shipped_stft", "shipped_packetssym5", "random_stft",
                                  "random_sym5", "random_coif4"])
def test_eval_logits_and_labels(name):
    g = torch.load(os.path.join(GOLD, f"dcnn_{name}.pt"), map_location="cpu")
    x = g["x"]
    net = DCNN(_args(x.shape, time_dim_add=g["time_dim_add"]))
    net.load_state_dict(strip_ddp_prefix(g["state_dict"]), strict=True)
    net.cuda().eval()
    with torch.no_grad():
        y = net(x.cuda())
    assert (y.cpu() - g["logits"]).abs().max().item() <= 1e-4
    assert torch.equal(y.argmax(-1).cpu(), g["labels"])


def test_packet_memory_order_input_is_consumed_without_copy():
    # packets arrive as a permuted view (memory [B,C,T,P]); stft as dense [B,C,F,T]
    g = torch.load(os.path.join(GOLD, "dcnn_random_sym5.pt"), map_location="cpu")
    x = g["x"]
    net = DCNN(_args(x.shape, time_dim_add=1))
    net.load_state_dict(g["state_dict"])
    net.cuda().eval()
    view = x.permute(0, 1, 3, 2).contiguous().cuda().permute(0, 1, 3, 2)
    with torch.no_grad():
        y = net(view)
    assert (y.cpu() - g["logits"]).abs().max().item() <= 1e-4


def test_train_step_matches_reference():
    g = torch.load(os.path.join(GOLD, "dcnn_train_step.pt"), map_location="cpu")
    net = DCNN(_args(g["x"].shape, dropout_cnn=0.0, dropout_lstm=0.0))
    net.load_state_dict(g["state_dict"], strict=True)
    net.cuda().train()
    opt = ops.FusedAdam(net.parameters(), lr=g["lr"], weight_decay=g["weight_decay"])
    opt.zero_grad()
    out = net(g["x"].cuda())
    loss = ops.CrossEntropyLoss()(out, g["labels"].cuda())
    loss.backward()
    assert (out.detach().cpu() - g["logits"]).abs().max().item() <= 1e-4
    assert abs(loss.item() - g["loss"].item()) <= 1e-5
    gmax = max(v.abs().max().item() for v in g["grads"].values())
    for k, p in net.named_parameters():
        err = (p.grad.cpu() - g["grads"][k]).abs().max().item()
        assert err <= 1e-4 * gmax, f"grad {k}: {err:.3e} (max grad {gmax:.3e})"
    opt.step()
    after = net.state_dict()
    for k, v in g["state_dict_after"].items():
        if v.dtype.is_floating_point:
            assert (after[k].cpu() - v).abs().max().item() <= 2e-6 + 1e-5 * v.abs().max().item(), k
        else:
            assert int(after[k]) == int(v), k


def test_get_model_factory_and_name():
    a = _args((4, 1, 256, 101))
    a.module = DCNN
    net = get_model(a, "modules")
    assert isinstance(net, DCNN) and net.get_name() == "DCNN"


def test_level14_coif4_shape_runs():
    # BASELINE config 2 geometry [B,1,16384,24], flattend_size 80960, time_dim 3 (small batch)
    torch.manual_seed(0)
    net = DCNN(_args((2, 1, 16384, 24), flattend_size=80960)).cuda().train()
    x = torch.randn(2, 1, 24, 16384, device="cuda").permute(0, 1, 3, 2)
    out = net(x)
    assert out.shape == (2, 2) and torch.isfinite(out).all()
    loss = ops.CrossEntropyLoss()(out, torch.tensor([0, 1], device="cuda"))
    loss.backward()
    for k, p in net.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), k
