"""GPU parity of the LCNN model (BASELINE config 5 head) against a vector produced by the
reference's own LCNN class (tests/golden/dcnn_lcnn_eval.pt, reference models.py:68-131)."""

import os
import sys

import pytest
import torch

from audiofakedetect.lcnn import LCNN, gemm_nt, max_feature_map

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, GOLD)


def test_gemm_nt_and_mfm():
    g = torch.Generator().manual_seed(0)
    a = torch.randn(200, 300, generator=g)
    b = torch.randn(130, 300, generator=g)
    bias = torch.randn(130, generator=g)
    ref = a.double() @ b.double().t() + bias.double()
    got = gemm_nt(a.cuda(), b.cuda(), bias.cuda())
    assert (got.cpu().double() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()
    got2 = gemm_nt(a.cuda(), b.cuda(), None, out=got.clone(), accumulate=True)
    assert (got2.cpu().double() - (2 * ref - bias.double())).abs().max().item() <= 5e-5 * ref.abs().max().item()
    x = torch.randn(3, 8, 5, 7, generator=g)
    xr = x.double().requires_grad_()
    yr = xr.reshape(3, 2, 4, 5, 7).max(1)[0]
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy.double())
    xg = x.cuda().requires_grad_()
    yg = max_feature_map(xg)
    yg.backward(dy.cuda())
    assert torch.equal(yg.detach().cpu().double(), yr.detach())
    assert torch.equal(xg.grad.cpu().double(), xr.grad)


def test_lcnn_eval_logits_match_reference():
    from recipes import fill_state_dict

    g = torch.load(os.path.join(GOLD, "dcnn_lcnn_eval.pt"), map_location="cpu")
    net = LCNN()
    assert {k: tuple(v.shape) for k, v in net.state_dict().items()} == g["shapes"]
    net.load_state_dict(fill_state_dict(g["shapes"]), strict=True)
    net.cuda().eval()
    with torch.no_grad():
        y = net(g["x"].cuda())
    assert (y.cpu() - g["logits"]).abs().max().item() <= 1e-4
    assert torch.equal(y.argmax(-1).cpu(), g["labels"])


def test_lcnn_lstm_backward_is_loud():
    net = LCNN().cuda().train()
    out = net(torch.randn(2, 1, 256, 101, device="cuda"))
    with pytest.raises(NotImplementedError):
        out.sum().backward()
