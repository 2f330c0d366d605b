"""GPU parity of the DCNN layer kernels (through the C ABI) against plain PyTorch references.

Each HIP op is compared with the same op computed by torch on the CPU in float64 (forward
and backward); tolerances are relative to the largest reference magnitude:
  conv / linear (fp32 MFMA / fma accumulation over K <= 1152 terms): 2e-5
  elementwise / pooling / batch-norm                                 : 1e-5
"""

import os

import pytest
import torch
import torch.nn.functional as F

from audiofakedetect import _native, ops

pytestmark = pytest.mark.gpu


def _close(got, ref, rtol, what=""):
    ref = ref.double()
    got = got.detach().cpu().double()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    scale = ref.abs().max().item() + 1e-30
    err = (got - ref).abs().max().item()
    assert err <= rtol * scale, f"{what}: err {err:.3e} scale {scale:.3e}"


CONV_CASES = [
    # n, cin, h, w, cout, k, pad, dil      (DCNN on STFT features, reference models.py:255-291)
    (2, 1, 101, 256, 64, 3, 2, 1),
    (2, 64, 51, 129, 64, 1, 0, 1),
    (2, 64, 51, 129, 96, 3, 1, 1),
    (2, 96, 25, 64, 128, 3, 1, 1),
    (2, 128, 25, 64, 32, 3, 1, 1),
    (2, 32, 25, 64, 64, 3, 1, 1),
    (2, 12, 64, 32, 12, 3, 1, 1),
    (2, 12, 64, 32, 12, 5, 2, 2),
    (2, 12, 60, 28, 12, 7, 2, 4),
    # the same stack at time_dim 13 (coif4 level 8) and on odd sizes: narrow-image direct kernel, dilnarrow.hip
    (3, 13, 64, 32, 13, 3, 1, 1),
    (3, 13, 64, 32, 13, 5, 2, 2),
    (3, 13, 60, 28, 13, 7, 2, 4),
    (2, 13, 37, 30, 13, 7, 2, 4),
    (2, 12, 33, 41, 12, 5, 2, 2),
    # level-14 shaped (wide) images: RECT tiles
    (1, 1, 24, 1500, 64, 3, 2, 1),
    (1, 64, 6, 2050, 96, 3, 1, 1),
    (1, 3, 64, 1100, 3, 7, 2, 4),
    (1, 3, 64, 1030, 3, 5, 2, 2),
    # few-channel dilated stack (time_dim = 3 at level 14): direct VALU kernels, dilconv.hip
    (2, 3, 64, 300, 3, 3, 1, 1),
    (1, 3, 60, 2044, 3, 7, 2, 4),
    (2, 4, 61, 259, 4, 7, 2, 4),
    (2, 1, 40, 70, 1, 5, 2, 2),
    (3, 2, 33, 513, 2, 7, 2, 4),
    (2, 4, 30, 64, 4, 3, 1, 1),
    # 1x1 layers: streaming GEMM kernels, conv1x1.hip (ragged pixel tiles, padded channels)
    (2, 32, 13, 300, 64, 1, 0, 1),
    (2, 48, 7, 65, 96, 1, 0, 1),
    (2, 128, 5, 33, 128, 1, 0, 1),
    (3, 20, 9, 50, 40, 1, 0, 1),
    (1, 64, 13, 1031, 64, 1, 0, 1),
    # 3x3 on wide images: compile-time-geometry kernel, conv3x3.hip (all four wave tilings,
    # both directions; border tiles, ragged right edge, padded output channels)
    (2, 32, 6, 1100, 64, 3, 1, 1),
    (1, 128, 5, 1300, 32, 3, 1, 1),
    (1, 96, 4, 1025, 128, 3, 1, 1),
    (1, 8, 3, 1024, 40, 3, 1, 1),
    (1, 64, 13, 1157, 96, 3, 1, 1),
    # 32 output channels: backward-weight with the operand roles swapped (interior + edge launches; 96 channels;
    # an image size the bias kernel's 16-byte loads do not take, i.e. the unswapped path)
    (2, 128, 6, 2052, 32, 3, 1, 1),
    (1, 96, 4, 1100, 32, 3, 1, 1),
    (1, 64, 5, 1301, 32, 3, 1, 1),
    # the same kernels with 32-column tiles (narrow level-8 / STFT / LCNN images): ragged tile rows
    # and columns, 4- and 8-row tiles
    (2, 32, 12, 32, 64, 3, 1, 1),
    (2, 64, 12, 32, 32, 3, 1, 1),
    (2, 96, 7, 40, 128, 3, 1, 1),
    (1, 32, 9, 24, 96, 3, 1, 1),
    # every other shape the few-channel matrix-core kernels accept (dilconv_mfma.hip: Cin == Cout in 5..16, (k, dilation) in
    # (3, 1) / (5, 2) / (7, 4), any padding up to the full one): plain 3x3 layers, no padding, "same" and full padding, wide
    # and odd images, 5 and 16 channels
    (2, 8, 20, 50, 8, 3, 1, 1),
    (2, 16, 17, 33, 16, 3, 0, 1),
    (2, 10, 24, 24, 10, 3, 2, 1),
    (1, 5, 40, 41, 5, 7, 2, 4),
    (2, 16, 30, 36, 16, 5, 4, 2),
    (2, 16, 30, 36, 16, 5, 8, 2),
    (1, 6, 30, 600, 6, 3, 1, 1),
    (1, 9, 45, 200, 9, 5, 2, 2),
    (2, 7, 50, 30, 7, 7, 12, 4),
    (2, 11, 31, 29, 11, 7, 24, 4),
    # LCNN shapes (models.py:85-110)
    (2, 1, 101, 256, 64, 5, 2, 1),
    (2, 48, 25, 64, 128, 3, 1, 1),
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_forward_backward(case):
    n, cin, h, w, cout, k, pad, dil = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    b = torch.randn(cout, generator=g)
    xr, wr, br = (t.double().requires_grad_() for t in (x, wt, b))
    yr = F.conv2d(xr, wr, br, padding=pad, dilation=dil)
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy.double())

    xg, wg, bg = (t.cuda().requires_grad_() for t in (x, wt, b))
    yg = ops.conv2d(xg, wg, bg, pad, dil)
    _close(yg, yr.detach(), 2e-5, "conv fwd")
    yg.backward(dy.cuda())
    _close(xg.grad, xr.grad, 2e-5, "conv dgrad")
    _close(wg.grad, wr.grad, 3e-5, "conv wgrad")
    _close(bg.grad, br.grad, 2e-5, "conv dbias")


@pytest.mark.parametrize("shape", [(2, 64, 103, 258), (3, 5, 7, 9), (1, 96, 51, 129), (2, 4, 26, 1002)])
@pytest.mark.parametrize("slope", [0.25, -0.3, None])
def test_prelu_maxpool(shape, slope):
    g = torch.Generator().manual_seed(1)
    z = torch.randn(shape, generator=g)
    zr = z.double().requires_grad_()
    ar = torch.tensor([slope if slope is not None else 1.0], dtype=torch.float64, requires_grad=True)
    act = F.prelu(zr, ar) if slope is not None else zr
    ur = F.max_pool2d(act, 2, 2)
    du = torch.randn(ur.shape, generator=g)
    ur.backward(du.double())
    zg = z.cuda().requires_grad_()
    ag = torch.tensor([slope], device="cuda", requires_grad=True) if slope is not None else None
    ug = ops.prelu_maxpool2x2(zg, ag)
    _close(ug, ur.detach(), 1e-6, "pool fwd")
    ug.backward(du.cuda())
    _close(zg.grad, zr.grad, 1e-6, "pool dz")
    if slope is not None:
        _close(ag.grad, ar.grad, 1e-4, "pool dslope")


@pytest.mark.parametrize("affine", [False, True])
@pytest.mark.parametrize("slope", [None, 0.25])
def test_batch_norm_train_and_eval(affine, slope):
    g = torch.Generator().manual_seed(2)
    x = torch.randn(4, 12, 9, 31, generator=g) * 2 + 0.7
    bn_r = torch.nn.BatchNorm2d(12, affine=affine).double()
    bn_g = torch.nn.BatchNorm2d(12, affine=affine).cuda()
    if affine:
        with torch.no_grad():
            bn_r.weight.copy_(torch.randn(12, generator=g))
            bn_r.bias.copy_(torch.randn(12, generator=g))
            bn_g.weight.copy_(bn_r.weight.float())
            bn_g.bias.copy_(bn_r.bias.float())
    xr = x.double().requires_grad_()
    ar = torch.tensor([slope or 1.0], dtype=torch.float64, requires_grad=True)
    yr = bn_r(F.prelu(xr, ar) if slope else xr)
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy.double())
    xg = x.cuda().requires_grad_()
    ag = torch.tensor([slope], device="cuda", requires_grad=True) if slope else None
    yg = ops.batch_norm(xg, bn_g, ag, sync=False)
    _close(yg, yr.detach(), 1e-5, "bn fwd")
    yg.backward(dy.cuda())
    _close(xg.grad, xr.grad, 2e-5, "bn dx")
    _close(bn_g.running_mean, bn_r.running_mean, 1e-5, "running_mean")
    _close(bn_g.running_var, bn_r.running_var, 1e-5, "running_var")
    assert int(bn_g.num_batches_tracked) == 1
    if affine:
        _close(bn_g.weight.grad, bn_r.weight.grad, 2e-5, "dgamma")
        _close(bn_g.bias.grad, bn_r.bias.grad, 2e-5, "dbeta")
    if slope:
        _close(ag.grad, ar.grad, 1e-4, "bn dslope")
    bn_r.eval()
    bn_g.eval()
    with torch.no_grad():
        ye = bn_r(F.prelu(x.double(), ar.detach()) if slope else x.double())
        yg = ops.batch_norm(x.cuda(), bn_g, ag.detach() if slope else None, sync=False)
    _close(yg, ye, 1e-5, "bn eval")


def test_dropout_permute_and_prelu_dropout():
    g = torch.Generator().manual_seed(3)
    x = torch.randn(3, 64, 12, 32, generator=g)
    y = ops.dropout_permute(x.cuda(), 0.6, training=False)
    assert torch.equal(y.cpu(), x.permute(0, 2, 1, 3).contiguous())
    xg = x.cuda().requires_grad_()
    y = ops.dropout_permute(xg, 0.6, training=True)
    kept = (y != 0).float().mean().item()
    assert abs(kept - 0.4) < 0.02
    yp = y.permute(0, 2, 1, 3)
    mask = (yp != 0)
    assert torch.allclose(yp[mask], xg.detach()[mask] / 0.4, rtol=1e-6)
    dy = torch.randn_like(y)
    y.backward(dy)
    assert torch.allclose(xg.grad, dy.permute(0, 2, 1, 3) * mask / 0.4, rtol=1e-6)

    z = torch.randn(3, 12, 40, 8, generator=g)
    a = torch.tensor([0.25], device="cuda", requires_grad=True)
    zg = z.cuda().requires_grad_()
    y = ops.prelu_dropout(zg, a, 0.2, training=False)
    _close(y, F.prelu(z.double(), torch.tensor([0.25], dtype=torch.float64)), 1e-6, "prelu")
    y = ops.prelu_dropout(zg, a, 0.2, training=True)
    assert abs((y != 0).float().mean().item() - 0.8) < 0.03
    zr = z.double().requires_grad_()
    ar = torch.tensor([0.25], dtype=torch.float64, requires_grad=True)
    mask = (y != 0).cpu().double() / 0.8
    (F.prelu(zr, ar) * mask).sum().backward()
    y.sum().backward()
    _close(zg.grad, zr.grad, 1e-6, "prelu dz")
    _close(a.grad, ar.grad, 1e-4, "prelu dslope")


@pytest.mark.parametrize("td,f", [(12, 320), (3, 5000), (1, 777)])
def test_linear_mean(td, f):
    g = torch.Generator().manual_seed(4)
    x = torch.randn(5, td, f, generator=g)
    w = torch.randn(2, f, generator=g) / f ** 0.5
    b = torch.randn(2, generator=g)
    xr, wr, br = (t.double().requires_grad_() for t in (x, w, b))
    yr = (xr @ wr.t() + br).mean(1)
    dy = torch.randn(5, 2, generator=g)
    yr.backward(dy.double())
    xg, wg, bg = (t.cuda().requires_grad_() for t in (x, w, b))
    yg = ops.linear_mean(xg, wg, bg)
    _close(yg, yr.detach(), 1e-5, "linear fwd")
    yg.backward(dy.cuda())
    _close(xg.grad, xr.grad, 1e-5, "linear dx")
    _close(wg.grad, wr.grad, 1e-5, "linear dw")
    _close(bg.grad, br.grad, 1e-5, "linear db")


def test_cross_entropy_and_accuracy():
    g = torch.Generator().manual_seed(5)
    logits = torch.randn(128, 2, generator=g) * 3
    labels = torch.randint(0, 2, (128,), generator=g)
    lr = logits.double().requires_grad_()
    loss_r = F.cross_entropy(lr, labels)
    loss_r.backward()
    lg = logits.cuda().requires_grad_()
    loss, stats = ops.cross_entropy(lg, labels.cuda())
    _close(loss, loss_r.detach(), 1e-6, "ce")
    (loss * 2.0).backward()
    _close(lg.grad, 2.0 * lr.grad, 1e-5, "ce grad")
    assert int(stats[1].item()) == int((logits.argmax(-1) == labels).sum())


def test_fused_adam_matches_torch_adam():
    g = torch.Generator().manual_seed(6)
    shapes = [(64, 1, 3, 3), (64,), (1,), (2, 320)]
    ps_r = [torch.randn(s, generator=g).requires_grad_() for s in shapes]
    ps_g = [p.detach().clone().cuda().requires_grad_() for p in ps_r]
    opt_r = torch.optim.Adam(ps_r, lr=4e-4, weight_decay=1e-3)
    opt_g = ops.FusedAdam(ps_g, lr=4e-4, weight_decay=1e-3)
    for _ in range(3):
        grads = [torch.randn(s, generator=g) for s in shapes]
        opt_r.zero_grad()
        opt_g.zero_grad()
        for p, q, gr in zip(ps_r, ps_g, grads):
            p.grad = gr.clone()
            q.grad = gr.clone().cuda()
        opt_r.step()
        opt_g.step()
    for p, q in zip(ps_r, ps_g):
        assert torch.allclose(q.detach().cpu(), p.detach(), atol=1e-6, rtol=1e-5)


def test_normalize_and_transpose():
    x = torch.randn(2, 1, 95, 256).cuda()
    view = x.permute(0, 1, 3, 2)
    y = ops.normalize_forward(view, -3.0, 2.5)
    assert y.shape == view.shape
    assert torch.allclose(y, (view + 3.0) / 2.5, atol=1e-6)
    t = ops.transpose_last2(x)
    assert torch.equal(t, x.transpose(-1, -2).contiguous())


# (the two wide shapes: rows of more than 1024 pooled columns -- the forward's four pixels per thread with a ragged last
# thread, several workgroup columns per row in the backward; 24 and 64 channels)
@pytest.mark.parametrize("shape,cout,pad", [((3, 1, 24, 300), 64, 2), ((2, 1, 109, 256), 64, 2),
                                            ((2, 1, 11, 37), 20, 1), ((2, 1, 6, 2053), 24, 2), ((2, 1, 7, 4099), 64, 1)])
@pytest.mark.parametrize("slope", [0.25, -0.4])
def test_fused_conv1_prelu_pool(shape, cout, pad, slope):
    """Single-channel first block fused (conv 3x3 + PReLU + MaxPool2d) vs the three torch ops."""
    g = torch.Generator().manual_seed(sum(shape) + cout)
    x = torch.randn(shape, generator=g)
    w = torch.randn(cout, 1, 3, 3, generator=g) / 3
    b = torch.randn(cout, generator=g)
    wr, br = w.double().requires_grad_(), b.double().requires_grad_()
    ar = torch.tensor([slope], dtype=torch.float64, requires_grad=True)
    ur = F.max_pool2d(F.prelu(F.conv2d(x.double(), wr, br, padding=pad), ar), 2, 2)
    du = torch.randn(ur.shape, generator=g)
    ur.backward(du.double())
    wg, bg = w.cuda().requires_grad_(), b.cuda().requires_grad_()
    ag = torch.tensor([slope], device="cuda", requires_grad=True)
    ug = ops.conv1_prelu_maxpool(x.cuda(), wg, bg, ag, pad)
    _close(ug, ur.detach(), 2e-6, "conv1 fused fwd")
    ug.backward(du.cuda())
    _close(wg.grad, wr.grad, 2e-5, "conv1 fused dw")
    _close(bg.grad, br.grad, 2e-5, "conv1 fused db")
    _close(ag.grad, ar.grad, 1e-4, "conv1 fused dslope")


def test_scalar_moments_match_welford_and_torch():
    # calc_normalization's statistics (wavelet_math.py:387-452): fused double-precision
    # reduction vs the reference's Welford update chain and vs torch in float64
    from audiofakedetect.data_loader import WelfordEstimator

    g = torch.Generator().manual_seed(5)
    batches = [(-12.0 + 5.0 * torch.randn(3, 1, 257, 41, generator=g)) for _ in range(4)]
    mom = ops.ScalarMoments(torch.device("cuda"))
    wel = WelfordEstimator()
    for b in batches:
        mom.update(b.cuda())
        mom.update(b.cuda()[..., 1:])  # non-contiguous view, odd element count
        wel.update(b.permute(0, 3, 2, 1))
        wel.update(b[..., 1:].permute(0, 3, 2, 1))
    mean, std = mom.finalize()
    allv = torch.cat([torch.cat([b.reshape(-1), b[..., 1:].reshape(-1)]) for b in batches]).double()
    assert abs(mean.item() - allv.mean().item()) <= 1e-6 * abs(allv.mean().item())
    assert abs(std.item() - allv.std(unbiased=False).item()) <= 1e-6 * allv.std().item()
    wm, ws = wel.finalize()
    assert abs(mean.item() - wm.item()) <= 1e-4 and abs(std.item() - ws.item()) <= 1e-4


@pytest.mark.parametrize("shape", [(1, 64, 7, 1027, 96), (2, 32, 6, 1100, 64), (1, 16, 5, 1025, 32)])
def test_conv_feeding_maxpool_skips_the_unused_row_and_column(shape):
    # conv2d(pooled=True) + prelu_maxpool2x2 against conv -> PReLU -> MaxPool2d(2, 2) in float64:
    # the odd last row / column of the conv output is neither produced nor differentiated
    n, cin, h, w, cout = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    b = torch.randn(cout, generator=g)
    slope = torch.tensor([0.25])
    xr, wr, br, sr = (t.double().requires_grad_() for t in (x, wt, b, slope))
    yr = F.max_pool2d(F.prelu(F.conv2d(xr, wr, br, padding=1), sr), 2, 2)
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy.double())
    xg, wg, bg, sg = (t.cuda().requires_grad_() for t in (x, wt, b, slope))
    yg = ops.prelu_maxpool2x2(ops.conv2d(xg, wg, bg, 1, 1, pooled=True), sg)
    _close(yg, yr.detach(), 2e-5, "conv+pool fwd")
    yg.backward(dy.cuda())
    _close(xg.grad, xr.grad, 2e-5, "dgrad")
    _close(wg.grad, wr.grad, 3e-5, "wgrad")
    _close(bg.grad, br.grad, 2e-5, "dbias")
    _close(sg.grad, sr.grad, 1e-4, "dslope")


@pytest.mark.parametrize("shape,cout,training", [
    ((4, 64, 13, 257), 64, True), ((3, 64, 5, 1031), 64, True), ((2, 32, 7, 130), 96, True),
    ((4, 64, 13, 257), 64, False),
])
def test_batchnorm_folded_into_the_1x1_convolution(shape, cout, training):
    """BatchNorm2d(affine=False) -> Conv2d(k=1) in one pass per direction (reference
    models.py:260-262) against the same two layers in float64: output, input gradient (the
    BatchNorm backward rebuilt from the 1x1 weight / bias gradients), weight and bias gradients,
    running statistics."""
    torch.manual_seed(5)
    n, c, h, w = shape
    # post-pool activations: positive, mean comparable to the spread
    u = (torch.randn(shape).abs() * 0.7 + 0.3 * torch.rand(1, c, 1, 1)).cuda().requires_grad_(True)
    conv = torch.nn.Conv2d(c, cout, 1).cuda()
    bn = torch.nn.BatchNorm2d(c, affine=False).cuda()
    bn.train(training)
    if not training:
        bn.running_mean.copy_(torch.rand(c) * 0.5)
        bn.running_var.copy_(torch.rand(c) + 0.5)
    assert ops.bn_conv1x1_applicable(bn, conv)
    ref_bn = torch.nn.BatchNorm2d(c, affine=False).double().cuda()
    ref_bn.load_state_dict(bn.state_dict())
    ref_bn.train(training)
    ref_conv = torch.nn.Conv2d(c, cout, 1).double().cuda()
    ref_conv.load_state_dict(conv.state_dict())
    u64 = u.detach().double().requires_grad_(True)
    dz = torch.randn(n, cout, h, w, device="cuda")

    z = ops.bn_conv1x1(u, bn, conv.weight, conv.bias, sync=False)
    z.backward(dz)
    zr = ref_conv(ref_bn(u64))
    zr.backward(dz.double())

    def close(a, b, tol):
        scale = b.abs().max().item()
        assert (a.double() - b).abs().max().item() <= tol * scale, ((a.double() - b).abs().max().item(), scale)

    close(z, zr, 2e-6)
    close(u.grad, u64.grad, 2e-5)
    close(conv.weight.grad, ref_conv.weight.grad, 2e-5)
    close(conv.bias.grad, ref_conv.bias.grad, 2e-6)
    if training:
        close(bn.running_mean, ref_bn.running_mean, 1e-6)
        close(bn.running_var, ref_bn.running_var, 1e-6)
        assert int(bn.num_batches_tracked) == 1


@pytest.mark.gpu
@pytest.mark.parametrize("shape,cout,linked", [((2, 64, 13, 1157), 64, True), ((3, 64, 5, 333), 64, False),
                                               ((2, 48, 3, 130), 40, False), ((1, 64, 2, 31), 64, True)])
def test_block2_one_pass_backward(shape, cout, linked):
    """DCNN block 2 in training mode, u -> BatchNorm -> Conv2d(k=1) -> PReLU -> BatchNorm (reference
    models.py:260-264), with its backward in one pass over the activations (`afd_conv1x1_prelu_bn_backward`)
    against the four layers in float64: output, weight / bias / slope gradients, running statistics and the
    input gradient -- directly, or (`linked`) as the (t, alpha, beta) hand-over that the first block's backward
    consumes (`afd_conv1_pool_backward_affine`).  Shapes cover whole and partial 64-pixel tiles, unaligned rows
    and fewer than 64 channels."""
    torch.manual_seed(11)
    n, c, h, w = shape
    u = (torch.randn(shape).abs() * 0.7 + 0.3 * torch.rand(1, c, 1, 1)).cuda().requires_grad_(True)
    conv = torch.nn.Conv2d(c, cout, 1).cuda()
    bn1 = torch.nn.BatchNorm2d(c, affine=False).cuda().train()
    bn2 = torch.nn.BatchNorm2d(cout, affine=False).cuda().train()
    act = torch.nn.PReLU().cuda()
    assert ops.bn_conv1x1_prelu_bn_applicable(bn1, conv, bn2)
    ref = torch.nn.Sequential(torch.nn.BatchNorm2d(c, affine=False), torch.nn.Conv2d(c, cout, 1), torch.nn.PReLU(),
                              torch.nn.BatchNorm2d(cout, affine=False)).double().cuda().train()
    ref[1].load_state_dict(conv.state_dict())
    u64 = u.detach().double().requires_grad_(True)
    g = torch.randn(n, cout, h, w, device="cuda")

    link = {} if linked else None
    y = ops.bn_conv1x1_prelu_bn(u, bn1, conv.weight, conv.bias, act.weight, bn2, sync=False, link=link)
    y.backward(g)
    yr = ref(u64)
    yr.backward(g.double())

    def close(a, b, tol):
        scale = b.abs().max().item()
        assert (a.double() - b).abs().max().item() <= tol * scale, ((a.double() - b).abs().max().item(), scale)

    close(y, yr, 5e-6)
    du = u.grad
    if linked:
        alpha, beta = link.pop("affine")
        du = du + alpha.view(1, -1, 1, 1) * u.detach() + beta.view(1, -1, 1, 1)
    close(du, u64.grad, 5e-5)
    close(conv.weight.grad, ref[1].weight.grad, 5e-5)
    close(conv.bias.grad, ref[1].bias.grad, 5e-5)
    close(act.weight.grad, ref[2].weight.grad, 5e-5)
    close(bn1.running_mean, ref[0].running_mean, 1e-6)
    close(bn2.running_var, ref[3].running_var, 1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("geometry", [(2, 10, 300, 24, 2), (2, 6, 2055, 64, 2)])
def test_conv1_pool_backward_with_affine_gradient(geometry):
    """`afd_conv1_pool_backward_affine`: the first block's backward with its incoming gradient given as
    du + alpha[c] * u + beta[c] equals the plain backward on the materialised sum."""
    torch.manual_seed(12)
    lib = _native.load()
    n, h, w, cout, pad = geometry
    x = torch.randn(n, 1, h, w, device="cuda")
    conv = torch.nn.Conv2d(1, cout, 3, padding=pad).cuda()
    slope = torch.full((1,), 0.25, device="cuda")
    hp, wp = (h + 2 * pad - 2) // 2, (w + 2 * pad - 2) // 2
    u = torch.empty(n, cout, hp, wp, device="cuda")
    idx = torch.empty(n, cout, hp, wp, dtype=torch.uint8, device="cuda")
    _native.check(lib.afd_conv1_pool_forward(_native.ptr(x), _native.ptr(conv.weight.detach().contiguous()),
                                             _native.ptr(conv.bias.detach()), _native.ptr(slope), _native.ptr(u),
                                             _native.ptr(idx), None, None, 0, n, h, w, cout, pad, _native.stream_ptr()), "fwd")
    # the same launch with the BatchNorm batch sums in its epilogue (round 5): identical u / idx, sums equal to float64
    # sums of u (ragged last workgroup column, 24 channels)
    u2, idx2 = torch.empty_like(u), torch.empty_like(idx)
    sums = torch.empty(2 * cout + 1, dtype=torch.float64, device="cuda")
    sws = torch.empty(lib.afd_conv1_pool_stats_workspace_bytes(n, h, w, cout, pad), dtype=torch.uint8, device="cuda")
    _native.check(lib.afd_conv1_pool_forward(_native.ptr(x), _native.ptr(conv.weight.detach().contiguous()),
                                             _native.ptr(conv.bias.detach()), _native.ptr(slope), _native.ptr(u2),
                                             _native.ptr(idx2), _native.ptr(sums), _native.ptr(sws), sws.numel(), n, h, w,
                                             cout, pad, _native.stream_ptr()), "fwd + sums")
    assert torch.equal(u2, u) and torch.equal(idx2, idx)
    want1, want2 = u.double().sum((0, 2, 3)), (u.double() ** 2).sum((0, 2, 3))
    assert (sums[:cout] - want1).abs().max().item() <= 2e-6 * u.double().abs().sum((0, 2, 3)).max().item()
    assert (sums[cout:2 * cout] - want2).abs().max().item() <= 2e-6 * want2.max().item()
    du = torch.randn_like(u)
    alpha, beta = torch.randn(cout, device="cuda"), torch.randn(cout, device="cuda")
    ws = torch.empty(lib.afd_conv1_pool_workspace_bytes(n, h, w, cout, pad), dtype=torch.uint8, device="cuda")

    def run(grad, a, b):
        dw = torch.empty(cout, 1, 3, 3, device="cuda")
        db = torch.empty(cout, device="cuda")
        ds = torch.zeros(1, device="cuda")
        _native.check(lib.afd_conv1_pool_backward_affine(
            _native.ptr(x), _native.ptr(grad), _native.ptr(idx), _native.ptr(u), _native.ptr(slope), _native.ptr(a),
            _native.ptr(b), _native.ptr(dw), _native.ptr(db), _native.ptr(ds), n, h, w, cout, pad, _native.ptr(ws),
            ws.numel(), _native.stream_ptr()), "bwd")
        return dw, db, ds

    full = du + alpha.view(1, -1, 1, 1) * u + beta.view(1, -1, 1, 1)
    for got, want in zip(run(du, alpha, beta), run(full, None, None)):
        assert (got - want).abs().max().item() <= 2e-5 * want.abs().max().item()


def test_conv1x1_epilogue_returns_batchnorm_sums():
    """`afd_conv1x1_forward_stats`: the output equals the plain launch bit for bit and the per-channel sums of
    PReLU(y) and its square match float64 sums of that output."""
    torch.manual_seed(13)
    lib = _native.load()
    slope = torch.full((1,), 0.25, device="cuda")

    # (the 64 -> 64 launch is the two-pass kernel whose stores leave under the next pass's matrix instructions: the last
    # case gives every wave several tiles, so rows of one tile leave during the next tile's first pass)
    for cin, co, hw, n in ((64, 64, 13 * 1157, 2), (32, 40, 999, 2), (64, 96, 4097, 2), (64, 64, 13 * 8193, 6)):
        xx = torch.randn(n, cin, hw, device="cuda")
        c1 = torch.nn.Conv2d(cin, co, 1).cuda()
        w1, b1 = c1.weight.detach().reshape(co, cin).contiguous(), c1.bias.detach()
        y0 = torch.empty(n, co, hw, device="cuda")
        wsz = lib.afd_conv2d_workspace_bytes(n, cin, 1, hw, co, 1, 0, 1)
        ws0 = torch.empty(max(wsz, 16), dtype=torch.uint8, device="cuda")
        _native.check(lib.afd_conv2d_forward(_native.ptr(xx), _native.ptr(w1), _native.ptr(b1), _native.ptr(y0), n,
                                             cin, 1, hw, co, 1, 0, 1, _native.ptr(ws0), ws0.numel(),
                                             _native.stream_ptr()), "conv")
        y1 = torch.empty_like(y0)
        s2 = torch.empty(2 * co + 1, dtype=torch.float64, device="cuda")
        ws1 = torch.empty(lib.afd_conv1x1_forward_stats_workspace_bytes(co), dtype=torch.uint8, device="cuda")
        _native.check(lib.afd_conv1x1_forward_stats(
            _native.ptr(xx), _native.ptr(w1), _native.ptr(b1), _native.ptr(slope), _native.ptr(y1), _native.ptr(s2),
            n, cin, co, hw, _native.ptr(ws1), ws1.numel(), _native.stream_ptr()), "conv stats")
        assert torch.equal(y0, y1)
        pz = torch.where(y0 > 0, y0, 0.25 * y0).double()
        ref = torch.cat([pz.sum((0, 2)), (pz ** 2).sum((0, 2))])
        err = (s2[:2 * co] - ref).abs()
        assert (err <= 3e-6 * ref.abs() + 1e-4).all(), err.max().item()


@pytest.mark.parametrize("shape,cout,act", [((2, 64, 13, 1157), 96, True), ((1, 96, 6, 1030), 128, False),
                                            ((2, 64, 4, 200), 96, True), ((1, 64, 3, 131), 96, False)])
def test_backward_data_with_batchnorm_sums(shape, cout, act):
    """`afd_conv3x3_backward_data_bnstats`: dx equals the plain backward-data launch bit for bit, and the sums of
    dx and dx * xhat (xhat the convolution's input) match float64 sums; then end to end through autograd:
    BatchNorm -> 3x3 convolution with the linked backward equals the unlinked one."""
    torch.manual_seed(14)
    lib = _native.load()
    n, cin, h, w = shape
    assert lib.afd_conv3x3_backward_data_bnstats_applicable(cin, h, w, cout)
    dy = torch.randn(n, cout, h, w, device="cuda")
    wt = (torch.randn(cout, cin, 3, 3, device="cuda") * 0.05).contiguous()
    xhat = torch.randn(shape, device="cuda")
    slope = torch.full((1,), 0.25, device="cuda") if act else None
    ws = torch.empty(lib.afd_conv2d_workspace_bytes(n, cin, h, w, cout, 3, 1, 1), dtype=torch.uint8, device="cuda")
    dx0 = torch.empty(shape, device="cuda")
    _native.check(lib.afd_conv2d_backward_data(_native.ptr(dy), _native.ptr(wt), _native.ptr(dx0), n, cin, h, w, cout,
                                               3, 1, 1, _native.ptr(ws), ws.numel(), _native.stream_ptr()), "dgrad")
    dx1 = torch.empty_like(dx0)
    sums = torch.empty(2 * cin, dtype=torch.float64, device="cuda")
    sws = torch.empty(lib.afd_conv3x3_backward_data_bnstats_workspace_bytes(n, cin, h, w), dtype=torch.uint8,
                      device="cuda")
    _native.check(lib.afd_conv3x3_backward_data_bnstats(
        _native.ptr(dy), _native.ptr(wt), _native.ptr(dx1), _native.ptr(xhat), _native.ptr(sums), n, cin, h, w, cout,
        _native.ptr(ws), ws.numel(), _native.ptr(sws), sws.numel(), _native.stream_ptr()), "dgrad + sums")
    assert torch.equal(dx0, dx1)
    ref = torch.cat([dx0.double().sum((0, 2, 3)), (dx0.double() * xhat.double()).sum((0, 2, 3))])
    scale = (dx0.double().abs() * xhat.double().abs()).sum((0, 2, 3)).max().item()
    assert (sums - ref).abs().max().item() <= 2e-6 * scale
    bn_in = xhat

    # the same launch without the activations (where the kernel allows it): sum(dx * xhat) is left to
    # afd_conv_weight_dot -- for a convolution, sum_px dx[c] x[c] = sum_{co,k} w[co][c][k] dw[co][c][k]
    if not lib.afd_conv3x3_backward_data_bnstats_needs_input(cin, h, w, cout):
        dx2 = torch.empty_like(dx0)
        sums2 = torch.full((2 * cin,), float("nan"), dtype=torch.float64, device="cuda")
        _native.check(lib.afd_conv3x3_backward_data_bnstats(
            _native.ptr(dy), _native.ptr(wt), _native.ptr(dx2), None, _native.ptr(sums2), n, cin, h, w, cout,
            _native.ptr(ws), ws.numel(), _native.ptr(sws), sws.numel(), _native.stream_ptr()), "dgrad + sum(dx)")
        assert torch.equal(dx0, dx2)
        assert (sums2[:cin] - ref[:cin]).abs().max().item() <= 2e-6 * scale
        assert (sums2[cin:] == 0).all()
        dw = torch.empty_like(wt)
        _native.check(lib.afd_conv2d_backward_weight(
            _native.ptr(xhat), _native.ptr(dy), _native.ptr(dw), None, n, cin, h, w, cout, 3, 1, 1,
            _native.ptr(ws), ws.numel(), _native.stream_ptr()), "wgrad")
        _native.check(lib.afd_conv_weight_dot(_native.ptr(wt), _native.ptr(dw), cout, cin, 9,
                                              sums2.data_ptr() + 8 * cin, _native.stream_ptr()), "weight dot")
        assert (sums2[cin:] - ref[cin:]).abs().max().item() <= 2e-5 * scale

    # autograd: BatchNorm -> Conv2d(3x3) with and without the link
    bn = torch.nn.BatchNorm2d(cin, affine=False).cuda().train()
    conv = torch.nn.Conv2d(cin, cout, 3, padding=1).cuda()
    grads = []
    for linked in (False, True):
        z = bn_in.clone().requires_grad_(True)
        link = {} if linked else None
        y = ops.conv2d(ops.batch_norm(z, bn, slope, False, link), conv.weight, conv.bias, 1, 1, bn_link=link)
        y.backward(dy)
        grads.append(z.grad)
        assert not linked or "bwd_sums" not in link  # consumed by the BatchNorm's backward
    _close(grads[1], grads[0].cpu(), 2e-6, "linked BatchNorm backward")


@pytest.mark.parametrize("shape,cout", [((2, 64, 13, 1157), 96), ((2, 64, 12, 1028), 96), ((1, 32, 6, 517), 64),
                                        ((2, 64, 4, 200), 96), ((1, 64, 7, 256), 96), ((1, 32, 3, 64), 64)])
def test_conv_pool_backward_from_the_pooled_gradient(shape, cout):
    """Conv3x3 -> PReLU -> MaxPool2d(2, 2) backward without the dense gradient (`afd_prelu_pool_backward_compact`,
    `afd_conv3x3_backward_data_bnstats_pooled`, `afd_conv3x3_backward_weight_pooled`): dx is the dense path's bit for bit,
    the weight / bias gradients and the BatchNorm sums agree to rounding; then through autograd with the switch on / off."""
    import os
    torch.manual_seed(41)
    lib = _native.load()
    n, cin, h, w = shape
    assert lib.afd_conv3x3_pooled_backward_applicable(cin, h, w, cout)
    x = torch.randn(shape, device="cuda")
    conv = torch.nn.Conv2d(cin, cout, 3, padding=1).cuda()
    wt, bias = conv.weight.detach().contiguous(), conv.bias.detach()
    slope = torch.full((1,), 0.25, device="cuda")
    hp, wp = h // 2, w // 2
    u = torch.empty(n, cout, hp, wp, device="cuda")
    idx = ops._empty_with_slack((n, cout, hp, wp), torch.uint8, "cuda")
    ws = torch.empty(lib.afd_conv2d_workspace_bytes(n, cin, h, w, cout, 3, 1, 1), dtype=torch.uint8, device="cuda")
    _native.check(lib.afd_conv3x3_prelu_pool_forward(_native.ptr(x), _native.ptr(wt), _native.ptr(bias), _native.ptr(slope),
                                                     _native.ptr(u), _native.ptr(idx), n, cin, h, w, cout, _native.ptr(ws),
                                                     ws.numel(), _native.stream_ptr()), "conv + pool")
    du = torch.randn_like(u)
    coef = torch.randn(cout, 4, device="cuda")
    sws = torch.empty(lib.afd_conv3x3_backward_data_bnstats_workspace_bytes(n, cin, h, w), dtype=torch.uint8, device="cuda")
    # dense path
    dz = torch.empty(n, cout, h, w, device="cuda")
    ds0 = torch.zeros(1, device="cuda")
    _native.check(lib.afd_prelu_pool_backward_affine(_native.ptr(u), _native.ptr(slope), _native.ptr(idx), _native.ptr(du),
                                                     _native.ptr(coef), cout, _native.ptr(dz), _native.ptr(ds0), n * cout, h, w,
                                                     _native.stream_ptr()), "pool bwd")
    dx0 = torch.empty_like(x)
    sums0 = torch.empty(2 * cin, dtype=torch.float64, device="cuda")
    _native.check(lib.afd_conv3x3_backward_data_bnstats(
        _native.ptr(dz), _native.ptr(wt), _native.ptr(dx0), None, _native.ptr(sums0), n, cin, h, w, cout,
        _native.ptr(ws), ws.numel(), _native.ptr(sws), sws.numel(), _native.stream_ptr()), "dense dgrad")
    dw0, db0 = torch.empty_like(wt), torch.empty(cout, device="cuda")
    _native.check(lib.afd_conv2d_backward_weight(_native.ptr(x), _native.ptr(dz), _native.ptr(dw0), _native.ptr(db0), n, cin, h, w,
                                                 cout, 3, 1, 1, _native.ptr(ws), ws.numel(), _native.stream_ptr()), "dense wgrad")
    # pooled path
    gg = ops._empty_with_slack(u.shape, torch.float32, "cuda")
    ds1 = torch.zeros(1, device="cuda")
    _native.check(lib.afd_prelu_pool_backward_compact(_native.ptr(u), _native.ptr(slope), _native.ptr(idx), _native.ptr(du),
                                                      _native.ptr(coef), cout, _native.ptr(gg), _native.ptr(ds1), n * cout, hp, wp,
                                                      _native.stream_ptr()), "pool bwd compact")
    dx1 = torch.empty_like(x)
    sums1 = torch.empty(2 * cin, dtype=torch.float64, device="cuda")
    _native.check(lib.afd_conv3x3_backward_data_bnstats_pooled(
        _native.ptr(gg), _native.ptr(idx), _native.ptr(wt), _native.ptr(dx1), _native.ptr(sums1), n, cin, h, w, cout,
        _native.ptr(ws), ws.numel(), _native.ptr(sws), sws.numel(), _native.stream_ptr()), "pooled dgrad")
    dw1, db1 = torch.empty_like(wt), torch.empty(cout, device="cuda")
    _native.check(lib.afd_conv3x3_backward_weight_pooled(_native.ptr(x), _native.ptr(gg), _native.ptr(idx), _native.ptr(dw1),
                                                         _native.ptr(db1), n, cin, h, w, cout, _native.ptr(ws), ws.numel(),
                                                         _native.stream_ptr()), "pooled wgrad")
    assert torch.equal(dx0, dx1)
    assert torch.allclose(ds0, ds1, rtol=3e-4, atol=1e-5)  # float atomics over the workgroups: the order differs run to run
    assert (sums0[:cin] - sums1[:cin]).abs().max().item() <= 1e-9 * max(1.0, sums0[:cin].abs().max().item())
    _close(dw1, dw0.cpu().double(), 3e-6, "pooled wgrad")  # (two float32 summation orders over the same products)
    _close(db1, db0.cpu().double(), 2e-6, "pooled dbias")

    # autograd: BatchNorm -> conv + PReLU + pool -> BatchNorm, with and without the pooled backward
    bn_in = torch.nn.BatchNorm2d(cin, affine=False).cuda().train()
    bn_out = torch.nn.BatchNorm2d(cout, affine=False).cuda().train()
    sl = torch.nn.Parameter(torch.full((1,), 0.25, device="cuda"))
    dy = torch.randn(n, cout, hp, wp, device="cuda")
    res = []
    for off in (True, False):
        if off:
            os.environ["AFD_NO_POOLED_BWD"] = "1"
        else:
            os.environ.pop("AFD_NO_POOLED_BWD", None)
        try:
            conv.zero_grad()
            sl.grad = None
            z = x.clone().requires_grad_(True)
            link, pool_link = {}, {}
            hcur = ops.batch_norm(z, bn_in, None, False, link)
            hcur = ops.conv3x3_prelu_maxpool(hcur, conv.weight, conv.bias, sl, link, pool_link)
            y = ops.batch_norm(hcur, bn_out, None, False, None, pool_link)
            y.backward(dy)
            res.append((z.grad.clone(), conv.weight.grad.clone(), conv.bias.grad.clone(), sl.grad.clone()))
        finally:
            os.environ.pop("AFD_NO_POOLED_BWD", None)
    for got, want, what in zip(res[1], res[0], ("dx", "dw", "db", "dslope")):
        # (dslope: one float atomic per workgroup, the order differs run to run)
        _close(got, want.cpu().double(), 3e-4 if what == "dslope" else 5e-6, "pooled backward through autograd: " + what)


@pytest.mark.parametrize("shape,cout,pooled", [((2, 64, 12, 1030), 96, True), ((1, 64, 13, 259), 96, True),
                                               ((2, 96, 6, 1101), 128, False), ((2, 128, 6, 1027), 32, False)])
def test_batchnorm_batch_sums_from_the_convolution_epilogue(shape, cout, pooled):
    """`afd_conv3x3_forward_stats` (the F(4x4) Winograd kernel's statistics epilogue): the convolution output (or
    pooled value and code) is the plain launch's bit for bit, the sums match float64 sums of PReLU(y) / u, and through
    autograd conv [-> pool] -> BatchNorm gives the same output and gradients with and without the hand-over."""
    torch.manual_seed(31)
    lib = _native.load()
    n, cin, h, w = shape
    assert lib.afd_conv3x3_forward_stats_applicable(cin, h, w, cout, int(pooled))
    x = torch.randn(shape, device="cuda")
    conv = torch.nn.Conv2d(cin, cout, 3, padding=1).cuda()
    bn = torch.nn.BatchNorm2d(cout, affine=False).cuda().train()
    slope = torch.full((1,), 0.25, device="cuda")
    outs = []
    for linked in (False, True):
        bn.reset_running_stats()
        conv.zero_grad()
        xg = x.clone().requires_grad_(True)
        link = {"want_stats": True, "stats_slope": slope} if linked else {}
        if pooled:
            u = ops.conv3x3_prelu_maxpool(xg, conv.weight, conv.bias, slope, None, link)
            y = ops.batch_norm(u, bn, None, False, None, link)
            mid = u
        else:
            z = ops.conv2d(xg, conv.weight, conv.bias, 1, 1, out_link=link)
            y = ops.batch_norm(z, bn, slope, False, None, sum_link=link)
            mid = z
        assert "fwd_sums" not in link  # consumed by the BatchNorm's forward
        y.backward(torch.ones_like(y) * torch.linspace(-1, 1, y.shape[-1], device="cuda"))
        outs.append((mid.detach().clone(), y.detach().clone(), xg.grad.clone(), conv.weight.grad.clone(),
                     bn.running_mean.clone(), bn.running_var.clone()))
    assert torch.equal(outs[0][0], outs[1][0])
    _close(outs[1][1], outs[0][1].cpu(), 2e-6, "BatchNorm output")
    _close(outs[1][2], outs[0][2].cpu(), 1e-5, "input gradient")
    _close(outs[1][3], outs[0][3].cpu(), 1e-5, "weight gradient")
    _close(outs[1][4], outs[0][4].cpu(), 2e-6, "running mean")
    _close(outs[1][5], outs[0][5].cpu(), 2e-6, "running variance")
    v = outs[0][0].double() if pooled else torch.where(outs[0][0] > 0, outs[0][0], 0.25 * outs[0][0]).double()
    mean_ref = v.mean((0, 2, 3))
    _close(outs[1][4] / 0.1, mean_ref.cpu(), 1e-5, "batch mean against float64")


@pytest.mark.parametrize("shape,cout", [((2, 128, 6, 1028), 32), ((1, 96, 4, 1100), 128), ((2, 16, 9, 70), 24)])
def test_bias_gradient_from_the_batchnorm_backward(shape, cout):
    """conv -> PReLU -> BatchNorm with the two calls linked (`out_link` / `sum_link`): the convolution's bias
    gradient is the per-channel sum the BatchNorm backward accumulates while it writes its result
    (`afd_bn_backward_apply_sums` -> `afd_conv2d_backward_weight_sums`), equal to the unlinked pass over dy; the
    first two shapes are the layers whose backward-weight product runs with swapped operands (conv.hip)."""
    torch.manual_seed(21)
    n, cin, h, w = shape
    x = torch.randn(shape, device="cuda")
    conv = torch.nn.Conv2d(cin, cout, 3, padding=1).cuda()
    bn = torch.nn.BatchNorm2d(cout, affine=False).cuda().train()
    slope = torch.full((1,), 0.25, device="cuda", requires_grad=True)
    dy = torch.randn(n, cout, h, w, device="cuda")
    grads = []
    for linked in (False, True):
        conv.zero_grad()
        link = {} if linked else None
        z = ops.conv2d(x, conv.weight, conv.bias, 1, 1, out_link=link)
        y = ops.batch_norm(z, bn, slope, False, None, sum_link=link)
        y.backward(dy)
        grads.append((conv.bias.grad.clone(), conv.weight.grad.clone()))
        assert not linked or "dy_sums" not in link  # consumed by the convolution's backward
    _close(grads[1][0], grads[0][0].cpu(), 2e-6, "bias gradient from the BatchNorm backward")
    assert torch.equal(grads[1][1], grads[0][1])


@pytest.mark.parametrize("shape", [(2, 96, 12, 1030), (3, 24, 7, 131)])
def test_batchnorm_backward_inside_the_pool_backward(shape):
    """PReLU + MaxPool2d(2, 2) -> BatchNorm(affine=False): with the two calls linked the BatchNorm's backward is
    applied inside the pool's backward (`afd_prelu_pool_backward_affine`); gradients equal the unlinked chain and
    the same layers in float64."""
    torch.manual_seed(15)
    n, c, h, w = shape
    z0 = torch.randn(shape, device="cuda")
    act = torch.nn.PReLU().cuda()
    bn = torch.nn.BatchNorm2d(c, affine=False).cuda().train()
    g = torch.randn(n, c, h // 2, w // 2, device="cuda")
    res = []
    for linked in (False, True):
        z = z0.clone().requires_grad_(True)
        act.weight.grad = None
        link = {} if linked else None
        y = ops.batch_norm(ops.prelu_maxpool2x2(z, act.weight, link), bn, None, False, None, link)
        y.backward(g)
        assert not linked or not link
        res.append((y.detach(), z.grad, act.weight.grad.clone()))
    ref = torch.nn.Sequential(torch.nn.PReLU(), torch.nn.MaxPool2d(2, 2), torch.nn.BatchNorm2d(c, affine=False)).double().cuda().train()
    z64 = z0.double().requires_grad_(True)
    yr = ref(z64)
    yr.backward(g.double())
    for y, dz, ds in res:
        _close(y, yr.detach().cpu(), 1e-5, "output")
        _close(dz, z64.grad.cpu(), 2e-5, "dz")
        _close(ds, ref[0].weight.grad.cpu(), 1e-4, "dslope")


WIDE3X3 = [(2, 32, 6, 1100, 64), (1, 96, 4, 1025, 128), (1, 64, 13, 1157, 96), (2, 64, 51, 129, 96)]


@pytest.mark.parametrize("shape", WIDE3X3)
@pytest.mark.parametrize("winograd", [True, False])
def test_3x3_layers_on_both_kernels(shape, winograd, monkeypatch):
    """The 3x3 / pad 1 layers with more than 32 output channels run on a Winograd kernel -- F(2x2,3x3)
    (wino.hip), or F(4x4,3x3) (wino44.hip) on images at least 256 wide; AFD_NO_WINOGRAD=1 keeps them on the
    direct implicit GEMM (conv3x3.hip).  Both against float64, forward and backward-data: the F(2x2) transform adds
    rounding (inputs and filters are combined before the products) but stays inside the direct kernel's 1e-5 bar;
    the F(4x4) kernels' stated bar is 2e-5 (transform constants up to 8; test_winograd_f44_layers)."""
    if not winograd:
        monkeypatch.setenv("AFD_NO_WINOGRAD", "1")
    n, cin, h, w, cout = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    b = torch.randn(cout, generator=g)
    xr, wr, br = (t.double().requires_grad_() for t in (x, wt, b))
    yr = F.conv2d(xr, wr, br, padding=1)
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy.double())
    xg, wg, bg = (t.cuda().requires_grad_() for t in (x, wt, b))
    yg = ops.conv2d(xg, wg, bg, 1, 1)
    yg.backward(dy.cuda())
    bar = 2e-5 if (winograd and w >= 256) else 1e-5
    _close(yg, yr.detach(), bar, "fwd")
    _close(xg.grad, xr.grad, bar, "dgrad")
    _close(wg.grad, wr.grad, 3e-5, "wgrad")


def test_winograd_class_is_what_runs():
    """The timing classes tell the two 3x3 kernels apart (bench.py's roofline object relies on it)."""
    from audiofakedetect import _native

    x = torch.randn(1, 64, 6, 260, device="cuda")
    wt = torch.randn(96, 64, 3, 3, device="cuda") / 24
    _native.timing_reset()
    _native.timing_enable(True)
    try:
        ops.conv2d(x, wt, None, 1, 1)
        torch.cuda.synchronize()
        cw = _native.timing_collect("conv_winograd")
        n_w, work = cw["launches"], cw["work"]
        n_d = _native.timing_collect("conv_igemm")["launches"]
    finally:
        _native.timing_enable(False)
        _native.timing_reset()
    assert n_w == 1 and n_d == 0
    assert work == 2.0 * 96 * 6 * 260 * 64 * 9
    # issued on the matrix cores: 16 GEMMs of [96 x 64] x [64 x tiles], tiles padded to the workgroup's
    # 64 (3 tile rows x 130 tiles -> 3 x 192), against 36/16 more in direct form
    assert cw["issued"] == 2.0 * 16 * 96 * 64 * 3 * 192
    assert cw["bytes"] == 4.0 * (64 + 96) * 6 * 260


def _issued_winograd(fn):
    from audiofakedetect import _native

    _native.timing_reset()
    _native.timing_enable(True)
    try:
        out = fn()
        torch.cuda.synchronize()
        cw = _native.timing_collect("conv_winograd")
    finally:
        _native.timing_enable(False)
        _native.timing_reset()
    return out, cw


@pytest.mark.parametrize("case", [
    # (n, cin, h, w, cout, direction): the layer shapes wino44.hip takes at level 14, with ragged right edges
    (2, 96, 6, 1101, 128, "fwd"),     # block 4 forward: 8 waves
    (1, 96, 13, 1030, 64, "dgrad"),   # block 3 backward-data: forward 64 -> 96, the GEMM's output is 64 channels
    (2, 32, 6, 1027, 128, "dgrad"),   # block 5 backward-data: forward 128 -> 32
    (1, 96, 3, 300, 64, "dgrad"),     # a single tile row with one row unused
    (1, 128, 6, 1030, 96, "dgrad"),   # block 4 backward-data: six matrix waves + two helper waves
    (2, 128, 6, 1027, 32, "fwd"),     # block 5 forward: two waves, 8-channel chunks
    (2, 64, 7, 300, 32, "dgrad"),     # block 6 backward-data
    (2, 64, 13, 1030, 96, "dgrad"),   # 13 rows: the last tile row has ONE live row
    (2, 128, 5, 300, 96, "dgrad"),    # five rows: a full tile row and one live row
    (1, 96, 10, 517, 128, "fwd"),     # ten rows: two full tile rows and a two-row one
    (3, 64, 55, 129, 96, "dgrad"),    # block 3 backward-data at level 8 / STFT: 33 tile columns = 2 x 16 + 1 -> tail form
    (2, 64, 70, 65, 96, "dgrad"),     # 17 tile columns, 18 tile rows: two tail workgroups per image, the second ragged
])
def test_winograd_f44_layers(case):
    """wino44.hip, Winograd F(4x4, 3x3): against float64 at the stated bar for that kernel -- 2e-5 of the largest
    output (measured <= 1.2e-5; the F(2x2) kernels: 4e-7) -- and the issued-flop count says it is the kernel that ran:
    36 GEMMs per 4x4 tile, 16 tiles per workgroup."""
    n, cg_in, h, w, cg_out, direction = case
    g = torch.Generator().manual_seed(sum(case[:5]))
    # (tile rows x 36 positions; 30 in a last tile row with at most three live output rows)
    tx = -(-w // 4)
    tiles_x = -(-tx // 16) * 16
    ty = -(-h // 4)
    # (the last tile row: 24 positions -- F(2x4), round 6 -- with at most two live output rows, 30 with three)
    last = h - 4 * (ty - 1)
    pos_rows = (ty - 1) * 36 + (24 if last <= 2 else (30 if last <= 3 else 36))
    pos_tiles = pos_rows * tiles_x
    if tx % 16 == 1 and tx > 16 and ty >= 2:
        # round 6, tail form: the one live tile of the last workgroup column is packed 16 tile rows per workgroup
        # (all 36 positions there) instead of one workgroup with 15 dead tiles per tile row
        pos_tiles = pos_rows * (tiles_x - 16) + 36 * 16 * -(-ty // 16)
    if direction == "fwd":
        x = torch.randn(n, cg_in, h, w, generator=g)
        wt = torch.randn(cg_out, cg_in, 3, 3, generator=g) / (cg_in * 9) ** 0.5
        b = torch.randn(cg_out, generator=g)
        ref = F.conv2d(x.double(), wt.double(), b.double(), padding=1)
        got, cw = _issued_winograd(lambda: ops.conv2d(x.cuda(), wt.cuda(), b.cuda(), 1, 1))
    else:
        # backward-data of a forward layer cg_out -> cg_in channels: dy has cg_in channels, dx has cg_out
        dy = torch.randn(n, cg_in, h, w, generator=g)
        wt = torch.randn(cg_in, cg_out, 3, 3, generator=g) / (cg_out * 9) ** 0.5
        ref = torch.nn.grad.conv2d_input((n, cg_out, h, w), wt.double(), dy.double(), padding=1)
        lib = _native.load()
        dx = torch.empty(n, cg_out, h, w, device="cuda")
        ws = torch.empty(lib.afd_conv2d_workspace_bytes(n, cg_out, h, w, cg_in, 3, 1, 1), dtype=torch.uint8, device="cuda")
        dyc, wc = dy.cuda(), wt.cuda()

        def run():
            _native.check(lib.afd_conv2d_backward_data(_native.ptr(dyc), _native.ptr(wc), _native.ptr(dx), n, cg_out, h, w,
                                                       cg_in, 3, 1, 1, _native.ptr(ws), ws.numel(),
                                                       _native.stream_ptr()), "dgrad")
            return dx
        got, cw = _issued_winograd(run)
    _close(got, ref, 2e-5, f"F(4x4) {direction}")
    assert cw["launches"] == 1
    assert cw["issued"] == 2.0 * pos_tiles * cg_out * cg_in * n


@pytest.mark.parametrize("shape", [(1, 64, 7, 1027, 96), (2, 32, 6, 1100, 64), (2, 64, 13, 257, 96)])
def test_conv_prelu_pool_in_one_launch(shape):
    """conv3x3_prelu_maxpool (the pool folded into the Winograd epilogue, reference models.py:263-265)
    against conv -> PReLU -> MaxPool2d(2, 2) in float64, and bit-for-bit against the two-launch path."""
    n, cin, h, w, cout = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    b = torch.randn(cout, generator=g)
    slope = torch.tensor([0.25])
    conv = torch.nn.Conv2d(cin, cout, 3, padding=1)
    assert ops.conv3x3_prelu_maxpool_applicable(x.cuda(), conv)
    xr, wr, br, sr = (t.double().requires_grad_() for t in (x, wt, b, slope))
    yr = F.max_pool2d(F.prelu(F.conv2d(xr, wr, br, padding=1), sr), 2, 2)
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy.double())
    xg, wg, bg, sg = (t.cuda().requires_grad_() for t in (x, wt, b, slope))
    yg = ops.conv3x3_prelu_maxpool(xg, wg, bg, sg)
    _close(yg, yr.detach(), 2e-5, "fused fwd")
    yg.backward(dy.cuda())
    _close(xg.grad, xr.grad, 2e-5, "dgrad")
    _close(wg.grad, wr.grad, 3e-5, "wgrad")
    _close(bg.grad, br.grad, 2e-5, "dbias")
    _close(sg.grad, sr.grad, 1e-4, "dslope")
    with torch.no_grad():
        two = ops.prelu_maxpool2x2(ops.conv2d(xg, wg, bg, 1, 1, pooled=True), sg)
    if cout in (64, 96) and cin % 16 == 0 and w >= 256 and h >= 4:
        # one launch: Winograd F(4x4, 3x3) (wino44.hip); two launches: F(2x2, 3x3) -- equal to rounding
        _close(two, yg.detach().double().cpu(), 2e-5, "one launch against two")
    else:
        assert torch.equal(two, yg.detach())


def _issued_class(fn, name):
    _native.timing_reset()
    _native.timing_enable(True)
    try:
        out = fn()
        torch.cuda.synchronize()
    finally:
        _native.timing_enable(False)
    c = _native.timing_collect(name)
    _native.timing_reset()
    return out, c


@pytest.mark.parametrize("case", [
    # n, cin, h, w, cout, crop rows, crop cols      (level-14 block shapes at reduced width)
    (2, 64, 13, 1029, 96, 12, 1028),   # block 3: 3 x 2 channel groups, pooled crop, three tile rows, first / last column groups
    (1, 96, 6, 1024, 128, 6, 1024),    # block 4: 4 x 3 channel groups, a half-used second tile row, right border by one column
    (2, 128, 6, 516, 32, 6, 516),      # block 5: ragged last column group (129 tiles)
    (3, 32, 6, 260, 64, 6, 260),       # block 6
    (2, 64, 27, 64, 96, 26, 64),       # a level-8 shape: 16 tiles per row, seven tile rows (one ragged)
    (1, 64, 3, 1029, 64, 2, 1028),     # the sym5 level-14 geometry: one tile row, two live rows
    (2, 32, 9, 132, 32, 9, 132),       # one channel group, three tile rows (the last with one live row), ragged column group
    (1, 96, 5, 200, 96, 5, 200),       # 3 x 3 channel groups, two tile rows (the last with one live row)
])
@pytest.mark.parametrize("with_sums", [False, True])
def test_winograd_domain_backward_weight(case, with_sums):
    """wino44_wgrad.hip: dW = G^T [sum over tiles (A dy A^T) . (B^T x B)] G against float64, 3e-5 of the largest
    entry (the bar of the direct kernels).  dy holds garbage outside the crop (a pooled layer's odd last row /
    column): the kernel must not read it into the sums.  The issued-flop count says which kernel ran."""
    n, cin, h, w, cout, rows, cols = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(n, cin, h, w, generator=g)
    dy = torch.randn(n, cout, h, w, generator=g)
    dyz = torch.zeros_like(dy)
    dyz[:, :, :rows, :cols] = dy[:, :, :rows, :cols]
    ref_w = torch.nn.grad.conv2d_weight(x.double(), (cout, cin, 3, 3), dyz.double(), padding=1)
    ref_b = dyz.double().sum((0, 2, 3))
    lib = _native.load()
    xc, dyc = x.cuda(), dy.cuda()
    dw = torch.empty(cout, cin, 3, 3, device="cuda")
    db = torch.empty(cout, device="cuda")
    sums = ref_b.cuda() if with_sums else None
    ws = torch.empty(lib.afd_conv2d_workspace_bytes(n, cin, h, w, cout, 3, 1, 1), dtype=torch.uint8, device="cuda")
    def run():
        _native.check(lib.afd_conv2d_backward_weight_sums(
            _native.ptr(xc), _native.ptr(dyc), _native.ptr(dw), _native.ptr(db), _native.ptr(sums), n, cin, h, w,
            cout, 3, 1, 1, rows, cols, _native.ptr(ws), ws.numel(), _native.stream_ptr()), "wgrad")
        return dw
    _, c = _issued_class(run, "conv_wgrad")
    _close(dw, ref_w, 3e-5, "winograd-domain wgrad")
    _close(db, ref_b, 2e-5, "dbias")
    groups = -(-(-(-cols // 4)) // 4)
    assert c["launches"] == 1
    # 36 products per tile, channel pair and k-step of four tiles; 30 in the k-steps of a last tile row with fewer than
    # four live rows (A dy A^T is zero at the positions 30..35 there, the kernel skips them)
    short = n * groups if rows % 4 else 0
    assert c["issued"] == 2.0 * cout * cin * 4 * (36 * n * groups * -(-rows // 4) - 6 * short)


# ---- gradient parity at the grid sizes the benchmark runs (level-14 blocks 3 and 4, N = 128) ----------------------
BENCH_WGRAD = [
    # n, cin, h, w, cout, crop rows, crop cols
    (128, 64, 13, 8193, 96, 12, 8192),   # block 3: the pooled crop of the 13 x 8193 image
    (128, 96, 6, 4096, 128, 6, 4096),    # block 4
]


@pytest.mark.parametrize("case", BENCH_WGRAD)
def test_backward_weight_at_benchmark_size_equals_its_two_frame_pieces(case):
    """At N = 128 the backward-weight launch splits its tile sequence over three rounds of workgroups (768 with
    8-aligned split counts); at N = 2 the same entry point runs a handful of splits.  The weight gradient is linear in
    the batch, so the N = 128 result must equal the sum (in float64) of the 64 two-frame results, and one two-frame
    piece is checked against torch's float64 gradient -- both at 3e-5 of the largest entry, the layer bar.  (Measured:
    2.2e-5 for block 3, whose float32 accumulators each sum ~50 000 Winograd-domain products whose transform constants
    reach 8 x 20; the review's 2e-6 would hold for a direct-form sum, not for F(4x4) in float32.)
    Reference: what autograd returns for nn.Conv2d(k=3, padding=1) at models.py:264-270."""
    n, cin, h, w, cout, rows, cols = case
    torch.manual_seed(sum(case))
    lib = _native.load()
    x = torch.randn(n, cin, h, w, device="cuda")
    dy = torch.randn(n, cout, h, w, device="cuda")
    ws = torch.empty(lib.afd_conv2d_workspace_bytes(n, cin, h, w, cout, 3, 1, 1), dtype=torch.uint8, device="cuda")

    def wgrad(xs, dys, nn):
        dw = torch.empty(cout, cin, 3, 3, device="cuda")
        db = torch.empty(cout, device="cuda")
        _native.check(lib.afd_conv2d_backward_weight_sums(
            _native.ptr(xs), _native.ptr(dys), _native.ptr(dw), _native.ptr(db), None, nn, cin, h, w, cout, 3, 1, 1,
            rows, cols, _native.ptr(ws), ws.numel(), _native.stream_ptr()), "wgrad")
        return dw, db

    (dw_all, db_all), c = _issued_class(lambda: wgrad(x, dy, n), "conv_wgrad")
    assert c["launches"] == 1 and c["issued"] > 0  # the Winograd-domain kernel, one launch
    acc_w = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, device="cuda")
    acc_b = torch.zeros(cout, dtype=torch.float64, device="cuda")
    first = None
    for i in range(0, n, 2):
        dw2, db2 = wgrad(x[i:i + 2], dy[i:i + 2], 2)
        if first is None:
            first = dw2.clone()
        acc_w += dw2.double()
        acc_b += db2.double()
    _close(dw_all, acc_w.cpu(), 3e-5, "N = 128 wgrad vs the sum of its two-frame pieces")
    _close(db_all, acc_b.cpu(), 2e-6, "N = 128 dbias vs the sum of its two-frame pieces")
    dyz = torch.zeros(2, cout, h, w, dtype=torch.float64)
    dyz[:, :, :rows, :cols] = dy[:2, :, :rows, :cols].cpu().double()
    ref = torch.nn.grad.conv2d_weight(x[:2].cpu().double(), (cout, cin, 3, 3), dyz, padding=1)
    _close(first, ref, 3e-5, "two-frame piece vs float64")


def test_pooled_backward_at_benchmark_size_equals_its_two_frame_pieces():
    """Block 3 as the step runs it: the gradient of the pooled convolution arrives pooled (gg + argmax codes).  The
    N = 128 pooled backward-weight launch equals the sum of its 64 two-frame launches (3e-5, see above), and the N = 128 pooled
    backward-data launch gives frames 0-1 exactly what the N = 2 launch gives (per-frame work, bit-equal)."""
    n, cin, h, w, cout = 128, 64, 13, 8193, 96
    hp, wp = h // 2, w // 2
    torch.manual_seed(7)
    lib = _native.load()
    assert lib.afd_conv3x3_pooled_backward_applicable(cin, h, w, cout)
    x = torch.randn(n, cin, h, w, device="cuda")
    gg = ops._empty_with_slack((n, cout, hp, wp), torch.float32, "cuda")
    gg.normal_()
    idx = ops._empty_with_slack((n, cout, hp, wp), torch.uint8, "cuda")
    idx.copy_(torch.randint(0, 8, (n, cout, hp, wp), device="cuda", dtype=torch.uint8))
    wt = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
    ws = torch.empty(lib.afd_conv2d_workspace_bytes(n, cin, h, w, cout, 3, 1, 1), dtype=torch.uint8, device="cuda")

    def slices(i, nn):
        # two-frame pieces keep the slack behind them: they are views into the big tensors (the last piece of all
        # ends where the big tensor's own slack begins)
        return x[i:i + nn], gg[i:i + nn], idx[i:i + nn]

    def wgrad(i, nn):
        xs, gs, cs = slices(i, nn)
        dw = torch.empty(cout, cin, 3, 3, device="cuda")
        db = torch.empty(cout, device="cuda")
        _native.check(lib.afd_conv3x3_backward_weight_pooled(_native.ptr(xs), _native.ptr(gs), _native.ptr(cs), _native.ptr(dw),
                                                             _native.ptr(db), nn, cin, h, w, cout, _native.ptr(ws), ws.numel(),
                                                             _native.stream_ptr()), "pooled wgrad")
        return dw, db

    dw_all, db_all = wgrad(0, n)
    acc_w = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, device="cuda")
    acc_b = torch.zeros(cout, dtype=torch.float64, device="cuda")
    for i in range(0, n, 2):
        dw2, db2 = wgrad(i, 2)
        acc_w += dw2.double()
        acc_b += db2.double()
    _close(dw_all, acc_w.cpu(), 3e-5, "pooled N = 128 wgrad vs the sum of its two-frame pieces")
    _close(db_all, acc_b.cpu(), 2e-6, "pooled N = 128 dbias vs the sum of its two-frame pieces")

    def dgrad(nn):
        xs, gs, cs = slices(0, nn)
        dx = torch.empty(nn, cin, h, w, device="cuda")
        sums = torch.empty(2 * cin, dtype=torch.float64, device="cuda")
        sws = torch.empty(lib.afd_conv3x3_backward_data_bnstats_workspace_bytes(nn, cin, h, w), dtype=torch.uint8, device="cuda")
        _native.check(lib.afd_conv3x3_backward_data_bnstats_pooled(
            _native.ptr(gs), _native.ptr(cs), _native.ptr(wt), _native.ptr(dx), _native.ptr(sums), nn, cin, h, w, cout,
            _native.ptr(ws), ws.numel(), _native.ptr(sws), sws.numel(), _native.stream_ptr()), "pooled dgrad")
        return dx

    dx_all = dgrad(n)
    dx_two = dgrad(2)
    assert torch.equal(dx_all[:2], dx_two)


def test_batchnorm_backward_sums_from_the_weight_gradient_at_full_width(monkeypatch):
    """the BatchNorm backward's second sum, sum(dx * xhat), comes from the convolution's weights and
    their fp32 gradient (afd_conv_weight_dot: sum_px dx[c] xhat[c] = sum_{co,k} w dw) instead of a pass over the
    activations.  At the full level-14 width of block 4 ([N, 96, 6, 4096] -> 128 channels) and a batch of 32 the
    BatchNorm's input gradient through that identity equals the one with AFD_BNSTATS_FROM_INPUT=1 (the sums taken from
    the activations in double precision) to 5e-6 of its largest entry -- a precision regression in the weight
    gradient would show here."""
    torch.manual_seed(23)
    n, cin, h, w, cout = 32, 96, 6, 4096, 128
    x = torch.randn(n, cin, h, w, device="cuda")
    dy = torch.randn(n, cout, h, w, device="cuda")
    bn = torch.nn.BatchNorm2d(cin, affine=False).cuda().train()
    conv = torch.nn.Conv2d(cin, cout, 3, padding=1).cuda()
    slope = torch.full((1,), 0.25, device="cuda")
    grads = []
    for from_input in (True, False):
        if from_input:
            monkeypatch.setenv("AFD_BNSTATS_FROM_INPUT", "1")
        else:
            monkeypatch.delenv("AFD_BNSTATS_FROM_INPUT", raising=False)
        conv.zero_grad()
        z = x.clone().requires_grad_(True)
        link = {}
        y = ops.conv2d(ops.batch_norm(z, bn, slope, False, link), conv.weight, conv.bias, 1, 1, bn_link=link)
        y.backward(dy)
        grads.append((z.grad.clone(), conv.weight.grad.clone()))
    _close(grads[1][0], grads[0][0].cpu(), 5e-6, "BatchNorm input gradient: w . dw sums vs activation sums")
    _close(grads[1][1], grads[0][1].cpu(), 1e-6, "weight gradient on both paths")


FOLD_CASES = [
    # shape of the BatchNorm's input, slope in front of it (None: no PReLU), cout, pooled, statistics behind the conv
    ((2, 64, 13, 1029), 0.25, 96, True, True),     # block 2 -> 3 (z2, PReLU, pooled convolution + sums)
    ((2, 96, 6, 1100), None, 128, False, True),    # block 3 -> 4 (pooled tensor, no PReLU)
    ((2, 128, 6, 1028), 1.7, 32, False, True),     # block 4 -> 5, slope above 1 (the general PReLU form)
    ((2, 32, 6, 1029), -0.2, 64, True, False),     # block 5 -> 6, negative slope
    ((1, 64, 13, 261), 0.0, 96, True, True),       # ragged last workgroup column, slope 0
    ((3, 64, 51, 129), 0.25, 96, True, True),      # level-8 / STFT geometry: every workgroup a border workgroup
    ((3, 96, 25, 64), None, 128, False, True),
    ((2, 128, 25, 64), 0.6, 32, False, True),
]


@pytest.mark.parametrize("shape,slope_in,cout,pooled,stats", FOLD_CASES)
def test_batchnorm_applied_while_the_convolution_loads(shape, slope_in, cout, pooled, stats, monkeypatch):
    """`batch_norm(defer=True)` -> 3x3 convolution (`afd_conv3x3_forward_fold` / `afd_conv3x3_backward_weight_fold`):
    the BatchNorm's result is never stored, the convolution's forward and backward-weight launches build
    (PReLU(z) - mean) * invstd from z while they load -- with afd_bn_apply_forward's arithmetic, so outputs and
    gradients equal the two-pass chain bit for bit (the PReLU slope's gradient ends in float atomics: tolerance)."""
    torch.manual_seed(41)
    n, cin, h, w = shape
    x = torch.randn(shape, device="cuda") * 1.5 + 0.3
    conv = torch.nn.Conv2d(cin, cout, 3, padding=1).cuda()
    bn_in = torch.nn.BatchNorm2d(cin, affine=False).cuda().train()
    bn_out = torch.nn.BatchNorm2d(cout, affine=False).cuda().train()
    a_in = None if slope_in is None else torch.full((1,), float(slope_in), device="cuda", requires_grad=True)
    a_out = torch.full((1,), 0.25, device="cuda")
    assert ops.conv3x3_input_fold_applicable(bn_in, conv, shape, pooled, stats)
    res = []
    for defer in (False, True, "and the backward in the backward-data launch"):
        # (second run: the BatchNorm's backward is its own pass, as in the two-pass chain; third run: the default --
        # the convolution's backward-data launch applies it, `afd_conv3x3_backward_data_bnapply`)
        if defer is True:
            monkeypatch.setenv("AFD_NO_BWD_BNAPPLY", "1")
        else:
            monkeypatch.delenv("AFD_NO_BWD_BNAPPLY", raising=False)
        bn_in.reset_running_stats()
        bn_out.reset_running_stats()
        conv.zero_grad()
        if a_in is not None:
            a_in.grad = None
        xg = x.clone().requires_grad_(True)
        link = {}
        hh = ops.batch_norm(xg, bn_in, a_in, False, link, defer=defer)
        assert ("fold" in link) == bool(defer)
        if defer:
            assert hh.data_ptr() == xg.data_ptr()  # nothing was written
        if pooled:
            out_link = {"want_stats": True} if stats else None
            u = ops.conv3x3_prelu_maxpool(hh, conv.weight, conv.bias, a_out, link, out_link)
            y = ops.batch_norm(u, bn_out, None, False, None, out_link) if stats else u
            mid = u
        else:
            out_link = {"want_stats": True, "stats_slope": a_out}
            z = ops.conv2d(hh, conv.weight, conv.bias, 1, 1, bn_link=link, out_link=out_link)
            y = ops.batch_norm(z, bn_out, a_out, False, None, sum_link=out_link)
            mid = z
        assert "fold" not in link  # taken by the convolution
        gen = torch.Generator(device="cuda").manual_seed(5)
        y.backward(torch.randn(y.shape, device="cuda", generator=gen))
        res.append((mid.detach().clone(), y.detach().clone(), xg.grad.clone(), conv.weight.grad.clone(),
                    conv.bias.grad.clone(), None if a_in is None else a_in.grad.clone(), bn_in.running_var.clone()))
    names = ("convolution output", "output", "input gradient", "weight gradient", "bias gradient", "slope gradient",
             "running variance")
    for got, want, what in zip(res[1], res[0], names):
        if want is None:
            continue
        if what == "slope gradient":
            _close(got, want.cpu(), 3e-4, what)
        else:
            assert torch.equal(got, want), f"{what}: {(got - want).abs().max().item():.3e}"
    # with the BatchNorm's backward inside the backward-data launch its two batch sums come from the weights, the weight
    # gradient and the border sums of the output gradient instead of a sum over g: the input gradient agrees to rounding
    for got, want, what in zip(res[2], res[0], names):
        if want is None:
            continue
        if what in ("input gradient", "slope gradient"):
            _close(got, want.cpu(), 3e-4 if what == "slope gradient" else 3e-6, what + " (backward in the launch)")
        else:
            assert torch.equal(got, want), f"{what} (backward in the launch): {(got - want).abs().max().item():.3e}"
    # and the chain against float64
    xd = x.double().cpu()
    v = xd if slope_in is None else torch.where(xd > 0, xd, float(slope_in) * xd)
    xh = (v - v.mean((0, 2, 3), keepdim=True)) / torch.sqrt(v.var((0, 2, 3), unbiased=False, keepdim=True) + bn_in.eps)
    zr = F.conv2d(xh, conv.weight.detach().double().cpu(), conv.bias.detach().double().cpu(), padding=1)
    if pooled:
        zr = F.max_pool2d(torch.where(zr > 0, zr, 0.25 * zr), 2, 2)
    _close(res[1][0], zr, 2e-5, "folded convolution against float64")


@pytest.mark.parametrize("shape", [(2, 64, 13, 1033), (3, 64, 51, 129), (1, 32, 12, 265)])
def test_pool_and_batchnorm_backward_inside_the_next_backward_data_launch(shape, monkeypatch):
    """DCNN blocks 2 -> 3 -> 4: BatchNorm -> [conv3x3 + PReLU + pool] -> BatchNorm -> conv3x3 -> BatchNorm.  By default the
    second convolution's backward-data launch applies the middle BatchNorm's backward AND the pool's / PReLU's
    (`afd_conv3x3_backward_data_bnapply` with the pool's codes) and hands the pooled gradient straight to the first
    convolution's backward; with AFD_NO_BWD_BNAPPLY=1 the pool's backward pass does it.  Same gradients to rounding."""
    torch.manual_seed(43)
    n, cin, h, w = shape
    c1, c2 = (96, 128) if cin == 64 else (64, 32)
    x = torch.randn(shape, device="cuda") * 1.3 + 0.2
    conv1 = torch.nn.Conv2d(cin, c1, 3, padding=1).cuda()
    conv2 = torch.nn.Conv2d(c1, c2, 3, padding=1).cuda()
    bn0 = torch.nn.BatchNorm2d(cin, affine=False).cuda().train()
    bn1 = torch.nn.BatchNorm2d(c1, affine=False).cuda().train()
    bn2 = torch.nn.BatchNorm2d(c2, affine=False).cuda().train()
    a0 = torch.full((1,), 0.3, device="cuda", requires_grad=True)
    a1 = torch.full((1,), 0.25, device="cuda", requires_grad=True)
    a2 = torch.full((1,), 0.2, device="cuda")
    lib = _native.load()
    if not (lib.afd_conv3x3_pooled_backward_applicable(cin, h, w, c1)
            and ops.conv3x3_input_fold_applicable(bn1, conv2, (n, c1, h // 2, w // 2), False, True)):
        pytest.skip("geometry off the kernels this chain needs")
    res = []
    for off in (True, False):
        if off:
            monkeypatch.setenv("AFD_NO_BWD_BNAPPLY", "1")
        else:
            monkeypatch.delenv("AFD_NO_BWD_BNAPPLY", raising=False)
        for m in (conv1, conv2):
            m.zero_grad()
        a0.grad = a1.grad = None
        xg = x.clone().requires_grad_(True)
        l0, pool_link, l1 = {}, {"want_stats": True}, {}
        h0 = ops.batch_norm(xg, bn0, a0, False, l0, defer=True)
        u = ops.conv3x3_prelu_maxpool(h0, conv1.weight, conv1.bias, a1, l0, pool_link)
        h1 = ops.batch_norm(u, bn1, None, False, l1, pool_link, defer=True)
        out_link = {"want_stats": True, "stats_slope": a2}
        z = ops.conv2d(h1, conv2.weight, conv2.bias, 1, 1, bn_link=l1, out_link=out_link)
        y = ops.batch_norm(z, bn2, a2, False, None, sum_link=out_link)
        gen = torch.Generator(device="cuda").manual_seed(7)
        y.backward(torch.randn(y.shape, device="cuda", generator=gen))
        assert ("compacted" not in pool_link) and ("applied" not in l1)  # consumed
        res.append((xg.grad.clone(), conv1.weight.grad.clone(), conv1.bias.grad.clone(), conv2.weight.grad.clone(),
                    conv2.bias.grad.clone(), a0.grad.clone(), a1.grad.clone()))
    names = ("input gradient", "first weight gradient", "first bias gradient", "second weight gradient",
             "second bias gradient", "slope in front", "pool's slope")
    # (the first convolution's bias gradient sums the pooled gradient, whose terms the BatchNorm backward centred: the
    # two forms of that arithmetic -- the pool pass's A g + B u + K, the launch's invstd (g - mean g - xhat mean g xhat) --
    # differ in the last bits of every element and the sum keeps little else)
    for got, want, what in zip(res[1], res[0], names):
        _close(got, want.cpu(), 3e-4 if "slope" in what else (5e-5 if what == "first bias gradient" else 2e-5), what)
