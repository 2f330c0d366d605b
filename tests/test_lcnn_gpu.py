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


def _bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float64)


@pytest.mark.parametrize("cin,cout,k,pad,h,w", [(1, 64, 5, 2, 21, 70), (32, 96, 3, 1, 13, 37), (48, 128, 3, 1, 9, 64),
                                                (64, 128, 1, 0, 12, 32), (32, 64, 3, 1, 12, 32)])
def test_conv2d_bf16_matches_bf16_rounded_reference(cin, cout, k, pad, h, w):
    """afd_conv2d_forward_bf16: operands rounded to bf16 (RNE), fp32 accumulation.  Against float64 on the
    SAME rounded operands the only error is the accumulation order (1e-5 of the largest output); against
    the unrounded fp32 convolution it is the bf16 rounding itself (stated bar 2e-2 of the largest output)."""
    from audiofakedetect.lcnn import conv2d_bf16

    g = torch.Generator().manual_seed(cin * 100 + cout)
    x = torch.randn(3, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    b = torch.randn(cout, generator=g)
    got = conv2d_bf16(x.cuda(), wt.cuda(), b.cuda(), pad).cpu().double()
    ref_r = torch.nn.functional.conv2d(_bf16_round(x), _bf16_round(wt), b.double(), padding=pad)
    ref = torch.nn.functional.conv2d(x.double(), wt.double(), b.double(), padding=pad)
    assert got.shape == ref.shape
    scale = ref.abs().max().item()
    assert (got - ref_r).abs().max().item() <= 1e-5 * scale
    assert (got - ref).abs().max().item() <= 2e-2 * scale
    # max-feature-map fused into the epilogue (reference models.py:203-209)
    fused = conv2d_bf16(x.cuda(), wt.cuda(), b.cuda(), pad, mfm=True).cpu().double()
    want = ref_r.reshape(3, 2, cout // 2, *ref_r.shape[2:]).max(1)[0]
    assert fused.shape == want.shape and (fused - want).abs().max().item() <= 1e-5 * scale


def test_gemm_nt_bf16():
    g = torch.Generator().manual_seed(1)
    a = torch.randn(200, 300, generator=g)
    b = torch.randn(130, 300, generator=g)
    bias = torch.randn(130, generator=g)
    ref_r = _bf16_round(a) @ _bf16_round(b).t() + bias.double()
    got = gemm_nt(a.cuda(), b.cuda(), bias.cuda(), bf16=True)
    assert (got.cpu().double() - ref_r).abs().max().item() <= 1e-5 * ref_r.abs().max().item()
    got2 = gemm_nt(a.cuda(), b.cuda(), None, out=got.clone(), accumulate=True, bf16=True)
    assert (got2.cpu().double() - (2 * ref_r - bias.double())).abs().max().item() <= 2e-5 * ref_r.abs().max().item()


def test_gemm_nt_bf16_on_128_tiles():
    """The 128 x 128-tile kernel (K % 32 == 0, 16-byte aligned rows) on ragged M and N, with and without accumulation,
    against float64 on the bf16-rounded operands -- and equal to the 64 x 64 kernel's result (same rounding, same
    accumulation order along K)."""
    import os
    g = torch.Generator().manual_seed(2)
    a = torch.randn(333, 320, generator=g)
    b = torch.randn(200, 320, generator=g)
    bias = torch.randn(200, generator=g)
    ref_r = _bf16_round(a) @ _bf16_round(b).t() + bias.double()
    got = gemm_nt(a.cuda(), b.cuda(), bias.cuda(), bf16=True)
    assert (got.cpu().double() - ref_r).abs().max().item() <= 1e-5 * ref_r.abs().max().item()
    got2 = gemm_nt(a.cuda(), b.cuda(), None, out=got.clone(), accumulate=True, bf16=True)
    assert (got2.cpu().double() - (2 * ref_r - bias.double())).abs().max().item() <= 2e-5 * ref_r.abs().max().item()
    os.environ["AFD_GEMM_BF16_SMALL"] = "1"
    try:
        small = gemm_nt(a.cuda(), b.cuda(), bias.cuda(), bf16=True)
    finally:
        del os.environ["AFD_GEMM_BF16_SMALL"]
    assert (got - small).abs().max().item() <= 1e-5 * ref_r.abs().max().item()


def test_lcnn_bf16_eval_labels_and_logits():
    """BASELINE configs[4] precision: LCNN(precision="bf16") evaluation forward.  Bars (stated up front):
    class labels bit-exact on the reference fixture; logits within 3e-2 of the largest fp32 logit magnitude
    of the fixture (bf16 has 8 significant bits; nine convolutions, two BLSTM layers); the fp32 path of the
    same module stays within 1e-4."""
    from recipes import fill_state_dict

    g = torch.load(os.path.join(GOLD, "dcnn_lcnn_eval.pt"), map_location="cpu")
    net = LCNN(precision="bf16")
    net.load_state_dict(fill_state_dict(g["shapes"]), strict=True)
    net.cuda().eval()
    with torch.no_grad():
        y = net(g["x"].cuda())
        net.precision = "fp32"
        y32 = net(g["x"].cuda())
    assert (y32.cpu() - g["logits"]).abs().max().item() <= 1e-4
    err = (y.cpu() - g["logits"]).abs().max().item()
    assert err <= 3e-2 * g["logits"].abs().max().item(), err
    assert torch.equal(y.argmax(-1).cpu(), g["labels"])
    # training in bf16 is not built: the module says so instead of silently running fp32
    net.precision = "bf16"
    net.train()
    with pytest.raises(RuntimeError):
        net(g["x"].cuda())
    # a batch of 128 random frames: labels of the bf16 and fp32 paths agree wherever the fp32 margin is
    # larger than the bf16 error bar
    net.eval()
    x = torch.randn(128, 1, 256, 101, generator=torch.Generator().manual_seed(9)).cuda()
    with torch.no_grad():
        yb = net(x)
        net.precision = "fp32"
        yf = net(x)
    bar = 3e-2 * yf.abs().max().item()
    assert (yb - yf).abs().max().item() <= bar
    sure = (yf[:, 0] - yf[:, 1]).abs() > 2 * bar
    assert torch.equal(yb.argmax(-1)[sure], yf.argmax(-1)[sure])


def test_blstm_layer_forward_backward_matches_nn_lstm():
    # BLSTMLayer (reference models.py:212-237) = bidirectional nn.LSTM over [T, B, D] seen as
    # [B, T, D]; float64 torch on the CPU is the reference
    from audiofakedetect.lcnn import BLSTMLayer

    torch.manual_seed(3)
    bsz, steps, d = 5, 6, 64
    layer = BLSTMLayer(d, d)
    ref = torch.nn.LSTM(d, d // 2, bidirectional=True).double()
    ref.load_state_dict({k: v.double() for k, v in layer.l_blstm.state_dict().items()})
    x = torch.randn(bsz, steps, d)
    xr = x.double().requires_grad_()
    yr = ref(xr.permute(1, 0, 2))[0].permute(1, 0, 2)
    dy = torch.randn(yr.shape)
    yr.backward(dy.double())
    layer.cuda()
    xg = x.cuda().requires_grad_()
    yg = layer(xg)
    yg.backward(dy.cuda())
    assert (yg.detach().cpu().double() - yr.detach()).abs().max().item() <= 2e-6
    assert (xg.grad.cpu().double() - xr.grad).abs().max().item() <= 2e-5 * xr.grad.abs().max().item()
    refg = dict(ref.named_parameters())
    for k, p in layer.l_blstm.named_parameters():
        err = (p.grad.cpu().double() - refg[k].grad).abs().max().item()
        assert err <= 3e-5 * refg[k].grad.abs().max().item(), (k, err)


def test_lcnn_train_step_matches_cpu_restatement():
    # forward + backward of the whole LCNN (dropout off) against oracle/torch_ref.LCNNRef, which
    # tests/test_oracle_dcnn.py pins on the reference's own LCNN class
    from oracle import torch_ref
    from recipes import fill_state_dict

    g = torch.load(os.path.join(GOLD, "dcnn_lcnn_eval.pt"), map_location="cpu")
    net = LCNN()
    net.load_state_dict(fill_state_dict(g["shapes"]), strict=True)
    ref = torch_ref.LCNNRef()
    ref.load_state_dict(net.state_dict())
    for m in (net, ref):
        m.train()
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
    net.cuda()
    x = g["x"]
    labels = torch.arange(x.shape[0]) % 2
    out = net(x.cuda())
    loss = torch.nn.functional.cross_entropy(out, labels.cuda())
    loss.backward()
    out_ref = ref(x)
    loss_ref = torch.nn.functional.cross_entropy(out_ref, labels)
    loss_ref.backward()
    assert (out.detach().cpu() - out_ref.detach()).abs().max().item() <= 1e-4
    assert abs(loss.item() - loss_ref.item()) <= 1e-5
    num = den = 0.0
    refp = dict(ref.named_parameters())
    for k, p in net.named_parameters():
        assert p.grad is not None, k
        d = p.grad.cpu() - refp[k].grad
        num += d.pow(2).sum().item()
        den += refp[k].grad.pow(2).sum().item()
    # A max-feature-map / max-pool decision at a near tie can fall the other way under another
    # fp32 summation order; one such flip re-routes one gradient element and shows up as ~2e-3
    # relative L2 in every layer upstream of it, while the layers downstream of it agree to ~5e-6
    # (tools/lcnn_grad_debug.py prints the per-tensor picture; see also test_dcnn_gpu.py).
    assert (num / den) ** 0.5 <= 2e-2, (num / den) ** 0.5
    # the recurrent / linear tail is downstream of every max decision: near-exact
    for k, p in net.named_parameters():
        if k.startswith(("lstm.", "fc.")):
            err = (p.grad.cpu() - refp[k].grad).norm().item() / refp[k].grad.norm().item()
            assert err <= 1e-4, (k, err)


@pytest.mark.parametrize("cin,cout,k,pad,h,w,bn", [(1, 64, 5, 2, 21, 40, False), (32, 64, 1, 0, 10, 24, True),
                                                     (32, 96, 3, 1, 10, 24, True), (48, 128, 3, 1, 9, 17, False),
                                                     (64, 64, 3, 1, 12, 32, True), (48, 96, 1, 0, 7, 19, True)])
def test_lcnn_nhwc_bf16_layers(cin, cout, k, pad, h, w, bn):
    """csrc/lcnn_nhwc.hip: Conv2d -> MaxFeatureMap2D [-> BatchNorm2d(affine=False), evaluation mode] on channels-last
    bf16 tensors, against float64 on the SAME bf16-rounded operands (folded weights rounded as the kernel rounds
    them): 1e-5 of the largest value before the output is rounded to bf16, i.e. half a bf16 ulp (4e-3) after it.
    Then MaxPool2d(2, 2) on the bf16 result: exact."""
    from audiofakedetect import _native
    lib = _native.load()
    g = torch.Generator().manual_seed(cin * 100 + cout + k)
    n = 3
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    b = torch.randn(cout, generator=g)
    half = cout // 2
    mean = torch.randn(half, generator=g) * 0.3
    var = torch.rand(half, generator=g) + 0.5
    eps = 1e-5
    scale = torch.rsqrt(var + eps) if bn else torch.ones(half)
    shift = mean if bn else torch.zeros(half)
    # the reference computation on the operands the kernel sees: bf16 x, bf16 (w * scale), fp32 bias (b - mean) * scale
    wf = (wt * torch.cat([scale, scale]).view(-1, 1, 1, 1)).to(torch.bfloat16).double()
    bf = ((b - torch.cat([shift, shift])) * torch.cat([scale, scale])).double()
    xq = x.to(torch.bfloat16) if cin > 1 else x  # the first layer reads the fp32 image and rounds while it gathers
    yr = torch.nn.functional.conv2d(xq.to(torch.bfloat16).double(), wf, bf, padding=pad)
    yr = torch.maximum(yr[:, :half], yr[:, half:])
    buf = torch.empty(lib.afd_lcnn_prep_bytes(cin, cout, k), dtype=torch.uint8, device="cuda")
    wc, bc = wt.cuda(), b.cuda()
    mc, vc = (mean.cuda(), var.cuda()) if bn else (None, None)
    _native.check(lib.afd_lcnn_prep_conv_bf16(_native.ptr(wc), _native.ptr(bc), _native.ptr(mc), _native.ptr(vc), eps,
                                              _native.ptr(buf), cin, cout, k, _native.stream_ptr()), "prep")
    ho, wo = h + 2 * pad - (k - 1), w + 2 * pad - (k - 1)
    y = torch.empty(n, ho, wo, half, dtype=torch.bfloat16, device="cuda")
    if cin == 1:
        xc = x[:, 0].contiguous().cuda()
        _native.check(lib.afd_lcnn_conv1_nhwc_bf16(_native.ptr(xc), _native.ptr(buf), _native.ptr(y), n, h, w, cout, k, pad,
                                                   0, _native.stream_ptr()), "conv1")
    else:
        xc = xq.permute(0, 2, 3, 1).contiguous().cuda()
        _native.check(lib.afd_lcnn_conv_nhwc_bf16(_native.ptr(xc), _native.ptr(buf), _native.ptr(y), n, h, w, cin, cout, k,
                                                  pad, 0, _native.stream_ptr()), "conv")
    got = y.float().cpu().permute(0, 3, 1, 2).double()
    scale_ref = yr.abs().max().item()
    assert (got - yr).abs().max().item() <= 4.5e-3 * scale_ref
    # pooling of the bf16 tensor: exact, both output types
    for f32 in (0, 1):
        z = torch.empty(n, ho // 2, wo // 2, half, dtype=torch.float32 if f32 else torch.bfloat16, device="cuda")
        _native.check(lib.afd_lcnn_pool_nhwc_bf16(_native.ptr(y), _native.ptr(z), n, ho, wo, half, f32,
                                                  _native.stream_ptr()), "pool")
        zr = torch.nn.functional.max_pool2d(y.float().cpu().permute(0, 3, 1, 2), 2, 2)
        assert torch.equal(z.float().cpu().permute(0, 3, 1, 2), zr)
        # the pool in the convolution's epilogue (the four pixels of a window in the four lanes of a quad): the same
        # values bit for bit -- rounding to bf16 is monotone, so it commutes with the maximum
        if f32 and cin == 1:
            continue  # the first layer has no fp32 form
        zf = torch.full_like(z, float("nan"))
        if cin == 1:
            _native.check(lib.afd_lcnn_conv1_nhwc_bf16(_native.ptr(xc), _native.ptr(buf), _native.ptr(zf), n, h, w, cout, k,
                                                       pad, 1, _native.stream_ptr()), "conv1 + pool")
        else:
            _native.check(lib.afd_lcnn_conv_nhwc_bf16(_native.ptr(xc), _native.ptr(buf), _native.ptr(zf), n, h, w, cin, cout,
                                                      k, pad, 2 if f32 else 1, _native.stream_ptr()), "conv + pool")
        assert torch.equal(zf, z), (f32, (zf.float() - z.float()).abs().max().item())


def test_lstm_step_bf16_matches_the_two_launch_form():
    """afd_lstm_step_bf16 (recurrent projection + cell in one launch) against the GEMM + cell kernels it replaces in the
    bf16 evaluation path: same bf16-rounded operands, fp32 accumulation in a different order -- 1e-5 of the largest
    output over six steps of both directions; and against nn.LSTM in float64 at the bf16 bar."""
    from audiofakedetect import _native
    from audiofakedetect.lcnn import blstm_forward_bf16
    lib = _native.load()
    torch.manual_seed(4)
    m = torch.nn.LSTM(64, 32, batch_first=True, bidirectional=True).cuda()
    x = torch.randn(40, 6, 64, device="cuda")
    whh = {}
    for sfx in ("", "_reverse"):
        w = getattr(m, "weight_hh_l0" + sfx).detach().contiguous()
        wb = torch.empty(w.shape, dtype=torch.bfloat16, device="cuda")
        _native.check(lib.afd_f32_to_bf16(_native.ptr(w), _native.ptr(wb), w.numel(), _native.stream_ptr()), "cvt")
        assert torch.equal(wb, w.to(torch.bfloat16))
        whh[sfx] = wb
    with torch.no_grad():
        two = blstm_forward_bf16(x, m)
        one = blstm_forward_bf16(x, m, None, whh)
        ref, _ = m.double()(x.double())
    # `one` now runs the whole layer in one launch with hardware exp / reciprocal in the gates; h is rounded to bf16 for
    # the next step's product, so a 1e-7 difference in a gate can flip a rounding (2^-9 relative of one h): the bar
    # between two bf16 recurrences is a few bf16 ulp of the largest output, the mean difference stays tiny
    assert (one - two).abs().max().item() <= 2e-3 * two.abs().max().item()
    assert (one - two).abs().mean().item() <= 2e-5 * two.abs().max().item()
    assert (one.double() - ref).abs().max().item() <= 2e-2 * ref.abs().max().item()


@pytest.mark.parametrize("hidden,batch", [(256, 70), (64, 33), (32, 5)])
def test_blstm_layer_in_one_launch_equals_the_stepwise_form(hidden, batch):
    """afd_blstm_layer_bf16 (every step of both directions of a layer in one launch: h in LDS, c in registers) against
    the per-step launches it replaces: the same bf16 operands and an fp32 cell (hardware exp / reciprocal in the layer
    kernel) -- at the model's hidden size (256), with ragged batch blocks, and at small sizes."""
    import os
    from audiofakedetect import _native
    from audiofakedetect.lcnn import blstm_forward_bf16
    lib = _native.load()
    torch.manual_seed(hidden + batch)
    m = torch.nn.LSTM(2 * hidden, hidden, batch_first=True, bidirectional=True).cuda()
    x = torch.randn(batch, 6, 2 * hidden, device="cuda")
    whh = {}
    for sfx in ("", "_reverse"):
        w = getattr(m, "weight_hh_l0" + sfx).detach().contiguous()
        wb = torch.empty(w.shape, dtype=torch.bfloat16, device="cuda")
        _native.check(lib.afd_f32_to_bf16(_native.ptr(w), _native.ptr(wb), w.numel(), _native.stream_ptr()), "cvt")
        whh[sfx] = wb
    with torch.no_grad():
        layer = blstm_forward_bf16(x, m, None, whh)
        os.environ["AFD_LSTM_STEPWISE"] = "1"
        try:
            steps = blstm_forward_bf16(x, m, None, whh)
        finally:
            del os.environ["AFD_LSTM_STEPWISE"]
        ref, _ = m.double()(x.double())
    assert layer.shape == steps.shape == (batch, 6, 2 * hidden)
    # two bf16 recurrences: a few bf16 ulp at the worst element (a flipped rounding of one h), tiny on average
    assert (layer - steps).abs().max().item() <= 2e-3 * steps.abs().max().item()
    assert (layer - steps).abs().mean().item() <= 2e-5 * steps.abs().max().item()
    assert (layer.double() - ref).abs().max().item() <= 2e-2 * ref.abs().max().item()
