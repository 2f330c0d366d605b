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
    # Gradients: fp32 rounding of a conv output can flip a 2x2 max-pool argmax at a near
    # tie or the PReLU branch at a zero crossing; each flip re-routes one gradient element, so
    # element-wise agreement with another fp32 implementation is bounded by those flips, not
    # by accumulation error (measured with tools/grad_debug.py: with no flip between two
    # runs every tensor agrees to ~1e-6; one pool flip in block 3 costs 1e-3 L2 upstream of
    # it, and the CPU fp32 reference differs from its own fp64 run by 2e-4 .. 2e-3 for the
    # same reason).  Bars: per tensor relative L2 <= 2e-2 and max <= 1e-1 of its largest
    # entry; over all parameters together relative L2 <= 5e-3.  (The overall figure IS the flip
    # noise: 2.9e-3 with the first BatchNorm's sums from a pass over its input, 3.5e-3 with the
    # same sums from the first block's epilogue (round 5) -- whose logits (1.2e-6 against 1.5e-6)
    # and running mean (3.7e-9 against 7.5e-9) are CLOSER to the reference; the routing-aligned
    # test below holds 1e-4 per tensor either way.)
    num = den = 0.0
    for k, p in net.named_parameters():
        ref = g["grads"][k]
        diff = (p.grad.cpu() - ref)
        l2 = diff.norm().item() / (ref.norm().item() + 1e-30)
        mx = diff.abs().max().item() / (ref.abs().max().item() + 1e-30)
        assert l2 <= 2e-2 and mx <= 1e-1, f"grad {k}: rel L2 {l2:.3e}, rel max {mx:.3e}"
        num += diff.norm().item() ** 2
        den += ref.norm().item() ** 2
    assert (num / den) ** 0.5 <= 5e-3
    opt.step()
    after = net.state_dict()
    for k, v in g["state_dict_after"].items():
        if v.dtype.is_floating_point:
            # Adam's first step moves every parameter by ~lr * sign(grad): a parameter whose
            # gradient is within rounding of zero may step the other way (2 * lr = 8e-4)
            d = (after[k].cpu() - v).abs()
            assert d.max().item() <= 2.1 * g["lr"], k
            assert int((d > 2e-5).sum()) <= max(2, d.numel() // 100), k
        else:
            assert int(after[k]) == int(v), k


def _routing_flips(taps, routing):
    """Per layer: positions where the GPU run's routing decision differs from the reference's."""
    import numpy as np

    assert len(taps) == len(routing), (len(taps), [r["name"] for r in routing])
    out = []
    for (kind, t), rec in zip(taps, routing):
        assert kind == rec["kind"], (kind, rec["name"])
        if kind == "pool":
            ref = rec["code"].to(t.device)
            assert ref.shape == t.shape, rec["name"]
            out.append((rec["name"], kind, t, ref, t != ref))
        else:
            n = int(np.prod(rec["shape"]))
            ref = torch.from_numpy(np.unpackbits(rec["bits"].numpy())[:n].astype(bool)).reshape(rec["shape"]).to(t.device)
            assert ref.shape == t.shape, rec["name"]
            out.append((rec["name"], kind, t, ref, (t <= 0) != ref))
    return out


def test_train_step_gradients_with_reference_routing():
    """Demonstrates the claim behind the loose gradient bars above: the only disagreement with
    the reference's fp32 run is WHICH way a max-pool near tie / a PReLU zero crossing went.

    tests/golden/dcnn_train_step_routing.pt holds the reference run's pool argmax codes and PReLU
    branch masks (make_golden.record_routing).  The GPU forward's own decisions (the tensors it
    saves for backward, exposed through ops.debug_taps) differ from them in a handful of
    positions out of 8.6 M; with those positions set to the reference's decision before backward
    (a pool code byte, or the sign of a pre-activation that is within rounding of zero), EVERY
    gradient tensor agrees with the reference to 1e-4 of its largest entry."""
    g = torch.load(os.path.join(GOLD, "dcnn_train_step.pt"), map_location="cpu")
    routing = torch.load(os.path.join(GOLD, "dcnn_train_step_routing.pt"), map_location="cpu")["routing"]
    net = DCNN(_args(g["x"].shape, dropout_cnn=0.0, dropout_lstm=0.0))
    net.load_state_dict(g["state_dict"], strict=True)
    net.cuda().train()
    opt = ops.FusedAdam(net.parameters(), lr=g["lr"], weight_decay=g["weight_decay"])
    opt.zero_grad()
    ops.debug_taps = []
    try:
        out = net(g["x"].cuda())
        taps = ops.debug_taps
    finally:
        ops.debug_taps = None
    loss = ops.CrossEntropyLoss()(out, g["labels"].cuda())
    total = flips = 0
    for name, kind, t, ref, diff in _routing_flips(taps, routing):
        nd = int(diff.sum())
        total += diff.numel()
        flips += nd
        if nd == 0:
            continue
        if kind == "pool":
            t.data[diff] = ref[diff]
        else:
            # a flipped PReLU branch means |z| is at rounding level: give z the reference's sign
            # (the F(4x4) / F(2x4) layers are within 2e-5 of their largest output, which is O(1) here)
            assert t.data[diff].abs().max().item() <= 2e-5, name
            tiny = torch.full_like(t.data[diff], 1e-30)
            t.data[diff] = torch.where(ref[diff], -tiny, tiny)
    assert flips <= max(8, total // 100000), f"{flips} routing differences in {total} decisions"
    loss.backward()
    for k, p in net.named_parameters():
        ref = g["grads"][k]
        mx = (p.grad.cpu() - ref).abs().max().item() / (ref.abs().max().item() + 1e-30)
        assert mx <= 1e-4, f"grad {k}: max diff {mx:.3e} of the largest entry ({flips} flips patched)"


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


PLAN_L14 = ["block1: conv1+prelu+pool | bn folded into the next 1x1",
            "block2: bn+conv1x1+prelu+bn one-pass backward | bn applied by the next conv",
            "block3: conv3x3+prelu+pool (winograd epilogue) | input bn applied on load | bn sums from the epilogue",
            "block4: conv3x3 | input bn applied on load | bn sums from the epilogue",
            "block5: conv3x3 | input bn applied on load | bn sums from the epilogue",
            "block6: conv3x3+prelu+pool (winograd epilogue) | input bn applied on load"]


@pytest.mark.parametrize("input_dim,flat,add", [((2, 1, 16384, 24), 80960, 0), ((2, 1, 256, 109), 320, 0),
                                                ((2, 1, 256, 95), 320, 1), ((2, 1, 256, 101), 320, 0)])
def test_the_fused_units_of_a_training_step_are_the_documented_plan(input_dim, flat, add):
    """`DCNN.last_plan` lists the fused units a forward pass ran.  In training mode every BASELINE / shipped geometry
    (coif4 level 14, coif4 / sym5 level 8, STFT) runs the SAME six units -- the plan DESIGN.md section 4.5 describes;
    in evaluation mode the BatchNorms are plain passes and nothing is deferred."""
    torch.manual_seed(0)
    args = _args(input_dim, flattend_size=flat, dropout_cnn=0.0, dropout_lstm=0.0, time_dim_add=add)
    net = DCNN(args).cuda().train()
    n, _, p, t = input_dim
    x = torch.randn(n, 1, t, p, device="cuda").permute(0, 1, 3, 2)
    out = net(x)
    assert net.last_plan == PLAN_L14, net.last_plan
    ops.CrossEntropyLoss()(out, torch.tensor([0, 1], device="cuda")).backward()
    net.eval()
    with torch.no_grad():
        net(x)
    assert not any("applied on load" in u or "applied by the next" in u or "one-pass" in u for u in net.last_plan), net.last_plan
    assert net.last_plan[0].startswith("block1: conv1+prelu+pool") and len(net.last_plan) == 6


@pytest.mark.parametrize("input_dim,flat,expect", [((3, 1, 16384, 24), 80960, 4), ((3, 1, 256, 101), 320, 4)])
def test_training_step_without_the_normalised_tensors(input_dim, flat, expect, monkeypatch):
    """The BatchNorms in front of blocks 3-6 hand their statistics to the next convolution instead of writing their
    result (ops.batch_norm(defer=True), decided by DCNN._plan): at the level-14 and the level-8 geometry all four do,
    and logits and every gradient equal those of the step with AFD_NO_INPUT_FOLD=1 to the run-to-run noise of either
    path (the layer-level test holds the launches bit-equal)."""
    torch.manual_seed(3)
    args = _args(input_dim, flattend_size=flat, dropout_cnn=0.0, dropout_lstm=0.0)
    net = DCNN(args).cuda().train()
    n, _, p, t = input_dim
    x = torch.randn(n, 1, t, p, device="cuda").permute(0, 1, 3, 2)
    labels = torch.tensor([0, 1, 1][:n], device="cuda")
    res = []
    for off in (False, True):
        if off:
            monkeypatch.setenv("AFD_NO_INPUT_FOLD", "1")
        net.zero_grad()
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.reset_running_stats()
        out = net(x)
        loss = ops.CrossEntropyLoss()(out, labels)
        loss.backward()
        # (the plan's units whose convolution applies the previous BatchNorm while it loads)
        assert sum("input bn applied on load" in u for u in net.last_plan) == (0 if off else expect), net.last_plan
        res.append((out.detach().clone(), {k: v.grad.clone() for k, v in net.named_parameters()},
                    {k: v.clone() for k, v in net.named_buffers() if "running" in k}))
    # (two runs of ONE path already differ in the last bits -- the BatchNorm sums end in double atomics, the slopes in
    # float atomics -- and a near-tie of a max-pool window may then go the other way: the bar is on the relative L2
    # error of each gradient, which a handful of such windows cannot reach)
    assert (res[0][0] - res[1][0]).abs().max().item() < 1e-5
    for k in res[0][1]:
        a, b = res[0][1][k].double(), res[1][1][k].double()
        tol = 3e-4 if a.numel() == 1 else 1e-4
        assert (a - b).norm().item() <= tol * (b.norm().item() + 1e-30), k
    for k in res[0][2]:
        assert torch.allclose(res[0][2][k], res[1][2][k], rtol=1e-6, atol=1e-7), k


@pytest.mark.parametrize("tag", ["coif4", "sym5"])
def test_level14_eval_matches_reference_class(tag):
    """BASELINE configs[1] / [2] geometry against the REFERENCE CLASS itself: tests/golden/dcnn_level14_eval.pt holds
    the eval logits of the reference's DCNN (models.py:240-313) on [2,1,16384,24] (coif4) and [2,1,16384,10] (sym5),
    flattend_size 80 960 (tests/golden/make_golden.py level14).  Weights and inputs are rebuilt from the recipes the
    generator used; logits within 1e-4, labels bit-exact."""
    import sys

    sys.path.insert(0, GOLD)
    from recipes import fill_dcnn_state_dict, level14_input

    g = torch.load(os.path.join(GOLD, "dcnn_level14_eval.pt"), map_location="cpu")[tag]
    x = level14_input(g["t_len"])
    assert torch.equal(x.flatten()[:16], g["x_head"])
    assert torch.allclose(torch.stack([x.double().sum(), (x.double() ** 2).sum()]), g["x_digest"], rtol=1e-12)
    net = DCNN(_args(x.shape, flattend_size=80960))
    assert {k: tuple(v.shape) for k, v in net.state_dict().items()} == g["shapes"]
    net.load_state_dict(fill_dcnn_state_dict(g["shapes"]), strict=True)
    net.cuda().eval()
    with torch.no_grad():
        y = net(x.cuda())  # the permuted view [B, C, P, T] over memory [B, C, T, P], as Packets returns it
    assert (y.cpu() - g["logits"]).abs().max().item() <= 1e-4
    assert torch.equal(y.argmax(-1).cpu(), g["labels"])


@pytest.mark.parametrize("wavelet,t_len", [("coif4", 24), ("sym5", 10)])
def test_full_width_level14_step_matches_cpu_restatement(wavelet, t_len):
    """BASELINE configs[1] / configs[2] geometry at full width (16384 packets x 24 time steps for
    coif4; x 10 for sym5, where time_dim = 10 // 8 = 1 and block 6 collapses to one row -- reference
    models.py:280), 2 frames.

    Exercises the wide-image kernels (conv3x3 / wgrad3x3 / conv1x1 / dilconv / fused conv1)
    inside the real model: features -> DCNN -> loss -> backward against oracle/torch_ref.DCNNRef
    on the CPU (fp32).  Bars as in test_train_step_matches_reference: logits 1e-4, loss 1e-5,
    gradients bounded by pool / PReLU near-tie flips (relative L2 over all parameters <= 5e-3).
    """
    from oracle import torch_ref
    from audiofakedetect.wavelet_math import Packets

    torch.manual_seed(11)
    x = (0.1 * torch.randn(2, 1, 22050)).clamp_(-1, 1)
    feats, _ = Packets(wavelet, max_lev=14, log_scale=True)(x.cuda())  # view [B, 1, 16384, T]
    feats = (feats - feats.mean()) / feats.std()
    assert tuple(feats.shape) == (2, 1, 16384, t_len)
    args = _args(feats.shape, flattend_size=40 * (16384 // 8 - 24), dropout_cnn=0.0, dropout_lstm=0.0)
    net = DCNN(args)
    ref = torch_ref.DCNNRef(args.input_dim, dropout_cnn=0.0, dropout_lstm=0.0,
                            flattend_size=args.flattend_size)
    ref.load_state_dict(net.state_dict())
    net.cuda().train()
    ref.train()
    labels = torch.tensor([0, 1])
    out = net(feats)
    loss = ops.CrossEntropyLoss()(out, labels.cuda())
    loss.backward()
    out_ref = ref(feats.cpu())
    loss_ref = torch.nn.functional.cross_entropy(out_ref, labels)
    loss_ref.backward()
    assert (out.detach().cpu() - out_ref.detach()).abs().max().item() <= 1e-4
    assert abs(loss.item() - loss_ref.item()) <= 1e-5
    assert torch.equal(out.argmax(-1).cpu(), out_ref.argmax(-1))
    num = den = 0.0
    refp = dict(ref.named_parameters())
    for k, p in net.named_parameters():
        d = (p.grad.cpu() - refp[k].grad)
        num += d.pow(2).sum().item()
        den += refp[k].grad.pow(2).sum().item()
    # (unpatched routing: a handful of near-tie pool / PReLU decisions out of 2-4e7 go the other way and each moves
    # the gradient by a whole activation's worth; test_full_width_level14_gradients_with_oracle_routing below shows
    # the arithmetic itself agrees to 1e-4 per tensor once those decisions are aligned)
    assert (num / den) ** 0.5 <= 5e-3, (num / den) ** 0.5


@pytest.mark.parametrize("wavelet,t_len", [("coif4", 24), ("sym5", 10)])
def test_full_width_level14_gradients_with_oracle_routing(wavelet, t_len):
    """The full-width step of the test above with the routing argument made explicit (as
    test_train_step_gradients_with_reference_routing does on the reference's own fixture): the CPU restatement's
    pool argmax codes and PReLU branch masks are recorded by the hooks of tests/golden/make_golden.record_routing;
    the GPU forward's decisions differ from them in a few positions per hundred thousand (near ties within the
    kernels' rounding: 1e-5 of the layer's largest value on the F(4x4) Winograd layers); with those set to the
    restatement's decision before backward, every gradient tensor agrees to 1e-4 of its largest entry (5e-4 for the
    scalar-like parameters, see below) -- the
    3e-3 bar of the unpatched comparison above is the routing, not the arithmetic."""
    import sys

    sys.path.insert(0, GOLD)
    from make_golden import record_routing
    from oracle import torch_ref
    from audiofakedetect.wavelet_math import Packets

    torch.manual_seed(11)
    x = (0.1 * torch.randn(2, 1, 22050)).clamp_(-1, 1)
    feats, _ = Packets(wavelet, max_lev=14, log_scale=True)(x.cuda())
    feats = (feats - feats.mean()) / feats.std()
    args = _args(feats.shape, flattend_size=40 * (16384 // 8 - 24), dropout_cnn=0.0, dropout_lstm=0.0)
    net = DCNN(args)
    ref = torch_ref.DCNNRef(args.input_dim, dropout_cnn=0.0, dropout_lstm=0.0, flattend_size=args.flattend_size)
    ref.load_state_dict(net.state_dict())
    ref.train()
    routing = record_routing(ref)
    labels = torch.tensor([0, 1])
    torch.nn.functional.cross_entropy(ref(feats.cpu()), labels).backward()
    net.cuda().train()
    ops.debug_taps = []
    try:
        out = net(feats)
        taps = ops.debug_taps
    finally:
        ops.debug_taps = None
    loss = ops.CrossEntropyLoss()(out, labels.cuda())
    total = flips = 0
    for name, kind, t, rt, diff in _routing_flips(taps, routing):
        nd = int(diff.sum())
        total += diff.numel()
        flips += nd
        if nd == 0:
            continue
        if kind == "pool":
            t.data[diff] = rt[diff]
        else:
            assert t.data[diff].abs().max().item() <= 1e-4, name
            tiny = torch.full_like(t.data[diff], 1e-30)
            t.data[diff] = torch.where(rt[diff], -tiny, tiny)
    assert flips <= total // 20000, f"{flips} routing differences in {total} decisions"
    loss.backward()
    refp = dict(ref.named_parameters())
    worst = {}
    for k, p in net.named_parameters():
        r = refp[k].grad
        worst[k] = (p.grad.cpu() - r).abs().max().item() / (r.abs().max().item() + 1e-30)
    # measured: <= 7e-5 for every convolution / linear weight and bias (the F(4x4) layers included); up to 3.6e-4 for
    # the one- to three-element parameters -- PReLU slopes, the dilated stack's BatchNorm weights -- whose gradient is
    # ONE cancelling sum over millions of products, accumulated in fp32 on the CPU side
    for k, mx in worst.items():
        bar = 5e-4 if refp[k].numel() <= 4 else 1e-4
        assert mx <= bar, f"grad {k}: max diff {mx:.3e} of the largest entry ({flips} of {total} decisions patched)"


def _level14_net(t_len, seed=5):
    """Eval-mode DCNN at a level-14 geometry with non-trivial running statistics and weights."""
    torch.manual_seed(seed)
    args = _args((2, 1, 16384, t_len), flattend_size=40 * (16384 // 8 - 24), dropout_cnn=0.0, dropout_lstm=0.0)
    net = DCNN(args)
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0.0, 0.2)
                m.running_var.uniform_(0.5, 1.5)
    return args, net


@pytest.mark.parametrize("wavelet,t_len", [("coif4", 24), ("sym5", 10)])
def test_level14_batch128_forward_is_batch_independent_and_matches_oracle(wavelet, t_len):
    """BASELINE configs[1]/[2] at the batch size the metric is quoted on (B = 128).

    The CPU oracle cannot run 128 level-14 frames in test time; in eval mode every frame's logits
    depend on that frame alone, so: (i) frames 0-1 of the B = 128 GPU forward must equal the B = 2
    GPU forward of the same two frames (different grid sizes / tile walks of the same kernels) to
    1e-5, (ii) the B = 2 forward matches oracle/torch_ref.DCNNRef on the CPU to 1e-4 with equal
    labels, (iii) a frame repeated at another batch position gives the same logits."""
    from oracle import torch_ref
    from audiofakedetect.wavelet_math import Packets

    args, net = _level14_net(t_len)
    ref = torch_ref.DCNNRef(args.input_dim, dropout_cnn=0.0, dropout_lstm=0.0, flattend_size=args.flattend_size)
    ref.load_state_dict(net.state_dict())
    ref.eval()
    net.cuda().eval()
    g = torch.Generator().manual_seed(3)
    x = (0.1 * torch.randn(128, 1, 22050, generator=g)).clamp_(-1, 1)
    x[77] = x[1]
    with torch.no_grad():
        tr = Packets(wavelet, max_lev=14, log_scale=True)
        tr.fused_norm = (-9.0, 3.0)
        f128, _ = tr(x.cuda())
        f2, _ = tr(x[:2].cuda())
        assert torch.equal(f128[:2], f2)  # the front end is per frame
        y128 = net(f128)
        y2 = net(f2)
        yref = ref(f2.cpu())
    assert tuple(y128.shape) == (128, 2) and torch.isfinite(y128).all()
    assert (y128[:2] - y2).abs().max().item() <= 1e-5
    assert (y128[77] - y128[1]).abs().max().item() <= 1e-5
    assert (y2.cpu() - yref).abs().max().item() <= 1e-4
    assert torch.equal(y2.argmax(-1).cpu(), yref.argmax(-1))
    assert y128.std(0).min().item() > 0  # the frames are told apart


def _slope_along_gradient(net, feats, labels):
    """(loss(theta + h g) - loss(theta - h g)) / (2 h |g|^2) for the gradient g the backward pass left in .grad: 1 when
    g is the gradient of the loss the forward pass computes (central difference: the curvature term cancels).  The loss
    is taken in float64 from the float logits; h moves the loss by about 1e-3 each way, a thousand times the noise."""
    params = list(net.parameters())
    grads = [p.grad.detach().clone() for p in params]
    saved = [p.detach().clone() for p in params]
    g2 = sum((g.double() ** 2).sum() for g in grads).item()
    h = 1e-3 / g2

    def loss_at(step):
        with torch.no_grad():
            for p, s0, g in zip(params, saved, grads):
                p.copy_(s0).add_(g, alpha=step)
            val = torch.nn.functional.cross_entropy(net(feats).double(), labels).item()
            for p, s0 in zip(params, saved):
                p.copy_(s0)
        return val

    return (loss_at(h) - loss_at(-h)) / (2 * h * g2)



@pytest.mark.parametrize("wavelet,t_len", [("sym5", 10), ("coif4", 24)])
def test_level14_batch128_train_step_properties(wavelet, t_len):
    """configs[1] / configs[2] per-GPU step at B = 128 (coif4 / sym5 level 14): finite loss and gradients,
    the bias gradient of the last layer equals the batch mean of dlogits (a closed form that needs
    no oracle), the loss's central difference along the whole-network gradient equals |g|^2 to 1 % (every backward
    kernel at the benchmark batch), and Adam reaches a lower loss within a few steps on a fixed batch."""
    from audiofakedetect.wavelet_math import Packets

    torch.manual_seed(2)
    args = _args((128, 1, 16384, t_len), flattend_size=40 * (16384 // 8 - 24), dropout_cnn=0.0, dropout_lstm=0.0)
    net = DCNN(args).cuda().train()
    g = torch.Generator().manual_seed(4)
    x = (0.1 * torch.randn(128, 1, 22050, generator=g)).clamp_(-1, 1)
    labels = torch.randint(0, 2, (128,), generator=g).cuda()
    with torch.no_grad():
        tr = Packets(wavelet, max_lev=14, log_scale=True)
        tr.fused_norm = (-9.0, 3.0)
        feats, _ = tr(x.cuda())
    opt = ops.FusedAdam(net.parameters(), lr=4e-4, weight_decay=0.0)
    losses = []
    for it in range(24):
        opt.zero_grad()
        out = net(feats)
        loss = ops.CrossEntropyLoss()(out, labels)
        loss.backward()
        torch.cuda.synchronize()
        if it == 0:
            for k, p in net.named_parameters():
                assert torch.isfinite(p.grad).all(), k
            dlogits = (torch.softmax(out.detach().double(), -1)
                       - torch.nn.functional.one_hot(labels, 2).double()) / 128
            assert (net.fc[1].bias.grad.double() - dlogits.sum(0)).abs().max().item() <= 1e-6
            slope = _slope_along_gradient(net, feats, labels)
            assert abs(slope - 1.0) <= 0.01, slope  # measured 0.99994 / 0.99998 (level 14), 0.9975 (STFT)
        losses.append(loss.item())
        opt.step()
    # the fixed batch is learned: at lr 4e-4 the first Adam steps on the 80 960-wide Linear overshoot (the loss is above
    # its start for about ten steps, profiles/r04_overfit.txt), after 24 steps it is well below it
    print("losses", [round(v, 4) for v in losses])
    assert all(map(lambda v: v == v and v < 10, losses)) and losses[-1] < 0.8 * losses[0], losses


def test_stft_dcnn_batch128_eval_and_train_step_properties():
    """configs[0] at its stated batch (STFT(n_fft 511, hop 220) + DCNN, B = 128): in evaluation mode the logits of
    the 128-frame batch equal those of its 4-frame slices run on their own (bit for bit: no kernel's result may
    depend on the batch around a frame) and the labels agree; in training mode the loss and every gradient are finite,
    the last layer's bias gradient equals the batch mean of dlogits, the central difference of the loss along the
    gradient equals |g|^2 to 1 %, and the loss falls over a few Adam steps."""
    from audiofakedetect.wavelet_math import STFTLayer

    torch.manual_seed(5)
    g = torch.Generator().manual_seed(6)
    x = (0.1 * torch.randn(128, 1, 22050, generator=g)).clamp_(-1, 1).cuda()
    labels = torch.randint(0, 2, (128,), generator=g).cuda()
    with torch.no_grad():
        feats, _ = STFTLayer(n_fft=511, hop_length=220, log_scale=True, power=2.0)(x)
    assert tuple(feats.shape) == (128, 1, 256, 101)
    feats = (feats + 9.0) / 3.0
    net = DCNN(_args(feats.shape, dropout_cnn=0.0, dropout_lstm=0.0)).cuda().eval()
    with torch.no_grad():
        full = net(feats)
        for lo in (0, 60, 124):
            part = net(feats[lo:lo + 4].contiguous())
            assert torch.equal(part, full[lo:lo + 4]), lo
    net.train()
    opt = ops.FusedAdam(net.parameters(), lr=4e-4, weight_decay=0.0)
    losses = []
    for it in range(24):
        opt.zero_grad()
        out = net(feats)
        loss = ops.CrossEntropyLoss()(out, labels)
        loss.backward()
        if it == 0:
            for k, p in net.named_parameters():
                assert torch.isfinite(p.grad).all(), k
            dlogits = (torch.softmax(out.detach().double(), -1)
                       - torch.nn.functional.one_hot(labels, 2).double()) / 128
            assert (net.fc[1].bias.grad.double() - dlogits.sum(0)).abs().max().item() <= 1e-6
            slope = _slope_along_gradient(net, feats, labels)
            assert abs(slope - 1.0) <= 0.01, slope  # measured 0.99994 / 0.99998 (level 14), 0.9975 (STFT)
        losses.append(loss.item())
        opt.step()
    print("losses", [round(v, 4) for v in losses])
    assert all(v == v and v < 10 for v in losses) and losses[-1] < 0.8 * losses[0], losses
