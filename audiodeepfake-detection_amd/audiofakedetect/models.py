"""Model plugin of the hot path: DCNN (and the ``get_model`` factory) on libafd_hip.

Keeps the reference's plugin contract (``src/audiofakedetect/models.py``): a module class
takes ``args: DotDict`` (reads ``input_dim, ochannels1..5, kernel1, dropout_cnn,
dropout_lstm, time_dim_add, flattend_size, ddp``, models.py:255-299), maps
``x[B, C, P, T] -> logits[B, 2]``, has ``get_name()``, and its ``state_dict()`` carries the
reference's key names (``cnn.N.*``, ``dil_conv.N.*``, ``fc.1.*``) so the shipped
checkpoints load after stripping ``module.``.

The torch.nn layer objects below only HOLD parameters / buffers (and give the reference's
default initialisation); ``forward`` never calls them -- it drives the fused HIP kernels:
conv (MFMA implicit GEMM) -> [PReLU+MaxPool] -> BatchNorm with the PReLU folded into its
input, dropout folded into the cnn->dil_conv permute, Linear+mean in one kernel.
"""

from __future__ import annotations

from types import SimpleNamespace

import torch
import torch.nn as nn

from . import ops
from .utils import DotDict


class _Unit:
    """One block of the convolution stack as the plan decided it (`DCNN._plan`)."""

    def __init__(self, **kw) -> None:
        self.form = ""              # conv1_pool | onepass | bn_conv1x1 | wino_pool | conv
        self.fold_next = False      # the block's BatchNorm is folded into the next block's 1x1 convolution
        self.stats = False          # the block's launch also produces the following BatchNorm's batch sums
        self.epilogue_stats = False  # (what the input fold's applicability asks about that launch)
        self.pool_link = False      # pool -> BatchNorm hand-over exists
        self.sum_link = False       # conv -> BatchNorm hand-over exists
        self.pool_pass = False      # PReLU + max-pool is a pass of its own
        self.defer = False          # the block's BatchNorm leaves its normalisation to the next 3x3 convolution
        self.in_fold = False        # ... and this block's convolution applies the previous block's while it loads
        self.in_shape = self.out_shape = None
        self.text: list = []
        self.__dict__.update(kw)

    def describe(self, cnn) -> list:
        """The `last_plan` entries of this unit."""
        b = f"block{self.step + 1}: "
        on_load = " | input bn applied on load" if self.in_fold else ""
        sums = " | bn sums from the epilogue" if self.stats else ""
        if self.form == "conv1_pool":
            return [b + "conv1+prelu+pool" + (" | bn folded into the next 1x1" if self.fold_next else "")]
        if self.form == "onepass":
            return [b + "bn+conv1x1+prelu+bn one-pass backward" + (" | bn applied by the next conv" if self.defer else "")]
        if self.form == "bn_conv1x1":
            out = [b + "bn folded into conv1x1"]
        elif self.form == "wino_pool":
            out = [b + "conv3x3+prelu+pool (winograd epilogue)" + on_load + sums]
        else:
            k = cnn[self.conv_i].kernel_size
            out = [b + f"conv{k[0]}x{k[1]}" + on_load + sums]
        if self.pool_pass:
            out.append(b + "prelu+pool pass")
        return out


class DCNN(nn.Module):
    """Deep CNN with dilated convolutions (reference models.py:240-317)."""

    def __init__(self, args: DotDict) -> None:
        super().__init__()
        oc = [args.input_dim[1], args.ochannels1, args.ochannels2, args.ochannels3,
              args.ochannels4, args.ochannels5, 64]
        # (kernel, padding, pooled, normalised) per conv block
        blocks = ((args.kernel1, 2, True, True), (1, 0, False, True), (3, 1, True, True),
                  (3, 1, False, True), (3, 1, False, True), (3, 1, True, False))
        layers = []
        self._cnn_plan = []  # (conv idx, prelu idx, pooled, bn idx or None)
        for i, (k, pad, pooled, normed) in enumerate(blocks):
            conv_i = len(layers)
            layers += [nn.Conv2d(oc[i], oc[i + 1], k, stride=1, padding=pad), nn.PReLU()]
            if pooled:
                layers.append(nn.MaxPool2d(2, 2))
            bn_i = None
            if normed:
                bn_i = len(layers)
                layers.append(nn.BatchNorm2d(oc[i + 1], affine=False))
            self._cnn_plan.append((conv_i, conv_i + 1, pooled, bn_i))
        layers.append(nn.Dropout(args.dropout_cnn))
        self.cnn = nn.Sequential(*layers)

        time_dim = args.input_dim[-1] // 8 + args.time_dim_add
        dil = []
        for k, pad, d in ((3, 1, 1), (5, 2, 2), (7, 2, 4)):
            dil += [nn.BatchNorm2d(time_dim, affine=True),
                    nn.Conv2d(time_dim, time_dim, k, 1, padding=pad, dilation=d), nn.PReLU()]
        dil.append(nn.Dropout(args.dropout_lstm))
        self.dil_conv = nn.Sequential(*dil)
        self.fc = nn.Sequential(nn.Flatten(2), nn.Linear(args.flattend_size, 2))
        self.single_gpu = not args.ddp
        self.sync_bn = bool(args.ddp)
        self._plans: dict = {}
        self.last_plan: list = []

    # -- the fusion plan --------------------------------------------------------------------------------------------
    def _plan(self, shape, in_grad: bool) -> tuple:
        """The fused units of the convolution stack for an NCHW input of `shape`, decided ONCE per (geometry, mode):
        a tuple of `_Unit` records that `forward` executes without deciding anything again.

        Pass 1 walks the blocks with shape arithmetic only and picks each block's launch form; pass 2 lets a block's
        BatchNorm leave its normalisation to the NEXT block's 3x3 convolution (`defer`) from what pass 1 recorded for
        that block -- its form and whether its launch produces the following BatchNorm's batch sums.  Cached per
        (shape, training, gradient mode, input gradient, data-parallel statistics, the AFD_NO_* switches)."""
        key = (tuple(shape), self.training, torch.is_grad_enabled(), bool(in_grad), self.sync_bn, ops.plan_switches())
        hit = self._plans.get(key)
        if hit is not None:
            return hit
        plan, cnn, training = self._cnn_plan, self.cnn, self.training
        n, c, hh, ww = shape
        units = []
        pending = False  # a BatchNorm waiting to be folded into the 1x1 convolution after it
        grad = bool(in_grad)
        for step, (conv_i, prelu_i, pooled, bn_i) in enumerate(plan):
            conv = cnn[conv_i]
            nxt = plan[step + 1] if step + 1 < len(plan) else None
            fold_next = (bn_i is not None and pooled and nxt is not None
                         and ops.bn_conv1x1_applicable(cnn[bn_i], cnn[nxt[0]]))
            in_shape = (n, c, hh, ww)
            k, pad, dil = conv.kernel_size[0], conv.padding[0], conv.dilation[0]
            zh, zw = hh + 2 * pad - dil * (k - 1), ww + 2 * pad - dil * (k - 1)
            cout = conv.out_channels
            u = _Unit(step=step, conv_i=conv_i, prelu_i=prelu_i, pooled=pooled, bn_i=bn_i, fold_next=fold_next)
            if (pooled and conv.in_channels == 1 and conv.kernel_size == (3, 3) and conv.dilation == (1, 1) and not grad):
                u.form = "conv1_pool"  # single-channel first block: conv + PReLU + pool in one kernel
                u.stats = fold_next and training  # the folded BatchNorm takes its batch sums from this launch
            elif (pending and not pooled and bn_i is not None
                  and ops.bn_conv1x1_prelu_bn_applicable(cnn[plan[step - 1][3]], conv, cnn[bn_i])):
                u.form = "onepass"  # BatchNorm -> 1x1 convolution -> PReLU -> BatchNorm with a one-pass backward
            else:
                u.pool_link = pooled and bn_i is not None and not fold_next
                probe = SimpleNamespace(is_cuda=True, shape=in_shape)
                if pending:
                    u.form = "bn_conv1x1"  # BatchNorm (no affine) -> 1x1 convolution: one pass
                elif pooled and ops.conv3x3_prelu_maxpool_applicable(probe, conv):
                    u.form = "wino_pool"   # 3x3 conv + PReLU + 2x2 max-pool in the Winograd epilogue
                    u.stats = u.pool_link and training
                    # ... and whether that launch will produce the sums (what the input fold's applicability asks)
                    u.epilogue_stats = (bn_i is not None and not fold_next and bool(
                        ops._lib().afd_conv3x3_forward_stats_applicable(c, hh, ww, cout, 1)))
                else:
                    u.form = "conv"
                    u.sum_link = not pooled and bn_i is not None and conv.bias is not None
                    u.stats = u.sum_link and training
                    u.epilogue_stats = u.sum_link
                u.pool_pass = pooled and u.form != "wino_pool"
            u.in_shape = in_shape
            pending = fold_next
            grad = grad or (torch.is_grad_enabled() and any(p.requires_grad for p in conv.parameters()))
            hh, ww, c = (zh // 2, zw // 2, cout) if pooled else (zh, zw, cout)
            u.out_shape = (n, c, hh, ww)  # what the block's BatchNorm sees (pooled tensor, or z)
            units.append(u)
        # pass 2: may the BatchNorm that ends block s leave its normalisation to block s + 1's convolution?
        for s_, u in enumerate(units[:-1]):
            v = units[s_ + 1]
            if u.form == "conv1_pool":
                # only the one-pass block consumes the first launch's batch sums (`bn_conv1x1` takes its own pass)
                u.stats = u.stats and v.form == "onepass"
            if not training or u.bn_i is None or u.fold_next or u.form == "conv1_pool":
                continue
            if v.form not in ("wino_pool", "conv") or (v.pooled and v.form != "wino_pool") or cnn[v.conv_i].in_channels == 1:
                continue
            u.defer = ops.conv3x3_input_fold_applicable(cnn[u.bn_i], cnn[v.conv_i], u.out_shape, v.pooled,
                                                        v.epilogue_stats)
            v.in_fold = u.defer
        for u in units:
            u.text = u.describe(cnn)
        self._plans[key] = hit = tuple(units)
        return hit

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        # [batch, channels, packets, time] -> NCHW [batch, channels, time, packets]
        h = x.permute(0, 1, 3, 2)
        if not h.is_contiguous():
            # STFT features are dense [B, C, F, T]: one tiled transpose on the GPU
            h = ops.transpose_contiguous(x.contiguous())
        cnn = self.cnn
        units = self._plan(tuple(h.shape), h.requires_grad)
        # the fused units this forward pass runs, in order: one entry per launch group (read by tests / tools)
        self.last_plan = [t for u in units for t in u.text]
        # Run-time hand-overs between the autograd functions of neighbouring units (tensors of THIS step: batch sums,
        # gradient terms); which of them exist is the plan's decision:
        link = None     # block 1 -> block 2: the folded BatchNorm's batch sums forward, a gradient term backward
        bn_link = None  # a BatchNorm -> the 3x3 convolution after it: statistics forward (a deferred BatchNorm's
        #                 (mean, invstd)), the BatchNorm's backward sums from that convolution's backward-data launch
        z = None
        for u in units:
            conv, slope = cnn[u.conv_i], cnn[u.prelu_i].weight
            bn = cnn[u.bn_i] if u.bn_i is not None else None
            if u.form == "conv1_pool":
                link = {} if u.fold_next else None
                if u.stats:
                    link["want_stats"] = True
                h = ops.conv1_prelu_maxpool(h, conv.weight, conv.bias, slope, conv.padding[0], link)
                if bn is not None and not u.fold_next:
                    h = ops.batch_norm(h, bn, None, self.sync_bn)
                continue
            if u.form == "onepass":
                bn_link = {}
                h = ops.bn_conv1x1_prelu_bn(h, cnn[units[u.step - 1].bn_i], conv.weight, conv.bias, slope, bn,
                                            self.sync_bn, link, bn_link, u.defer)
                link = None
                continue
            link = None
            in_link, bn_link = bn_link, None
            pool_link = {} if u.pool_link else None  # pool -> the BatchNorm right behind it (its backward runs in the pool's)
            sum_link = {} if u.sum_link else None    # conv -> PReLU -> BatchNorm: the bias gradient comes from the BatchNorm's backward
            if u.form == "bn_conv1x1":
                z = ops.bn_conv1x1(h, cnn[units[u.step - 1].bn_i], conv.weight, conv.bias, self.sync_bn)
            elif u.form == "wino_pool":
                if u.stats:
                    pool_link["want_stats"] = True  # the BatchNorm behind it takes its batch sums from that launch
                h = ops.conv3x3_prelu_maxpool(h, conv.weight, conv.bias, slope, in_link, pool_link)
            else:
                if u.stats:
                    sum_link["want_stats"] = True  # the BatchNorm of PReLU(z) takes its batch sums from this launch
                    sum_link["stats_slope"] = slope
                z = ops.conv2d(h, conv.weight, conv.bias, conv.padding[0], conv.dilation[0], pooled=u.pooled,
                               bn_link=in_link, out_link=sum_link)
            if u.pooled:
                if u.pool_pass:
                    h = ops.prelu_maxpool2x2(z, slope, pool_link)
                if bn is not None and not u.fold_next:
                    bn_link = {}
                    h = ops.batch_norm(h, bn, None, self.sync_bn, bn_link, pool_link, defer=u.defer)
            else:
                bn_link = {}
                h = ops.batch_norm(z, bn, slope, self.sync_bn, bn_link, sum_link=sum_link, defer=u.defer)
        # Dropout + [batch, channels, time, packets] -> [batch, time, channels, packets]
        h = ops.dropout_permute(h, cnn[-1].p, self.training)
        dil = self.dil_conv
        slope = None
        z = h
        for j in range(3):
            bn, conv = dil[3 * j], dil[3 * j + 1]
            h = ops.batch_norm(z, bn, slope, self.sync_bn)
            z = ops.conv2d(h, conv.weight, conv.bias, conv.padding[0], conv.dilation[0])
            slope = dil[3 * j + 2].weight
        h = ops.prelu_dropout(z, slope, dil[-1].p, self.training)
        lin = self.fc[1]
        b, td = h.shape[0], h.shape[1]
        return ops.linear_mean(h.reshape(b, td, -1), lin.weight, lin.bias)

    def get_name(self) -> str:
        return "DCNN"


def strip_ddp_prefix(state_dict: dict) -> dict:
    """Checkpoints of the reference carry ``module.module.`` (double DDP wrap)."""
    out = {}
    for k, v in state_dict.items():
        while k.startswith("module."):
            k = k[len("module."):]
        out[k] = v
    return out


def get_model(args: DotDict, model_name: str, nclasses: int = 2, in_channels: int = 1,
              lead: bool = False) -> nn.Module:
    """Model factory (reference models.py:710-765); 'modules' instantiates ``args.module``."""
    if model_name == "modules":
        model = args.module(args)
    elif model_name == "lcnn":
        from .lcnn import LCNN

        model = LCNN(classes=nclasses, in_channels=in_channels,
                     lstm_channels=args.num_of_scales if args.features == "none" else 60)
    else:
        raise NotImplementedError(f"Model {model_name!r} is outside the hot path (SURVEY.md #16)")
    return model
