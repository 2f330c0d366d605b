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

    def _next_normalises(self, step: int, shape) -> bool:
        """May the BatchNorm that ends block `step`, applied to a tensor of `shape`, leave its normalisation to the
        next block's 3x3 convolution (ops.batch_norm(defer=True))?  Mirrors the branch that block takes in `forward`."""
        plan, cnn = self._cnn_plan, self.cnn
        if not self.training or step + 1 >= len(plan) or plan[step][3] is None:
            return False
        conv_i, _, pooled, bn_i = plan[step + 1]
        conv = cnn[conv_i]
        after = plan[step + 2] if step + 2 < len(plan) else None
        lib = ops._lib()
        n, cin, h, w = shape
        if pooled:
            probe = SimpleNamespace(is_cuda=True, shape=shape)
            if conv.in_channels == 1 or not ops.conv3x3_prelu_maxpool_applicable(probe, conv):
                return False
            folds_on = bn_i is not None and after is not None and ops.bn_conv1x1_applicable(cnn[bn_i], cnn[after[0]])
            want_stats = (bn_i is not None and not folds_on
                          and bool(lib.afd_conv3x3_forward_stats_applicable(cin, h, w, conv.out_channels, 1)))
        else:
            want_stats = bn_i is not None and conv.bias is not None
        return ops.conv3x3_input_fold_applicable(cnn[plan[step][3]], conv, shape, pooled, want_stats)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        # [batch, channels, packets, time] -> NCHW [batch, channels, time, packets]
        h = x.permute(0, 1, 3, 2)
        if not h.is_contiguous():
            # STFT features are dense [B, C, F, T]: one tiled transpose on the GPU
            h = ops.transpose_contiguous(x.contiguous())
        cnn = self.cnn
        # the fused units this forward pass runs, in order: one entry per launch group (read by tests / tools through
        # `last_plan`; the decisions themselves are the branches below -- shape and mode decide, nothing is cached)
        plan_log = self.last_plan = []
        pending_bn = None  # a BatchNorm waiting to be folded into the 1x1 convolution after it
        link = None        # hand-over of a gradient term from block 2's backward to block 1's
        bn_link = None     # from a BatchNorm to the 3x3 convolution after it: that layer's backward-data launch
        #                    also produces the BatchNorm's backward sums
        for step, (conv_i, prelu_i, pooled, bn_i) in enumerate(self._cnn_plan):
            conv = cnn[conv_i]
            slope = cnn[prelu_i].weight
            nxt = self._cnn_plan[step + 1] if step + 1 < len(self._cnn_plan) else None
            fold_next = (bn_i is not None and pooled and nxt is not None
                         and ops.bn_conv1x1_applicable(cnn[bn_i], cnn[nxt[0]]))
            if (pooled and conv.in_channels == 1 and conv.kernel_size == (3, 3)
                    and conv.dilation == (1, 1) and not h.requires_grad):
                # single-channel first block: conv + PReLU + pool in one kernel
                link = {} if fold_next else None
                if fold_next and self.training:
                    link["want_stats"] = True  # the folded BatchNorm takes its batch sums from the first block's launch
                plan_log.append(f"block{step + 1}: conv1+prelu+pool" + (" | bn folded into the next 1x1" if fold_next else ""))
                h = ops.conv1_prelu_maxpool(h, conv.weight, conv.bias, slope, conv.padding[0], link)
                if fold_next:
                    pending_bn = cnn[bn_i]
                elif bn_i is not None:
                    h = ops.batch_norm(h, cnn[bn_i], None, self.sync_bn)
                continue
            fused_pool = False
            if (pending_bn is not None and not pooled and bn_i is not None
                    and ops.bn_conv1x1_prelu_bn_applicable(pending_bn, conv, cnn[bn_i])):
                # training step of block 2: BatchNorm -> 1x1 convolution -> PReLU -> BatchNorm with a one-pass backward
                bn_link = {}
                # (the second BatchNorm's result is not stored when block 3's convolution can build it while it loads)
                zshape = (h.shape[0], conv.out_channels, h.shape[2], h.shape[3])
                plan_log.append(f"block{step + 1}: bn+conv1x1+prelu+bn one-pass backward"
                                + (" | bn applied by the next conv" if self._next_normalises(step, zshape) else ""))
                h = ops.bn_conv1x1_prelu_bn(h, pending_bn, conv.weight, conv.bias, slope, cnn[bn_i],
                                            self.sync_bn, link, bn_link, self._next_normalises(step, zshape))
                pending_bn = link = None
                continue
            link = None
            sum_link = None
            in_link, bn_link = bn_link, None
            # from a pool to the BatchNorm right behind it: that BatchNorm's backward runs inside the pool's
            pool_link = {} if (pooled and bn_i is not None and not fold_next) else None
            if pending_bn is not None:
                # BatchNorm (no affine) -> 1x1 convolution: one pass, normalised tensor never written
                plan_log.append(f"block{step + 1}: bn folded into conv1x1")
                z = ops.bn_conv1x1(h, pending_bn, conv.weight, conv.bias, self.sync_bn)
                pending_bn = None
            elif pooled and ops.conv3x3_prelu_maxpool_applicable(h, conv):
                # 3x3 conv + PReLU + 2x2 max-pool in the Winograd epilogue: the conv output is never written
                if pool_link is not None and self.training:
                    pool_link["want_stats"] = True  # the BatchNorm behind it takes its batch sums from that launch
                plan_log.append(f"block{step + 1}: conv3x3+prelu+pool (winograd epilogue)"
                                + (" | input bn applied on load" if in_link and in_link.get("fold") is not None else "")
                                + (" | bn sums from the epilogue" if pool_link and pool_link.get("want_stats") else ""))
                h = ops.conv3x3_prelu_maxpool(h, conv.weight, conv.bias, slope, in_link, pool_link)
                fused_pool = True
            else:
                # conv -> (PReLU) -> BatchNorm without a pool in between: the BatchNorm's backward hands the
                # per-channel sums of its result (this layer's bias gradient) over
                sum_link = {} if (not pooled and bn_i is not None and conv.bias is not None) else None
                if sum_link is not None and self.training:
                    sum_link["want_stats"] = True  # the BatchNorm of PReLU(z) takes its batch sums from this launch
                    sum_link["stats_slope"] = slope
                plan_log.append(f"block{step + 1}: conv{conv.kernel_size[0]}x{conv.kernel_size[1]}"
                                + (" | input bn applied on load" if in_link and in_link.get("fold") is not None else "")
                                + (" | bn sums from the epilogue" if sum_link and sum_link.get("want_stats") else ""))
                z = ops.conv2d(h, conv.weight, conv.bias, conv.padding[0], conv.dilation[0], pooled=pooled,
                               bn_link=in_link, out_link=sum_link)
            if pooled:
                if not fused_pool:
                    plan_log.append(f"block{step + 1}: prelu+pool pass")
                    h = ops.prelu_maxpool2x2(z, slope, pool_link)
                if fold_next:
                    pending_bn = cnn[bn_i]
                elif bn_i is not None:
                    bn_link = {}
                    h = ops.batch_norm(h, cnn[bn_i], None, self.sync_bn, bn_link, pool_link,
                                       defer=self._next_normalises(step, tuple(h.shape)))
            else:
                bn_link = {}
                h = ops.batch_norm(z, cnn[bn_i], slope, self.sync_bn, bn_link, sum_link=sum_link,
                                   defer=self._next_normalises(step, tuple(z.shape)))
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
