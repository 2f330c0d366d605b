"""torch.autograd bindings of the libafd_hip layer kernels (host plumbing only).

Every function here marshals ``data_ptr()`` + the current HIP stream into one or two C-ABI
calls (``include/afd_hip.h``); no arithmetic on activations happens in PyTorch.  The small
per-channel vectors of BatchNorm (mean / invstd / running statistics, C <= 128 values) and
the cross-rank reductions go through torch / torch.distributed.
"""

from __future__ import annotations

import itertools
import os
from typing import Optional

import torch
import torch.distributed as dist

from . import _native

_ws_cache: dict = {}
_seed_counter = itertools.count(1)

# Tests only: when set to a list, every PReLU / max-pool forward appends (kind, tensor) --
# ("pool", the uint8 argmax code tensor) or ("prelu", the pre-activation tensor) -- i.e. the
# tensors whose signs / codes decide how the backward pass routes gradients.  They are the very
# tensors saved for backward, so a test can compare them with another implementation's decisions.
debug_taps: Optional[list] = None


def _tap(kind: str, t: torch.Tensor) -> None:
    if debug_taps is not None:
        debug_taps.append((kind, t))


def _lib():
    return _native.load()


def plan_switches() -> tuple:
    """The development switches (AFD_* environment variables) a fusion plan was decided under: part of its cache key
    (the applicability predicates here and in the library read them when they are asked)."""
    import os
    return tuple(sorted((k, v) for k, v in os.environ.items() if k.startswith("AFD_")))


_zero_chunks: dict = {}


def _zeros(n: int, dtype, device) -> torch.Tensor:
    """`n` zeros for an accumulator a kernel adds into (slope gradients, per-channel sums): a never-used slice of a chunk
    that ONE memset zeroed, instead of a `torch.zeros` -- a memset launch -- per accumulator (a level-8 step had seven of
    them).  A slice is handed out once and never again, so it is zero whatever happened to its neighbours; chunks are
    replaced when used up and stay alive as long as something still refers to a slice."""
    device = torch.device(device)
    key = (device.type, device.index, dtype)
    chunk = _zero_chunks.get(key)
    if chunk is None or chunk[1] + n > chunk[0].numel():
        chunk = [torch.zeros(max(4096, 4 * n), dtype=dtype, device=device), 0]
        _zero_chunks[key] = chunk
    out = chunk[0][chunk[1]:chunk[1] + n]
    chunk[1] += (n + 3) // 4 * 4  # (16-byte steps for the float slots, 32-byte for doubles)
    return out


def _ws(nbytes: int, device, lane: str = "main") -> torch.Tensor:
    """Grow-only scratch buffer per device and stream lane (conv weight slabs / wgrad partial slabs)."""
    key = (device.type, device.index, lane)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


# --------------------------------------------------------------------------------------
# Every kernel of a step runs on the caller's stream.  Rounds 1-2 issued the convolutions' backward-weight kernels
# on a second HIP stream; measured at the end of round 2 it bought 0.7 ms of a 63.6 ms step while its MFMA-bound
# and HBM-bound partners queued on CU residency (launch durations stretched 4-10x), and streams confined to disjoint
# CU partitions (hipExtStreamCreateWithCUMask) were 1.4-3x slower -- profiles/r03_stream_modes.json.  Removed.

# One-shot hook run by the autograd engine at the very end of the next backward pass: the data-parallel step starts its gradient all-reduce from here, without
# a trip back through the Python step code (train_classifier.sync_gradients then only waits for it).
_end_of_backward: list = []


def at_end_of_backward(fn) -> None:
    _end_of_backward.append(fn)


def _run_end_of_backward() -> None:
    while _end_of_backward:
        _end_of_backward.pop(0)()


def next_seed() -> int:
    """Seed of the next dropout mask: a function of torch's seed and a call counter."""
    return (torch.initial_seed() * 0x9E3779B97F4A7C15 + next(_seed_counter) * 0xD1B54A32D192ED03) \
        & 0x7FFFFFFFFFFFFFFF


def _f32c(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


class ScalarMoments:
    """Scalar mean / std of everything it is shown, accumulated on the GPU in double precision.

    One fused reduction per batch (``afd_moments_accumulate``) in place of the reference's
    ``WelfordEstimator.update`` chain of elementwise passes (data_loader.py:27-71); ``finalize``
    returns the same population statistics, ``sqrt(M2 / count)``.
    """

    def __init__(self, device) -> None:
        self.acc = torch.zeros(3, dtype=torch.float64, device=device)

    def update(self, t: torch.Tensor) -> None:
        _native.require_gpu()
        src = t if t.is_contiguous() else t.contiguous()
        src = src.to(torch.float32)
        _native.check(_lib().afd_moments_accumulate(_native.ptr(src), src.numel(), _native.ptr(self.acc),
                                                    _native.stream_ptr()), "afd_moments_accumulate")

    def finalize(self):
        n, s, q = (float(v) for v in self.acc.cpu())
        mean = s / n
        var = max(q / n - mean * mean, 0.0)
        return (torch.tensor([mean], dtype=torch.float32), torch.tensor([var ** 0.5], dtype=torch.float32))


def normalize_forward(t: torch.Tensor, mean: float, std: float) -> torch.Tensor:
    """(t - mean) / std, out of place, in the memory order of `t`."""
    _native.require_gpu()
    src = t
    perm = None
    if not t.is_contiguous():
        # features come as a permuted view of a dense buffer: normalise the buffer
        if t.dim() == 4 and t.permute(0, 1, 3, 2).is_contiguous():
            src = t.permute(0, 1, 3, 2)
            perm = (0, 1, 3, 2)
        else:
            src = t.contiguous()
    src = src if src.is_cuda else src.cuda()
    out = torch.empty_like(src)
    _native.check(_lib().afd_normalize_forward(_native.ptr(src), _native.ptr(out), src.numel(),
                                               float(mean), float(std), _native.stream_ptr()),
                  "afd_normalize_forward")
    return out.permute(*perm) if perm else out


def normalize_channels_forward(t: torch.Tensor, means, stds) -> torch.Tensor:
    """(t[:, c] - means[c]) / stds[c] for [B, C, ...] features, out of place, memory order kept:
    one `afd_normalize_channels_forward` launch."""
    _native.require_gpu()
    src, perm = t, None
    if not t.is_contiguous():
        if t.dim() == 4 and t.permute(0, 1, 3, 2).is_contiguous():
            src, perm = t.permute(0, 1, 3, 2), (0, 1, 3, 2)
        else:
            src = t.contiguous()
    src = src if src.is_cuda else src.cuda()
    out = torch.empty_like(src)
    b, c = src.shape[0], src.shape[1]
    plane = src.numel() // (b * c)
    if c <= 8 and b * c <= 65535:
        import ctypes

        ms = (ctypes.c_float * c)(*[float(means[ch]) for ch in range(c)])
        sd = (ctypes.c_float * c)(*[float(stds[ch]) for ch in range(c)])
        _native.check(_lib().afd_normalize_channels_forward(
            _native.ptr(src), _native.ptr(out), b, c, plane, ms, sd, _native.stream_ptr()), "afd_normalize_channels_forward")
    else:
        for i in range(b):
            for ch in range(c):
                off = (i * c + ch) * plane * 4
                _native.check(_lib().afd_normalize_forward(
                    _native.c_p(src.data_ptr() + off), _native.c_p(out.data_ptr() + off), plane,
                    float(means[ch]), float(stds[ch]), _native.stream_ptr()), "afd_normalize_forward")
    return out.permute(*perm) if perm else out


def transpose_last2(x: torch.Tensor) -> torch.Tensor:
    """Contiguous [.., C, R] from contiguous [.., R, C]."""
    r, c = x.shape[-2], x.shape[-1]
    planes = x.numel() // (r * c)
    out = torch.empty(x.shape[:-2] + (c, r), dtype=x.dtype, device=x.device)
    _native.check(_lib().afd_transpose_last2(_native.ptr(x), _native.ptr(out), planes, r, c,
                                             _native.stream_ptr()), "afd_transpose_last2")
    return out


class _Transpose(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return transpose_last2(_f32c(x))

    @staticmethod
    def backward(ctx, dy):
        return transpose_last2(_f32c(dy))


def transpose_contiguous(x: torch.Tensor) -> torch.Tensor:
    return _Transpose.apply(x)


def _conv3x3_forward_stats(lib, x, w, b, slope, y, u, idx, geom, ws, link, fold=None):
    """Runs the forward 3x3 convolution through `afd_conv3x3_forward_stats` and leaves the BatchNorm batch sums of
    its result (packed [sum | sum of squares | count slot], as afd_bn_stats) in link["fwd_sums"].  `fold`: see
    `_take_fold`."""
    n, cin, h, wd, cout = geom
    sums = torch.empty(2 * cout + 1, dtype=torch.float64, device=x.device)
    sws = _ws(lib.afd_conv3x3_forward_stats_workspace_bytes(n, h, wd, cout), x.device, "fwdstats")
    if fold is not None:
        _native.check(lib.afd_conv3x3_forward_fold(
            _native.ptr(x), _native.ptr(fold[0]), _native.ptr(fold[1]), _native.ptr(w), _native.ptr(b), _native.ptr(slope),
            _native.ptr(y), _native.ptr(u), _native.ptr(idx), _native.ptr(sums), n, cin, h, wd, cout, _native.ptr(ws),
            ws.numel(), _native.ptr(sws), sws.numel(), _native.stream_ptr()), "afd_conv3x3_forward_fold")
    else:
        _native.check(lib.afd_conv3x3_forward_stats(
            _native.ptr(x), _native.ptr(w), _native.ptr(b), _native.ptr(slope), _native.ptr(y), _native.ptr(u),
            _native.ptr(idx), _native.ptr(sums), n, cin, h, wd, cout, _native.ptr(ws), ws.numel(), _native.ptr(sws),
            sws.numel(), _native.stream_ptr()), "afd_conv3x3_forward_stats")
    link["fwd_sums"] = sums


def _take_fold(bn_link, lib, geom, pooled: bool, want_stats: bool):
    """The (aff, slope) pair a deferred BatchNorm left for this convolution (`batch_norm(..., defer=True)`): x is then
    that BatchNorm's INPUT and the launches normalise it while they load.  Fails loudly when the layer is not on the
    kernels that can (the caller asked `conv3x3_input_fold_applicable` before deferring)."""
    fold = bn_link.pop("fold", None) if bn_link is not None else None
    if fold is None:
        return None
    n, cin, h, wd, cout, k, pad, dil = geom
    if not (k == 3 and pad == 1 and dil == 1
            and lib.afd_conv3x3_input_fold_applicable(cin, h, wd, cout, int(pooled), int(want_stats))):
        raise RuntimeError("a BatchNorm deferred its normalisation to a convolution that cannot apply it: "
                           f"geometry {geom}, pooled={pooled}, statistics={want_stats}")
    return fold


def conv3x3_input_fold_applicable(bn: torch.nn.Module, conv: torch.nn.Module, shape, pooled: bool,
                                  want_stats: bool) -> bool:
    """May a training-mode BatchNorm(affine=False) on a tensor of `shape` leave its normalisation to `conv`, its only
    consumer (`batch_norm(..., defer=True)`)?"""
    import os
    if os.environ.get("AFD_NO_INPUT_FOLD") or not torch.is_grad_enabled():
        return False
    if not (bn.training and bn.weight is None and bn.bias is None and bn.running_mean is not None):
        return False
    if (conv.kernel_size != (3, 3) or conv.padding != (1, 1) or conv.dilation != (1, 1) or conv.stride != (1, 1)
            or conv.bias is None or not conv.weight.requires_grad):
        return False
    n, cin, h, w = shape
    return bool(_lib().afd_conv3x3_input_fold_applicable(cin, h, w, conv.out_channels, int(pooled), int(want_stats)))


def _pack_fold(mean, invstd):
    """[C][2] = (mean, invstd): the table the folded launches read."""
    return torch.stack((mean, invstd), dim=1).contiguous()


# --------------------------------------------------------------------------------------
class _Conv2d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, pad, dil, pooled, bn_link=None, out_link=None):
        ctx.bn_link = bn_link
        ctx.out_link = out_link
        lib = _lib()
        x = _f32c(x)
        w = _f32c(w)
        n, cin, h, wd = x.shape
        cout, _, k, _ = w.shape
        ho = h + 2 * pad - dil * (k - 1)
        wo = wd + 2 * pad - dil * (k - 1)
        # a 2x2/2 max-pool follows: the odd last row / column of y is never read and its
        # gradient is zero (nn.MaxPool2d floor mode)
        crop = (2 * (ho // 2), 2 * (wo // 2)) if pooled else (ho, wo)
        if ho < 1 or wo < 1:
            raise ValueError(f"conv2d: empty output for input {tuple(x.shape)}, k={k}, pad={pad}, dil={dil}")
        y = torch.empty((n, cout, ho, wo), dtype=torch.float32, device=x.device)
        nbytes = lib.afd_conv2d_workspace_bytes(n, cin, h, wd, cout, k, pad, dil)
        ws = _ws(nbytes, x.device)
        want_stats = bool(out_link is not None and out_link.get("want_stats") and k == 3 and pad == 1 and dil == 1
                          and not pooled and x.is_cuda and lib.afd_conv3x3_forward_stats_applicable(cin, h, wd, cout, 0))
        fold = _take_fold(bn_link, lib, (n, cin, h, wd, cout, k, pad, dil), False, True)
        if fold is not None and not want_stats:
            raise RuntimeError("a BatchNorm deferred its normalisation to a convolution launch that cannot apply it")
        ctx.fold = fold
        if want_stats:
            # the only consumer is a training-mode BatchNorm of PReLU(y): its batch sums from this launch's epilogue
            _conv3x3_forward_stats(lib, x, w, b, out_link.get("stats_slope"), y, None, None, (n, cin, h, wd, cout), ws,
                                   out_link, fold)
        else:
            _native.check(lib.afd_conv2d_forward_cropped(
                _native.ptr(x), _native.ptr(w), _native.ptr(b), _native.ptr(y), n, cin, h, wd, cout, k,
                pad, dil, crop[0], crop[1], _native.ptr(ws), ws.numel(), _native.stream_ptr()),
                "afd_conv2d_forward")
        ctx.save_for_backward(x, w)
        ctx.geom = (n, cin, h, wd, cout, k, pad, dil)
        ctx.crop = crop
        ctx.has_bias = b is not None
        ctx.bias_ref = b  # the Parameter itself: its .grad may be an optimizer arena view
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        # per-channel sums of dy left by its producer (a BatchNorm backward given the same dict as `sum_link`)
        dy_sums = ctx.out_link.pop("dy_sums", None) if ctx.out_link is not None else None
        dx, dw, db = _conv2d_backward(x, w, ctx.bias_ref, ctx.has_bias, ctx.geom, ctx.crop, dy,
                                      ctx.needs_input_grad[0], ctx.needs_input_grad[1],
                                      ctx.has_bias and ctx.needs_input_grad[2], ctx.bn_link, dy_sums, ctx.fold)
        return dx, dw, db, None, None, None, None, None


def _bn_backward_in_dgrad_ok(lib, bn_link, fold, geom, pooled: bool) -> bool:
    if bn_link is None or fold is None or "bn_ctx" not in bn_link:
        return False
    n, cin, h, wd, cout, k, pad, dil = geom
    return bool(k == 3 and pad == 1 and dil == 1
                and lib.afd_conv3x3_backward_data_bnapply_applicable(cin, h, wd, cout, int(pooled)))


def _dgrad_with_bn_backward(lib, z, w, dw, dy, codes, dy_sums, db, geom, crop, fold, bn_ctx, ws, pool=None):
    """Backward-data of a 3x3 convolution whose input was a deferred BatchNorm (`batch_norm(defer=True)`), with that
    BatchNorm's -- and the PReLU's in front of it -- backward applied in the launch's epilogue
    (`afd_conv3x3_backward_data_bnapply`): returns (dL/dz, its per-channel sums (double), dslope or None).  The two batch
    sums the BatchNorm backward needs come from small tensors: sum(g * xhat) = w . dw (so the weight gradient must exist),
    sum(g) from the weights and the border sums of dy (`afd_conv3x3_input_grad_sums`)."""
    n, cin, h, wd, cout = geom[:5]
    aff, slope = fold
    bn_codes = None
    if pool is not None:  # the BatchNorm sits behind PReLU + max-pool: z is the pooled tensor, the slope the pool's
        bn_codes, slope = pool
    mean, invstd, count, sync = bn_ctx
    dev = z.device
    buf = torch.empty(max(2 * cin, cin + 8 * cout), dtype=torch.float64, device=dev)
    _native.check(lib.afd_conv3x3_input_grad_sums(
        _native.ptr(dy), _native.ptr(codes), _native.ptr(w), _native.ptr(dy_sums), None if dy_sums is not None else _native.ptr(db),
        _native.ptr(buf), n, cin, h, wd, cout, crop[0], crop[1], _native.stream_ptr()), "afd_conv3x3_input_grad_sums")
    _native.check(lib.afd_conv_weight_dot(_native.ptr(w), _native.ptr(dw), cout, cin, 9, buf.data_ptr() + 8 * cin,
                                          _native.stream_ptr()), "afd_conv_weight_dot")
    sums = buf[:2 * cin]
    if _dist_on(sync):
        sums = sums.clone()
        all_reduce_sum(sums)
    mdy = torch.empty(cin, dtype=torch.float32, device=dev)
    mdyx = torch.empty(cin, dtype=torch.float32, device=dev)
    on_dev = torch.is_tensor(count)
    tab = torch.empty((cin, 4), dtype=torch.float32, device=dev)  # (mean, invstd, mdy, mdyx): written by the same launch
    _native.check(lib.afd_bn_backward_means(
        _native.ptr(sums), cin, -1.0 if on_dev else float(count), _native.ptr(count) if on_dev else None,
        _native.ptr(mdy), _native.ptr(mdyx), _native.ptr(mean), _native.ptr(invstd), _native.ptr(tab),
        _native.stream_ptr()), "afd_bn_backward_means")
    dz = _empty_with_slack(z.shape, torch.float32, dev)  # (a pooled gradient is read in whole vectors past its end)
    out = torch.empty(2 * cin, dtype=torch.float64, device=dev)
    sws = _ws(lib.afd_conv3x3_backward_data_bnstats_workspace_bytes(n, cin, h, wd), dev, "bnstats")
    _native.check(lib.afd_conv3x3_backward_data_bnapply(
        _native.ptr(dy), _native.ptr(codes), _native.ptr(w), _native.ptr(z), _native.ptr(tab), _native.ptr(slope),
        _native.ptr(bn_codes), _native.ptr(dz), _native.ptr(out), n, cin, h, wd, cout, _native.ptr(ws), ws.numel(),
        _native.ptr(sws), sws.numel(), _native.stream_ptr()), "afd_conv3x3_backward_data_bnapply")
    dslope = out[cin:].sum(dtype=torch.float32).reshape(1) if slope is not None else None
    return dz, out[:cin], dslope


def _conv2d_backward(x, w, b, has_bias, geom, crop, dy, need_dx, need_dw, need_db, bn_link=None, dy_sums=None,
                     fold=None):
    """Backward-data and backward-weight, both issued on the CURRENT stream in program order (the second stream of
    rounds 1-2 is gone, profiles/r03_stream_modes.json): the launches share the workspace `ws` and the statistics
    buffers, and the fold / bnapply branches rely on backward-weight having finished before backward-data reads its
    sums -- which same-stream ordering gives.  `bn_link`: when x was the
    output of a training-mode BatchNorm (which set "bn" there), the backward-data launch also produces that
    BatchNorm's backward sums -- of dx and of dx * x -- and leaves them in the dict."""
    lib = _lib()
    n, cin, h, wd, cout, k, pad, dil = geom
    dy = _f32c(dy)
    nbytes = lib.afd_conv2d_workspace_bytes(n, cin, h, wd, cout, k, pad, dil)
    ws = _ws(nbytes, x.device)
    dx = dw = db = None
    dot_sums = False
    bn = bn_link.get("bn") if bn_link is not None else None
    if (need_dx and need_dw and (dy_sums is not None or (has_bias and need_db))
            and _bn_backward_in_dgrad_ok(lib, bn_link, fold, geom, False)):
        # backward-weight first: its result gives the BatchNorm's second backward sum, and the backward-data launch then
        # applies the BatchNorm / PReLU backward to its own result (no pass over the activations for them)
        dw = torch.empty_like(w)
        db = torch.empty(cout, dtype=torch.float32, device=x.device) if has_bias else None
        _native.check(lib.afd_conv3x3_backward_weight_fold(
            _native.ptr(x), _native.ptr(fold[0]), _native.ptr(fold[1]), _native.ptr(dy), None, _native.ptr(dw),
            _native.ptr(db), _native.ptr(dy_sums), n, cin, h, wd, cout, crop[0], crop[1], _native.ptr(ws), ws.numel(),
            _native.stream_ptr()), "afd_conv3x3_backward_weight_fold")
        pool = bn_link.pop("bn_pool", None)
        dz, dxs, dslope = _dgrad_with_bn_backward(lib, x, w, dw, dy, None, dy_sums, db, geom, crop, fold,
                                                  bn_link.pop("bn_ctx"), ws, pool)
        bn_link["applied"] = (dslope, dxs, pool is not None)
        return dz, dw, db
    if (need_dx and bn is not None and k == 3 and pad == 1 and dil == 1
            and lib.afd_conv3x3_backward_data_bnstats_applicable(cin, h, wd, cout)):
        dx = torch.empty_like(x)
        sums = torch.empty(2 * cin, dtype=torch.float64, device=x.device)
        sws = _ws(lib.afd_conv3x3_backward_data_bnstats_workspace_bytes(n, cin, h, wd), x.device, "bnstats")
        # sum(dx * x) per channel equals sum over (cout, taps) of w * dw: with the weight gradient coming anyway, the
        # launch need not read x again
        dot_sums = need_dw and not lib.afd_conv3x3_backward_data_bnstats_needs_input(cin, h, wd, cout)
        if fold is not None and not dot_sums:
            raise RuntimeError("input fold: the backward-data launch of this layer needs the normalised tensor")
        _native.check(lib.afd_conv3x3_backward_data_bnstats(
            _native.ptr(dy), _native.ptr(w), _native.ptr(dx), None if dot_sums else _native.ptr(x), _native.ptr(sums),
            n, cin, h, wd, cout, _native.ptr(ws), ws.numel(), _native.ptr(sws), sws.numel(), _native.stream_ptr()),
            "afd_conv3x3_backward_data_bnstats")
        bn_link["bwd_sums"] = sums
    elif need_dx:
        if fold is not None:
            raise RuntimeError("input fold: the BatchNorm in front of this layer gets no backward sums")
        dx = torch.empty_like(x)
        _native.check(lib.afd_conv2d_backward_data(
            _native.ptr(dy), _native.ptr(w), _native.ptr(dx), n, cin, h, wd, cout, k, pad, dil,
            _native.ptr(ws), ws.numel(), _native.stream_ptr()), "afd_conv2d_backward_data")
    if need_dw or need_db:
        dw = torch.empty_like(w)
        db = torch.empty(cout, dtype=torch.float32, device=x.device) if has_bias else None
        if fold is not None:
            # x is the input of the BatchNorm in front of this layer: normalised while the launch loads it
            _native.check(lib.afd_conv3x3_backward_weight_fold(
                _native.ptr(x), _native.ptr(fold[0]), _native.ptr(fold[1]), _native.ptr(dy), None, _native.ptr(dw),
                _native.ptr(db), _native.ptr(dy_sums), n, cin, h, wd, cout, crop[0], crop[1], _native.ptr(ws), ws.numel(),
                _native.stream_ptr()), "afd_conv3x3_backward_weight_fold")
        else:
            _native.check(lib.afd_conv2d_backward_weight_sums(
                _native.ptr(x), _native.ptr(dy), _native.ptr(dw), _native.ptr(db), _native.ptr(dy_sums), n, cin, h,
                wd, cout, k, pad, dil, crop[0], crop[1], _native.ptr(ws), ws.numel(),
                _native.stream_ptr()), "afd_conv2d_backward_weight")
        if dot_sums:
            _native.check(lib.afd_conv_weight_dot(_native.ptr(w), _native.ptr(dw), cout, cin, k * k,
                                                  sums.data_ptr() + 8 * cin, _native.stream_ptr()), "afd_conv_weight_dot")
    return dx, dw, db


def conv2d(x, w, b=None, padding: int = 0, dilation: int = 1, pooled: bool = False,
           bn_link: Optional[dict] = None, out_link: Optional[dict] = None):
    """``pooled=True``: the result only feeds ``prelu_maxpool2x2`` (whose backward zeroes the
    gradient of an odd last row / column), so those need not be computed.  ``bn_link``: the dict given to the
    BatchNorm call that produced x (its ONLY consumer being this convolution), see `_conv2d_backward`.
    ``out_link``: the dict given as `sum_link` to the BatchNorm call that is the ONLY consumer of the result: its
    backward leaves the per-channel sums of the output gradient there (the bias gradient)."""
    return _Conv2d.apply(x, w, b, int(padding), int(dilation), bool(pooled), bn_link, out_link)


# --------------------------------------------------------------------------------------
def _pool_backward(u, slope, idx, du, zshape, out_link):
    """(dz, dslope) of PReLU + MaxPool2d(2, 2) from the pooled output and the argmax codes; when the BatchNorm
    behind the pool left its backward coefficients in `out_link`, du is that layer's incoming gradient and the
    BatchNorm backward is applied on load (`afd_prelu_pool_backward_affine`)."""
    lib = _lib()
    n, c, h, w = zshape
    du = _f32c(du)
    dz = torch.empty((n, c, h, w), dtype=torch.float32, device=u.device)
    dslope = _zeros(1, torch.float32, u.device) if slope is not None else None
    coef = out_link.pop("affine_coef", None) if out_link is not None else None
    _native.check(lib.afd_prelu_pool_backward_affine(
        _native.ptr(u), _native.ptr(slope), _native.ptr(idx), _native.ptr(du), _native.ptr(coef), c,
        _native.ptr(dz), _native.ptr(dslope), n * c, h, w, _native.stream_ptr()), "afd_prelu_pool_backward")
    return dz, dslope


class _PReLUPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, slope, out_link=None):
        ctx.out_link = out_link
        lib = _lib()
        z = _f32c(z)
        n, c, h, w = z.shape
        u = torch.empty((n, c, h // 2, w // 2), dtype=torch.float32, device=z.device)
        idx = torch.empty((n, c, h // 2, w // 2), dtype=torch.uint8, device=z.device)
        _native.check(lib.afd_prelu_pool_forward(_native.ptr(z), _native.ptr(slope), _native.ptr(u),
                                                 _native.ptr(idx), n * c, h, w,
                                                 _native.stream_ptr()), "afd_prelu_pool_forward")
        # backward needs only the pooled output and the 3-bit argmax code: the 4x larger
        # pre-pool tensor z is not kept alive
        _tap("pool", idx)
        ctx.save_for_backward(u, idx, slope if slope is not None else torch.empty(0))
        ctx.has_slope = slope is not None
        ctx.zshape = (n, c, h, w)
        return u

    @staticmethod
    def backward(ctx, du):
        u, idx, slope = ctx.saved_tensors
        dz, dslope = _pool_backward(u, slope if ctx.has_slope else None, idx, du, ctx.zshape, ctx.out_link)
        return dz, dslope, None


class _Conv1PReLUPool(torch.autograd.Function):
    """Conv2d(1 -> Cout, 3x3, pad) + PReLU + MaxPool2d(2,2), fused (single-channel first block)."""

    @staticmethod
    def forward(ctx, x, w, b, slope, pad, link=None):
        lib = _lib()
        x = _f32c(x)
        w = _f32c(w)
        n, _, h, wd = x.shape
        cout = w.shape[0]
        ctx.link = link
        hp, wp = (h + 2 * pad - 2) // 2, (wd + 2 * pad - 2) // 2
        u = torch.empty((n, cout, hp, wp), dtype=torch.float32, device=x.device)
        idx = torch.empty((n, cout, hp, wp), dtype=torch.uint8, device=x.device)
        # a training-mode BatchNorm behind the pool (the consumer set link["want_stats"]) takes its batch sums from this
        # launch: no statistics pass over u
        sums = sws = None
        if (link is not None and link.get("want_stats") and not os.environ.get("AFD_NO_FWD_STATS")
                and lib.afd_conv1_pool_stats_applicable(n, h, wd, cout, pad)):  # wide rows only (level 14)
            sums = torch.empty(2 * cout + 1, dtype=torch.float64, device=x.device)
            sws = _ws(lib.afd_conv1_pool_stats_workspace_bytes(n, h, wd, cout, pad), x.device, "stats")
        _native.check(lib.afd_conv1_pool_forward(
            _native.ptr(x), _native.ptr(w), _native.ptr(b), _native.ptr(slope), _native.ptr(u),
            _native.ptr(idx), _native.ptr(sums), _native.ptr(sws), sws.numel() if sws is not None else 0, n, h, wd, cout,
            pad, _native.stream_ptr()), "afd_conv1_pool_forward")
        if sums is not None:
            link["fwd_sums"] = sums
        _tap("pool", idx)
        ctx.save_for_backward(x, u, idx, slope)
        ctx.cfg = (n, h, wd, cout, pad, b is not None, tuple(w.shape))
        return u

    @staticmethod
    def backward(ctx, du):
        lib = _lib()
        x, u, idx, slope = ctx.saved_tensors
        n, h, wd, cout, pad, has_bias, wshape = ctx.cfg
        du = _f32c(du)
        dw = torch.empty(wshape, dtype=torch.float32, device=x.device)
        db = torch.empty(cout, dtype=torch.float32, device=x.device) if has_bias else None
        dslope = _zeros(1, torch.float32, x.device)
        ws = _ws(lib.afd_conv1_pool_workspace_bytes(n, h, wd, cout, pad), x.device)
        # the consumer of u (`_BNConv1x1PReLUBN`) may have left the affine part of its input gradient,
        # alpha[c] * u + beta[c], to this kernel, which reads du and u anyway
        aff = ctx.link.pop("affine", None) if ctx.link is not None else None
        alpha, beta = aff if aff is not None else (None, None)
        _native.check(lib.afd_conv1_pool_backward_affine(
            _native.ptr(x), _native.ptr(du), _native.ptr(idx), _native.ptr(u), _native.ptr(slope),
            _native.ptr(alpha), _native.ptr(beta), _native.ptr(dw), _native.ptr(db), _native.ptr(dslope),
            n, h, wd, cout, pad, _native.ptr(ws), ws.numel(), _native.stream_ptr()),
            "afd_conv1_pool_backward")
        return None, dw, db, dslope, None, None


def conv1_prelu_maxpool(x, w, b, slope, padding: int, link: Optional[dict] = None):
    """Fused first block for single-channel inputs (no gradient w.r.t. x: the features are
    produced under no_grad, reference train_classifier.py:965-967).  `link`: a dict shared with the
    consumer of the result (`bn_conv1x1_prelu_bn`), through which that layer's backward hands over the
    affine part of the gradient."""
    return _Conv1PReLUPool.apply(x, w, b, slope, int(padding), link)


def prelu_maxpool2x2(z, slope: Optional[torch.Tensor], out_link: Optional[dict] = None):
    """MaxPool2d(2,2)(PReLU(z)); slope=None -> plain max pool.  `out_link`: a dict shared with the BatchNorm that
    is the ONLY consumer of the result (`batch_norm(.., prod_link=)`): its backward is then applied inside this
    layer's backward."""
    return _PReLUPool.apply(z, slope, out_link)


def _empty_with_slack(shape, dtype, device, slack: int = 16):
    """A tensor of `shape` whose storage continues `slack` elements past its end: the kernels that expand a pooled
    gradient read whole 2- / 4-element vectors whose last elements may lie past the last row (and are masked)."""
    numel = 1
    for d in shape:
        numel *= int(d)
    return torch.empty(numel + slack, dtype=dtype, device=device)[:numel].view(shape)


class _Conv3x3PReLUPool(torch.autograd.Function):
    """Conv2d(k=3, padding=1) + PReLU + MaxPool2d(2, 2) in one launch (DCNN blocks 3 and 6): on the
    Winograd kernel the 2x2 output tile is the pooling window, so the convolution output is never
    written.  Backward: pool/PReLU backward from (u, code), then the convolution's backward."""

    @staticmethod
    def forward(ctx, x, w, b, slope, bn_link=None, out_link=None):
        ctx.bn_link = bn_link
        ctx.out_link = out_link
        lib = _lib()
        x = _f32c(x)
        w = _f32c(w)
        n, cin, h, wd = x.shape
        cout = w.shape[0]
        u = torch.empty((n, cout, h // 2, wd // 2), dtype=torch.float32, device=x.device)
        idx = _empty_with_slack((n, cout, h // 2, wd // 2), torch.uint8, x.device)
        nbytes = lib.afd_conv2d_workspace_bytes(n, cin, h, wd, cout, 3, 1, 1)
        ws = _ws(nbytes, x.device)
        want_stats = bool(out_link is not None and out_link.get("want_stats")
                          and lib.afd_conv3x3_forward_stats_applicable(cin, h, wd, cout, 1))
        fold = _take_fold(bn_link, lib, (n, cin, h, wd, cout, 3, 1, 1), True, want_stats)
        ctx.fold = fold
        if want_stats:
            # the only consumer is a training-mode BatchNorm of u: its batch sums from this launch's epilogue
            _conv3x3_forward_stats(lib, x, w, b, slope, None, u, idx, (n, cin, h, wd, cout), ws, out_link, fold)
        elif fold is not None:
            _native.check(lib.afd_conv3x3_forward_fold(
                _native.ptr(x), _native.ptr(fold[0]), _native.ptr(fold[1]), _native.ptr(w), _native.ptr(b),
                _native.ptr(slope), None, _native.ptr(u), _native.ptr(idx), None, n, cin, h, wd, cout, _native.ptr(ws),
                ws.numel(), None, 0, _native.stream_ptr()), "afd_conv3x3_forward_fold")
        else:
            _native.check(lib.afd_conv3x3_prelu_pool_forward(
                _native.ptr(x), _native.ptr(w), _native.ptr(b), _native.ptr(slope), _native.ptr(u),
                _native.ptr(idx), n, cin, h, wd, cout, _native.ptr(ws), ws.numel(), _native.stream_ptr()),
                "afd_conv3x3_prelu_pool_forward")
        _tap("pool", idx)
        import os
        if (out_link is not None and x.is_cuda and bn_link is not None and bn_link.get("bn")
                and not os.environ.get("AFD_NO_BWD_POOLAPPLY")
                and lib.afd_conv3x3_pooled_backward_applicable(cin, h, wd, cout)):
            # a BatchNorm right behind the pool may have its backward -- and this pool's / PReLU's -- applied by ITS
            # consumer's backward-data launch, which then hands back the pooled gradient (`_dgrad_with_bn_backward`)
            out_link["pool_ctx"] = (idx, slope)
        ctx.save_for_backward(x, w, u, idx, slope)
        ctx.geom = (n, cin, h, wd, cout, 3, 1, 1)
        ctx.crop = (2 * (h // 2), 2 * (wd // 2))
        ctx.has_bias = b is not None
        ctx.bias_ref = b
        return u

    @staticmethod
    def backward(ctx, du):
        x, w, u, idx, slope = ctx.saved_tensors
        n, cin, h, wd, cout = ctx.geom[:5]
        lib = _lib()
        bn = ctx.bn_link.get("bn") if ctx.bn_link is not None else None
        if (bn is not None and ctx.needs_input_grad[0] and ctx.needs_input_grad[1]
                and lib.afd_conv3x3_pooled_backward_applicable(cin, h, wd, cout)):
            # the pooled gradient + the argmax codes stand for the dense one (three quarters zeros, 4x the bytes,
            # written once and read twice): both convolution kernels expand them while they load
            du = _f32c(du)
            done = ctx.out_link.pop("compacted", None) if ctx.out_link is not None else None
            if done is not None:
                # du IS the pooled gradient of this convolution: the BatchNorm behind the pool and this PReLU / pool were
                # differentiated by the next convolution's backward-data launch
                gg, dslope = du, done
            else:
                gg = _empty_with_slack(u.shape, torch.float32, u.device)
                dslope = _zeros(1, torch.float32, u.device)
            coef = ctx.out_link.pop("affine_coef", None) if ctx.out_link is not None else None
            if done is None:
                _native.check(lib.afd_prelu_pool_backward_compact(
                    _native.ptr(u), _native.ptr(slope), _native.ptr(idx), _native.ptr(du), _native.ptr(coef), cout,
                    _native.ptr(gg), _native.ptr(dslope), n * cout, h // 2, wd // 2, _native.stream_ptr()),
                    "afd_prelu_pool_backward_compact")
            ws = _ws(lib.afd_conv2d_workspace_bytes(n, cin, h, wd, cout, 3, 1, 1), x.device)
            if ctx.has_bias and _bn_backward_in_dgrad_ok(lib, ctx.bn_link, ctx.fold, ctx.geom, True):
                # as in `_conv2d_backward`: backward-weight first, then backward-data with the BatchNorm / PReLU
                # backward of its result in the epilogue
                dw = torch.empty_like(w)
                db = torch.empty(cout, dtype=torch.float32, device=x.device)
                _native.check(lib.afd_conv3x3_backward_weight_fold(
                    _native.ptr(x), _native.ptr(ctx.fold[0]), _native.ptr(ctx.fold[1]), _native.ptr(gg), _native.ptr(idx),
                    _native.ptr(dw), _native.ptr(db), None, n, cin, h, wd, cout, h, wd, _native.ptr(ws), ws.numel(),
                    _native.stream_ptr()), "afd_conv3x3_backward_weight_fold")
                pool = ctx.bn_link.pop("bn_pool", None)
                dz, dxs, dsl_in = _dgrad_with_bn_backward(lib, x, w, dw, gg, idx, None, db, ctx.geom, ctx.crop, ctx.fold,
                                                         ctx.bn_link.pop("bn_ctx"), ws, pool)
                ctx.bn_link["applied"] = (dsl_in, dxs, pool is not None)
                return dz, dw, db, dslope, None, None
            dx = torch.empty_like(x)
            sums = torch.empty(2 * cin, dtype=torch.float64, device=x.device)
            sws = _ws(lib.afd_conv3x3_backward_data_bnstats_workspace_bytes(n, cin, h, wd), x.device, "bnstats")
            _native.check(lib.afd_conv3x3_backward_data_bnstats_pooled(
                _native.ptr(gg), _native.ptr(idx), _native.ptr(w), _native.ptr(dx), _native.ptr(sums), n, cin, h, wd,
                cout, _native.ptr(ws), ws.numel(), _native.ptr(sws), sws.numel(), _native.stream_ptr()),
                "afd_conv3x3_backward_data_bnstats_pooled")
            dw = torch.empty_like(w)
            db = torch.empty(cout, dtype=torch.float32, device=x.device) if ctx.has_bias else None
            if ctx.fold is not None:
                _native.check(lib.afd_conv3x3_backward_weight_fold(
                    _native.ptr(x), _native.ptr(ctx.fold[0]), _native.ptr(ctx.fold[1]), _native.ptr(gg), _native.ptr(idx),
                    _native.ptr(dw), _native.ptr(db), None, n, cin, h, wd, cout, h, wd, _native.ptr(ws), ws.numel(),
                    _native.stream_ptr()), "afd_conv3x3_backward_weight_fold")
            else:
                _native.check(lib.afd_conv3x3_backward_weight_pooled(
                    _native.ptr(x), _native.ptr(gg), _native.ptr(idx), _native.ptr(dw), _native.ptr(db), n, cin, h, wd,
                    cout, _native.ptr(ws), ws.numel(), _native.stream_ptr()), "afd_conv3x3_backward_weight_pooled")
            _native.check(lib.afd_conv_weight_dot(_native.ptr(w), _native.ptr(dw), cout, cin, 9,
                                                  sums.data_ptr() + 8 * cin, _native.stream_ptr()), "afd_conv_weight_dot")
            ctx.bn_link["bwd_sums"] = sums
            return dx, dw, db, dslope, None, None
        if ctx.out_link is not None and "compacted" in ctx.out_link:
            raise RuntimeError("conv3x3 + pool backward: a pooled gradient arrived on the dense-gradient path")
        dz, dslope = _pool_backward(u, slope, idx, du, (n, cout, h, wd), ctx.out_link)
        dx, dw, db = _conv2d_backward(x, w, ctx.bias_ref, ctx.has_bias, ctx.geom, ctx.crop, dz,
                                      ctx.needs_input_grad[0], ctx.needs_input_grad[1],
                                      ctx.has_bias and ctx.needs_input_grad[2], ctx.bn_link, None, ctx.fold)
        return dx, dw, db, dslope, None, None


def conv3x3_prelu_maxpool_applicable(x, conv: torch.nn.Module) -> bool:
    import os
    if os.environ.get("AFD_NO_CONV_POOL_FUSE") or not x.is_cuda:
        return False
    if (conv.kernel_size != (3, 3) or conv.padding != (1, 1) or conv.dilation != (1, 1)
            or conv.stride != (1, 1)):
        return False
    n, cin, h, w = x.shape
    return bool(_lib().afd_conv3x3_prelu_pool_applicable(cin, h, w, conv.out_channels))


def conv3x3_prelu_maxpool(x, w, b, slope, bn_link: Optional[dict] = None, out_link: Optional[dict] = None):
    """MaxPool2d(2, 2)(PReLU(conv2d(x, w, b, padding=1))) without materialising the convolution output.
    ``bn_link``: as for `conv2d`; ``out_link``: as for `prelu_maxpool2x2`."""
    return _Conv3x3PReLUPool.apply(x, w, b, slope, bn_link, out_link)


# --------------------------------------------------------------------------------------
def multi_rank() -> bool:
    """A process group with more than one rank is up -- or one rank with AFD_FORCE_COLLECTIVES=1, which makes a
    one-GPU run issue every collective of the data-parallel step (tools/ddp_collectives.py counts and times them:
    the only multi-GPU evidence a one-GPU box can give)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or bool(os.environ.get("AFD_FORCE_COLLECTIVES"))


def _dist_on(sync: bool) -> bool:
    return sync and multi_rank()


_direct_rccl = False
# collectives issued by the training step since the last `collective_counters(reset=True)`: the first run with more
# than one GPU records what it exchanged (bench.py prints them per step on its line)
_collectives = {"count": 0, "bytes": 0}


def note_collective(t: torch.Tensor) -> None:
    _collectives["count"] += 1
    _collectives["bytes"] += t.numel() * t.element_size()


def collective_counters(reset: bool = False) -> dict:
    out = dict(_collectives)
    if reset:
        _collectives["count"] = 0
        _collectives["bytes"] = 0
    return out


def enable_direct_rccl() -> bool:
    """OPT-IN (AFD_RCCL_DIRECT=1, called by `ddp_setup` / the tools): a second RCCL communicator inside libafd_hip,
    created from a unique id that rank 0 draws and the existing process group broadcasts; the in-step collectives
    (`all_reduce_sum`) then run as ncclAllReduce ON THE COMPUTE STREAM instead of going through c10d's stream hand-off
    (~55 us of GPU idle per collective, profiles/r04_ddp1_collectives.json).  Validated with one rank only -- no box
    with two GPUs has run it -- which is why it is not the default.  Returns whether the direct path is up."""
    global _direct_rccl
    if _direct_rccl:
        return True
    if not (dist.is_available() and dist.is_initialized() and dist.get_backend() == "nccl"):
        return False
    lib = _native.load()
    import ctypes
    buf = ctypes.create_string_buffer(128)
    dev = torch.device("cuda", torch.cuda.current_device())
    # byte 128 = rank 0's status: a failure to draw the id must not leave the other ranks waiting in the broadcast
    uid = torch.zeros(129, dtype=torch.uint8, device=dev)
    err = None
    if dist.get_rank() == 0:
        rc = lib.afd_rccl_unique_id(buf)
        if rc != 0:
            err = lib.afd_last_error().decode(errors="replace")
            uid[128] = 1
        else:
            uid[:128].copy_(torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8))
    dist.broadcast(uid, src=0)
    host = uid.cpu().tolist()
    if host[128]:
        raise RuntimeError(f"direct RCCL: rank 0 could not draw a unique id ({err or 'see rank 0'})")
    rc = lib.afd_rccl_init(ctypes.c_char_p(bytes(host[:128])), dist.get_rank(), dist.get_world_size())
    # every rank learns whether every rank's communicator came up, so that all raise (or all proceed) together
    ok = torch.tensor([1 if rc == 0 else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if int(ok.item()) == 0:
        if rc == 0:
            lib.afd_rccl_destroy()
        raise RuntimeError("direct RCCL: ncclCommInitRank failed on at least one rank"
                           + (f" (this rank: {lib.afd_last_error().decode(errors='replace')})" if rc != 0 else ""))
    _direct_rccl = True
    return True


def disable_direct_rccl() -> None:
    global _direct_rccl
    if _direct_rccl:
        _native.load().afd_rccl_destroy()
        _direct_rccl = False


def all_reduce_sum(t: torch.Tensor) -> None:
    """In-place sum of `t` over the ranks: ncclAllReduce on the current stream when the direct communicator is up
    (contiguous float32 / float64 GPU tensors only -- anything else raises), `dist.all_reduce` otherwise."""
    note_collective(t)
    if _direct_rccl:
        # exactly one communicator carries the in-step collectives: RCCL orders launches within a communicator, not
        # across two of them on one device, so a tensor the direct path cannot take is an error, not a detour via c10d
        if not (t.is_cuda and t.is_contiguous() and t.dtype in (torch.float32, torch.float64)):
            raise RuntimeError(f"direct RCCL: all_reduce_sum takes contiguous float32 / float64 GPU tensors, got "
                               f"{t.dtype} on {t.device} (contiguous: {t.is_contiguous()})")
        _native.check(_native.load().afd_rccl_all_reduce_sum(_native.ptr(t), t.numel(), 1 if t.dtype == torch.float64 else 0,
                                                             _native.stream_ptr()), "afd_rccl_all_reduce_sum")
    else:
        dist.all_reduce(t)


def bn_finalize(sums, c, local_count, eps, sync, running_mean=None, running_var=None, nbt=None,
                momentum=0.1):
    """Batch statistics from the packed per-channel [sum | sum of squares | count] vector.

    With a process group up (and `sync`) the packed vector is all-reduced first, so mean and
    variance are those of the GLOBAL batch with per-rank counts, as nn.SyncBatchNorm computes
    them (reference models.py:260-289).  Updates the running statistics in place (unbiased
    variance, reference momentum).  Device agnostic: also used by the CPU gloo tests.
    """
    if _dist_on(sync):
        sums[2 * c] = local_count
        all_reduce_sum(sums)
        cnt = sums[2 * c].clone()
    else:
        cnt = torch.tensor(float(local_count), dtype=torch.float64, device=sums.device)
    mean64 = sums[:c] / cnt
    var64 = (sums[c:2 * c] / cnt - mean64 * mean64).clamp_(min=0.0)
    mean = mean64.float()
    invstd = torch.rsqrt(var64 + eps).float()
    if running_mean is not None:
        with torch.no_grad():
            mom = momentum
            if nbt is not None:
                nbt += 1
                if mom is None:
                    mom = 1.0 / float(nbt)
            unbiased = var64 * (cnt / (cnt - 1.0).clamp_(min=1.0))
            running_mean.mul_(1.0 - mom).add_(mean, alpha=mom)
            running_var.mul_(1.0 - mom).add_(unbiased.float(), alpha=mom)
    return mean, invstd, cnt


class _BatchNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, slope, gamma, beta, running_mean, running_var, nbt, training, momentum,
                eps, sync, link=None, prod_link=None, sum_link=None, pre_sums=None, defer=False):
        ctx.link = link
        ctx.prod_link = prod_link
        ctx.sum_link = sum_link
        lib = _lib()
        x = _f32c(x)
        n, c = x.shape[0], x.shape[1]
        hw = x.numel() // (n * c)
        dev = x.device
        count = float(n * hw)
        fold_tab = None
        if training:
            if pre_sums is not None:
                sums = pre_sums  # from the producer's epilogue (afd_conv3x3_forward_stats)
            else:
                sums = torch.empty(2 * c + 1, dtype=torch.float64, device=dev)
                _native.check(lib.afd_bn_stats(_native.ptr(x), _native.ptr(slope), _native.ptr(sums), n, c,
                                               hw, _native.stream_ptr()), "afd_bn_stats")
            if x.is_cuda:
                # one kernel for mean / invstd / running statistics / num_batches_tracked
                dist_on = _dist_on(sync)
                if dist_on:
                    sums[2 * c] = count
                    all_reduce_sum(sums)
                mean = torch.empty(c, dtype=torch.float32, device=dev)
                invstd = torch.empty(c, dtype=torch.float32, device=dev)
                cnt = torch.empty(1, dtype=torch.float64, device=dev) if dist_on else None
                # (a deferred BatchNorm's (mean, invstd) pairs come out of the same launch: no torch.stack)
                fold_tab = (torch.empty((c, 2), dtype=torch.float32, device=dev)
                            if defer and gamma is None and link is not None else None)
                mom = momentum
                if mom is None:  # cumulative moving average (nn.BatchNorm2d(momentum=None))
                    mom = 1.0 / float(int(nbt) + 1) if nbt is not None else 0.0
                with torch.no_grad():
                    _native.check(lib.afd_bn_finalize(
                        _native.ptr(sums), c, -1.0 if dist_on else count, float(eps), float(mom),
                        _native.ptr(mean), _native.ptr(invstd), _native.ptr(running_mean),
                        _native.ptr(running_var), _native.ptr(nbt), _native.ptr(cnt), _native.ptr(fold_tab),
                        _native.stream_ptr()), "afd_bn_finalize")
                ctx.count = cnt if dist_on else count
            else:
                mean, invstd, cnt = bn_finalize(sums, c, count, eps, sync, running_mean, running_var,
                                                nbt, momentum)
                ctx.count = cnt
        else:
            mean = running_mean
            invstd = torch.rsqrt(running_var + eps)
        if defer and training and gamma is None and link is not None and x.is_cuda:
            # the convolution that consumes the result normalises x while it loads (`_take_fold`): no pass here, the
            # result is never stored.  The backward below is unchanged: it takes the gradient of the normalised
            # tensor from that convolution's backward-data launch, and never read the normalised tensor itself.
            link["fold"] = (fold_tab if fold_tab is not None else _pack_fold(mean, invstd), slope)
            link["bn"] = True
            if prod_link is None:
                # the consumer's backward-data launch may apply this layer's (and the PReLU's) backward itself
                # (`_dgrad_with_bn_backward`)
                link["bn_ctx"] = (mean, invstd, ctx.count, sync)
            elif slope is None and "pool_ctx" in prod_link:
                # ... and behind a fused convolution + PReLU + pool, that pool's and PReLU's backward with it (the pool's
                # codes decide the slope factor); otherwise the BatchNorm hands its backward to the pool (below)
                link["bn_ctx"] = (mean, invstd, ctx.count, sync)
                link["bn_pool"] = prod_link["pool_ctx"]
            empty = torch.empty(0)
            if slope is not None:
                _tap("prelu", x)
            ctx.save_for_backward(x, slope if slope is not None else empty, mean, invstd, empty)
            ctx.flags = (slope is not None, False, training, sync)
            return x
        y = torch.empty_like(x)
        _native.check(lib.afd_bn_apply_forward(
            _native.ptr(x), _native.ptr(slope), _native.ptr(mean), _native.ptr(invstd),
            _native.ptr(gamma), _native.ptr(beta), _native.ptr(y), n, c, hw, _native.stream_ptr()),
            "afd_bn_apply_forward")
        empty = torch.empty(0)
        if slope is not None:
            _tap("prelu", x)
        ctx.save_for_backward(x, slope if slope is not None else empty, mean, invstd,
                              gamma if gamma is not None else empty)
        ctx.flags = (slope is not None, gamma is not None, training, sync)
        if link is not None and training and gamma is None:
            # the consumer's backward-data launch can produce this layer's backward sums (`_conv2d_backward`)
            link["bn"] = True
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib()
        x, slope, mean, invstd, gamma = ctx.saved_tensors
        has_slope, has_gamma, training, sync = ctx.flags
        if not training:
            # running statistics are constants: a per-channel scale (attribution passes such as
            # integrated gradients differentiate the evaluation-mode model with respect to its
            # input).  Off the training path: elementwise torch ops on the device.
            slope = slope if has_slope else None
            gamma = gamma if has_gamma else None
            shp = (1, -1) + (1,) * (x.dim() - 2)
            dy = _f32c(dy)
            v = torch.where(x > 0, x, x * slope) if has_slope else x
            scale = invstd * gamma if has_gamma else invstd
            g = dy * scale.reshape(shp)
            dgamma = dbeta = dslope = None
            if has_gamma:
                red = [d for d in range(x.dim()) if d != 1]
                dgamma = (dy * (v - mean.reshape(shp)) * invstd.reshape(shp)).sum(red)
                dbeta = dy.sum(red)
            if has_slope:
                neg = x <= 0
                dslope = (g * x * neg).sum().reshape(1)
                g = torch.where(neg, g * slope, g)
            return g, dslope, dgamma, dbeta, None, None, None, None, None, None, None, None, None, None, None, None
        applied = ctx.link.pop("applied", None) if ctx.link is not None else None
        if applied is not None:
            # the consumer's backward-data launch has applied this layer's and the PReLU's backward: dy IS dL/dx
            dslope, dxs, pooled = applied
            if pooled:
                ctx.prod_link["compacted"] = dslope  # (the slope is the pool's: see `_Conv3x3PReLUPool.backward`)
                dslope = None
            elif ctx.sum_link is not None:
                ctx.sum_link["dy_sums"] = dxs
            return dy, dslope, None, None, None, None, None, None, None, None, None, None, None, None, None, None
        slope = slope if has_slope else None
        gamma = gamma if has_gamma else None
        n, c = x.shape[0], x.shape[1]
        hw = x.numel() // (n * c)
        dy = _f32c(dy)
        sums = ctx.link.pop("bwd_sums", None) if ctx.link is not None else None
        if sums is None:
            sums = torch.empty(2 * c, dtype=torch.float64, device=x.device)
            _native.check(lib.afd_bn_backward_stats(
                _native.ptr(x), _native.ptr(slope), _native.ptr(dy), _native.ptr(mean),
                _native.ptr(invstd), _native.ptr(sums), n, c, hw, _native.stream_ptr()),
                "afd_bn_backward_stats")
        dgamma = dbeta = None
        if has_gamma:
            both = sums.float()  # (one cast launch for both halves)
            dbeta, dgamma = both[:c], both[c:]
        if _dist_on(sync):
            sums = sums.clone()
            all_reduce_sum(sums)
        mdy = torch.empty(c, dtype=torch.float32, device=x.device)
        mdyx = torch.empty(c, dtype=torch.float32, device=x.device)
        on_dev = torch.is_tensor(ctx.count)
        _native.check(lib.afd_bn_backward_means(
            _native.ptr(sums), c, -1.0 if on_dev else float(ctx.count),
            _native.ptr(ctx.count) if on_dev else None, _native.ptr(mdy), _native.ptr(mdyx), None, None, None,
            _native.stream_ptr()), "afd_bn_backward_means")
        if ctx.prod_link is not None and not has_slope and not has_gamma:
            # the producer of x (a PReLU + max-pool backward) applies dx = A dy + B x + K where it reads dy and x
            coef = torch.empty((c, 4), dtype=torch.float32, device=x.device)
            _native.check(lib.afd_bn_backward_coef(_native.ptr(mean), _native.ptr(invstd), _native.ptr(mdy),
                                                   _native.ptr(mdyx), _native.ptr(coef), c, _native.stream_ptr()),
                          "afd_bn_backward_coef")
            ctx.prod_link["affine_coef"] = coef
            return dy, None, None, None, None, None, None, None, None, None, None, None, None, None, None, None
        dx = torch.empty_like(x)
        dslope = _zeros(1, torch.float32, x.device) if has_slope else None
        # the per-channel sums of dx for the convolution that produced x (its bias gradient), from the same pass
        dxs = _zeros(c, torch.float64, x.device) if ctx.sum_link is not None else None
        _native.check(lib.afd_bn_backward_apply_sums(
            _native.ptr(x), _native.ptr(slope), _native.ptr(dy), _native.ptr(mean),
            _native.ptr(invstd), _native.ptr(gamma), _native.ptr(mdy), _native.ptr(mdyx),
            _native.ptr(dx), _native.ptr(dslope), _native.ptr(dxs), n, c, hw, _native.stream_ptr()),
            "afd_bn_backward_apply")
        if dxs is not None:
            ctx.sum_link["dy_sums"] = dxs
        return dx, dslope, dgamma, dbeta, None, None, None, None, None, None, None, None, None, None, None, None


def _fold_forward(w2, b, mean, invstd):
    """(wf, bf): weights / bias of the 1x1 convolution with the BatchNorm in front of it folded in."""
    cout, c = w2.shape
    wf = torch.empty_like(w2)
    bf = torch.empty(cout, dtype=torch.float32, device=w2.device)
    _native.check(_lib().afd_bn_fold_forward(_native.ptr(w2), _native.ptr(b), _native.ptr(mean), _native.ptr(invstd),
                                             _native.ptr(wf), _native.ptr(bf), cout, c, _native.stream_ptr()),
                  "afd_bn_fold_forward")
    return wf, bf


def _fold_backward(gw, db, w2, mean, invstd, count, sync, need_affine):
    """dw against the normalised input from the gradient gw against the un-normalised one, and (alpha, beta) of
    the input gradient du = wf^T dz + alpha u + beta (the BatchNorm backward rebuilt from the small matrices)."""
    lib = _lib()
    cout, c = w2.shape
    dw = torch.empty_like(w2)
    sums = torch.empty(2 * c, dtype=torch.float64, device=w2.device)
    _native.check(lib.afd_bn_fold_backward_weights(
        _native.ptr(gw), _native.ptr(db), _native.ptr(w2), _native.ptr(mean), _native.ptr(invstd), _native.ptr(dw),
        _native.ptr(sums), cout, c, _native.stream_ptr()), "afd_bn_fold_backward_weights")
    if not need_affine:
        return dw, None, None
    if _dist_on(sync):
        all_reduce_sum(sums)
    alpha = torch.empty(c, dtype=torch.float32, device=w2.device)
    beta = torch.empty(c, dtype=torch.float32, device=w2.device)
    on_dev = torch.is_tensor(count)
    _native.check(lib.afd_bn_fold_backward_affine(
        _native.ptr(sums), -1.0 if on_dev else float(count), _native.ptr(count) if on_dev else None,
        _native.ptr(mean), _native.ptr(invstd), _native.ptr(alpha), _native.ptr(beta), c, _native.stream_ptr()),
        "afd_bn_fold_backward_affine")
    return dw, alpha, beta


class _BNConv1x1(torch.autograd.Function):
    """BatchNorm2d(affine=False) followed by Conv2d(k=1, pad 0) as ONE pass over the activations
    in each direction (DCNN blocks 1 -> 2, reference models.py:260-262).

    The normalisation is a per-channel scale and shift of the convolution's input, so it folds into
    the weights: z = (w * invstd) . u + (b - (w * invstd) . mean), with u the un-normalised tensor
    (the normalised one is never written).  Backward: the weight / bias gradients against u give
    those against xhat by the same algebra, the two batch means of the BatchNorm backward are
    sum_co w dbias and sum_co w dw, and the input gradient is the backward-data GEMM with the folded
    weights plus alpha * u + beta (`afd_conv1x1_bn_backward_data`)."""

    @staticmethod
    def forward(ctx, u, w, b, running_mean, running_var, nbt, training, momentum, eps, sync):
        lib = _lib()
        u = _f32c(u)
        n, c, h, wd = u.shape
        cout = w.shape[0]
        hw = h * wd
        dev = u.device
        count = float(n * hw)
        if training:
            sums = torch.empty(2 * c + 1, dtype=torch.float64, device=dev)
            _native.check(lib.afd_bn_stats(_native.ptr(u), None, _native.ptr(sums), n, c, hw,
                                           _native.stream_ptr()), "afd_bn_stats")
            dist_on = _dist_on(sync)
            if dist_on:
                sums[2 * c] = count
                all_reduce_sum(sums)
            mean = torch.empty(c, dtype=torch.float32, device=dev)
            invstd = torch.empty(c, dtype=torch.float32, device=dev)
            cnt = torch.empty(1, dtype=torch.float64, device=dev) if dist_on else None
            mom = momentum
            if mom is None:
                mom = 1.0 / float(int(nbt) + 1) if nbt is not None else 0.0
            with torch.no_grad():
                _native.check(lib.afd_bn_finalize(
                    _native.ptr(sums), c, -1.0 if dist_on else count, float(eps), float(mom),
                    _native.ptr(mean), _native.ptr(invstd), _native.ptr(running_mean),
                    _native.ptr(running_var), _native.ptr(nbt), _native.ptr(cnt), None,
                    _native.stream_ptr()), "afd_bn_finalize")
            ctx.count = cnt if dist_on else count
        else:
            mean = running_mean
            invstd = torch.rsqrt(running_var + eps)
        w2 = _f32c(w).reshape(cout, c)
        wf, bf = _fold_forward(w2, b, mean.contiguous(), invstd.contiguous())
        z = torch.empty((n, cout, h, wd), dtype=torch.float32, device=dev)
        nbytes = lib.afd_conv2d_workspace_bytes(n, c, h, wd, cout, 1, 0, 1)
        ws = _ws(nbytes, dev)
        _native.check(lib.afd_conv2d_forward(
            _native.ptr(u), _native.ptr(wf), _native.ptr(bf), _native.ptr(z), n, c, h, wd, cout, 1, 0, 1,
            _native.ptr(ws), ws.numel(), _native.stream_ptr()), "afd_conv2d_forward")
        ctx.save_for_backward(u, w2, wf, mean, invstd)
        ctx.geom = (n, c, h, wd, cout)
        ctx.flags = (training, sync, b is not None, tuple(w.shape))
        return z

    @staticmethod
    def backward(ctx, dz):
        lib = _lib()
        u, w2, wf, mean, invstd = ctx.saved_tensors
        n, c, h, wd, cout = ctx.geom
        training, sync, has_bias, wshape = ctx.flags
        dz = _f32c(dz)
        dev = u.device
        nbytes = lib.afd_conv2d_workspace_bytes(n, c, h, wd, cout, 1, 0, 1)
        ws = _ws(nbytes, dev)
        # gradients against the un-normalised input, then against xhat = (u - mean) * invstd
        g = torch.empty((cout, c), dtype=torch.float32, device=dev)
        db = torch.empty(cout, dtype=torch.float32, device=dev)
        _native.check(lib.afd_conv2d_backward_weight(
            _native.ptr(u), _native.ptr(dz), _native.ptr(g), _native.ptr(db), n, c, h, wd, cout, 1, 0, 1,
            _native.ptr(ws), ws.numel(), _native.stream_ptr()), "afd_conv2d_backward_weight")
        need_aff = bool(ctx.needs_input_grad[0] and training)
        # sums over pixels of dxhat and of dxhat * xhat come from the small matrices alone
        dw, alpha, beta = _fold_backward(g, db, w2, mean.contiguous(), invstd.contiguous(),
                                         ctx.count if training else 1.0, sync, need_aff)
        du = None
        if ctx.needs_input_grad[0]:
            du = torch.empty_like(u)
            if training:
                _native.check(lib.afd_conv1x1_bn_backward_data(
                    _native.ptr(dz), _native.ptr(wf), _native.ptr(u), _native.ptr(alpha),
                    _native.ptr(beta), _native.ptr(du), n, c, cout, h * wd, _native.stream_ptr()),
                    "afd_conv1x1_bn_backward_data")
            else:
                _native.check(lib.afd_conv2d_backward_data(
                    _native.ptr(dz), _native.ptr(wf), _native.ptr(du), n, c, h, wd, cout, 1, 0, 1,
                    _native.ptr(ws), ws.numel(), _native.stream_ptr()), "afd_conv2d_backward_data")
        return (du, dw.reshape(wshape) if ctx.needs_input_grad[1] else None,
                db if (has_bias and ctx.needs_input_grad[2]) else None,
                None, None, None, None, None, None, None)


def bn_conv1x1(u, bn: torch.nn.Module, w, b, sync: bool = True):
    """conv1x1(batch_norm(u)) for a BatchNorm without affine parameters, without materialising the
    normalised tensor (see `_BNConv1x1`)."""
    training = bn.training or bn.running_mean is None
    return _BNConv1x1.apply(u, w, b, bn.running_mean, bn.running_var, bn.num_batches_tracked,
                            training, bn.momentum, bn.eps, sync)


def bn_conv1x1_applicable(bn: torch.nn.Module, conv: torch.nn.Module) -> bool:
    import os
    return (not os.environ.get("AFD_NO_BN_FOLD") and bn.weight is None and bn.bias is None
            and bn.running_mean is not None
            and conv.kernel_size == (1, 1) and conv.padding == (0, 0) and conv.stride == (1, 1)
            and conv.in_channels <= 128 and conv.out_channels <= 128)


def _bn_finalize_sums(sums, c, count, bn, sync, fold_tab=None):
    """(mean, invstd, count) from the packed [sum | sum of squares | count slot] double vector of a training
    batch (all-reduced first when a process group is up); updates bn's running buffers.  `fold_tab`: a [c][2] tensor
    that takes the (mean, invstd) pairs in the same launch."""
    lib = _lib()
    dev = sums.device
    dist_on = _dist_on(sync)
    if dist_on:
        sums[2 * c] = count
        all_reduce_sum(sums)
    mean = torch.empty(c, dtype=torch.float32, device=dev)
    invstd = torch.empty(c, dtype=torch.float32, device=dev)
    cnt = torch.empty(1, dtype=torch.float64, device=dev) if dist_on else None
    mom = bn.momentum
    nbt = bn.num_batches_tracked
    if mom is None:
        mom = 1.0 / float(int(nbt) + 1) if nbt is not None else 0.0
    with torch.no_grad():
        _native.check(lib.afd_bn_finalize(
            _native.ptr(sums), c, -1.0 if dist_on else count, float(bn.eps), float(mom),
            _native.ptr(mean), _native.ptr(invstd), _native.ptr(bn.running_mean),
            _native.ptr(bn.running_var), _native.ptr(nbt), _native.ptr(cnt), _native.ptr(fold_tab),
            _native.stream_ptr()), "afd_bn_finalize")
    return mean, invstd, (cnt if dist_on else count)


def _bn_batch_stats(x, slope, c, n, hw, bn, sync, pre_sums=None):
    """Training-mode statistics of PReLU(x) (or x): (mean, invstd, count); updates bn's running buffers.  `pre_sums`:
    the packed sums from the producer's epilogue (no pass over x)."""
    sums = pre_sums
    if sums is None:
        sums = torch.empty(2 * c + 1, dtype=torch.float64, device=x.device)
        _native.check(_lib().afd_bn_stats(_native.ptr(x), _native.ptr(slope), _native.ptr(sums), n, c, hw,
                                          _native.stream_ptr()), "afd_bn_stats")
    return _bn_finalize_sums(sums, c, float(n * hw), bn, sync)


class _BNConv1x1PReLUBN(torch.autograd.Function):
    """DCNN block 2 in training mode (reference models.py:260-264):

        u -> BatchNorm2d(affine=False) -> Conv2d(k=1) -> z -> PReLU -> BatchNorm2d(affine=False) -> xhat

    Forward: the first normalisation is folded into the weights (as `_BNConv1x1`), the second one runs on
    PReLU(z) without materialising it.  Backward: ONE pass over the activations
    (`afd_conv1x1_prelu_bn_backward`) does the second BatchNorm's backward, the PReLU backward and both
    GEMMs of the convolution's backward; the affine term of the first BatchNorm's backward is handed to the
    producer of u through `link` (or added here when there is no such producer)."""

    @staticmethod
    def forward(ctx, u, w, b, slope, bn1, bn2, sync, link, out_link=None, defer=False):
        ctx.out_link = out_link
        lib = _lib()
        u = _f32c(u)
        n, c, h, wd = u.shape
        cout = w.shape[0]
        hw = h * wd
        dev = u.device
        pre1 = link.pop("fwd_sums", None) if link is not None else None  # from the first block's forward launch
        mean1, invstd1, cnt1 = _bn_batch_stats(u, None, c, n, hw, bn1, sync, pre1)
        w2 = _f32c(w).reshape(cout, c)
        wf, bf = _fold_forward(w2, b, mean1, invstd1)
        z = torch.empty((n, cout, h, wd), dtype=torch.float32, device=dev)
        # the convolution's epilogue also sums PReLU(z) and its square per channel: no statistics pass
        ws = _ws(lib.afd_conv1x1_forward_stats_workspace_bytes(cout), dev)
        sums2 = torch.empty(2 * cout + 1, dtype=torch.float64, device=dev)
        _native.check(lib.afd_conv1x1_forward_stats(
            _native.ptr(u), _native.ptr(wf), _native.ptr(bf), _native.ptr(slope), _native.ptr(z),
            _native.ptr(sums2), n, c, cout, hw, _native.ptr(ws), ws.numel(), _native.stream_ptr()),
            "afd_conv1x1_forward_stats")
        fold2 = torch.empty((cout, 2), dtype=torch.float32, device=dev) if defer and out_link is not None else None
        mean2, invstd2, cnt2 = _bn_finalize_sums(sums2, cout, float(n * hw), bn2, sync, fold2)
        if defer and out_link is not None:
            # the 3x3 convolution behind the second BatchNorm applies PReLU and the normalisation while it loads z
            out_link["fold"] = (fold2, slope)
            y = z
        else:
            y = torch.empty_like(z)
            _native.check(lib.afd_bn_apply_forward(
                _native.ptr(z), _native.ptr(slope), _native.ptr(mean2), _native.ptr(invstd2), None, None,
                _native.ptr(y), n, cout, hw, _native.stream_ptr()), "afd_bn_apply_forward")
        _tap("prelu", z)
        if out_link is not None:
            out_link["bn"] = True
        ctx.save_for_backward(u, z, w2, wf, mean1, invstd1, mean2, invstd2, slope)
        ctx.geom = (n, c, h, wd, cout)
        ctx.cfg = (sync, b is not None, tuple(w.shape), cnt1, cnt2, link)
        return y

    @staticmethod
    def backward(ctx, g):
        lib = _lib()
        u, z, w2, wf, mean1, invstd1, mean2, invstd2, slope = ctx.saved_tensors
        n, c, h, wd, cout = ctx.geom
        sync, has_bias, wshape, cnt1, cnt2, link = ctx.cfg
        hw = h * wd
        dev = u.device
        g = _f32c(g)
        # second BatchNorm: batch sums of g and g * xhat -- from the launch that produced g, or one reduction
        # pass -- folded into per-channel constants
        sums = ctx.out_link.pop("bwd_sums", None) if ctx.out_link is not None else None
        if sums is None:
            sums = torch.empty(2 * cout, dtype=torch.float64, device=dev)
            _native.check(lib.afd_bn_backward_stats(
                _native.ptr(z), _native.ptr(slope), _native.ptr(g), _native.ptr(mean2), _native.ptr(invstd2),
                _native.ptr(sums), n, cout, hw, _native.stream_ptr()), "afd_bn_backward_stats")
        if _dist_on(sync):
            all_reduce_sum(sums)
        mdy = torch.empty(cout, dtype=torch.float32, device=dev)
        mdyx = torch.empty(cout, dtype=torch.float32, device=dev)
        on_dev = torch.is_tensor(cnt2)
        _native.check(lib.afd_bn_backward_means(
            _native.ptr(sums), cout, -1.0 if on_dev else float(cnt2), _native.ptr(cnt2) if on_dev else None,
            _native.ptr(mdy), _native.ptr(mdyx), None, None, None, _native.stream_ptr()), "afd_bn_backward_means")
        coef = torch.empty((cout, 4), dtype=torch.float32, device=dev)
        _native.check(lib.afd_bn_backward_coef(_native.ptr(mean2), _native.ptr(invstd2), _native.ptr(mdy),
                                               _native.ptr(mdyx), _native.ptr(coef), cout, _native.stream_ptr()),
                      "afd_bn_backward_coef")
        t = torch.empty_like(u)
        gw = torch.empty((cout, c), dtype=torch.float32, device=dev)
        db = torch.empty(cout, dtype=torch.float32, device=dev)
        dslope = _zeros(1, torch.float32, dev)
        ws = _ws(lib.afd_conv1x1_prelu_bn_backward_workspace_bytes(c, cout), dev)
        _native.check(lib.afd_conv1x1_prelu_bn_backward(
            _native.ptr(g), _native.ptr(z), _native.ptr(u), _native.ptr(wf), _native.ptr(coef),
            _native.ptr(slope), _native.ptr(t), _native.ptr(gw), _native.ptr(db), _native.ptr(dslope),
            n, c, cout, hw, _native.ptr(ws), ws.numel(), _native.stream_ptr()),
            "afd_conv1x1_prelu_bn_backward")
        # first BatchNorm, from the small matrices alone (as in `_BNConv1x1`)
        dw, alpha, beta = _fold_backward(gw, db, w2, mean1, invstd1, cnt1, sync, bool(ctx.needs_input_grad[0]))
        du = None
        if ctx.needs_input_grad[0]:
            if link is not None:
                link["affine"] = (alpha, beta)  # added by the producer of u where it reads du and u
                du = t
            else:
                du = torch.addcmul(t + beta.view(1, -1, 1, 1), u, alpha.view(1, -1, 1, 1))
        return (du, dw.reshape(wshape) if ctx.needs_input_grad[1] else None,
                db if (has_bias and ctx.needs_input_grad[2]) else None,
                dslope if ctx.needs_input_grad[3] else None, None, None, None, None, None, None)


def bn_conv1x1_prelu_bn_applicable(bn1: torch.nn.Module, conv: torch.nn.Module, bn2: torch.nn.Module) -> bool:
    if not bn_conv1x1_applicable(bn1, conv):
        return False
    if not (bn1.training and bn2.training and bn2.weight is None and bn2.bias is None
            and bn2.running_mean is not None):
        return False
    return bool(_lib().afd_conv1x1_prelu_bn_backward_applicable(conv.in_channels, conv.out_channels))


def bn_conv1x1_prelu_bn(u, bn1, w, b, slope, bn2, sync: bool = True, link: Optional[dict] = None,
                        out_link: Optional[dict] = None, defer: bool = False):
    """batch_norm2(PReLU(conv1x1(batch_norm1(u)))) in training mode with the one-pass backward of
    `_BNConv1x1PReLUBN`.  `link`: the dict given to `conv1_prelu_maxpool` when that call produced u;
    `out_link`: as `batch_norm`'s `link`, for the second BatchNorm and the convolution that consumes the result;
    `defer`: as `batch_norm`'s, for the second BatchNorm (z comes back instead of the normalised tensor)."""
    return _BNConv1x1PReLUBN.apply(u, w, b, slope, bn1, bn2, sync, link, out_link, bool(defer))


def batch_norm(x, bn: torch.nn.Module, slope: Optional[torch.Tensor] = None, sync: bool = True,
               link: Optional[dict] = None, prod_link: Optional[dict] = None, sum_link: Optional[dict] = None,
               defer: bool = False):
    """BatchNorm (batch statistics across all ranks when a process group is up) of
    PReLU(x) if `slope` is given, else of x.  `bn` carries weight/bias/running stats.  `link`: a dict shared with
    the convolution that is the ONLY consumer of the result (its `bn_link`): that layer's backward-data launch
    then also produces this layer's backward sums.  `prod_link`: a dict shared with the PReLU + max-pool call that
    produced x (its `out_link`, this layer being the ONLY consumer of x): in training mode and without affine
    parameters this layer's backward is then applied inside the pool's backward.  `sum_link`: a dict shared with the
    convolution that produced x (its `out_link`, this layer being the ONLY consumer of x): the backward pass leaves
    the per-channel sums of its result there -- that convolution's bias gradient.  `defer` (with `link`; the caller
    asked `conv3x3_input_fold_applicable`): the result is NOT computed -- x itself comes back, and the consuming
    convolution, handed `link` as its `bn_link`, normalises it while it loads."""
    training = bn.training or bn.running_mean is None
    # batch sums left by the producer of x (`afd_conv3x3_forward_stats`, asked for through "want_stats")
    pre = None
    for lk in (prod_link, sum_link):
        if lk is not None and "fwd_sums" in lk:
            pre = lk.pop("fwd_sums")
    if not training or bn.weight is not None or slope is not None:
        prod_link = None
    if not training:
        sum_link = pre = None
    return _BatchNorm.apply(x, slope, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                            bn.num_batches_tracked, training, bn.momentum, bn.eps, sync, link, prod_link, sum_link, pre,
                            bool(defer))


# --------------------------------------------------------------------------------------
class _DropoutPermute(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, seed):
        x = _f32c(x)
        b, c, h, w = x.shape
        y = torch.empty((b, h, c, w), dtype=torch.float32, device=x.device)
        _native.check(_lib().afd_dropout_permute(_native.ptr(x), _native.ptr(y), b, c, h, w, p, seed,
                                                 0, _native.stream_ptr()), "afd_dropout_permute")
        ctx.cfg = (b, c, h, w, p, seed)
        return y

    @staticmethod
    def backward(ctx, dy):
        b, c, h, w, p, seed = ctx.cfg
        dy = _f32c(dy)
        dx = torch.empty((b, c, h, w), dtype=torch.float32, device=dy.device)
        _native.check(_lib().afd_dropout_permute(_native.ptr(dy), _native.ptr(dx), b, c, h, w, p,
                                                 seed, 1, _native.stream_ptr()),
                      "afd_dropout_permute(inverse)")
        return dx, None, None


def dropout_permute(x, p: float, training: bool):
    """Dropout(p)(x).permute(0, 2, 1, 3).contiguous()."""
    p = float(p) if training else 0.0
    return _DropoutPermute.apply(x, p, next_seed() if p > 0 else 0)


class _PReLUDropout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, slope, p, seed):
        z = _f32c(z)
        y = torch.empty_like(z)
        _native.check(_lib().afd_prelu_dropout_forward(_native.ptr(z), _native.ptr(slope),
                                                       _native.ptr(y), z.numel(), p, seed,
                                                       _native.stream_ptr()),
                      "afd_prelu_dropout_forward")
        _tap("prelu", z)
        ctx.save_for_backward(z, slope)
        ctx.cfg = (p, seed)
        return y

    @staticmethod
    def backward(ctx, dy):
        z, slope = ctx.saved_tensors
        p, seed = ctx.cfg
        dy = _f32c(dy)
        dz = torch.empty_like(z)
        dslope = _zeros(1, torch.float32, z.device)
        _native.check(_lib().afd_prelu_dropout_backward(
            _native.ptr(z), _native.ptr(slope), _native.ptr(dy), _native.ptr(dz), _native.ptr(dslope),
            z.numel(), p, seed, _native.stream_ptr()), "afd_prelu_dropout_backward")
        return dz, dslope, None, None


def prelu_dropout(z, slope, p: float, training: bool):
    p = float(p) if training else 0.0
    return _PReLUDropout.apply(z, slope, p, next_seed() if p > 0 else 0)


# --------------------------------------------------------------------------------------
class _LinearMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        x = _f32c(x)
        bsz, td, f = x.shape
        o = w.shape[0]
        if w.dim() != 2 or w.shape[1] != f or (b is not None and b.numel() != o):
            # (nn.Linear raises here too: a wrong `flattend_size` must not read past the weight)
            raise RuntimeError(f"linear_mean: features of width {f} against a weight of shape {tuple(w.shape)}")
        y = torch.empty((bsz, o), dtype=torch.float32, device=x.device)
        _native.check(_lib().afd_linear_mean_forward(_native.ptr(x), _native.ptr(_f32c(w)),
                                                     _native.ptr(b), _native.ptr(y), bsz, td, f, o,
                                                     _native.stream_ptr()), "afd_linear_mean_forward")
        ctx.save_for_backward(x, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        bsz, td, f = x.shape
        o = w.shape[0]
        dy = _f32c(dy)
        dx = torch.empty_like(x)
        dw = torch.empty_like(w)
        db = torch.empty(o, dtype=torch.float32, device=x.device)
        _native.check(_lib().afd_linear_mean_backward(
            _native.ptr(x), _native.ptr(_f32c(w)), _native.ptr(dy), _native.ptr(dx), _native.ptr(dw),
            _native.ptr(db), bsz, td, f, o, _native.stream_ptr()), "afd_linear_mean_backward")
        return dx, dw, db


def linear_mean(x, w, b):
    """(x @ w.T + b).mean(1) for x [B, TD, F]."""
    return _LinearMean.apply(x, w, b)


# --------------------------------------------------------------------------------------
class _CrossEntropy(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels):
        logits = _f32c(logits)
        labels = labels.to(torch.int64).contiguous()
        b, o = logits.shape
        out = torch.empty(2, dtype=torch.float32, device=logits.device)  # loss, correct
        dl = torch.empty_like(logits)
        _native.check(_lib().afd_cross_entropy(
            _native.ptr(logits), _native.ptr(labels), _native.ptr(out),
            _native.ptr(dl), _native.c_p(out.data_ptr() + 4), b, o, _native.stream_ptr()),
            "afd_cross_entropy")
        ctx.save_for_backward(dl)
        ctx.mark_non_differentiable(out)
        loss = out[0].clone()
        ctx.stats = out
        return loss, out

    @staticmethod
    def backward(ctx, dloss, _dout):
        (dl,) = ctx.saved_tensors
        if _end_of_backward:
            # the loss is the first node of every backward pass: callbacks queued here run once the
            # engine has finished the whole graph
            torch.autograd.Variable._execution_engine.queue_callback(_run_end_of_backward)
        return dl * dloss, None


def cross_entropy(logits, labels):
    """Mean cross entropy; returns (loss, stats) with stats = [loss, #correct]."""
    return _CrossEntropy.apply(logits, labels)


class CrossEntropyLoss(torch.nn.Module):
    """Drop-in for torch.nn.CrossEntropyLoss() on [B, O] logits (train_classifier.py:1212)."""

    def forward(self, logits, labels):
        loss, stats = cross_entropy(logits, labels)
        self.last_stats = stats
        return loss


# --------------------------------------------------------------------------------------
class FusedAdam(torch.optim.Optimizer):
    """Adam with coupled L2 over ONE flat fp32 arena (params, grads, m, v contiguous).

    Same update rule as ``torch.optim.Adam(params, lr, weight_decay=wd)`` (reference
    train_classifier.py:1215-1219).  The parameters are re-pointed into ``self.flat``; the gradient all-reduce of the
    data-parallel step is one collective over ``flat_grad`` and the update is one launch.

    Gradients: ``zero_grad`` drops them (``p.grad = None``), so the backward pass hands every parameter its own fresh
    gradient tensor without a launch; ``gather_grads`` -- implied by reading ``flat_grad`` and by ``step`` -- copies
    them into the arena in ONE launch (`afd_multi_gather`) and re-points ``p.grad`` at the arena slices.  (Until
    round 3 the gradients were arena views from the start and autograd added every one of them in with a launch of its
    own: ~50 four-microsecond kernels per step.)  A second backward pass before the step accumulates into the slices in
    place, as torch does.
    """

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        params = [p for p in params]
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._params = [p for g in self.param_groups for p in g["params"] if p.requires_grad]
        dev = self._params[0].device
        total = sum(p.numel() for p in self._params)
        self.flat = torch.empty(total, dtype=torch.float32, device=dev)
        self._flat_grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.m = torch.zeros(total, dtype=torch.float32, device=dev)
        self.v = torch.zeros(total, dtype=torch.float32, device=dev)
        off = 0
        self._offsets = []
        with torch.no_grad():
            for p in self._params:
                n = p.numel()
                self.flat[off:off + n].copy_(p.detach().reshape(-1))
                p.data = self.flat[off:off + n].view_as(p)
                self._offsets.append(off)
                off += n
        self._relink()
        self.step_count = 0

    @property
    def flat_grad(self) -> torch.Tensor:
        """The gradient arena, with every parameter's gradient in its slice."""
        self.gather_grads()
        return self._flat_grad

    def zero_grad(self, set_to_none: bool = True):
        for p in self._params:
            p.grad = None

    def _relink(self):
        base = self._flat_grad.data_ptr()
        for p, off in zip(self._params, self._offsets):
            if p.grad is None or p.grad.data_ptr() != base + 4 * off:
                p.grad = self._flat_grad[off:off + p.numel()].view_as(p)

    @torch.no_grad()
    def gather_grads(self) -> None:
        """Every gradient that is not in its arena slice yet goes there (a missing one as zeros)."""
        base = self._flat_grad.data_ptr()
        todo = []
        for p, off in zip(self._params, self._offsets):
            g = p.grad
            if g is not None and g.data_ptr() == base + 4 * off:
                continue
            if g is not None and (g.dtype != torch.float32 or not g.is_contiguous()):
                g = g.float().contiguous()
            todo.append((g, off, p.numel()))
        if not todo:
            return
        if self._flat_grad.is_cuda:
            n = len(todo)
            srcs = (_native.c_p * n)(*[(g.data_ptr() if g is not None else 0) for g, _, _ in todo])
            offs = (_native.c_l * n)(*[off for _, off, _ in todo])
            cnts = (_native.c_l * n)(*[cnt for _, _, cnt in todo])
            _native.check(_lib().afd_multi_gather(srcs, offs, cnts, n, _native.ptr(self._flat_grad),
                                                  _native.stream_ptr()), "afd_multi_gather")
        else:
            for g, off, cnt in todo:
                if g is None:
                    self._flat_grad[off:off + cnt].zero_()
                else:
                    self._flat_grad[off:off + cnt].copy_(g.reshape(-1))
        self._relink()  # (the sources die here; the allocator is stream-ordered)

    @torch.no_grad()
    def step(self, closure=None, grad_scale: float = 1.0):
        g = self.param_groups[0]
        self.step_count += 1
        self.gather_grads()
        _native.check(_lib().afd_adam_step(
            _native.ptr(self.flat), _native.ptr(self._flat_grad), _native.ptr(self.m),
            _native.ptr(self.v), self.flat.numel(), g["lr"], g["betas"][0], g["betas"][1], g["eps"],
            g["weight_decay"], self.step_count, grad_scale, _native.stream_ptr()), "afd_adam_step")
        return None
