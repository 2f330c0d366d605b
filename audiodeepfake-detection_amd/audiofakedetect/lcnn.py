"""LCNN (ASVspoof-2021 LA baseline head) on libafd_hip -- reference models.py:68-131.

Layer objects hold parameters / buffers under the reference's state_dict keys (``lcnn.N.*``,
``lstm.K.l_blstm.*``, ``fc.*``); ``forward`` drives HIP kernels: conv (MFMA implicit GEMM),
max-feature-map, max-pool, BatchNorm, dropout+permute, and the two bidirectional LSTM layers as
input/recurrent projections on ``afd_gemm_nt`` plus the fused cell kernel.  The convolutional
trunk and the LSTM layers support backward (BPTT over the handful of time steps left after the
four pools, ``afd_lstm_cell_backward`` + the same GEMM kernel).
"""

from __future__ import annotations

from typing import Optional

import os

import torch
import torch.nn as nn

from . import _native, ops


class MaxFeatureMap2D(nn.Module):
    """Max over the two channel halves (reference models.py:161-209)."""

    def __init__(self, max_dim: int = 1) -> None:
        super().__init__()
        self.max_dim = max_dim

    def forward(self, inputs: torch.Tensor) -> torch.Tensor:
        return max_feature_map(inputs)


class _MFM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = ops._f32c(x)
        n, c = x.shape[0], x.shape[1]
        if c % 2:
            raise ValueError("MaxFeatureMap: odd number of channels")
        hw = x.numel() // (n * c)
        y = torch.empty((n, c // 2) + tuple(x.shape[2:]), dtype=torch.float32, device=x.device)
        sel = torch.empty(y.shape, dtype=torch.uint8, device=x.device)
        _native.check(_native.load().afd_mfm_forward(_native.ptr(x), _native.ptr(y), _native.ptr(sel),
                                                     n, c, hw, _native.stream_ptr()), "afd_mfm_forward")
        ctx.save_for_backward(sel)
        ctx.shape = tuple(x.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        (sel,) = ctx.saved_tensors
        n, c = ctx.shape[0], ctx.shape[1]
        hw = sel.numel() // (n * (c // 2))
        dx = torch.empty(ctx.shape, dtype=torch.float32, device=dy.device)
        _native.check(_native.load().afd_mfm_backward(_native.ptr(ops._f32c(dy)), _native.ptr(sel),
                                                      _native.ptr(dx), n, c, hw, _native.stream_ptr()),
                      "afd_mfm_backward")
        return dx


def max_feature_map(x: torch.Tensor) -> torch.Tensor:
    return _MFM.apply(x)


def gemm_nt(a: torch.Tensor, b: torch.Tensor, bias=None, out=None, accumulate: bool = False,
            bf16: bool = False):
    """out[M,N] (+)= a[M,K] @ b[N,K].T + bias; `a`/`out` may be row-strided views.  ``bf16``: operands
    rounded to bf16, fp32 accumulation (``afd_gemm_nt_bf16``)."""
    m, k = a.shape
    n = b.shape[0]
    if out is None:
        out = torch.empty((m, n), dtype=torch.float32, device=a.device)
    assert a.stride(1) == 1 and b.stride(1) == 1 and out.stride(1) == 1
    fn = _native.load().afd_gemm_nt_bf16 if bf16 else _native.load().afd_gemm_nt
    _native.check(fn(
        _native.ptr(a), _native.ptr(b), _native.ptr(bias), _native.ptr(out), m, n, k, a.stride(0),
        b.stride(0), out.stride(0), 1 if accumulate else 0, _native.stream_ptr()), "afd_gemm_nt")
    return out


def conv2d_bf16(x: torch.Tensor, w: torch.Tensor, b, padding: int, mfm: bool = False) -> torch.Tensor:
    """conv2d(x, w, b, padding) with bf16 operands on the matrix cores (inference only: no autograd node);
    ``mfm``: max-feature-map over the two channel halves applied in the epilogue (Cout / 2 channels out)."""
    lib = _native.load()
    x = ops._f32c(x)
    w = ops._f32c(w)
    n, cin, h, wd = x.shape
    cout, _, k, _ = w.shape
    ho, wo = h + 2 * padding - (k - 1), wd + 2 * padding - (k - 1)
    y = torch.empty((n, cout // 2 if mfm else cout, ho, wo), dtype=torch.float32, device=x.device)
    nbytes = lib.afd_conv2d_bf16_workspace_bytes(cin, cout, k)
    ws = ops._ws(nbytes, x.device)
    _native.check(lib.afd_conv2d_forward_bf16(
        _native.ptr(x), _native.ptr(w), _native.ptr(b), _native.ptr(y), n, cin, h, wd, cout, k, int(padding),
        1 if mfm else 0, _native.ptr(ws), ws.numel(), _native.stream_ptr()), "afd_conv2d_forward_bf16")
    return y


def lstm_fragment_order(wh: torch.Tensor) -> torch.Tensor:
    """weight_hh [4H][H] -> the matrix-fragment order `afd_blstm_layer_bf16` reads ([H/8][H/16][64 lanes][8])."""
    h = wh.shape[1]
    return wh.view(4, h // 8, 8, h // 16, 2, 8).permute(1, 3, 4, 0, 2, 5).contiguous()


def blstm_forward_bf16(x: torch.Tensor, m: nn.LSTM, wih: Optional[dict] = None, whh: Optional[dict] = None,
                       bias2: Optional[dict] = None) -> torch.Tensor:
    """Inference forward of a bidirectional LSTM layer with bf16 projections (cell update in fp32).  `wih`:
    replacement input weights per direction suffix (columns permuted to the caller's feature order); `whh`:
    recurrent weights per direction suffix already converted to bf16 (`afd_f32_to_bf16`) -- the step then is one
    launch (`afd_lstm_step_bf16`: projection + cell)."""
    lib = _native.load()
    bsz, steps, d = x.shape
    h = m.weight_hh_l0.shape[1]
    xt = x.permute(1, 0, 2).contiguous().view(steps * bsz, d)
    out = torch.empty((steps, bsz, 2 * h), dtype=torch.float32, device=x.device)
    if whh is not None and h % 16 == 0 and h <= 256 and not os.environ.get("AFD_LSTM_STEPWISE"):
        # the whole layer -- every step of both directions -- in one launch (`afd_blstm_layer_bf16`)
        pres = []
        for sfx in ("", "_reverse"):
            wi = wih[sfx] if wih is not None else getattr(m, "weight_ih_l0" + sfx)
            bias = bias2[sfx] if bias2 is not None else getattr(m, "bias_ih_l0" + sfx) + getattr(m, "bias_hh_l0" + sfx)
            pres.append(gemm_nt(xt, ops._f32c(wi), bias, bf16=True))
        frag = [whh.get("frag" + sfx) for sfx in ("", "_reverse")]
        if frag[0] is None:  # callers without a prepared plan: pack on the fly
            frag = [lstm_fragment_order(whh[sfx]) for sfx in ("", "_reverse")]
        _native.check(lib.afd_blstm_layer_bf16(_native.ptr(pres[0]), _native.ptr(pres[1]), _native.ptr(frag[0]),
                                               _native.ptr(frag[1]), _native.ptr(out), steps, bsz, h,
                                               _native.stream_ptr()), "afd_blstm_layer_bf16")
        return out.permute(1, 0, 2).contiguous()
    if whh is not None and h % 16 == 0:
        # both directions advance together: one launch per time step (`afd_lstm_step_bf16_pair`)
        pres, hs2, cs2 = [], [], []
        state = torch.zeros((2, 3, bsz, h), dtype=torch.float32, device=x.device)  # per direction: two h buffers and c
        for d, sfx in enumerate(("", "_reverse")):
            wi = wih[sfx] if wih is not None else getattr(m, "weight_ih_l0" + sfx)
            bias = bias2[sfx] if bias2 is not None else getattr(m, "bias_ih_l0" + sfx) + getattr(m, "bias_hh_l0" + sfx)
            pres.append(gemm_nt(xt, ops._f32c(wi), bias, bf16=True).view(steps, bsz, 4 * h))
            hs2.append(state[d, :2])
            cs2.append(state[d, 2])
        two = _native.c_p * 2
        whp = two(whh[""].data_ptr(), whh["_reverse"].data_ptr())
        cp = two(cs2[0].data_ptr(), cs2[1].data_ptr())
        for n_step in range(steps):
            tf, tr = n_step, steps - 1 - n_step
            a, b2 = n_step & 1, 1 - (n_step & 1)
            _native.check(lib.afd_lstm_step_bf16_pair(
                two(pres[0][tf].data_ptr(), pres[1][tr].data_ptr()), whp,
                two(hs2[0][a].data_ptr(), hs2[1][a].data_ptr()), cp,
                two(out[tf, :, 0:h].data_ptr(), out[tr, :, h:2 * h].data_ptr()), 2 * h,
                two(hs2[0][b2].data_ptr(), hs2[1][b2].data_ptr()), bsz, h, _native.stream_ptr()), "afd_lstm_step_bf16_pair")
        return out.permute(1, 0, 2).contiguous()
    for direction, sfx in enumerate(("", "_reverse")):
        wi, wh = getattr(m, "weight_ih_l0" + sfx), getattr(m, "weight_hh_l0" + sfx)
        bias = getattr(m, "bias_ih_l0" + sfx) + getattr(m, "bias_hh_l0" + sfx)
        if wih is not None:
            wi = wih[sfx]
        pre = gemm_nt(xt, ops._f32c(wi), bias, bf16=True).view(steps, bsz, 4 * h)
        hs = torch.zeros((2, bsz, h), dtype=torch.float32, device=x.device)
        cs = torch.zeros((bsz, h), dtype=torch.float32, device=x.device)
        order = range(steps) if direction == 0 else range(steps - 1, -1, -1)
        if whh is not None and h % 16 == 0:
            for n_step, t in enumerate(order):
                hout = out[t, :, direction * h:(direction + 1) * h]
                _native.check(lib.afd_lstm_step_bf16(
                    _native.ptr(pre[t]), _native.ptr(whh[sfx]), _native.ptr(hs[n_step & 1]), _native.ptr(cs), _native.ptr(hout),
                    2 * h, _native.ptr(hs[1 - (n_step & 1)]), bsz, h, _native.stream_ptr()), "afd_lstm_step_bf16")
            continue
        whc = ops._f32c(wh)
        for t in order:
            gates = pre[t]
            gemm_nt(hs[0], whc, None, out=gates, accumulate=True, bf16=True)
            hout = out[t, :, direction * h:(direction + 1) * h]
            _native.check(lib.afd_lstm_cell(_native.ptr(gates), _native.ptr(cs), _native.ptr(hout),
                                            _native.ptr(hs[0]), bsz, h, 2 * h, _native.stream_ptr()), "afd_lstm_cell")
    return out.permute(1, 0, 2).contiguous()


class _BLSTM(torch.autograd.Function):
    """Bidirectional single-layer LSTM, x [B,T,D] -> [B,T,2H] (nn.LSTM semantics).

    Forward: one input projection GEMM per direction, then per step the recurrent projection
    accumulated into the step's gate rows (``afd_gemm_nt``) and the fused cell kernel.  For the
    backward pass the gate pre-activations and the cell states of every step are kept; BPTT runs
    ``afd_lstm_cell_backward`` per step and the same GEMM kernel for dh, dx, dW.
    """

    @staticmethod
    def forward(ctx, x, w_ih, w_hh, b_ih, b_hh, w_ih_r, w_hh_r, b_ih_r, b_hh_r):
        lib = _native.load()
        bsz, steps, d = x.shape
        h = w_hh.shape[1]
        xt = x.permute(1, 0, 2).contiguous().view(steps * bsz, d)  # [T*B, D]
        out = torch.empty((steps, bsz, 2 * h), dtype=torch.float32, device=x.device)
        keep = any(ctx.needs_input_grad)
        saved = []
        for direction, (wi, wh, bi, bh) in enumerate(((w_ih, w_hh, b_ih, b_hh),
                                                      (w_ih_r, w_hh_r, b_ih_r, b_hh_r))):
            pre = gemm_nt(xt, ops._f32c(wi), bi + bh).view(steps, bsz, 4 * h)
            hs = torch.zeros((bsz, h), dtype=torch.float32, device=x.device)
            cs = torch.zeros((bsz, h), dtype=torch.float32, device=x.device)
            c_all = torch.empty((steps, bsz, h), dtype=torch.float32, device=x.device) if keep else None
            whc = ops._f32c(wh)
            order = range(steps) if direction == 0 else range(steps - 1, -1, -1)
            for t in order:
                gates = pre[t]
                gemm_nt(hs, whc, None, out=gates, accumulate=True)
                hout = out[t, :, direction * h:(direction + 1) * h]
                _native.check(lib.afd_lstm_cell(_native.ptr(gates), _native.ptr(cs), _native.ptr(hout),
                                                _native.ptr(hs), bsz, h, 2 * h, _native.stream_ptr()),
                              "afd_lstm_cell")
                if keep:
                    c_all[t].copy_(cs)
            saved += [pre, c_all]
        if keep:
            ctx.save_for_backward(xt, out, ops._f32c(w_ih), ops._f32c(w_hh), ops._f32c(w_ih_r),
                                  ops._f32c(w_hh_r), *saved)
            ctx.dims = (bsz, steps, d, h)
        return out.permute(1, 0, 2).contiguous()

    @staticmethod
    def backward(ctx, dy):
        lib = _native.load()
        xt, out, w_ih, w_hh, w_ih_r, w_hh_r, pre_f, c_f, pre_r, c_r = ctx.saved_tensors
        bsz, steps, d, h = ctx.dims
        dout = ops._f32c(dy).permute(1, 0, 2).contiguous()  # [T, B, 2H]
        dx = torch.zeros((steps * bsz, d), dtype=torch.float32, device=dy.device)
        grads = []
        for direction, (wi, wh, pre, c_all) in enumerate(((w_ih, w_hh, pre_f, c_f),
                                                          (w_ih_r, w_hh_r, pre_r, c_r))):
            wh_t = wh.t().contiguous()  # [H, 4H]: dh_prev = dpre @ W_hh
            dpre = torch.empty((steps, bsz, 4 * h), dtype=torch.float32, device=dy.device)
            dc = torch.zeros((bsz, h), dtype=torch.float32, device=dy.device)
            dh = torch.zeros((bsz, h), dtype=torch.float32, device=dy.device)
            hprev = torch.zeros((steps, bsz, h), dtype=torch.float32, device=dy.device)
            order = list(range(steps)) if direction == 0 else list(range(steps - 1, -1, -1))
            for pos in range(len(order) - 1, -1, -1):  # reverse of the processing order
                t = order[pos]
                tprev = order[pos - 1] if pos > 0 else None
                dh.add_(dout[t, :, direction * h:(direction + 1) * h])
                _native.check(lib.afd_lstm_cell_backward(
                    _native.ptr(pre[t]), _native.ptr(c_all[t]),
                    _native.ptr(c_all[tprev]) if tprev is not None else None, _native.ptr(dh), h,
                    _native.ptr(dc), _native.ptr(dpre[t]), bsz, h, _native.stream_ptr()),
                    "afd_lstm_cell_backward")
                if tprev is not None:
                    hprev[t].copy_(out[tprev, :, direction * h:(direction + 1) * h])
                    gemm_nt(dpre[t], wh_t, None, out=dh)  # gradient reaching h_{t_prev}
            dpre2 = dpre.view(steps * bsz, 4 * h)
            dpre2_t = dpre2.t().contiguous()  # [4H, T*B]
            gemm_nt(dpre2, wi.t().contiguous(), None, out=dx, accumulate=True)       # dx += dpre @ W_ih
            dwi = gemm_nt(dpre2_t, xt.t().contiguous())                              # dpre^T @ x
            dwh = gemm_nt(dpre2_t, hprev.view(steps * bsz, h).t().contiguous())      # dpre^T @ h_prev
            db = dpre2.sum(0)
            grads += [dwi, dwh, db, db.clone()]
        dx = dx.view(steps, bsz, d).permute(1, 0, 2).contiguous()
        return (dx, *grads)


class BLSTMLayer(nn.Module):
    """Bi-directional LSTM wrapper (reference models.py:212-237)."""

    def __init__(self, input_dim: int, output_dim: int) -> None:
        super().__init__()
        if output_dim % 2 != 0:
            raise ValueError("BLSTMLayer expects an even layer size")
        self.l_blstm = nn.LSTM(input_dim, output_dim // 2, bidirectional=True)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        m = self.l_blstm
        return _BLSTM.apply(ops._f32c(x), m.weight_ih_l0, m.weight_hh_l0, m.bias_ih_l0, m.bias_hh_l0,
                            m.weight_ih_l0_reverse, m.weight_hh_l0_reverse, m.bias_ih_l0_reverse,
                            m.bias_hh_l0_reverse)


class _Bf16Plan:
    """Weights of the bf16-storage evaluation forward (csrc/lcnn_nhwc.hip), prepared once per model state: per
    convolution the bf16 rows in matrix-tile order with the evaluation-mode BatchNorm behind it folded in, and the
    first BLSTM layer's input weights with their columns permuted to the channels-last feature order."""

    def __init__(self, net: "LCNN") -> None:
        lib = _native.load()
        self.key = net._bf16_key()
        self.convs = []
        dev = net.fc.weight.device
        for conv_i, pooled, bn_i in net._plan:
            conv = net.lcnn[conv_i]
            cout, cin, k, _ = conv.weight.shape
            buf = torch.empty(lib.afd_lcnn_prep_bytes(cin, cout, k), dtype=torch.uint8, device=dev)
            bn = net.lcnn[bn_i] if bn_i is not None else None
            mean = ops._f32c(bn.running_mean) if bn is not None else None
            var = ops._f32c(bn.running_var) if bn is not None else None
            _native.check(lib.afd_lcnn_prep_conv_bf16(
                _native.ptr(ops._f32c(conv.weight.detach())), _native.ptr(ops._f32c(conv.bias.detach())),
                _native.ptr(mean), _native.ptr(var), float(bn.eps) if bn is not None else 0.0, _native.ptr(buf), cin, cout,
                k, _native.stream_ptr()), "afd_lcnn_prep_conv_bf16")
            self.convs.append((buf, cin, cout, k, conv.padding[0], pooled))
        # channels-last features: index w * C + c instead of the reference's c * W + w (models.py:118-119)
        lstm0 = net.lstm[0].l_blstm
        c_last = self.convs[-1][2] // 2
        feat = lstm0.weight_ih_l0.shape[1]
        wlast = feat // c_last
        perm = torch.arange(feat, device=dev).view(c_last, wlast).t().reshape(-1)
        self.wih0 = {sfx: ops._f32c(getattr(lstm0, "weight_ih_l0" + sfx).detach()[:, perm]) for sfx in ("", "_reverse")}
        # recurrent weights as bf16, once
        self.whh = []
        for layer in net.lstm:
            d = {}
            for sfx in ("", "_reverse"):
                w = ops._f32c(getattr(layer.l_blstm, "weight_hh_l0" + sfx).detach())
                wb = torch.empty(w.shape, dtype=torch.bfloat16, device=dev)
                _native.check(lib.afd_f32_to_bf16(_native.ptr(w), _native.ptr(wb), w.numel(), _native.stream_ptr()),
                              "afd_f32_to_bf16")
                d[sfx] = wb
                if w.shape[1] % 16 == 0 and w.shape[1] <= 256:
                    d["frag" + sfx] = lstm_fragment_order(wb)  # what the one-launch layer kernel reads
            self.whh.append(d)
        # b_ih + b_hh per layer and direction, once (two launches per layer and step otherwise)
        self.bias = [{sfx: (getattr(layer.l_blstm, "bias_ih_l0" + sfx) + getattr(layer.l_blstm, "bias_hh_l0" + sfx)).detach()
                      for sfx in ("", "_reverse")} for layer in net.lstm]


class LCNN(nn.Module):
    """Light CNN + 2 x BLSTM + Linear (reference models.py:68-131)."""

    def __init__(self, classes: int = 2, in_channels: int = 1, lstm_channels: int = 256,
                 precision: str = "fp32") -> None:
        super().__init__()
        if precision not in ("fp32", "bf16"):
            raise ValueError("precision must be 'fp32' or 'bf16'")
        # "bf16": matrix products of the evaluation forward on the bf16 matrix cores (BASELINE configs[4]);
        # parameters, BatchNorm, max-feature-map, the LSTM cell and training stay fp32
        self.precision = precision
        # (cin, cout, k, pad, pooled, bn channels or 0)
        spec = ((in_channels, 64, 5, 2, True, 0), (32, 64, 1, 0, False, 32), (32, 96, 3, 1, True, 48),
                (48, 96, 1, 0, False, 48), (48, 128, 3, 1, True, 0), (64, 128, 1, 0, False, 64),
                (64, 64, 3, 1, False, 32), (32, 64, 1, 0, False, 32), (32, 64, 3, 1, True, 0))
        layers = []
        self._plan = []  # (conv idx, pooled, bn idx or None)
        for cin, cout, k, pad, pooled, bn in spec:
            conv_i = len(layers)
            layers += [nn.Conv2d(cin, cout, k, 1, padding=pad), MaxFeatureMap2D()]
            if pooled:
                layers.append(nn.MaxPool2d(2, 2))
            bn_i = None
            if bn:
                bn_i = len(layers)
                layers.append(nn.BatchNorm2d(bn, affine=False))
            self._plan.append((conv_i, pooled, bn_i))
        layers.append(nn.Dropout(0.7))
        self.lcnn = nn.Sequential(*layers)
        hid = (lstm_channels // 16) * 32
        self.lstm = nn.Sequential(BLSTMLayer(hid, hid), BLSTMLayer(hid, hid))
        self.fc = nn.Linear(hid, classes)
        self.sync_bn = True

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        h = x.permute(0, 1, 3, 2)
        if not h.is_contiguous():
            h = ops.transpose_contiguous(x.contiguous())
        net = self.lcnn
        bf16 = self.precision == "bf16"
        if bf16 and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise RuntimeError("LCNN(precision='bf16') is the evaluation path: call it under torch.no_grad() "
                               "(training runs in fp32)")
        if bf16:
            return self._forward_bf16(x)
        for conv_i, pooled, bn_i in self._plan:
            conv = net[conv_i]
            if bf16:
                # convolution + max-feature-map in one launch: the full-width output is never written
                h = conv2d_bf16(h, conv.weight, conv.bias, conv.padding[0], mfm=True)
            else:
                h = ops.conv2d(h, conv.weight, conv.bias, conv.padding[0], conv.dilation[0])
                h = max_feature_map(h)
            if pooled:
                h = ops.prelu_maxpool2x2(h, None)
            if bn_i is not None:
                h = ops.batch_norm(h, net[bn_i], None, self.sync_bn)
        h = ops.dropout_permute(h, net[-1].p, self.training)  # [B, T', C, W']
        h = h.reshape(h.shape[0], h.shape[1], -1)
        for layer in self.lstm:
            h = blstm_forward_bf16(ops._f32c(h), layer.l_blstm) if bf16 else layer(h)
        return ops.linear_mean(h, self.fc.weight, self.fc.bias)

    def _bf16_key(self):
        return tuple((p.data_ptr(), p._version) for p in self.parameters()) + tuple(
            (b.data_ptr(), b._version) for b in self.buffers())

    def _forward_bf16(self, x: torch.Tensor) -> torch.Tensor:
        """Evaluation forward with bf16 storage: channels-last bf16 activations between the layers, BatchNorm
        (evaluation mode) folded into the convolutions, bf16 matrix products with fp32 accumulation."""
        if self.training:
            raise RuntimeError("LCNN(precision='bf16') is the evaluation path (call .eval()): BatchNorm is folded "
                               "into the convolutions from its running statistics")
        lib = _native.load()
        plan = getattr(self, "_bf16_plan", None)
        if plan is None or plan.key != self._bf16_key():
            plan = self._bf16_plan = _Bf16Plan(self)
        n = x.shape[0]
        if x.shape[1] != 1:
            raise RuntimeError("LCNN(precision='bf16'): one input channel")
        # the reference permutes [B, 1, F, T] -> [B, 1, T, F]: a single-channel image is its own channels-last form
        img = ops.transpose_contiguous(ops._f32c(x)) if x.is_contiguous() else ops._f32c(x.permute(0, 1, 3, 2))
        h, w = img.shape[2], img.shape[3]
        cur = img
        last = len(plan.convs) - 1
        for i, (buf, cin, cout, k, pad, pooled) in enumerate(plan.convs):
            ho, wo = h + 2 * pad - (k - 1), w + 2 * pad - (k - 1)
            # the MaxPool2d(2, 2) behind a layer runs in that layer's epilogue (pool = 1; 2 = fp32 out for the tensor that
            # feeds the BLSTM layers): the pre-pool tensor is never written
            f32 = pooled and i == last
            pool = (2 if f32 else 1) if pooled else 0
            if pooled:
                ho, wo = ho // 2, wo // 2
            y = torch.empty((n, ho, wo, cout // 2), dtype=torch.float32 if f32 else torch.bfloat16, device=x.device)
            if i == 0:
                if pool == 2:
                    raise RuntimeError("LCNN(precision='bf16'): the first layer cannot be the last")
                _native.check(lib.afd_lcnn_conv1_nhwc_bf16(_native.ptr(cur), _native.ptr(buf), _native.ptr(y), n, h, w, cout,
                                                           k, pad, pool, _native.stream_ptr()), "afd_lcnn_conv1_nhwc_bf16")
            else:
                _native.check(lib.afd_lcnn_conv_nhwc_bf16(_native.ptr(cur), _native.ptr(buf), _native.ptr(y), n, h, w, cin,
                                                          cout, k, pad, pool, _native.stream_ptr()), "afd_lcnn_conv_nhwc_bf16")
            cur, h, w = y, ho, wo
        seq = cur.reshape(n, h, -1)  # [B, T', W' C] fp32 (dropout is the identity in evaluation mode)
        for li, layer in enumerate(self.lstm):
            seq = blstm_forward_bf16(seq, layer.l_blstm, plan.wih0 if li == 0 else None, plan.whh[li], plan.bias[li])
        return ops.linear_mean(seq, self.fc.weight, self.fc.bias)

    def get_name(self) -> str:
        return "LCNN"
