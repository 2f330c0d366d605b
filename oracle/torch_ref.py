"""ORACLE (test infrastructure only) -- PyTorch-CPU restatement of the reference hot path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this module.  It is what BASELINE.md section 3 calls the CPU baseline
("port"): the reference's algorithm, op for op, on plain torch CPU tensors.

Restated reference symbols:

* ptwt analysis step + packet tree + per-node Welford + stack + log-power + sign
  channel: ``src/audiofakedetect/wavelet_math.py:167-263`` (ptwt: third party,
  unpinned, algorithm restated -- see oracle/wpt_oracle.py header).
* ``WelfordEstimator``: ``src/audiofakedetect/data_loader.py:27-71``.
* ``STFTLayer``: ``wavelet_math.py:25-68``; torchaudio ``Spectrogram`` defaults are
  restated through ``torch.stft`` (periodic Hann, centre, reflect, one-sided).
* ``torchvision.transforms.Normalize`` with scalar statistics: ``wavelet_math.py:380-382``.
* ``DCNN``: ``src/audiofakedetect/models.py:240-313`` (SyncBatchNorm without a
  process group == BatchNorm2d).
* ``LCNN`` / ``MaxFeatureMap2D`` / ``BLSTMLayer``: ``models.py:68-131,161-237``.
* the train step: ``src/audiofakedetect/train_classifier.py:945-995``.
"""

from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from .wpt_oracle import dec_hi_from_lo, graycode_paths


class Welford:
    """Running mean / M2 over every axis but the last (data_loader.py:41-71)."""

    def __init__(self) -> None:
        self.count = None

    def update(self, vals: torch.Tensor) -> None:
        if self.count is None:
            self.axes = tuple(range(vals.dim() - 1))
            self.count = torch.zeros(1, dtype=torch.float32)
            self.mean = torch.zeros(vals.shape[-1], dtype=torch.float32)
            self.m2 = torch.zeros(vals.shape[-1], dtype=torch.float32)
        n_new = 1
        for s in vals.shape[:-1]:
            n_new *= s
        self.count += n_new
        delta = vals - self.mean
        self.mean += torch.sum(delta / self.count, self.axes)
        delta2 = vals - self.mean
        self.m2 += torch.sum(delta * delta2, self.axes)

    def finalize(self) -> Tuple[torch.Tensor, torch.Tensor]:
        return self.mean, torch.sqrt(self.m2 / self.count)


def analysis_step_torch(x: torch.Tensor, filt: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """x [B, n] -> (cA, cD) [B, n'] through F.pad(reflect) + F.conv1d(stride 2)."""
    filt_len = filt.shape[-1]
    n = x.shape[-1]
    padl = filt_len - 2
    padr = filt_len - 2 + (n % 2)
    xp = F.pad(x.unsqueeze(1), (padl, padr), mode="reflect")
    res = F.conv1d(xp, filt, stride=2)
    return res[:, 0], res[:, 1]


def packets_torch(
    x: torch.Tensor,
    dec_lo: Sequence[float],
    level: int,
    log_scale: bool = False,
    loss_less: bool = False,
    power: float = 2.0,
    compute_welford: bool = True,
    block_norm: bool = False,
    welford_dict: Optional[Dict[str, Welford]] = None,
    per_node: bool = True,
) -> Tuple[torch.Tensor, Dict[str, Welford]]:
    """The reference's packet path on CPU; returns logical [B, C, P, T].

    per_node=True runs it the way the reference does (one pad + conv1d per node, one
    Welford update per node); per_node=False folds a level's nodes into the batch axis
    (same arithmetic, far fewer calls) for the deep-level tests.
    """
    if x.dim() == 3:
        x = x[:, 0, :]
    if not per_node:
        return _packets_torch_levelwise(x, dec_lo, level, log_scale, loss_less, power), {}
    lo = torch.tensor(list(dec_lo), dtype=x.dtype)
    hi = torch.tensor(dec_hi_from_lo(dec_lo), dtype=x.dtype)
    filt = torch.stack([lo.flip(0), hi.flip(0)], 0).unsqueeze(1)
    nodes = {"": x}
    for _ in range(level):
        nxt = {}
        for path, data in nodes.items():
            ca, cd = analysis_step_torch(data, filt)
            nxt[path + "a"] = ca
            nxt[path + "d"] = cd
        nodes = nxt
    if welford_dict is None:
        welford_dict = {}
    plist = []
    for path in graycode_paths(level):
        node = nodes[path]
        if compute_welford:
            if path not in welford_dict:
                welford_dict[path] = Welford()
            welford_dict[path].update(node.unsqueeze(-1))
        if block_norm:
            node = node / torch.max(torch.abs(node))
        plist.append(node)
    wp = torch.stack(plist, dim=-1)  # [B, T, P]
    if log_scale:
        wlog = torch.log(torch.abs(wp).pow(power) + 1e-12)
        if loss_less:
            sign = ((wp < 0).type(torch.float32) * (-1) + 0.5) * 2
            wp = torch.stack([wlog, sign], 1)
        else:
            wp = wlog.unsqueeze(1)
    else:
        wp = wp.unsqueeze(1)
    return wp.permute(0, 1, 3, 2), welford_dict


def _packets_torch_levelwise(x, dec_lo, level, log_scale, loss_less, power):
    lo = torch.tensor(list(dec_lo), dtype=x.dtype)
    hi = torch.tensor(dec_hi_from_lo(dec_lo), dtype=x.dtype)
    filt = torch.stack([lo.flip(0), hi.flip(0)], 0).unsqueeze(1)
    b = x.shape[0]
    cur = x.unsqueeze(1)  # [B, nodes, n] in path order
    for _ in range(level):
        nodes, n = cur.shape[1], cur.shape[2]
        ca, cd = analysis_step_torch(cur.reshape(b * nodes, n), filt)
        cur = torch.stack([ca.reshape(b, nodes, -1), cd.reshape(b, nodes, -1)], 2)
        cur = cur.reshape(b, 2 * nodes, -1)
    f = torch.arange(1 << level)
    wp = cur[:, f ^ (f >> 1), :].permute(0, 2, 1)  # [B, T, P]
    if log_scale:
        wlog = torch.log(torch.abs(wp).pow(power) + 1e-12)
        if loss_less:
            sign = ((wp < 0).type(torch.float32) * (-1) + 0.5) * 2
            wp = torch.stack([wlog, sign], 1)
        else:
            wp = wlog.unsqueeze(1)
    else:
        wp = wp.unsqueeze(1)
    return wp.permute(0, 1, 3, 2)


def stft_torch(
    x: torch.Tensor,
    n_fft: int = 511,
    hop_length: int = 220,
    log_scale: bool = False,
    power: float = 2.0,
) -> torch.Tensor:
    """torchaudio ``Spectrogram(n_fft, hop_length, power)`` (+ log) via torch.stft."""
    shape = x.shape
    flat = x.reshape(-1, shape[-1])
    win = torch.hann_window(n_fft, periodic=True, dtype=x.dtype)
    spec = torch.stft(
        flat,
        n_fft=n_fft,
        hop_length=hop_length,
        win_length=n_fft,
        window=win,
        center=True,
        pad_mode="reflect",
        normalized=False,
        onesided=True,
        return_complex=True,
    )
    spec = spec.abs().pow(power) if power != 2.0 else (spec.real**2 + spec.imag**2)
    spec = spec.reshape(shape[:-1] + spec.shape[-2:])
    if log_scale:
        spec = torch.log(spec + 1e-12)
    return spec


def normalize_torch(t: torch.Tensor, mean: float, std: float) -> torch.Tensor:
    return (t - mean) / std


DCNN_DEFAULTS = dict(
    ochannels1=64, ochannels2=64, ochannels3=96, ochannels4=128, ochannels5=32,
    kernel1=3, dropout_cnn=0.6, dropout_lstm=0.2, time_dim_add=0, flattend_size=320,
)


class DCNNRef(nn.Module):
    """Plain-torch DCNN with the reference's layer order and state_dict key names."""

    def __init__(self, input_dim: Sequence[int], **kw) -> None:
        super().__init__()
        cfg = dict(DCNN_DEFAULTS)
        cfg.update(kw)
        chans = [input_dim[1], cfg["ochannels1"], cfg["ochannels2"], cfg["ochannels3"],
                 cfg["ochannels4"], cfg["ochannels5"], 64]
        # (kernel, padding, pool after activation, batch-norm after)
        spec = [
            (cfg["kernel1"], 2, True, True),
            (1, 0, False, True),
            (3, 1, True, True),
            (3, 1, False, True),
            (3, 1, False, True),
            (3, 1, True, False),
        ]
        layers: List[nn.Module] = []
        for idx, (k, p, pool, bn) in enumerate(spec):
            layers.append(nn.Conv2d(chans[idx], chans[idx + 1], k, stride=1, padding=p))
            layers.append(nn.PReLU())
            if pool:
                layers.append(nn.MaxPool2d(2, 2))
            if bn:
                layers.append(nn.BatchNorm2d(chans[idx + 1], affine=False))
        layers.append(nn.Dropout(cfg["dropout_cnn"]))
        self.cnn = nn.Sequential(*layers)
        tdim = input_dim[-1] // 8 + cfg["time_dim_add"]
        dil: List[nn.Module] = []
        for k, p, d in ((3, 1, 1), (5, 2, 2), (7, 2, 4)):
            dil += [nn.BatchNorm2d(tdim, affine=True),
                    nn.Conv2d(tdim, tdim, k, 1, padding=p, dilation=d), nn.PReLU()]
        dil.append(nn.Dropout(cfg["dropout_lstm"]))
        self.dil_conv = nn.Sequential(*dil)
        self.fc = nn.Sequential(nn.Flatten(2), nn.Linear(cfg["flattend_size"], 2))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = self.cnn(x.permute(0, 1, 3, 2))
        x = x.permute(0, 2, 1, 3).contiguous()
        x = self.dil_conv(x)
        return self.fc(x).mean(1)


class _MFM(nn.Module):
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        b, c = x.shape[:2]
        return x.reshape(b, 2, c // 2, *x.shape[2:]).max(1)[0]


class _BLSTM(nn.Module):
    def __init__(self, din: int, dout: int) -> None:
        super().__init__()
        self.l_blstm = nn.LSTM(din, dout // 2, bidirectional=True)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        y, _ = self.l_blstm(x.permute(1, 0, 2))
        return y.permute(1, 0, 2)


class LCNNRef(nn.Module):
    """Plain-torch LCNN with the reference's layer order and state_dict key names."""

    def __init__(self, classes: int = 2, in_channels: int = 1, lstm_channels: int = 256) -> None:
        super().__init__()
        # (cin, cout, k, pad, pool, bn-channels or 0)
        spec = [
            (in_channels, 64, 5, 2, True, 0),
            (32, 64, 1, 0, False, 32),
            (32, 96, 3, 1, True, 48),
            (48, 96, 1, 0, False, 48),
            (48, 128, 3, 1, True, 0),
            (64, 128, 1, 0, False, 64),
            (64, 64, 3, 1, False, 32),
            (32, 64, 1, 0, False, 32),
            (32, 64, 3, 1, True, 0),
        ]
        layers: List[nn.Module] = []
        for cin, cout, k, p, pool, bn in spec:
            layers += [nn.Conv2d(cin, cout, k, 1, padding=p), _MFM()]
            if pool:
                layers.append(nn.MaxPool2d(2, 2))
            if bn:
                layers.append(nn.BatchNorm2d(bn, affine=False))
        layers.append(nn.Dropout(0.7))
        self.lcnn = nn.Sequential(*layers)
        hid = (lstm_channels // 16) * 32
        self.lstm = nn.Sequential(_BLSTM(hid, hid), _BLSTM(hid, hid))
        self.fc = nn.Linear(hid, classes)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = self.lcnn(x.permute(0, 1, 3, 2))
        x = x.permute(0, 2, 1, 3).contiguous()
        x = self.lstm(x.view(x.shape[0], x.shape[1], -1))
        return self.fc(x).mean(1)


def train_step_torch(model, optimizer, feats: torch.Tensor, labels: torch.Tensor):
    """train_classifier.py:964-986 without the front end: zero_grad, fwd, CE, bwd, step."""
    optimizer.zero_grad()
    out = model(feats)
    loss = F.cross_entropy(out, labels)
    loss.backward()
    optimizer.step()
    return out.detach(), loss.detach()
