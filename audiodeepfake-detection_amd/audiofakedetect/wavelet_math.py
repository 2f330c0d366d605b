"""Transform plugin of the hot path: wavelet packets / STFT -> normalise (MI355X host side).

Keeps the public surface of the reference's ``src/audiofakedetect/wavelet_math.py``
(``Packets`` :223-263, ``STFTLayer`` :25-68, ``compute_pytorch_packet_representation``
:167-220, ``get_transforms`` :266-384, ``calc_normalization`` :387-452); the arithmetic is
one HIP launch per batch in ``libafd_hip.so`` (``afd_wpt_forward`` / ``afd_stft_forward``).
There is no CPU path: CPU inputs are computed on the GPU and the result is returned on the
input's device; a missing library or GPU raises.
"""

from __future__ import annotations

import os
import pickle
from collections.abc import Mapping
from math import log
from typing import Optional, Tuple

import numpy as np
import torch

from . import _native
from .data_loader import WelfordEstimator, get_costum_dataset
from .utils import DotDict
from .wavelets import Wavelet

_WPT_LOG, _WPT_SIGN, _WPT_NORM = 1, 2, 4
_STFT_LOG, _STFT_NORM = 1, 4


_ws_cache: dict = {}


def _workspace(nbytes: int, device) -> torch.Tensor:
    """Grow-only device scratch for the level-8 node hand-off of deep transforms."""
    buf = _ws_cache.get(device.index)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
        _ws_cache[device.index] = buf
    return buf


def _as_frames(x: torch.Tensor) -> torch.Tensor:
    """[B,1,N] / [B,N] / [1,N] / [N] -> contiguous f32 [B,N] on the GPU."""
    _native.require_gpu()
    if x.dim() == 3:
        if x.shape[1] != 1:
            raise ValueError(f"expected mono frames [B,1,N], got {tuple(x.shape)}")
        x = x[:, 0, :]
    elif x.dim() == 1:
        x = x.unsqueeze(0)
    elif x.dim() != 2:
        raise ValueError(f"expected [B,1,N] or [B,N] frames, got {tuple(x.shape)}")
    if not x.is_cuda:
        x = x.cuda(non_blocking=True)
    return x.to(torch.float32).contiguous()


def wpt_forward(
    frames: torch.Tensor,
    wavelet: Wavelet,
    max_lev: int,
    log_scale: bool = False,
    loss_less: bool = False,
    power: float = 2.0,
    mean=None,
    std=None,
) -> torch.Tensor:
    """Launch the fused packet transform; returns memory-order [B, C, T, P].

    ``mean`` / ``std``: a scalar, or one value per channel (coefficients, sign)."""
    lib = _native.load()
    x = _as_frames(frames)
    b, n = x.shape
    length = wavelet.dec_len
    t_len = lib.afd_wpt_out_len(n, length, max_lev)
    if t_len <= 0:
        raise ValueError(f"wavelet packets: level {max_lev} is not defined for N={n}, L={length}")
    flags = 0
    nch = 1
    if log_scale:
        flags |= _WPT_LOG
        if loss_less:
            flags |= _WPT_SIGN
            nch = 2
    if mean is not None:
        flags |= _WPT_NORM
    out = torch.empty((b, nch, t_len, 1 << max_lev), dtype=torch.float32, device=x.device)
    ws_bytes = lib.afd_wpt_workspace_bytes(b, n, length, max_lev)
    ws = _workspace(ws_bytes, x.device) if ws_bytes else None
    rc = lib.afd_wpt_forward(
        _native.ptr(x), b, n, _native.float_array(wavelet.dec_lo),
        _native.float_array(wavelet.dec_hi), length, max_lev, flags, float(power), 1e-12,
        *_channel_stats(mean, std), _native.ptr(out),
        _native.ptr(ws), ws_bytes, _native.stream_ptr(),
    )
    if rc == _ERR_UNSUPPORTED and n > _LDS_FRAME:
        # The kernels keep a frame's packet tree in LDS: frames beyond about 27 000 samples (1.2 s at 22 050 Hz) do not fit.
        # Split level 1 off from global memory (afd_wpt_analysis_step) and transform the two children -- frames of half the
        # length -- on their own (recursively: a 4 s frame splits twice).  In frequency order the packets of the
        # approximation child come first and those of the detail child follow REVERSED (the Gray-code rule of
        # ptwt's get_level: order_{k+1} = [a + p for p in order_k] + [d + p for p in reversed(order_k)]).
        raw = _wpt_long_raw(x, wavelet, max_lev)
        if flags == 0:
            return raw
        _native.check(lib.afd_packet_block_norm(
            _native.ptr(raw), b, t_len, 1 << max_lev, None, flags, float(power), 1e-12, *_channel_stats(mean, std),
            _native.ptr(out), _native.stream_ptr()), "afd_packet_block_norm")
        return out
    _native.check(rc, "afd_wpt_forward")
    return out


_ERR_UNSUPPORTED = -3
_LDS_FRAME = 20000  # (frames at least this long may exceed what the LDS-resident kernels hold; shorter ones never do)


def _wpt_long_raw(x: torch.Tensor, wavelet: Wavelet, level: int) -> torch.Tensor:
    """Raw packet coefficients [B, 1, T, 2^level] of frames too long for the LDS-resident kernels: one analysis step
    from global memory, then both children as frames of their own."""
    lib = _native.load()
    b, n = x.shape
    length = wavelet.dec_len
    n1 = lib.afd_wpt_out_len(n, length, 1)
    ca = torch.empty((b, n1), dtype=torch.float32, device=x.device)
    cd = torch.empty((b, n1), dtype=torch.float32, device=x.device)
    lo, hi = _native.float_array(wavelet.dec_lo), _native.float_array(wavelet.dec_hi)
    step = 65535  # (grid limit of the step kernel's batch axis)
    for b0 in range(0, b, step):
        nb = min(step, b - b0)
        _native.check(lib.afd_wpt_analysis_step(_native.ptr(x[b0:]), nb, n, lo, hi, length, _native.ptr(ca[b0:]),
                                                _native.ptr(cd[b0:]), _native.stream_ptr()), "afd_wpt_analysis_step")
    if level == 1:
        return torch.stack((ca, cd), dim=-1).unsqueeze(1)
    pa = wpt_forward(ca, wavelet, level - 1)   # [B, 1, T, P/2] each, in their own frequency order
    pd = wpt_forward(cd, wavelet, level - 1)
    return torch.cat((pa, pd.flip(-1)), dim=-1)


def _channel_stats(mean, std) -> Tuple[float, float, float, float]:
    """(mean, std, sign_mean, sign_std) of the C-ABI from a scalar or per-channel pair."""
    def two(v, default):
        if v is None:
            return default, default
        if isinstance(v, (tuple, list)):
            return float(v[0]), float(v[-1])
        return float(v), float(v)

    m0, m1 = two(mean, 0.0)
    s0, s1 = two(std, 1.0)
    return m0, s0, m1, s1


def graycode_keys(level: int) -> list:
    """Node paths of a level in frequency order (ptwt ``WaveletPacket.get_level``)."""
    order = ["a", "d"]
    for _ in range(level - 1):
        order = ["a" + path for path in order] + ["d" + path for path in order[::-1]]
    return order


class _NodeEstimator:
    """One node's view onto a `PacketWelford` (the reference keeps a WelfordEstimator per node)."""

    def __init__(self, owner: "PacketWelford", index: int) -> None:
        self.owner, self.index = owner, index

    def finalize(self) -> Tuple[torch.Tensor, torch.Tensor]:
        mean, std = self.owner.moments()
        return mean[self.index:self.index + 1], std[self.index:self.index + 1]


class PacketWelford(Mapping):
    """The reference's ``block_norm_dict`` (node path -> running estimator,
    wavelet_math.py:188-200) with all of a level's estimators in one device buffer: every
    update is one fused reduction over the raw coefficients (``afd_packet_stats``) in place
    of 2^level WelfordEstimator updates.  ``finalize()`` gives the dict the reference stores,
    ``{path: {"mean": [1], "std": [1]}}`` with population statistics ``sqrt(M2 / count)``."""

    def __init__(self, level: int, device) -> None:
        self.level = level
        self.paths = graycode_keys(level)
        self._index = {k: i for i, k in enumerate(self.paths)}
        self.sums = torch.zeros((2, 1 << level), dtype=torch.float64, device=device)
        self.count = 0

    def __getitem__(self, key: str) -> _NodeEstimator:
        return _NodeEstimator(self, self._index[key])

    def __iter__(self):
        return iter(self.paths)

    def __len__(self) -> int:
        return len(self.paths)

    def moments(self) -> Tuple[torch.Tensor, torch.Tensor]:
        if self.count == 0:
            raise RuntimeError("PacketWelford: no data seen")
        mean = self.sums[0] / self.count
        var = (self.sums[1] / self.count - mean * mean).clamp_min(0.0)
        return mean.float(), var.sqrt().float()

    def finalize(self) -> dict:
        mean, std = self.moments()
        return {k: {"mean": mean[i:i + 1], "std": std[i:i + 1]} for i, k in enumerate(self.paths)}


def packet_block_norm(
    frames: torch.Tensor,
    wavelet: Wavelet,
    max_lev: int,
    log_scale: bool,
    loss_less: bool,
    power: float,
    block_norm: bool,
    estimators: Optional[PacketWelford],
    mean=None,
    std=None,
) -> torch.Tensor:
    """Packet image with per-node statistics / max normalisation (wavelet_math.py:194-218).

    Raw coefficients -> one statistics pass (per-packet sums into `estimators`, max |v| of this
    batch) -> divide by the batch's per-node maximum and apply the log / sign / normalise
    epilogue.  Memory order [B, C, T, P] like `wpt_forward`."""
    lib = _native.load()
    raw = wpt_forward(frames, wavelet, max_lev)
    b, _, t_len, npk = raw.shape
    if estimators is not None and len(estimators) != npk:
        raise ValueError("block_norm_dict belongs to another decomposition level")
    sums = estimators.sums if estimators is not None else torch.zeros(
        (2, npk), dtype=torch.float64, device=raw.device)
    absmax = torch.zeros(npk, dtype=torch.float32, device=raw.device)
    _native.check(lib.afd_packet_stats(_native.ptr(raw), b * t_len, npk, _native.ptr(sums),
                                       _native.ptr(absmax), _native.stream_ptr()), "afd_packet_stats")
    if estimators is not None:
        estimators.count += b * t_len
    flags = 0
    nch = 1
    if log_scale:
        flags |= _WPT_LOG
        if loss_less:
            flags |= _WPT_SIGN
            nch = 2
    if mean is not None:
        flags |= _WPT_NORM
    if flags == 0 and not block_norm:
        return raw
    out = torch.empty((b, nch, t_len, npk), dtype=torch.float32, device=raw.device)
    _native.check(lib.afd_packet_block_norm(
        _native.ptr(raw), b, t_len, npk, _native.ptr(absmax) if block_norm else None, flags,
        float(power), 1e-12, *_channel_stats(mean, std),
        _native.ptr(out), _native.stream_ptr()), "afd_packet_block_norm")
    return out


def compute_pytorch_packet_representation(
    pt_data: torch.Tensor,
    wavelet: Wavelet,
    max_lev: int = 8,
    log_scale: bool = False,
    loss_less: bool = False,
    power: float = 2.0,
    block_norm: bool = False,
    compute_welford: bool = False,
    block_norm_dict=None,
) -> Tuple[torch.Tensor, dict]:
    """Packet image [B, C, T, P] (+ the per-node statistics of the reference API).

    The reference updates one Welford estimator per node on every call
    (wavelet_math.py:194-200) although the result is only read when ``--block-norm`` is set;
    here the estimators run when the caller hands in a `PacketWelford` as ``block_norm_dict``
    (``calc_normalization`` does under ``--block-norm``), otherwise the dict is passed through.
    """
    estimators = block_norm_dict if (compute_welford and isinstance(block_norm_dict, PacketWelford)) else None
    if block_norm_dict is None:
        block_norm_dict = {}
    if block_norm or estimators is not None:
        out = packet_block_norm(pt_data, wavelet, max_lev, log_scale, loss_less, power,
                                block_norm, estimators)
    else:
        out = wpt_forward(pt_data, wavelet, max_lev, log_scale, loss_less, power)
    return (out if pt_data.is_cuda else out.cpu()), block_norm_dict


class Packets(torch.nn.Module):
    """Wavelet-packet representation as a module (reference wavelet_math.py:223-263)."""

    def __init__(
        self,
        wavelet_str: str = "sym8",
        max_lev: int = 8,
        log_scale: bool = False,
        loss_less: bool = False,
        power: float = 2.0,
        block_norm: bool = False,
        compute_welford: bool = False,
        block_norm_dict=None,
    ) -> None:
        super().__init__()
        self.wavelet = Wavelet(wavelet_str)
        self.max_lev = max_lev
        self.log_scale = log_scale
        self.loss_less = loss_less
        self.power = power
        self.block_norm = block_norm
        self.compute_welford = compute_welford
        self.block_norm_dict = block_norm_dict
        # set by fuse_normalization(): (mean, std) applied in the kernel epilogue
        self.fused_norm: Optional[tuple] = None  # scalars, or (coefficients, sign) pairs

    def forward(self, pt_data: torch.Tensor) -> Tuple[torch.Tensor, dict]:
        mean, std = self.fused_norm if self.fused_norm is not None else (None, None)
        estimators = self.block_norm_dict if (
            self.compute_welford and isinstance(self.block_norm_dict, PacketWelford)) else None
        if self.block_norm or estimators is not None:
            packets = packet_block_norm(pt_data, self.wavelet, self.max_lev, self.log_scale,
                                        self.loss_less, self.power, self.block_norm, estimators,
                                        mean, std)
        else:
            packets = wpt_forward(pt_data, self.wavelet, self.max_lev, self.log_scale,
                                  self.loss_less, self.power, mean, std)
        bdict = self.block_norm_dict if self.block_norm_dict is not None else {}
        if not pt_data.is_cuda:
            packets = packets.cpu()  # the result lives where the input lives, as in the reference
        # logical [B, C, P, T]; memory stays [B, C, T, P] exactly like the reference's view
        return packets.permute(0, 1, 3, 2), bdict


class STFTLayer(torch.nn.Module):
    """Power spectrogram (reference wavelet_math.py:25-68; torchaudio Spectrogram defaults)."""

    def __init__(self, n_fft: int = 511, hop_length: int = 220, log_offset: float = 1e-12,
                 log_scale: bool = False, power: float = 2.0) -> None:
        super().__init__()
        self.n_fft = n_fft
        self.hop_length = hop_length
        self.log_scale = log_scale
        self.log_offset = log_offset
        self.power = power
        self.block_norm_dict = None
        self.fused_norm: Optional[Tuple[float, float]] = None
        self._basis: Optional[torch.Tensor] = None

    def forward(self, input: torch.Tensor) -> Tuple[torch.Tensor, None]:
        from .stft import stft_forward

        mean, std = self.fused_norm if self.fused_norm is not None else (None, None)
        spec = stft_forward(self, input, mean, std)
        return (spec if input.is_cuda else spec.cpu()), None


class Normalize(torch.nn.Module):
    """``torchvision.transforms.Normalize`` (wavelet_math.py:380-382): ``(t - mean[c]) / std[c]``.

    The statistics are per channel, as ``calc_normalization``'s Welford produces them (one value
    for the usual single-channel features, two with ``--loss-less True``: log-magnitude and sign);
    a single value is applied to every channel, as torchvision broadcasts it."""

    def __init__(self, mean, std) -> None:
        super().__init__()
        self.means = [float(v) for v in torch.as_tensor(mean, dtype=torch.float64).reshape(-1).tolist()]
        self.stds = [float(v) for v in torch.as_tensor(std, dtype=torch.float64).reshape(-1).tolist()]
        if len(self.means) != len(self.stds) and 1 not in (len(self.means), len(self.stds)):
            raise ValueError("mean and std must hold one value per channel")
        if any(s == 0.0 for s in self.stds):
            raise ValueError("std evaluated to zero")
        self.identity = False  # True once the statistics are fused into the transform

    @property
    def mean(self) -> float:
        return self.means[0]

    @property
    def std(self) -> float:
        return self.stds[0]

    def channel_stats(self, channels: int):
        """([mean per channel], [std per channel]) for a `channels`-channel tensor."""
        nstat = max(len(self.means), len(self.stds))
        if nstat not in (1, channels):
            raise ValueError(f"Normalize holds {nstat} channel statistics, the tensor has {channels} channels")
        means = self.means * channels if len(self.means) == 1 else self.means
        stds = self.stds * channels if len(self.stds) == 1 else self.stds
        return means, stds

    def forward(self, t: torch.Tensor) -> torch.Tensor:
        if self.identity:
            return t
        from .ops import normalize_forward

        means, stds = self.channel_stats(t.shape[1] if t.dim() >= 3 else 1)
        if len(set(means)) == 1 and len(set(stds)) == 1:
            out = normalize_forward(t, means[0], stds[0])
        else:
            # per-channel statistics (loss-less features) outside the fused epilogue
            from .ops import normalize_channels_forward

            out = normalize_channels_forward(t, means, stds)
        return out if t.is_cuda else out.cpu()


def fuse_normalization(transforms: torch.nn.Sequential, normalize: torch.nn.Sequential) -> bool:
    """Move the scalar (x - mean) / std into the transform kernel's epilogue.

    Saves the clone + two elementwise passes of the reference's Normalize.  Returns False
    (and changes nothing) when the pair cannot be fused.
    """
    if len(transforms) != 1 or len(normalize) != 1:
        return False
    tr, nm = transforms[0], normalize[0]
    if not isinstance(nm, Normalize) or not hasattr(tr, "fused_norm"):
        return False
    if len(nm.means) == 1 and len(nm.stds) == 1:
        tr.fused_norm = (nm.mean, nm.std)
    else:
        # per-channel statistics: the packet transform's epilogue takes (coefficients, sign) pairs
        if not isinstance(tr, Packets) or max(len(nm.means), len(nm.stds)) != 2 or not tr.loss_less:
            return False
        means, stds = nm.channel_stats(2)
        tr.fused_norm = (tuple(means), tuple(stds))
    nm.identity = True
    return True


def get_transforms(
    args: DotDict,
    features: str,
    device: str,
    normalization: bool,
    pbar: bool = False,
    verbose: bool = True,
) -> Tuple[torch.nn.Sequential, torch.nn.Sequential]:
    """Build the frequency-space transform and the normalisation (wavelet_math.py:266-384)."""
    if features not in (None, "none"):
        raise NotImplementedError("lfcc/delta features are outside the hot path (SURVEY.md #13)")
    log_scale = bool(args.features == "none" and args.log_scale)
    if args.transform == "stft":
        transform: torch.nn.Module = STFTLayer(
            n_fft=args.num_of_scales * 2 - 1, hop_length=args.hop_length,
            log_scale=log_scale, power=args.power,
        )
    elif args.transform == "packets":
        transform = Packets(
            wavelet_str=args.wavelet, max_lev=int(log(args.num_of_scales, 2)),
            log_scale=log_scale, loss_less=False if args.loss_less == "False" else True,
            power=args.power, block_norm_dict=None, block_norm=False, compute_welford=True,
        )
    else:
        raise ValueError(f"unknown transform {args.transform!r}")
    transforms = torch.nn.Sequential(transform)

    loss_less = "_loss_less" if args.loss_less == "True" else ""
    norm_dir = "{}/norms/{}_{}_{}_{}_{}_{}{}_{}_{}secs".format(
        args.log_dir, str(args.data_path).replace("/", "_"), "-".join(args.only_use or []),
        args.transform, args.wavelet, args.num_of_scales, args.power, loss_less,
        args.sample_rate, args.seconds,
    )
    welford_dict: dict = {}
    bn_files = [f"{norm_dir}_mean_std_bn.pt", f"{norm_dir}_mean_std_bn.pkl"]
    if os.path.exists(f"{norm_dir}_mean_std.pkl") and not args.block_norm:
        if verbose:
            print("Loading pre calculated mean and std from file.")
        with open(f"{norm_dir}_mean_std.pkl", "rb") as file:
            mean, std = pickle.load(file)
        mean = torch.from_numpy(np.asarray(mean, dtype=np.float32))
        std = torch.from_numpy(np.asarray(std, dtype=np.float32))
    elif args.block_norm and any(os.path.exists(f) for f in bn_files):
        # the reference writes `_mean_std_bn.pkl` (:447) and looks for `_mean_std_bn.pt` (:356);
        # both names are accepted here, `.pt` is what calc_normalization writes
        if verbose:
            print("Loading pre calculated mean and std from file.")
        welford_dict = torch.load(next(f for f in bn_files if os.path.exists(f)), map_location=device)
    elif normalization:
        if verbose:
            print("computing mean and std values.", flush=True)
        welford_dict, mean, std = calc_normalization(args, pbar, transforms, norm_dir)
    else:
        if verbose:
            print("Using default mean and std.")
        mean = torch.tensor(args.mean if args.mean is not None else 0.0)
        std = torch.tensor(args.std if args.std is not None else 1.0)
    if args.block_norm:
        if args.transform != "packets":
            raise ValueError("--block-norm is defined for the packet transform only")
        # per-node max normalisation inside the transform, identity Normalize (:373-378)
        mean, std = torch.tensor(0.0), torch.tensor(1.0)
        transforms[0].block_norm_dict = welford_dict
        transforms[0].compute_welford = False
        transforms[0].block_norm = True
    normalize = torch.nn.Sequential(Normalize(mean, std))
    return transforms, normalize


def calc_normalization(args: DotDict, pbar: bool, transforms: torch.nn.Sequential,
                       norm_dir: str) -> tuple:
    """Scalar mean / std of the transformed training set (wavelet_math.py:387-452)."""
    dataset = get_costum_dataset(
        data_path=args.data_path, ds_type="train", only_use=args.only_use,
        save_path=args.save_path, limit=args.limit_train[0] if args.limit_train else None,
        file_type=args.file_type, resample_rate=args.sample_rate, seconds=args.seconds,
        synthetic=bool(args.synthetic),
    )
    loader = torch.utils.data.DataLoader(dataset, batch_size=4000, shuffle=False,
                                         num_workers=0)
    from .ops import ScalarMoments

    # single-channel features: the reference's per-channel Welford over [B, T, P, C=1] is a
    # scalar mean / std; two-channel (loss-less) features keep the per-channel estimator
    welford = None
    welford_dict = None
    if args.block_norm:
        # per-node estimators ride along with the transform (:436-439)
        transforms[0].block_norm_dict = PacketWelford(transforms[0].max_lev, "cuda")
        transforms[0].compute_welford = True
    with torch.no_grad():
        for batch in loader:
            feats, welford_dict = transforms(batch["audio"].cuda(non_blocking=True))
            if welford is None:
                welford = ScalarMoments(feats.device) if feats.shape[1] == 1 else WelfordEstimator()
            if isinstance(welford, ScalarMoments):
                welford.update(feats)
            else:
                welford.update(feats.permute(0, 3, 2, 1))
        mean, std = welford.finalize()
    os.makedirs(os.path.dirname(norm_dir), exist_ok=True)
    if args.block_norm:
        welford_dict = welford_dict.finalize()
        torch.save(welford_dict, f"{norm_dir}_mean_std_bn.pt")
    else:
        with open(f"{norm_dir}_mean_std.pkl", "wb") as f:
            pickle.dump([mean.cpu().numpy(), std.cpu().numpy()], f)
    return welford_dict, mean, std
