"""Host side of the STFT front end: basis cache + launch of afd_stft_forward."""

from __future__ import annotations

import ctypes
from typing import Optional

import torch

from . import _native

_basis_cache: dict = {}


def _basis(n_fft: int, device) -> torch.Tensor:
    key = (n_fft, device.index)
    t = _basis_cache.get(key)
    if t is None:
        lib = _native.load()
        rows, cols = ctypes.c_int(), ctypes.c_int()
        _native.check(lib.afd_stft_dims(4 * n_fft, n_fft, 1, None, None, ctypes.byref(rows),
                                        ctypes.byref(cols)), "afd_stft_dims")
        host = torch.empty((rows.value, cols.value), dtype=torch.float32)
        _native.check(lib.afd_stft_basis(n_fft, ctypes.c_void_p(host.data_ptr())), "afd_stft_basis")
        t = host.to(device)
        _basis_cache[key] = t
    return t


def stft_forward(layer, x: torch.Tensor, mean: Optional[float] = None,
                 std: Optional[float] = None) -> torch.Tensor:
    """Power spectrogram of x[..., N] -> [..., F, T] (the reference's STFTLayer output)."""
    _native.require_gpu()
    lib = _native.load()
    lead = x.shape[:-1]
    n = x.shape[-1]
    frames = x.reshape(-1, n)
    if not frames.is_cuda:
        frames = frames.cuda(non_blocking=True)
    frames = frames.to(torch.float32).contiguous()
    b = frames.shape[0]
    f, t = ctypes.c_int(), ctypes.c_int()
    rc = lib.afd_stft_dims(n, layer.n_fft, layer.hop_length, ctypes.byref(f), ctypes.byref(t), None, None)
    if rc != 0:
        raise ValueError(f"stft: n_fft {layer.n_fft} / hop {layer.hop_length} not defined for N={n}")
    out = torch.empty((b, f.value, t.value), dtype=torch.float32, device=frames.device)
    flags = (1 if layer.log_scale else 0) | (4 if mean is not None else 0)
    step = 65535
    for b0 in range(0, b, step):
        nb = min(step, b - b0)
        _native.check(lib.afd_stft_forward(
            _native.ptr(frames[b0:]), nb, n, layer.n_fft, layer.hop_length,
            _native.ptr(_basis(layer.n_fft, frames.device)), flags, float(layer.power),
            float(layer.log_offset), float(mean or 0.0), float(std if std is not None else 1.0),
            _native.ptr(out[b0:]), _native.stream_ptr()), "afd_stft_forward")
    return out.reshape(lead + (f.value, t.value))
