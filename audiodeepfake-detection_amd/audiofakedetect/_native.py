"""ctypes binding of libafd_hip.so (the C ABI declared in include/afd_hip.h).

The library is built in-tree (``audiodeepfake-detection_amd/lib/libafd_hip.so``, see
``__graft_entry__.build``).  There is no CPU fallback: if the library is missing or no
MI355X is visible every entry point raises.
"""

from __future__ import annotations

import ctypes
import os
from typing import Optional

import torch

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# AFD_LIB: another build of the same library (A/B runs of bench.py, tools/ab_*.py); the in-tree build otherwise
LIB_PATH = os.environ.get("AFD_LIB") or os.path.join(_PKG, "lib", "libafd_hip.so")

_lib: Optional[ctypes.CDLL] = None

c_f = ctypes.c_float
c_i = ctypes.c_int
c_u = ctypes.c_uint
c_p = ctypes.c_void_p
c_sz = ctypes.c_size_t
c_l = ctypes.c_longlong
c_ul = ctypes.c_ulonglong

# name -> (restype, argtypes); mirrors include/afd_hip.h
_SIGNATURES = {
    "afd_last_error": (ctypes.c_char_p, []),
    "afd_version": (c_i, []),
    "afd_timing_enable": (c_i, [c_i]),
    "afd_timing_collect": (c_i, [c_i, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(c_l)]
                           + [ctypes.POINTER(ctypes.c_double)] * 3),
    "afd_timing_reset": (c_i, []),
    "afd_wpt_out_len": (c_i, [c_i, c_i, c_i]),
    "afd_wpt_workspace_bytes": (c_sz, [c_i, c_i, c_i, c_i]),
    "afd_wpt_forward": (c_i, [c_p, c_i, c_i, ctypes.POINTER(c_f), ctypes.POINTER(c_f), c_i, c_i,
                              c_u, c_f, c_f, c_f, c_f, c_f, c_f, c_p, c_p, c_sz, c_p]),
    "afd_wpt_analysis_step": (c_i, [c_p, c_i, c_i, ctypes.POINTER(c_f), ctypes.POINTER(c_f), c_i, c_p, c_p, c_p]),
    "afd_wpt_lattice": (c_i, [ctypes.POINTER(c_f), ctypes.POINTER(c_f), c_i] + [ctypes.POINTER(ctypes.c_double)] * 4),
    "afd_rccl_unique_id": (c_i, [c_p]),
    "afd_rccl_init": (c_i, [c_p, c_i, c_i]),
    "afd_rccl_all_reduce_sum": (c_i, [c_p, c_l, c_i, c_p]),
    "afd_rccl_world": (c_i, []),
    "afd_rccl_destroy": (c_i, []),
    "afd_stft_dims": (c_i, [c_i, c_i, c_i] + [ctypes.POINTER(c_i)] * 4),
    "afd_stft_basis": (c_i, [c_i, c_p]),
    "afd_stft_forward": (c_i, [c_p, c_i, c_i, c_i, c_i, c_p, c_u, c_f, c_f, c_f, c_f, c_p, c_p]),
    "afd_conv2d_workspace_bytes": (c_sz, [c_i] * 8),
    "afd_conv2d_forward": (c_i, [c_p, c_p, c_p, c_p] + [c_i] * 8 + [c_p, c_sz, c_p]),
    "afd_conv3x3_prelu_pool_applicable": (c_i, [c_i] * 4),
    "afd_conv3x3_prelu_pool_forward": (c_i, [c_p] * 6 + [c_i] * 5 + [c_p, c_sz, c_p]),
    "afd_conv1x1_forward_stats_workspace_bytes": (c_sz, [c_i]),
    "afd_conv1x1_forward_stats": (c_i, [c_p] * 6 + [c_i, c_i, c_i, c_l, c_p, c_sz, c_p]),
    "afd_conv3x3_backward_data_bnstats_applicable": (c_i, [c_i] * 4),
    "afd_conv3x3_backward_data_bnstats_needs_input": (c_i, [c_i] * 4),
    "afd_conv_weight_dot": (c_i, [c_p, c_p, c_i, c_i, c_i, c_p, c_p]),
    "afd_multi_gather": (c_i, [c_p, c_p, c_p, c_i, c_p, c_p]),
    "afd_conv3x3_pooled_backward_applicable": (c_i, [c_i] * 4),
    "afd_conv3x3_backward_data_bnstats_pooled": (c_i, [c_p] * 5 + [c_i] * 5 + [c_p, c_sz, c_p, c_sz, c_p]),
    "afd_conv3x3_backward_weight_pooled": (c_i, [c_p] * 5 + [c_i] * 5 + [c_p, c_sz, c_p]),
    "afd_prelu_pool_backward_compact": (c_i, [c_p] * 5 + [c_i] + [c_p, c_p] + [c_i] * 3 + [c_p]),
    "afd_conv3x3_backward_data_bnstats_workspace_bytes": (c_sz, [c_i] * 4),
    "afd_conv3x3_backward_data_bnstats": (c_i, [c_p] * 5 + [c_i] * 5 + [c_p, c_sz, c_p, c_sz, c_p]),
    "afd_conv3x3_forward_stats_applicable": (c_i, [c_i] * 5),
    "afd_conv3x3_forward_stats_workspace_bytes": (c_sz, [c_i] * 4),
    "afd_conv3x3_forward_stats": (c_i, [c_p] * 8 + [c_i] * 5 + [c_p, c_sz, c_p, c_sz, c_p]),
    "afd_conv3x3_input_fold_applicable": (c_i, [c_i] * 6),
    "afd_conv3x3_backward_data_bnapply_applicable": (c_i, [c_i] * 5),
    "afd_conv3x3_input_grad_sums": (c_i, [c_p] * 6 + [c_i] * 7 + [c_p]),
    "afd_conv3x3_backward_data_bnapply": (c_i, [c_p] * 9 + [c_i] * 5 + [c_p, c_sz, c_p, c_sz, c_p]),
    "afd_conv3x3_forward_fold": (c_i, [c_p] * 10 + [c_i] * 5 + [c_p, c_sz, c_p, c_sz, c_p]),
    "afd_conv3x3_backward_weight_fold": (c_i, [c_p] * 8 + [c_i] * 7 + [c_p, c_sz, c_p]),
    "afd_bn_fold_forward": (c_i, [c_p] * 6 + [c_i, c_i, c_p]),
    "afd_bn_fold_backward_weights": (c_i, [c_p] * 7 + [c_i, c_i, c_p]),
    "afd_bn_fold_backward_affine": (c_i, [c_p, ctypes.c_double, c_p, c_p, c_p, c_p, c_p, c_i, c_p]),
    "afd_bn_backward_coef": (c_i, [c_p] * 5 + [c_i, c_p]),
    "afd_wav_read_windows": (c_i, [c_p, c_p, c_i, c_i, c_p, c_p, c_i]),
    "afd_pcm16_resample": (c_i, [c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_i, c_p]),
    "afd_conv1x1_bn_backward_data": (c_i, [c_p] * 6 + [c_i, c_i, c_i, c_l, c_p]),
    "afd_conv2d_backward_data": (c_i, [c_p, c_p, c_p] + [c_i] * 8 + [c_p, c_sz, c_p]),
    "afd_conv2d_backward_weight": (c_i, [c_p, c_p, c_p, c_p] + [c_i] * 8 + [c_p, c_sz, c_p]),
    "afd_conv2d_forward_cropped": (c_i, [c_p, c_p, c_p, c_p] + [c_i] * 10 + [c_p, c_sz, c_p]),
    "afd_conv2d_backward_weight_cropped": (c_i, [c_p, c_p, c_p, c_p] + [c_i] * 10 + [c_p, c_sz, c_p]),
    "afd_conv2d_backward_weight_sums": (c_i, [c_p, c_p, c_p, c_p, c_p] + [c_i] * 10 + [c_p, c_sz, c_p]),
    "afd_conv1_pool_workspace_bytes": (c_sz, [c_i] * 5),
    "afd_conv1_pool_stats_workspace_bytes": (c_sz, [c_i] * 5),
    "afd_conv1_pool_stats_applicable": (c_i, [c_i] * 5),
    "afd_conv1_pool_forward": (c_i, [c_p] * 8 + [c_sz] + [c_i] * 5 + [c_p]),
    "afd_conv1_pool_backward": (c_i, [c_p] * 8 + [c_i] * 5 + [c_p, c_sz, c_p]),
    "afd_conv1_pool_backward_affine": (c_i, [c_p] * 10 + [c_i] * 5 + [c_p, c_sz, c_p]),
    "afd_conv1x1_prelu_bn_backward_applicable": (c_i, [c_i, c_i]),
    "afd_conv1x1_prelu_bn_backward_workspace_bytes": (c_sz, [c_i, c_i]),
    "afd_conv1x1_prelu_bn_backward": (c_i, [c_p] * 10 + [c_i, c_i, c_i, c_l, c_p, c_sz, c_p]),
    "afd_moments_accumulate": (c_i, [c_p, c_sz, c_p, c_p]),
    "afd_normalize_forward": (c_i, [c_p, c_p, c_sz, c_f, c_f, c_p]),
    "afd_normalize_channels_forward": (c_i, [c_p, c_p, c_i, c_i, c_sz, ctypes.POINTER(c_f), ctypes.POINTER(c_f), c_p]),
    "afd_packet_stats": (c_i, [c_p, c_l, c_i, c_p, c_p, c_p]),
    "afd_packet_block_norm": (c_i, [c_p, c_i, c_i, c_i, c_p, c_u, c_f, c_f, c_f, c_f, c_f, c_f, c_p, c_p]),
    "afd_transpose_last2": (c_i, [c_p, c_p, c_i, c_i, c_i, c_p]),
    "afd_prelu_dropout_forward": (c_i, [c_p, c_p, c_p, c_sz, c_f, c_ul, c_p]),
    "afd_prelu_dropout_backward": (c_i, [c_p, c_p, c_p, c_p, c_p, c_sz, c_f, c_ul, c_p]),
    "afd_prelu_pool_forward": (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p]),
    "afd_prelu_pool_backward": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p]),
    "afd_prelu_pool_backward_affine": (c_i, [c_p] * 5 + [c_i, c_p, c_p, c_i, c_i, c_i, c_p]),
    "afd_bn_stats": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_p]),
    "afd_bn_apply_forward": (c_i, [c_p] * 7 + [c_i, c_i, c_i, c_p]),
    "afd_bn_backward_stats": (c_i, [c_p] * 6 + [c_i, c_i, c_i, c_p]),
    "afd_bn_backward_apply": (c_i, [c_p] * 10 + [c_i, c_i, c_i, c_p]),
    "afd_bn_backward_apply_sums": (c_i, [c_p] * 11 + [c_i, c_i, c_i, c_p]),
    "afd_bn_finalize": (c_i, [c_p, c_i, ctypes.c_double, c_f, c_f, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "afd_bn_backward_means": (c_i, [c_p, c_i, ctypes.c_double, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "afd_dropout_permute": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_ul, c_i, c_p]),
    "afd_linear_mean_forward": (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p]),
    "afd_linear_mean_backward": (c_i, [c_p] * 6 + [c_i, c_i, c_i, c_i, c_p]),
    "afd_mfm_forward": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_p]),
    "afd_mfm_backward": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_p]),
    "afd_gemm_nt": (c_i, [c_p, c_p, c_p, c_p] + [c_i] * 7 + [c_p]),
    "afd_gemm_nt_bf16": (c_i, [c_p, c_p, c_p, c_p] + [c_i] * 7 + [c_p]),
    "afd_conv2d_bf16_workspace_bytes": (c_sz, [c_i, c_i, c_i]),
    "afd_conv2d_forward_bf16": (c_i, [c_p, c_p, c_p, c_p] + [c_i] * 8 + [c_p, c_sz, c_p]),
    "afd_lcnn_prep_bytes": (c_sz, [c_i, c_i, c_i]),
    "afd_lcnn_prep_conv_bf16": (c_i, [c_p, c_p, c_p, c_p, c_f, c_p, c_i, c_i, c_i, c_p]),
    "afd_lcnn_conv1_nhwc_bf16": (c_i, [c_p, c_p, c_p] + [c_i] * 7 + [c_p]),
    "afd_lcnn_conv_nhwc_bf16": (c_i, [c_p, c_p, c_p] + [c_i] * 8 + [c_p]),
    "afd_lcnn_pool_nhwc_bf16": (c_i, [c_p, c_p] + [c_i] * 5 + [c_p]),
    "afd_f32_to_bf16": (c_i, [c_p, c_p, c_sz, c_p]),
    "afd_lstm_step_bf16": (c_i, [c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_i, c_i, c_p]),
    "afd_lstm_step_bf16_pair": (c_i, [c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_i, c_i, c_p]),
    "afd_blstm_layer_bf16": (c_i, [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p]),
    "afd_lstm_cell": (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p]),
    "afd_lstm_cell_backward": (c_i, [c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_i, c_i, c_p]),
    "afd_cross_entropy": (c_i, [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_p]),
    "afd_adam_step": (c_i, [c_p, c_p, c_p, c_p, c_sz, c_f, c_f, c_f, c_f, c_f, c_i, c_f, c_p]),
}


def signatures():
    return dict(_SIGNATURES)


def load() -> ctypes.CDLL:
    """Load libafd_hip.so and declare every prototype; raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"libafd_hip.so not found at {LIB_PATH}: build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` "
            "(make -C audiodeepfake-detection_amd/csrc). There is no CPU fallback."
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().afd_last_error().decode(errors="replace")
        raise RuntimeError(f"libafd_hip {what} failed ({rc}): {msg}")


def require_gpu() -> None:
    if not torch.cuda.is_available():
        raise RuntimeError(
            "audiofakedetect (MI355X build) needs a visible gfx950 GPU; there is no CPU path."
        )


def stream_ptr() -> c_p:
    return c_p(torch.cuda.current_stream().cuda_stream)


def ptr(t: Optional[torch.Tensor]) -> c_p:
    return c_p(0) if t is None else c_p(t.data_ptr())


def float_array(vals):
    return (c_f * len(vals))(*[float(v) for v in vals])


KERNEL_CLASSES = {"wpt": 0, "conv_igemm": 1, "conv_wgrad": 2, "stft": 3, "conv_direct": 4,
                  "conv_winograd": 5, "conv_wgrad_1x1": 6, "lcnn_bf16": 7, "conv_first": 8, "batchnorm": 9,
                  "elementwise": 10, "conv1x1": 11}


def timing_enable(on: bool) -> None:
    load().afd_timing_enable(1 if on else 0)


def timing_collect(name: str) -> dict:
    """Sums over one kernel class's launches since the last reset: total_ms, launches, work
    (algorithmic bytes or direct-form flops), issued (matrix-core flops) and bytes (algorithmic)."""
    ms, n, work = ctypes.c_double(), c_l(), ctypes.c_double()
    issued, nbytes = ctypes.c_double(), ctypes.c_double()
    check(load().afd_timing_collect(KERNEL_CLASSES[name], ctypes.byref(ms), ctypes.byref(n),
                                    ctypes.byref(work), ctypes.byref(issued), ctypes.byref(nbytes)),
          "afd_timing_collect")
    return {"total_ms": ms.value, "launches": n.value, "work": work.value, "issued": issued.value,
            "bytes": nbytes.value}


def timing_reset() -> None:
    load().afd_timing_reset()
