"""ORACLE (test infrastructure only) -- float64 numpy restatement of the WPT front end.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this package.  It is the checker, never the product.

Restates, for the wavelet-packet path of the reference:

* ``compute_pytorch_packet_representation``  reference
  ``src/audiofakedetect/wavelet_math.py:167-220``
* ``Packets.forward``                         ``wavelet_math.py:249-263``
* the ptwt analysis step the reference calls at ``wavelet_math.py:182,192``
  (ptwt is a third-party dependency, unpinned in the reference's
  ``requirements.txt:4``; its published algorithm is restated here):
  ``F.pad(x, (L-2, L-2 + (n odd)), "reflect")`` then
  ``F.conv1d(., stack(flip(dec_lo), flip(dec_hi)), stride=2)``, applied recursively
  to both outputs; ``get_level`` returns the nodes in frequency (Gray code) order.

PARITY PIN: pinned on PyWavelets.  The reference's own tests hold SHAPES only for this
path (``tests/test_transforms.py:54-142``); its dependency pywt (``requirements.txt:5``,
the source of the taps at ``wavelet_math.py:239``) computes the same tree as
``pywt.WaveletPacket(x, w, mode="reflect").get_level(l, order="freq")`` -- the call the
reference itself makes in ``scripts/freq_visual/fingerprints.py:101-106``.
``tests/golden/make_wpt_golden.py`` (run in the build container with the interpreter
that has pywt 1.1.1) wrote that call's float64 outputs and pywt's tap tables to
``tests/golden/pywt_*.npz``; ``tests/test_oracle_wpt.py`` holds this restatement, the C
one (``oracle/wpt_oracle.c``) and the torch ``F.pad``/``F.conv1d`` one
(``oracle/torch_ref.py``) to them (<= 1e-13 with pywt's taps at levels 1, 3, 8, 14), next
to the reference's shape asserts and the mathematical known answers (Haar closed form,
constant input, pure-tone packet index = Gray-code order, energy).  Residual assumption:
ptwt == pywt for ``mode="reflect"`` (ptwt is installed nowhere on this machine).
"""

from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple

import numpy as np

# Independent copies of the tap tables (the product has its own in
# audiofakedetect/wavelets.py; tests compare the two).
HAAR = [1.0 / math.sqrt(2.0), 1.0 / math.sqrt(2.0)]
SYM5 = [
    0.027333068345077982, 0.029519490925774643, -0.039134249302383094,
    0.1993975339773936, 0.7234076904024206, 0.6339789634582119,
    0.01660210576452232, -0.17532808990845047, -0.021101834024758855,
    0.019538882735286728,
]
COIF4 = [
    -1.7849850030882614e-06, -3.2596802368833675e-06, 3.1229875865345646e-05,
    6.233903446100713e-05, -0.00025997455248771324, -0.0005890207562443383,
    0.0012665619292989445, 0.003751436157278457, -0.00565828668661072,
    -0.015211731527946259, 0.025082261844864097, 0.03933442712333749,
    -0.09622044203398798, -0.06662747426342504, 0.4343860564914685,
    0.782238930920499, 0.41530840703043026, -0.05607731331675481,
    -0.08126669968087875, 0.026682300156053072, 0.016068943964776348,
    -0.0073461663276420935, -0.0016294920126017326, 0.0008923136685823146,
]
TAPS = {"haar": HAAR, "db1": HAAR, "sym5": SYM5, "coif4": COIF4}


def dec_hi_from_lo(dec_lo: Sequence[float]) -> List[float]:
    """pywt convention: dec_hi[k] = (-1)^(k+1) dec_lo[L-1-k]."""
    length = len(dec_lo)
    return [(-1.0) ** (k + 1) * dec_lo[length - 1 - k] for k in range(length)]


def child_length(n: int, filt_len: int) -> int:
    return (n + filt_len - 2 + (n % 2)) // 2


def analysis_step(x: np.ndarray, dec_lo: Sequence[float],
                  dec_hi: Optional[Sequence[float]] = None) -> Tuple[np.ndarray, np.ndarray]:
    """One two-channel analysis step along the last axis (ptwt ``wavedec(level=1)``).

    cA[i] = sum_m dec_lo[m] * xe[2i+1-m], cD likewise with dec_hi, where xe is the
    whole-sample reflect extension of x.  `dec_hi` omitted: the quadrature mirror of
    `dec_lo` (orthogonal wavelets); the biorthogonal families pass their own.
    """
    x = np.asarray(x, dtype=np.float64)
    lo = np.asarray(dec_lo, dtype=np.float64)
    hi = np.asarray(dec_hi_from_lo(dec_lo) if dec_hi is None else dec_hi, dtype=np.float64)
    filt_len = len(lo)
    n = x.shape[-1]
    padl = filt_len - 2
    padr = filt_len - 2 + (n % 2)
    if max(padl, padr) >= n:
        raise ValueError("reflect padding needs pad < node length")
    pad = [(0, 0)] * (x.ndim - 1) + [(padl, padr)]
    p = np.pad(x, pad, mode="reflect")
    n_out = (p.shape[-1] - filt_len) // 2 + 1
    assert n_out == child_length(n, filt_len)
    # correlation with the flipped taps == convolution: out[i] = sum_k lo[L-1-k] p[2i+k]
    ca = np.zeros(x.shape[:-1] + (n_out,), dtype=np.float64)
    cd = np.zeros_like(ca)
    for k in range(filt_len):
        seg = p[..., k : k + 2 * n_out : 2]
        ca += lo[filt_len - 1 - k] * seg
        cd += hi[filt_len - 1 - k] * seg
    return ca, cd


def graycode_paths(level: int) -> List[str]:
    """Node paths of one level in frequency order (ptwt ``get_level``)."""
    order = ["a", "d"]
    for _ in range(level - 1):
        order = ["a" + p for p in order] + ["d" + p for p in order[::-1]]
    return order


def wpt_nodes_by_path(x: np.ndarray, dec_lo: Sequence[float], level: int) -> np.ndarray:
    """Reference-shaped traversal: a dict of path -> node, gathered by `graycode_paths`.

    Slow (one analysis step per node); used to pin `wpt_nodes` on small cases.
    """
    x = np.asarray(x, dtype=np.float64)
    nodes = {"": x}
    for _ in range(level):
        nxt = {}
        for path, data in nodes.items():
            ca, cd = analysis_step(data, dec_lo)
            nxt[path + "a"] = ca
            nxt[path + "d"] = cd
        nodes = nxt
    return np.stack([nodes[p] for p in graycode_paths(level)], axis=-2)


def wpt_nodes(x: np.ndarray, dec_lo: Sequence[float], level: int,
              dec_hi: Optional[Sequence[float]] = None) -> np.ndarray:
    """Level-`level` packet nodes of x[..., N] in frequency order -> [..., P, T].

    All nodes of a level have the same length, so a level is one vectorised analysis
    step over a [..., nodes, n] array in path order (a=0, d=1, level 1 = MSB); the
    frequency-order gather is node f <- path index f ^ (f >> 1) (Gray code), which
    `tests/test_oracle_wpt.py` checks against `graycode_paths`.
    """
    cur = np.asarray(x, dtype=np.float64)[..., None, :]
    for _ in range(level):
        ca, cd = analysis_step(cur, dec_lo, dec_hi)
        cur = np.stack([ca, cd], axis=-2).reshape(ca.shape[:-2] + (-1, ca.shape[-1]))
    f = np.arange(1 << level)
    return cur[..., f ^ (f >> 1), :]


def packet_features(
    x: np.ndarray,
    dec_lo: Sequence[float],
    level: int,
    log_scale: bool = False,
    loss_less: bool = False,
    power: float = 2.0,
    block_norm: bool = False,
    dec_hi: Optional[Sequence[float]] = None,
) -> np.ndarray:
    """``Packets.forward`` output, logical [B, C, P, T] (float64).

    Input [B, N] (a [B, 1, N] batch is squeezed, as the reference's training batches
    are, data_loader.py:351-353).
    """
    x = np.asarray(x, dtype=np.float64)
    if x.ndim == 3:
        x = x[:, 0, :]
    nodes = wpt_nodes(x, dec_lo, level, dec_hi)  # [B, P, T]
    if block_norm:
        # wavelet_math.py:202-203: node / max|node| over the whole batch node tensor
        mx = np.max(np.abs(nodes), axis=(0, 2), keepdims=True)
        nodes = nodes / mx
    if log_scale:
        logp = np.log(np.abs(nodes) ** power + 1e-12)
        if loss_less:
            sign = ((nodes < 0).astype(np.float64) * (-1) + 0.5) * 2
            return np.stack([logp, sign], axis=1)
        return logp[:, None]
    return nodes[:, None]
