"""Orthogonal wavelet filter taps for the wavelet-packet front end.

The reference obtains its filters from ``pywt.Wavelet(wavelet_str)``
(reference ``src/audiofakedetect/wavelet_math.py:239``) and hands them to ptwt.
pywt is a third-party dependency that is not part of the reference tree, so this
module carries the decomposition low-pass tables itself (pywt convention:
``dec_hi[k] = (-1)**(k+1) * dec_lo[L-1-k]``) and exposes the small subset of the
``pywt.Wavelet`` surface the hot path touches (``dec_lo``, ``dec_hi``,
``dec_len``, ``name``).

Tables: haar, db2..db10 (computed by spectral factorisation, minimum phase, as
pywt stores them), sym5 and coif4 (literal tables), and every
other discrete wavelet pywt names -- db11..db38, sym2..sym20, coif1..coif17, dmey,
bior*/rbio* -- from ``wavelet_tables.py`` (data lifted from PyWavelets 1.1.1 by
``tools/gen_wavelet_tables.py``), so that every ``--wavelet`` of the reference's launch
scripts (scripts/start_exps.sh:3-31) and its default sym8 (utils.py:84-89) construct.
``tests/test_oracle_wpt.py`` checks all of them against the pywt fixture
``tests/golden/pywt_taps.npz``.  coif4 keeps the higher-precision table (it meets the
coiflet conditions to 3e-13; pywt 1.1.1's differs from it by 2.3e-8, below half an ulp
of the fp32 taps the kernels use).  Other wavelets can be registered with
:func:`register_wavelet`.
"""

from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import numpy as np

from .wavelet_tables import TABLES as _PYWT_TABLES

_SYM5 = [
    0.027333068345077982,
    0.029519490925774643,
    -0.039134249302383094,
    0.1993975339773936,
    0.7234076904024206,
    0.6339789634582119,
    0.01660210576452232,
    -0.17532808990845047,
    -0.021101834024758855,
    0.019538882735286728,
]

_COIF4 = [
    -1.7849850030882614e-06,
    -3.2596802368833675e-06,
    3.1229875865345646e-05,
    6.233903446100713e-05,
    -0.00025997455248771324,
    -0.0005890207562443383,
    0.0012665619292989445,
    0.003751436157278457,
    -0.00565828668661072,
    -0.015211731527946259,
    0.025082261844864097,
    0.03933442712333749,
    -0.09622044203398798,
    -0.06662747426342504,
    0.4343860564914685,
    0.782238930920499,
    0.41530840703043026,
    -0.05607731331675481,
    -0.08126669968087875,
    0.026682300156053072,
    0.016068943964776348,
    -0.0073461663276420935,
    -0.0016294920126017326,
    0.0008923136685823146,
]


def _daubechies_dec_lo(order: int) -> List[float]:
    """Minimum-phase Daubechies filter of `order` vanishing moments (2*order taps).

    Spectral factorisation of the Daubechies polynomial
    P(y) = sum_k C(order-1+k, k) y^k with y = (2 - z - 1/z)/4; the roots inside the
    unit circle are kept.  Returned in pywt's ``dec_lo`` orientation (the reversed
    minimum-phase sequence).
    """
    if order == 1:
        s = 1.0 / math.sqrt(2.0)
        return [s, s]
    coeffs = [math.comb(order - 1 + k, k) for k in range(order)]
    yroots = np.roots(coeffs[::-1])
    zroots = []
    for y in yroots:
        # z^2 - (2 - 4y) z + 1 = 0
        b = 2.0 - 4.0 * y
        disc = np.sqrt(b * b - 4.0 + 0j)
        z1 = (b + disc) / 2.0
        z2 = (b - disc) / 2.0
        zroots.append(z1 if abs(z1) < 1.0 else z2)
    poly = np.array([1.0 + 0j])
    for _ in range(order):
        poly = np.convolve(poly, [1.0, 1.0])
    for z in zroots:
        poly = np.convolve(poly, [1.0, -z])
    h = np.real(poly)
    h = h * (math.sqrt(2.0) / h.sum())
    # h is the minimum-phase scaling filter (pywt rec_lo); dec_lo is its reverse.
    return [float(v) for v in h[::-1]]


_TABLES: Dict[str, List[float]] = {}
_HI_TABLES: Dict[str, List[float]] = {}  # only the banks whose dec_hi is not the mirror of dec_lo


def register_wavelet(name: str, dec_lo: Sequence[float], dec_hi: Optional[Sequence[float]] = None) -> None:
    """Register the decomposition filters of a two-channel bank (`dec_hi` omitted: the quadrature mirror of
    `dec_lo`, an orthogonal wavelet)."""
    taps = [float(v) for v in dec_lo]
    if len(taps) % 2 != 0 or len(taps) < 2:
        raise ValueError("a two-channel wavelet filter has an even number of taps")
    _TABLES[name] = taps
    _HI_TABLES.pop(name, None)
    if dec_hi is not None:
        hi = [float(v) for v in dec_hi]
        if len(hi) != len(taps):
            raise ValueError("dec_lo and dec_hi must have the same length")
        _HI_TABLES[name] = hi


register_wavelet("haar", _daubechies_dec_lo(1))
register_wavelet("db1", _daubechies_dec_lo(1))
for _n in range(2, 11):
    register_wavelet(f"db{_n}", _daubechies_dec_lo(_n))
register_wavelet("sym5", _SYM5)
register_wavelet("coif4", _COIF4)
for _name, (_lo, _hi) in _PYWT_TABLES.items():
    if _name not in _TABLES:
        register_wavelet(_name, _lo, _hi)


class Wavelet:
    """The part of ``pywt.Wavelet`` the wavelet-packet front end reads."""

    def __init__(self, name: str) -> None:
        if name not in _TABLES:
            raise ValueError(
                f"Unknown wavelet name '{name}', register its dec_lo taps with "
                "audiofakedetect.wavelets.register_wavelet(name, dec_lo)."
            )
        self.name = name
        self.dec_lo = list(_TABLES[name])
        length = len(self.dec_lo)
        if name in _HI_TABLES:
            self.dec_hi = list(_HI_TABLES[name])
        else:
            self.dec_hi = [
                (-1.0) ** (k + 1) * self.dec_lo[length - 1 - k] for k in range(length)
            ]
        # synthesis pair of a perfect-reconstruction bank (for an orthogonal one: the reversed analysis filters)
        self.rec_lo = [(-1.0) ** (k + 1) * self.dec_hi[k] for k in range(length)]
        self.rec_hi = [(-1.0) ** k * self.dec_lo[k] for k in range(length)]
        self.dec_len = length
        self.rec_len = length

    def __len__(self) -> int:
        return self.dec_len

    def __repr__(self) -> str:
        return f"Wavelet({self.name!r}, dec_len={self.dec_len})"


def wavelist() -> List[str]:
    """Names of the registered wavelets."""
    return sorted(_TABLES)


def node_length(n: int, filt_len: int) -> int:
    """Length of a child node of a length-`n` parent (reflect mode, ptwt padding).

    pad_l = L-2, pad_r = L-2 + (n odd); stride-2 valid correlation with L taps.
    """
    return (n + filt_len - 2 + (n % 2)) // 2


def level_lengths(n: int, filt_len: int, level: int) -> List[int]:
    """Node lengths n_0..n_level of the packet tree."""
    out = [n]
    for _ in range(level):
        n = node_length(n, filt_len)
        out.append(n)
    return out
