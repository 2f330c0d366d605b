"""Generates the wavelet-packet golden vectors in tests/golden/ from PyWavelets.

Run in the build container only, with the interpreter that has PyWavelets (1.1.1 here):

    /opt/conda/bin/python3.9 tests/golden/make_wpt_golden.py

What is pinned.  The reference takes its filters from ``pywt.Wavelet(wavelet_str)``
(reference src/audiofakedetect/wavelet_math.py:239; pywavelets is requirements.txt:5)
and computes the packet tree with ``ptwt.WaveletPacket(data, wavelet, mode="reflect")``
followed by ``get_level(max_lev)`` (wavelet_math.py:182-192).  ptwt is not installed
anywhere on this machine; pywt is, and the reference itself runs exactly the pywt call

    pywt.WaveletPacket(data=clip_array, wavelet=wavelet, mode="reflect").get_level(14, order="freq")

for the same analysis (scripts/freq_visual/fingerprints.py:101-106).  The fixtures
written here are the outputs of that call (float64 arithmetic) on seeded 1 s frames.
The residual assumption is ptwt == pywt for ``mode="reflect"`` (ptwt's own test-suite
is built on that equality).

Only arrays are written (inputs, taps, node coefficients); no source of pywt or of the
reference.  Files:

* ``pywt_taps.npz``      dec_lo / dec_hi of every discrete wavelet pywt knows
* ``pywt_wpt_core.npz``  haar, sym5, coif4, db8, sym8 at levels 1, 3, 8: three frames
                         (noise, four tones, impulses at 0 / 11 025 / 22 049), float64
* ``pywt_wpt_l14.npz``   haar, sym5, coif4, sym8 at level 14, one frame: float32 nodes
                         plus float64 per-node sums and sums of squares
* ``pywt_wpt_names.npz`` every wavelet of scripts/start_exps.sh:3-31 (+ the longest of
                         each family) at level 8 on one composite frame, float32 nodes
                         plus float64 per-node sums and sums of squares
"""

import os

import numpy as np
import pywt

OUT = os.path.dirname(os.path.abspath(__file__))
N = 22050
CORE = ["haar", "sym5", "coif4", "db8", "sym8"]
L14 = ["haar", "sym5", "coif4", "sym8"]
START_EXPS = ([f"sym{k}" for k in range(2, 11)] + [f"db{k}" for k in range(2, 11)]
              + [f"coif{k}" for k in range(2, 11)])
EXTREMES = ["db11", "db20", "db38", "sym11", "sym20", "coif1", "coif11", "coif17", "dmey",
            "bior2.2", "bior4.4", "bior6.8", "rbio3.9"]


def frames() -> np.ndarray:
    """float32 [4, N]: noise, four tones, three impulses, composite (noise + tones + impulses)."""
    rng = np.random.default_rng(20261005)
    t = np.arange(N) / 22050.0
    noise = 0.1 * rng.standard_normal(N)
    tones = sum(a * np.sin(2 * np.pi * f * t + p) for a, f, p in
                ((0.5, 440.0, 0.0), (0.3, 3000.0, 0.4), (0.2, 7500.0, 1.1), (0.1, 10500.0, 2.3)))
    imp = np.zeros(N)
    imp[0], imp[11025], imp[22049] = 1.0, -0.75, 0.5
    comp = 0.05 * rng.standard_normal(N) + 0.3 * tones + 0.4 * imp
    return np.stack([noise, tones, imp, comp]).astype(np.float32)


def packet_level(x: np.ndarray, name: str, level: int) -> np.ndarray:
    """float64 [F, 2^level, T]: the call of fingerprints.py:101-106 with maxlevel stated."""
    out = []
    for frame in x.astype(np.float64):  # pywt 1.1.1's WaveletPacket takes one 1-D signal
        tree = pywt.WaveletPacket(data=frame, wavelet=pywt.Wavelet(name), mode="reflect", maxlevel=level)
        out.append(np.stack([n.data for n in tree.get_level(level, order="freq")]))
    return np.stack(out)


def main() -> None:
    x = frames()
    taps = {}
    for name in pywt.wavelist(kind="discrete"):
        w = pywt.Wavelet(name)
        taps[name + "/lo"] = np.asarray(w.dec_lo, dtype=np.float64)
        taps[name + "/hi"] = np.asarray(w.dec_hi, dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "pywt_taps.npz"), **taps)

    core = {"x": x[:3], "pywt_version": np.array(pywt.__version__)}
    for name in CORE:
        for level in (1, 3, 8):
            core[f"{name}/{level}"] = packet_level(x[:3], name, level)
    np.savez_compressed(os.path.join(OUT, "pywt_wpt_core.npz"), **core)

    deep = {"x": x[:1]}
    for name in L14:
        nodes = packet_level(x[:1], name, 14)[0]
        deep[name] = nodes.astype(np.float32)
        deep[name + "/sum"] = nodes.sum(-1)
        deep[name + "/sumsq"] = (nodes * nodes).sum(-1)
    np.savez_compressed(os.path.join(OUT, "pywt_wpt_l14.npz"), **deep)

    names = {"x": x[3:4]}
    for name in START_EXPS + EXTREMES:
        nodes = packet_level(x[3:4], name, 8)[0]
        names[name] = nodes.astype(np.float32)
        names[name + "/sum"] = nodes.sum(-1)
        names[name + "/sumsq"] = (nodes * nodes).sum(-1)
    np.savez_compressed(os.path.join(OUT, "pywt_wpt_names.npz"), **names)
    for f in ("pywt_taps", "pywt_wpt_core", "pywt_wpt_l14", "pywt_wpt_names"):
        print(f, os.path.getsize(os.path.join(OUT, f + ".npz")))


if __name__ == "__main__":
    main()
