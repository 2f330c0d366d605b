"""CPU checks of the C-ABI boundary: the library loads and exports what include/afd_hip.h declares."""

import os
import re

from audiofakedetect import _native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "afd_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(afd_[a-z0-9_]+)\s*\(", text)))


def test_library_loads_and_exports_every_declared_symbol():
    lib = _native.load()
    declared = _declared_symbols()
    assert "afd_wpt_forward" in declared
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in afd_hip.h but not exported"
    assert lib.afd_version() >= 1


def test_python_binding_covers_the_header():
    assert set(_native.signatures()) == set(_declared_symbols())


def test_out_len_matches_host_table():
    from audiofakedetect import wavelets

    lib = _native.load()
    for length, table in ((2, "haar"), (10, "sym5"), (24, "coif4")):
        for level in (1, 8, 14):
            assert lib.afd_wpt_out_len(22050, length, level) == \
                wavelets.level_lengths(22050, length, level)[level]
    assert lib.afd_wpt_out_len(20, 24, 3) == -1


def test_input_fold_is_offered_for_the_shipped_geometries_and_refused_elsewhere():
    """`afd_conv3x3_input_fold_applicable` (host-only arithmetic): the BatchNorms in front of DCNN blocks 3-6 may leave
    their normalisation to the next convolution at the level-14 coif4, level-8 and STFT geometries of BASELINE.json;
    shapes off the F(4x4) kernels' grid -- and the AFD_NO_INPUT_FOLD switch -- keep the two-pass chain."""
    lib = _native.load()
    blocks = {  # (cin, h, w, cout, pooled, statistics behind the convolution) per geometry, blocks 3..6
        "coif4 level 14": [(64, 13, 8193, 96, 1, 1), (96, 6, 4096, 128, 0, 1), (128, 6, 4096, 32, 0, 1), (32, 6, 4096, 64, 1, 0)],
        "coif4 level 8": [(64, 55, 129, 96, 1, 1), (96, 27, 64, 128, 0, 1), (128, 27, 64, 32, 0, 1), (32, 27, 64, 64, 1, 0)],
        "stft": [(64, 51, 129, 96, 1, 1), (96, 25, 64, 128, 0, 1), (128, 25, 64, 32, 0, 1), (32, 25, 64, 64, 1, 0)],
    }
    for name, layers in blocks.items():
        for geom in layers:
            assert lib.afd_conv3x3_input_fold_applicable(*geom) == 1, (name, geom)
    # sym5 level 14: block 3 works on 6 rows, blocks 4-6 on 3 rows -- the F(4x4) kernels' minimum
    assert lib.afd_conv3x3_input_fold_applicable(64, 6, 8193, 96, 1, 1) == 1
    assert lib.afd_conv3x3_input_fold_applicable(96, 3, 4096, 128, 0, 1) == 1
    assert lib.afd_conv3x3_input_fold_applicable(96, 2, 4096, 128, 0, 1) == 0
    # off the grid: 48 input channels (LCNN), an image under 48 columns, a crop the backward-weight kernel refuses
    assert lib.afd_conv3x3_input_fold_applicable(48, 16, 128, 96, 1, 1) == 0
    assert lib.afd_conv3x3_input_fold_applicable(64, 16, 32, 96, 1, 1) == 0
    assert lib.afd_conv3x3_input_fold_applicable(64, 13, 1030, 96, 1, 1) == 0
    os.environ["AFD_NO_INPUT_FOLD"] = "1"
    try:
        assert lib.afd_conv3x3_input_fold_applicable(64, 13, 8193, 96, 1, 1) == 0
    finally:
        del os.environ["AFD_NO_INPUT_FOLD"]
