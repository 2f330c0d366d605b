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
