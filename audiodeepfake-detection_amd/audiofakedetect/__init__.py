"""audiofakedetect hot path, MI355X-native (host side).

Mirrors the module layout of the reference package ``src/audiofakedetect`` for the
transform / model / trainer plugin API; the arithmetic runs in ``libafd_hip.so``.
"""

from .version import __version__  # noqa: F401
