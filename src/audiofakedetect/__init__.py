"""Import shim: ``src.audiofakedetect.*`` resolves to the MI355X host package.

The reference's launch scripts run ``python -m src.audiofakedetect.train_classifier`` and its
grid-search config does ``from src.audiofakedetect.models import DCNN``
(reference scripts/train.sh:33-68, scripts/gridsearch_config.py:8).  The implementation
lives in ``audiodeepfake-detection_amd/audiofakedetect``; this package only extends its
search path to there.
"""

import os as _os

_impl = _os.path.join(
    _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))),
    "audiodeepfake-detection_amd",
    "audiofakedetect",
)
__path__.append(_impl)
from .version import __version__  # noqa: E402,F401
