"""Grid-search config for the entry-point test, in the reference's plugin protocol
(scripts/gridsearch_config.py: a module exposing get_config() -> dict of lists; the model class
is passed as an object under "module")."""

import os

from src.audiofakedetect.models import DCNN


def get_config() -> dict:
    cfg = _grid()
    if os.environ.get("AFD_TEST_DEFAULT_WAVELET") == "1":
        # leave --wavelet to the parser's default (sym8: 101 time steps, time_dim_add 0 as scripts/start_exps.sh:9)
        del cfg["wavelet"]
        cfg["time_dim_add"] = [0]
    return cfg


def _grid() -> dict:
    return {
        "transform": ["packets"], "wavelet": ["sym5"], "num_of_scales": [256],
        "learning_rate": [0.0004], "weight_decay": [0.001], "epochs": [1], "batch_size": [8],
        "dropout_cnn": [0.6], "dropout_lstm": [0.2], "model": ["modules"], "module": [DCNN],
        "kernel1": [3], "ochannels1": [64], "ochannels2": [64], "ochannels3": [96],
        "ochannels4": [128], "ochannels5": [32], "hop_length": [220], "sample_rate": [22050],
        "seconds": [1], "time_dim_add": [1], "flattend_size": [320], "validation_interval": [1],
        "limit_train": [(16, 8, 8)], "block_norm": [os.environ.get("AFD_TEST_BLOCK_NORM") == "1"],
    }
