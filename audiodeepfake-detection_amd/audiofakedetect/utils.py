"""Host plumbing of the hot path: config container, seeds, CLI flags, grid iterator.

API-compatible with the reference's ``src/audiofakedetect/utils.py`` for the names the hot
path and ``scripts/train.sh`` touch (``DotDict`` :320-332, ``set_seed`` :18-27,
``add_default_parser_args`` :30-317, ``_Griderator``/``build_new_grid`` :480-586,
``get_input_dims`` :589-621).  Nothing here is accelerated.
"""

from __future__ import annotations

import itertools
import os
import random
from argparse import ArgumentParser
from typing import Any, Iterable, Optional

import numpy as np
import torch


def set_seed(seed: int) -> None:
    """Seed python/numpy/torch (CPU and every visible GPU)."""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)


class DotDict(dict):
    """dict whose keys are also attributes (missing attribute -> None, like dict.get)."""

    __getattr__ = dict.get
    __setattr__ = dict.__setitem__  # type: ignore[assignment]
    __delattr__ = dict.__delitem__  # type: ignore[assignment]


def _flag(parser: ArgumentParser, name: str, **kw) -> None:
    parser.add_argument(name, **kw)


def add_default_parser_args(parser: ArgumentParser) -> ArgumentParser:
    """The reference's training flags (same names, types and defaults where the path reads them)."""
    f = lambda name, **kw: _flag(parser, name, **kw)  # noqa: E731
    f("--log-dir", type=str, default="./exp/log")
    f("--batch-size", type=int, default=128)
    f("--learning-rate", type=float, default=1e-4)
    f("--weight-decay", type=float, default=1e-4)
    f("--epochs", type=int, default=10)
    f("--transform", choices=["stft", "packets"], default="stft")
    f("--features", choices=["lfcc", "delta", "doubledelta", "none"], default="none")
    f("--num-of-scales", type=int, default=256)
    f("--wavelet", type=str, default="sym8")
    f("--sample-rate", type=int, default=22050)
    f("--window-size", type=int, default=11025)
    f("--f-min", type=float, default=1000)
    f("--f-max", type=float, default=9500)
    f("--hop-length", type=int, default=100)
    f("--log-scale", action="store_true")
    f("--block-norm", action="store_true")
    f("--power", type=float, default=2.0)
    f("--dropout-cnn", type=float, default=0.6)
    f("--dropout-lstm", type=float, default=0.3)
    f("--loss-less", choices=["True", "False"], default="False")
    f("--random-seeds", action="store_true")
    f("--aug-contrast", action="store_true")
    f("--aug-noise", action="store_true")
    f("--calc-normalization", action="store_true")
    f("--mean", type=float, default=0.0)
    f("--std", type=float, default=1.0)
    f("--data-prefix", type=str, default="../data/fake")
    f("--unknown-prefix", type=str, default=None)
    f("--cross-dir", type=str, default=None)
    f("--cross-prefix", type=str, default=None)
    f("--cross-sources", type=str, nargs="+", default=None)
    f("--init-seeds", type=int, nargs="+", default=None)
    f("--seed", type=int, default=0)
    f("--flattend-size", type=int, default=21888)
    f("--model", choices=["lcnn", "gridmodel", "modules"], default="lcnn")
    f("--nclasses", type=int, default=2)
    f("--enable-gs", action="store_true")
    f("--tensorboard", action="store_true")
    f("--pbar", action="store_true")
    f("--validation-interval", type=int, default=1)
    f("--only-testing", action="store_true")
    f("--ckpt-every", type=int, default=500)
    f("--time-dim-add", type=int, default=0)
    f("--ddp", action="store_true")
    f("--only-ig", action="store_true")
    f("--config", type=str, default=None)
    # MI355X build only: run on synthetic frames (no dataset on disk)
    f("--synthetic", action="store_true")
    f("--synthetic-steps", type=int, default=8)
    f("--num-workers", type=int, default=None)  # loader workers; default: CPU share (the reference hard-codes 10)
    # batches straight from the library's threaded WAV reader + GPU PCM conversion / resampling instead of
    # DataLoader workers (16-bit PCM WAV datasets; same values)
    f("--native-loader", action="store_true")
    return parser


class _Griderator:
    """Iterates the cartesian product of the config lists, seeds first."""

    def __init__(self, config: dict, init_seeds: Optional[list] = None, num_exp: int = 5) -> None:
        if not isinstance(config, dict):
            raise TypeError(f"Config file must be of type dict but is {type(config)}.")
        if init_seeds is None:
            rng = random.SystemRandom()
            init_seeds = [rng.randrange(10000) for _ in range(num_exp)]
        self.init_config: dict[str, Any] = {"seed": list(init_seeds)}
        self.init_config.update(config)
        self.grid_values = list(itertools.product(*self.init_config.values()))
        self.current = 0

    def get_keys(self):
        return self.init_config.keys()

    def get_len(self) -> int:
        return len(self.grid_values)

    def __iter__(self):
        return self

    def __next__(self):
        self.current += 1
        if self.current < len(self.grid_values):
            return self.grid_values[self.current]
        raise StopIteration

    def next(self):
        return self.__next__()

    def reset(self) -> None:
        self.current = 0

    def update_args(self, args: DotDict) -> DotDict:
        for key, value in zip(self.get_keys(), self.grid_values[self.current]):
            args[key] = value
        return args

    def update_step(self, args: DotDict):
        new_args = self.update_args(args)
        try:
            step = self.__next__()
        except StopIteration:
            return new_args, StopIteration
        return new_args, step


def build_new_grid(config: dict, random_seeds: bool = False, seeds: Optional[list] = None) -> _Griderator:
    if random_seeds:
        return _Griderator(config, num_exp=3)
    init_seeds: Iterable[int] = [0, 1, 2, 3, 4]
    if isinstance(seeds, list):
        init_seeds = [int(s) for s in seeds]
    return _Griderator(config, init_seeds=list(init_seeds))


def get_input_dims(args: DotDict, transforms) -> list:
    """Shape of the transformed batch: runs the transform on one frame (utils.py:589-621)."""
    from .data_loader import get_costum_dataset

    dataset = get_costum_dataset(
        data_path=args.data_path,
        ds_type="train",
        only_use=args.only_use,
        save_path=args.save_path,
        limit=args.limit_train[0] if args.limit_train else None,
        file_type=args.file_type,
        resample_rate=args.sample_rate,
        seconds=args.seconds,
        synthetic=bool(args.synthetic),
    )
    with torch.no_grad():
        audio = dataset[0]["audio"]
        if torch.cuda.is_available():
            audio = audio.cuda(non_blocking=True)
        feats, _ = transforms(audio)
    shape = list(feats.shape)
    if len(shape) < 4:
        shape.insert(0, args.batch_size)
    else:
        shape[0] = args.batch_size
    return shape
