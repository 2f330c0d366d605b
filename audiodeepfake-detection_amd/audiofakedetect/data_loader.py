"""Data-side pieces the hot path touches.

``WelfordEstimator`` mirrors reference ``src/audiofakedetect/data_loader.py:27-71``.
The reference's folder indexer / windowed WAV reader (``CustomDataset``, :74-507) is
out of scope this round; ``SyntheticFrames`` produces the
item format it emits (``{"audio": f32[1, N], "label": int64}``, :351-353,392) so the
trainer and ``get_input_dims`` run unchanged on synthetic data.
"""

from __future__ import annotations

from typing import Optional, Tuple

import numpy as np
import torch
from torch.utils.data import Dataset


class WelfordEstimator:
    """Running mean / std over all axes but the last (the channel axis)."""

    def __init__(self) -> None:
        self.collapsed_axis: Optional[Tuple[int, ...]] = None

    def update(self, batch_vals: torch.Tensor) -> None:
        if not self.collapsed_axis:
            self.collapsed_axis = tuple(range(batch_vals.dim() - 1))
            dev = batch_vals.device
            nch = batch_vals.shape[-1]
            self.count = torch.zeros(1, device=dev, dtype=torch.float32)
            self.mean = torch.zeros(nch, device=dev, dtype=torch.float32)
            self.std = torch.zeros(nch, device=dev, dtype=torch.float32)
            self.m2 = torch.zeros(nch, device=dev, dtype=torch.float32)
        self.count += int(np.prod(batch_vals.shape[:-1]))
        delta = batch_vals - self.mean
        self.mean += torch.sum(delta / self.count, self.collapsed_axis)
        delta2 = batch_vals - self.mean
        self.m2 += torch.sum(delta * delta2, self.collapsed_axis)

    def finalize(self) -> Tuple[torch.Tensor, torch.Tensor]:
        return self.mean, torch.sqrt(self.m2 / self.count)


class SyntheticFrames(Dataset):
    """Seeded synthetic 1 s frames: 0.1 * randn clipped to [-1, 1]."""

    key = "audio"

    def __init__(self, length: int = 1024, num_samples: int = 22050, seed: int = 1234,
                 num_labels: int = 2) -> None:
        self.length = length
        self.num_samples = num_samples
        self.seed = seed
        self.num_labels = num_labels

    def __len__(self) -> int:
        return self.length

    def __getitem__(self, idx: int) -> dict:
        g = torch.Generator().manual_seed(self.seed * 1_000_003 + idx)
        audio = (0.1 * torch.randn(1, self.num_samples, generator=g)).clamp_(-1.0, 1.0)
        label = torch.randint(0, self.num_labels, (1,), generator=g)[0].to(torch.int64)
        return {"audio": audio, "label": label, "index": idx}


def get_costum_dataset(data_path=None, ds_type="train", only_use=None, save_path=None,
                       limit=None, asvspoof_name=None, file_type="wav", resample_rate=22050,
                       seconds=1, synthetic=False, **_):
    """Dataset factory (name kept from the reference, data_loader.py:397-507)."""
    if synthetic or data_path is None:
        length = int(limit) if limit else 1024
        seed = {"train": 1234, "val": 4321, "test": 9876}.get(ds_type, 1)
        return SyntheticFrames(length, int(resample_rate * (seconds or 1)), seed)
    raise NotImplementedError(
        "The on-disk dataset indexer of the reference is out of scope this round "
        "(SURVEY.md 8(f-3)); pass --synthetic or build the dataset yourself."
    )
