"""Integrated-gradients helpers (reference ``src/audiofakedetect/integrated_gradients.py``).

Only the numerical part of the attribution workflow is provided -- the running mean over
attribution maps (:13-47), the straight-line path between a baseline and an image (:104-123)
and the trapezoidal path integral (:126-138).  The plotting helpers of the reference
(matplotlib / tikzplotlib figures) are tooling outside the hot path; the ``.npy`` files the
driver writes (``Trainer.integrated_gradients``) carry the same arrays the reference plots.
The forward / backward passes behind the path gradients are the DCNN / LCNN HIP kernels.
"""

from __future__ import annotations

from typing import Optional

import torch


class Mean:
    """Running mean over equally shaped maps; ``finalize`` also averages the leading axis."""

    def __init__(self) -> None:
        self.init: Optional[bool] = None

    def update(self, batch_vals: torch.Tensor) -> None:
        vals = batch_vals.detach().to(torch.float32)
        if self.init is None:
            self.init = True
            self.count = 0
            self.mean = torch.zeros_like(vals)
        self.count += 1
        self.mean += vals

    def finalize(self) -> torch.Tensor:
        return torch.mean(self.mean, dim=0).squeeze() / self.count


def interpolate_images(baseline: torch.Tensor, image: torch.Tensor, alphas: torch.Tensor) -> torch.Tensor:
    """[len(alphas), *image.shape]: baseline + alpha * (image - baseline)."""
    shape = (-1,) + (1,) * image.dim()
    return baseline.unsqueeze(0) + alphas.reshape(shape) * (image - baseline).unsqueeze(0)


def integral_approximation(gradients: torch.Tensor) -> torch.Tensor:
    """Trapezoidal rule over the path axis (uniform alpha grid): mean of neighbour averages."""
    return (0.5 * (gradients[:-1] + gradients[1:])).mean(dim=0)
