"""Deterministic weight recipes shared by make_golden.py and the tests.

Large weight sets (LCNN: 3.3 M parameters) are not stored as fixtures; both sides rebuild
them from the parameter names and shapes with this recipe.
"""

import zlib

import torch


def fill_state_dict(shapes: dict, scale: float = 0.05) -> dict:
    """name -> tensor; values depend only on (name, shape)."""
    out = {}
    for name in sorted(shapes):
        shape = tuple(shapes[name])
        g = torch.Generator().manual_seed(zlib.crc32(name.encode()) & 0x7FFFFFFF)
        if name.endswith("num_batches_tracked"):
            out[name] = torch.tensor(3, dtype=torch.long)
        elif name.endswith("running_var"):
            out[name] = 0.5 + torch.rand(shape, generator=g)
        elif name.endswith("running_mean"):
            out[name] = 0.1 * torch.randn(shape, generator=g)
        else:
            out[name] = scale * torch.randn(shape, generator=g)
    return out
