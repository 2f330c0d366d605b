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


def fill_dcnn_state_dict(shapes: dict) -> dict:
    """Like fill_state_dict, with magnitudes a trained network has: convolution / linear weights of standard
    deviation sqrt(2 / fan_in), PReLU slopes 0.25, BatchNorm scales near one -- so that activations keep their size
    through the stack and the logits depend on the input (used for the level-14 geometry, whose Linear is 80 960
    wide)."""
    out = fill_state_dict(shapes, scale=0.05)
    bn = {k[: -len("running_mean")] for k in shapes if k.endswith("running_mean")}
    for name in sorted(shapes):
        shape = tuple(shapes[name])
        g = torch.Generator().manual_seed(zlib.crc32(name.encode()) & 0x7FFFFFFF)
        prefix = name.rsplit(".", 1)[0] + "."
        if name.endswith("weight") and len(shape) >= 2:
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            out[name] = (2.0 / fan_in) ** 0.5 * torch.randn(shape, generator=g)
        elif name.endswith("weight") and prefix in bn:
            out[name] = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif name.endswith("weight"):
            out[name] = torch.full(shape, 0.25)
    return out


def level14_input(t_len: int) -> torch.Tensor:
    """Seeded stand-in for a normalised log-packet image [2, 1, 16384, t_len] in the plugin's memory order
    ([B, C, T, P] permuted to logical [B, C, P, T], as `Packets.forward` returns it): unit noise plus a slow ripple
    along the packet axis, so that neighbouring packets are correlated as real features are."""
    g = torch.Generator().manual_seed(1400 + t_len)
    mem = torch.randn(2, 1, t_len, 16384, generator=g)
    mem = mem + 0.5 * torch.sin(torch.arange(16384) / 97.0)
    return mem.permute(0, 1, 3, 2)
