"""Host-side helpers with the reference's names (libraries/NeRF/utils.py:13-88): `to_local`, `in_cube`,
`positional_encoding`, `multi_part_positional_encoding`.

Inside the fused kernels the bone transform and the cube test never leave registers; these functions exist for callers of
the reference API that use them stand-alone, and for the ONE place the render path needs an encoding on the host: the
bone-length conditioning of the tri-plane producers (models/narf.py:277-290), a (B, P) tensor per call.
"""
import math
from typing import List, Union

import torch


def to_local(points: torch.Tensor, pose_to_camera: torch.Tensor) -> torch.Tensor:
    """points (B, 3, n), part frames (B, P, 4, 4) -> (B, 3 P, n): R^T (p - t) per part (utils.py:13-32)."""
    rot_t = pose_to_camera[:, :, :3, :3].transpose(-1, -2)
    shifted = points.unsqueeze(1) - pose_to_camera[:, :, :3, 3:4]
    local = rot_t @ shifted
    return local.flatten(1, 2)


def in_cube(p: torch.Tensor) -> torch.Tensor:
    """|coordinate| <= 1 on all three axes (inclusive, utils.py:35-43): (B, 3, n) -> (B, 1, n); (B, 3 G, n) -> (B, G, n)."""
    ok = p.abs() <= 1
    if p.shape[1] == 3:
        return ok.all(dim=1, keepdim=True)
    return ok.unflatten(1, (-1, 3)).all(dim=2)


def _octaves(num_frequency: int, like: torch.Tensor) -> torch.Tensor:
    return torch.exp2(torch.arange(num_frequency, device=like.device)).to(like.dtype)


def positional_encoding(x: torch.Tensor, num_frequency: int, cos_first: bool = True, cat_dim: int = 2) -> torch.Tensor:
    """NeRF encoding of (B, d, n) -> (B, 2 F d, n) (utils.py:74-88); the phase is (x 2^f) pi, rounded in that order.
    cat_dim 2 interleaves per input dimension ([d][2F]), cat_dim 1 puts the trig function and frequency outermost ([2F][d])."""
    if cat_dim not in (1, 2):
        raise ValueError("cat_dim must be 1 or 2")
    B, _, n = x.shape
    freq = _octaves(num_frequency, x)
    if cat_dim == 2:
        phase = (x[:, :, None, :] * freq[None, None, :, None]) * math.pi
    else:
        phase = (x[:, None, :, :] * freq[None, :, None, None]) * math.pi
    pair = (phase.cos(), phase.sin()) if cos_first else (phase.sin(), phase.cos())
    return torch.cat(pair, dim=cat_dim).reshape(B, -1, n)


def multi_part_positional_encoding(value: Union[List, torch.Tensor], num_frequency: int, num_bone: int) -> torch.Tensor:
    """Per-part encoding for grouped layers (utils.py:46-71): value (B, P d, n) -> (B, P 2F d, n), sin block first;
    every channel of a part whose value leaves [-1, 1] is zeroed."""
    if isinstance(value, (list, tuple)):
        raise NotImplementedError("mip-NeRF (value, sigma) encoding is not on the tri-plane path")
    B, _, n = value.shape
    per_part = value.reshape(B * num_bone, -1, n)
    enc = positional_encoding(per_part, num_frequency, cos_first=False, cat_dim=1).reshape(B, num_bone, -1, n)
    outside = (value.reshape(B, num_bone, -1, n).abs() > 1).any(dim=2, keepdim=True)
    return (enc * (~outside).to(enc.dtype)).reshape(B, -1, n)
