"""Ray samplers behind the reference's two entry points (libraries/NeRF/ray_sampler.py:7-39 `mask_based_sampler`,
:42-67 `whole_image_grid_ray_sampler`), on the device of their input (the reference hard-codes "cuda").

`mask_based_sampler` keeps the reference's law - the `ray_batchsize` largest values of (mask dilated by a 129 x 129
window) + U[0, 1) - and runs on the device through `ops.mask_dilate_topk` (HIP: separable window maximum + per-image
radix select); `noise` may be passed in to replay a draw. `whole_image_grid_ray_sampler` is closed-form index arithmetic.
"""
from typing import Optional, Tuple

import torch

DILATE_RADIUS = 64      # 129 x 129 window (ray_sampler.py:23-24)


def _homogeneous(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """pixel coordinates (B, n) each -> (B, 1, 3, n) rows (x, y, 1)"""
    return torch.stack((x, y, torch.ones_like(x)), dim=1).unsqueeze(1)


def mask_based_sampler(mask: torch.Tensor, ray_batchsize: int, noise: Optional[torch.Tensor] = None
                       ) -> Tuple[torch.Tensor, torch.Tensor]:
    """mask (B, h, w) -> ray_idx (B, n) int64 (flat pixel ids, unordered) and homo_img (B, 1, 3, n) pixel centres."""
    from ... import ops
    B, h, w = mask.shape
    if noise is None:
        noise = torch.rand(B, h * w, device=mask.device)
    ray_idx = ops.mask_dilate_topk(mask, noise.reshape(B, h * w), ray_batchsize, DILATE_RADIUS)
    col = (ray_idx % w).to(torch.float32) + 0.5
    row = torch.div(ray_idx, w, rounding_mode="floor").to(torch.float32) + 0.5
    return ray_idx, _homogeneous(col, row)


def whole_image_grid_ray_sampler(render_size: int, patch_size: int, batchsize: int, device="cuda"
                                 ) -> Tuple[torch.Tensor, torch.Tensor]:
    """Every pixel of a patch_size^2 image that covers a render_size^2 frame: grid (B, S, S, 2) in [-1, 1] and
    homo_img (B, 1, 3, S^2), row-major pixels."""
    centre = (torch.arange(patch_size, device=device, dtype=torch.float32) + 0.5) * render_size / patch_size
    flat = torch.arange(patch_size * patch_size, device=device)
    col, row = centre[flat % patch_size], centre[torch.div(flat, patch_size, rounding_mode="floor")]
    xy = torch.stack((col, row), dim=-1).reshape(1, patch_size, patch_size, 2).expand(batchsize, -1, -1, -1)
    grid = xy / (render_size / 2) - 1
    homo = _homogeneous(col[None], row[None]).expand(batchsize, -1, -1, -1)
    return grid.contiguous(), homo.contiguous()
