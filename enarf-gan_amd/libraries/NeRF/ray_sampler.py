"""Ray samplers with the reference's signatures (libraries/NeRF/ray_sampler.py:7-67), on the input's device
instead of the literal "cuda" (SURVEY Q13). Plain torch: they run once per step and are not on the timed path."""
from typing import Tuple

import torch
import torch.nn.functional as F


def mask_based_sampler(mask: torch.Tensor, ray_batchsize: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Top-k of (129x129-dilated mask + U[0,1)): ray_idx (B, n), homo_img (B, 1, 3, n)."""
    batchsize, h, w = mask.shape
    pad_size = 64
    m = F.max_pool2d(mask.float()[:, None], pad_size * 2 + 1, stride=1, padding=pad_size)[:, 0]
    m = m.reshape(batchsize, h * w)
    m = m + torch.empty_like(m).uniform_()
    ray_idx = torch.topk(m, ray_batchsize, dim=1, sorted=False)[1]
    x, y = ray_idx % w, torch.div(ray_idx, w, rounding_mode="floor")
    rays = (torch.stack([x, y], dim=2) + 0.5).permute(0, 2, 1)
    homo_img = torch.cat([rays, torch.ones(batchsize, 1, ray_batchsize, device=mask.device)], dim=1)
    return ray_idx, homo_img.reshape(batchsize, 1, 3, -1)


def whole_image_grid_ray_sampler(render_size: int, patch_size: int, batchsize: int, device="cuda"
                                 ) -> Tuple[torch.Tensor, torch.Tensor]:
    """grid (B, patch, patch, 2) in [-1, 1] and homo_img (B, 1, 3, patch^2) of pixel centres."""
    y, x = torch.meshgrid([torch.arange(patch_size, device=device), torch.arange(patch_size, device=device)],
                          indexing="ij")
    rays = torch.stack([x, y], dim=2)[None]
    rays = render_size * (rays + 0.5) / patch_size
    rays = rays.repeat(batchsize, 1, 1, 1)
    grid = rays / (render_size / 2) - 1
    rays = rays.reshape(batchsize, -1, 2).permute(0, 2, 1)
    homo_img = torch.cat([rays, torch.ones(batchsize, 1, patch_size ** 2, device=device)], dim=1)
    return grid, homo_img.reshape(batchsize, 1, 3, -1)
