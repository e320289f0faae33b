"""Sharding of the render path across the GPUs of one node.

Rays and frames are independent: the forward path has NO exchange step, so ranks only agree on who renders
what. Frames are dealt contiguously (each rank keeps its own tri-planes, poses and MLP packs); a single frame
is cut into contiguous ray ranges. The only collective is the one a caller wants for the outputs
(`all_gather_rays`), plus barriers / a MAX-reduce of wall time in bench.py.
"""
from __future__ import annotations

from typing import List, Tuple

import torch


def split_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """[start, stop) of `rank` when `total` items are dealt contiguously, sizes differing by at most one."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    q, r = divmod(total, world)
    start = rank * q + min(rank, r)
    return start, start + q + (1 if rank < r else 0)


def frames_for_rank(num_frames: int, rank: int, world: int) -> range:
    return range(*split_range(num_frames, rank, world))


def rays_for_rank(num_rays: int, rank: int, world: int) -> slice:
    return slice(*split_range(num_rays, rank, world))


def near_far_inputs_are_global(pose_to_camera: torch.Tensor) -> torch.Tensor:
    """The reference's near/far planes are min/max over the WHOLE batch of part centres (rendering.py:15-17).
    When frames of one logical batch are sharded, every rank must therefore hand the kernel the full batch of
    part frames for that reduction; this helper only documents and checks that contract."""
    if pose_to_camera.dim() != 4:
        raise ValueError("pose_to_camera must be (B, J, 4, 4)")
    return pose_to_camera


def all_gather_rays(local: torch.Tensor, num_rays: int, group=None) -> torch.Tensor:
    """Reassemble (..., n_local) per-rank ray outputs into (..., num_rays) on every rank (ragged ranges allowed)."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    sizes = [split_range(num_rays, r, world) for r in range(world)]
    maxn = max(b - a for a, b in sizes)
    pad = torch.zeros(*local.shape[:-1], maxn, dtype=local.dtype, device=local.device)
    pad[..., :local.shape[-1]] = local
    bufs: List[torch.Tensor] = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    return torch.cat([bufs[r][..., :sizes[r][1] - sizes[r][0]] for r in range(world)], dim=-1)
