"""Stand-alone tri-plane sampling with the reference's names and argument meaning (libraries/triplane/sampling.py:9-127):
`sample_feature`, `sample_triplane_part_prob`, `sample_weighted_feature_v2`.

On the render path these three never run: the fused kernels (enarf_query_fwd / enarf_render_fwd) do their arithmetic per
(part, sample) pair in registers. They exist for callers of the reference API that use them directly. The memory-bound
part - the bilinear gathers and their gradients - is the HIP operator (`enarf_triplane_sample_fwd` for the plain sum,
`enarf_triplane_sample_ex_*` for separate planes / per-point images); the small elementwise tails (sigmoid, product,
softmax, weighting) are torch on the device. Inputs of any floating dtype are computed in fp32 and returned in the
INPUT's dtype, as the reference operator does (TriplaneSampler.cpp:20, kernel.cu:244).
"""
from typing import Optional

import torch

from ... import ops
from ...cuda_extension.triplane_sampler import triplane_sampler


class _GatherPlanes(torch.autograd.Function):
    """planes (Bimg, 3C, H, W), position (B, 3, n) [, point_image (n,)] -> (B, 3, C, n): the three planes' bilinear samples"""

    @staticmethod
    def forward(ctx, planes, position, point_image):
        grid = position.detach().float().permute(0, 2, 1).contiguous()
        out = ops.triplane_sample_ex_fwd(planes.detach(), grid, separate=True, point_image=point_image)
        ctx.save_for_backward(planes, grid)
        ctx.point_image = point_image
        return out

    @staticmethod
    def backward(ctx, g):
        planes, grid = ctx.saved_tensors
        gi, gg = ops.triplane_sample_ex_bwd(g.contiguous(), planes.detach(), grid, True, ctx.point_image,
                                            ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        if gi is not None:
            gi = gi.to(planes.dtype)
        return gi, (None if gg is None else gg.permute(0, 2, 1)), None


def sample_feature(tri_plane_features: torch.Tensor, position: torch.Tensor, reduction: str = "sum", clamp_mask: bool = False,
                   batch_idx: Optional[torch.Tensor] = None) -> torch.Tensor:
    """tri_plane_features (B, 3C, h, w), position (B, 3, n) in [-1, 1] -> (B, C, n).

    reduction "sum": sum of the three planes' samples; "prod": product of their sigmoids (`clamp_mask`: values clamped to
    [-2, 5] first, gradient passed straight through, sampling.py:46-47). `batch_idx` (n,), with B == 1: the reference's
    planes-side-by-side form (sampling.py:34-38, input (1, 3C, h, (h + 1) * Bimg)) - here point i samples image
    batch_idx[i] of the un-concatenated planes, which is what that form computes."""
    B, _, h, w = tri_plane_features.shape
    assert B == 1 or batch_idx is None
    if reduction not in ("sum", "prod"):
        raise ValueError()
    dtype = tri_plane_features.dtype
    if batch_idx is None and reduction == "sum":
        grid = position.permute(0, 2, 1).contiguous()[:, :, None, :]
        return triplane_sampler(tri_plane_features, grid)[:, :, :, 0].to(dtype)
    planes = tri_plane_features
    if batch_idx is not None:                         # (1, 3C, h, (h + 1) * Bimg) -> (Bimg, 3C, h, h): drop the pad column
        n_img = w // (h + 1)
        planes = planes.reshape(-1, h, n_img, h + 1)[..., :h].permute(2, 0, 1, 3).contiguous()
    per_plane = _GatherPlanes.apply(planes, position, batch_idx)                    # (B, 3, C, n)
    if reduction == "sum":
        return per_plane.sum(dim=1).to(dtype)
    if clamp_mask:
        per_plane = (per_plane.detach().clamp(-2, 5) - per_plane.detach()) + per_plane
    return torch.sigmoid(per_plane).prod(dim=1).to(dtype)


def sample_triplane_part_prob(tri_plane_weights: torch.Tensor, position: torch.Tensor, position_validity: torch.Tensor,
                              mode: str = "prod", clamp_mask: bool = False) -> torch.Tensor:
    """tri_plane_weights (B * P, 3, h, w), position (B, P, 3, n), validity (B, P, n) bool -> part probability (B, P, n):
    "prod" (the default everywhere), "sum" + softmax over parts with invalid pairs pushed to -1e4, else uniform 1 / P."""
    B, P, _, n = position.shape
    flat = position.reshape(B * P, 3, n)
    if mode == "prod":
        return sample_feature(tri_plane_weights, flat, clamp_mask=clamp_mask, reduction="prod").reshape(B, P, n)
    if mode == "sum":
        logit = sample_feature(tri_plane_weights, flat, clamp_mask=clamp_mask).reshape(B, P, n)
        return torch.softmax(logit - (~position_validity) * 1e4, dim=1)
    return torch.full((B, P, n), 1.0 / P, device=position.device)


def sample_weighted_feature_v2(feat_dim: int, tri_plane_features: torch.Tensor, position: torch.Tensor, weight: torch.Tensor,
                               position_validity: torch.Tensor, clamp_mask: bool = False) -> torch.Tensor:
    """sum over the VALID parts of weight[b, k, i] * tri-plane feature at position[b, k, :, i]: (B, feat_dim, n).
    tri_plane_features (B, 3 * feat_dim, h, w); only valid (part, point) pairs are gathered, each from its own image."""
    B, P, n = position_validity.shape
    assert position_validity.dtype == torch.bool
    out = torch.zeros(B, feat_dim, n, device=position.device, dtype=torch.float32)
    pair = torch.nonzero(position_validity.reshape(-1)).reshape(-1)                 # flat (b, k, i) ids of the valid pairs
    if pair.numel() == 0:
        return out
    image = torch.div(pair, P * n, rounding_mode="floor")
    point = pair % n
    pos = position.permute(0, 1, 3, 2).reshape(B * P * n, 3)[pair].t()[None]        # (1, 3, V)
    per_plane = _GatherPlanes.apply(tri_plane_features, pos, image.to(torch.int32))  # (1, 3, feat_dim, V)
    value = per_plane.sum(dim=1)[0] * weight.reshape(-1)[pair][None]                # (feat_dim, V)
    out = out.permute(1, 0, 2).reshape(feat_dim, B * n).index_add(1, image * n + point, value.float())
    return out.reshape(feat_dim, B, n).permute(1, 0, 2).contiguous()
