"""Mask regularisers of the GAN's generator loss (models/loss.py:5-30): the lowest `background_ratio` of the rendered
foreground-mask values are pushed to zero, and the mask is pulled to one where the projected skeleton is
(`bone_mask` > 0.5, max-pooled down to the mask's resolution)."""
import torch
import torch.nn.functional as F


def push_to_background(fake_mask: torch.Tensor, background_ratio: float = 0.3):
    if background_ratio <= 0:
        return 0
    flat = fake_mask.reshape(-1)
    lowest = torch.topk(flat, k=int(flat.numel() * background_ratio), largest=False, sorted=False)[0]
    return lowest.square().mean()


def nerf_bone_loss(fake_mask: torch.Tensor, bone_mask: torch.Tensor) -> torch.Tensor:
    assert fake_mask.ndim == bone_mask.ndim
    if fake_mask.shape[-1] != bone_mask.shape[-1]:
        rate = bone_mask.shape[-1] // fake_mask.shape[-1]
        bone_mask = F.max_pool2d(bone_mask[:, None], rate, rate, 0)[:, 0]
    on_bone = bone_mask > 0.5
    return ((1 - fake_mask).square() * on_bone).sum() / on_bone.sum()


def nerf_patch_loss(fake_mask, bone_mask, background_ratio=0.3, coef=10):
    return (push_to_background(fake_mask, background_ratio) + nerf_bone_loss(fake_mask, bone_mask)) * coef
