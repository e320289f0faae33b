"""`render` / `render_entire_img` with the reference's signatures (libraries/NeRF/rendering.py:227-427).

The whole body of the reference's render() - frustum range, coarse pass, importance sampling, fine pass,
compositing - is ONE launch of the fused HIP ray march (enarf_render_fwd); nothing is computed in torch
except packing the already-transformed part frames into the kernel's 16-float records.
"""
from typing import Dict, Optional

import torch

from ... import ops


def _parts_from_part_poses(model, pose_to_camera: torch.Tensor, bone_length: torch.Tensor) -> torch.Tensor:
    """(B, P, 4, 4) part frames (unscaled translation) -> the kernel's (B, P, 16) records.

    Same arithmetic as rendering.py:258-260 (t *= coordinate_scale) and narf.py:165 (cbl / bl / cs)."""
    B, P = pose_to_camera.shape[:2]
    cs = model.coordinate_scale
    parts = torch.zeros(B, P, 16, dtype=torch.float32, device=pose_to_camera.device)
    parts[:, :, :9] = pose_to_camera[:, :, :3, :3].reshape(B, P, 9)
    parts[:, :, 9:12] = pose_to_camera[:, :, :3, 3] * cs if cs != 1 else pose_to_camera[:, :, :3, 3]
    parts[:, :, 12] = (model.canonical_bone_length[:, None] / bone_length / cs)[:, :, 0]
    return parts


def render(model, image_coord: torch.Tensor, pose_to_camera: torch.Tensor, inv_intrinsics: torch.Tensor,
           render_scale: float = 1, Nc: int = 64, Nf: int = 128, semantic_map: bool = False,
           return_intermediate: bool = False, camera_pose: Optional[torch.Tensor] = None,
           model_input: Dict = {}, _parts: Optional[torch.Tensor] = None, _pack: Optional[torch.Tensor] = None,
           bins: Optional[torch.Tensor] = None, seed: Optional[int] = None):
    """image_coord (B, 1, 3, n); pose_to_camera (B, P, 4, 4) part frames -> color (B,3,n), mask (B,n), disparity (B,n).

    Extra keyword arguments (not in the reference): `bins` (B, n, Nf) replays given importance samples;
    `seed` seeds the in-kernel Philox draw (default: a fresh 63-bit number from torch's CPU generator)."""
    if semantic_map:
        raise AssertionError("semantic map rendering will be implemented later")   # rendering.py:298
    if return_intermediate:
        raise NotImplementedError("return_intermediate=True (fine_points / fine_density of rendering.py:291) is not "
                                  "exposed by the fused kernel")
    if pose_to_camera.requires_grad:
        raise NotImplementedError("Currently pose should not be differentiable")   # rendering.py:216-217
    if torch.is_grad_enabled() and any(p.requires_grad for p in model.parameters()):
        raise NotImplementedError("backward of the fused renderer is not implemented yet (SURVEY.md §8f rank 1): "
                                  "call under torch.no_grad()")
    assert pose_to_camera.shape[1] == model.num_bone
    if not hasattr(model, "buffers_tensors"):
        model.buffers_tensors = {}
    tri, feat_cl = model._tri_plane_pair(model_input)
    if _parts is None:
        _parts = _parts_from_part_poses(model, pose_to_camera, model_input["bone_length"])
    if _pack is None:
        _pack = model._mlp_pack(model_input["z_rend"])
    if seed is None:
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    cfg = model.config
    out = ops.render_fwd(image_coord, inv_intrinsics, _parts, model.canonical_pose, tri, feat_cl, _pack, Nc, Nf,
                         render_scale=render_scale, bins=bins, seed=seed, mlp_mode=model.mlp_mode,
                         multiply_density_with_weight=bool(cfg.multiply_density_with_triplane_wieght))
    model.buffers_tensors["fine_weights"] = out.fine_weights      # (B, 1, n, Nf-1); zeros for dropped rays
    model.buffers_tensors["fine_depth"] = out.fine_depth          # (B, 1, n, Nf)
    return out.color, out.mask, out.disparity


def render_entire_img(model, pose_to_camera: torch.Tensor, inv_intrinsics: torch.Tensor,
                      camera_pose: Optional[torch.Tensor] = None, render_size: int = 128, Nc: int = 64,
                      Nf: int = 128, semantic_map: bool = False, use_normalized_intrinsics: bool = False,
                      no_grad: bool = True, model_input: Dict = {}, bbox=None):
    """Whole frame of image 0 (rendering.py:362-427) -> (3,H,W), (H,W), (H,W). One launch: the reference's
    render_bs chunk loop exists to bound its (B,P,3,n*N) temporaries, which the fused kernel never creates."""
    if bbox is not None:
        render_width, render_height = bbox[2] - bbox[0], bbox[3] - bbox[1]
        x_offset, y_offset = bbox[0], bbox[1]
    else:
        render_width, render_height = render_size, render_size
        x_offset, y_offset = 0, 0
    dev = pose_to_camera.device
    idx = torch.arange(render_width * render_height, device=dev)
    x = (idx % render_width + 0.5 + x_offset).float()
    y = (torch.div(idx, render_width, rounding_mode="floor") + 0.5 + y_offset).float()
    if use_normalized_intrinsics:
        x, y = x / render_size, y / render_size
    img_coord = torch.stack([x, y, torch.ones_like(x)], dim=0)[None, None]
    mi = dict(model_input)
    mi["bone_length"] = mi["bone_length"][:1]
    if mi.get("z_rend") is not None:
        mi["z_rend"] = mi["z_rend"][:1]
    if mi.get("tri_plane_feature") is not None:
        mi["tri_plane_feature"] = mi["tri_plane_feature"][:1]
    with torch.set_grad_enabled(not no_grad):
        color, mask, disparity = render(model, img_coord, pose_to_camera[:1], inv_intrinsics, Nc=Nc, Nf=Nf,
                                        camera_pose=camera_pose, model_input=mi)
    return (color.reshape(3, render_height, render_width), mask.reshape(render_height, render_width),
            disparity.reshape(render_height, render_width))
