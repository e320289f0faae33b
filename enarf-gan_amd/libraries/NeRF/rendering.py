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


_MLP_LEAVES = ("conv.weight", "conv.modulation.weight", "conv.modulation.bias", "bias")


class _RenderFunction(torch.autograd.Function):
    """Differentiable wrapper of the fused march: forward = enarf_prepare (MLP pack) + enarf_triplane_pack +
    enarf_render_fwd, backward = enarf_render_bwd + enarf_weight_grad (dW') + enarf_prepare_bwd + un-pack.
    Differentiable inputs: the tri-plane, z_rend and the 12 StyledMLP tensors (as in the reference, not the pose and
    not the importance samples)."""

    @staticmethod
    def forward(ctx, k, tri, z_rend, *params):
        mlp = {f"layers.{i}.{leaf}": params[4 * i + j] for i in range(3) for j, leaf in enumerate(_MLP_LEAVES)}
        tri_c = tri.detach().contiguous()
        feat_cl = ops.triplane_pack(tri_c)
        pack = k["pack_fn"](z_rend.detach(), {n: t.detach() for n, t in mlp.items()})
        out = ops.render_fwd(k["image_coord"], k["inv_intrinsics"], k["parts"], k["canonical_pose"], tri_c, feat_cl, pack,
                             k["Nc"], k["Nf"], render_scale=k["render_scale"], bins=k["bins"], seed=k["seed"],
                             mlp_mode=k["mlp_mode"], return_bins=True, **k["flags"])
        bins_used = out.taps["bins"]
        ctx.k = k
        ctx.save_for_backward(tri_c, feat_cl, pack, bins_used, z_rend.detach(), *[p.detach() for p in params])
        ctx.mark_non_differentiable(out.fine_weights, out.fine_depth, bins_used)
        return out.color, out.mask, out.disparity, out.fine_weights, out.fine_depth, bins_used

    @staticmethod
    def backward(ctx, g_color, g_mask, g_disp, _gfw, _gfd, _gb):
        k = ctx.k
        tri_c, feat_cl, pack, bins_used, z_rend = ctx.saved_tensors[:5]
        params = ctx.saved_tensors[5:]
        mlp = {f"layers.{i}.{leaf}": params[4 * i + j] for i in range(3) for j, leaf in enumerate(_MLP_LEAVES)}
        grad_tri, dW, db = ops.render_bwd(k["image_coord"], k["inv_intrinsics"], k["parts"], k["canonical_pose"], tri_c,
                                          feat_cl, pack, k["Nf"], bins_used, g_color, g_mask, g_disp,
                                          render_scale=k["render_scale"], **k["flags"])
        pg, dz = ops.prepare_bwd(z_rend, mlp, dW)
        grads = []
        for i in range(3):
            grads += [pg[f"layers.{i}.conv.weight"], pg[f"layers.{i}.conv.modulation.weight"],
                      pg[f"layers.{i}.conv.modulation.bias"], db[i].reshape(params[4 * i + 3].shape)]
        return (None, grad_tri, dz) + tuple(grads)


class _RenderFunctionCL(torch.autograd.Function):
    """The same march for producers that emit channel-last feature planes (the deformation-field warp): differentiable
    inputs are the NCHW tri-plane that holds the part-probability planes (1 or B images), the channel-last feature planes
    (B, 3, H, W, 32), z_rend and the StyledMLP tensors. No NCHW <-> channel-last copy in either direction."""

    @staticmethod
    def forward(ctx, k, tri, feat_cl, z_rend, *params):
        mlp = {f"layers.{i}.{leaf}": params[4 * i + j] for i in range(3) for j, leaf in enumerate(_MLP_LEAVES)}
        tri_c, feat_c = tri.detach().contiguous(), feat_cl.detach().contiguous()
        pack = k["pack_fn"](z_rend.detach(), {n: t.detach() for n, t in mlp.items()})
        out = ops.render_fwd(k["image_coord"], k["inv_intrinsics"], k["parts"], k["canonical_pose"], tri_c, feat_c, pack,
                             k["Nc"], k["Nf"], render_scale=k["render_scale"], bins=k["bins"], seed=k["seed"],
                             mlp_mode=k["mlp_mode"], return_bins=True, **k["flags"])
        bins_used = out.taps["bins"]
        ctx.k = k
        ctx.save_for_backward(tri_c, feat_c, pack, bins_used, z_rend.detach(), *[p.detach() for p in params])
        ctx.mark_non_differentiable(out.fine_weights, out.fine_depth, bins_used)
        return out.color, out.mask, out.disparity, out.fine_weights, out.fine_depth, bins_used

    @staticmethod
    def backward(ctx, g_color, g_mask, g_disp, _gfw, _gfd, _gb):
        k = ctx.k
        tri_c, feat_c, pack, bins_used, z_rend = ctx.saved_tensors[:5]
        params = ctx.saved_tensors[5:]
        mlp = {f"layers.{i}.{leaf}": params[4 * i + j] for i in range(3) for j, leaf in enumerate(_MLP_LEAVES)}
        grad_tri, dW, db, gfeat = ops.render_bwd(k["image_coord"], k["inv_intrinsics"], k["parts"], k["canonical_pose"],
                                                 tri_c, feat_c, pack, k["Nf"], bins_used, g_color, g_mask, g_disp,
                                                 render_scale=k["render_scale"], feat_grad_channel_last=True, **k["flags"])
        pg, dz = ops.prepare_bwd(z_rend, mlp, dW)
        grads = []
        for i in range(3):
            grads += [pg[f"layers.{i}.conv.weight"], pg[f"layers.{i}.conv.modulation.weight"],
                      pg[f"layers.{i}.conv.modulation.bias"], db[i].reshape(params[4 * i + 3].shape)]
        return (None, grad_tri, gfeat, dz) + tuple(grads)


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
    if pose_to_camera.requires_grad:
        raise NotImplementedError("Currently pose should not be differentiable")   # rendering.py:216-217
    assert pose_to_camera.shape[1] == model.num_bone
    if not hasattr(model, "buffers_tensors"):
        model.buffers_tensors = {}
    if _parts is None:
        _parts = _parts_from_part_poses(model, pose_to_camera, model_input["bone_length"])
    if seed is None:
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    cfg = model.config
    mult_w = bool(cfg.multiply_density_with_triplane_wieght)
    z_rend = model_input["z_rend"]
    params = model.mlp.as_dict()
    needs_grad = False
    cl_route = bool(getattr(model, "uses_warp", False)) and model_input.get("tri_plane_feature") is None
    if torch.is_grad_enabled():
        if cl_route:     # the producer emits channel-last planes: (part-probability planes NCHW, feature planes channel-last)
            tri_graph, feat_graph = model._tri_plane_pair_graph(model_input)
            needs_grad = feat_graph.requires_grad
        else:
            tri_graph = model._tri_plane_graph(model_input)    # tri-plane as the autograd graph sees it
        needs_grad = (needs_grad or tri_graph.requires_grad or z_rend.requires_grad or
                      any(p.requires_grad for p in params.values()))
    if needs_grad and return_intermediate:
        raise NotImplementedError("return_intermediate=True is served from the kernel's taps and is not differentiable; "
                                  "call it under torch.no_grad()")
    flags = model.kernel_flags() if hasattr(model, "kernel_flags") else dict(multiply_density_with_weight=mult_w)
    if needs_grad:
        k = dict(image_coord=image_coord.detach(), inv_intrinsics=inv_intrinsics.detach(), parts=_parts.detach(),
                 canonical_pose=model.canonical_pose, Nc=Nc, Nf=Nf, render_scale=float(render_scale), bins=bins, seed=seed,
                 mlp_mode=model.mlp_mode, pack_fn=model._mlp_pack_from, flags=flags)
        flat = [params[f"layers.{i}.{leaf}"] for i in range(3) for leaf in _MLP_LEAVES]
        if cl_route:
            color, mask, disparity, fw, fd, bins_used = _RenderFunctionCL.apply(k, tri_graph, feat_graph, z_rend, *flat)
            model.buffers_tensors.update(fine_weights=fw, fine_depth=fd, bins=bins_used, tri_plane_feature=None)
        else:
            color, mask, disparity, fw, fd, bins_used = _RenderFunction.apply(k, tri_graph, z_rend, *flat)
            model.buffers_tensors.update(fine_weights=fw, fine_depth=fd, bins=bins_used, tri_plane_feature=tri_graph)
        return color, mask, disparity
    tri, feat_cl = model._tri_plane_pair(model_input)
    if _pack is None:
        _pack = model._mlp_pack(z_rend)
    out = ops.render_fwd(image_coord, inv_intrinsics, _parts, model.canonical_pose, tri, feat_cl, _pack, Nc, Nf,
                         render_scale=render_scale, bins=bins, seed=seed, mlp_mode=model.mlp_mode,
                         return_bins=True, debug=return_intermediate, **flags)
    model.buffers_tensors["bins"] = out.taps["bins"]
    model.buffers_tensors["fine_weights"] = out.fine_weights      # (B, 1, n, Nf-1); zeros for dropped rays
    model.buffers_tensors["fine_depth"] = out.fine_depth          # (B, 1, n, Nf)
    if return_intermediate:
        # (fine_points (B, 3, n*Nf), fine_density (B, 1, n*Nf)) of rendering.py:291, for ALL n rays (the reference
        # compacts batch 1 to the rays that hit; dropped rays hold zero density here) - from the kernel's taps
        t = out.taps
        B, n = out.mask.shape
        coord = image_coord.reshape(B, 3, n).to(torch.float32)
        Ki = inv_intrinsics.to(torch.float32)
        if Ki.dim() == 2:
            Ki = Ki[None].expand(B, -1, -1)
        ray = torch.einsum("bij,bjn->bin", Ki, coord)
        b_ = t["bins"][:, None]                                                # (B, 1, n, Nf)
        start, end = (t["depth_min"][:, None] * ray)[..., None], (t["depth_max"][:, None] * ray)[..., None]
        fine_points = (start * (1 - b_) + end * b_).reshape(B, 3, n * Nf)
        fine_density = t["fine_density"].reshape(B, 1, n * Nf)
        model.buffers_tensors["fine_density"] = fine_density
        return out.color, out.mask, out.disparity, (fine_points, fine_density)
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
