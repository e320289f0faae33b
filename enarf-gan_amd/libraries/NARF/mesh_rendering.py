"""Density sweep behind `create_mesh` (libraries/NARF/mesh_rendering.py:50-81 of the reference).

The reference builds the (2/voxel_size + 1)^3 grid on the host, pushes it through
`calc_density_and_color_from_camera_coord_v2` in `render_bs` chunks and hands the volume to `mcubes`. Here the lattice
is generated inside `enarf_query_fwd` (lattice mode, density only, one launch; or chunk by chunk from device tensors when
`chunk` is given); the volume stays on the device. Marching cubes (`mcubes`) and the rasteriser (`pytorch3d`) are third-party and not part of this package:
`create_mesh` raises ImportError where the reference would, after the volume is available from `density_volume`.
"""
from __future__ import annotations

from typing import Dict

import torch

from ... import ops


def _grid_chunk(D: int, start: int, stop: int, center: torch.Tensor, scale: float, dev: torch.device) -> torch.Tensor:
    """Points [start, stop) of stack(meshgrid(bins, bins, bins)).reshape(1, 3, -1), bins = arange(-c, c + 1) / c."""
    c = (D - 1) // 2
    idx = torch.arange(start, stop, device=dev, dtype=torch.int64)
    ix = torch.div(idx, D * D, rounding_mode="floor")
    iy = torch.div(idx, D, rounding_mode="floor") % D
    iz = idx % D
    p = torch.stack([ix, iy, iz], dim=0).to(torch.float32)
    p = (p - c) / c                                           # == torch.arange(-c, c + 1) / c, elementwise
    return ((p[None] + center.to(dev).reshape(1, 3, 1)) * scale).contiguous()


@torch.no_grad()
def density_volume(model, pose_to_camera: torch.Tensor, center: torch.Tensor, voxel_size: float = 0.003,
                   model_input: Dict = {}, chunk: int = 1 << 31) -> torch.Tensor:
    """(D, D, D) density grid, D = 2 * int(1 / voxel_size) + 1, exactly the tensor the reference feeds to marching
    cubes. pose_to_camera (1, P, 4, 4): part frames with UNSCALED translation (scaled here, on a copy - the reference
    scales its argument in place)."""
    cube = int(1 / voxel_size)
    D = 2 * cube + 1
    dev = pose_to_camera.device
    pose = pose_to_camera.clone()
    if model.coordinate_scale != 1:
        pose[:, :, :3, 3] *= model.coordinate_scale
    B, P = pose.shape[:2]
    assert B == 1, "create_mesh sweeps one pose"
    parts = torch.zeros(B, P, 16, dtype=torch.float32, device=dev)
    parts[:, :, :9] = pose[:, :, :3, :3].reshape(B, P, 9)
    parts[:, :, 9:12] = pose[:, :, :3, 3]
    parts[:, :, 12] = (model.canonical_bone_length[:, None] / model_input["bone_length"] / model.coordinate_scale)[:, :, 0]
    tri, feat_cl = model._tri_plane_pair(model_input)
    pack = model._mlp_pack(model_input["z_rend"])
    flags = model.kernel_flags()
    total = D * D * D
    if chunk >= total and total < 2 ** 31:          # one launch, the lattice generated in the kernel (no point tensor)
        den, _ = ops.query_fwd(None, parts, model.canonical_pose, tri, feat_cl, pack, mlp_mode=model.mlp_mode,
                               need_color=False, **flags,
                               grid=(D, center.reshape(3).tolist(), float(model.coordinate_scale)))
        return den.reshape(D, D, D)
    out = torch.empty(total, dtype=torch.float32, device=dev)
    for s in range(0, total, chunk):
        e = min(s + chunk, total)
        pts = _grid_chunk(D, s, e, center, float(model.coordinate_scale), dev)
        den, _ = ops.query_fwd(pts, parts, model.canonical_pose, tri, feat_cl, pack, mlp_mode=model.mlp_mode,
                               need_color=False, **flags)
        out[s:e] = den.reshape(-1)
    return out.reshape(D, D, D)


def create_mesh(model, pose_to_camera, center, voxel_size=0.003, mesh_th=15, model_input={}):
    """mesh_rendering.py:50-81: density sweep + marching cubes -> (vertices, triangles, textures). The third-party
    imports come first, as in the reference (:52 and the module imports), so a missing PyMCubes / pytorch3d fails
    before any GPU work; `density_volume` is the sweep on its own."""
    try:
        import mcubes
        from pytorch3d.renderer import Textures
    except ImportError as e:      # same third-party requirements as the reference
        raise ImportError("create_mesh needs PyMCubes and pytorch3d (as the reference does); the density grid itself "
                          "is available from density_volume()") from e
    density = density_volume(model, pose_to_camera, center, voxel_size, model_input)
    cube = int(1 / voxel_size)
    dev = pose_to_camera.device
    vertices, triangles = mcubes.marching_cubes(density.cpu().numpy(), mesh_th)
    vertices = torch.tensor((vertices - cube) * voxel_size, device=dev).float() + center[:, :, 0]
    triangles = torch.tensor(triangles.astype("int64")).to(dev)
    return vertices, triangles, Textures(verts_rgb=torch.ones_like(vertices)[None])


def render_mesh_(meshes, intrinsics, img_size, render_size=512):
    """mesh_rendering.py:17-47: Phong-shaded rasterisation of (vertices, triangles, textures) with pytorch3d. The
    rasteriser is third-party and outside the hot path (out of scope); this raises ImportError where the
    reference's module import would."""
    try:
        import pytorch3d.renderer  # noqa: F401
    except ImportError as e:
        raise ImportError("render_mesh_ needs pytorch3d (as the reference does)") from e
    raise NotImplementedError("the pytorch3d rasteriser call is not rebuilt here (SURVEY.md §8, out of scope); "
                              "use create_mesh() / density_volume() and rasterise with pytorch3d directly")
