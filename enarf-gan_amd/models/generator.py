"""Generator shells around the HIP renderer: `TriNARFGenerator` (GAN) and `DSONARFGenerator` (single dynamic scene).

They keep the constructor arguments, method names, argument order and return tuples of the reference's
`models/generator.py` (TriNARFGenerator :14-118, DSONARFGenerator :182-300) so that its train / demo scripts can switch
over; everything inside is this package's own plumbing around `TriPlaneNARF`.

The background network is the StyleGAN2 skip generator of `libraries/custom_stylegan2/net.py` (generator.py:32-38:
n_mlp 4, `crop_background` from the config, or the pretrained church generator), built on this repo's HIP ops; with
`black_background=True` / `black_bg_if_possible=True` nothing of it runs.
"""
from typing import Optional

import numpy as np
import torch
from torch import nn

from ..libraries.custom_stylegan2.net import Generator as StyleGANGenerator
from ..libraries.custom_stylegan2.net import PretrainedStyleGAN
from ..libraries.NeRF.ray_sampler import mask_based_sampler, whole_image_grid_ray_sampler
from .narf import TriPlaneNARF


class _RendererShell(nn.Module):
    """What both generators share: the config / size bookkeeping and the tri-plane NARF they wrap."""

    def __init__(self, config, size, num_bone, parent_id, num_bone_param, z_dim, **nerf_kwargs):
        super().__init__()
        self.config, self.size, self.num_bone = config, size, num_bone
        params = config.nerf_params
        # with the head as an extra part every joint has a bone-length parameter (generator.py:30-31, :198-199)
        n_param = num_bone if params.origin_location == "center+head" else num_bone_param
        self.nerf = TriPlaneNARF(params, z_dim=z_dim, num_bone=num_bone, bone_length=True, parent=parent_id,
                                 num_bone_param=n_param, **nerf_kwargs)

    @property
    def memory_cost(self):
        return self.nerf.memory_cost

    @property
    def flops(self):
        return self.nerf.flops

    def register_canonical_pose(self, pose: np.ndarray):
        self.nerf.register_canonical_pose(pose)

    @property
    def _samples(self):
        p = self.config.nerf_params
        return dict(Nc=p.Nc, Nf=p.Nf)


class TriNARFGenerator(_RendererShell):
    def __init__(self, config, size, num_bone=1, parent_id=None, num_bone_param=None, black_background=False):
        super().__init__(config, size, num_bone, parent_id, num_bone_param, z_dim=[2 * config.z_dim, config.z_dim])
        self.ray_sampler = whole_image_grid_ray_sampler
        self.background_ratio = config.background_ratio
        self.black_background = black_background
        self.background_generator = None
        if not black_background:                   # generator.py:32-38
            if config.pretrained_background:
                self.background_generator = PretrainedStyleGAN()
            else:
                self.background_generator = StyleGANGenerator(size=size, style_dim=config.z_dim, n_mlp=4, last_channel=3,
                                                              crop_background=config.crop_background)

    def normalized_inv_intrinsics(self, intrinsics: torch.Tensor):
        bottom = intrinsics.new_tensor([[0, 0, 1]])
        return torch.linalg.inv(torch.cat([intrinsics[:2] / self.size, bottom], dim=0))

    def _latent_parts(self, z: torch.Tensor):
        """z = [tri-plane (2u) | renderer (u) | background (u, absent on black)] -> (z_nerf, z_render, z_bg or None)."""
        shares = 3 if self.black_background else 4
        u = z.shape[1] // shares
        pieces = torch.split(z, [2 * u, u] + [u] * (shares - 3), dim=1)
        return pieces[0], pieces[1], (pieces[2] if shares == 4 else None)

    def _background(self, z_bg, z_render, force_black: bool):
        if self.black_background or force_black:
            return -1
        bg = self.background_generator
        image, _ = bg([z_bg, z_render], inject_index=bg.n_latent - 4)
        return image

    def forward(self, pose_to_camera, pose_to_world, bone_length, z=None, inv_intrinsics=None,
                return_intermediate=False, truncation_psi=1, black_bg_if_possible=False, return_disparity=False,
                return_bg=False):
        if self.num_bone != 1 and (bone_length is None or pose_to_camera is None):
            raise AssertionError("bone_length and pose_to_camera are required")
        B, S = pose_to_camera.shape[0], self.size
        _, pixels = self.ray_sampler(S, S, B, device=pose_to_camera.device)
        z_nerf, z_render, z_bg = self._latent_parts(z)
        K_inv = torch.as_tensor(inv_intrinsics).float().to(pixels.device)
        rendered = self.nerf(B, pixels, pose_to_camera, K_inv, z_nerf, z_render, bone_length,
                             return_intermediate=return_intermediate, truncation_psi=truncation_psi,
                             return_disparity=return_disparity, **self._samples)
        person = rendered[0].reshape(B, 3, S, S)
        alpha = rendered[1].reshape(B, S, S)
        backdrop = self._background(z_bg, z_render, black_bg_if_possible)
        image = person + (1 - alpha[:, None]) * backdrop
        if return_intermediate:                      # (fine_points, fine_density) ride along as the last item
            points, density = rendered[-1]
            return image, alpha, points, density
        if return_disparity:                         # back to metric units
            return image, alpha, rendered[2] * self.config.nerf_params.coordinate_scale
        if return_bg:
            return person, alpha, backdrop
        side = self.nerf.buffers_tensors
        return image, alpha, side["fine_weights"], side["fine_depth"]


    def render_mesh(self, pose_to_camera, intrinsics, z, bone_length, voxel_size=0.003, mesh_th=15, truncation_psi=0.4):
        """models/generator.py:120-129."""
        z_nerf, z_render, _ = self._latent_parts(z)
        return self.nerf.render_mesh(pose_to_camera, intrinsics, z_nerf, z_render, bone_length, voxel_size, mesh_th,
                                     truncation_psi, self.size)

    def create_mesh(self, pose_to_camera, z, bone_length, voxel_size=0.003, mesh_th=15, truncation_psi=0.4):
        """models/generator.py:131-140 (whose call passes arguments `create_mesh` does not take and cannot run as
        written): the same sweep as render_mesh, returning (vertices, triangles, textures)."""
        from ..libraries.NARF.mesh_rendering import create_mesh
        z_nerf, z_render, _ = self._latent_parts(z)
        center, pose_parts, model_input = self.nerf._mesh_inputs(pose_to_camera, z_nerf, z_render, bone_length,
                                                                 truncation_psi)
        return create_mesh(self.nerf, pose_parts, center=center, voxel_size=voxel_size, mesh_th=mesh_th,
                           model_input=model_input)

    def density_volume(self, pose_to_camera, z, bone_length, voxel_size=0.003, truncation_psi=0.4):
        """The density grid behind render_mesh / create_mesh, without the third-party marching cubes."""
        z_nerf, z_render, _ = self._latent_parts(z)
        return self.nerf.density_volume(pose_to_camera, z_nerf, z_render, bone_length, voxel_size, truncation_psi)


class DSONARFGenerator(_RendererShell):
    def __init__(self, config, size, num_bone=1, parent_id=None, num_bone_param=None):
        if not config.use_triplane:
            raise NotImplementedError("MLPNARF (use_triplane: False) is not the tri-plane path (SURVEY.md §2, OUT)")
        params = config.nerf_params
        latent = (20 if params.time_conditional else 0) + (9 * (num_bone - 1) if params.pose_conditional else 0)
        super().__init__(config, size, num_bone, parent_id, num_bone_param, z_dim=latent,
                         view_dependent=not params.no_ray_direction)
        self.ray_sampler = mask_based_sampler
        self.time_conditional, self.pose_conditional = params.time_conditional, params.pose_conditional

    @staticmethod
    def positional_encoding(x: torch.Tensor, num_frequency: int) -> torch.Tensor:
        """(B,) -> (B, 2 F): cos then sin of x * 2^f * pi."""
        octaves = torch.exp2(torch.arange(num_frequency, device=x.device).float())
        phase = x[:, None] * octaves * np.pi          # (x 2^f) pi, in this order
        return torch.cat([phase.cos(), phase.sin()], dim=1)

    @staticmethod
    def pose_encoding(pose: torch.Tensor):
        """Joint rotations relative to the root, flattened: (B, J, 4, 4) -> (B, 9 (J - 1))."""
        root_T = pose[:, :1, :3, :3].transpose(-1, -2)
        return (root_T @ pose[:, 1:, :3, :3]).flatten(1)

    def get_latents(self, frame_time: torch.Tensor, pose_to_camera: torch.Tensor):
        codes = []
        if self.time_conditional:
            codes.append(self.positional_encoding(frame_time, num_frequency=10))
        if self.pose_conditional:
            codes.append(self.pose_encoding(pose_to_camera))
        if not codes:
            raise AssertionError("neither time_conditional nor pose_conditional")
        z = torch.cat(codes, dim=1)
        return z, z

    def forward(self, pose_to_camera, camera_pose, mask, frame_time, bone_length, inv_intrinsics,
                background: Optional[float] = None):
        if bone_length is None or pose_to_camera is None or not isinstance(inv_intrinsics, torch.Tensor):
            raise AssertionError("bone_length, pose_to_camera and a tensor inv_intrinsics are required")
        ray_idx, pixels = self.ray_sampler(mask, self.config.ray_batchsize)
        z_tri, z_render = self.get_latents(frame_time, pose_to_camera)
        color, alpha = self.nerf(pose_to_camera.shape[0], pixels, pose_to_camera, inv_intrinsics, z_tri, z_render,
                                 bone_length, return_intermediate=False, camera_pose=camera_pose, **self._samples)
        backdrop = -1 if background is None else background
        return color + backdrop * (1 - alpha[:, None]), alpha, ray_idx

    def render_entire_img(self, pose_to_camera, inv_intrinsics, frame_time, bone_length, camera_pose=None,
                          render_size=128, semantic_map=False, use_normalized_intrinsics=False, no_grad=True, bbox=None):
        z_tri, z_render = self.get_latents(frame_time, pose_to_camera)
        s = self._samples
        return self.nerf.render_entire_img(pose_to_camera, inv_intrinsics, z_tri, z_render, bone_length, camera_pose,
                                           render_size, s["Nc"], s["Nf"], semantic_map, use_normalized_intrinsics,
                                           no_grad=no_grad, bbox=bbox)
