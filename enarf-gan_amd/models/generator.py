"""TriNARFGenerator / DSONARFGenerator API shells with the reference's signatures (models/generator.py:14-300).

Only the renderer is in scope: the StyleGAN2 background generator is not rebuilt. Pass
`black_background=True`, or assign `gen.background_generator` (any module with the reference's call convention
`bg([z_bg, z_render], inject_index=n_latent - 4) -> (image, _)` and an `n_latent` attribute).
"""
from typing import Optional

import numpy as np
import torch
from torch import nn

from ..libraries.NeRF.ray_sampler import mask_based_sampler, whole_image_grid_ray_sampler
from .narf import TriPlaneNARF


class TriNARFGenerator(nn.Module):
    def __init__(self, config, size, num_bone=1, parent_id=None, num_bone_param=None, black_background=False):
        super().__init__()
        self.config = config
        self.size = size
        self.num_bone = num_bone
        self.ray_sampler = whole_image_grid_ray_sampler
        self.background_ratio = config.background_ratio
        self.black_background = black_background
        z_dim = config.z_dim
        if config.nerf_params.origin_location == "center+head":
            num_bone_param = num_bone
        self.nerf = TriPlaneNARF(config.nerf_params, z_dim=[z_dim * 2, z_dim], num_bone=num_bone, bone_length=True,
                                 parent=parent_id, num_bone_param=num_bone_param)
        self.background_generator = None   # out of scope: assign one, or use black_background / black_bg_if_possible

    def register_canonical_pose(self, pose: np.ndarray):
        self.nerf.register_canonical_pose(pose)

    def normalized_inv_intrinsics(self, intrinsics: torch.Tensor):
        normalized_intrinsics = torch.cat([intrinsics[:2] / self.size, intrinsics.new([[0, 0, 1]])], dim=0)
        return torch.linalg.inv(normalized_intrinsics)

    def _split_z(self, z):
        if not self.black_background:
            z_dim = z.shape[1] // 4
            return torch.split(z, [z_dim * 2, z_dim, z_dim], dim=1)
        z_dim = z.shape[1] // 3
        return tuple(torch.split(z, [z_dim * 2, z_dim], dim=1)) + (None,)

    def forward(self, pose_to_camera, pose_to_world, bone_length, z=None, inv_intrinsics=None,
                return_intermediate=False, truncation_psi=1, black_bg_if_possible=False, return_disparity=False,
                return_bg=False):
        assert self.num_bone == 1 or (bone_length is not None and pose_to_camera is not None)
        batchsize = pose_to_camera.shape[0]
        grid, homo_img = self.ray_sampler(self.size, self.size, batchsize, device=pose_to_camera.device)
        z_for_nerf, z_for_neural_render, z_for_background = self._split_z(z)
        if isinstance(inv_intrinsics, np.ndarray):
            inv_intrinsics = torch.from_numpy(inv_intrinsics)
        inv_intrinsics = inv_intrinsics.float().to(homo_img.device)
        nerf_output = self.nerf(batchsize, homo_img, pose_to_camera, inv_intrinsics, z_for_nerf, z_for_neural_render,
                                bone_length, Nc=self.config.nerf_params.Nc, Nf=self.config.nerf_params.Nf,
                                return_intermediate=return_intermediate, truncation_psi=truncation_psi,
                                return_disparity=return_disparity)
        fg_color, fg_mask = nerf_output[:2]
        fine_weights = self.nerf.buffers_tensors["fine_weights"]
        fine_depth = self.nerf.buffers_tensors["fine_depth"]
        fg_color = fg_color.reshape(batchsize, 3, self.size, self.size)
        fg_mask = fg_mask.reshape(batchsize, self.size, self.size)
        if not self.black_background and not black_bg_if_possible:
            if self.background_generator is None:
                raise NotImplementedError("the StyleGAN2 background generator is out of scope: assign "
                                          "gen.background_generator, or use black_background / black_bg_if_possible")
            n_latent = self.background_generator.n_latent
            bg_color, _ = self.background_generator([z_for_background, z_for_neural_render], inject_index=n_latent - 4)
        else:
            bg_color = -1
        rendered_color = fg_color + (1 - fg_mask[:, None]) * bg_color
        if return_intermediate:                          # models/generator.py:109-111
            fine_points, fine_density = nerf_output[-1]
            return rendered_color, fg_mask, fine_points, fine_density
        if return_disparity:
            disparity = nerf_output[2] * self.config.nerf_params.coordinate_scale
            return rendered_color, fg_mask, disparity
        if return_bg:
            return fg_color, fg_mask, bg_color
        return rendered_color, fg_mask, fine_weights, fine_depth


class DSONARFGenerator(nn.Module):
    def __init__(self, config, size, num_bone=1, parent_id=None, num_bone_param=None):
        super().__init__()
        self.config = config
        self.size = size
        self.num_bone = num_bone
        self.ray_sampler = mask_based_sampler
        if not config.use_triplane:
            raise NotImplementedError("MLPNARF (use_triplane: False) is not the tri-plane path (SURVEY.md §2, OUT)")
        self.time_conditional = config.nerf_params.time_conditional
        self.pose_conditional = config.nerf_params.pose_conditional
        z_dim = (20 if self.time_conditional else 0) + ((num_bone - 1) * 9 if self.pose_conditional else 0)
        if config.nerf_params.origin_location == "center+head":
            num_bone_param = num_bone
        view_dependent = not config.nerf_params.no_ray_direction
        self.nerf = TriPlaneNARF(config.nerf_params, z_dim=z_dim, num_bone=num_bone, bone_length=True,
                                 parent=parent_id, num_bone_param=num_bone_param, view_dependent=view_dependent)

    def register_canonical_pose(self, pose: np.ndarray):
        self.nerf.register_canonical_pose(pose)

    @staticmethod
    def positional_encoding(x: torch.Tensor, num_frequency: int) -> torch.Tensor:
        x = x[:, None] * 2 ** torch.arange(num_frequency, device=x.device) * np.pi
        return torch.cat([torch.cos(x), torch.sin(x)], dim=1)

    @staticmethod
    def pose_encoding(pose: torch.Tensor):
        rot, root_rot = pose[:, 1:, :3, :3], pose[:, :1, :3, :3]
        encoded = torch.matmul(root_rot.permute(0, 1, 3, 2), rot)
        return encoded.reshape(encoded.shape[0], -1)

    def get_latents(self, frame_time: torch.Tensor, pose_to_camera: torch.Tensor):
        zs = []
        if self.time_conditional:
            zs.append(self.positional_encoding(frame_time, num_frequency=10))
        if self.pose_conditional:
            zs.append(self.pose_encoding(pose_to_camera))
        assert len(zs) > 0
        z = torch.cat(zs, dim=1)
        return z, z

    def forward(self, pose_to_camera, camera_pose, mask, frame_time, bone_length, inv_intrinsics,
                background: Optional[float] = None):
        assert bone_length is not None and pose_to_camera is not None
        assert isinstance(inv_intrinsics, torch.Tensor)
        batchsize = pose_to_camera.shape[0]
        grid, img_coord = self.ray_sampler(mask, self.config.ray_batchsize)
        z1, z2 = self.get_latents(frame_time, pose_to_camera)
        rendered_color, rendered_mask = self.nerf(batchsize, img_coord, pose_to_camera, inv_intrinsics, z1, z2,
                                                  bone_length, Nc=self.config.nerf_params.Nc,
                                                  Nf=self.config.nerf_params.Nf, return_intermediate=False,
                                                  camera_pose=camera_pose)
        if background is None:
            background = -1
        rendered_color = rendered_color + background * (1 - rendered_mask[:, None])
        return rendered_color, rendered_mask, grid

    def render_entire_img(self, pose_to_camera, inv_intrinsics, frame_time, bone_length, camera_pose=None,
                          render_size=128, semantic_map=False, use_normalized_intrinsics=False, no_grad=True, bbox=None):
        z1, z2 = self.get_latents(frame_time, pose_to_camera)
        return self.nerf.render_entire_img(pose_to_camera, inv_intrinsics, z1, z2, bone_length, camera_pose, render_size,
                                           self.config.nerf_params.Nc, self.config.nerf_params.Nf, semantic_map,
                                           use_normalized_intrinsics, no_grad=no_grad, bbox=bbox)
