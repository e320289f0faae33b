"""TriPlaneNARF with the reference's constructor, method names and state-dict keys (models/narf.py:17-290,
libraries/NARF/base.py:11-83, libraries/NeRF/base.py:11-151), backed by the HIP kernels.

State-dict keys on the path: `tri_plane`, `mlp.layers.{0,1,2}.{bias, conv.weight,
conv.modulation.weight, conv.modulation.bias, noise.weight}`, buffers `canonical_pose`,
`canonical_bone_length`, `canonical_joints`, `canonical_parent_joints` - so reference snapshots load.

The StyleGAN2-ADA synthesis networks behind the producers (un-vendored in the reference) are this repo's restatement
(libraries/stylegan2_ada/networks.py), built and registered as the reference does (models/narf.py:31, :41, :71); any other
callable with the calling convention `net(z, encoded_length, truncation_psi=...)` can take a slot, where `encoded_length`
(B, P * 2 * num_frequency_for_other) is the bone-length positional encoding of models/narf.py:277-290:
  * default (GAN):        `model.tri_plane_gen` -> (B, (32 + P) * 3, 256, 256)
  * `constant_trimask`:   `model.generator`     -> (B, 96, 256, 256) feature planes; the part-probability planes are the
                          learned constant `model.tri_plane` (1, 3P, 256, 256) x `constant_trimask_lr_mul` (narf.py:32-38)
  * `deformation_field`:  `model.flow_generator` -> (B, 6, 256, 256) flow warping the constant feature planes (narf.py:39-58)
or the caller passes `model_input["tri_plane_feature"]`.
"""
import math
from typing import Dict, List, Optional, Union

import numpy as np
import torch
from torch import nn

from .. import ops
from ..libraries.NARF.pose_utils import transform_pose
from ..libraries.NeRF.utils import multi_part_positional_encoding
from ..libraries.NeRF.rendering import _parts_from_part_poses, render, render_entire_img


# ---- parameter containers mirroring libraries/NeRF/net.py:10-27 and custom_stylegan2/net.py:128-320 ----------
class _EqualLinearParams(nn.Module):            # EqualLinear(style_dim, in_channel, bias_init=1)
    def __init__(self, in_dim, out_dim, bias_init=1.0):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(out_dim, in_dim))
        self.bias = nn.Parameter(torch.zeros(out_dim).fill_(bias_init))


class _ModulatedConv1dParams(nn.Module):        # ModulatedConv1d(in, out, 1, style_dim)
    def __init__(self, in_channel, out_channel, style_dim):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(1, out_channel, in_channel, 1))
        self.modulation = _EqualLinearParams(style_dim, in_channel, bias_init=1.0)


class _NoiseParams(nn.Module):                  # NoiseInjection (unused on this path: use_noise=False)
    def __init__(self):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(1))


class _StyledConv1dParams(nn.Module):           # StyledConv(..., conv_1d=True, use_noise=False)
    def __init__(self, in_channel, out_channel, style_dim):
        super().__init__()
        self.conv = _ModulatedConv1dParams(in_channel, out_channel, style_dim)
        self.bias = nn.Parameter(torch.zeros(1, out_channel, 1))
        self.noise = _NoiseParams()


class StyledMLP(nn.Module):
    """Parameters of StyledMLP(in_dim, hidden_dim, out_dim, style_dim, num_layers=3); evaluated inside the HIP kernels."""

    def __init__(self, in_dim, hidden_dim, out_dim, style_dim=512, num_layers=3):
        super().__init__()
        assert (in_dim, hidden_dim, out_dim, num_layers) == (32, 64, 4, 3), \
            "the HIP kernels implement StyledMLP(32, 64, 4), the only shape the tri-plane NARF uses (narf.py:77)"
        self.layers = nn.ModuleList([_StyledConv1dParams(in_dim, hidden_dim, style_dim),
                                     _StyledConv1dParams(hidden_dim, hidden_dim, style_dim),
                                     _StyledConv1dParams(hidden_dim, out_dim, style_dim)])
        self.hidden_dim = hidden_dim

    def as_dict(self) -> Dict[str, torch.Tensor]:
        return {k: v for k, v in self.state_dict(keep_vars=True).items()}


class TriPlaneNARF(nn.Module):
    def __init__(self, config, z_dim: Union[int, List[int]] = 256, num_bone=1, bone_length=True, parent=None,
                 num_bone_param=None, view_dependent: bool = False):
        super().__init__()
        assert bone_length
        assert hasattr(config, "origin_location")
        if view_dependent:
            raise NotImplementedError("view-dependent colour (no_ray_direction: False) is not on the shipped tri-plane "
                                      "path (configs set no_ray_direction: True / GAN) and is not implemented")
        if getattr(config, "selector_mlp", False):
            raise NotImplementedError("nerf_params.selector_mlp=True (a learned per-part selector network instead of the "
                                      "part-probability planes, models/narf.py:59-70) is not implemented by the HIP kernels")
        self.no_selector = bool(getattr(config, "no_selector", False))          # uniform part weights (narf.py:133-134)
        self.clamp_mask = bool(getattr(config, "clamp_mask", False))            # sampling.py:46-47
        self.config = config
        self.tri_plane_based = True
        self.w_dim, self.feat_dim = 512, 32
        self.origin_location = config.origin_location
        self.coordinate_scale = config.coordinate_scale
        assert self.origin_location in ["center", "center_fixed", "center+head"]
        if isinstance(z_dim, list):
            self.z_dim, self.z2_dim = z_dim[0], z_dim[1]
        else:
            self.z_dim = self.z2_dim = z_dim
        self.num_frequency_for_position = getattr(config, "num_frequency_for_position", 10) \
            if not isinstance(config, dict) else config.get("num_frequency_for_position", 10)
        self.num_frequency_for_other = getattr(config, "num_frequency_for_other", 4) \
            if not isinstance(config, dict) else config.get("num_frequency_for_other", 4)
        self.view_dependent = False
        self.num_bone = num_bone - 1 if self.origin_location in ["center", "center_fixed"] else num_bone
        self.num_joints = num_bone
        self.use_bone_length = bone_length
        assert parent is not None
        self.parent_id = np.asarray(parent)
        self.temporal_state = {}
        self.buffers_tensors = {}
        self.mlp_mode = getattr(config, "mlp_mode", "f16x3") if not isinstance(config, dict) else config.get("mlp_mode", "f16x3")
        self.constant_trimask = bool(getattr(config, "constant_trimask", False)) and not config.constant_triplane
        self.trimask_lr_mul = float(getattr(config, "constant_trimask_lr_mul", 1)) if self.constant_trimask else 1.0
        if config.constant_triplane or (config.deformation_field and not self.constant_trimask):
            self.tri_plane = nn.Parameter(torch.zeros(1, 32 * 3 + self.num_bone * 3, 256, 256))
        elif self.constant_trimask:       # only the part-probability planes are a parameter (narf.py:34)
            self.tri_plane = nn.Parameter(torch.zeros(1, self.num_bone * 3, 256, 256) / self.trimask_lr_mul)
        # producer precedence as in models/narf.py:29-71: constant_triplane, constant_trimask, deformation_field, StyleGAN
        self.uses_warp = bool(config.deformation_field) and not config.constant_triplane and not self.constant_trimask
        # The StyleGAN2-ADA networks behind the producers (models/narf.py:31, :41, :71: prepare_stylegan2) are this repo's
        # restatement on the HIP ops (libraries/stylegan2_ada/networks.py), registered under the reference's attribute names so
        # that a snapshot's `tri_plane_gen.*` / `generator.*` / `flow_generator.*` keys load. Any of the three can be replaced
        # by a plain callable (z, encoded_length, truncation_psi=...) -> planes: see __setattr__.
        self.flow_generator = None         # deformation_field: (z, encoded_length, ...) -> flow (B, 6, 256, 256)
        self.generator = None              # constant_trimask: (z, encoded_length, ...) -> (B, 96, 256, 256)
        if config.constant_triplane:
            self.tri_plane_gen = lambda z, *args, **kwargs: self.tri_plane.expand(z.shape[0], -1, -1, -1)
        elif self.constant_trimask:
            self.generator = self.prepare_stylegan2(self.feat_dim * 3)
            self.tri_plane_gen = self._trimask_tri_plane    # models/narf.py:32-38
        elif self.uses_warp:
            self.flow_generator = self.prepare_stylegan2(2 * 3)
            self.tri_plane_gen = self._warped_tri_plane     # models/narf.py:40-58 with the HIP warp producer
        else:
            self.tri_plane_gen = self.prepare_stylegan2((self.feat_dim + self.num_bone) * 3)      # models/narf.py:71
        self.mlp = StyledMLP(32, 64, 4, style_dim=self.z2_dim)
        self._cl_cache = None              # (data_ptr, _version, shape) -> channel-last copy of a constant tri-plane

    # ---- canonical pose (models/narf.py:84-120) -------------------------------------------------------------------
    @property
    def memory_cost(self) -> int:
        """Activations per query point in the reference's units (output channels of every conv layer,
        libraries/custom_stylegan2/net.py:98-100): the StyledMLP's 64 + 64 + 4. (The reference's own
        TriPlaneNeRF.memory_cost, libraries/triplane/triplane_nerf.py:73-79, raises AttributeError on its StyledMLP child.)"""
        return sum(layer.conv.weight.shape[1] for layer in self.mlp.layers)

    @property
    def flops(self) -> int:
        """Multiply-adds x 2 of the per-point network (2 * in * out - out + bias, net.py:102-107, per layer): the
        32 -> 64 -> 64 -> 4 StyledMLP = 12 800 - the MFMA-eligible work SURVEY.md section 8(d) prices a query at."""
        return sum(2 * layer.conv.weight.shape[2] * layer.conv.weight.shape[1] for layer in self.mlp.layers)

    def register_canonical_pose(self, pose: np.ndarray) -> None:
        pose = np.asarray(pose)
        par = self.parent_id[1:]
        coordinate = pose[:, :3, 3]
        length = np.linalg.norm(coordinate[1:] - coordinate[par], axis=1)
        self.register_buffer("canonical_joints", torch.tensor(pose[1:, :3, 3], dtype=torch.float32))
        self.register_buffer("canonical_parent_joints", torch.tensor(pose[par, :3, 3], dtype=torch.float32))
        mid = (pose[1:, :, 3:] + pose[par, :, 3:]) / 2
        if self.origin_location == "center":
            cpose = np.concatenate([pose[1:, :, :3], mid], axis=-1)
        elif self.origin_location == "center_fixed":
            cpose = np.concatenate([pose[par, :, :3], mid], axis=-1)
        else:
            length = np.concatenate([length, np.ones(1)])
            cpose = np.concatenate([np.concatenate([pose[par, :, :3], mid], axis=-1), pose[15][None]])
        self.register_buffer("canonical_bone_length", torch.tensor(length, dtype=torch.float32))
        self.register_buffer("canonical_pose", torch.tensor(cpose, dtype=torch.float32))

    def transform_pose(self, pose_to_camera, bone_length):
        return transform_pose(pose_to_camera, bone_length, self.origin_location, self.parent_id)

    # ---- tri-plane handling ---------------------------------------------------------------------------------------
    _PRODUCERS = ("tri_plane_gen", "generator", "flow_generator")

    def __setattr__(self, name, value):
        """The three producer slots hold a network (registered: its parameters train and load with the model) or ANY callable
        with the producer's signature - e.g. a cached tri-plane, or another synthesis network; torch refuses a plain function
        where a child module is registered, so the slot is cleared first."""
        if name in self._PRODUCERS and not isinstance(value, nn.Module):
            self.__dict__.get("_modules", {}).pop(name, None)
            object.__setattr__(self, name, value)
            return
        if name in self._PRODUCERS:
            self.__dict__.pop(name, None)
        super().__setattr__(name, value)

    def prepare_stylegan2(self, out_channels):
        """models/narf.py:80-83: z -> planes, conditioned on the encoded bone lengths"""
        from ..libraries.stylegan2_ada.networks import prepare_triplane_generator
        return prepare_triplane_generator(self.z_dim, self.w_dim, out_channels, self.num_frequency_for_other * 2 * self.num_bone)

    def encode_bone_length(self, bone_length: torch.Tensor) -> torch.Tensor:
        """(B, P, 1) part bone lengths -> (B, P * 2F) conditioning vector of the tri-plane producers
        (models/narf.py:286-288: multi_part_positional_encoding(bone_length, num_frequency_for_other, num_bone)[:, :, 0])."""
        return multi_part_positional_encoding(bone_length, self.num_frequency_for_other, num_bone=self.num_bone)[:, :, 0]

    def compute_tri_plane_feature(self, z, bone_length, truncation_psi=1):
        if self.config.constant_triplane:
            bs = bone_length.shape[0] if z is None else z.shape[0]
            return self.tri_plane.expand(bs, -1, -1, -1)
        return self.tri_plane_gen(z, self.encode_bone_length(bone_length), truncation_psi=truncation_psi)

    # ---- constant_trimask producer (models/narf.py:32-38) -------------------------------------------------------------
    def _feature_planes(self, z, *args, **kwargs) -> torch.Tensor:
        return self.generator(z, *args, **kwargs)

    def _trimask_tri_plane(self, z, *args, **kwargs) -> torch.Tensor:
        """cat([generator(z, ...), tri_plane.expand(B) * lr_mul], dim=1): (B, 96 + 3P, 256, 256), differentiable."""
        feat = self._feature_planes(z, *args, **kwargs)
        return torch.cat([feat, self.tri_plane.expand(feat.shape[0], -1, -1, -1) * self.trimask_lr_mul], dim=1)

    # ---- deformation-field producer (models/narf.py:40-58) ----------------------------------------------------------
    def _flow(self, z, *args, **kwargs) -> torch.Tensor:
        return self.flow_generator(z, *args, **kwargs)

    def _constant_planes_cl(self) -> torch.Tensor:
        """Channel-last copy of the constant feature planes, re-laid out once per parameter version."""
        tri = self.tri_plane.detach()
        key = (tri.data_ptr(), tri._version, tuple(tri.shape))
        if self._cl_cache is None or self._cl_cache[0] != key:
            self._cl_cache = (key, ops.triplane_pack(tri))
        return self._cl_cache[1]

    def _warped_tri_plane(self, z, *args, **kwargs) -> torch.Tensor:
        """The reference's `warp` closure: (B, 96 + 3P, 256, 256) NCHW, differentiable w.r.t. the tri-plane parameter
        and the flow (the autograd path and `buffers_tensors`; rendering without gradients takes `_tri_plane_pair`'s
        direct channel-last route instead)."""
        flow = self._flow(z, *args, **kwargs)
        warped = _WarpFunction.apply(self.tri_plane, flow)                                  # (B, 96, H, W)
        return torch.cat([warped, self.tri_plane[:, 96:].expand(flow.shape[0], -1, -1, -1)], dim=1)

    def _tri_plane_pair_graph(self, model_input: Dict):
        """Deformation field under autograd: (the tri-plane parameter - it holds the part-probability planes -, warped
        feature planes channel-last (B, 3, H, W, 32)), both differentiable; no NCHW copy of the warped planes."""
        flow = self._flow(model_input.get("z"), self.encode_bone_length(model_input["bone_length"]),
                          truncation_psi=model_input.get("truncation_psi", 1))
        return self.tri_plane, _WarpCLFunction.apply(self.tri_plane, flow)

    def _tri_plane_pair(self, model_input: Dict):
        """(tri-plane NCHW (1 or B images), channel-last feature planes) for this call.

        A constant tri-plane (one image shared by the batch) is re-laid out once per parameter version; with a
        deformation field the constant planes are warped straight into the channel-last layout (one image per flow)
        and the part-probability planes stay shared."""
        tri = model_input.get("tri_plane_feature")
        if tri is None and self.uses_warp:
            flow = self._flow(model_input.get("z"), self.encode_bone_length(model_input["bone_length"]),
                              truncation_psi=model_input.get("truncation_psi", 1)).detach()
            feat_cl = ops.triplane_warp_fwd(self._constant_planes_cl(), flow)
            self.buffers_tensors["tri_plane_feature"] = None      # (not materialised NCHW on this route)
            return self.tri_plane.detach(), feat_cl
        if tri is None:
            tri = self.compute_tri_plane_feature(model_input.get("z"), model_input["bone_length"],
                                                 model_input.get("truncation_psi", 1))
        self.buffers_tensors["tri_plane_feature"] = tri
        if not self.training:
            self.temporal_state["tri_plane_feature"] = tri
        if tri.shape[0] > 1 and tri.stride(0) == 0:       # an expanded constant tri-plane
            tri = tri[:1]
        tri = tri.detach()
        if not tri.is_contiguous():
            tri = tri.contiguous()
        key = (tri.data_ptr(), tri._version, tuple(tri.shape))
        if self._cl_cache is not None and self._cl_cache[0] == key and tri.shape[0] == 1:
            return tri, self._cl_cache[1]
        cl = ops.triplane_pack(tri)
        if tri.shape[0] == 1:
            self._cl_cache = (key, cl)
        return tri, cl

    def _tri_plane_graph(self, model_input: Dict) -> torch.Tensor:
        """The tri-plane tensor autograd should differentiate: the (1, ...) parameter itself for a constant tri-plane
        (its expand() is undone so that the gradient is accumulated once, not per image), else the producer's output."""
        tri = model_input.get("tri_plane_feature")
        if tri is None:
            if self.config.constant_triplane:
                return self.tri_plane
            tri = self.compute_tri_plane_feature(model_input.get("z"), model_input["bone_length"],
                                                 model_input.get("truncation_psi", 1))
        return tri

    def kernel_flags(self) -> Dict[str, bool]:
        """nerf_params switches the kernels take (forward and backward)"""
        return dict(multiply_density_with_weight=bool(self.config.multiply_density_with_triplane_wieght),
                    clamp_mask=self.clamp_mask, uniform_part_weight=self.no_selector)

    def _mlp_pack(self, z_rend: torch.Tensor) -> torch.Tensor:
        return self._mlp_pack_from(z_rend, self.mlp.as_dict())

    def _mlp_pack_from(self, z_rend: torch.Tensor, mlp: Dict[str, torch.Tensor]) -> torch.Tensor:
        return ops.prepare_mlp(z_rend, mlp)

    # ---- the reference's entry points -------------------------------------------------------------------------------
    def forward(self, batchsize, sampled_img_coord, pose_to_camera, inv_intrinsics, z, z_rend, bone_length,
                render_scale=1, Nc=64, Nf=128, return_intermediate=False, truncation_psi=1,
                camera_pose: Optional[torch.Tensor] = None, return_disparity=False, bins=None, seed=None):
        """NARFBase.forward (libraries/NARF/base.py:26-51): raw 24-joint poses in, colour/mask(/disparity) out.

        One enarf_prepare launch (part frames + modulated MLP weights) then one enarf_render_fwd launch."""
        model_input = {"z": z, "z_rend": z_rend, "bone_length": bone_length, "truncation_psi": truncation_psi}
        parts, pack = ops.prepare(pose_to_camera, bone_length, self.canonical_bone_length, z_rend, self.mlp.as_dict(),
                                  self.parent_id, self.origin_location, self.coordinate_scale)
        # part-frame count only (the kernel reads `parts`); keeps render()'s shape assertion meaningful
        pose_parts = pose_to_camera.new_empty(pose_to_camera.shape[0], self.num_bone, 4, 4)
        res = render(self, sampled_img_coord, pose_parts, inv_intrinsics, render_scale, Nc, Nf,
                     return_intermediate=return_intermediate, camera_pose=camera_pose,
                     model_input=model_input, _parts=parts, _pack=pack, bins=bins, seed=seed)
        if return_intermediate:                       # libraries/NeRF/base.py:112-114
            return res[0], res[1], res[3]
        color, mask, disparity = res
        if return_disparity:
            return color, mask, disparity
        return color, mask

    def render_entire_img(self, pose_to_camera, inv_intrinsics, z, z_rend, bone_length, camera_pose=None,
                          render_size=128, Nc=64, Nf=128, semantic_map=False, use_normalized_intrinsics=False,
                          no_grad=True, truncation_psi=1, bbox=None):
        """NARFBase.render_entire_img (libraries/NARF/base.py:53-63)."""
        model_input = {"z": z, "z_rend": z_rend, "bone_length": bone_length, "truncation_psi": truncation_psi}
        pose_parts, model_input["bone_length"] = self.transform_pose(pose_to_camera, bone_length)
        model_input["tri_plane_feature"] = self.compute_tri_plane_feature(z, bone_length)
        return render_entire_img(self, pose_parts, inv_intrinsics, camera_pose, render_size, Nc, Nf, semantic_map,
                                 use_normalized_intrinsics, no_grad, model_input, bbox=bbox)

    def density_volume(self, pose_to_camera, z, z_rend, bone_length, voxel_size=0.003, truncation_psi=0.4):
        """The (2/voxel_size + 1)^3 density grid `render_mesh` thresholds (libraries/NARF/base.py:65-77 up to the
        marching-cubes call): one lattice-mode launch of the query kernel, the volume stays on the device."""
        from ..libraries.NARF.mesh_rendering import density_volume
        center, pose_parts, model_input = self._mesh_inputs(pose_to_camera, z, z_rend, bone_length, truncation_psi)
        return density_volume(self, pose_parts, center, voxel_size, model_input)

    def _mesh_inputs(self, pose_to_camera, z, z_rend, bone_length, truncation_psi):
        if not ((z is None or z.shape[0] == 1) and (bone_length is None or bone_length.shape[0] == 1)):
            raise AssertionError("render_mesh takes one sample (base.py:67-68)")
        center = pose_to_camera[:, 0, :3, 3:].clone()  # (1, 3, 1)
        model_input = {"z": z, "z_rend": z_rend, "bone_length": bone_length, "truncation_psi": truncation_psi}
        pose_parts, model_input["bone_length"] = self.transform_pose(pose_to_camera, bone_length)
        model_input["tri_plane_feature"] = self.compute_tri_plane_feature(z, bone_length)
        return center, pose_parts, model_input

    def render_mesh(self, pose_to_camera, intrinsics, z, z_rend, bone_length, voxel_size=0.003, mesh_th=15,
                    truncation_psi=0.4, img_size=128):
        """NARFBase.render_mesh (libraries/NARF/base.py:65-83): density sweep (HIP, lattice mode) -> marching cubes
        (PyMCubes) -> Phong render (pytorch3d). The two third-party stages raise ImportError when absent, exactly where
        the reference's imports would; `density_volume()` returns the swept grid without them."""
        from ..libraries.NARF.mesh_rendering import create_mesh, render_mesh_
        center, pose_parts, model_input = self._mesh_inputs(pose_to_camera, z, z_rend, bone_length, truncation_psi)
        meshes = create_mesh(self, pose_parts, center=center, voxel_size=voxel_size, mesh_th=mesh_th,
                             model_input=model_input)
        return render_mesh_(meshes, intrinsics, img_size), meshes

    def calc_density_and_color_from_camera_coord_v2(self, position: torch.Tensor, pose_to_camera: torch.Tensor,
                                                    ray_direction: Optional[torch.Tensor], model_input: Dict):
        """models/narf.py:176-211: position (B,3,n) camera coords, pose_to_camera (B,P,4,4) part frames whose
        translation is already x coordinate_scale (as render / create_mesh pass it) -> density (B,1,n), colour (B,3,n)."""
        B, P = pose_to_camera.shape[:2]
        if pose_to_camera.requires_grad or position.requires_grad:
            raise NotImplementedError("Currently pose and positions should not be differentiable")
        parts = torch.zeros(B, P, 16, dtype=torch.float32, device=position.device)
        parts[:, :, :9] = pose_to_camera[:, :, :3, :3].reshape(B, P, 9)
        parts[:, :, 9:12] = pose_to_camera[:, :, :3, 3]
        parts[:, :, 12] = (self.canonical_bone_length[:, None] / model_input["bone_length"] / self.coordinate_scale)[:, :, 0]
        mult_w = bool(self.config.multiply_density_with_triplane_wieght)
        z_rend = model_input["z_rend"]
        tri_graph = self._tri_plane_graph(model_input)
        params = self.mlp.as_dict()
        if torch.is_grad_enabled() and (tri_graph.requires_grad or z_rend.requires_grad or
                                        any(p.requires_grad for p in params.values())):
            k = dict(points=position.detach(), parts=parts, canonical_pose=self.canonical_pose, mlp_mode=self.mlp_mode,
                     pack_fn=self._mlp_pack_from, flags=self.kernel_flags())
            flat = [params[f"layers.{i}.{leaf}"] for i in range(3) for leaf in _MLP_LEAVES]
            return _QueryFunction.apply(k, tri_graph, z_rend, *flat)
        tri, feat_cl = self._tri_plane_pair(model_input)
        pack = self._mlp_pack(z_rend)
        den, col, vb = ops.query_fwd(position, parts, self.canonical_pose, tri, feat_cl, pack, mlp_mode=self.mlp_mode,
                                     need_valid=True, **self.kernel_flags())
        if not self.training:
            self.temporal_state["valid_bits"] = vb
        return den, col


_MLP_LEAVES = ("conv.weight", "conv.modulation.weight", "conv.modulation.bias", "bias")


class _WarpFunction(torch.autograd.Function):
    """Constant tri-plane parameter (1, 96 + 3P, H, W) + flow (B, 6, H, W) -> warped feature planes (B, 96, H, W) NCHW:
    enarf_triplane_pack + enarf_triplane_warp_fwd forward, enarf_triplane_warp_bwd + enarf_triplane_unpack_add backward."""

    @staticmethod
    def forward(ctx, tri, flow):
        tri_c, fl = tri.detach().contiguous(), flow.detach().contiguous().float()
        src_cl = ops.triplane_pack(tri_c)
        out_cl = ops.triplane_warp_fwd(src_cl, fl)
        B, _, H, W, C = out_cl.shape
        ctx.save_for_backward(src_cl, fl)
        ctx.tri_shape = tuple(tri_c.shape)
        return out_cl.permute(0, 1, 4, 2, 3).reshape(B, 3 * C, H, W)

    @staticmethod
    def backward(ctx, g):
        src_cl, fl = ctx.saved_tensors
        B, _, H, W = fl.shape
        g_cl = g.reshape(B, 3, 32, H, W).permute(0, 1, 3, 4, 2).contiguous()
        gs, gf = ops.triplane_warp_bwd(g_cl, src_cl, fl, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        g_tri = None
        if gs is not None:
            g_tri = torch.zeros(ctx.tri_shape, dtype=torch.float32, device=fl.device)
            ops.triplane_unpack_add(gs.reshape(1, 3, H, W, 32), g_tri)
        return g_tri, gf


class _WarpCLFunction(torch.autograd.Function):
    """As _WarpFunction, but the warped planes stay channel-last (B, 3, H, W, 32): what the march reads."""

    @staticmethod
    def forward(ctx, tri, flow):
        tri_c, fl = tri.detach().contiguous(), flow.detach().contiguous().float()
        src_cl = ops.triplane_pack(tri_c)
        ctx.save_for_backward(src_cl, fl)
        ctx.tri_shape = tuple(tri_c.shape)
        return ops.triplane_warp_fwd(src_cl, fl)

    @staticmethod
    def backward(ctx, g_cl):
        src_cl, fl = ctx.saved_tensors
        H, W = fl.shape[2:]
        gs, gf = ops.triplane_warp_bwd(g_cl.contiguous(), src_cl, fl, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        g_tri = None
        if gs is not None:
            g_tri = torch.zeros(ctx.tri_shape, dtype=torch.float32, device=fl.device)
            ops.triplane_unpack_add(gs.reshape(1, 3, H, W, 32), g_tri)
        return g_tri, gf


class _QueryFunction(torch.autograd.Function):
    """Differentiable point query: forward = enarf_prepare (MLP pack) + enarf_triplane_pack + enarf_query_fwd, backward =
    enarf_query_bwd + enarf_weight_grad + enarf_prepare_bwd + un-pack. Differentiable inputs: the tri-plane, z_rend and
    the 12 StyledMLP tensors."""

    @staticmethod
    def forward(ctx, k, tri, z_rend, *params):
        mlp = {f"layers.{i}.{leaf}": params[4 * i + j] for i in range(3) for j, leaf in enumerate(_MLP_LEAVES)}
        tri_c = tri.detach().contiguous()
        feat_cl = ops.triplane_pack(tri_c)
        pack = k["pack_fn"](z_rend.detach(), {n: t.detach() for n, t in mlp.items()})
        den, col = ops.query_fwd(k["points"], k["parts"], k["canonical_pose"], tri_c, feat_cl, pack, mlp_mode=k["mlp_mode"],
                                 **k["flags"])
        ctx.k = k
        ctx.save_for_backward(tri_c, feat_cl, pack, z_rend.detach(), *[p.detach() for p in params])
        return den, col

    @staticmethod
    def backward(ctx, g_den, g_col):
        k = ctx.k
        tri_c, feat_cl, pack, z_rend = ctx.saved_tensors[:4]
        params = ctx.saved_tensors[4:]
        mlp = {f"layers.{i}.{leaf}": params[4 * i + j] for i in range(3) for j, leaf in enumerate(_MLP_LEAVES)}
        grad_tri, dW, db = ops.query_bwd(k["points"], k["parts"], k["canonical_pose"], tri_c, feat_cl, pack, g_den, g_col,
                                         **k["flags"])
        pg, dz = ops.prepare_bwd(z_rend, mlp, dW)
        grads = []
        for i in range(3):
            grads += [pg[f"layers.{i}.conv.weight"], pg[f"layers.{i}.conv.modulation.weight"],
                      pg[f"layers.{i}.conv.modulation.bias"], db[i].reshape(params[4 * i + 3].shape)]
        return (None, grad_tri, dz) + tuple(grads)
