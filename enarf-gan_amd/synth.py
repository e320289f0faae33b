"""Seeded synthetic inputs for the ENARF ray-render hot path.

No dataset, SMPL model file or pretrained snapshot is reachable offline, so every
test, the bench and the golden-vector generator draw their inputs from here:
an SMPL-topology skeleton (24 joints, parents as in reference DSO_demo.py:26-27),
random articulated poses, a pinhole camera, a smoothed-noise tri-plane and
StyledMLP parameters with the reference's state-dict key names.

Everything is deterministic in (seed, shape) and independent of the device: tensors
are produced on the CPU with numpy / a seeded torch.Generator, callers move them.
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import numpy as np
import torch

# reference DSO_demo.py:26-27 / ENARF_GAN_demo.py
SMPL_PARENTS = np.array([-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13,
                         14, 16, 17, 18, 19, 20, 21], dtype=np.int64)
NUM_JOINTS = 24
FEAT_DIM = 32          # models/narf.py:22
PLANE_RES = 256        # models/narf.py:30
HIDDEN = 64            # models/narf.py:77  StyledMLP(32, 64, 4)

# Plausible T-pose joint positions in metres (x right, y up, z forward); a committed
# constant standing in for smpl_data/neutral_canonical.npy which is not in the tree.
_REST = np.array([
    [0.00, 0.00, 0.00],    # 0 pelvis
    [0.07, -0.09, 0.00],   # 1 l_hip
    [-0.07, -0.09, 0.00],  # 2 r_hip
    [0.00, 0.11, -0.01],   # 3 spine1
    [0.10, -0.47, 0.01],   # 4 l_knee
    [-0.10, -0.47, 0.01],  # 5 r_knee
    [0.00, 0.25, 0.00],    # 6 spine2
    [0.09, -0.87, -0.03],  # 7 l_ankle
    [-0.09, -0.87, -0.03],  # 8 r_ankle
    [0.00, 0.31, 0.02],    # 9 spine3
    [0.11, -0.92, 0.09],   # 10 l_foot
    [-0.11, -0.92, 0.09],  # 11 r_foot
    [0.00, 0.51, -0.02],   # 12 neck
    [0.08, 0.42, 0.00],    # 13 l_collar
    [-0.08, 0.42, 0.00],   # 14 r_collar
    [0.00, 0.60, 0.03],    # 15 head
    [0.17, 0.43, 0.00],    # 16 l_shoulder
    [-0.17, 0.43, 0.00],   # 17 r_shoulder
    [0.43, 0.43, -0.01],   # 18 l_elbow
    [-0.43, 0.43, -0.01],  # 19 r_elbow
    [0.68, 0.43, 0.00],    # 20 l_wrist
    [-0.68, 0.43, 0.00],   # 21 r_wrist
    [0.77, 0.42, 0.01],    # 22 l_hand
    [-0.77, 0.42, 0.01],   # 23 r_hand
], dtype=np.float64)


def rest_joints() -> np.ndarray:
    """(24, 3) rest-pose joints, camera convention (y down)."""
    j = _REST.copy()
    j[:, 1] *= -1.0
    return j


def canonical_pose() -> np.ndarray:
    """(24, 4, 4) float64 canonical pose: identity rotations, rest joints as translation.

    Plays the role of the array handed to `register_canonical_pose` (models/narf.py:84).
    """
    pose = np.tile(np.eye(4)[None], (NUM_JOINTS, 1, 1))
    pose[:, :3, 3] = rest_joints()
    return pose


def _axis_angle_to_matrix(v: np.ndarray) -> np.ndarray:
    theta = np.linalg.norm(v)
    if theta < 1e-12:
        return np.eye(3)
    k = v / theta
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + math.sin(theta) * K + (1 - math.cos(theta)) * (K @ K)


def random_pose(batch: int, seed: int = 1234, angle_std: float = 0.3,
                length_jitter: float = 0.1) -> Tuple[torch.Tensor, torch.Tensor]:
    """Random articulated poses.

    Returns pose_to_camera (B, 24, 4, 4) float32 and bone_length (B, 23, 1) float32,
    the two tensors the reference generators take (models/generator.py:56, :219).
    Per image b a RandomState(seed + b) draws per-joint axis-angles ~ N(0, angle_std^2)
    composed down the kinematic tree, per-bone length scales ~ U(1-j, 1+j) and a root
    translation with z in [2.5, 3.5] m.
    """
    rest = rest_joints()
    poses = np.zeros((batch, NUM_JOINTS, 4, 4), dtype=np.float64)
    lengths = np.zeros((batch, NUM_JOINTS - 1, 1), dtype=np.float64)
    for b in range(batch):
        rs = np.random.RandomState(seed + b)
        aa = rs.normal(0.0, angle_std, size=(NUM_JOINTS, 3))
        aa[0] = rs.normal(0.0, 0.5, size=3) * np.array([0.3, 1.0, 0.3])
        scale = rs.uniform(1 - length_jitter, 1 + length_jitter, size=NUM_JOINTS)
        root_t = np.array([rs.uniform(-0.3, 0.3), rs.uniform(-0.15, 0.15), rs.uniform(2.5, 3.5)])
        G = np.zeros((NUM_JOINTS, 4, 4))
        for j in range(NUM_JOINTS):
            L = np.eye(4)
            L[:3, :3] = _axis_angle_to_matrix(aa[j])
            p = SMPL_PARENTS[j]
            if p < 0:
                L[:3, 3] = root_t
                G[j] = L
            else:
                L[:3, 3] = (rest[j] - rest[p]) * scale[j]
                G[j] = G[p] @ L
        poses[b] = G
        jt = G[:, :3, 3]
        lengths[b, :, 0] = np.linalg.norm(jt[1:] - jt[SMPL_PARENTS[1:]], axis=1)
    return (torch.from_numpy(poses.astype(np.float32)),
            torch.from_numpy(lengths.astype(np.float32)))


def intrinsics(size: int, batch: int = 1) -> Tuple[torch.Tensor, torch.Tensor]:
    """K and K^-1, (B, 3, 3) float32: focal 1.2*S, principal point S/2."""
    K = np.array([[1.2 * size, 0.0, size / 2.0],
                  [0.0, 1.2 * size, size / 2.0],
                  [0.0, 0.0, 1.0]], dtype=np.float64)
    Kinv = np.linalg.inv(K)
    K = torch.from_numpy(np.tile(K[None], (batch, 1, 1)).astype(np.float32))
    Kinv = torch.from_numpy(np.tile(Kinv[None], (batch, 1, 1)).astype(np.float32))
    return K, Kinv


def pixel_centres(size: int, batch: int = 1) -> torch.Tensor:
    """(B, 1, 3, S*S) homogeneous pixel-centre coordinates, row-major pixels.

    Same values as whole_image_grid_ray_sampler (libraries/NeRF/ray_sampler.py:42-67) with
    render_size == patch_size, and as render_entire_img's img_coord (rendering.py:396-399).
    """
    idx = torch.arange(size * size)
    x = (idx % size).float() + 0.5
    y = torch.div(idx, size, rounding_mode="floor").float() + 0.5
    homo = torch.stack([x, y, torch.ones_like(x)], dim=0)
    return homo[None, None].repeat(batch, 1, 1, 1).contiguous()


def _box5(x: torch.Tensor) -> torch.Tensor:
    k = torch.ones(1, 1, 5, 5, dtype=x.dtype, device=x.device) / 25.0
    shp = x.shape
    y = torch.nn.functional.conv2d(x.reshape(-1, 1, shp[-2], shp[-1]), k, padding=2)
    return y.reshape(shp)


def part_centres(origin_location: str = "center_fixed") -> np.ndarray:
    """(P, 3) canonical part centres (bone mid-points; + head joint for center+head)."""
    j = rest_joints()
    mid = (j[1:] + j[SMPL_PARENTS[1:]]) / 2
    if origin_location == "center+head":
        mid = np.concatenate([mid, j[15][None]], axis=0)
    return mid


def make_triplane(batch: int, num_parts: int, seed: int = 7, device: str = "cpu",
                  origin_location: str = "center_fixed") -> torch.Tensor:
    """(B, (32+P)*3, 256, 256) float32 tri-plane in the reference's NCHW channel order.

    Channels [p*32, (p+1)*32) are feature plane p in {xy, yz, zx} (sampling.py:28-31);
    channel 96 + k*3 + p is the one-channel part-probability plane p of part k
    (models/narf.py:239). Box-filtered N(0,1) noise (x3 to keep O(1) magnitude); mask
    planes get +2 inside a 0.3-radius disc around the part centre so that the product of
    sigmoids is ~0.6-0.7 near the bone.
    """
    g = torch.Generator(device=device).manual_seed(seed)
    ch = (FEAT_DIM + num_parts) * 3
    tri = torch.empty(batch, ch, PLANE_RES, PLANE_RES, dtype=torch.float32, device=device)
    for b in range(batch):
        noise = torch.randn(ch, PLANE_RES, PLANE_RES, generator=g, dtype=torch.float32, device=device)
        tri[b] = _box5(noise) * 3.0
    centres = part_centres(origin_location)[:num_parts]
    lin = (torch.arange(PLANE_RES, dtype=torch.float32, device=device) + 0.5) / (PLANE_RES / 2) - 1.0
    yy, xx = torch.meshgrid(lin, lin, indexing="ij")   # yy: row coord, xx: col coord
    for k in range(num_parts):
        for p in range(3):
            cx, cy = float(centres[k][p]), float(centres[k][(p + 1) % 3])
            disc = ((xx - cx) ** 2 + (yy - cy) ** 2) < 0.3 ** 2
            tri[:, 3 * FEAT_DIM + k * 3 + p] += 2.0 * disc.float()
    return tri


def make_mlp_params(style_dim: int, seed: int = 11, bias_std: float = 0.1) -> Dict[str, torch.Tensor]:
    """StyledMLP(32, 64, 4, style_dim) parameters under the reference's state-dict names.

    Keys follow `nerf.mlp.layers.{i}.*` minus the `nerf.mlp.` prefix:
    conv.weight (1, out, in, 1) ~ randn (custom_stylegan2/net.py:216-218),
    conv.modulation.weight (in, style_dim) ~ randn, conv.modulation.bias (in) = 1 (:220),
    bias (1, out, 1) (zero in the reference's init; drawn ~N(0, bias_std^2) here so the
    bias path is exercised), noise.weight (1) = 0 (unused: use_noise=False).
    """
    g = torch.Generator().manual_seed(seed)
    dims = [(FEAT_DIM, HIDDEN), (HIDDEN, HIDDEN), (HIDDEN, 4)]
    sd = {}
    for i, (cin, cout) in enumerate(dims):
        sd[f"layers.{i}.conv.weight"] = torch.randn(1, cout, cin, 1, generator=g)
        sd[f"layers.{i}.conv.modulation.weight"] = torch.randn(cin, style_dim, generator=g)
        sd[f"layers.{i}.conv.modulation.bias"] = torch.ones(cin)
        sd[f"layers.{i}.bias"] = torch.randn(1, cout, 1, generator=g) * bias_std
        sd[f"layers.{i}.noise.weight"] = torch.zeros(1)
    return sd


def make_z_rend(batch: int, style_dim: int, seed: int = 13) -> torch.Tensor:
    g = torch.Generator().manual_seed(seed)
    return torch.randn(batch, style_dim, generator=g)


def make_scene(size: int, batch: int, origin_location: str = "center_fixed", style_dim: int = 256,
               pose_seed: int = 1234, tri_seed: int = 7, mlp_seed: int = 11, z_seed: int = 13,
               shared_triplane: bool = False) -> Dict[str, object]:
    """Bundle of every input of one synthetic render call (all CPU float32 tensors)."""
    num_parts = 24 if origin_location == "center+head" else 23
    pose, bone_length = random_pose(batch, pose_seed)
    K, Kinv = intrinsics(size, batch)
    tri = make_triplane(1 if shared_triplane else batch, num_parts, tri_seed,
                        origin_location=origin_location)
    if shared_triplane and batch > 1:
        tri = tri.expand(batch, -1, -1, -1)
    return {
        "size": size, "batch": batch, "origin_location": origin_location, "num_parts": num_parts,
        "parents": SMPL_PARENTS, "canonical_pose": canonical_pose(),
        "pose_to_camera": pose, "bone_length": bone_length,
        "intrinsics": K, "inv_intrinsics": Kinv,
        "image_coord": pixel_centres(size, batch),
        "tri_plane": tri, "mlp": make_mlp_params(style_dim, mlp_seed),
        "z_rend": make_z_rend(batch, style_dim, z_seed), "coordinate_scale": 3.0,
    }


class AttrDict(dict):
    """dict with attribute access: stands in for the reference's EasyDict configs (libraries/config.py)"""
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def nerf_config(**overrides) -> AttrDict:
    """`nerf_params` with the keys the render path reads and the shipping defaults
    (configs/enarfgan_train/SURREAL/config.yml:15-21, configs/DSO_demo/default.yml:16-34)."""
    c = AttrDict(hidden_size=32, Nc=48, Nf=64, origin_location="center_fixed", coordinate_scale=3, render_bs=16384,
                 no_ray_direction=True, multiply_density_with_triplane_wieght=False, clamp_mask=False, constant_triplane=True,
                 constant_trimask=False, constant_trimask_lr_mul=1, deformation_field=False, selector_mlp=False,
                 no_selector=False, time_conditional=True, pose_conditional=False)
    c.update(overrides)
    return c


def canonical_buffers(scene: Dict, origin_location: str = "center_fixed", style_dim: int = 20):
    """(canonical_pose (P,4,4), canonical_bone_length (P,1)) of a synthetic scene from the product's own
    TriPlaneNARF.register_canonical_pose (models/narf.py:84-120) - what the measurement tools hand to the C ABI."""
    from .models.narf import TriPlaneNARF
    m = TriPlaneNARF(nerf_config(origin_location=origin_location), style_dim, 24, parent=scene["parents"], num_bone_param=23)
    m.register_canonical_pose(scene["canonical_pose"])
    return m.canonical_pose, m.canonical_bone_length
