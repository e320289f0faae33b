"""ORACLE (test infrastructure, not product code): CPU restatement of ENARF-GAN's per-ray renderer.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
module. The product path (`enarf_gan_amd`) never does and fails loudly without its HIP library.

What it restates (file:line into the reference, nogu-atsu/ENARF-GAN):
  transform_pose                 libraries/NARF/pose_utils.py:129-148
  register_canonical_pose        models/narf.py:84-120
  to_local_and_canonical         models/narf.py:147-174
  in_cube / validity             libraries/NeRF/utils.py:35-43, models/narf.py:200-204
  sample_feature (grid_sample)   libraries/triplane/sampling.py:9-51 (+ ATen grid_sampler_2d,
                                 bilinear / zeros / align_corners=False)
  sample_triplane_part_prob      libraries/triplane/sampling.py:54-76
  sample_weighted_feature_v2     libraries/triplane/sampling.py:79-127
  ModulatedConv1d / StyledConv   libraries/custom_stylegan2/net.py:194-254, :270-320
  StyledMLP                      libraries/NeRF/net.py:10-27
  calc_density_and_color_...     libraries/triplane/triplane_nerf.py:32-48
  backbone / query               models/narf.py:176-275
  decide_frustrum_range          libraries/NeRF/rendering.py:10-79
  coarse_sample                  libraries/NeRF/rendering.py:82-135
  coarse_to_fine_sample          libraries/NeRF/rendering.py:138-224
  render                         libraries/NeRF/rendering.py:227-359
  triplane_sampler (fwd + bwd)   cuda_extension/TriplaneSampler_kernel.cu:13-229

Parity pinning: the reference ships no tests or golden vectors. This restatement is
pinned against outputs of the reference's own Python code imported in the build container
(`tests/golden/make_golden.py` -> `tests/golden/*.npz`; checked by `tests/test_oracle_golden.py`).

Arithmetic conventions (shared bit-for-bit with the HIP kernels so that validity masks can be
compared exactly):
  * every 3x3 product is spelled out as ((a0*b0 + a1*b1) + a2*b2) with separately rounded
    multiplies and adds (no FMA contraction); the reference uses torch.matmul whose summation
    order is backend-defined, so masks can differ from the reference only for points within a
    few ulp of a cube face;
  * for batch > 1 the reference remaps x into a (257*B)-wide concatenated plane
    (sampling.py:34-38, :96-97) which loses precision proportional to B; the oracle samples each
    image's own planes - identical in exact arithmetic;
  * the depth tables use torch.linspace's symmetric formula, spelled out in `linspace_sym`.

dtype: every function follows the dtype of its inputs; pass float64 tensors for a referee run.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

FEAT_DIM = 32
NEAR_PLANE = 0.3       # rendering.py:249
FAR_PLANE = 5.0        # rendering.py:83 default, never overridden by render()
N_RANGE_SAMPLES = 32   # rendering.py:18


# ----------------------------------------------------------------------------- pose plumbing
def transform_pose(pose_to_camera: torch.Tensor, bone_length: torch.Tensor, origin_location: str,
                   parent_id: np.ndarray) -> Tuple[torch.Tensor, torch.Tensor]:
    """24 joints -> P part frames (pose_utils.py:129-148)."""
    par = torch.as_tensor(np.asarray(parent_id)[1:], dtype=torch.long)
    mid = (pose_to_camera[:, 1:, :, 3:] + pose_to_camera[:, par, :, 3:]) / 2
    if origin_location == "center":
        pose = torch.cat([pose_to_camera[:, 1:, :, :3], mid], dim=-1)
    elif origin_location == "center_fixed":
        pose = torch.cat([pose_to_camera[:, par, :, :3], mid], dim=-1)
    elif origin_location == "center+head":
        bone_length = torch.cat([bone_length, torch.ones(bone_length.shape[0], 1, 1,
                                                         dtype=bone_length.dtype)], dim=1)
        _pose = torch.cat([pose_to_camera[:, par, :, :3], mid], dim=-1)
        pose = torch.cat([_pose, pose_to_camera[:, 15][:, None]], dim=1)
    else:
        raise ValueError(origin_location)
    return pose, bone_length


def register_canonical_pose(pose: np.ndarray, parent_id: np.ndarray, origin_location: str
                            ) -> Tuple[torch.Tensor, torch.Tensor]:
    """canonical_pose (P,4,4) and canonical_bone_length (P,) float32 buffers (narf.py:84-120)."""
    pose = np.asarray(pose)
    par = np.asarray(parent_id)[1:]
    coordinate = pose[:, :3, 3]
    length = np.linalg.norm(coordinate[1:] - coordinate[par], axis=1)
    mid = (pose[1:, :, 3:] + pose[par, :, 3:]) / 2
    if origin_location == "center":
        cpose = np.concatenate([pose[1:, :, :3], mid], axis=-1)
    elif origin_location == "center_fixed":
        cpose = np.concatenate([pose[par, :, :3], mid], axis=-1)
    elif origin_location == "center+head":
        length = np.concatenate([length, np.ones(1)])
        _pose = np.concatenate([pose[par, :, :3], mid], axis=-1)
        cpose = np.concatenate([_pose, pose[15][None]])
    else:
        raise ValueError(origin_location)
    return torch.tensor(cpose, dtype=torch.float32), torch.tensor(length, dtype=torch.float32)


def scale_pose_translation(pose: torch.Tensor, coordinate_scale: float) -> torch.Tensor:
    """rendering.py:258-260."""
    if coordinate_scale != 1:
        pose = pose.clone()
        pose[:, :, :3, 3] *= coordinate_scale
    return pose


def canonical_scale(canonical_bone_length: torch.Tensor, bone_length: torch.Tensor,
                    coordinate_scale: float) -> torch.Tensor:
    """s[b,k] = canonical_bone_length[k] / bone_length[b,k] / coordinate_scale (narf.py:165)."""
    cbl = canonical_bone_length.to(bone_length.dtype)
    return (cbl[:, None] / bone_length / coordinate_scale)[:, :, 0]   # (B, P)


# ----------------------------------------------------------------------------- bone transforms
def to_local_and_canonical(points: torch.Tensor, pose: torch.Tensor, scale: torch.Tensor,
                           canonical_pose: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """points (B,3,N), pose (B,P,4,4) [translation already x coordinate_scale], scale (B,P)
    -> local (B,P,3,N), canonical (B,P,3,N)   (narf.py:147-174), fixed op order."""
    dt = points.dtype
    R = pose[:, :, :3, :3].to(dt)
    t = pose[:, :, :3, 3].to(dt)
    d = points[:, None] - t[..., None]                     # (B,P,3,N)
    d0, d1, d2 = d[:, :, 0], d[:, :, 1], d[:, :, 2]
    loc = []
    for i in range(3):   # local_i = sum_j R[j][i] * d_j   (inv_R = R^T)
        loc.append((R[:, :, 0, i, None] * d0 + R[:, :, 1, i, None] * d1) + R[:, :, 2, i, None] * d2)
    s = scale.to(dt)[..., None]
    q0, q1, q2 = loc[0] * s, loc[1] * s, loc[2] * s
    Rc = canonical_pose[:, :3, :3].to(dt)
    tc = canonical_pose[:, :3, 3].to(dt)
    can = []
    for i in range(3):
        can.append(((Rc[None, :, i, 0, None] * q0 + Rc[None, :, i, 1, None] * q1)
                    + Rc[None, :, i, 2, None] * q2) + tc[None, :, i, None])
    return torch.stack(loc, dim=2), torch.stack(can, dim=2)


def validity(local: torch.Tensor, canonical: torch.Tensor) -> torch.Tensor:
    """(B,P,N) bool: all|local|<=1 (utils.py:42, inclusive) and all|canonical|<1 (narf.py:201, strict)."""
    return (local.abs() <= 1).all(dim=2) & (canonical.abs() < 1).all(dim=2)


# ----------------------------------------------------------------------------- bilinear sampling
def _unnormalize(coord: torch.Tensor, size: int) -> torch.Tensor:
    """ATen grid_sampler_unnormalize, align_corners=False: ((x + 1) * size - 1) / 2."""
    return ((coord + 1) * size - 1) / 2


def bilinear_taps(x: torch.Tensor, y: torch.Tensor, W: int, H: int):
    """Indices, weights and in-bounds masks of the 4 taps (nw, ne, sw, se), zeros padding.

    Follows ATen's grid_sampler_2d / the reference kernel (TriplaneSampler_kernel.cu:40-58)."""
    ix = _unnormalize(x, W)
    iy = _unnormalize(y, H)
    ix0 = torch.floor(ix)
    iy0 = torch.floor(iy)
    ix1 = ix0 + 1
    iy1 = iy0 + 1
    w_nw = (ix1 - ix) * (iy1 - iy)
    w_ne = (ix - ix0) * (iy1 - iy)
    w_sw = (ix1 - ix) * (iy - iy0)
    w_se = (ix - ix0) * (iy - iy0)
    taps = []
    for (xx, yy, ww) in ((ix0, iy0, w_nw), (ix1, iy0, w_ne), (ix0, iy1, w_sw), (ix1, iy1, w_se)):
        inb = (xx >= 0) & (xx <= W - 1) & (yy >= 0) & (yy <= H - 1)
        xi = xx.clamp(0, W - 1).long()
        yi = yy.clamp(0, H - 1).long()
        taps.append((xi, yi, ww, inb))
    return taps


def sample_plane(plane: torch.Tensor, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """plane (C,H,W), x/y (M,) in [-1,1] -> (C,M): explicit bilinear, zeros padding."""
    C, H, W = plane.shape
    flat = plane.reshape(C, H * W)
    out = torch.zeros(C, x.shape[0], dtype=plane.dtype)
    for (xi, yi, ww, inb) in bilinear_taps(x, y, W, H):
        v = flat[:, yi * W + xi]
        out = out + v * (ww * inb.to(ww.dtype))[None]
    return out


def sample_triplane_sum(planes: torch.Tensor, pos: torch.Tensor, use_grid_sample: bool = False
                        ) -> torch.Tensor:
    """planes (3C,H,W), pos (3,M) -> (C,M): sum over planes xy, yz, zx.

    Plane p is sampled at (x, y) = (pos[p], pos[(p+1)%3]) (sampling.py:30, kernel.cu:37-38)."""
    C3, H, W = planes.shape
    C = C3 // 3
    if use_grid_sample:
        grid = torch.stack([pos[[0, 1, 2]], pos[[1, 2, 0]]], dim=-1)[:, :, None]      # (3,M,1,2)
        out = F.grid_sample(planes.reshape(3, C, H, W), grid, align_corners=False)      # (3,C,M,1)
        return out[..., 0].sum(dim=0)
    out = None
    for p in range(3):
        v = sample_plane(planes[p * C:(p + 1) * C], pos[p], pos[(p + 1) % 3])
        out = v if out is None else out + v
    return out


def part_prob(mask_planes: torch.Tensor, canonical: torch.Tensor, valid: torch.Tensor,
              use_grid_sample: bool = False, clamp_mask: bool = False) -> torch.Tensor:
    """weight (B,P,N) = prod_plane sigmoid(bilinear(mask plane, masked position))  (sampling.py:54-76,
    :43-48; narf.py:237-240). Invalid (part, point) pairs sit at coordinate 2 -> sample 0 -> 0.125."""
    B, P, _, N = canonical.shape
    dt = canonical.dtype
    masked = canonical * valid[:, :, None] + 2 * (~valid[:, :, None]).to(dt)
    w = torch.empty(B, P, N, dtype=dt)
    for b in range(B):
        for k in range(P):
            planes = mask_planes[b, 3 * k:3 * k + 3].to(dt)                            # (3,H,W)
            pos = masked[b, k]
            prod = None
            for p in range(3):
                if use_grid_sample:
                    grid = torch.stack([pos[p], pos[(p + 1) % 3]], dim=-1)[None, :, None]
                    v = F.grid_sample(planes[p][None, None], grid, align_corners=False)[0, 0, :, 0]
                else:
                    v = sample_plane(planes[p][None], pos[p], pos[(p + 1) % 3])[0]
                if clamp_mask:      # value clamped, gradient straight through (sampling.py:46-47)
                    v = (v.detach().clamp(-2, 5) - v.detach()) + v
                s = torch.sigmoid(v)
                prod = s if prod is None else prod * s
            w[b, k] = prod
    return w


def weighted_feature(feat_planes: torch.Tensor, canonical: torch.Tensor, weight: torch.Tensor,
                     valid: torch.Tensor, use_grid_sample: bool = False) -> torch.Tensor:
    """feature (B,32,N) = sum over valid parts k (ascending) of weight[b,k,i] * sum_plane bilinear(...)
    (sampling.py:79-127). Only valid pairs are sampled, as in the reference."""
    B, P, _, N = canonical.shape
    dt = canonical.dtype
    out = torch.zeros(B, FEAT_DIM, N, dtype=dt)
    for b in range(B):
        planes = feat_planes[b].to(dt)
        for k in range(P):
            idx = torch.where(valid[b, k])[0]
            if idx.numel() == 0:
                continue
            pos = canonical[b, k][:, idx]
            v = sample_triplane_sum(planes, pos, use_grid_sample) * weight[b, k, idx][None]
            out[b][:, idx] = out[b][:, idx] + v
    return out


# ----------------------------------------------------------------------------- styled MLP
def modulated_weights(mlp: Dict[str, torch.Tensor], z_rend: torch.Tensor
                      ) -> List[Tuple[torch.Tensor, torch.Tensor]]:
    """Per-image demodulated 1x1-conv weights of the 3 StyledConv1d layers.

    s = z W_mod^T / sqrt(style_dim) + b_mod (EqualLinear, net.py:128-174, lr_mul 1);
    W' = normalize_rows(W * s / sqrt(in)) with F.normalize's eps 1e-12 (net.py:236-243).
    Returns [(W' (B,out,in), bias (out,))] * 3."""
    dt = z_rend.dtype
    out = []
    for i in range(3):
        W = mlp[f"layers.{i}.conv.weight"].to(dt)[0, :, :, 0]            # (out, in)
        Wm = mlp[f"layers.{i}.conv.modulation.weight"].to(dt)           # (in, style)
        bm = mlp[f"layers.{i}.conv.modulation.bias"].to(dt)
        bias = mlp[f"layers.{i}.bias"].to(dt).reshape(-1)
        style_dim = Wm.shape[1]
        s = F.linear(z_rend, Wm * (1 / math.sqrt(style_dim)), bias=bm)  # (B, in)
        w = (1 / math.sqrt(W.shape[1])) * W[None] * s[:, None, :]        # (B, out, in)
        w = F.normalize(w, dim=-1)
        out.append((w, bias))
    return out


def styled_mlp(feature: torch.Tensor, weights: List[Tuple[torch.Tensor, torch.Tensor]]) -> torch.Tensor:
    """(B,32,N) -> (B,4,N); every layer (also the last) is LeakyReLU(0.2)*sqrt(2) (net.py:313-320)."""
    h = feature
    for (w, bias) in weights:
        h = torch.bmm(w, h) + bias[None, :, None]
        h = F.leaky_relu(h, 0.2) * 2 ** 0.5
    return h


class MyReLU(torch.autograd.Function):
    """libraries/NeRF/activation.py:5-16: ReLU whose backward lets negative gradients through the negative region
    with slope 0.1 ("avoid zero gradient in the negative region")."""

    @staticmethod
    def forward(ctx, inp):
        ctx.save_for_backward(inp)
        return F.relu(inp)

    @staticmethod
    def backward(ctx, grad_output):
        inp, = ctx.saved_tensors
        return grad_output * (inp >= 0) + grad_output * (inp < 0) * (grad_output < 0) * 0.1


# ----------------------------------------------------------------------------- the query (a9)
def query(points: torch.Tensor, pose_scaled: torch.Tensor, scale: torch.Tensor,
          canonical_pose: torch.Tensor, tri_plane: torch.Tensor,
          weights: List[Tuple[torch.Tensor, torch.Tensor]], use_grid_sample: bool = False,
          multiply_density_with_weight: bool = False, return_taps: bool = False, clamp_mask: bool = False,
          no_selector: bool = False):
    """calc_density_and_color_from_camera_coord_v2 (narf.py:176-211) + backbone (:213-275).

    points (B,3,N) camera coords in the scaled space; pose_scaled (B,P,4,4).
    Returns density (B,1,N), color (B,3,N), valid (B,P,N) [+ taps dict]."""
    local, canonical = to_local_and_canonical(points, pose_scaled, scale, canonical_pose)
    valid = validity(local, canonical)
    P = pose_scaled.shape[1]
    if no_selector:          # models/narf.py:133-134: every entry of the weight tensor is 1 / P
        w = torch.full(valid.shape, 1.0 / P, dtype=points.dtype)
    else:
        w = part_prob(tri_plane[:, 3 * FEAT_DIM:], canonical, valid, use_grid_sample, clamp_mask)
    feat = weighted_feature(tri_plane[:, :3 * FEAT_DIM], canonical, w, valid, use_grid_sample)
    h = styled_mlp(feat, weights)
    color = torch.tanh(h[:, :3])
    density = MyReLU.apply(h[:, 3:])
    if multiply_density_with_weight:
        density = density * (10 * w.max(dim=1, keepdim=True)[0])
    else:
        density = density * 10
    density = density * valid.any(dim=1, keepdim=True)
    if return_taps:
        return density, color, valid, {"local": local, "canonical": canonical, "weight": w,
                                       "feature": feat, "mlp_out": h}
    return density, color, valid


# ----------------------------------------------------------------------------- ray set-up
def linspace_sym(start: float, end: float, steps: int, dtype=torch.float32) -> torch.Tensor:
    """torch.linspace's formula: i < steps//2 ? start + step*i : end - step*(steps-1-i), with
    step = (end - start)/(steps - 1), each op rounded in `dtype`."""
    s = torch.tensor(start, dtype=dtype)
    e = torch.tensor(end, dtype=dtype)
    step = (e - s) / (steps - 1)
    i = torch.arange(steps, dtype=dtype)
    lo = s + step * i
    hi = e - step * (steps - 1 - i)
    return torch.where(torch.arange(steps) < steps // 2, lo, hi)


def near_far(pose_scaled: torch.Tensor) -> Tuple[float, float]:
    """Batch-global planes (rendering.py:15-17): max(min_z - sqrt3, 0.3), max(max_z + sqrt3, 5)."""
    jz = pose_scaled[:, :, 2, 3].float()
    s3 = torch.tensor(3 ** 0.5, dtype=torch.float32)
    near = torch.clamp_min(jz.min() - s3, NEAR_PLANE)
    far = torch.clamp_min(jz.max() + s3, FAR_PLANE)
    return float(near), float(far)


def ray_directions(image_coord: torch.Tensor, inv_intrinsics: torch.Tensor) -> torch.Tensor:
    """(B,1,3,n), (B,3,3) -> (B,3,n): K^-1 [u,v,w], fixed op order (rendering.py:26-38)."""
    B = image_coord.shape[0]
    c = image_coord.reshape(B, 3, -1)
    Ki = inv_intrinsics.to(c.dtype)
    if Ki.ndim == 2:
        Ki = Ki[None].expand(B, -1, -1)
    rows = []
    for i in range(3):
        rows.append((Ki[:, i, 0, None] * c[:, 0] + Ki[:, i, 1, None] * c[:, 1]) + Ki[:, i, 2, None] * c[:, 2])
    return torch.stack(rows, dim=1)


def frustum_range(ray_dir: torch.Tensor, pose_scaled: torch.Tensor, near: float, far: float):
    """decide_frustrum_range (rendering.py:10-79, return_camera_coord=True branch).

    Returns depth_min (B,n), depth_max (B,n), validity (B,n) bool."""
    dt = ray_dir.dtype
    B, _, n = ray_dir.shape
    depths = linspace_sym(near, far, N_RANGE_SAMPLES, dt)                     # (32,)
    R = pose_scaled[:, :, :3, :3].to(dt)
    t = pose_scaled[:, :, :3, 3].to(dt)
    large = 1e3
    dmin = torch.full((B, n), large, dtype=dt)
    dmax = torch.full((B, n), -large, dtype=dt)
    for s in range(N_RANGE_SAMPLES):
        p = ray_dir * depths[s]                                                # (B,3,n)
        d = p[:, None] - t[..., None]                                          # (B,P,3,n)
        inside = None
        for i in range(3):
            li = (R[:, :, 0, i, None] * d[:, :, 0] + R[:, :, 1, i, None] * d[:, :, 1]) + R[:, :, 2, i, None] * d[:, :, 2]
            ok = li.abs() <= 1
            inside = ok if inside is None else (inside & ok)
        any_in = inside.any(dim=1)                                             # (B,n)
        dmin = torch.where(any_in, torch.minimum(dmin, depths[s]), dmin)
        dmax = torch.where(any_in, torch.maximum(dmax, depths[s]), dmax)
    valid = dmin != large
    dmin = torch.where(valid, dmin, torch.full_like(dmin, near))
    dmax = torch.where(dmax != -large, dmax, torch.full_like(dmax, far))
    dmin = torch.clamp_min(dmin, near)
    return dmin, dmax, valid


def coarse_points(ray_dir: torch.Tensor, dmin: torch.Tensor, dmax: torch.Tensor, Nc: int):
    """coarse_sample (rendering.py:119-131): Nc+1 bin edges, mid-points of consecutive edge points.

    ray_dir (B,3,n), dmin/dmax (B,n) -> coarse_depth (B,n,Nc+1), points (B,3,n,Nc)."""
    dt = ray_dir.dtype
    bins = linspace_sym(0.0, 1.0, Nc + 1, torch.float32).to(dt)
    start = dmin[:, None] * ray_dir
    end = dmax[:, None] * ray_dir
    depth = dmin[..., None] * (1 - bins) + dmax[..., None] * bins
    edge = start[..., None] * (1 - bins) + end[..., None] * bins
    mid = (edge[..., 1:] + edge[..., :-1]) / 2
    return depth, mid, start, end


def ray_weights(density: torch.Tensor, depth: torch.Tensor, render_scale: float = 1.0):
    """density (B,n,M) on M intervals, depth (B,n,M+1) -> T (B,n,M), weights (B,n,M)
    (rendering.py:180-184, :316-321)."""
    delta = depth[..., 1:] - depth[..., :-1]
    dd = density * delta * render_scale
    T = torch.exp(-(torch.cumsum(dd, dim=-1) - dd))
    return T, T * (1 - torch.exp(-dd))


def smooth_weights(w: torch.Tensor) -> torch.Tensor:
    """rendering.py:188-190 on the last axis."""
    wp = F.pad(w, (1, 1))
    return (torch.maximum(wp[..., :-2], wp[..., 1:-1]) + torch.maximum(wp[..., 1:-1], wp[..., 2:])) / 2 + 0.01


def draw_bins(w_smooth: torch.Tensor, Nf: int, Nc: int, generator: Optional[torch.Generator] = None
              ) -> torch.Tensor:
    """bins = sort(multinomial(w, Nf, replacement)/Nc + U[0,1)/Nc) (rendering.py:192-197). (…,Nc)->(…,Nf)."""
    shp = w_smooth.shape[:-1]
    flat = w_smooth.reshape(-1, w_smooth.shape[-1]).float()
    idx = torch.multinomial(flat, Nf, replacement=True, generator=generator).float()
    u = torch.rand(flat.shape[0], Nf, generator=generator)
    bins = idx / Nc + u / Nc
    return torch.sort(bins, dim=-1)[0].reshape(*shp, Nf).to(w_smooth.dtype)


def fine_points(bins: torch.Tensor, dmin, dmax, start, end):
    """rendering.py:198-200. bins (B,n,Nf) -> fine_depth (B,n,Nf), fine points (B,3,n,Nf)."""
    depth = dmin[..., None] * (1 - bins) + dmax[..., None] * bins
    pts = start[..., None] * (1 - bins[:, None]) + end[..., None] * bins[:, None]
    return depth, pts


def composite(density: torch.Tensor, color: torch.Tensor, depth: torch.Tensor, render_scale: float = 1.0):
    """rendering.py:307-335: integrate the first Nf-1 fine samples.

    density (B,n,Nf), color (B,3,n,Nf), depth (B,n,Nf) -> color (B,3,n), mask (B,n), disparity (B,n),
    weights (B,n,Nf-1)."""
    T, w = ray_weights(density[..., :-1], depth, render_scale)
    rc = torch.sum(w[:, None] * color[..., :-1], dim=-1)
    rm = torch.sum(w, dim=-1)
    rd = torch.sum(w * 1 / depth[..., :-1], dim=-1)
    return rc, rm, rd, w


def render(image_coord: torch.Tensor, pose_parts: torch.Tensor, bone_length_parts: torch.Tensor,
           inv_intrinsics: torch.Tensor, canonical_pose: torch.Tensor, canonical_bone_length: torch.Tensor,
           tri_plane: torch.Tensor, mlp: Dict[str, torch.Tensor], z_rend: torch.Tensor,
           coordinate_scale: float = 3.0, Nc: int = 48, Nf: int = 64, render_scale: float = 1.0,
           bins: Optional[torch.Tensor] = None, generator: Optional[torch.Generator] = None,
           use_grid_sample: bool = False, return_taps: bool = False, multiply_density_with_weight: bool = False,
           clamp_mask: bool = False, no_selector: bool = False, near_far_planes=None):
    """render() (rendering.py:227-359) after transform_pose: pose_parts (B,P,4,4) unscaled.
    `near_far_planes` = (near, far) of a LARGER batch these images were taken from (the planes are reduced over the whole
    batch, rendering.py:15-17): lets a test restate some images of a batch without restating all of them.

    `bins` (B,n,Nf), sorted, replaces the random importance samples (for parity runs); rays that
    the reference would drop (B == 1 and no cube hit, rendering.py:107-110) produce zeros and their
    bins are ignored. Returns color (B,3,n), mask (B,n), disparity (B,n) [+ taps]."""
    dt = image_coord.dtype
    B = image_coord.shape[0]
    n = image_coord.shape[-1]
    pose = scale_pose_translation(pose_parts, coordinate_scale).to(dt)
    scale = canonical_scale(canonical_bone_length, bone_length_parts.to(dt), coordinate_scale)
    weights = modulated_weights(mlp, z_rend.to(dt))
    near, far = near_far(pose) if near_far_planes is None else near_far_planes
    rd = ray_directions(image_coord, inv_intrinsics.to(dt))
    dmin, dmax, rvalid = frustum_range(rd, pose, near, far)
    drop = (~rvalid) if B == 1 else torch.zeros_like(rvalid)

    cdepth, cpts, start, end = coarse_points(rd, dmin, dmax, Nc)
    qkw = dict(multiply_density_with_weight=multiply_density_with_weight, clamp_mask=clamp_mask, no_selector=no_selector)
    cden, _, cvalid = query(cpts.reshape(B, 3, -1), pose, scale, canonical_pose, tri_plane.to(dt), weights,
                            use_grid_sample, **qkw)
    cden = cden.reshape(B, n, Nc)
    _, cw = ray_weights(cden, cdepth, render_scale)
    cws = smooth_weights(cw)
    if bins is None:
        bins = draw_bins(cws, Nf, Nc, generator)
    fdepth, fpts = fine_points(bins.to(dt), dmin, dmax, start, end)
    fden, fcol, fvalid = query(fpts.reshape(B, 3, -1), pose, scale, canonical_pose, tri_plane.to(dt), weights,
                               use_grid_sample, **qkw)
    fden = fden.reshape(B, n, Nf)
    fcol = fcol.reshape(B, 3, n, Nf)
    rc, rm, rdisp, fw = composite(fden, fcol, fdepth, render_scale)
    keep = (~drop).to(dt)
    rc, rm, rdisp = rc * keep[:, None], rm * keep, rdisp * keep
    if return_taps:
        taps = {"near": near, "far": far, "ray_dir": rd, "depth_min": dmin, "depth_max": dmax,
                "ray_validity": rvalid, "coarse_depth": cdepth, "coarse_density": cden,
                "coarse_weights_smooth": cws, "coarse_valid": cvalid.reshape(B, -1, n, Nc),
                "bins": bins, "fine_depth": fdepth, "fine_density": fden, "fine_color": fcol,
                "fine_valid": fvalid.reshape(B, -1, n, Nf), "fine_weights": fw}
        return rc, rm, rdisp, taps
    return rc, rm, rdisp


# ----------------------------------------------------------------------------- the a1 operator
def triplane_sampler_forward(inp: torch.Tensor, grid: torch.Tensor) -> torch.Tensor:
    """triplane_sampler forward, bilinear / zeros / align_corners=False (kernel.cu:13-92).

    inp (B,3C,H,W), grid (B,h,w,3) -> (B,C,h,w)."""
    B, C3, H, W = inp.shape
    _, h, w, _ = grid.shape
    out = torch.empty(B, C3 // 3, h, w, dtype=inp.dtype)
    for b in range(B):
        pos = grid[b].reshape(-1, 3).t()
        out[b] = sample_triplane_sum(inp[b], pos).reshape(C3 // 3, h, w)
    return out


def triplane_sampler_backward(grad_out: torch.Tensor, inp: torch.Tensor, grid: torch.Tensor
                              ) -> Tuple[torch.Tensor, torch.Tensor]:
    """Gradients of triplane_sampler_forward w.r.t. (input, grid) via autograd through the explicit
    bilinear formula; equals kernel.cu:94-229 (the true gradients; the reference wrapper's
    inverted output_mask bug, triplane_sampler.py:59-62, is deliberately not reproduced)."""
    inp = inp.detach().clone().requires_grad_(True)
    grid = grid.detach().clone().requires_grad_(True)
    out = triplane_sampler_forward(inp, grid)
    gi, gg = torch.autograd.grad(out, (inp, grid), grad_out)
    return gi, gg
