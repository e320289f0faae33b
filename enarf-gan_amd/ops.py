"""Tensor-level wrappers of the C ABI: torch supplies device memory and the current HIP stream,
nothing else. Every function here requires CUDA(=HIP) tensors and raises otherwise.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import FEAT_DIM, MARCH, MLP_MODE, ORIGIN

PLANE_CH = 3 * FEAT_DIM   # 96 feature channels precede the part-probability planes (models/narf.py:239,255)


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _first_cuda_tensor(args, kwargs) -> Optional[torch.Tensor]:
    for v in list(args) + list(kwargs.values()):
        if isinstance(v, torch.Tensor) and v.is_cuda:
            return v
        if isinstance(v, dict):
            for w in v.values():
                if isinstance(w, torch.Tensor) and w.is_cuda:
                    return w
    return None


def _on_tensor_device(fn):
    """The library launches on the stream it is handed, and a HIP stream belongs to one device: make the device of the
    call's first device tensor current for the duration of the call (a process that drives several GPUs may have another
    one current)."""
    import functools

    @functools.wraps(fn)
    def wrapped(*args, **kwargs):
        t = _first_cuda_tensor(args, kwargs)
        if t is None or t.device.index == torch.cuda.current_device():
            return fn(*args, **kwargs)
        with torch.cuda.device(t.device):
            return fn(*args, **kwargs)
    return wrapped


def _stream(dev: torch.device) -> int:
    return torch.cuda.current_stream(dev).cuda_stream


def _dev_f32(t: torch.Tensor, what: str) -> torch.Tensor:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.EnarfHipError(f"{what} must be a CUDA/HIP tensor (enarf_gan_amd has no CPU path)")
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def mlp_pack_bytes() -> int:
    return int(_lib.load().enarf_mlp_pack_bytes())


def num_parts(num_joints: int, origin_location: str) -> int:
    return num_joints if origin_location == "center+head" else num_joints - 1


# ---------------------------------------------------------------------------------------- a1 operator
@_on_tensor_device
def triplane_sample_fwd(inp: torch.Tensor, grid: torch.Tensor, mode: int = 0, padding_mode: int = 0,
                        align_corners: bool = False, use_workspace: bool = True) -> torch.Tensor:
    """input (B,3C,H,W), grid (B,h,w,3) -> (B,C,h,w); cuda_extension/TriplaneSampler.cpp:15-24."""
    lib = _lib.load()
    inp = _dev_f32(inp, "input")
    grid = _dev_f32(grid, "grid")
    B, C3, H, W = inp.shape
    if C3 % 3 != 0 or grid.shape[0] != B or grid.shape[-1] != 3 or grid.dim() != 4:
        raise ValueError(f"triplane_sampler: input {tuple(inp.shape)} / grid {tuple(grid.shape)} shapes do not match "
                         "(B,3C,H,W) / (B,h,w,3)")
    h, w = grid.shape[1], grid.shape[2]
    Cc = C3 // 3
    out = torch.empty(B, Cc, h, w, dtype=torch.float32, device=inp.device)
    ws = None
    if use_workspace:
        nbytes = lib.enarf_triplane_sample_workspace_bytes(B, Cc, H, W)
        if nbytes:
            ws = torch.empty(nbytes // 4, dtype=torch.float32, device=inp.device)
    rc = lib.enarf_triplane_sample_fwd(_p(inp), _p(grid), _p(out), B, Cc, H, W, h * w, mode, padding_mode,
                                       int(bool(align_corners)), _p(ws), _stream(inp.device))
    _lib.check(rc, "enarf_triplane_sample_fwd")
    return out


@_on_tensor_device
def triplane_sample_bwd(grad_out: torch.Tensor, inp: torch.Tensor, grid: torch.Tensor, mode: int, padding_mode: int,
                        align_corners: bool, need_input: bool, need_grid: bool, use_workspace: bool = True
                        ) -> Tuple[Optional[torch.Tensor], Optional[torch.Tensor]]:
    """cuda_extension/TriplaneSampler.cpp:26-52; returns (grad_input, grad_grid), None where not needed."""
    lib = _lib.load()
    grad_out = _dev_f32(grad_out, "grad_output")
    inp = _dev_f32(inp, "input")
    grid = _dev_f32(grid, "grid")
    B, C3, H, W = inp.shape
    h, w = grid.shape[1], grid.shape[2]
    gi = torch.zeros_like(inp) if need_input else None
    gg = torch.empty_like(grid) if need_grid else None
    ws = None
    if use_workspace:
        nbytes = lib.enarf_triplane_sample_bwd_workspace_bytes(B, C3 // 3, H, W)
        if nbytes:
            ws = torch.empty(nbytes // 4, dtype=torch.float32, device=inp.device)
    rc = lib.enarf_triplane_sample_bwd(_p(grad_out), _p(inp), _p(grid), _p(gi), _p(gg), B, C3 // 3, H, W, h * w,
                                       mode, padding_mode, int(bool(align_corners)), _p(ws), _stream(inp.device))
    _lib.check(rc, "enarf_triplane_sample_bwd")
    return gi, gg


@_on_tensor_device
def triplane_sample_ex_fwd(inp: torch.Tensor, grid: torch.Tensor, separate: bool = False,
                           point_image: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Generalised gather (bilinear, zeros, align_corners False): input (Bimg, 3C, H, W), grid (B, n, 3) -> (B, C, n), or
    (B, 3, C, n) with `separate` (the planes' samples side by side). point_image (n,) int32 with B == 1: point i samples
    image point_image[i]."""
    lib = _lib.load()
    inp, grid = _dev_f32(inp, "input"), _dev_f32(grid, "grid")
    Bi, C3, H, W = inp.shape
    B, n, _ = grid.shape
    Cc = C3 // 3
    pi = None if point_image is None else point_image.to(device=inp.device, dtype=torch.int32).contiguous()
    out = torch.empty((B, 3, Cc, n) if separate else (B, Cc, n), dtype=torch.float32, device=inp.device)
    _lib.check(lib.enarf_triplane_sample_ex_fwd(_p(inp), _p(grid), _p(out), B, Cc, H, W, n, 0, 0, 0, int(separate), _p(pi), Bi,
                                                _stream(inp.device)), "enarf_triplane_sample_ex_fwd")
    return out


@_on_tensor_device
def triplane_sample_ex_bwd(grad_out: torch.Tensor, inp: torch.Tensor, grid: torch.Tensor, separate: bool,
                           point_image: Optional[torch.Tensor], need_input: bool, need_grid: bool):
    lib = _lib.load()
    go, inp, grid = _dev_f32(grad_out, "grad_output"), _dev_f32(inp, "input"), _dev_f32(grid, "grid")
    Bi, C3, H, W = inp.shape
    B, n, _ = grid.shape
    pi = None if point_image is None else point_image.to(device=inp.device, dtype=torch.int32).contiguous()
    gi = torch.zeros_like(inp) if need_input else None
    gg = torch.empty_like(grid) if need_grid else None
    _lib.check(lib.enarf_triplane_sample_ex_bwd(_p(go), _p(inp), _p(grid), _p(gi), _p(gg), B, C3 // 3, H, W, n, 0, 0, 0,
                                                int(separate), _p(pi), Bi, _stream(inp.device)), "enarf_triplane_sample_ex_bwd")
    return gi, gg


# ---------------------------------------------------------------------------------------- a16 ray sampler
@_on_tensor_device
def mask_dilate_topk(mask: torch.Tensor, noise: torch.Tensor, k: int, radius: int = 64) -> torch.Tensor:
    """mask (B, h, w), noise (B, h*w) -> (B, k) int64 flat pixel ids of the k largest dilate(mask) + noise per image
    (libraries/NeRF/ray_sampler.py:23-30), unordered."""
    lib = _lib.load()
    m = _dev_f32(mask, "mask")
    nz = _dev_f32(noise, "noise")
    B, h, w = m.shape
    if nz.shape != (B, h * w):
        raise ValueError(f"noise {tuple(nz.shape)} must be (B, h*w) = {(B, h * w)}")
    out = torch.empty(B, k, dtype=torch.int64, device=m.device)
    ws = torch.empty(int(lib.enarf_mask_topk_workspace_bytes(B, h, w)) // 4, dtype=torch.float32, device=m.device)
    _lib.check(lib.enarf_mask_dilate_topk(_p(m), _p(nz), _p(out), B, h, w, int(k), int(radius), _p(ws), _stream(m.device)),
               "enarf_mask_dilate_topk")
    return out


# ---------------------------------------------------------------------------------------- re-layout
@_on_tensor_device
def triplane_pack(tri_nchw: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """(B, 96+3P, H, W) NCHW -> feature planes channel-last (B, 3, H, W, 32)."""
    lib = _lib.load()
    tri = _dev_f32(tri_nchw, "tri_plane")
    B, Ct, H, W = tri.shape
    if out is None:
        out = torch.empty(B, 3, H, W, FEAT_DIM, dtype=torch.float32, device=tri.device)
    rc = lib.enarf_triplane_pack(_p(tri), _p(out), B, Ct, H, W, _stream(tri.device))
    _lib.check(rc, "enarf_triplane_pack")
    return out


@_on_tensor_device
def triplane_unpack_add(grad_feat_cl: torch.Tensor, grad_tri_nchw: torch.Tensor) -> torch.Tensor:
    """grad_tri[:, :96] += channel-last gradient (B, 3, H, W, 32) (the inverse re-layout)."""
    lib = _lib.load()
    g = _dev_f32(grad_feat_cl, "grad_feat_cl")
    B, Ct, H, W = grad_tri_nchw.shape
    _lib.check(lib.enarf_triplane_unpack_add(_p(g), _p(grad_tri_nchw), B, Ct, H, W, _stream(g.device)),
               "enarf_triplane_unpack_add")
    return grad_tri_nchw


@_on_tensor_device
def triplane_warp_fwd(src_cl: torch.Tensor, flow: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Deformation-field producer: src_cl (1|-,3,H,W,32) channel-last constant planes, flow (B,6,H,W) -> (B,3,H,W,32)."""
    lib = _lib.load()
    src = _dev_f32(src_cl, "src_cl").reshape(3, *src_cl.shape[-3:])
    fl = _dev_f32(flow, "flow")
    B, six, H, W = fl.shape
    if six != 6 or tuple(src.shape) != (3, H, W, FEAT_DIM):
        raise ValueError(f"flow {tuple(fl.shape)} / src_cl {tuple(src_cl.shape)}: expected (B,6,H,W) and (3,H,W,{FEAT_DIM})")
    if out is None:
        out = torch.empty(B, 3, H, W, FEAT_DIM, dtype=torch.float32, device=fl.device)
    _lib.check(lib.enarf_triplane_warp_fwd(_p(src), _p(fl), _p(out), B, H, W, _stream(fl.device)), "enarf_triplane_warp_fwd")
    return out


@_on_tensor_device
def triplane_warp_bwd(g_out_cl: torch.Tensor, src_cl: torch.Tensor, flow: torch.Tensor, need_src: bool = True,
                      need_flow: bool = True):
    """-> (g_src_cl (3,H,W,32) or None, g_flow (B,6,H,W) or None)."""
    lib = _lib.load()
    g = _dev_f32(g_out_cl, "g_out_cl")
    src = _dev_f32(src_cl, "src_cl").reshape(3, *src_cl.shape[-3:])
    fl = _dev_f32(flow, "flow")
    B, _, H, W = fl.shape
    gs = torch.zeros_like(src) if need_src else None
    gf = torch.empty_like(fl) if need_flow else None
    _lib.check(lib.enarf_triplane_warp_bwd(_p(g), _p(src), _p(fl), _p(gs), _p(gf), B, H, W, _stream(fl.device)),
               "enarf_triplane_warp_bwd")
    return gs, gf


# ---------------------------------------------------------------------------------------- prepare
def _prepare_args(pose_to_camera, bone_length, canonical_bone_length, z_rend, mlp, parents, origin_location,
                  coordinate_scale, parts_out, pack_out):
    """Validated enarf_prepare_args + the tensors its pointers borrow (kept alive by the caller)."""
    pose = _dev_f32(pose_to_camera, "pose_to_camera")
    B, J = pose.shape[0], pose.shape[1]
    P = num_parts(J, origin_location)
    bl = _dev_f32(bone_length, "bone_length").reshape(B, J - 1)
    cbl = _dev_f32(canonical_bone_length, "canonical_bone_length")
    z = _dev_f32(z_rend, "z_rend")
    if cbl.numel() != P:
        raise ValueError(f"canonical_bone_length has {cbl.numel()} entries, expected {P}")
    a = _lib.PrepareArgs()
    a.B, a.num_joints, a.origin_location, a.style_dim = B, J, ORIGIN[origin_location], z.shape[1]
    a.coordinate_scale = float(coordinate_scale)
    par = np.asarray(parents, dtype=np.int64)
    for j in range(J):
        a.parents[j] = int(par[j])
    a.pose_to_camera, a.bone_length, a.canonical_bone_length, a.z_rend = _p(pose), _p(bl), _p(cbl), _p(z)
    keep = []
    dims = [(FEAT_DIM, 64), (64, 64), (64, 4)]
    for i, (cin, cout) in enumerate(dims):
        cw = _dev_f32(mlp[f"layers.{i}.conv.weight"], "conv.weight")
        mw = _dev_f32(mlp[f"layers.{i}.conv.modulation.weight"], "modulation.weight")
        mb = _dev_f32(mlp[f"layers.{i}.conv.modulation.bias"], "modulation.bias")
        bs = _dev_f32(mlp[f"layers.{i}.bias"], "bias")
        if cw.numel() != cin * cout or mw.shape != (cin, z.shape[1]) or mb.numel() != cin or bs.numel() != cout:
            raise ValueError(f"StyledMLP layer {i}: unexpected parameter shapes "
                             f"{tuple(cw.shape)} {tuple(mw.shape)} {tuple(mb.shape)} {tuple(bs.shape)}")
        keep += [cw, mw, mb, bs]
        a.conv_weight[i], a.mod_weight[i], a.mod_bias[i], a.bias[i] = _p(cw), _p(mw), _p(mb), _p(bs)
    if parts_out is None:
        parts_out = torch.empty(B, P, 16, dtype=torch.float32, device=pose.device)
    if pack_out is None:
        pack_out = torch.empty(B, mlp_pack_bytes(), dtype=torch.uint8, device=pose.device)
    a.parts, a.mlp_pack = _p(parts_out), _p(pack_out)
    keep += [pose, bl, cbl, z]
    return a, keep, parts_out, pack_out


@_on_tensor_device
def prepare(pose_to_camera: torch.Tensor, bone_length: torch.Tensor, canonical_bone_length: torch.Tensor,
            z_rend: torch.Tensor, mlp: Dict[str, torch.Tensor], parents: Sequence[int], origin_location: str,
            coordinate_scale: float, parts_out: Optional[torch.Tensor] = None,
            pack_out: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """24 joints -> part frames (B,P,16) and the per-image MLP pack (B, pack_bytes) uint8."""
    lib = _lib.load()
    a, _keep, parts_out, pack_out = _prepare_args(pose_to_camera, bone_length, canonical_bone_length, z_rend, mlp,
                                                  parents, origin_location, coordinate_scale, parts_out, pack_out)
    rc = lib.enarf_prepare(C.byref(a), _stream(parts_out.device))
    _lib.check(rc, "enarf_prepare")
    return parts_out, pack_out


@_on_tensor_device
def prepare_mlp(z_rend: torch.Tensor, mlp: Dict[str, torch.Tensor], pack_out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """enarf_prepare with parts == NULL: only the per-image modulated, demodulated MLP pack (B, pack_bytes) uint8."""
    lib = _lib.load()
    z = _dev_f32(z_rend.detach(), "z_rend")
    B = z.shape[0]
    a = _lib.PrepareArgs()
    a.B, a.num_joints, a.origin_location, a.style_dim = B, 2, ORIGIN["center_fixed"], z.shape[1]   # joints unused without parts
    a.coordinate_scale = 1.0
    a.parents[0], a.parents[1] = -1, 0
    a.z_rend = _p(z)
    keep = [z]
    dims = [(FEAT_DIM, 64), (64, 64), (64, 4)]
    for i, (cin, cout) in enumerate(dims):
        ts = [_dev_f32(mlp[f"layers.{i}.{leaf}"].detach(), leaf) for leaf in
              ("conv.weight", "conv.modulation.weight", "conv.modulation.bias", "bias")]
        if ts[0].numel() != cin * cout or ts[1].shape != (cin, z.shape[1]) or ts[2].numel() != cin or ts[3].numel() != cout:
            raise ValueError(f"StyledMLP layer {i}: unexpected parameter shapes {[tuple(t.shape) for t in ts]}")
        keep += ts
        a.conv_weight[i], a.mod_weight[i], a.mod_bias[i], a.bias[i] = [_p(t) for t in ts]
    if pack_out is None:
        pack_out = torch.empty(B, mlp_pack_bytes(), dtype=torch.uint8, device=z.device)
    a.parts, a.mlp_pack = None, _p(pack_out)
    _lib.check(lib.enarf_prepare(C.byref(a), _stream(z.device)), "enarf_prepare")
    return pack_out


@_on_tensor_device
def mlp_unpack(pack_one_image: torch.Tensor):
    """Dense (W1 (64,32), W2 (64,64), W3 (4,64), b1, b2, b3) from one image's pack (tests / interop)."""
    lib = _lib.load()
    dense = torch.empty(64 * 32 + 64 * 64 + 4 * 64 + 132, dtype=torch.float32, device=pack_one_image.device)
    _lib.check(lib.enarf_mlp_unpack(_p(pack_one_image), _p(dense), _stream(dense.device)), "enarf_mlp_unpack")
    o = 0
    outs = []
    for shp in ((64, 32), (64, 64), (4, 64), (64,), (64,), (4,)):
        n = int(np.prod(shp))
        outs.append(dense[o:o + n].reshape(shp))
        o += n
    return outs


def _plane_strides(tri_nchw: torch.Tensor, feat_cl: torch.Tensor, B: int):
    """Batch strides (in floats) of the mask planes inside tri_nchw and of feat_cl; 0 = shared tri-plane."""
    Ct, H, W = tri_nchw.shape[1:]
    mask_stride = 0 if tri_nchw.shape[0] == 1 else Ct * H * W
    feat_stride = 0 if feat_cl.shape[0] == 1 else 3 * H * W * FEAT_DIM
    if tri_nchw.shape[0] not in (1, B) or feat_cl.shape[0] not in (1, B):
        raise ValueError("tri-plane batch must be 1 (shared) or B")
    return mask_stride, feat_stride


# ---------------------------------------------------------------------------------------- a9 query
@_on_tensor_device
def query_fwd(points: Optional[torch.Tensor], parts: torch.Tensor, canonical_pose: torch.Tensor, tri_nchw: torch.Tensor,
              feat_cl: torch.Tensor, mlp_pack: torch.Tensor, mlp_mode: str = "f32",
              multiply_density_with_weight: bool = False, need_color: bool = True, need_valid: bool = False,
              debug: bool = False, grid: Optional[Tuple[int, Sequence[float], float]] = None, clamp_mask: bool = False,
              uniform_part_weight: bool = False):
    """points (B,3,N) -> density (B,1,N), color (B,3,N) [, valid_bits (B,N) int32 [, canonical, weight]].

    grid = (D, centre (3,), scale) with points = None: the D^3 lattice of create_mesh, generated in the kernel."""
    lib = _lib.load()
    P = parts.shape[1]
    if grid is not None:
        if points is not None:
            raise ValueError("pass either points or grid")
        B, N = parts.shape[0], int(grid[0]) ** 3
        pts = parts                       # device / dtype carrier only
    else:
        pts = _dev_f32(points, "points")
        B, _, N = pts.shape
    tri = _dev_f32(tri_nchw, "tri_plane")
    H, W = tri.shape[2], tri.shape[3]
    mstride, fstride = _plane_strides(tri, feat_cl, B)
    dev = pts.device
    if N == 0:   # nothing to launch (an empty tensor has no device pointer)
        def z(*shape, dt=torch.float32):
            return torch.empty(*shape, dtype=dt, device=dev)
        out = (z(B, 1, 0), z(B, 3, 0) if need_color else None)
        if need_valid or debug:
            out = out + (z(B, 0, dt=torch.int32),)
        if debug:
            out = out + (z(B, P, 3, 0), z(B, P, 0))
        return out
    den = torch.empty(B, 1, N, dtype=torch.float32, device=dev)
    col = torch.empty(B, 3, N, dtype=torch.float32, device=dev) if need_color else None
    vb = torch.empty(B, N, dtype=torch.int32, device=dev) if (need_valid or debug) else None
    dc = torch.zeros(B, P, 3, N, dtype=torch.float32, device=dev) if debug else None
    dw = torch.zeros(B, P, N, dtype=torch.float32, device=dev) if debug else None
    a = _lib.QueryArgs()
    a.B, a.N, a.P, a.H, a.W = B, N, P, H, W
    a.mlp_mode, a.multiply_density_with_weight = MLP_MODE[mlp_mode], int(multiply_density_with_weight)
    a.clamp_mask, a.uniform_part_weight = int(bool(clamp_mask)), int(bool(uniform_part_weight))
    a.points, a.parts, a.canonical_pose = (None if grid is not None else _p(pts)), _p(parts), _p(_dev_f32(canonical_pose, "canonical_pose"))
    if grid is not None:
        a.grid_D, a.grid_scale = int(grid[0]), float(grid[2])
        for j in range(3):
            a.grid_center[j] = float(grid[1][j])
    a.feat_cl, a.feat_batch_stride = _p(feat_cl), fstride
    a.mask_planes, a.mask_batch_stride = tri.data_ptr() + PLANE_CH * H * W * 4, mstride
    a.mlp_pack, a.density, a.color, a.valid_bits = _p(mlp_pack), _p(den), _p(col), _p(vb)
    a.dbg_canonical, a.dbg_weight = _p(dc), _p(dw)
    _lib.check(lib.enarf_query_fwd(C.byref(a), _stream(dev)), "enarf_query_fwd")
    out = (den, col)
    if need_valid or debug:
        out = out + (vb,)
    if debug:
        out = out + (dc, dw)
    return out


# ---------------------------------------------------------------------------------------- a13 render
_render_ws = {}


_render_epoch = {}


def _ws_key(dev: torch.device):
    return (dev.index, torch.cuda.current_stream(dev).cuda_stream)


def _render_workspace(dev: torch.device, B: int, n: int) -> torch.Tensor:
    """Workspace of enarf_render_fwd, cached per (device, stream): launches on one stream are ordered, so they can
    share it; it only grows."""
    key = _ws_key(dev)
    need = int(_lib.load().enarf_render_workspace_bytes(B, n))
    ws = _render_ws.get(key)
    if ws is None or ws.numel() * 4 < need:
        ws = torch.empty((need + 3) // 4, dtype=torch.int32, device=dev)
        _render_ws[key] = ws
        _render_epoch[key] = 0
    return ws


_render_shape = {}


class _Epoch:
    """ws_epoch bookkeeping of the cached workspace of the current stream (include/enarf_hip.h, enarf_render_args):
    `with _Epoch(dev, shape) as k:` hands out the epoch for one forward call and, when the call went through, advances the
    count; anything that raises - and every call that does not keep count, like the backward - falls back to epoch 0, and
    so does a call whose (B, n, group_frames, per-frame planes) differs from the previous one's: a batch marched in groups
    keeps one queue-header pair per group inside the workspace, at offsets that depend on those."""

    def __init__(self, dev: torch.device, shape=None, counted: bool = True):
        self.key, self.counted, self.shape = _ws_key(dev), counted, shape

    def __enter__(self) -> int:
        same = self.shape is not None and _render_shape.get(self.key) == self.shape
        self.k = _render_epoch.get(self.key, 0) if (self.counted and same) else 0
        _render_epoch[self.key] = 0
        _render_shape[self.key] = self.shape if self.counted else None
        return self.k

    def __exit__(self, exc_type, exc, tb):
        if exc_type is None:
            _render_epoch[self.key] = self.k + 1
        return False


def _shape_key(a) -> tuple:
    return (a.B, a.n, a.group_frames, a.feat_batch_stride != 0)


class RenderOutputs:
    __slots__ = ("color", "mask", "disparity", "fine_weights", "fine_depth", "taps", "counters")


@_on_tensor_device
def render_fwd(image_coord: torch.Tensor, inv_intrinsics: torch.Tensor, parts: torch.Tensor,
               canonical_pose: torch.Tensor, tri_nchw: torch.Tensor, feat_cl: torch.Tensor, mlp_pack: torch.Tensor,
               Nc: int, Nf: int, render_scale: float = 1.0, bins: Optional[torch.Tensor] = None, seed: int = 0,
               mlp_mode: str = "f32", multiply_density_with_weight: bool = False,
               drop_invalid_rays: Optional[bool] = None, want_fine: bool = True, debug: bool = False,
               count: bool = False, early_stop_eps: float = 0.0, return_bins: bool = False, clamp_mask: bool = False,
               uniform_part_weight: bool = False, march: str = "auto", group_frames: int = 0) -> RenderOutputs:
    """The fused ray march. image_coord (B,1,3,n) or (B,3,n); returns color (B,3,n), mask (B,n), disparity (B,n),
    fine_weights (B,1,n,Nf-1), fine_depth (B,1,n,Nf) and, with debug=True, the parity taps."""
    lib = _lib.load()
    a, o, _keep = _render_args(image_coord, inv_intrinsics, parts, canonical_pose, tri_nchw, feat_cl, mlp_pack, Nc, Nf,
                               render_scale, bins, seed, mlp_mode, multiply_density_with_weight, drop_invalid_rays,
                               want_fine, debug, count, early_stop_eps, return_bins, clamp_mask, uniform_part_weight, march)
    a.group_frames = int(group_frames)
    with _Epoch(o.color.device, _shape_key(a)) as k:
        a.ws_epoch = k
        _lib.check(lib.enarf_render_fwd(C.byref(a), _stream(o.color.device)), "enarf_render_fwd")
    return o


def _render_args(image_coord, inv_intrinsics, parts, canonical_pose, tri_nchw, feat_cl, mlp_pack, Nc, Nf, render_scale,
                 bins, seed, mlp_mode, multiply_density_with_weight, drop_invalid_rays, want_fine, debug, count,
                 early_stop_eps, return_bins, clamp_mask=False, uniform_part_weight=False, march="auto"):
    """enarf_render_args with freshly allocated outputs + the tensors its pointers borrow."""
    coord = _dev_f32(image_coord, "image_coord")
    B, n = coord.shape[0], coord.shape[-1]
    coord = coord.reshape(B, 3, n)
    Ki = _dev_f32(inv_intrinsics, "inv_intrinsics")
    if Ki.dim() == 2:
        Ki = Ki[None].expand(B, -1, -1).contiguous()
    P = parts.shape[1]
    tri = _dev_f32(tri_nchw, "tri_plane")
    H, W = tri.shape[2], tri.shape[3]
    mstride, fstride = _plane_strides(tri, feat_cl, B)
    dev = coord.device
    o = RenderOutputs()
    o.color = torch.empty(B, 3, n, dtype=torch.float32, device=dev)
    o.mask = torch.empty(B, n, dtype=torch.float32, device=dev)
    o.disparity = torch.empty(B, n, dtype=torch.float32, device=dev)
    o.fine_weights = torch.empty(B, 1, n, Nf - 1, dtype=torch.float32, device=dev) if want_fine else None
    o.fine_depth = torch.empty(B, 1, n, Nf, dtype=torch.float32, device=dev) if want_fine else None
    o.taps, o.counters = None, None
    a = _lib.RenderArgs()
    a.B, a.n, a.P, a.Nc, a.Nf, a.H, a.W = B, n, P, Nc, Nf, H, W
    a.mlp_mode, a.multiply_density_with_weight = MLP_MODE[mlp_mode], int(multiply_density_with_weight)
    a.drop_invalid_rays = int(B == 1 if drop_invalid_rays is None else drop_invalid_rays)
    a.render_scale, a.early_stop_eps = float(render_scale), float(early_stop_eps)
    a.clamp_mask, a.uniform_part_weight = int(bool(clamp_mask)), int(bool(uniform_part_weight))
    a.march = MARCH[march]
    a.image_coord, a.inv_intrinsics, a.parts = _p(coord), _p(Ki), _p(parts)
    cpose = _dev_f32(canonical_pose, "canonical_pose")
    a.canonical_pose = _p(cpose)
    a.feat_cl, a.feat_batch_stride = _p(feat_cl), fstride
    a.mask_planes, a.mask_batch_stride = tri.data_ptr() + PLANE_CH * H * W * 4, mstride
    a.mlp_pack = _p(mlp_pack)
    if bins is not None:
        bins = _dev_f32(bins, "bins").reshape(B, n, Nf)
    a.bins, a.seed = _p(bins), int(seed) & 0xFFFFFFFFFFFFFFFF
    a.color, a.mask, a.disparity = _p(o.color), _p(o.mask), _p(o.disparity)
    a.fine_weights, a.fine_depth = _p(o.fine_weights), _p(o.fine_depth)
    if debug:
        t = {
            "depth_min": torch.zeros(B, n, device=dev), "depth_max": torch.zeros(B, n, device=dev),
            "ray_validity": torch.zeros(B, n, dtype=torch.uint8, device=dev),
            "coarse_density": torch.zeros(B, n, Nc, device=dev), "fine_density": torch.zeros(B, n, Nf, device=dev),
            "fine_color": torch.zeros(B, 3, n, Nf, device=dev),
            "fine_valid": torch.zeros(B, n, Nf, dtype=torch.int32, device=dev),
            "bins": torch.zeros(B, n, Nf, device=dev),
        }
        a.dbg_depth_min, a.dbg_depth_max, a.dbg_ray_valid = _p(t["depth_min"]), _p(t["depth_max"]), _p(t["ray_validity"])
        a.dbg_coarse_density, a.dbg_fine_density = _p(t["coarse_density"]), _p(t["fine_density"])
        a.dbg_fine_color, a.dbg_fine_valid, a.dbg_bins = _p(t["fine_color"]), _p(t["fine_valid"]), _p(t["bins"])
        o.taps = t
    if return_bins and not debug:      # the importance samples actually used (needed to replay / differentiate)
        o.taps = {"bins": torch.zeros(B, n, Nf, dtype=torch.float32, device=dev)}     # rays that are dropped keep zeros
        a.dbg_bins = _p(o.taps["bins"])
    if count:
        o.counters = torch.zeros(8, dtype=torch.int64, device=dev)
        a.counters = _p(o.counters)
    ws = _render_workspace(dev, B, n)
    a.workspace = _p(ws)
    return a, o, [coord, Ki, cpose, tri, bins, ws]


STEP_PRE, STEP_MARCH, STEP_ALL = 1, 2, 3


class RenderStep:
    """One forward step bound to its arguments: `RenderStep(...).run()` = enarf_render_step_fwd (re-layout + prepare +
    ray set-up in ONE launch, then the march). `run(STEP_PRE)` / `run(STEP_MARCH)` issue the two launches separately
    (bench.py brackets the march with events that way)."""

    def __init__(self, pose_to_camera, bone_length, canonical_bone_length, z_rend, mlp, parents, origin_location,
                 coordinate_scale, image_coord, inv_intrinsics, canonical_pose, tri_nchw, feat_cl, Nc, Nf,
                 parts_out=None, pack_out=None, relayout=True, render_scale=1.0, bins=None, seed=0, mlp_mode="f32",
                 multiply_density_with_weight=False, drop_invalid_rays=None, want_fine=True, debug=False, count=False,
                 early_stop_eps=0.0, return_bins=False, clamp_mask=False, uniform_part_weight=False, march="auto",
                 group_frames=0):
        self.lib = _lib.load()
        self.pa, k1, self.parts, self.pack = _prepare_args(pose_to_camera, bone_length, canonical_bone_length, z_rend,
                                                           mlp, parents, origin_location, coordinate_scale, parts_out,
                                                           pack_out)
        self.ra, self.out, k2 = _render_args(image_coord, inv_intrinsics, self.parts, canonical_pose, tri_nchw, feat_cl,
                                             self.pack, Nc, Nf, render_scale, bins, seed, mlp_mode,
                                             multiply_density_with_weight, drop_invalid_rays, want_fine, debug, count,
                                             early_stop_eps, return_bins, clamp_mask, uniform_part_weight, march)
        self.ra.group_frames = int(group_frames)
        self.tri = _dev_f32(tri_nchw, "tri_plane")
        self.feat_cl, self.relayout = feat_cl, relayout
        if feat_cl.shape[0] != self.tri.shape[0]:
            raise ValueError("feat_cl and tri_plane must have the same batch")
        self._keep = k1 + k2

    def run(self, phases: int = STEP_ALL) -> RenderOutputs:
        if self.tri.device.index != torch.cuda.current_device():
            with torch.cuda.device(self.tri.device):
                return self._run(phases)
        return self._run(phases)

    def _run(self, phases: int) -> RenderOutputs:
        t = self.tri

        def call():
            rc = self.lib.enarf_render_step_fwd(C.byref(self.pa), _p(t) if self.relayout else None, _p(self.feat_cl),
                                                t.shape[0], t.shape[1], C.byref(self.ra), phases, _stream(t.device))
            _lib.check(rc, "enarf_render_step_fwd")

        if phases & STEP_PRE:          # a new epoch starts with the pre-march phase; a march-only call stays in it
            with _Epoch(t.device, _shape_key(self.ra)) as k:
                self.ra.ws_epoch = k
                call()
        else:
            call()
        return self.out


def render_step_fwd(*args, **kw) -> RenderOutputs:
    """prepare + tri-plane re-layout + render in two launches; arguments as RenderStep. Returns RenderOutputs
    (plus .parts / .pack of the prepare stage as attributes of the RenderStep it ran: use RenderStep to keep them)."""
    return RenderStep(*args, **kw).run()


# ---------------------------------------------------------------------------------------- backward
@_on_tensor_device
def render_bwd(image_coord, inv_intrinsics, parts, canonical_pose, tri_nchw, feat_cl, mlp_pack, Nf, bins,
               g_color, g_mask, g_disparity=None, render_scale: float = 1.0, drop_invalid_rays: Optional[bool] = None,
               feat_grad_channel_last: bool = False, clamp_mask: bool = False, uniform_part_weight: bool = False,
               multiply_density_with_weight: bool = False, counters: Optional[torch.Tensor] = None, group_frames: int = 0):
    """Backward of render_fwd w.r.t. the tri-plane and the per-image demodulated MLP weights / biases.

    Returns (grad_tri (same batch as tri_nchw: 1 for a shared tri-plane), dW [3 x (B,out,in)], db [3 x (out,)]).
    feat_grad_channel_last: the feature-plane gradient stays channel-last (same shape as feat_cl) and is returned as a
    fourth value instead of being folded into grad_tri[:, :96] (for producers that emit channel-last planes).
    The weight gradients are formed by enarf_weight_grad from the kernel's compact per-tile rows (features in, dL/dz3 out:
    it re-runs the MLP forward and backward on them; no library GEMM, no host sync).
    counters: optional zeroed int64 tensor [8] on the device (enarf_render_bwd_args.counters: pairs, tiles, rays, 128-B
    feature-gradient lines added, part-probability adds, gather rounds)."""
    lib = _lib.load()
    coord = _dev_f32(image_coord, "image_coord")
    B, n = coord.shape[0], coord.shape[-1]
    coord = coord.reshape(B, 3, n)
    Ki = _dev_f32(inv_intrinsics, "inv_intrinsics")
    if Ki.dim() == 2:
        Ki = Ki[None].expand(B, -1, -1).contiguous()
    P = parts.shape[1]
    tri = _dev_f32(tri_nchw, "tri_plane")
    Ct, H, W = tri.shape[1:]
    mstride, fstride = _plane_strides(tri, feat_cl, B)
    dev = coord.device
    rows = int(lib.enarf_render_bwd_rows_per_image(n, Nf))
    bufs, blocks = _row_buffers(B, rows, dev)
    grad_tri, gfeat = _zeros_pair(tri, feat_cl)
    a = _lib.RenderBwdArgs()
    a.B, a.n, a.P, a.Nf, a.H, a.W = B, n, P, Nf, H, W
    a.drop_invalid_rays = int(B == 1 if drop_invalid_rays is None else drop_invalid_rays)
    a.render_scale = float(render_scale)
    a.clamp_mask, a.uniform_part_weight = int(bool(clamp_mask)), int(bool(uniform_part_weight))
    a.multiply_density_with_weight = int(bool(multiply_density_with_weight))
    a.image_coord, a.inv_intrinsics, a.parts = _p(coord), _p(Ki), _p(parts)
    a.canonical_pose = _p(_dev_f32(canonical_pose, "canonical_pose"))
    a.feat_cl, a.feat_batch_stride = _p(feat_cl), fstride
    a.mask_planes, a.mask_batch_stride = tri.data_ptr() + PLANE_CH * H * W * 4, mstride
    a.mlp_pack = _p(mlp_pack)
    bins = _dev_f32(bins, "bins").reshape(B, n, Nf)
    a.bins = _p(bins)
    gc = None if g_color is None else _dev_f32(g_color, "g_color").reshape(B, 3, n)
    gm = None if g_mask is None else _dev_f32(g_mask, "g_mask").reshape(B, n)
    gd = None if g_disparity is None else _dev_f32(g_disparity, "g_disparity").reshape(B, n)
    a.g_color, a.g_mask, a.g_disparity = _p(gc), _p(gm), _p(gd)
    a.grad_feat_cl, a.grad_feat_batch_stride = _p(gfeat), fstride
    a.grad_mask_planes, a.grad_mask_batch_stride = grad_tri.data_ptr() + PLANE_CH * H * W * 4, mstride
    a.rows_x, a.rows_dz3 = _p(bufs["x"]), _p(bufs["dz3"])
    a.rows_per_image, a.row_blocks = rows, _p(blocks)
    a.workspace = _p(_render_workspace(dev, B, n))
    a.group_frames = int(group_frames)
    if counters is not None:
        if counters.dtype != torch.int64 or counters.numel() < 8 or counters.device != dev:
            raise ValueError("counters: an int64 tensor of 8 elements on the inputs' device")
        a.counters = _p(counters)
    with _Epoch(dev, counted=False):      # the backward's set-up clears the headers itself and uses header 0 (of every group)
        _lib.check(lib.enarf_render_bwd(C.byref(a), _stream(dev)), "enarf_render_bwd")
    if not feat_grad_channel_last:
        _lib.check(lib.enarf_triplane_unpack_add(_p(gfeat), _p(grad_tri), grad_tri.shape[0], Ct, H, W, _stream(dev)),
                   "enarf_triplane_unpack_add")
    dW, db = _weight_grad(bufs, blocks, mlp_pack, B, rows, dev)
    if feat_grad_channel_last:
        return grad_tri, dW, db, gfeat
    return grad_tri, dW, db


def _zeros_pair(a: torch.Tensor, b: torch.Tensor):
    """zero-filled tensors shaped like a and b from ONE allocation and one fill launch (the gradient planes of a backward)"""
    na = (a.numel() + 63) // 64 * 64                      # keeps the second tensor 256-byte aligned
    buf = torch.zeros(na + b.numel(), dtype=torch.float32, device=a.device)
    return buf[:a.numel()].view(a.shape), buf[na:].view(b.shape)


def _sum_images(t: torch.Tensor) -> torch.Tensor:
    """per-image gradients of a shared parameter -> the parameter's gradient; one image: a view, not a launch (the 13 such
    sums of a C1 backward were 57 us of 4 us reduction kernels: profiles/r03_bwd_final_kernel_stats.csv)"""
    return t[0] if t.shape[0] == 1 else t.sum(dim=0)


def _row_buffers(B: int, rows: int, dev: torch.device):
    bufs = {k: torch.empty(B, rows, w, dtype=torch.float32, device=dev) for k, w in (("x", 32), ("dz3", 4))}
    return bufs, torch.zeros(B, dtype=torch.int32, device=dev)


def _weight_grad(bufs, blocks, mlp_pack, B: int, rows: int, dev: torch.device):
    """dW'_l = dZ_l^T H_{l-1} per image and db_l = column sums of dZ_l, from the compact rows (x, dz3) a backward kernel
    exported: enarf_weight_grad re-runs the MLP forward and backward on them."""
    lib = _lib.load()
    dW = [torch.empty(B, 64, 32, device=dev), torch.empty(B, 64, 64, device=dev), torch.empty(B, 4, 64, device=dev)]
    dbb = [torch.empty(B, 64, device=dev), torch.empty(B, 64, device=dev), torch.empty(B, 4, device=dev)]
    w = _lib.WeightGradArgs()
    w.B, w.rows_per_image, w.row_blocks = B, rows, _p(blocks)
    w.rows_x, w.rows_dz3, w.mlp_pack = _p(bufs["x"]), _p(bufs["dz3"]), _p(mlp_pack)
    w.dW1, w.dW2, w.dW3 = _p(dW[0]), _p(dW[1]), _p(dW[2])
    w.db1, w.db2, w.db3 = _p(dbb[0]), _p(dbb[1]), _p(dbb[2])
    wws = torch.empty(int(lib.enarf_weight_grad_workspace_bytes(B, rows)) // 4, dtype=torch.float32, device=dev)
    w.workspace = _p(wws)
    _lib.check(lib.enarf_weight_grad(C.byref(w), _stream(dev)), "enarf_weight_grad")
    return dW, [_sum_images(t) for t in dbb]


@_on_tensor_device
def query_bwd(points, parts, canonical_pose, tri_nchw, feat_cl, mlp_pack, g_density, g_color, clamp_mask: bool = False,
              uniform_part_weight: bool = False, multiply_density_with_weight: bool = False):
    """Backward of query_fwd w.r.t. the tri-plane and the per-image demodulated MLP weights / biases.

    points (B,3,N); g_density (B,1,N) or None; g_color (B,3,N) or None. Returns (grad_tri, dW [3 x (B,out,in)], db)."""
    lib = _lib.load()
    pts = _dev_f32(points, "points")
    B, _, N = pts.shape
    P = parts.shape[1]
    tri = _dev_f32(tri_nchw, "tri_plane")
    Ct, H, W = tri.shape[1:]
    mstride, fstride = _plane_strides(tri, feat_cl, B)
    dev = pts.device
    rows = max(int(lib.enarf_query_bwd_rows_per_image(N)), 16)
    bufs, blocks = _row_buffers(B, rows, dev)
    grad_tri, gfeat = _zeros_pair(tri, feat_cl)
    a = _lib.QueryBwdArgs()
    a.B, a.P, a.H, a.W, a.N = B, P, H, W, N
    a.clamp_mask, a.uniform_part_weight = int(bool(clamp_mask)), int(bool(uniform_part_weight))
    a.multiply_density_with_weight = int(bool(multiply_density_with_weight))
    a.points, a.parts = _p(pts) if N else _p(torch.zeros(1, device=dev)), _p(parts)
    a.canonical_pose = _p(_dev_f32(canonical_pose, "canonical_pose"))
    a.feat_cl, a.feat_batch_stride = _p(feat_cl), fstride
    a.mask_planes, a.mask_batch_stride = tri.data_ptr() + PLANE_CH * H * W * 4, mstride
    a.mlp_pack = _p(mlp_pack)
    gd = None if g_density is None else _dev_f32(g_density, "g_density").reshape(B, N)
    gc = None if g_color is None else _dev_f32(g_color, "g_color").reshape(B, 3, N)
    a.g_density, a.g_color = (_p(gd) if N else None), (_p(gc) if N else None)
    a.grad_feat_cl, a.grad_feat_batch_stride = _p(gfeat), fstride
    a.grad_mask_planes, a.grad_mask_batch_stride = grad_tri.data_ptr() + PLANE_CH * H * W * 4, mstride
    a.rows_x, a.rows_dz3 = _p(bufs["x"]), _p(bufs["dz3"])
    a.rows_per_image, a.row_blocks = rows, _p(blocks)
    _lib.check(lib.enarf_query_bwd(C.byref(a), _stream(dev)), "enarf_query_bwd")
    _lib.check(lib.enarf_triplane_unpack_add(_p(gfeat), _p(grad_tri), grad_tri.shape[0], Ct, H, W, _stream(dev)),
               "enarf_triplane_unpack_add")
    dW, db = _weight_grad(bufs, blocks, mlp_pack, B, rows, dev)
    return grad_tri, dW, db


@_on_tensor_device
def prepare_bwd(z_rend: torch.Tensor, mlp: Dict[str, torch.Tensor], dW):
    """dW' (3 x (B,out,in)) -> gradients of conv.weight, modulation.weight, modulation.bias (summed over the batch,
    in the parameters' own shapes) and of z_rend (B, style_dim)."""
    lib = _lib.load()
    z = _dev_f32(z_rend, "z_rend")
    B, D = z.shape
    dev = z.device
    a = _lib.PrepareBwdArgs()
    a.B, a.style_dim, a.z_rend = B, D, _p(z)
    keep, outs = [], []
    dims = [(FEAT_DIM, 64), (64, 64), (64, 4)]
    for i, (cin, cout) in enumerate(dims):
        cw = _dev_f32(mlp[f"layers.{i}.conv.weight"].detach(), "conv.weight")
        mw = _dev_f32(mlp[f"layers.{i}.conv.modulation.weight"].detach(), "modulation.weight")
        mb = _dev_f32(mlp[f"layers.{i}.conv.modulation.bias"].detach(), "modulation.bias")
        d = _dev_f32(dW[i], "dW")
        o = (torch.empty(B, cout, cin, device=dev), torch.empty(B, cin, D, device=dev), torch.empty(B, cin, device=dev))
        keep += [cw, mw, mb, d]
        outs.append(o)
        a.conv_weight[i], a.mod_weight[i], a.mod_bias[i], a.dW[i] = _p(cw), _p(mw), _p(mb), _p(d)
        a.d_conv_weight[i], a.d_mod_weight[i], a.d_mod_bias[i] = _p(o[0]), _p(o[1]), _p(o[2])
    dz = torch.empty(B, 3, D, device=dev)
    a.d_z_rend = _p(dz)
    _lib.check(lib.enarf_prepare_bwd(C.byref(a), _stream(dev)), "enarf_prepare_bwd")
    grads = {}
    for i, (cin, cout) in enumerate(dims):
        grads[f"layers.{i}.conv.weight"] = _sum_images(outs[i][0]).reshape(1, cout, cin, 1)
        grads[f"layers.{i}.conv.modulation.weight"] = _sum_images(outs[i][1])
        grads[f"layers.{i}.conv.modulation.bias"] = _sum_images(outs[i][2])
    return grads, dz.sum(1)
