"""ctypes binding of libenarf_hip.so (the C ABI declared in include/enarf_hip.h).

There is no CPU fallback: if the library is missing or a call fails, an exception is raised.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
# The product loads the in-tree library and nothing else: no environment variable changes what runs.
# Measurement tools that A/B a variant build of the same ABI call `use_variant(path)` explicitly, before the first load().
LIB_PATH = os.path.join(_HERE, "csrc", "libenarf_hip.so")
_variant = False


def use_variant(path: str) -> None:
    """Measurement only (bench.py --allow-variant, tools/): load another build of the same ABI instead of the in-tree one."""
    global LIB_PATH, _variant
    if _lib is not None:
        raise EnarfHipError("use_variant() must be called before the library is loaded")
    LIB_PATH, _variant = os.path.abspath(path), True


def library_info() -> dict:
    """What is (or will be) loaded: path and whether it is a variant build - bench.py prints this."""
    return {"path": LIB_PATH, "variant": _variant}

ABI_VERSION = 4
STATUS_MARCH_WATCHDOG = 1                      # ENARF_STATUS_* (include/enarf_hip.h)
MAX_JOINTS = 32
MAX_PARTS = 32
FEAT_DIM = 32
HIDDEN = 64

INTERP = {"bilinear": 0, "nearest": 1}            # cuda_extension/triplane_sampler.py:7-10
PADDING = {"zeros": 0, "border": 1, "reflection": 2}   # :12-16
ORIGIN = {"center": 0, "center_fixed": 1, "center+head": 2}
MLP_MODE = {"f32": 0, "bf16x3": 1, "bf16": 2, "f16x3": 3}
MARCH = {"auto": 0, "ray": 1, "task": 2}          # ENARF_MARCH_* (include/enarf_hip.h)

_f32p = C.c_void_p   # device pointers travel as integers


class PrepareArgs(C.Structure):
    _fields_ = [
        ("B", C.c_int), ("num_joints", C.c_int), ("origin_location", C.c_int), ("style_dim", C.c_int),
        ("coordinate_scale", C.c_float), ("parents", C.c_int * MAX_JOINTS),
        ("pose_to_camera", _f32p), ("bone_length", _f32p), ("canonical_bone_length", _f32p), ("z_rend", _f32p),
        ("conv_weight", _f32p * 3), ("mod_weight", _f32p * 3), ("mod_bias", _f32p * 3), ("bias", _f32p * 3),
        ("parts", _f32p), ("mlp_pack", _f32p),
    ]


class QueryArgs(C.Structure):
    _fields_ = [
        ("B", C.c_int), ("N", C.c_longlong), ("P", C.c_int), ("H", C.c_int), ("W", C.c_int),
        ("mlp_mode", C.c_int), ("multiply_density_with_weight", C.c_int),
        ("points", _f32p), ("parts", _f32p), ("canonical_pose", _f32p),
        ("feat_cl", _f32p), ("feat_batch_stride", C.c_longlong),
        ("mask_planes", _f32p), ("mask_batch_stride", C.c_longlong),
        ("mlp_pack", _f32p), ("density", _f32p), ("color", _f32p), ("valid_bits", _f32p),
        ("dbg_canonical", _f32p), ("dbg_weight", _f32p),
        ("grid_D", C.c_int), ("grid_center", C.c_float * 3), ("grid_scale", C.c_float),
        ("clamp_mask", C.c_int), ("uniform_part_weight", C.c_int),
    ]


class RenderArgs(C.Structure):
    _fields_ = [
        ("B", C.c_int), ("n", C.c_int), ("P", C.c_int), ("Nc", C.c_int), ("Nf", C.c_int),
        ("H", C.c_int), ("W", C.c_int), ("mlp_mode", C.c_int), ("multiply_density_with_weight", C.c_int),
        ("drop_invalid_rays", C.c_int), ("render_scale", C.c_float), ("early_stop_eps", C.c_float),
        ("image_coord", _f32p), ("inv_intrinsics", _f32p), ("parts", _f32p), ("canonical_pose", _f32p),
        ("feat_cl", _f32p), ("feat_batch_stride", C.c_longlong),
        ("mask_planes", _f32p), ("mask_batch_stride", C.c_longlong),
        ("mlp_pack", _f32p), ("bins", _f32p), ("seed", C.c_uint64),
        ("color", _f32p), ("mask", _f32p), ("disparity", _f32p), ("fine_weights", _f32p), ("fine_depth", _f32p),
        ("dbg_depth_min", _f32p), ("dbg_depth_max", _f32p), ("dbg_ray_valid", _f32p),
        ("dbg_coarse_density", _f32p), ("dbg_fine_density", _f32p), ("dbg_fine_color", _f32p),
        ("dbg_fine_valid", _f32p), ("dbg_bins", _f32p), ("counters", _f32p), ("workspace", _f32p),
        ("clamp_mask", C.c_int), ("uniform_part_weight", C.c_int), ("march", C.c_int), ("ws_epoch", C.c_int),
        ("near_far", _f32p), ("ray_id_base", C.c_ulonglong), ("group_frames", C.c_int),
    ]


class RenderBwdArgs(C.Structure):
    _fields_ = [
        ("B", C.c_int), ("n", C.c_int), ("P", C.c_int), ("Nf", C.c_int), ("H", C.c_int), ("W", C.c_int),
        ("drop_invalid_rays", C.c_int), ("render_scale", C.c_float),
        ("image_coord", _f32p), ("inv_intrinsics", _f32p), ("parts", _f32p), ("canonical_pose", _f32p),
        ("feat_cl", _f32p), ("feat_batch_stride", C.c_longlong),
        ("mask_planes", _f32p), ("mask_batch_stride", C.c_longlong),
        ("mlp_pack", _f32p), ("bins", _f32p),
        ("g_color", _f32p), ("g_mask", _f32p), ("g_disparity", _f32p),
        ("grad_feat_cl", _f32p), ("grad_feat_batch_stride", C.c_longlong),
        ("grad_mask_planes", _f32p), ("grad_mask_batch_stride", C.c_longlong),
        ("rows_x", _f32p), ("rows_dz3", _f32p), ("rows_per_image", C.c_longlong), ("row_blocks", _f32p),
        ("workspace", _f32p), ("near_far", _f32p), ("group_frames", C.c_int), ("counters", _f32p),
        ("clamp_mask", C.c_int), ("uniform_part_weight", C.c_int), ("multiply_density_with_weight", C.c_int),
    ]


class QueryBwdArgs(C.Structure):
    _fields_ = [
        ("B", C.c_int), ("P", C.c_int), ("H", C.c_int), ("W", C.c_int), ("N", C.c_longlong),
        ("points", _f32p), ("parts", _f32p), ("canonical_pose", _f32p),
        ("feat_cl", _f32p), ("feat_batch_stride", C.c_longlong),
        ("mask_planes", _f32p), ("mask_batch_stride", C.c_longlong),
        ("mlp_pack", _f32p), ("g_density", _f32p), ("g_color", _f32p),
        ("grad_feat_cl", _f32p), ("grad_feat_batch_stride", C.c_longlong),
        ("grad_mask_planes", _f32p), ("grad_mask_batch_stride", C.c_longlong),
        ("rows_x", _f32p), ("rows_dz3", _f32p), ("rows_per_image", C.c_longlong), ("row_blocks", _f32p),
        ("clamp_mask", C.c_int), ("uniform_part_weight", C.c_int), ("multiply_density_with_weight", C.c_int),
    ]


class WeightGradArgs(C.Structure):
    _fields_ = [
        ("B", C.c_int),
        ("rows_x", _f32p), ("rows_dz3", _f32p), ("mlp_pack", _f32p), ("rows_per_image", C.c_longlong),
        ("row_blocks", _f32p),
        ("dW1", _f32p), ("dW2", _f32p), ("dW3", _f32p), ("db1", _f32p), ("db2", _f32p), ("db3", _f32p),
        ("workspace", _f32p),
    ]


class PrepareBwdArgs(C.Structure):
    _fields_ = [
        ("B", C.c_int), ("style_dim", C.c_int), ("z_rend", _f32p),
        ("conv_weight", _f32p * 3), ("mod_weight", _f32p * 3), ("mod_bias", _f32p * 3), ("dW", _f32p * 3),
        ("d_conv_weight", _f32p * 3), ("d_mod_weight", _f32p * 3), ("d_mod_bias", _f32p * 3), ("d_z_rend", _f32p),
    ]


# every symbol include/enarf_hip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "enarf_abi_version": (C.c_int, []),
    "enarf_version": (C.c_int, []),
    "enarf_query_bwd_rows_per_image": (C.c_longlong, [C.c_longlong]),
    "enarf_query_bwd": (C.c_int, [C.POINTER(QueryBwdArgs), C.c_void_p]),
    "enarf_last_error": (C.c_char_p, []),
    "enarf_device_status": (C.c_int, [C.POINTER(C.c_uint), C.c_int]),
    "enarf_triplane_sample_workspace_bytes": (C.c_size_t, [C.c_int] * 4),
    "enarf_triplane_sample_bwd_workspace_bytes": (C.c_size_t, [C.c_int] * 4),
    "enarf_triplane_sample_fwd": (C.c_int, [_f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_longlong,
                                            C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "enarf_triplane_sample_bwd": (C.c_int, [_f32p, _f32p, _f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int,
                                            C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "enarf_triplane_sample_ex_fwd": (C.c_int, [_f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_longlong,
                                               C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "enarf_triplane_sample_ex_bwd": (C.c_int, [_f32p, _f32p, _f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int,
                                               C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "enarf_triplane_pack": (C.c_int, [_f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "enarf_mlp_pack_bytes": (C.c_size_t, []),
    "enarf_prepare": (C.c_int, [C.POINTER(PrepareArgs), C.c_void_p]),
    "enarf_mlp_unpack": (C.c_int, [C.c_void_p, _f32p, C.c_void_p]),
    "enarf_query_fwd": (C.c_int, [C.POINTER(QueryArgs), C.c_void_p]),
    "enarf_render_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "enarf_near_far": (C.c_int, [_f32p, C.c_int, C.c_int, _f32p, C.c_void_p]),
    "enarf_render_fwd": (C.c_int, [C.POINTER(RenderArgs), C.c_void_p]),
    "enarf_render_step_fwd": (C.c_int, [C.POINTER(PrepareArgs), _f32p, _f32p, C.c_int, C.c_int, C.POINTER(RenderArgs),
                                        C.c_int, C.c_void_p]),
    "enarf_triplane_warp_fwd": (C.c_int, [_f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "enarf_triplane_warp_bwd": (C.c_int, [_f32p, _f32p, _f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "enarf_weight_grad_workspace_bytes": (C.c_size_t, [C.c_int, C.c_longlong]),
    "enarf_weight_grad": (C.c_int, [C.POINTER(WeightGradArgs), C.c_void_p]),
    "enarf_render_bwd_rows_per_image": (C.c_longlong, [C.c_int, C.c_int]),
    "enarf_render_bwd": (C.c_int, [C.POINTER(RenderBwdArgs), C.c_void_p]),
    "enarf_prepare_bwd": (C.c_int, [C.POINTER(PrepareBwdArgs), C.c_void_p]),
    "enarf_triplane_unpack_add": (C.c_int, [_f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "enarf_mask_topk_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "enarf_mask_dilate_topk": (C.c_int, [_f32p, _f32p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                         C.c_void_p]),
    "enarf_bias_act": (C.c_int, [_f32p, _f32p, _f32p, _f32p, C.c_longlong, C.c_int, C.c_longlong, C.c_float, C.c_float, C.c_void_p]),
    "enarf_upfirdn2d_out_size": (C.c_int, [C.c_int] * 6),
    "enarf_upfirdn2d": (C.c_int, [_f32p, _f32p, C.c_longlong, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
}

_lib: Optional[C.CDLL] = None


class EnarfHipError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load libenarf_hip.so (once). Raises if it has not been built: there is no fallback path."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own HIP runtime (torch/lib/libamdhip64.so). It must be in the process BEFORE this
    # library is dlopen'ed so that both resolve to ONE runtime; loaded the other way round, this library binds
    # /opt/rocm's copy and its launches fail with "no ROCm-capable device is detected" next to torch's.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise EnarfHipError(
            f"{LIB_PATH} is missing: build it with `python -m enarf_gan_amd.build` (hipcc, gfx950). "
            "enarf_gan_amd has no CPU or eager fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.enarf_abi_version() != ABI_VERSION:
        raise EnarfHipError(f"libenarf_hip.so ABI {lib.enarf_abi_version()} != {ABI_VERSION}")
    _lib = lib
    return lib


def device_status(clear: bool = True) -> int:
    """ENARF_STATUS_* flags kernels of the CURRENT device have raised since the last clear (enarf_device_status): a word
    of pinned host memory, read without a synchronisation - it covers every launch that has completed."""
    flags = C.c_uint(0)
    lib = load()
    rc = lib.enarf_device_status(C.byref(flags), int(clear))
    if rc != 0:
        raise EnarfHipError(f"enarf_device_status failed (code {rc}): {lib.enarf_last_error().decode(errors='replace')}")
    return int(flags.value)


def raise_on_device_status(what: str) -> None:
    flags = device_status(clear=True)
    if flags & STATUS_MARCH_WATCHDOG:
        raise EnarfHipError(f"{what}: an earlier enarf_render_fwd launch on this device was abandoned by its scheduler "
                            "watchdog (ENARF_STATUS_MARCH_WATCHDOG): the outputs of that launch are incomplete")
    if flags:
        raise EnarfHipError(f"{what}: device status {flags:#x}")


def check(rc: int, what: str) -> None:
    """Every call through the C ABI ends here: its own return code first, then the device's sticky status word - a kernel
    that gave up (the task march's watchdog) is reported by the next call on that device, whatever the caller asked for."""
    if rc != 0:
        msg = load().enarf_last_error().decode(errors="replace")
        if rc == -2:
            raise NotImplementedError(f"{what}: {msg}")
        raise EnarfHipError(f"{what} failed (code {rc}): {msg}")
    raise_on_device_status(what)
