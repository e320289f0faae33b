"""On-disk formats of the reference that sit either side of the render path: training snapshots and the
pose / camera caches. Readers execute nothing from the file.

* snapshots - `torch.save({"iteration", "start_time", "gen", "dis", "gen_opt", "dis_opt"})` (train_ENARF_GAN.py:278-294;
  train_DSO.py:287-298); demos load `["gen"]` with `strict=False` (DSO_demo.py:37-42, ENARF_GAN_demo.py). The mirror
  generators keep the reference's state-dict keys (`nerf.tri_plane`, `nerf.mlp.layers.{0,1,2}.*`, the canonical-pose
  buffers; `background_generator.*` with libraries/custom_stylegan2/net.py's; `nerf.tri_plane_gen.*` with
  libraries/stylegan2_ada/networks.py's), so a reference snapshot's weights load by name; a key with no counterpart or another
  shape is reported, not loaded. `["dis"]` loads into libraries.custom_stylegan2.net.Discriminator with a plain
  `load_state_dict`. Read with `torch.load(weights_only=True)`.
* `cache.pickle` (dataset/dataset.py:152-185; README.md:40-48) - {"img": [blosc-packed uint8 (3, S, S)], "camera_intrinsic"
  (N, 3, 3), "smpl_pose" (N, 24, 4, 4) [, "camera_rotation" (N, 3, 3), "camera_translation" (N, 3, 1), "frame_id" (N,)]}.
* `sample_data.pickle` (data_preprocess/ZJU/prepare_sample_data.py:59-66) - [{"pose_3d" (24, 4, 4), "intrinsics" (3, 3),
  "bone_length" (23, 1)}, ...].
  Both are read with an unpickler that only rebuilds numpy arrays and plain containers: any other global in the stream
  (the way a pickle runs code) raises `UnsafePickleError`.
"""
from __future__ import annotations

import io
import pickle
import time
from typing import Any, Dict, List, NamedTuple, Optional

import numpy as np
import torch


# ------------------------------------------------------------------------------------------------------------ snapshots
class SnapshotReport(NamedTuple):
    loaded: List[str]          # keys copied into the generator
    missing: List[str]         # generator keys the snapshot does not hold (left as they were)
    ignored: List[str]         # snapshot keys of networks this repo does not build, or with another shape
    iteration: Optional[int]


def _strip_module(sd: Dict[str, Any]) -> Dict[str, Any]:
    """DistributedDataParallel prefixes every key with "module." (train_ENARF_GAN.py:203-206 saves gen.module's, but
    snapshots written from the wrapped model exist too)."""
    return {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}


def read_snapshot(path) -> Dict[str, Any]:
    """The snapshot dictionary, tensors only (`torch.load(weights_only=True)`: optimiser states and plain numbers pass,
    anything that would need code from the file is refused by torch)."""
    snap = torch.load(path, map_location="cpu", weights_only=True)
    if not isinstance(snap, dict):
        raise ValueError(f"{path}: not a snapshot dictionary")
    return snap


def load_generator_snapshot(path_or_snapshot, generator: torch.nn.Module, strict: bool = False) -> SnapshotReport:
    """Load a reference snapshot's `["gen"]` weights (or a bare state dict) into a mirror generator
    (models.generator.TriNARFGenerator / DSONARFGenerator), as DSO_demo.py:37-42 does with strict=False.
    strict=True raises if a generator key is missing or has another shape (train_DSO.py:222-230 back-fills, then strict)."""
    snap = path_or_snapshot if isinstance(path_or_snapshot, dict) else read_snapshot(path_or_snapshot)
    sd = snap["gen"] if "gen" in snap and isinstance(snap["gen"], dict) else snap
    sd = _strip_module({k: v for k, v in sd.items() if torch.is_tensor(v)})
    own = generator.state_dict()
    take, ignored = {}, []
    for k, v in sd.items():
        if k in own and tuple(own[k].shape) == tuple(v.shape):
            take[k] = v
        else:
            ignored.append(k)
    missing = [k for k in own if k not in take]
    if strict and missing:
        raise RuntimeError(f"snapshot lacks (or has another shape for) generator keys: {missing[:8]}{' ...' if len(missing) > 8 else ''}")
    generator.load_state_dict(take, strict=False)
    it = snap.get("iteration") if isinstance(snap, dict) else None
    return SnapshotReport(sorted(take), missing, sorted(ignored), int(it) if it is not None else None)


def save_snapshot(path, generator: torch.nn.Module, iteration: int, discriminator: Optional[torch.nn.Module] = None,
                  gen_optimizer=None, dis_optimizer=None, start_time: Optional[float] = None) -> None:
    """Write the reference's snapshot schema (train_ENARF_GAN.py:278-294), so that its demos and its resume path read a
    snapshot of the mirror generator."""
    torch.save({"iteration": int(iteration), "start_time": float(time.time() if start_time is None else start_time),
                "gen": generator.state_dict(), "dis": discriminator.state_dict() if discriminator is not None else {},
                "gen_opt": gen_optimizer.state_dict() if gen_optimizer is not None else {},
                "dis_opt": dis_optimizer.state_dict() if dis_optimizer is not None else {}}, path)


# ------------------------------------------------------------------------------------------------------- pickled caches
class UnsafePickleError(pickle.UnpicklingError):
    pass


def _numpy_globals():
    try:                                      # numpy >= 2
        import numpy._core.multiarray as _ma
        import numpy._core.numeric as _num
    except ImportError:                       # numpy 1.x
        import numpy.core.multiarray as _ma
        import numpy.core.numeric as _num
    table = {("numpy", "ndarray"): np.ndarray, ("numpy", "dtype"): np.dtype}
    for root in ("numpy.core", "numpy._core"):     # files written by numpy 1.x name numpy.core, by numpy 2 numpy._core
        table[(root + ".multiarray", "_reconstruct")] = _ma._reconstruct
        table[(root + ".multiarray", "scalar")] = _ma.scalar
        if hasattr(_num, "_frombuffer"):
            table[(root + ".numeric", "_frombuffer")] = _num._frombuffer
    return table


class _ArrayUnpickler(pickle.Unpickler):
    """Rebuilds numpy arrays / scalars / dtypes and the built-in containers; refuses every other global."""
    _allowed = None

    def find_class(self, module, name):
        if _ArrayUnpickler._allowed is None:
            _ArrayUnpickler._allowed = _numpy_globals()
        obj = _ArrayUnpickler._allowed.get((module, name))
        if obj is None:
            raise UnsafePickleError(f"refused global {module}.{name}: the cache readers only rebuild numpy arrays and plain containers")
        return obj


def _load_arrays(path):
    with open(path, "rb") as f:
        return _ArrayUnpickler(io.BytesIO(f.read())).load()


class HumanCache(NamedTuple):
    """cache.pickle as HumanDataset.load_cache leaves it (dataset/dataset.py:152-185)."""
    img: List[bytes]                       # blosc-packed uint8 (3, S, S) per frame (unpack_image)
    intrinsics: np.ndarray                 # (N, 3, 3)
    inv_intrinsics: np.ndarray             # (N, 3, 3) np.linalg.inv, as the reference (:163)
    pose_to_world: np.ndarray              # (N, 24, 4, 4) "smpl_pose"
    pose_to_camera: np.ndarray             # (N, 24, 4, 4) extrinsic @ pose_to_world (:172), or pose_to_world
    camera_rotation: Optional[np.ndarray]  # (N, 3, 3) or None
    frame_id: Optional[np.ndarray]


def read_cache(path) -> HumanCache:
    d = _load_arrays(path)
    if not isinstance(d, dict) or "img" not in d:
        raise ValueError(f"{path}: not a cache.pickle dictionary (no 'img')")
    for key in ("camera_intrinsic", "smpl_pose"):
        if key not in d:
            raise ValueError(f"{path}: cache.pickle lacks '{key}'")
    K = np.asarray(d["camera_intrinsic"])
    pose_w = np.asarray(d["smpl_pose"])
    n = len(d["img"])
    if K.shape != (n, 3, 3) or pose_w.shape[0] != n or pose_w.shape[-2:] != (4, 4):
        raise ValueError(f"{path}: {n} images but camera_intrinsic {K.shape}, smpl_pose {pose_w.shape}")
    rot = None
    if "camera_rotation" in d:
        rot = np.asarray(d["camera_rotation"])
        ext = np.broadcast_to(np.eye(4), (n, 4, 4)).copy()
        ext[:, :3, :3] = rot
        ext[:, :3, 3:] = np.asarray(d["camera_translation"])
        pose_c = np.matmul(ext[:, None], pose_w)
    else:
        pose_c = pose_w
    return HumanCache(list(d["img"]), K, np.linalg.inv(K), pose_w, pose_c, rot,
                      np.asarray(d["frame_id"]) if "frame_id" in d else None)


def unpack_image(packed: bytes) -> np.ndarray:
    """One entry of cache.pickle["img"] -> uint8 (3, S, S) (dataset.py:196: blosc.unpack_array). blosc is a third-party
    package the reference requires (requirements.txt); without it this raises ImportError, as the reference would."""
    import blosc
    return blosc.unpack_array(packed)


class SampleData(NamedTuple):
    """sample_data.pickle (prepare_sample_data.py:59-66), stacked."""
    pose_3d: np.ndarray          # (N, 24, 4, 4) pose_to_camera
    intrinsics: np.ndarray       # (N, 3, 3)
    bone_length: np.ndarray      # (N, 23, 1)


def read_sample_data(path) -> SampleData:
    rows = _load_arrays(path)
    if not isinstance(rows, list) or not rows or not all(isinstance(r, dict) for r in rows):
        raise ValueError(f"{path}: not a sample_data.pickle list of dictionaries")
    for key in ("pose_3d", "intrinsics", "bone_length"):
        if any(key not in r for r in rows):
            raise ValueError(f"{path}: an entry lacks '{key}'")
    return SampleData(np.stack([np.asarray(r["pose_3d"]) for r in rows]), np.stack([np.asarray(r["intrinsics"]) for r in rows]),
                      np.stack([np.asarray(r["bone_length"]) for r in rows]))
