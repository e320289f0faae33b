#!/usr/bin/env python3
"""Throughput of the create_mesh density sweep (667^3 = 297 M points at the reference's voxel_size 0.003)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from _helpers import Scene
from test_gpu_api import _model
from enarf_gan_amd.libraries.NARF.mesh_rendering import density_volume

sc = Scene(128, 1, "center_fixed", 20)
m = _model(sc)
s = sc.raw
mi = {"z": None, "z_rend": s["z_rend"].cuda(), "bone_length": sc.bl_parts.cuda(), "truncation_psi": 1}
center = sc.pose_parts[0, :, :3, 3].mean(0).reshape(1, 3, 1)
pose = sc.pose_parts.cuda()
for voxel in (0.01, 0.003):
    density_volume(m, pose, center, 0.05, mi)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    vol = density_volume(m, pose, center, voxel, mi)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"voxel {voxel}: {vol.shape[0]}^3 = {vol.numel() / 1e6:.1f} M points in {dt * 1e3:.1f} ms = {vol.numel() / dt / 1e9:.2f} G points/s, "
          f"occupied (> 15): {float((vol > 15).float().mean()) * 100:.2f} %", flush=True)
