#!/usr/bin/env python3
"""Throughput of the create_mesh density sweep (667^3 = 297 M points at the reference's voxel_size 0.003), through the
model mirror: TriPlaneNARF.density_volume -> one lattice-mode launch of enarf_query_fwd. OFFSET=x moves the lattice x
units away from the body (an all-empty sweep: the fixed cost per tile)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from enarf_gan_amd import synth
from enarf_gan_amd.models.narf import TriPlaneNARF

sc = synth.make_scene(128, 1, "center_fixed", 20)
m = TriPlaneNARF(synth.nerf_config(origin_location="center_fixed"), 20, 24, parent=sc["parents"], num_bone_param=23)
m.register_canonical_pose(sc["canonical_pose"])
m.load_state_dict({f"mlp.{k}": v for k, v in sc["mlp"].items()}, strict=False)
with torch.no_grad():
    m.tri_plane.copy_(sc["tri_plane"][:1])
m = m.cuda().eval()
pose, bl, z = sc["pose_to_camera"].cuda(), sc["bone_length"].cuda(), sc["z_rend"].cuda()
from enarf_gan_amd.libraries.NARF.mesh_rendering import density_volume
center, pose_parts, mi = m._mesh_inputs(pose, None, z, bl, 0.4)
center = center.clone()
center[:, 0] += float(os.environ.get("OFFSET", "0"))                # the lattice moved sideways, away from the body
for voxel in (0.01, 0.003):
    density_volume(m, pose_parts, center, 0.05, mi)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    vol = density_volume(m, pose_parts, center, voxel, mi)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"voxel {voxel}: {vol.shape[0]}^3 = {vol.numel() / 1e6:.1f} M points in {dt * 1e3:.1f} ms = {vol.numel() / dt / 1e9:.2f} G points/s, "
          f"occupied (> 15): {float((vol > 15).float().mean()) * 100:.2f} %", flush=True)
