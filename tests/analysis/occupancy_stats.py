"""CPU analysis (not product, not a test): valid-part counts per coarse sample of the bench scene, to size the
lane utilisation / wave balance of the gather rounds under different sample->wave assignments."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from enarf_gan_amd import synth
from oracle import enarf_oracle as O

S, Nc = 128, 48
sc = synth.make_scene(S, 1, "center_fixed", 20, pose_seed=1234, shared_triplane=True)
pose_p, bl_p = O.transform_pose(sc["pose_to_camera"], sc["bone_length"], "center_fixed", sc["parents"])
cpose, cbl = O.register_canonical_pose(sc["canonical_pose"], sc["parents"], "center_fixed")
pose = O.scale_pose_translation(pose_p, 3.0)
scale = O.canonical_scale(cbl, bl_p, 3.0)
near, far = O.near_far(pose)
rd = O.ray_directions(sc["image_coord"], sc["inv_intrinsics"])
dmin, dmax, rvalid = O.frustum_range(rd, pose, near, far)
live = rvalid[0]
rd, dmin, dmax = rd[:, :, live], dmin[:, live], dmax[:, live]
_, pts, _, _ = O.coarse_points(rd, dmin, dmax, Nc)
m = pts.shape[2]
loc, can = O.to_local_and_canonical(pts.reshape(1, 3, -1), pose, scale, cpose)
val = O.validity(loc, can).reshape(-1, m, Nc)          # (P, m, Nc)
cnt = val.sum(0).numpy()                                # (m, Nc) valid parts per sample
print("live rays", m, "pairs/ray", cnt.sum() / m, "valid samples frac", (cnt > 0).mean(), "mean parts per valid sample",
      cnt.sum() / (cnt > 0).sum(), "max", cnt.max())
# current: wave w owns samples [12w, 12w+12)
seg = cnt.reshape(m, 4, 12).max(2)                      # rounds per wave
print("contiguous 4x12: rounds/wave mean", seg.mean(), "critical wave mean", seg.max(1).mean(), "sum", seg.sum(1).mean(),
      "lane util", cnt.sum() / (16 * seg.sum()))
# 3 waves x 16
seg3 = cnt.reshape(m, 3, 16).max(2)
print("contiguous 3x16: critical", seg3.max(1).mean(), "sum", seg3.sum(1).mean(), "lane util", cnt.sum() / (16 * seg3.sum()))
# interleaved: sample i -> wave i % 4
il = np.stack([cnt[:, w::4].max(1) for w in range(4)], 1)
print("interleaved: critical", il.max(1).mean(), "sum", il.sum(1).mean())
# sorted by count (ideal sample-level balance): sort samples by count desc, deal out in tiles of 16 -> rounds = first of each tile
srt = -np.sort(-cnt, axis=1)
tiles = srt.reshape(m, 3, 16)[:, :, 0]
print("count-sorted tiles of 16: rounds per tile", tiles.mean(0), "sum", tiles.sum(1).mean())
# compacted valid samples only, split evenly over 4 waves
nv = (cnt > 0).sum(1)
print("valid samples per ray: mean", nv.mean(), "hist", np.bincount(nv, minlength=49)[::4])
# pair-level bound
print("pair-level bound rounds (pairs/64 per WG round):", np.ceil(cnt.sum(1) / 64).mean())
# ---- per-ray cost distribution (tail analysis): coarse pairs per ray and parts touched
pairs = cnt.sum(1)
parts_touched = val.any(2).sum(0).numpy()      # (m,) parts with >= 1 valid coarse sample
print("pairs/ray percentiles 10/50/90/99/max", np.percentile(pairs, [10, 50, 90, 99, 100]))
print("critical rounds/ray percentiles", np.percentile(seg.max(1), [10, 50, 90, 99, 100]))
print("parts touched percentiles", np.percentile(parts_touched, [10, 50, 90, 99, 100]))
print("corr(pairs, parts touched)", np.corrcoef(pairs, parts_touched)[0, 1], "corr(crit rounds, parts)", np.corrcoef(seg.max(1), parts_touched)[0, 1])
for T in (4, 6, 8, 10):
    hv = parts_touched >= T
    print(f"T={T}: heavy frac {hv.mean():.2f}  mean crit rounds heavy {seg.max(1)[hv].mean():.2f} light {seg.max(1)[~hv].mean() if (~hv).any() else 0:.2f}")
