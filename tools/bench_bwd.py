#!/usr/bin/env python3
"""Timing of the renderer backward (not the headline metric) at C1-like shapes: enarf_render_bwd, the tri-plane un-pack,
enarf_weight_grad and enarf_prepare_bwd, each bracketed by events on the launch stream, plus the kernel's own counters
(pairs, tiles, 128-B feature-gradient lines added, part-probability adds) against the float-atomic ceiling.
Env: SIZE, BATCH, NC, NF, DISTINCT=1 (one tri-plane per frame), ITERS, ENARF_VARIANT=<name> (tools/build_variant.sh)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from enarf_gan_amd import _lib, ops, synth  # noqa: E402

ATOMIC_CEILING_GBS = 1300.0      # MI355X_MICROARCH.md, Global float atomics: ~1.3 TB/s of added bytes chip-wide
HBM_PEAK_GBS = 8000.0


def _maybe_variant():
    """tools only: ENARF_VARIANT=<name> loads variants/libenarf_<name>.so (tools/build_variant.sh) instead of the in-tree build"""
    v = os.environ.get("ENARF_VARIANT")
    if v:
        _lib.use_variant(v if os.path.sep in v else os.path.join(ROOT, "variants", f"libenarf_{v}.so"))
        print("variant library:", _lib.library_info()["path"], flush=True)


_maybe_variant()

S, B = int(os.environ.get("SIZE", 128)), int(os.environ.get("BATCH", 1))
Nc, Nf = int(os.environ.get("NC", 48)), int(os.environ.get("NF", 64))
iters = int(os.environ.get("ITERS", 6))
dev = torch.device("cuda:0")
sc = synth.make_scene(S, B, "center_fixed", 20, shared_triplane=True)
cpose, cbl = synth.canonical_buffers(sc, "center_fixed")
d = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in sc.items()}
tri = sc["tri_plane"][:1].contiguous().to(dev)
distinct = bool(os.environ.get("DISTINCT")) and B > 1
if distinct:      # GAN style: one tri-plane per frame
    g = torch.Generator(device=dev).manual_seed(5)
    tri = (tri + 0.05 * torch.randn(B, *tri.shape[1:], device=dev, generator=g)).contiguous()
mlp = {k: v.to(dev) for k, v in sc["mlp"].items()}
n = S * S
coord = d["image_coord"].reshape(B, 3, n).contiguous()
parts, pack = ops.prepare(d["pose_to_camera"], d["bone_length"], cbl.to(dev), d["z_rend"], mlp, sc["parents"], "center_fixed", 3.0)
feat_cl = ops.triplane_pack(tri)
fwd = ops.render_fwd(coord, d["inv_intrinsics"], parts, cpose.to(dev), tri, feat_cl, pack, Nc, Nf, seed=1, mlp_mode="f16x3", return_bins=True)
bins = fwd.taps["bins"]
torch.manual_seed(0)          # the same output gradients for every build: |grad_tri| is comparable across variants
gc, gm = torch.randn(B, 3, n, device=dev), torch.randn(B, n, device=dev)
cpose_d = cpose.to(dev)


def backward(counters=None):
    grad_tri, dW, db = ops.render_bwd(coord, d["inv_intrinsics"], parts, cpose_d, tri, feat_cl, pack, Nf, bins, gc, gm,
                                      counters=counters)
    pg, dz = ops.prepare_bwd(d["z_rend"], mlp, dW)
    return grad_tri


cnt = torch.zeros(8, dtype=torch.int64, device=dev)
backward(cnt)
torch.cuda.synchronize()
pairs, tiles, rays, lines, madds, rounds = [int(x) for x in cnt[:6].tolist()]
best = None
for it in range(iters):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    grad_tri = backward()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    best = ms if best is None else min(best, ms)
    print(f"backward {S}x{S} B={B} Nf={Nf}{' per-frame tri-planes' if distinct else ''}: wall {1e3 * (time.perf_counter() - t0):.2f} ms, "
          f"device {ms:.3f} ms, |grad_tri| {float(grad_tri.abs().sum()):.3f}", flush=True)
atomic_bytes = lines * 128 + madds * 4
print(json.dumps({
    "workload": f"backward {S}x{S} B={B} Nc {Nc} Nf {Nf}{' per-frame tri-planes' if distinct else ''}", "ms": best,
    "library": _lib.library_info(),
    "pairs": pairs, "tiles": tiles, "rays": rays, "gather_rounds": rounds,
    "feature_lines_added": lines, "feature_line_bytes_per_pair": lines * 128 / max(pairs, 1),
    "unmerged_feature_bytes_per_pair": 12 * 128,
    "mask_adds": madds, "atomic_bytes": atomic_bytes,
    "atomic_floor_ms": atomic_bytes / ATOMIC_CEILING_GBS / 1e6,
    "atomic_frac_of_ceiling_over_whole_backward": atomic_bytes / (best * 1e-3) / 1e9 / ATOMIC_CEILING_GBS,
    "row_bytes": tiles * 16 * 144,
}), flush=True)
