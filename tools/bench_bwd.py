#!/usr/bin/env python3
"""Timing of the renderer backward (not the headline metric): enarf_render_bwd kernel + GEMMs + prepare_bwd at C1."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from enarf_gan_amd import ops, synth  # noqa: E402

def _maybe_variant():
    """tools only: ENARF_VARIANT=<name> loads variants/libenarf_<name>.so (tools/build_variant.sh) instead of the in-tree build"""
    import os
    v = os.environ.get("ENARF_VARIANT")
    if v:
        from enarf_gan_amd import _lib
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        _lib.use_variant(v if os.path.sep in v else os.path.join(root, "variants", f"libenarf_{v}.so"))
        print("variant library:", _lib.library_info()["path"], flush=True)


_maybe_variant()

S, B, Nc, Nf = int(os.environ.get("SIZE", 128)), int(os.environ.get("BATCH", 1)), 48, 64
dev = torch.device("cuda:0")
sc = synth.make_scene(S, B, "center_fixed", 20, shared_triplane=True)
cpose, cbl = synth.canonical_buffers(sc, "center_fixed")
d = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in sc.items()}
tri = sc["tri_plane"][:1].contiguous().to(dev)
if os.environ.get("DISTINCT") and B > 1:      # GAN style: one tri-plane per frame
    g = torch.Generator(device=dev).manual_seed(5)
    tri = (tri + 0.05 * torch.randn(B, *tri.shape[1:], device=dev, generator=g)).contiguous()
mlp = {k: v.to(dev) for k, v in sc["mlp"].items()}
n = S * S
coord = d["image_coord"].reshape(B, 3, n).contiguous()
parts, pack = ops.prepare(d["pose_to_camera"], d["bone_length"], cbl.to(dev), d["z_rend"], mlp, sc["parents"], "center_fixed", 3.0)
feat_cl = ops.triplane_pack(tri)
fwd = ops.render_fwd(coord, d["inv_intrinsics"], parts, cpose.to(dev), tri, feat_cl, pack, Nc, Nf, seed=1, mlp_mode="f32", return_bins=True)
bins = fwd.taps["bins"]
gc, gm = torch.randn(B, 3, n, device=dev), torch.randn(B, n, device=dev)
for it in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    grad_tri, dW, db = ops.render_bwd(coord, d["inv_intrinsics"], parts, cpose.to(dev), tri, feat_cl, pack, Nf, bins, gc, gm)
    pg, dz = ops.prepare_bwd(d["z_rend"], mlp, dW)
    e1.record()
    torch.cuda.synchronize()
    print(f"backward {S}x{S} B={B}: wall {1e3 * (time.perf_counter() - t0):.2f} ms, device {e0.elapsed_time(e1):.2f} ms, "
          f"|grad_tri| {float(grad_tri.abs().sum()):.3f}", flush=True)
