#!/usr/bin/env python3
"""Deformation-field tri-plane producer: enarf_triplane_warp_fwd / _bwd at B = 16 frames (256^2 planes), beside the NCHW ->
channel-last re-layout of 16 per-frame tri-planes that a generic producer needs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from enarf_gan_amd import ops

dev = torch.device("cuda:0")
B = int(os.environ.get("BATCH", 16))
g = torch.Generator(device=dev).manual_seed(0)
tri = torch.randn(1, 165, 256, 256, device=dev, generator=g)
tri_b = torch.randn(B, 165, 256, 256, device=dev, generator=g)
flow = 3 * torch.randn(B, 6, 256, 256, device=dev, generator=g)
src_cl = ops.triplane_pack(tri)
out = torch.empty(B, 3, 256, 256, 32, device=dev)
gout = torch.randn_like(out)
feat = torch.empty_like(out)
for name, fn, nbytes in (("warp fwd (channel-last out)", lambda: ops.triplane_warp_fwd(src_cl, flow, out), out.numel() * 4 * 2),
                         ("re-layout of B NCHW tri-planes", lambda: ops.triplane_pack(tri_b, feat), out.numel() * 4 * 2),
                         ("warp bwd (d planes + d flow)", lambda: ops.triplane_warp_bwd(gout, src_cl, flow), out.numel() * 4 * 2)):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{name:36s} B={B}: {ms:7.3f} ms  ({nbytes / ms / 1e9:.2f} TB/s of plane bytes)", flush=True)
