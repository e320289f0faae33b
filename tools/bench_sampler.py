#!/usr/bin/env python3
"""Throughput of the a1 operator (the TriplaneSampler replacement) in the shape the reference would use it
(libraries/triplane/sampling.py:25-26): input (1, 96, 256, 256), grid (1, N, 1, 3), N = 128*128*64 points."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from enarf_gan_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
inp = torch.randn(1, 96, 256, 256, device=dev, generator=g)
N = 128 * 128 * 64
grid = (torch.rand(1, N, 1, 3, device=dev, generator=g) * 2 - 1).contiguous()
go = torch.randn(1, 32, N, 1, device=dev, generator=g)
for name, fn in (("fwd channel-last (pack + gather)", lambda: ops.triplane_sample_fwd(inp, grid, use_workspace=True)),
                 ("fwd direct NCHW", lambda: ops.triplane_sample_fwd(inp, grid, use_workspace=False)),
                 ("bwd channel-last atomics (input + grid)", lambda: ops.triplane_sample_bwd(go, inp, grid, 0, 0, False, True, True)),
                 ("bwd direct NCHW (grad input + grid)", lambda: ops.triplane_sample_bwd(go, inp, grid, 0, 0, False, True, True, use_workspace=False))):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"{name:40s} {ms:8.3f} ms  {N / ms / 1e6:8.2f} G points/s  ({N / (128 * 128 * 64) * 16384 / ms / 1e3:.1f} M rays/s at 64 samples/ray)", flush=True)
