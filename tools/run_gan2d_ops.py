#!/usr/bin/env python3
"""The 2-D GAN ops a few times each at the discriminator's widest map, for `rocprofv3 --kernel-trace --stats` (kernel
durations without the host side): tools/gpu_gan2d_prof.sh."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from enarf_gan_amd.libraries.custom_stylegan2 import op  # noqa: E402

dev = torch.device("cuda:0")
x = torch.randn(16, 256, 128, 128, device=dev)
half = x[:, :, :64, :64].contiguous()
bias = torch.randn(256, device=dev)
k = op.make_kernel([1, 3, 3, 1]).to(dev)
k4 = k * 4
for _ in range(int(os.environ.get("REPS", 30))):
    op.fused_leaky_relu(x, bias)
    op.upfirdn2d(x, k, pad=(2, 1))
    op.upfirdn2d(x, k, pad=(2, 2))
    op.upfirdn2d(x, k, down=2, pad=(2, 2))
    op.upfirdn2d(half, k4, up=2, pad=(2, 1))
torch.cuda.synchronize()
