"""Rate of the 4 x 4 blur (enarf_upfirdn2d) against the output width: is it the odd row pitch or the tile quantisation that
costs the 129-wide case? (profiles/r03_gan2d_traffic.json: the latter.)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from enarf_gan_amd.libraries.custom_stylegan2 import op
dev = torch.device("cuda:0")
k = op.make_kernel([1, 3, 3, 1]).to(dev)
def timed(fn, n=30):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
for (H, W, pad) in [(128, 128, (2, 1)), (128, 128, (2, 2)), (128, 159, (2, 2, 2, 1)), (128, 127, (2, 2, 2, 1)), (128, 160, (2, 1)), (128, 125, (2, 2, 2, 1))]:
    x = torch.randn(16, 256, H, W, device=dev)
    y = op.upfirdn2d(x, k, pad=pad)
    t = timed(lambda: op.upfirdn2d(x, k, pad=pad))
    mb = (x.numel() + y.numel()) * 4 / 1e6
    print(f"in {H}x{W} pad {pad} -> out {tuple(y.shape[2:])}: {t:.4f} ms  {mb / t / 1e3:.2f} TB/s", flush=True)
