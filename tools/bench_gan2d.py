#!/usr/bin/env python3
"""Timing of the 2-D GAN side (not the headline metric): the two HIP ops against the HBM roofline
and against what torch alone would launch for them, then the discriminator (forward, backward, R1 step) and the background
generator at the training configs' shapes (configs/enarfgan_train/*/config.yml: 128 x 128, forward batch 16).
Env: SIZE (128), BATCH (16), ITERS (20)."""
import json
import os
import sys
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from enarf_gan_amd.libraries.custom_stylegan2 import net, op  # noqa: E402
from enarf_gan_amd.libraries.gan.loss import adv_loss_dis, d_r1_loss  # noqa: E402

HBM_PEAK_GBS = 8000.0
S, B, iters = int(os.environ.get("SIZE", 128)), int(os.environ.get("BATCH", 16)), int(os.environ.get("ITERS", 20))
dev = torch.device("cuda:0")


def timed(fn, n=iters, warm=3):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


out = {"workload": f"2-D GAN side, {S}x{S}, batch {B}"}
# ---- ops at the discriminator's widest map: (B, 256, S, S)
C = 256
x = torch.randn(B, C, S, S, device=dev)
bias = torch.randn(C, device=dev)
nbytes = x.numel() * 4
t = timed(lambda: op.fused_leaky_relu(x, bias))
t_ref = timed(lambda: F.leaky_relu(x + bias.view(1, -1, 1, 1), 0.2) * 2 ** 0.5)
out["bias_act"] = {"shape": list(x.shape), "ms": t, "GBps": 2 * nbytes / t / 1e6, "hbm_frac": 2 * nbytes / t / 1e6 / HBM_PEAK_GBS,
                   "torch_three_kernels_ms": t_ref}
k = op.make_kernel([1, 3, 3, 1]).to(dev)
for name, kw, scale in (("blur_pad21", dict(pad=(2, 1)), 1.0), ("blur_down_pad22", dict(pad=(2, 2)), 1.0),
                        ("upsample2", dict(up=2, pad=(2, 1)), 4.0)):
    xin = x if "up" not in kw else x[:, :, : S // 2, : S // 2].contiguous()
    ks = k * scale                      # a module's filter is a buffer: its host copy is taken once
    y = op.upfirdn2d(xin, ks, **kw)
    moved = (xin.numel() + y.numel()) * 4
    t = timed(lambda: op.upfirdn2d(xin, ks, **kw))
    out["upfirdn2d_" + name] = {"in": list(xin.shape), "out": list(y.shape), "ms": t, "GBps": moved / t / 1e6,
                                "hbm_frac": moved / t / 1e6 / HBM_PEAK_GBS}
# the same blur as torch would run it without the op: a depth-wise conv2d on the padded map
w = torch.flip(k, [0, 1]).view(1, 1, 4, 4).repeat(C, 1, 1, 1)
out["upfirdn2d_blur_pad21"]["torch_pad_plus_depthwise_conv_ms"] = timed(lambda: F.conv2d(F.pad(x, [2, 1, 2, 1]), w, groups=C))
del x

# ---- networks
dis = net.Discriminator(SimpleNamespace(minibatch_std=False), size=S).to(dev)
img = torch.randn(B, 3, S, S, device=dev)
real = torch.randn(B, 3, S, S, device=dev, requires_grad=True)


def dis_fwd():
    with torch.no_grad():
        return dis(img)


def dis_step():
    dis.zero_grad(set_to_none=True)
    adv_loss_dis(dis(real), dis(img), "ce").backward()


def r1_step():
    dis.zero_grad(set_to_none=True)
    (0.5 * d_r1_loss(dis(real), real) * 16 * 10).backward()


out["discriminator"] = {"params_M": sum(p.numel() for p in dis.parameters()) / 1e6, "forward_ms": timed(dis_fwd),
                        "real_plus_fake_forward_backward_ms": timed(dis_step), "r1_step_ms": timed(r1_step, n=max(3, iters // 4))}
bg = net.Generator(S, 256, 4, crop_background=True).to(dev)
zb, zr = torch.randn(B, 256, device=dev), torch.randn(B, 256, device=dev)


def bg_fwd():
    with torch.no_grad():
        return bg([zb, zr], inject_index=bg.n_latent - 4)


def bg_step():
    bg.zero_grad(set_to_none=True)
    im, _ = bg([zb, zr], inject_index=bg.n_latent - 4)
    im.square().mean().backward()


out["background_generator"] = {"params_M": sum(p.numel() for p in bg.parameters()) / 1e6, "forward_ms": timed(bg_fwd),
                               "forward_backward_ms": timed(bg_step)}
print(json.dumps(out))
