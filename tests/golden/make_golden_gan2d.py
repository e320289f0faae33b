#!/usr/bin/env python3
"""Generate tests/golden/gan2d_*.npz: the reference's OWN 2-D GAN classes (libraries/custom_stylegan2/net.py: EqualConv2d,
EqualLinear, StyledConv, ToRGB, Generator, ConvLayer, ResBlock, Discriminator) run on the CPU in this container.

Those classes import five names from an un-vendored, empty submodule (net.py:12-14). The harness binds those names to this
repo's CPU restatement of the published ops (oracle/gan_ops_oracle.py) BEFORE any class is instantiated - the reference's
files are imported unmodified, nothing of them is copied - and `kornia.augmentation.RandomCrop` (constructed in
Generator.__init__, used only in training mode) to an identity. What the fixtures therefore pin is everything the
reference's file itself defines: state-dict keys and shapes, the padding arithmetic, the residual / skip wiring, the
minibatch-stddev feature, latent injection, the crop. Weights are a function of the state-dict key
(gan_ops_oracle.fill_by_name), so a fixture holds inputs and outputs only.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_gan2d.py
"""
import os
import sys
import types

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
from torch import nn  # noqa: E402

import make_golden as harness  # noqa: E402  (puts /root/reference on sys.path; placeholders for what is not installed)
from oracle import gan_ops_oracle as third  # noqa: E402


def reference_net():
    harness.install_placeholders()
    import libraries.custom_stylegan2.net as refnet          # the reference's file, unmodified
    for name in ("FusedLeakyReLU", "fused_leaky_relu", "PixelNorm", "Upsample", "Blur", "ModulatedConv2d"):
        setattr(refnet, name, getattr(third, name))
    refnet.kornia = types.SimpleNamespace(augmentation=types.SimpleNamespace(RandomCrop=lambda *a, **k: nn.Identity()))
    return refnet


def keys_and_shapes(module):
    sd = module.state_dict()
    return np.array(sorted(sd.keys())), np.array([",".join(map(str, sd[k].shape)) for k in sorted(sd.keys())])


def discriminator_case(refnet, name, size, batch, minibatch_std, seed):
    torch.manual_seed(seed)
    dis = refnet.Discriminator(types.SimpleNamespace(minibatch_std=minibatch_std), size=size)
    third.fill_by_name(dis)
    x = torch.randn(batch, 3, size, size, generator=torch.Generator().manual_seed(seed)).requires_grad_(True)
    out = dis(x)
    # R1 (libraries/gan/loss.py:25-31) and the gradient of the penalty with respect to two parameters: second derivatives
    (gx,) = torch.autograd.grad(out.sum(), x, create_graph=True)
    r1 = gx.pow(2).reshape(batch, -1).sum(1).mean()
    wrt = [dis.final_linear[1].weight, dis.convs[0][0].weight]
    g_r1 = torch.autograd.grad(r1, wrt)
    keys, shapes = keys_and_shapes(dis)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), size=size, minibatch_std=int(minibatch_std), x=x.detach().numpy(),
                        out=out.detach().numpy(), grad_x=gx.detach().numpy(), r1=r1.detach().numpy(),
                        r1_grad_final_linear_1_weight=g_r1[0].numpy(), r1_grad_convs_0_0_weight=g_r1[1].numpy(),
                        keys=keys, shapes=shapes)
    print(name, "out", out.detach().flatten().tolist(), "r1", float(r1))


def generator_case(refnet, name, size, style_dim, batch, crop_background, seed):
    torch.manual_seed(seed)
    gen = refnet.Generator(size=size, style_dim=style_dim, n_mlp=4, last_channel=3, crop_background=crop_background).eval()
    third.fill_by_name(gen)
    g = torch.Generator().manual_seed(seed)
    z_bg, z_render = torch.randn(batch, style_dim, generator=g), torch.randn(batch, style_dim, generator=g)
    # explicit per-layer noise maps (the registered square buffers do not fit the 1 : 2 maps of crop_background)
    wide = 2 if crop_background else 1
    res = [4] + [2 ** lv for lv in range(3, gen.log_size + 1) for _ in range(2)]
    noise = [torch.randn(1, 1, r, r * wide, generator=g) for r in res]
    with torch.no_grad():
        # as models/generator.py:102-103 calls it
        img, _ = gen([z_bg, z_render], inject_index=gen.n_latent - 4, noise=noise)
        one, lat = gen([z_bg], return_latents=True, noise=noise)
    keys, shapes = keys_and_shapes(gen)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), size=size, style_dim=style_dim, crop_background=int(crop_background),
                        z_bg=z_bg.numpy(), z_render=z_render.numpy(), image=img.numpy(), image_one_style=one.numpy(),
                        latent_one_style=lat.numpy(), n_latent=gen.n_latent, keys=keys, shapes=shapes,
                        **{f"noise_{i}": t.numpy() for i, t in enumerate(noise)})
    print(name, tuple(img.shape), float(img.abs().mean()))


def main():
    refnet = reference_net()
    discriminator_case(refnet, "gan2d_dis_32_std", 32, 4, True, 3)
    discriminator_case(refnet, "gan2d_dis_16_nostd", 16, 2, False, 4)
    generator_case(refnet, "gan2d_gen_32_crop", 32, 16, 2, True, 5)
    generator_case(refnet, "gan2d_gen_16", 16, 8, 2, False, 6)


if __name__ == "__main__":
    main()
