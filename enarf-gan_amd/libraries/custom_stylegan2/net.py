"""The 2-D GAN networks either side of the renderer: the StyleGAN2 background generator and the
residual discriminator of libraries/custom_stylegan2/net.py:346-536 and :539-676, on this repo's HIP ops
(`op.py`: fused bias + leaky ReLU, upfirdn2d) and torch's library convolutions (MIOpen / hipBLASLt on ROCm).

State-dict keys follow the reference's, so its snapshots load by name (`formats.load_generator_snapshot`,
`Discriminator.load_state_dict(snapshot["dis"])`): `convs.N.{conv1,conv2,skip}.M.{weight,bias,kernel}`, `final_conv.*`,
`final_linear.*` for the discriminator; `style.N.*`, `input.input`, `conv1.*`, `to_rgb1.*`, `convs.N.*`, `to_rgbs.N.*`,
`noises.noise_N` for the generator. The 1-D classes of the same reference file (EqualConv1d, ModulatedConv1d, the StyledMLP)
live with the renderer: `models/narf.py`, fused into the HIP kernels.

Parity: the classes defined IN the reference file are pinned by `tests/golden/gan2d_*.npz`, captured from the imported
reference with this repo's CPU restatement of the two un-vendored ops standing in for the missing submodule
(`tests/golden/make_golden.py`); the ops themselves (rosinality/stylegan2-pytorch `fused_leaky_relu`, `upfirdn2d`,
`ModulatedConv2d`; no commit recorded in the reference checkout) follow their published definitions: parity unpinned.
"""
from __future__ import annotations

import math
import random
from typing import List, Optional, Sequence, Tuple

import torch
from torch import nn
from torch.nn import functional as F

from .op import Blur, FusedLeakyReLU, PixelNorm, Upsample, fused_leaky_relu

SQRT2 = math.sqrt(2.0)


def channel_table(multiplier: int = 2) -> dict:
    """feature maps per resolution (net.py:383-393, :605-615)"""
    table = {res: 512 for res in (4, 8, 16, 32)}
    table.update({64: 256 * multiplier, 128: 128 * multiplier, 256: 64 * multiplier, 512: 32 * multiplier, 1024: 16 * multiplier})
    return table


# ---------------------------------------------------------------------------------------------------- equalised layers
class EqualConv2d(nn.Module):
    """conv2d with N(0, 1) weights and the He constant applied at run time (net.py:29-66)"""

    def __init__(self, in_channel, out_channel, kernel_size, stride=1, padding=0, groups=1, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(out_channel, in_channel // groups, kernel_size, kernel_size))
        self.scale = 1.0 / math.sqrt(in_channel // groups * kernel_size ** 2)
        self.stride, self.padding, self.groups = stride, padding, groups
        self.bias = nn.Parameter(torch.zeros(out_channel)) if bias else None

    def forward(self, input):
        return F.conv2d(input, self.weight * self.scale, self.bias, self.stride, self.padding, 1, self.groups)

    def extra_repr(self):
        o, i, k, _ = self.weight.shape
        return f"{i}, {o}, {k}, stride={self.stride}, padding={self.padding}"


class EqualLinear(nn.Module):
    """net.py:128-174; `activation` (any non-None value, the reference passes 'fused_lrelu') fuses bias and leaky ReLU"""

    def __init__(self, in_dim, out_dim, bias=True, bias_init=0, lr_mul=1, activation=None, w=1):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(out_dim, in_dim).div_(lr_mul))
        self.bias = nn.Parameter(torch.full((out_dim,), float(bias_init))) if bias else None
        self.activation = activation
        self.scale = (w / math.sqrt(in_dim)) * lr_mul
        self.lr_mul = lr_mul
        self.in_dim, self.out_dim = in_dim, out_dim

    def forward(self, input):
        bias = None if self.bias is None else self.bias * self.lr_mul
        if self.activation is None:
            return F.linear(input, self.weight * self.scale, bias)
        if bias is None:
            raise ValueError("EqualLinear with an activation needs a bias (net.py:163)")
        return fused_leaky_relu(F.linear(input, self.weight * self.scale), bias)

    def extra_repr(self):
        return f"{self.in_dim}, {self.out_dim}"


class ScaledLeakyReLU(nn.Module):
    def __init__(self, negative_slope=0.2):
        super().__init__()
        self.negative_slope = negative_slope

    def forward(self, input):
        return F.leaky_relu(input, self.negative_slope) * SQRT2


class NoiseInjection(nn.Module):
    def __init__(self):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(1))

    def forward(self, image, noise: Optional[torch.Tensor] = None):
        if noise is None:
            noise = torch.randn_like(image[:, :1])
        return image + self.weight * noise


class ConstantInput(nn.Module):
    def __init__(self, channel, size=4, size2=4):
        super().__init__()
        self.input = nn.Parameter(torch.randn(1, channel, size, size2))

    def forward(self, input):
        return self.input.expand(input.shape[0], -1, -1, -1)


# ---------------------------------------------------------------------------------------------------- modulated conv
class ModulatedConv2d(nn.Module):
    """StyleGAN2 weight (de)modulation as one grouped convolution over the batch (published algorithm: Karras et al. 2020,
    sec. 2.2; rosinality/stylegan2-pytorch model.py ModulatedConv2d): w'_{b,o,i,:,:} = scale * w_{o,i} * s_{b,i}, optionally
    normalised per (b, o) to unit L2 norm; up-sampling = stride-2 transposed convolution then a 4-tap blur scaled by 4,
    down-sampling = blur then stride-2 convolution."""

    def __init__(self, in_channel, out_channel, kernel_size, style_dim, demodulate=True, upsample=False, downsample=False,
                 blur_kernel: Sequence[int] = (1, 3, 3, 1)):
        super().__init__()
        self.eps = 1e-8
        self.kernel_size, self.in_channel, self.out_channel = kernel_size, in_channel, out_channel
        self.upsample, self.downsample = upsample, downsample
        taps = len(blur_kernel)
        if upsample:
            p = (taps - 2) - (kernel_size - 1)
            self.blur = Blur(blur_kernel, pad=((p + 1) // 2 + 1, p // 2 + 1), upsample_factor=2)
        if downsample:
            p = (taps - 2) + (kernel_size - 1)
            self.blur = Blur(blur_kernel, pad=((p + 1) // 2, p // 2))
        self.scale = 1.0 / math.sqrt(in_channel * kernel_size ** 2)
        self.padding = kernel_size // 2
        self.weight = nn.Parameter(torch.randn(1, out_channel, in_channel, kernel_size, kernel_size))
        self.modulation = EqualLinear(style_dim, in_channel, bias_init=1)
        self.demodulate = demodulate

    def extra_repr(self):
        return f"{self.in_channel}, {self.out_channel}, {self.kernel_size}, upsample={self.upsample}, downsample={self.downsample}"

    def forward(self, input, style):
        """The modulation is applied to the ACTIVATIONS and the demodulation to the output, x -> d_{b,o} * conv(x * s_b, scale * w),
        which is the same function as convolving with the per-sample weights scale * w_{o,i} * s_{b,i} * d_{b,o} (the published
        form: one grouped convolution with batch * out filters) - but a dense convolution with ONE filter bank, which the library
        runs 2-3x faster than groups = batch (tools/bench_gan2d.py). d_{b,o} = rsqrt(sum_i s_{b,i}^2 sum_k (scale * w_{o,i,k})^2 + eps)."""
        b, cin, _, _ = input.shape
        s = self.modulation(style)                                              # (b, in)
        weight = self.scale * self.weight[0]                                    # (out, in, k, k)
        x = input * s.view(b, cin, 1, 1)
        if self.upsample:
            out = self.blur(F.conv_transpose2d(x, weight.transpose(0, 1), padding=0, stride=2))
        elif self.downsample:
            out = F.conv2d(self.blur(x), weight, padding=0, stride=2)
        else:
            out = F.conv2d(x, weight, padding=self.padding)
        if self.demodulate:
            energy = weight.pow(2).sum([2, 3])                                  # (out, in)
            d = torch.rsqrt(F.linear(s * s, energy) + self.eps)                 # (b, out)
            out = out * d.view(b, self.out_channel, 1, 1)
        return out


class StyledConv(nn.Module):
    """modulated conv + noise + bias + leaky ReLU * sqrt 2, in the reference's order (net.py:309-320: the bias is added after
    the noise and the activation is a plain LeakyReLU, not the fused op). 2-D only: the 1-D variant (`conv_1d=True`) is the
    renderer's StyledMLP, models/narf.py."""

    def __init__(self, in_channel, out_channel, kernel_size, style_dim, upsample=False, blur_kernel=(1, 3, 3, 1), demodulate=True,
                 use_noise=True):
        super().__init__()
        self.use_noise = use_noise
        self.conv = ModulatedConv2d(in_channel, out_channel, kernel_size, style_dim, demodulate=demodulate, upsample=upsample,
                                    blur_kernel=blur_kernel)
        self.bias = nn.Parameter(torch.zeros(1, out_channel, 1, 1))
        self.noise = NoiseInjection()
        self.activate = nn.LeakyReLU(0.2)

    def forward(self, input, style, noise: Optional[torch.Tensor] = None):
        out = self.conv(input, style)
        if self.use_noise:
            out = self.noise(out, noise=noise)
        return self.activate(out + self.bias) * SQRT2


class ToRGB(nn.Module):
    def __init__(self, in_channel, style_dim, upsample=True, blur_kernel=(1, 3, 3, 1), out_channel=3):
        super().__init__()
        if upsample:
            self.upsample = Upsample(blur_kernel)
        self.conv = ModulatedConv2d(in_channel, out_channel, 1, style_dim, demodulate=False)
        self.bias = nn.Parameter(torch.zeros(1, out_channel, 1, 1))

    def forward(self, input, style, skip=None):
        out = self.conv(input, style) + self.bias
        if skip is not None:
            out = out + self.upsample(skip)
        return out


# ---------------------------------------------------------------------------------------------------- background generator
def _random_window(image: torch.Tensor, width: int) -> torch.Tensor:
    """one random horizontal window of `width` columns per sample (the reference: kornia RandomCrop((size, size),
    resample='NEAREST') on a size x 2 size image, net.py:437,529 - an integer crop)"""
    b, _, h, w = image.shape
    if w == width:
        return image
    x0 = torch.randint(0, w - width + 1, (b,), device=image.device)
    cols = x0[:, None] + torch.arange(width, device=image.device)[None, :]          # (b, width)
    return torch.gather(image, 3, cols[:, None, None, :].expand(b, image.shape[1], h, width))


class Generator(nn.Module):
    """StyleGAN2 skip generator used as the background network (net.py:346-536): `size` x `size` output, or `size` x
    2 `size` cropped to a window when `crop_background` (a 4 x 8 constant input)."""

    def __init__(self, size, style_dim, n_mlp, channel_multiplier=2, blur_kernel=(1, 3, 3, 1), lr_mlp=0.01, last_channel=3,
                 crop_background=False):
        super().__init__()
        self.size, self.style_dim, self.crop_background = size, style_dim, crop_background
        self.style = nn.Sequential(PixelNorm(), *[EqualLinear(style_dim, style_dim, lr_mul=lr_mlp, activation="fused_lrelu")
                                                  for _ in range(n_mlp)])
        self.channels = channel_table(channel_multiplier)
        self.log_size = int(math.log2(size))
        self.num_layers = 2 * (self.log_size - 2) + 1
        self.n_latent = 2 * self.log_size - 2

        width = self.channels[4]
        self.input = ConstantInput(width, size2=8 if crop_background else 4)
        self.conv1 = StyledConv(width, width, 3, style_dim, blur_kernel=blur_kernel)
        self.to_rgb1 = ToRGB(width, style_dim, upsample=False, out_channel=last_channel)
        self.convs, self.upsamples, self.to_rgbs, self.noises = nn.ModuleList(), nn.ModuleList(), nn.ModuleList(), nn.Module()
        for layer in range(self.num_layers):
            res = 2 ** ((layer + 5) // 2)
            self.noises.register_buffer(f"noise_{layer}", torch.randn(1, 1, res, res))
        for level in range(3, self.log_size + 1):
            nxt = self.channels[2 ** level]
            self.convs.append(StyledConv(width, nxt, 3, style_dim, upsample=True, blur_kernel=blur_kernel))
            self.convs.append(StyledConv(nxt, nxt, 3, style_dim, blur_kernel=blur_kernel))
            self.to_rgbs.append(ToRGB(nxt, style_dim, out_channel=last_channel))
            width = nxt

    # -- latents
    def make_noise(self) -> List[torch.Tensor]:
        dev = self.input.input.device
        sizes = [4] + [2 ** level for level in range(3, self.log_size + 1) for _ in range(2)]
        return [torch.randn(1, 1, s, s, device=dev) for s in sizes]

    def mean_latent(self, n_latent):
        z = torch.randn(n_latent, self.style_dim, device=self.input.input.device)
        return self.style(z).mean(0, keepdim=True)

    def get_latent(self, input):
        return self.style(input)

    def _latent_stack(self, styles, inject_index):
        if len(styles) == 1:
            w = styles[0]
            return w if w.dim() == 3 else w.unsqueeze(1).expand(-1, self.n_latent, -1)
        if inject_index is None:
            inject_index = random.randint(1, self.n_latent - 1)
        head = styles[0].unsqueeze(1).expand(-1, inject_index, -1)
        tail = styles[1].unsqueeze(1).expand(-1, self.n_latent - inject_index, -1)
        return torch.cat([head, tail], dim=1)

    def forward(self, styles, return_latents=False, inject_index=None, truncation=1, truncation_latent=None,
                input_is_latent=False, noise=None, randomize_noise=True):
        if not input_is_latent:
            styles = [self.style(s) for s in styles]
        if noise is None:
            noise = [None] * self.num_layers if randomize_noise else [getattr(self.noises, f"noise_{i}") for i in range(self.num_layers)]
        if truncation < 1:
            styles = [truncation_latent + truncation * (s - truncation_latent) for s in styles]
        latent = self._latent_stack(styles, inject_index)

        out = self.conv1(self.input(latent), latent[:, 0], noise=noise[0])
        skip = self.to_rgb1(out, latent[:, 1])
        for level, to_rgb in enumerate(self.to_rgbs):
            i = 1 + 2 * level
            out = self.convs[2 * level](out, latent[:, i], noise=noise[i])
            out = self.convs[2 * level + 1](out, latent[:, i + 1], noise=noise[i + 1])
            skip = to_rgb(out, latent[:, i + 2], skip)
        image = skip
        if self.crop_background:
            image = _random_window(image, self.size) if self.training else image[:, :, :, self.size // 2: self.size * 3 // 2]
        return image, (latent if return_latents else None)


class PretrainedStyleGAN(nn.Module):
    """A 256 x 256 church generator cropped to 128-pixel windows (net.py:679-712). The reference loads
    `__stylegan2_pytorch/stylegan2-church-config-f.pt` from the working directory - a download that ships with neither
    repository; pass the path of that checkpoint (read with `weights_only=True`)."""

    def __init__(self, ckpt: str = "__stylegan2_pytorch/stylegan2-church-config-f.pt"):
        super().__init__()
        g = Generator(256, 512, 8, channel_multiplier=2)
        state = torch.load(ckpt, map_location="cpu", weights_only=True)
        g.load_state_dict(state["g_ema"])
        g.input.input = nn.Parameter(g.input.input[:, :, 1:-1].data)
        self.size, self.gen, self.n_latent = 128, g, g.n_latent

    def forward(self, z: Tuple[torch.Tensor, torch.Tensor], inject_index):
        sample, _ = self.gen([torch.cat(z, dim=1)], inject_index=inject_index)
        sample = _random_window(sample, self.size) if self.training else sample[:, :, :, self.size // 2: self.size * 3 // 2]
        return sample, None


# ---------------------------------------------------------------------------------------------------- discriminator
class ConvLayer(nn.Sequential):
    """[Blur] -> EqualConv2d -> [FusedLeakyReLU | ScaledLeakyReLU] (net.py:539-585); a Sequential so that the state-dict
    keys are the positions 0, 1, 2"""

    def __init__(self, in_channel, out_channel, kernel_size, downsample=False, blur_kernel=(1, 3, 3, 1), bias=True, activate=True):
        stages: List[nn.Module] = []
        if downsample:
            p = (len(blur_kernel) - 2) + (kernel_size - 1)
            stages.append(Blur(blur_kernel, pad=((p + 1) // 2, p // 2)))
            stride, self.padding = 2, 0
        else:
            stride, self.padding = 1, kernel_size // 2
        stages.append(EqualConv2d(in_channel, out_channel, kernel_size, padding=self.padding, stride=stride, bias=bias and not activate))
        if activate:
            stages.append(FusedLeakyReLU(out_channel) if bias else ScaledLeakyReLU(0.2))
        super().__init__(*stages)


class ResBlock(nn.Module):
    def __init__(self, in_channel, out_channel, blur_kernel=(1, 3, 3, 1)):
        super().__init__()
        self.conv1 = ConvLayer(in_channel, in_channel, 3)
        self.conv2 = ConvLayer(in_channel, out_channel, 3, downsample=True)
        self.skip = ConvLayer(in_channel, out_channel, 1, downsample=True, activate=False, bias=False)

    def forward(self, input):
        return (self.conv2(self.conv1(input)) + self.skip(input)) / SQRT2


def minibatch_stddev(feat: torch.Tensor, group: int, ddp: bool = False, world_size: int = 1) -> torch.Tensor:
    """One extra feature map: the standard deviation over groups of `group` samples, averaged over channels and pixels
    (net.py:652-669; stddev_feat = 1). With `ddp` the statistic is averaged over the ranks (an all-reduce of batch / group
    floats, net.py:665-667) - a detail of the reference kept as it is: gradients do not flow through the collective."""
    b, c, h, w = feat.shape
    g = min(b, group)
    y = feat.view(g, -1, 1, c, h, w)
    y = torch.sqrt(y.var(0, unbiased=False) + 1e-8).mean([2, 3, 4], keepdim=True).squeeze(2)       # (b / g, 1, 1, 1)
    if ddp:
        torch.distributed.all_reduce(y)
        y = y / world_size
    return torch.cat([feat, y.repeat(g, 1, h, w)], dim=1)


class Discriminator(nn.Module):
    """Residual StyleGAN2 discriminator (net.py:602-676); `config.minibatch_std` switches the stddev feature."""

    def __init__(self, config, size, in_dim=3, channel_multiplier=2, blur_kernel=(1, 3, 3, 1)):
        super().__init__()
        ch = channel_table(channel_multiplier)
        blocks: List[nn.Module] = [ConvLayer(in_dim, ch[size], 1)]
        width = ch[size]
        for level in range(int(math.log2(size)), 2, -1):
            blocks.append(ResBlock(width, ch[2 ** (level - 1)], blur_kernel))
            width = ch[2 ** (level - 1)]
        self.convs = nn.Sequential(*blocks)
        self.minibatch_std = config.minibatch_std
        self.stddev_group, self.stddev_feat = 4, 1
        self.final_conv = ConvLayer(width + (1 if self.minibatch_std else 0), ch[4], 3)
        self.final_linear = nn.Sequential(EqualLinear(ch[4] * 4 * 4, ch[4], activation="fused_lrelu"), EqualLinear(ch[4], 1))

    def forward(self, input, ddp=False, world_size=1):
        out = self.convs(input)
        if self.minibatch_std:
            out = minibatch_stddev(out, self.stddev_group, ddp, world_size)
        out = self.final_conv(out)
        return self.final_linear(out.view(out.shape[0], -1))
