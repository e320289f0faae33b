"""CPU restatement (plain PyTorch, fp32 / fp64) of the third-party ops under the reference's 2-D GAN networks.

TEST INFRASTRUCTURE ONLY: imported by tests/ and tests/golden/make_golden_gan2d.py, never by the product
(enarf_gan_amd runs these ops on the HIP library: libraries/custom_stylegan2/op.py).

What is restated, and from where. The reference's libraries/custom_stylegan2/net.py:12-14 imports
    FusedLeakyReLU, fused_leaky_relu                         from libraries.stylegan2_pytorch.op
    PixelNorm, Upsample, Blur, ModulatedConv2d, Generator    from libraries.stylegan2_pytorch.model
i.e. rosinality/stylegan2-pytorch (.gitmodules:7-9), a submodule whose directory is EMPTY in /root/reference and whose
commit is not recorded there. The restatement follows the published algorithms those names stand for:
  * upfirdn2d - Karras et al., StyleGAN2 (CVPR 2020), official `upfirdn_2d` reference semantics: insert up - 1 zeros after
    every sample, pad (negative pad = crop), correlate with the FLIPPED filter (a convolution), keep every down-th sample;
    1-D form = scipy.signal.upfirdn, which tests/test_gan2d_cpu.py uses as an independent check;
  * fused_leaky_relu(x, b, slope 0.2, scale sqrt 2) = scale * leaky_relu(x + b[None, :, None, ...], slope);
  * Blur(taps, pad, upsample_factor): the normalised outer-product filter (times upsample_factor^2) through upfirdn2d;
    Upsample(taps, factor 2): the same filter times factor^2, up = factor, pad = ((p + 1) // 2 + factor - 1, p // 2), p = taps - factor;
  * PixelNorm: x * rsqrt(mean_c(x^2) + 1e-8);
  * ModulatedConv2d: weight (de)modulation as a grouped convolution (StyleGAN2 sec. 2.2), up-sampling by a stride-2
    transposed convolution followed by Blur(pad = ((p + 1) // 2 + 1, p // 2 + 1), factor 2) with p = taps - 2 - (k - 1),
    down-sampling by Blur(pad = ((p + 1) // 2, p // 2)) with p = taps - 2 + (k - 1) followed by a stride-2 convolution.
PARITY UNPINNED for these ops against the reference's own dependency (absent, version unknown); the classes that ARE in the
reference's file (EqualConv2d, EqualLinear, StyledConv, ToRGB, Generator, ConvLayer, ResBlock, Discriminator) are pinned by
running the imported reference on top of this restatement (tests/golden/gan2d_*.npz).
"""
import math

import torch
from torch import nn
from torch.nn import functional as F


def upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0)):
    """input (N, C, H, W), kernel (kh, kw); pad (p0, p1) for both axes or (x0, x1, y0, y1)"""
    pad = tuple(pad)
    if len(pad) == 2:
        pad = (pad[0], pad[1], pad[0], pad[1])
    px0, px1, py0, py1 = pad
    n, c, h, w = input.shape
    kh, kw = kernel.shape
    x = input.reshape(n * c, 1, h, w)
    if up > 1:                                                   # zeros AFTER every sample
        z = x.new_zeros(n * c, 1, h * up, w * up)
        z[:, :, ::up, ::up] = x
        x = z
    x = F.pad(x, [max(px0, 0), max(px1, 0), max(py0, 0), max(py1, 0)])
    x = x[:, :, max(-py0, 0): x.shape[2] - max(-py1, 0), max(-px0, 0): x.shape[3] - max(-px1, 0)]
    x = F.conv2d(x, torch.flip(kernel, [0, 1]).to(x.dtype).view(1, 1, kh, kw))
    x = x[:, :, ::down, ::down]
    return x.reshape(n, c, x.shape[2], x.shape[3])


def fused_leaky_relu(input, bias=None, negative_slope=0.2, scale=2 ** 0.5):
    if bias is not None:
        input = input + bias.view(1, -1, *([1] * (input.dim() - 2)))
    return F.leaky_relu(input, negative_slope) * scale


class FusedLeakyReLU(nn.Module):
    def __init__(self, channel, bias=True, negative_slope=0.2, scale=2 ** 0.5):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(channel)) if bias else None
        self.negative_slope, self.scale = negative_slope, scale

    def forward(self, input):
        return fused_leaky_relu(input, self.bias, self.negative_slope, self.scale)


def make_kernel(k):
    k = torch.as_tensor(k, dtype=torch.float32)
    if k.dim() == 1:
        k = k[None, :] * k[:, None]
    return k / k.sum()


class Blur(nn.Module):
    def __init__(self, kernel, pad, upsample_factor=1):
        super().__init__()
        k = make_kernel(kernel)
        if upsample_factor > 1:
            k = k * upsample_factor ** 2
        self.register_buffer("kernel", k)
        self.pad = pad

    def forward(self, input):
        return upfirdn2d(input, self.kernel, pad=self.pad)


class Upsample(nn.Module):
    def __init__(self, kernel, factor=2):
        super().__init__()
        self.factor = factor
        self.register_buffer("kernel", make_kernel(kernel) * factor ** 2)
        p = self.kernel.shape[0] - factor
        self.pad = ((p + 1) // 2 + factor - 1, p // 2)

    def forward(self, input):
        return upfirdn2d(input, self.kernel, up=self.factor, down=1, pad=self.pad)


class PixelNorm(nn.Module):
    def forward(self, input):
        return input * torch.rsqrt(torch.mean(input ** 2, dim=1, keepdim=True) + 1e-8)


class _EqualLinear(nn.Module):
    """the modulation layer of ModulatedConv2d (weights N(0, 1), run-time scale 1 / sqrt(in), bias initialised to 1)"""

    def __init__(self, in_dim, out_dim, bias_init=0.0):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(out_dim, in_dim))
        self.bias = nn.Parameter(torch.full((out_dim,), float(bias_init)))
        self.scale = 1 / math.sqrt(in_dim)

    def forward(self, input):
        return F.linear(input, self.weight * self.scale, self.bias)


class ModulatedConv2d(nn.Module):
    def __init__(self, in_channel, out_channel, kernel_size, style_dim, demodulate=True, upsample=False, downsample=False,
                 blur_kernel=(1, 3, 3, 1)):
        super().__init__()
        self.kernel_size, self.in_channel, self.out_channel = kernel_size, in_channel, out_channel
        self.upsample, self.downsample, self.demodulate = upsample, downsample, demodulate
        if upsample:
            p = (len(blur_kernel) - 2) - (kernel_size - 1)
            self.blur = Blur(blur_kernel, pad=((p + 1) // 2 + 1, p // 2 + 1), upsample_factor=2)
        if downsample:
            p = (len(blur_kernel) - 2) + (kernel_size - 1)
            self.blur = Blur(blur_kernel, pad=((p + 1) // 2, p // 2))
        self.scale = 1 / math.sqrt(in_channel * kernel_size ** 2)
        self.padding = kernel_size // 2
        self.weight = nn.Parameter(torch.randn(1, out_channel, in_channel, kernel_size, kernel_size))
        self.modulation = _EqualLinear(style_dim, in_channel, bias_init=1)

    def forward(self, input, style):
        b, cin, h, w = input.shape
        k, cout = self.kernel_size, self.out_channel
        weight = self.scale * self.weight * self.modulation(style).view(b, 1, cin, 1, 1)
        if self.demodulate:
            weight = weight * torch.rsqrt(weight.pow(2).sum([2, 3, 4]) + 1e-8).view(b, cout, 1, 1, 1)
        if self.upsample:
            wt = weight.transpose(1, 2).reshape(b * cin, cout, k, k)
            out = F.conv_transpose2d(input.reshape(1, b * cin, h, w), wt, padding=0, stride=2, groups=b)
            return self.blur(out.view(b, cout, out.shape[2], out.shape[3]))
        if self.downsample:
            x = self.blur(input)
            out = F.conv2d(x.reshape(1, b * cin, x.shape[2], x.shape[3]), weight.view(b * cout, cin, k, k), padding=0, stride=2, groups=b)
        else:
            out = F.conv2d(input.reshape(1, b * cin, h, w), weight.view(b * cout, cin, k, k), padding=self.padding, groups=b)
        return out.view(b, cout, out.shape[2], out.shape[3])


def fill_by_name(module: nn.Module, seed: int = 0) -> None:
    """Deterministic values for every parameter and noise buffer as a function of its state-dict KEY (not of construction
    order), so that two implementations with the same keys hold the same numbers without a weights fixture. Filters
    (`*.kernel`) are constants and keep their values."""
    import zlib
    with torch.no_grad():
        for key, t in module.state_dict().items():
            if key.endswith(".kernel"):
                continue
            g = torch.Generator().manual_seed(zlib.crc32(key.encode()) + seed)
            v = torch.randn(t.shape, generator=g)
            if key.endswith("bias") or key.endswith("noise.weight"):
                v = 0.1 * v
            if "modulation.bias" in key:
                v = 1 + v
            t.copy_(v.to(t.dtype))
