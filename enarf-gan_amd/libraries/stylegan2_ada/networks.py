"""The tri-plane producer of the GAN path: the StyleGAN2-ADA generator that
`libraries/triplane/triplane_nerf.py:17-29` (`prepare_triplane_generator`) builds from an un-vendored submodule -
`training.networks.Generator` of NVlabs/stylegan2-ada-pytorch (`.gitmodules:1-3`; the directory is empty in the reference
checkout and no commit is recorded) with mapping depth 8, `channel_base` 32768, `channel_max` 512, 256 x 256 output of
(32 + P) * 3 channels, `use_noise=False`, fp32 throughout, no clamping, the default "skip" architecture.

Restated from the published architecture (Karras et al., "Analyzing and Improving the Image Quality of StyleGAN", CVPR 2020;
"Training Generative Adversarial Networks with Limited Data", NeurIPS 2020) on this repo's HIP ops - that code base's two
CUDA plugins are exactly `bias_act` and `upfirdn2d` (libraries/custom_stylegan2/op.py: `enarf_bias_act`, `enarf_upfirdn2d`) -
with torch's library convolutions. Module and parameter names follow the published implementation (`mapping.fc{i}`,
`mapping.embed`, `mapping.w_avg`, `synthesis.b{res}.{const, conv0, conv1, torgb}.{weight, bias, affine.*}`,
`resample_filter` buffers), which is what a snapshot of the reference's generator holds under `nerf.tri_plane_gen.*`.
PARITY UNPINNED: there is neither source nor fixture of this network in the reference; the tests check it against an
independent plain-PyTorch composition of the same published equations.

Only what `prepare_triplane_generator` configures is built: no noise inputs, no fp16 blocks, no `conv_clamp`, "skip"
architecture, `lrelu` activations.
"""
from __future__ import annotations

import math
import torch
from torch import nn
from torch.nn import functional as F

from ..custom_stylegan2.op import fused_leaky_relu, make_kernel, upfirdn2d

SQRT2 = math.sqrt(2.0)


def normalize_2nd_moment(x: torch.Tensor, dim: int = 1, eps: float = 1e-8) -> torch.Tensor:
    return x * (x.square().mean(dim=dim, keepdim=True) + eps).rsqrt()


class FullyConnectedLayer(nn.Module):
    """y = act(x W^T * (lr / sqrt(in)) + b * lr); weights are stored divided by the learning-rate multiplier"""

    def __init__(self, in_features, out_features, bias=True, activation="linear", lr_multiplier=1.0, bias_init=0.0):
        super().__init__()
        self.activation = activation
        self.weight = nn.Parameter(torch.randn(out_features, in_features) / lr_multiplier)
        self.bias = nn.Parameter(torch.full((out_features,), float(bias_init))) if bias else None
        self.weight_gain = lr_multiplier / math.sqrt(in_features)
        self.bias_gain = lr_multiplier

    def forward(self, x):
        w = self.weight * self.weight_gain
        b = None if self.bias is None else self.bias * self.bias_gain
        if self.activation == "linear":
            return F.linear(x, w, b)
        if self.activation != "lrelu":
            raise NotImplementedError(f"activation {self.activation!r}: the tri-plane generator uses 'linear' and 'lrelu' only")
        return fused_leaky_relu(F.linear(x, w), b, 0.2, SQRT2)


class MappingNetwork(nn.Module):
    """z (and the conditioning vector c: the encoded bone lengths, models/narf.py:80-83) -> num_ws copies of w"""

    def __init__(self, z_dim, c_dim, w_dim, num_ws, num_layers=8, embed_features=None, layer_features=None, activation="lrelu",
                 lr_multiplier=0.01, w_avg_beta=0.995):
        super().__init__()
        self.z_dim, self.c_dim, self.w_dim, self.num_ws, self.num_layers, self.w_avg_beta = z_dim, c_dim, w_dim, num_ws, num_layers, w_avg_beta
        embed_features = (w_dim if embed_features is None else embed_features) if c_dim > 0 else 0
        layer_features = w_dim if layer_features is None else layer_features
        widths = [z_dim + embed_features] + [layer_features] * (num_layers - 1) + [w_dim]
        if c_dim > 0:
            self.embed = FullyConnectedLayer(c_dim, embed_features)
        for i in range(num_layers):
            setattr(self, f"fc{i}", FullyConnectedLayer(widths[i], widths[i + 1], activation=activation, lr_multiplier=lr_multiplier))
        if num_ws is not None and w_avg_beta is not None:
            self.register_buffer("w_avg", torch.zeros(w_dim))

    def forward(self, z, c, truncation_psi=1, truncation_cutoff=None, skip_w_avg_update=False):
        x = None
        if self.z_dim > 0:
            x = normalize_2nd_moment(z.to(torch.float32))
        if self.c_dim > 0:
            y = normalize_2nd_moment(self.embed(c.to(torch.float32)))
            x = torch.cat([x, y], dim=1) if x is not None else y
        for i in range(self.num_layers):
            x = getattr(self, f"fc{i}")(x)
        if self.w_avg_beta is not None and self.training and not skip_w_avg_update:
            self.w_avg.copy_(x.detach().mean(dim=0).lerp(self.w_avg, self.w_avg_beta))
        if self.num_ws is not None:
            x = x.unsqueeze(1).repeat(1, self.num_ws, 1)
        if truncation_psi != 1:
            if self.num_ws is None or truncation_cutoff is None:
                x = self.w_avg.lerp(x, truncation_psi)
            else:
                x[:, :truncation_cutoff] = self.w_avg.lerp(x[:, :truncation_cutoff], truncation_psi)
        return x


def modulated_conv2d(x, weight, styles, up=1, padding=0, resample_filter=None, demodulate=True):
    """StyleGAN2's modulated convolution in its activation form: x * s -> convolution with the shared weights -> * d, with
    d_{b,o} = rsqrt(sum_{i,k} (w_{o,i,k} s_{b,i})^2 + 1e-8). up = 2: a stride-2 transposed convolution followed by the
    4 x 4 low-pass at gain 4 (pads 1, 1), the published "up-sampling convolution"; 2 H x 2 W out."""
    b, cin = x.shape[0], x.shape[1]
    x = x * styles.view(b, cin, 1, 1)
    if up == 1:
        x = F.conv2d(x, weight, padding=padding)
    else:
        k = weight.shape[-1]
        x = F.conv_transpose2d(x, weight.transpose(0, 1), stride=up, padding=0)
        # pads of the low-pass after a transposed convolution with a k-tap kernel (k = 3: 1 and 1)
        fw = resample_filter.shape[-1]
        p0 = padding + (fw + up - 1) // 2 - (k - 1)
        p1 = padding + (fw - up) // 2 - (k - up)
        x = upfirdn2d(x, resample_filter, pad=(p0, p1), gain=up ** 2)
    if demodulate:
        d = torch.rsqrt(F.linear(styles.square(), weight.square().sum([2, 3])) + 1e-8)       # (b, out)
        x = x * d.view(b, -1, 1, 1)
    return x


class SynthesisLayer(nn.Module):
    def __init__(self, in_channels, out_channels, w_dim, resolution, kernel_size=3, up=1, activation="lrelu",
                 resample_filter=(1, 3, 3, 1)):
        super().__init__()
        self.resolution, self.up, self.activation = resolution, up, activation
        self.register_buffer("resample_filter", make_kernel(resample_filter))
        self.padding = kernel_size // 2
        self.affine = FullyConnectedLayer(w_dim, in_channels, bias_init=1)
        self.weight = nn.Parameter(torch.randn(out_channels, in_channels, kernel_size, kernel_size))
        self.bias = nn.Parameter(torch.zeros(out_channels))

    def forward(self, x, w, gain=1):
        styles = self.affine(w)
        x = modulated_conv2d(x, self.weight, styles, up=self.up, padding=self.padding, resample_filter=self.resample_filter)
        return fused_leaky_relu(x, self.bias, 0.2, SQRT2 * gain)


class ToRGBLayer(nn.Module):
    def __init__(self, in_channels, out_channels, w_dim, kernel_size=1):
        super().__init__()
        self.affine = FullyConnectedLayer(w_dim, in_channels, bias_init=1)
        self.weight = nn.Parameter(torch.randn(out_channels, in_channels, kernel_size, kernel_size))
        self.bias = nn.Parameter(torch.zeros(out_channels))
        self.weight_gain = 1.0 / math.sqrt(in_channels * kernel_size ** 2)

    def forward(self, x, w):
        styles = self.affine(w) * self.weight_gain
        x = modulated_conv2d(x, self.weight, styles, demodulate=False)
        return x + self.bias.view(1, -1, 1, 1)


class SynthesisBlock(nn.Module):
    """one resolution of the "skip" architecture: [up-sampling conv0,] conv1, and a ToRGB whose output is added to the
    up-sampled image of the previous resolution"""

    def __init__(self, in_channels, out_channels, w_dim, resolution, img_channels, is_last, resample_filter=(1, 3, 3, 1)):
        super().__init__()
        self.in_channels, self.resolution, self.is_last = in_channels, resolution, is_last
        self.register_buffer("resample_filter", make_kernel(resample_filter))
        self.num_conv, self.num_torgb = 0, 0
        if in_channels == 0:
            self.const = nn.Parameter(torch.randn(out_channels, resolution, resolution))
        else:
            self.conv0 = SynthesisLayer(in_channels, out_channels, w_dim, resolution, up=2, resample_filter=resample_filter)
            self.num_conv += 1
        self.conv1 = SynthesisLayer(out_channels, out_channels, w_dim, resolution)
        self.num_conv += 1
        self.torgb = ToRGBLayer(out_channels, img_channels, w_dim)
        self.num_torgb += 1

    def forward(self, x, img, ws):
        w = iter(ws.unbind(dim=1))
        if self.in_channels == 0:
            x = self.const.unsqueeze(0).expand(ws.shape[0], -1, -1, -1)
        else:
            x = self.conv0(x, next(w))
        x = self.conv1(x, next(w))
        if img is not None:                                            # 2x up-sampling of the running image (pads 2, 1; gain 4)
            img = upfirdn2d(img, self.resample_filter, up=2, pad=(2, 1), gain=4)
        y = self.torgb(x, next(w))
        img = y if img is None else img + y
        return x, img


class SynthesisNetwork(nn.Module):
    def __init__(self, w_dim, img_resolution, img_channels, channel_base=32768, channel_max=512):
        super().__init__()
        assert img_resolution >= 4 and img_resolution & (img_resolution - 1) == 0
        self.w_dim, self.img_resolution, self.img_channels = w_dim, img_resolution, img_channels
        self.block_resolutions = [2 ** i for i in range(2, int(math.log2(img_resolution)) + 1)]
        ch = {res: min(channel_base // res, channel_max) for res in self.block_resolutions}
        self.num_ws = 0
        for res in self.block_resolutions:
            block = SynthesisBlock(ch[res // 2] if res > 4 else 0, ch[res], w_dim, res, img_channels, is_last=res == img_resolution)
            self.num_ws += block.num_conv + (block.num_torgb if res == img_resolution else 0)
            setattr(self, f"b{res}", block)

    def forward(self, ws):
        x = img = None
        i = 0
        for res in self.block_resolutions:
            block = getattr(self, f"b{res}")
            x, img = block(x, img, ws.narrow(1, i, block.num_conv + block.num_torgb))
            i += block.num_conv                                        # the ToRGB's w is the next block's first
        return img


class Generator(nn.Module):
    def __init__(self, z_dim, c_dim, w_dim, img_resolution, img_channels, mapping_kwargs=None, synthesis_kwargs=None):
        super().__init__()
        self.z_dim, self.c_dim, self.w_dim, self.img_resolution, self.img_channels = z_dim, c_dim, w_dim, img_resolution, img_channels
        synthesis_kwargs = dict(synthesis_kwargs or {})
        if synthesis_kwargs.pop("use_noise", False) or synthesis_kwargs.pop("num_fp16_res", 0) or synthesis_kwargs.pop("conv_clamp", None):
            raise NotImplementedError("noise inputs, fp16 blocks and conv_clamp are not part of the tri-plane generator's configuration")
        self.synthesis = SynthesisNetwork(w_dim, img_resolution, img_channels, **synthesis_kwargs)
        self.num_ws = self.synthesis.num_ws
        self.mapping = MappingNetwork(z_dim, c_dim, w_dim, self.num_ws, **dict(mapping_kwargs or {}))

    def forward(self, z, c=None, truncation_psi=1, truncation_cutoff=None, **_unused):
        return self.synthesis(self.mapping(z, c, truncation_psi=truncation_psi, truncation_cutoff=truncation_cutoff))


def prepare_triplane_generator(z_dim, w_dim, out_channels, c_dim=0) -> Generator:
    """libraries/triplane/triplane_nerf.py:17-29: 8 mapping layers, channel_base 32768, channel_max 512, no noise, no fp16
    blocks, no clamping; 256 x 256 planes of `out_channels`"""
    return Generator(z_dim=z_dim, c_dim=c_dim, w_dim=w_dim, img_resolution=256, img_channels=out_channels,
                     mapping_kwargs=dict(num_layers=8), synthesis_kwargs=dict(channel_base=32768, channel_max=512))

