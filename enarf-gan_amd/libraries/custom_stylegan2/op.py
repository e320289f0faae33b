"""fused_leaky_relu / FusedLeakyReLU and upfirdn2d on the HIP library.

The reference takes these from `libraries.stylegan2_pytorch.op` (libraries/custom_stylegan2/net.py:12-14), an un-vendored
submodule (rosinality/stylegan2-pytorch; no commit is recorded in the reference checkout) whose CUDA extensions are
`fused_bias_act` and `upfirdn2d`. Same names, argument meaning and defaults here, on `enarf_bias_act` / `enarf_upfirdn2d`
(include/enarf_hip.h). Both ops are differentiable to any order: bias_act's derivative is linear in its argument with the
mask taken from the forward output, upfirdn2d is linear and its adjoint is upfirdn2d with the filter flipped and up / down
swapped - R1 (libraries/gan/loss.py:25-31) differentiates the discriminator twice. Device tensors only.
"""
from __future__ import annotations

import ctypes as C
from typing import Sequence, Tuple

import torch
from torch import nn

from ... import _lib
from ...ops import _dev_f32, _on_tensor_device, _p, _stream


# ------------------------------------------------------------------------------------------------------- bias + act
@_on_tensor_device
def _bias_act_call(x: torch.Tensor, bias, ref, slope: float, gain: float) -> torch.Tensor:
    lib = _lib.load()
    x = _dev_f32(x, "input")
    if x.dim() < 2:
        raise ValueError(f"fused_leaky_relu: input {tuple(x.shape)} needs a channel axis (dim 1)")
    Cn = x.shape[1]
    outer = x.shape[0]
    inner = 1
    for s in x.shape[2:]:
        inner *= s
    b = None
    if bias is not None:
        b = _dev_f32(bias, "bias").reshape(-1)
        if b.numel() != Cn:
            raise ValueError(f"fused_leaky_relu: bias has {b.numel()} entries, input has {Cn} channels")
    r = None if ref is None else _dev_f32(ref, "ref")
    out = torch.empty_like(x)
    if x.numel() == 0:                     # an empty batch has no storage to point at
        return out
    _lib.check(lib.enarf_bias_act(_p(x), _p(b), _p(r), _p(out), outer, Cn, inner, float(slope), float(gain), _stream(x.device)),
               "enarf_bias_act")
    return out


class _BiasActGrad(torch.autograd.Function):
    """g -> g * gain * (out > 0 ? 1 : slope): linear in g, so it is its own derivative"""

    @staticmethod
    def forward(ctx, g, out, slope, gain):
        ctx.save_for_backward(out)
        ctx.slope, ctx.gain = slope, gain
        return _bias_act_call(g, None, out, slope, gain)

    @staticmethod
    def backward(ctx, gg):
        (out,) = ctx.saved_tensors
        return _BiasActGrad.apply(gg.contiguous(), out, ctx.slope, ctx.gain), None, None, None


class _BiasAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bias, slope, gain):
        out = _bias_act_call(x, bias, None, slope, gain)
        ctx.save_for_backward(out)
        ctx.slope, ctx.gain, ctx.has_bias = slope, gain, bias is not None
        return out

    @staticmethod
    def backward(ctx, g):
        (out,) = ctx.saved_tensors
        gx = _BiasActGrad.apply(g.contiguous(), out, ctx.slope, ctx.gain)
        gb = None
        if ctx.has_bias and ctx.needs_input_grad[1]:
            gb = gx.sum(dim=[0] + list(range(2, gx.dim())))
        return gx, gb, None, None


def fused_leaky_relu(input: torch.Tensor, bias=None, negative_slope: float = 0.2, scale: float = 2 ** 0.5) -> torch.Tensor:
    """scale * leaky_relu(input + bias[None, :, None, ...], negative_slope); bias indexes dim 1."""
    return _BiasAct.apply(input, bias, float(negative_slope), float(scale))


class FusedLeakyReLU(nn.Module):
    def __init__(self, channel: int, bias: bool = True, negative_slope: float = 0.2, scale: float = 2 ** 0.5):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(channel)) if bias else None
        self.negative_slope = negative_slope
        self.scale = scale

    def forward(self, input):
        return fused_leaky_relu(input, self.bias, self.negative_slope, self.scale)


# ------------------------------------------------------------------------------------------------------- upfirdn2d
def _host_filter(kernel: torch.Tensor) -> Tuple[Tuple[float, ...], int, int]:
    """the FIR filter as host floats (the library takes it by value); filters are constants of a module, so the copy is
    cached on the tensor"""
    cached = getattr(kernel, "_enarf_host_filter", None)
    if cached is not None and cached[3] == kernel._version:
        return cached[:3]
    k = kernel.detach().to("cpu", torch.float32)
    if k.dim() == 1:
        k = k[None, :] * k[:, None]
    if k.dim() != 2:
        raise ValueError(f"upfirdn2d: filter {tuple(kernel.shape)} must be 1-D or 2-D")
    vals = tuple(float(v) for v in k.reshape(-1).tolist())
    try:
        kernel._enarf_host_filter = (vals, k.shape[0], k.shape[1], kernel._version)
    except Exception:
        pass
    return vals, k.shape[0], k.shape[1]


@_on_tensor_device
def _upfirdn_call(x: torch.Tensor, filt: Tuple[float, ...], kh: int, kw: int, up: int, down: int, pads: Tuple[int, int, int, int]):
    lib = _lib.load()
    x = _dev_f32(x, "input")
    if x.dim() != 4:
        raise ValueError(f"upfirdn2d: input {tuple(x.shape)} must be (N, C, H, W)")
    N, Cn, H, W = x.shape
    px0, px1, py0, py1 = pads
    OH = lib.enarf_upfirdn2d_out_size(H, kh, up, down, py0, py1)
    OW = lib.enarf_upfirdn2d_out_size(W, kw, up, down, px0, px1)
    if OH <= 0 or OW <= 0:
        raise ValueError(f"upfirdn2d: empty output for input {H}x{W}, filter {kh}x{kw}, up {up}, down {down}, pad {pads}")
    out = torch.empty(N, Cn, OH, OW, dtype=torch.float32, device=x.device)
    if out.numel() == 0:
        return out
    karr = (C.c_float * (kh * kw))(*filt)
    _lib.check(lib.enarf_upfirdn2d(_p(x), _p(out), N * Cn, H, W, karr, kh, kw, up, down, px0, px1, py0, py1, _stream(x.device)),
               "enarf_upfirdn2d")
    return out


def adjoint_pads(H: int, W: int, OH: int, OW: int, kh: int, kw: int, up: int, down: int, pads):
    """pads of the adjoint: upfirdn2d(g, flip(k), up = down, down = up, these) maps (OH, OW) back to exactly (H, W)"""
    px0, _, py0, _ = pads
    return (kw - px0 - 1, W * up - OW * down + px0 - up + 1, kh - py0 - 1, H * up - OH * down + py0 - up + 1)


class _UpFirDn2d(torch.autograd.Function):
    """The adjoint of upfirdn2d(filter k, up, down, pad) on an H x W input is upfirdn2d(flip(k), up = down, down = up, pad')
    on the output, with pad' chosen so that it returns exactly H x W; the adjoint's adjoint is the op itself."""

    @staticmethod
    def forward(ctx, x, filt, kh, kw, up, down, pads):
        ctx.cfg = (filt, kh, kw, up, down, pads, x.shape[2], x.shape[3])
        return _upfirdn_call(x, filt, kh, kw, up, down, pads)

    @staticmethod
    def backward(ctx, g):
        filt, kh, kw, up, down, pads, H, W = ctx.cfg
        flipped = tuple(reversed(filt))                                   # both axes of a row-major kh x kw array
        gx = _UpFirDn2d.apply(g.contiguous(), flipped, kh, kw, down, up,
                              adjoint_pads(H, W, g.shape[2], g.shape[3], kh, kw, up, down, pads))
        return gx, None, None, None, None, None, None


def upfirdn2d(input: torch.Tensor, kernel: torch.Tensor, up: int = 1, down: int = 1, pad: Sequence[int] = (0, 0),
              gain: float = 1.0) -> torch.Tensor:
    """input (N, C, H, W); kernel (kh, kw); pad = (pad0, pad1) for both axes or (x0, x1, y0, y1); `gain` scales the filter
    (on the host copy: `kernel * 4` would be a new tensor, and a device-to-host copy, on every call)."""
    pads = tuple(int(p) for p in pad)
    if len(pads) == 2:
        pads = (pads[0], pads[1], pads[0], pads[1])
    if len(pads) != 4:
        raise ValueError(f"upfirdn2d: pad {pad} must have 2 or 4 entries")
    filt, kh, kw = _host_filter(kernel)
    if gain != 1.0:
        filt = tuple(v * float(gain) for v in filt)
    return _UpFirDn2d.apply(input, filt, kh, kw, int(up), int(down), pads)


def make_kernel(k) -> torch.Tensor:
    """separable taps -> normalised 2-D filter (outer product / sum)"""
    k = torch.as_tensor(k, dtype=torch.float32)
    if k.dim() == 1:
        k = k[None, :] * k[:, None]
    return k / k.sum()


class Blur(nn.Module):
    """FIR low-pass with explicit padding; the filter is a buffer named `kernel`, as in the snapshots' state dicts"""

    def __init__(self, kernel, pad, upsample_factor: int = 1):
        super().__init__()
        k = make_kernel(kernel)
        if upsample_factor > 1:
            k = k * (upsample_factor ** 2)
        self.register_buffer("kernel", k)
        self.pad = tuple(pad)

    def forward(self, input):
        return upfirdn2d(input, self.kernel, pad=self.pad)


class Upsample(nn.Module):
    def __init__(self, kernel, factor: int = 2):
        super().__init__()
        self.factor = factor
        k = make_kernel(kernel) * (factor ** 2)
        self.register_buffer("kernel", k)
        p = k.shape[0] - factor
        self.pad = ((p + 1) // 2 + factor - 1, p // 2)

    def forward(self, input):
        return upfirdn2d(input, self.kernel, up=self.factor, down=1, pad=self.pad)


class PixelNorm(nn.Module):
    def forward(self, input):
        return input * torch.rsqrt(torch.mean(input * input, dim=1, keepdim=True) + 1e-8)


__all__ = ["fused_leaky_relu", "FusedLeakyReLU", "upfirdn2d", "make_kernel", "Blur", "Upsample", "PixelNorm"]
