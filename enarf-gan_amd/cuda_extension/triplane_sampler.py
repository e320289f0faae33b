"""Drop-in for the reference's `cuda_extension/triplane_sampler.py` (its :7-16, :19-68).

Same names and argument meaning: GRID_SAMPLE_* enums, TriplaneSamplerFunction, triplane_sampler(input, grid,
mode, padding_mode, align_corners). The arithmetic runs in libenarf_hip.so (enarf_triplane_sample_fwd/bwd).
Unlike the reference wrapper, whose `output_mask` handling is inverted and drops exactly the gradients that
are required (triplane_sampler.py:59-62), backward returns the true gradients.
`triplane_sampler_cuda` below exposes the two functions of the reference's pybind module
(TriplaneSampler.cpp:55-59) for code that calls them directly.
"""
import types

import torch

from .. import ops

GRID_SAMPLE_INTERPOLATION_MODES = {"bilinear": 0, "nearest": 1}
GRID_SAMPLE_PADDING_MODES = {"zeros": 0, "border": 1, "reflection": 2}


def _forward(input, grid, interpolation_mode, padding_mode, align_corners):
    # the reference dispatches float / double / half and returns `input.new_zeros(...)`, i.e. the INPUT's dtype
    # (TriplaneSampler_kernel.cu:244, TriplaneSampler.cpp:20); the HIP kernels compute in fp32
    out = ops.triplane_sample_fwd(input, grid, int(interpolation_mode), int(padding_mode), bool(align_corners))
    return out.to(input.dtype)


def _backward(grad_output, input, grid, interpolation_mode, padding_mode, align_corners, output_mask):
    gi, gg = ops.triplane_sample_bwd(grad_output, input, grid, int(interpolation_mode), int(padding_mode),
                                     bool(align_corners), bool(output_mask[0]), bool(output_mask[1]))
    # the pybind module returns placeholder tensors where a gradient is not needed (TriplaneSampler.cpp:37,:44)
    gi = input.new_zeros((1, 1, 1, 1)) if gi is None else gi.to(input.dtype)
    gg = torch.zeros((0, 0, 0, 0)) if gg is None else gg.to(grid.dtype)
    return gi, gg


triplane_sampler_cuda = types.SimpleNamespace(triplane_sampler_forward=_forward, triplane_sampler_backward=_backward)


class TriplaneSamplerFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input, grid, mode="bilinear", padding_mode="zeros", align_corners=False,
                output_mask=(True, True)):
        """input (B, 3C, H, W), grid (B, h, w, 3) -> (B, C, h, w): sum over planes xy, yz, zx of grid_sample."""
        mode_enum = GRID_SAMPLE_INTERPOLATION_MODES[mode]
        padding_mode_enum = GRID_SAMPLE_PADDING_MODES[padding_mode]
        sampled = _forward(input, grid, mode_enum, padding_mode_enum, align_corners)
        ctx.save_for_backward(input, grid)
        ctx.mode_enum, ctx.padding_mode_enum, ctx.align_corners = mode_enum, padding_mode_enum, align_corners
        ctx.output_mask = output_mask
        return sampled

    @staticmethod
    def backward(ctx, grad_output):
        input, grid = ctx.saved_tensors
        need_i = bool(ctx.output_mask[0]) and ctx.needs_input_grad[0]
        need_g = bool(ctx.output_mask[1]) and ctx.needs_input_grad[1]
        gi, gg = ops.triplane_sample_bwd(grad_output, input, grid, ctx.mode_enum, ctx.padding_mode_enum,
                                         ctx.align_corners, need_i, need_g)
        gi = None if gi is None else gi.to(input.dtype)
        gg = None if gg is None else gg.to(grid.dtype)
        return gi, gg, None, None, None, None


def triplane_sampler(input, grid, mode="bilinear", padding_mode="zeros", align_corners=False):
    output_mask = (input.requires_grad, grid.requires_grad)
    return TriplaneSamplerFunction.apply(input, grid, mode, padding_mode, align_corners, output_mask)
