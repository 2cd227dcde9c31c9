"""Bilinear ``grid_sample`` (zeros padding, align_corners=False) that can be differentiated twice w.r.t. its input.

API of the reference's ``src/torch_utils/ops/grid_sample_gradfix.py`` (``grid_sample(input, grid)``, module switch
``enabled``), which the ADA geometric path calls (augment.py:300).  Why it is needed: stock autograd has no derivative for
``aten::grid_sampler_2d_backward``, so the R1 penalty (a gradient of a gradient) through the augmentation pipeline raises;
the reference's own fix switches itself off on torch >= 1.10 (grid_sample_gradfix.py:37).

Formulation used here: for a FIXED grid, sampling is a linear map ``S`` from the input image to the output image, and its
input-gradient is the adjoint map ``S^T`` applied to the output gradient.  One autograd node represents either direction;
differentiating ``S`` gives ``S^T`` and differentiating ``S^T`` gives ``S`` back, so every order of derivative w.r.t. the
input (and w.r.t. incoming gradients) is available from two ATen calls.  The grid itself receives a first-order gradient
(API parity with ``F.grid_sample``); higher-order terms through the grid are not provided -- ADA's grids come from random
transform parameters and never require grad.
"""

import torch

enabled = True   # False -> plain torch.nn.functional.grid_sample (first-order only w.r.t. R1 through ADA)

_MODE_BILINEAR, _PAD_ZEROS, _ALIGN = 0, 0, False


def grid_sample(input, grid):
    if not enabled:
        return torch.nn.functional.grid_sample(input=input, grid=grid, mode='bilinear', padding_mode='zeros', align_corners=False)
    assert input.ndim == 4 and grid.ndim == 4 and grid.shape[-1] == 2
    return _SamplingMap.apply(input, grid, None)


class _SamplingMap(torch.autograd.Function):
    """``geometry is None``: y = S_grid(t), t the image [N,C,H,W].  ``geometry = (N,C,H,W)``: y = S_grid^T(t), t an
    output-shaped gradient [N,C,Ho,Wo] scattered back onto an image of that geometry."""

    @staticmethod
    def forward(ctx, t, grid, geometry):
        ctx.adjoint = geometry is not None
        ctx.image_shape = tuple(geometry) if ctx.adjoint else tuple(t.shape)
        if ctx.adjoint:
            # the input operand of the ATen backward only provides the image geometry when the grid gradient is masked out
            like = t.new_empty(ctx.image_shape)
            out, _ = torch.ops.aten.grid_sampler_2d_backward(t, like, grid, _MODE_BILINEAR, _PAD_ZEROS, _ALIGN, [True, False])
            ctx.save_for_backward(grid, None)
        else:
            out = torch.ops.aten.grid_sampler_2d(t, grid, _MODE_BILINEAR, _PAD_ZEROS, _ALIGN)
            ctx.save_for_backward(grid, t if grid.requires_grad else None)
        return out

    @staticmethod
    def backward(ctx, g):
        grid, image = ctx.saved_tensors
        d_t = d_grid = None
        if ctx.needs_input_grad[0]:
            # d/dt of S is S^T and vice versa; the result is again a _SamplingMap node, hence differentiable
            d_t = _SamplingMap.apply(g, grid, None if ctx.adjoint else ctx.image_shape)
        if ctx.needs_input_grad[1]:
            if ctx.adjoint:
                raise RuntimeError('grid_sample: gradients through the sampling grid are first-order only')
            with torch.no_grad():
                _, d_grid = torch.ops.aten.grid_sampler_2d_backward(g, image, grid, _MODE_BILINEAR, _PAD_ZEROS, _ALIGN, [False, True])
        return d_t, d_grid, None
