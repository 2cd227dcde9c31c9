"""Bilinear ``grid_sample`` with a working second-order gradient w.r.t. the input.

Boundary names of the reference's ``src/torch_utils/ops/grid_sample_gradfix.py`` (``grid_sample`` :27,
``enabled`` :23).  The reference's custom op disables itself on torch >= 1.10 (:37) and its ATen
lookup is stale (:64-65), so with the stock ``F.grid_sample`` the R1 penalty through the ADA
geometric path raises "derivative for aten::grid_sampler_2d_backward is not implemented"
(SURVEY.md section 0.9).  This version is always active: backward calls
``torch.ops.aten.grid_sampler_2d_backward`` and the double-backward w.r.t. ``grad_output`` is another
``grid_sample`` of the incoming gradient -- grid_sample is linear in ``input`` -- exactly the
structure of grid_sample_gradfix.py:61-81.  No gradient flows to ``grid`` at second order.
"""

import torch

enabled = True  # kept for API compatibility


def grid_sample(input, grid):
    if enabled:
        return _GridSample2dForward.apply(input, grid)
    return torch.nn.functional.grid_sample(input=input, grid=grid, mode='bilinear', padding_mode='zeros', align_corners=False)


class _GridSample2dForward(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input, grid):
        assert input.ndim == 4 and grid.ndim == 4
        out = torch.nn.functional.grid_sample(input=input, grid=grid, mode='bilinear', padding_mode='zeros', align_corners=False)
        ctx.save_for_backward(input, grid)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        input, grid = ctx.saved_tensors
        grad_input, grad_grid = _GridSample2dBackward.apply(grad_output, input, grid)
        return grad_input, grad_grid


class _GridSample2dBackward(torch.autograd.Function):
    @staticmethod
    def forward(ctx, grad_output, input, grid):
        mask = [ctx.needs_input_grad[1], ctx.needs_input_grad[2]]
        grad_input, grad_grid = torch.ops.aten.grid_sampler_2d_backward(
            grad_output, input, grid, 0, 0, False, [True, True])  # bilinear, zeros, align_corners=False
        del mask
        ctx.save_for_backward(grid)
        return grad_input, grad_grid

    @staticmethod
    def backward(ctx, grad2_grad_input, grad2_grad_grid):
        (grid,) = ctx.saved_tensors
        grad2_grad_output = None
        if ctx.needs_input_grad[0]:
            grad2_grad_output = _GridSample2dForward.apply(grad2_grad_input, grid)
        assert not ctx.needs_input_grad[2], 'second-order gradient w.r.t. the sampling grid is not supported'
        return grad2_grad_output, None, None
