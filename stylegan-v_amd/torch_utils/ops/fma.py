"""Fused multiply-add ``a * b + c`` with a broadcast-aware backward.

Boundary name of the reference's ``src/torch_utils/ops/fma.py`` (``fma`` :15): the post-convolution
demodulation + noise step of ``modulated_conv2d`` (networks.py:69).  Forward is one ``addcmul``;
the hand-written backward avoids keeping the full-size product alive and sums broadcast dimensions
back to each operand's shape.  Differentiable to any order (backward uses plain tensor ops).
"""

import torch


def fma(a, b, c):
    return _FMA.apply(a, b, c)


def _sum_to_shape(t, shape):
    """Reduce a broadcast result ``t`` back to ``shape`` (inverse of broadcasting)."""
    lead = t.ndim - len(shape)
    assert lead >= 0
    dims = [d for d in range(t.ndim) if t.shape[d] > 1 and (d < lead or shape[d - lead] == 1)]
    if dims:
        t = t.sum(dim=dims, keepdim=True)
    if lead:
        t = t.reshape(-1, *t.shape[lead + 1:])
    assert t.shape == shape
    return t


class _FMA(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, c):
        ctx.save_for_backward(a, b)
        ctx.c_shape = c.shape
        return torch.addcmul(c, a, b)

    @staticmethod
    def backward(ctx, dout):
        a, b = ctx.saved_tensors
        da = _sum_to_shape(dout * b, a.shape) if ctx.needs_input_grad[0] else None
        db = _sum_to_shape(dout * a, b.shape) if ctx.needs_input_grad[1] else None
        dc = _sum_to_shape(dout, ctx.c_shape) if ctx.needs_input_grad[2] else None
        return da, db, dc
