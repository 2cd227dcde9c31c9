"""Fused bias + activation + gain + clamp.

Host-side mirror of the reference op ``src/torch_utils/ops/bias_act.py``: ``bias_act`` (:55) and the
``activation_funcs`` table (:23-33, read by the modules for default gains).  GPU tensors run
``csrc/bias_act.hip`` through the C ABI ``sgv_bias_act`` (include/sgv_ops.h), the replacement of
``_plugin.bias_act`` (bias_act.cpp:32).  First and second order gradients re-use the same kernel in
its ``grad=1`` / ``grad=2`` forms (bias_act.cu:51-142), the structure of bias_act.py:145-206.

Dispatch: ``impl='cuda'`` on a GPU tensor -> native kernel, failure to load it is an error.
CPU tensors or ``impl='ref'`` -> plain PyTorch (bias_act.py:94-123 behaviour).
"""

from types import SimpleNamespace

import numpy as np
import torch
import torch.nn.functional as F

from .. import custom_ops
from .upfirdn2d import _DTYPE_CODES


def _spec(func, def_alpha, def_gain, cuda_idx, ref, has_2nd_grad):
    return SimpleNamespace(func=func, def_alpha=def_alpha, def_gain=def_gain, cuda_idx=cuda_idx, ref=ref, has_2nd_grad=has_2nd_grad)


# name -> func / default alpha / default gain / native activation index / which tensor the gradient
# kernel needs ('x', 'y' or '') / whether a second derivative exists.  Values as bias_act.py:23-33.
activation_funcs = {
    'linear':   _spec(lambda x, **_: x,                               0,   1,          1, '',  False),
    'relu':     _spec(lambda x, **_: F.relu(x),                       0,   np.sqrt(2), 2, 'y', False),
    'lrelu':    _spec(lambda x, alpha, **_: F.leaky_relu(x, alpha),   0.2, np.sqrt(2), 3, 'y', False),
    'tanh':     _spec(lambda x, **_: torch.tanh(x),                   0,   1,          4, 'y', True),
    'sigmoid':  _spec(lambda x, **_: torch.sigmoid(x),                0,   1,          5, 'y', True),
    'elu':      _spec(lambda x, **_: F.elu(x),                        0,   1,          6, 'y', True),
    'selu':     _spec(lambda x, **_: F.selu(x),                       0,   1,          7, 'y', True),
    'softplus': _spec(lambda x, **_: F.softplus(x),                   0,   1,          8, 'y', True),
    'swish':    _spec(lambda x, **_: torch.sigmoid(x) * x,            0,   np.sqrt(2), 9, 'x', True),
}


def _resolve(act, alpha, gain, clamp):
    assert clamp is None or clamp >= 0
    spec = activation_funcs[act]
    alpha = float(spec.def_alpha if alpha is None else alpha)
    gain = float(spec.def_gain if gain is None else gain)
    clamp = float(-1 if clamp is None else clamp)
    return spec, alpha, gain, clamp


# ----------------------------------------------------------------------------------------------
# Plain-PyTorch path.


def _bias_act_ref(x, b=None, dim=1, act='linear', alpha=None, gain=None, clamp=None):
    """bias -> activation -> gain -> clamp with standard PyTorch ops (order of bias_act.py:104-122)."""
    assert isinstance(x, torch.Tensor)
    spec, alpha, gain, clamp = _resolve(act, alpha, gain, clamp)
    if b is not None:
        assert isinstance(b, torch.Tensor) and b.ndim == 1
        assert 0 <= dim < x.ndim
        assert b.shape[0] == x.shape[dim]
        view = [1] * x.ndim
        view[dim] = -1
        x = x + b.reshape(view)
    x = spec.func(x, alpha=alpha)
    if gain != 1:
        x = x * gain
    if clamp >= 0:
        x = x.clamp(-clamp, clamp)
    return x


# ----------------------------------------------------------------------------------------------
# Native path.


def _dense_like(t, memory_format):
    return t.contiguous(memory_format=memory_format)


def _same_layout(a, b):
    if a.ndim != b.ndim:
        return False
    return all(sa == sb and (sa < 2 or ta == tb) for sa, sb, ta, tb in zip(a.shape, b.shape, a.stride(), b.stride()))


def _native_call(x, b, xref, yref, dy, grad, dim, act_idx, alpha, gain, clamp):
    """One ``sgv_bias_act`` launch on x's current stream; ``None`` marks an absent stream."""
    lib = custom_ops.get_native()
    if x.dtype not in _DTYPE_CODES:
        raise RuntimeError(f'bias_act: unsupported dtype {x.dtype}')
    if x.numel() > 2 ** 31 - 1:
        raise RuntimeError('x is too large')
    # Same layout rules as bias_act.cpp:46-51.
    dense = x.is_contiguous() or (x.ndim == 4 and x.is_contiguous(memory_format=torch.channels_last))
    if not dense:
        raise RuntimeError('x must be non-overlapping and dense')
    for name, t in (('xref', xref), ('yref', yref), ('dy', dy)):
        if t is not None:
            if t.shape != x.shape or t.dtype != x.dtype or t.device != x.device:
                raise RuntimeError(f'{name} must have the same shape, dtype, and device as x')
            if not _same_layout(t, x):
                raise RuntimeError(f'{name} must have the same layout as x')
    if b is not None:
        if b.dtype != x.dtype or b.device != x.device:
            raise RuntimeError('b must have the same dtype and device as x')
        if b.ndim != 1:
            raise RuntimeError('b must have rank 1')
        if not (0 <= dim < x.ndim):
            raise RuntimeError('dim is out of bounds')
        if b.numel() != x.shape[dim]:
            raise RuntimeError('b has wrong number of elements')
        if not b.is_contiguous():
            raise RuntimeError('b must be contiguous')
    y = torch.empty_like(x)
    if x.numel() == 0:
        return y
    p = custom_ops.BiasActParams(x.data_ptr(), b.data_ptr() if b is not None else None, xref.data_ptr() if xref is not None else None,
                                 yref.data_ptr() if yref is not None else None, dy.data_ptr() if dy is not None else None, y.data_ptr(),
                                 grad, act_idx, alpha, gain, clamp, x.numel(), b.numel() if b is not None else 0,
                                 x.stride(dim) if b is not None else 1)
    with custom_ops.device_guard(x):
        custom_ops.check(lib.sgv_bias_act(p, _DTYPE_CODES[x.dtype], custom_ops.raw_stream(x)), lib)
    return y


def _memory_format_of(t):
    return torch.channels_last if t.ndim == 4 and t.stride(1) == 1 and t.shape[1] > 1 else torch.contiguous_format


class _BiasActFn(torch.autograd.Function):
    """y = bias_act(x, b).  cfg = (dim, act, alpha, gain, clamp) with resolved floats."""

    @staticmethod
    def forward(ctx, x, b, cfg):
        dim, act, alpha, gain, clamp = cfg
        spec = activation_funcs[act]
        ctx.memory_format = _memory_format_of(x)
        x = _dense_like(x, ctx.memory_format)
        b = b.contiguous() if b is not None else None
        y = x
        if act != 'linear' or gain != 1 or clamp >= 0 or b is not None:
            y = _native_call(x, b, None, None, None, 0, dim, spec.cuda_idx, alpha, gain, clamp)
        keep_x = 'x' in spec.ref or spec.has_2nd_grad
        ctx.cfg = cfg
        ctx.has_b = b is not None
        ctx.save_for_backward(x if keep_x else None, b if keep_x else None, y if 'y' in spec.ref else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        dim, act, alpha, gain, clamp = ctx.cfg
        x, b, y = ctx.saved_tensors
        dy = _dense_like(dy, ctx.memory_format)
        dx = db = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            dx = dy
            if act != 'linear' or gain != 1 or clamp >= 0:
                dx = _BiasActGradFn.apply(dy, x, b, y, ctx.cfg)
        if ctx.has_b and ctx.needs_input_grad[1]:
            db = dx.sum([i for i in range(dx.ndim) if i != dim])
        return dx, db, None


class _BiasActGradFn(torch.autograd.Function):
    """dx = d(bias_act)/dx * dy, itself differentiable (w.r.t. dy always; w.r.t. x, b where a 2nd derivative exists)."""

    @staticmethod
    def forward(ctx, dy, x, b, y, cfg):
        dim, act, alpha, gain, clamp = cfg
        spec = activation_funcs[act]
        ctx.memory_format = _memory_format_of(dy)
        dx = _native_call(dy, b, x, y, None, 1, dim, spec.cuda_idx, alpha, gain, clamp)
        ctx.cfg = cfg
        ctx.save_for_backward(dy if spec.has_2nd_grad else None, x, b, y)
        return dx

    @staticmethod
    def backward(ctx, d_dx):
        dim, act, alpha, gain, clamp = ctx.cfg
        spec = activation_funcs[act]
        d_dx = _dense_like(d_dx, ctx.memory_format)
        dy, x, b, y = ctx.saved_tensors
        d_dy = d_x = d_b = None
        if ctx.needs_input_grad[0]:
            d_dy = _BiasActGradFn.apply(d_dx, x, b, y, ctx.cfg)
        if spec.has_2nd_grad and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]):
            d_x = _native_call(d_dx, b, x, y, dy, 2, dim, spec.cuda_idx, alpha, gain, clamp)
        if spec.has_2nd_grad and b is not None and ctx.needs_input_grad[2]:
            d_b = d_x.sum([i for i in range(d_x.ndim) if i != dim])
        return d_dy, d_x, d_b, None, None


def bias_act(x, b=None, dim=1, act='linear', alpha=None, gain=None, clamp=None, impl='cuda'):
    """Add bias ``b`` along ``dim``, apply ``act``, scale by ``gain``, clamp to ``[-clamp, clamp]``
    (contract of bias_act.py:55-89; each step optional; ``alpha``/``gain`` default per activation).
    Supports first and second order gradients."""
    assert isinstance(x, torch.Tensor)
    assert impl in ('ref', 'cuda')
    if impl == 'cuda' and x.device.type == 'cuda':
        _, alpha_f, gain_f, clamp_f = _resolve(act, alpha, gain, clamp)
        return _BiasActFn.apply(x, b, (dim, act, alpha_f, gain_f, clamp_f))
    return _bias_act_ref(x=x, b=b, dim=dim, act=act, alpha=alpha, gain=gain, clamp=clamp)
