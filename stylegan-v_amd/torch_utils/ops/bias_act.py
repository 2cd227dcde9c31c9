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


def _native_call(x, b, xref, yref, dy, grad, dim, act_idx, alpha, gain, clamp, db=None):
    """One ``sgv_bias_act`` launch on x's current stream; ``None`` marks an absent stream.  ``db`` (fp32, zero-initialised,
    one entry per bias element) additionally receives the per-channel sum of the result (``sgv_bias_act_db``)."""
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
    # The kernel moves 16-byte vectors: a dense view whose storage offset is not 16-byte aligned (x[1:] of an [N, 3] tensor is still
    # "contiguous") is re-materialised here -- the reference op accepts such views, so this one does too.
    realign = lambda t: t.clone(memory_format=torch.preserve_format) if (t is not None and t.data_ptr() % 16 != 0) else t   # noqa: E731
    x, xref, yref, dy = realign(x), realign(xref), realign(yref), realign(dy)
    p = custom_ops.BiasActParams(x.data_ptr(), b.data_ptr() if b is not None else None, xref.data_ptr() if xref is not None else None,
                                 yref.data_ptr() if yref is not None else None, dy.data_ptr() if dy is not None else None, y.data_ptr(),
                                 grad, act_idx, alpha, gain, clamp, x.numel(), b.numel() if b is not None else 0,
                                 x.stride(dim) if b is not None else 1)
    with custom_ops.device_guard(x):
        if db is None:
            custom_ops.check(lib.sgv_bias_act(p, _DTYPE_CODES[x.dtype], custom_ops.raw_stream(x)), lib)
        else:
            custom_ops.check(lib.sgv_bias_act_db(p, db.data_ptr(), db.shape[0], _DTYPE_CODES[x.dtype], custom_ops.raw_stream(x)), lib)
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
        ctx.b_shape = tuple(b.shape) if b is not None else None
        ctx.b_dtype = b.dtype if b is not None else None
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
                if ctx.has_b and ctx.needs_input_grad[1] and _fused_db_ok(dy, ctx.b_shape, dim):
                    dx, db = _BiasActGradDbFn.apply(dy, x, b, y, ctx.cfg, ctx.b_shape[0], ctx.b_dtype)
                else:
                    dx = _BiasActGradFn.apply(dy, x, b, y, ctx.cfg)
        if ctx.has_b and ctx.needs_input_grad[1] and db is None:
            db = dx.sum([i for i in range(dx.ndim) if i != dim])
        return dx, db, None


fused_bias_grad = True   # bias gradient accumulated inside the grad = 1 kernel (one pass over dy less per layer)


def _fused_db_ok(dy, b_shape, dim):
    """The in-kernel bias-gradient sum needs dense NCHW-like storage: a whole 16-byte vector inside one bias element."""
    if not (fused_bias_grad and dy.is_cuda and dy.is_contiguous() and dy.dtype in (torch.float32, torch.float16, torch.bfloat16)):
        return False
    nvec = 16 // dy.element_size()
    return dy.numel() > 0 and dy.stride(dim) % nvec == 0 and dy.numel() % nvec == 0


def _grad_fn_backward(ctx, d_dx):
    """Shared second-order logic of _BiasActGradFn / _BiasActGradDbFn (the structure of bias_act.py:188-206)."""
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
    return d_dy, d_x, d_b


class _BiasActGradDbFn(torch.autograd.Function):
    """(dx, db) = (d(bias_act)/dx * dy, sum of dx over everything but `dim`) in ONE kernel.  Differentiable like _BiasActGradFn:
    db = sum(dx), so an incoming d_db is broadcast onto d_dx."""

    @staticmethod
    def forward(ctx, dy, x, b, y, cfg, nb, b_dtype):
        dim, act, alpha, gain, clamp = cfg
        spec = activation_funcs[act]
        ctx.memory_format = _memory_format_of(dy)
        slots = 64 if dy.numel() >= (1 << 22) else 1   # big tensors: spread the per-channel atomics over 64 partial rows
        from . import amax as _amax
        db32 = _amax.zeros([slots, nb], dy.device) if dy.is_cuda else torch.zeros([slots, nb], dtype=torch.float32, device=dy.device)      # (zero arena: one fill per many buffers)
        bb = b if b is not None else torch.zeros([nb], dtype=dy.dtype, device=dy.device)   # carries size_b / step_b; its values are only read by 'x'-referencing activations
        dx = _native_call(dy, bb, x, y, None, 1, dim, spec.cuda_idx, alpha, gain, clamp, db=db32)
        ctx.cfg = cfg
        ctx.dim = dim
        ctx.dx_meta = (tuple(dx.shape), dx.dtype, dx.device)   # 'linear' + gain/clamp saves no tensor at all: the shape comes from here
        ctx.save_for_backward(dy if spec.has_2nd_grad else None, x, b, y)
        return dx, (db32.sum(0) if slots > 1 else db32[0]).to(b_dtype)

    @staticmethod
    def backward(ctx, d_dx, d_db):
        shape, dtype, device = ctx.dx_meta
        if d_dx is None:
            d_dx = torch.zeros(shape, dtype=dtype, device=device)
        if d_db is not None:
            view = [1] * len(shape)
            view[ctx.dim] = -1
            d_dx = d_dx + d_db.reshape(view).to(d_dx.dtype)
        d_dy, d_x, d_b = _grad_fn_backward(ctx, d_dx)
        return d_dy, d_x, d_b, None, None, None, None


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
        d_dy, d_x, d_b = _grad_fn_backward(ctx, d_dx)
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
