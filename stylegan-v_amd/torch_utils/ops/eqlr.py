"""Equalised learning-rate parameter scaling of a whole module in ONE launch.

The reference's ``Conv2dLayer.forward`` (src/training/layers.py:184-185) evaluates ``self.weight * (self.weight_gain * self.lr_multiplier)`` and
``self.bias * self.lr_multiplier`` per layer and per call, and autograd multiplies each gradient once more: for the discriminator of the FFS-256
configuration that is ~250 element-wise launches per training iteration (three forward passes of 21 layers, each with its backward), every one a
few microseconds of a tensor that is kilobytes long.  ``batched(module)`` scales every recorded parameter of the module's ``Conv2dLayer``s with one
multi-tensor kernel (``sgv_multi_scale_f32``, csrc/multi_tensor.hip; the gradients come back through the same kernel, and the node is itself
differentiable for the R1 double backward); inside the block ``lookup`` hands a layer its scaled parameter.

A layer's factors depend on the ``gain`` argument of the call (a linear un-clamped layer folds it into its weights, ``Conv2dLayer._scaled_parameters``),
so every direct evaluation RECORDS the factors it used (``layer._eqlr_scales``) and the next ``batched`` block prepares exactly those; a layer that asks
for other factors than the prepared ones, or runs outside a block, multiplies directly as before.  CPU tensors and non-fp32 parameters take torch's op.
``SGV_EQLR_BATCH=0`` switches the batching off.
"""
import ctypes
import os
import threading

import torch

from .. import custom_ops

enabled = os.environ.get('SGV_EQLR_BATCH', '1') != '0'
_tls = threading.local()


def _native_ok(t):
    return t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()


def _batchable(t):     # (CPU tensors go through the same node -- torch's product per tensor inside it -- so that the CPU suite covers the bookkeeping)
    return t.dtype == torch.float32 and t.is_contiguous()


def _scale_many(tensors, scales):
    if not tensors:
        return []
    if not all(_native_ok(t) for t in tensors):
        return [t * s for t, s in zip(tensors, scales)]
    outs = [torch.empty_like(t) for t in tensors]
    lib = custom_ops.get_native()
    n = len(tensors)
    src = (ctypes.c_void_p * n)(*[t.data_ptr() for t in tensors])
    dst = (ctypes.c_void_p * n)(*[t.data_ptr() for t in outs])
    sizes = (ctypes.c_int64 * n)(*[t.numel() for t in tensors])
    fac = (ctypes.c_float * n)(*[float(s) for s in scales])
    with custom_ops.device_guard(tensors[0]):
        custom_ops.check(lib.sgv_multi_scale_f32(src, dst, sizes, fac, n, custom_ops.raw_stream(tensors[0])), lib)
    return outs


class _MultiScale(torch.autograd.Function):
    """outs[i] = tensors[i] * scales[i]; linear, so its gradient is the same node applied to the incoming gradients (any order of differentiation)."""

    @staticmethod
    def forward(ctx, scales, *tensors):
        ctx.scales = scales
        return tuple(_scale_many([t.contiguous() for t in tensors], scales))

    @staticmethod
    def backward(ctx, *grads):
        idx = [i for i, g in enumerate(grads) if g is not None and ctx.needs_input_grad[i + 1]]
        out = [None] * len(grads)
        if idx:
            res = _MultiScale.apply(tuple(ctx.scales[i] for i in idx), *[grads[i] for i in idx])
            for i, r in zip(idx, res):
                out[i] = r
        return (None, *out)


def scale_many(tensors, scales):
    """[t * s for t, s in zip(tensors, scales)] as one autograd node (one launch per 64 dense fp32 GPU tensors)."""
    return list(_MultiScale.apply(tuple(float(s) for s in scales), *tensors)) if tensors else []


class batched:
    """``with batched(module):`` -- the recorded equalised-lr products of the module's layers, prepared in one launch; see the module docstring."""

    def __init__(self, module, layer_type):
        self.module, self.layer_type, self.pushed = module, layer_type, False

    def __enter__(self):
        if not enabled:
            return self
        layers = self.module.__dict__.get('_eqlr_layers')
        if layers is None:
            layers = [m for m in self.module.modules() if isinstance(m, self.layer_type)]
            self.module.__dict__['_eqlr_layers'] = layers
        params, scales = [], []
        for layer in layers:
            rec = layer.__dict__.get('_eqlr_scales')
            if rec is None:
                continue
            ws, bs = rec
            if _batchable(layer.weight):
                params.append(layer.weight)
                scales.append(ws)
            if layer.bias is not None and bs != 1.0 and _batchable(layer.bias):
                params.append(layer.bias)
                scales.append(bs)
        table = {}
        if len(params) > 1:
            for p, s, o in zip(params, scales, scale_many(params, scales)):
                table[id(p)] = (s, o)
        stack = _tls.__dict__.setdefault('stack', [])
        stack.append(table)
        self.pushed = True
        return self

    def __exit__(self, *exc):
        if self.pushed:
            _tls.stack.pop()
        return False


def lookup(param, scale):
    """The prepared ``param * scale`` of the innermost ``batched`` block, or None (not prepared, or prepared with another factor)."""
    stack = _tls.__dict__.get('stack')
    if not stack:
        return None
    hit = stack[-1].get(id(param))
    return hit[1] if hit is not None and hit[0] == scale else None
