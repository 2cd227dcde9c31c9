"""Fused "FIR -> per-sample channel scale -> bias -> activation -> clamp" epilogue of an up-sampling synthesis layer.

The reference has no such op (there is no ``filtered_lrelu`` in this StyleGAN2-ADA-vintage code
base).  After every ``up=2`` modulated convolution it runs three full passes over the activation:
``upfirdn2d`` (conv2d_resample.py:138-139) -> ``x * dcoefs`` (networks.py:70-71) -> ``bias_act`` (networks.py:141-143).
``fir_bias_act`` computes exactly that composition; on the GPU it is ONE kernel forward
(``sgv_upfirdn2d_fused`` mode 1) and ONE kernel backward (mode 2: activation derivative and clamp mask applied while the
gradient rows are loaded, transposed FIR, and the per-plane sums that give the bias and scale gradients accumulated on
the way).  For fp32 the forward result is bit-identical to the three-op composition (same fp32 operations in the same
order).  Anything the fused kernel does not cover -- CPU tensors, other paddings, double backward -- runs the
composition of the three ops, which is the definition of the result.
"""

import contextlib
import os

import torch

from .. import custom_ops
from . import amax as _amax
from . import bias_act as _ba
from . import fused_conv_act as _fca
from . import modulation as _mod
from . import upfirdn2d as _ufd
from .upfirdn2d import _DTYPE_CODES

enabled = True  # module switch.  Second order: passes known to be differentiated twice (path-length regularisation, R1; training/loss.py) run under
                # ``fused_conv_act.composition_only()`` and take the composition.
# A fused node whose gradient is differentiated WITHOUT that announcement (``create_graph=True``) can switch its backward to the composition on the saved
# inputs -- but only if the forward kept its input, the (2H+3)^2 output of the up-sampling convolution, alive: one extra largest-resolution activation
# per up-layer for every ordinary training step.  So that is opt-in (`second_order_support()` / SGV_FIR_FUSED_SECOND_ORDER=1); without it the
# node is once-differentiable and says so.
keep_inputs_for_second_order = os.environ.get('SGV_FIR_FUSED_SECOND_ORDER', '0') == '1'


@contextlib.contextmanager
def second_order_support(on=True):
    """Fused nodes created inside keep their inputs so that `create_graph=True` differentiates the composition instead of raising."""
    global keep_inputs_for_second_order
    prev, keep_inputs_for_second_order = keep_inputs_for_second_order, bool(on)
    try:
        yield
    finally:
        keep_inputs_for_second_order = prev


def fir_bias_act_composed(x, f, scale=None, bias=None, padding=1, fir_gain=1, act='lrelu', alpha=None, gain=None, clamp=None, flip_filter=False):
    """The definition: three ops, differentiable to any order."""
    y = _ufd.upfirdn2d(x, f, padding=padding, gain=fir_gain, flip_filter=flip_filter)
    if scale is not None:
        y = _mod.scale_channels(y, scale)
    return _ba.bias_act(y, bias.to(y.dtype) if bias is not None else None, act=act, alpha=alpha, gain=gain, clamp=clamp)


def _ufd_params(x, f, y, pads, flip, gain, up=1, down=1):
    n, c, h, w = x.shape
    xs, fs, ys = x.stride(), f.stride(), y.stride()
    return custom_ops.Upfirdn2dParams(x.data_ptr(), f.data_ptr(), y.data_ptr(), up, up, down, down, pads[0], pads[1], pads[2], pads[3],
                                      int(bool(flip)), float(gain), w, h, c, n, xs[3], xs[2], xs[1], xs[0], f.shape[1], f.shape[0], fs[1], fs[0],
                                      y.shape[3], y.shape[2], ys[3], ys[2], ys[1], ys[0])


class _FusedFirBiasActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, f, scale, bias, cfg):
        pads, fir_gain, flip, act, alpha, gain, clamp = cfg
        lib = custom_ops.get_native()
        x_in = x
        x = x.contiguous()
        n, c, h, w = x.shape
        fh, fw = f.shape
        oh = _ufd.output_size(h, 1, 1, pads[2], pads[3], fh)
        ow = _ufd.output_size(w, 1, 1, pads[0], pads[1], fw)
        y = torch.empty([n, c, oh, ow], dtype=x.dtype, device=x.device)
        sc = scale.reshape(-1).contiguous().float() if scale is not None else None
        bi = bias.contiguous().float() if bias is not None else None
        e = custom_ops.FirEpilogue(1, sc.data_ptr() if sc is not None else None, bi.data_ptr() if bi is not None else None, None, None, None,
                                   _ba.activation_funcs[act].cuda_idx, alpha, gain, clamp)
        with custom_ops.device_guard(x):
            # (y feeds the layer's next convolution: the kernel leaves its magnitude bound behind)
            custom_ops.check(_amax.launch_tracking(y, lambda: lib.sgv_upfirdn2d_fused(_ufd_params(x, f, y, pads, flip, fir_gain), e, _DTYPE_CODES[x.dtype], custom_ops.raw_stream(x))), lib)
        ctx.cfg = cfg
        ctx.x_shape = x.shape
        ctx.has_scale, ctx.has_bias = scale is not None, bias is not None
        ctx.scale_shape = scale.shape if scale is not None else None
        # x, scale and bias themselves (with their history) serve the create_graph path of backward -- kept only on request (see above)
        ctx.keeps_inputs = keep_inputs_for_second_order
        if ctx.keeps_inputs:
            ctx.save_for_backward(y, f, sc, bi, x_in, scale, bias)
        else:
            ctx.save_for_backward(y, f, sc, bi)
        return y

    @staticmethod
    def backward(ctx, dy):
        pads, fir_gain, flip, act, alpha, gain, clamp = ctx.cfg
        y, f, sc, bi = ctx.saved_tensors[:4]
        if torch.is_grad_enabled():
            if not ctx.keeps_inputs:
                raise RuntimeError('fir_bias_act: this fused node is differentiated twice (create_graph=True) but did not keep its inputs; run the pass under '
                                   'fused_conv_act.composition_only() (as the R1 / path-length passes of training/loss.py do) or create the node under '
                                   'fused_fir_act.second_order_support() / SGV_FIR_FUSED_SECOND_ORDER=1')
            x_in, scale, bias = ctx.saved_tensors[4:]
            # create_graph=True (this gradient is differentiated again, e.g. a path-length or R1 pass that did not announce itself through
            # `composition_only()`): differentiate the composition on the saved inputs instead -- one extra forward, gradients of any order.
            ins = [t for t, need in zip((x_in, scale, bias), (ctx.needs_input_grad[0], ctx.needs_input_grad[2], ctx.needs_input_grad[3])) if need and t is not None]
            with torch.enable_grad():
                y2 = fir_bias_act_composed(x_in, f, scale=scale, bias=bias, padding=list(pads), fir_gain=fir_gain, act=act, alpha=alpha, gain=gain,
                                           clamp=(clamp if clamp >= 0 else None), flip_filter=flip)
                grads = iter(torch.autograd.grad(y2, ins, dy, create_graph=True, allow_unused=True))
            g_x = next(grads) if (ctx.needs_input_grad[0]) else None
            g_s = next(grads) if (ctx.needs_input_grad[2] and scale is not None) else None
            g_b = next(grads) if (ctx.needs_input_grad[3] and bias is not None) else None
            return g_x, None, g_s, g_b, None
        lib = custom_ops.get_native()
        dy = dy.contiguous()
        n, c, ih, iw = ctx.x_shape
        oh, ow = y.shape[2], y.shape[3]
        fh, fw = f.shape
        # the gradient of upfirdn2d is upfirdn2d with the padding of upfirdn2d.py:251-261 and the filter flip inverted
        bpads = (fw - pads[0] - 1, iw - ow + pads[0], fh - pads[2] - 1, ih - oh + pads[2])
        dx = torch.empty(ctx.x_shape, dtype=dy.dtype, device=dy.device)
        sums = _amax.zeros([2, n * c], dy.device)
        e = custom_ops.FirEpilogue(2, sc.data_ptr() if sc is not None else None, None, y.data_ptr(), sums[0].data_ptr(), sums[1].data_ptr(),
                                   _ba.activation_funcs[act].cuda_idx, alpha, gain, clamp)
        with custom_ops.device_guard(dy):
            # (dx feeds the data-gradient and weight-gradient convolutions of the layer in front: the kernel leaves its magnitude bound behind)
            custom_ops.check(_amax.launch_tracking(dx, lambda: lib.sgv_upfirdn2d_fused(_ufd_params(dy, f, dx, bpads, not flip, fir_gain), e, _DTYPE_CODES[dy.dtype],
                                                                                      custom_ops.raw_stream(dy))), lib)
        d_scale = d_bias = None
        sum_g, sum_gv = sums[0].reshape(n, c), sums[1].reshape(n, c)
        if ctx.has_bias and ctx.needs_input_grad[3]:
            d_bias = sum_g.sum(0)
        if ctx.has_scale and ctx.needs_input_grad[2]:
            b_row = bi.reshape(1, c) if bi is not None else 0.0
            d_scale = ((sum_gv - b_row * sum_g) / sc.reshape(n, c)).reshape(ctx.scale_shape)
        return dx, None, d_scale, d_bias, None


def _fusable(x, f, scale, bias, pads, act, alpha):
    if _fca._composition_depth > 0:     # a pass that is differentiated twice: evaluate the definition
        return False
    if not (enabled and x.is_cuda and x.ndim == 4 and x.dtype in (torch.float32, torch.float16, torch.bfloat16)):
        return False
    if f is None or f.ndim != 2 or f.shape[0] > 4 or f.shape[1] > 4 or not f.is_cuda:
        return False
    if tuple(pads) != (1, 1, 1, 1) or act not in ('linear', 'lrelu') or (act == 'lrelu' and alpha == 0):
        return False
    if scale is not None and (scale.dtype != torch.float32 or scale.numel() != x.shape[0] * x.shape[1]):
        return False
    return True


def fir_bias_act(x, f, scale=None, bias=None, padding=1, fir_gain=1, act='lrelu', alpha=None, gain=None, clamp=None, flip_filter=False):
    """clamp(act(upfirdn2d(x, f, padding, gain=fir_gain) * scale[n,c] + bias[c]) * gain).

    x [N,C,H,W]; f 2-D fp32 filter; scale [N,C] fp32 or None; bias [C] or None; act/alpha/gain/clamp as ``bias_act``."""
    pads = _ufd._parse_padding(padding)
    spec, alpha_f, gain_f, clamp_f = _ba._resolve(act, alpha, gain, clamp)
    if _fusable(x, f, scale, bias, pads, act, alpha_f):
        return _FusedFirBiasActFn.apply(x, f, scale, bias, (pads, float(fir_gain), bool(flip_filter), act, alpha_f, gain_f, clamp_f))
    return fir_bias_act_composed(x, f, scale=scale, bias=bias, padding=padding, fir_gain=fir_gain, act=act, alpha=alpha, gain=gain, clamp=clamp,
                                 flip_filter=flip_filter)


# ---------------------------------------------------------------------------------------------------------------------------------------
# FIR + decimate of a tensor that has a second consumer (the residual discriminator block: the skip branch's `upfirdn2d(x, f, down=2)` next
# to conv0(x), networks.py:343-344).  The node returns x itself as a second output for that other consumer; the other consumer's gradient
# then arrives HERE and is added in the store of this node's own gradient pass (a 2x up-sampling FIR: one streaming kernel,
# sgv_upfirdn2d_fused mode 4) instead of by a separate full-tensor addition.

class _FirDownAliasFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, f, cfg):
        down, pads, flip = cfg
        y = _ufd.upfirdn2d(x, f, down=down, padding=list(pads), flip_filter=flip)
        ctx.set_materialize_grads(False)
        ctx.cfg = cfg
        ctx.in_shape = tuple(x.shape)
        ctx.save_for_backward(f)
        from . import amax as _amax
        return y, _amax.share(x.view_as(x), x)

    @staticmethod
    def backward(ctx, g_y, g_alias=None):
        (f,) = ctx.saved_tensors
        down, pads, flip = ctx.cfg
        if g_y is None:
            return g_alias, None, None
        n, c, ih, iw = ctx.in_shape
        oh, ow = g_y.shape[2], g_y.shape[3]
        fh, fw = f.shape
        # the gradient of upfirdn2d: up / down swapped, the padding of upfirdn2d.py:251-261, the filter flip inverted
        bpads = (fw - pads[0] - 1, iw - ow * down + pads[0], fh - pads[2] - 1, ih - oh * down + pads[2])
        fusable = (not torch.is_grad_enabled() and g_alias is not None and enabled and g_y.is_cuda and g_y.dtype == torch.float32 and g_alias.dtype == torch.float32
                   and g_alias.is_contiguous() and tuple(g_alias.shape) == ctx.in_shape and tuple(f.shape) == (4, 4) and f.dtype == torch.float32)
        if fusable:
            lib = custom_ops.get_native()
            g_y = g_y.contiguous()
            gx = torch.empty(ctx.in_shape, dtype=torch.float32, device=g_y.device)
            e = custom_ops.FirEpilogue(4, None, None, g_alias.data_ptr(), None, None, 1, 0.0, 1.0, -1.0)
            with custom_ops.device_guard(g_y):
                rc = lib.sgv_upfirdn2d_fused(_ufd_params(g_y, f, gx, bpads, not flip, 1.0, up=down, down=1), e, 0, custom_ops.raw_stream(g_y))
            if rc == 0:
                return gx, None, None
            if rc != -3:   # anything but SGV_ERR_UNSUPPORTED (a geometry / width the lane-exchange kernel does not serve) is an error
                custom_ops.check(rc, lib)
        gx = _ufd.upfirdn2d(g_y, f, up=down, padding=list(bpads), flip_filter=not flip)
        return (gx + g_alias if g_alias is not None else gx), None, None


def fir_down_with_input_alias(x, f, down, pads, flip_filter=False):
    """(upfirdn2d(x, f, down=down, padding=pads), x) -- hand the second result to x's other consumer (see _FirDownAliasFn)."""
    pads = tuple(int(v) for v in pads)
    if enabled and x.is_cuda and x.ndim == 4 and f is not None and f.ndim == 2 and f.is_cuda:
        return _FirDownAliasFn.apply(x, f, (int(down), pads, bool(flip_filter)))
    return _ufd.upfirdn2d(x, f, down=down, padding=list(pads), flip_filter=flip_filter), x
