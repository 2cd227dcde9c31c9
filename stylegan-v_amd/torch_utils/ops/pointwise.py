"""1x1 convolutions with at most 4 channels on one side (ToRGB, fromRGB) as HBM-bound streaming kernels.

Reference: the 1x1 ``conv2d`` of ``ToRGBLayer`` (src/training/networks.py:148-163, C_out = 3, modulated) and of the
discriminator's ``fromrgb`` layer (networks.py:447, C_in = 3), dispatched through conv2d_resample.py:40-54 to cuDNN.
MIOpen runs these shapes at ~1.8 TFLOP/s; they are memory streams (1.4 flop/B) and ``csrc/pointwise.hip`` treats them
so.  ``pointwise_conv(x, w)`` takes per-sample (or shared) weights ``w [N or 1, Cout, Cin]`` -- ToRGB folds its styles
into them, so the separate x*s pass goes away.  Both autograd nodes below are written in terms of each other, so
gradients of any order are available (R1 differentiates the discriminator's fromRGB twice).
"""

import torch


def _amax_mod():
    from . import amax
    return amax

from .. import custom_ops
from . import conv2d_gradfix
from .upfirdn2d import _DTYPE_CODES

enabled = True


def pointwise_conv_ref(x, w):
    """y[n,o,p] = sum_i w[n or 0, o, i] x[n,i,p] with plain PyTorch."""
    n, ci, h, wd = x.shape
    y = torch.matmul(w.to(x.dtype), x.reshape(n, ci, h * wd))
    return y.reshape(n, w.shape[1], h, wd)


def outer_ref(a, b):
    """out[n,f,m] = sum_p a[n,f,p] b[n,m,p] (fp32)."""
    n = a.shape[0]
    return torch.matmul(a.reshape(n, a.shape[1], -1).float(), b.reshape(n, b.shape[1], -1).float().transpose(1, 2))


def _native_ok(x, few, many):
    return (enabled and x.is_cuda and x.ndim == 4 and x.dtype in (torch.float32, torch.float16, torch.bfloat16) and 1 <= few <= 4 and few <= many
            and (x.shape[2] * x.shape[3]) % 4 == 0 and x.shape[0] <= 65535)


class _PointwiseFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w):
        lib = custom_ops.get_native()
        xc = x.contiguous()
        wc = w.contiguous().float()
        n, ci, h, wd = xc.shape
        b, co, _ = wc.shape
        kind = 0 if co <= ci else 1  # 0: many -> few
        y = torch.empty([n, co, h, wd], dtype=xc.dtype, device=xc.device)
        p = custom_ops.PointwiseParams(xc.data_ptr(), wc.data_ptr(), y.data_ptr(), n, max(ci, co), min(ci, co), h * wd,
                                       co * ci if b > 1 else 0, kind)
        with custom_ops.device_guard(xc):
            custom_ops.check(lib.sgv_pointwise_small(p, _DTYPE_CODES[xc.dtype], custom_ops.raw_stream(xc)), lib)
        ctx.save_for_backward(x, w)
        ctx.shared_w = b == 1  # a layer's own weight (not ToRGB's per-sample product of weight and styles)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = pointwise_conv(dy, w.transpose(1, 2))
        if ctx.needs_input_grad[1] and not (ctx.shared_w and conv2d_gradfix.weight_gradients_disabled):
            co, ci = w.shape[1], w.shape[2]
            dw = outer(dy, x) if co <= ci else outer(x, dy).transpose(1, 2)  # [N, co, ci]
            if w.shape[0] == 1:
                dw = dw.sum(0, keepdim=True)
            dw = dw.to(w.dtype)
        return dx, dw


class _OuterFn(torch.autograd.Function):
    """out[n,f,m] = sum_p a[n,f,p] * b[n,m,p]: a has <= 4 channels."""

    @staticmethod
    def forward(ctx, a, b):
        lib = custom_ops.get_native()
        ac, bc = a.contiguous(), b.contiguous()
        n, f, h, wd = ac.shape
        m = bc.shape[1]
        out = _amax_mod().zeros([n, f, m], ac.device)
        with custom_ops.device_guard(ac):
            custom_ops.check(lib.sgv_pointwise_outer(ac.data_ptr(), bc.data_ptr(), out.data_ptr(), n, f, m, h * wd, _DTYPE_CODES[ac.dtype],
                                                     custom_ops.raw_stream(ac)), lib)
        ctx.save_for_backward(a, b)
        return out

    @staticmethod
    def backward(ctx, dout):
        a, b = ctx.saved_tensors
        da = db = None
        if ctx.needs_input_grad[0]:  # da[n,f,p] = sum_m dout[n,f,m] b[n,m,p]
            da = pointwise_conv(b, dout).to(a.dtype)
        if ctx.needs_input_grad[1]:  # db[n,m,p] = sum_f dout[n,f,m] a[n,f,p]
            db = pointwise_conv(a, dout.transpose(1, 2)).to(b.dtype)
        return da, db


class _PointwiseActFn(torch.autograd.Function):
    """y = bias_act(pointwise_conv(x, w), b) for the few -> many form as ONE kernel (sgv_pointwise_act; same operations in the same order as
    the two-pass composition, so bit-identical in fp32).  The backward pass is assembled from the differentiable pieces the composition itself
    uses (bias_act's gradient node, pointwise_conv, outer), so gradients of any order exist."""

    @staticmethod
    def forward(ctx, x, w, b, cfg):
        from . import bias_act as _ba
        act, alpha, gain, clamp = cfg
        lib = custom_ops.get_native()
        xc, wc = x.contiguous(), w.contiguous().float()
        bc = b.contiguous().float() if b is not None else None
        n, ci, h, wd = xc.shape
        co = wc.shape[1]
        y = torch.empty([n, co, h, wd], dtype=torch.float32, device=xc.device)
        p = custom_ops.PointwiseParams(xc.data_ptr(), wc.data_ptr(), y.data_ptr(), n, co, ci, h * wd, 0, 1)
        with custom_ops.device_guard(xc):
            from . import amax as _amax      # (fromRGB's output feeds a 3x3 convolution: the kernel leaves its magnitude bound behind)
            custom_ops.check(_amax.launch_tracking(y, lambda: lib.sgv_pointwise_act(p, bc.data_ptr() if bc is not None else None, _ba.activation_funcs[act].cuda_idx,
                                                                                    alpha, gain, clamp, _DTYPE_CODES[xc.dtype], custom_ops.raw_stream(xc))), lib)
        ctx.cfg = cfg
        ctx.save_for_backward(x, w, b, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import bias_act as _ba
        act, alpha, gain, clamp = ctx.cfg
        x, w, b, y = ctx.saved_tensors
        bcfg = (1, act, alpha, gain, clamp)
        dy = dy.contiguous()
        if not torch.is_grad_enabled() and dy.dtype == torch.float32 and x.shape[1] + 1 <= 4 and dy.data_ptr() % 16 == 0:
            # first-order pass: the activation gradient dz is never written -- the weight + bias gradient kernel and the input gradient
            # kernel evaluate it from (dy, y) on their way in
            lib = custom_ops.get_native()
            n, ci, h, wd = x.shape
            co = w.shape[1]
            aidx = _ba.activation_funcs[act].cuda_idx
            dx = dw = db = None
            need_dw = ctx.needs_input_grad[1] and not conv2d_gradfix.weight_gradients_disabled
            need_db = b is not None and ctx.needs_input_grad[2]
            with custom_ops.device_guard(dy):
                stream = custom_ops.raw_stream(dy)
                if need_dw or need_db:
                    xc = x.contiguous()
                    out = _amax_mod().zeros([n, ci + 1, co], dy.device)
                    custom_ops.check(lib.sgv_pointwise_outer_act(xc.data_ptr(), dy.data_ptr(), y.data_ptr(), out.data_ptr(), n, ci + 1, co, h * wd, 1, aidx, alpha, gain,
                                                                 clamp, _DTYPE_CODES[dy.dtype], stream), lib)
                    tot = out.sum(0)                                  # [ci + 1, co]
                    if need_dw:
                        dw = tot[:ci].t().reshape(1, co, ci).to(w.dtype)
                    if need_db:
                        db = tot[ci].to(b.dtype)
                if ctx.needs_input_grad[0]:
                    wt = w.reshape(co, ci).t().contiguous().float()    # [ci, co]: dx[f] = sum_m w[m, f] dz[m]
                    dx = torch.empty([n, ci, h, wd], dtype=torch.float32, device=dy.device)
                    p = custom_ops.PointwiseParams(dy.data_ptr(), wt.data_ptr(), dx.data_ptr(), n, co, ci, h * wd, 0, 0)
                    custom_ops.check(lib.sgv_pointwise_small_gradin(p, y.data_ptr(), aidx, alpha, gain, clamp, _DTYPE_CODES[dy.dtype], stream), lib)
            return dx, dw, db, None
        dz, db = dy, None
        need_db = b is not None and ctx.needs_input_grad[2]
        if act != 'linear' or gain != 1 or clamp >= 0:
            if need_db and _ba._fused_db_ok(dy, tuple(b.shape), 1):
                dz, db = _ba._BiasActGradDbFn.apply(dy, None, b, y, bcfg, b.shape[0], b.dtype)
            else:
                dz = _ba._BiasActGradFn.apply(dy, None, b, y, bcfg)
        if need_db and db is None:
            db = dz.sum([0, 2, 3]).to(b.dtype)
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = pointwise_conv(dz, w.transpose(1, 2))
        if ctx.needs_input_grad[1] and not conv2d_gradfix.weight_gradients_disabled:
            dw = outer(x, dz).transpose(1, 2).sum(0, keepdim=True).to(w.dtype)
        return dx, dw, db, None


def pointwise_conv_bias_act(x, w, b=None, act='linear', alpha=None, gain=None, clamp=None):
    """bias_act(pointwise_conv(x, w), b, ...) -- fused into the convolution's store for the shared-weight few -> many fp32 case (the
    discriminator's fromRGB: 3 -> C channels + bias + lrelu, layers.py Conv2dLayer.forward), the two-pass composition otherwise."""
    from . import bias_act as _ba
    co, ci = w.shape[1], w.shape[2]
    if (act in ('linear', 'lrelu') and w.shape[0] == 1 and co > ci and x.dtype == torch.float32 and w.is_cuda and _native_ok(x, ci, co)
            and (b is None or (b.is_cuda and tuple(b.shape) == (co,)))):
        _, alpha_f, gain_f, clamp_f = _ba._resolve(act, alpha, gain, clamp)
        return _PointwiseActFn.apply(x, w, b, (act, alpha_f, gain_f, clamp_f))
    return _ba.bias_act(pointwise_conv(x, w), b.to(x.dtype) if b is not None else None, act=act, alpha=alpha, gain=gain, clamp=clamp)


def pointwise_conv(x, w):
    """x [N,Cin,H,W], w [N or 1, Cout, Cin] -> [N,Cout,H,W]; one of Cin / Cout must be <= 4 for the native path."""
    assert x.ndim == 4 and w.ndim == 3 and w.shape[2] == x.shape[1] and w.shape[0] in (1, x.shape[0])
    co, ci = w.shape[1], w.shape[2]
    if _native_ok(x, min(co, ci), max(co, ci)) and w.is_cuda:
        return _PointwiseFn.apply(x, w)
    return pointwise_conv_ref(x, w)


def outer(a, b):
    """a [N,F,H,W] (F <= 4), b [N,M,H,W] -> fp32 [N,F,M]."""
    if _native_ok(a, a.shape[1], b.shape[1]) and b.is_cuda and b.dtype == a.dtype and b.shape[1] >= a.shape[1]:
        return _OuterFn.apply(a, b)
    return outer_ref(a, b)
