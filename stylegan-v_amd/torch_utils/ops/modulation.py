"""Weight (de)modulation primitives of ``modulated_conv2d`` (reference: src/training/networks.py:57-74).

The reference forms w[N,O,I,kh,kw] = W * s in PyTorch only to reduce it to the demodulation
coefficients d[N,O]; here d comes from two small kernels that never build that tensor
(csrc/modulate.hip: ``sgv_weight_sqsum`` + ``sgv_demod_coefs``), and the per-sample channel scaling
x * s[n, c] (pre-conv styles, post-conv dcoefs; networks.py:66,70-71) is one streaming kernel
(``sgv_scale_channels``).  GPU fp32 tensors take the native path; everything else (CPU, fp64
gradcheck) evaluates the same algebra with PyTorch ops.  Backward passes are written with
differentiable tensor ops, so gradients of any order are available.
"""

import torch


def _amax_mod():
    from . import amax
    return amax

from .. import custom_ops
from .upfirdn2d import _DTYPE_CODES


def _stream(t):
    return custom_ops.raw_stream(t)


def weight_sqsum_ref(weight):
    return weight.float().square().sum(dim=[2, 3])


def demod_coefs_ref(weight, styles, eps=1e-8):
    """d[n,o] = rsqrt(sum_i s[n,i]^2 * sum_k W[o,i,k]^2 + eps): networks.py:59-61 without w[N,O,I,k,k]."""
    q = weight.square().sum(dim=[2, 3])  # [O, I]
    return (styles.square() @ q.t() + eps).rsqrt()


class _DemodCoefsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, weight, styles, eps):
        lib = custom_ops.get_native()
        w = weight.contiguous()
        s = styles.contiguous()
        oc, ic, kh, kw = w.shape
        n = s.shape[0]
        q = torch.empty([oc, ic], dtype=torch.float32, device=w.device)
        d = torch.empty([n, oc], dtype=torch.float32, device=w.device)
        with custom_ops.device_guard(w):
            custom_ops.check(lib.sgv_weight_sqsum(w.data_ptr(), q.data_ptr(), oc, ic, kh * kw, _stream(w)), lib)
            custom_ops.check(lib.sgv_demod_coefs(s.data_ptr(), q.data_ptr(), d.data_ptr(), n, oc, ic, float(eps), _stream(w)), lib)
        ctx.eps = eps
        ctx.save_for_backward(weight, styles, q, d)
        return d

    @staticmethod
    def backward(ctx, grad_d):
        weight, styles, q0, d0 = ctx.saved_tensors
        if not torch.is_grad_enabled():
            # First-order pass (every backward of the FFS / SkyTimelapse configurations: pl_weight = 0): nothing will differentiate these gradients, so q and d come
            # from the forward pass and the chain below is 8 launches instead of 17 on [N, O] / [O, I]-sized tensors (12 demodulated layers per generator
            # backward: ~100 launches per iteration, profiles/r05_c30_small_launch_sources.txt).  Same formulas: g = -1/2 grad_d d^3, grad_w = 2 W (g^T s^2),
            # grad_s = 2 s (g q); the factor 2 * (-1/2) is applied once, to g.
            # Round 6: ONE launch (sgv_demod_coefs_backward) for both gradients.
            if grad_d.is_cuda and weight.dtype == torch.float32 and styles.dtype == torch.float32 and grad_d.dtype == torch.float32:
                lib = custom_ops.get_native()
                gd, wc, sc = grad_d.contiguous(), weight.contiguous(), styles.contiguous()
                oc, ic, kh, kw = wc.shape
                grad_w = torch.empty_like(wc) if ctx.needs_input_grad[0] else None
                grad_s = torch.empty_like(sc) if ctx.needs_input_grad[1] else None
                if grad_w is None and grad_s is None:
                    return None, None, None
                with custom_ops.device_guard(gd):
                    custom_ops.check(lib.sgv_demod_coefs_backward(gd.data_ptr(), d0.data_ptr(), sc.data_ptr(), q0.data_ptr(), wc.data_ptr(), grad_w.data_ptr() if grad_w is not None else None,
                                                                  grad_s.data_ptr() if grad_s is not None else None, sc.shape[0], oc, ic, kh * kw, _stream(gd)), lib)
                return grad_w, grad_s, None
            g = (grad_d * d0.pow(3)).neg_()                                  # [N, O] = 2 g
            grad_w = grad_s = None
            if ctx.needs_input_grad[0]:
                grad_w = weight * (g.t() @ styles.square())[:, :, None, None]
            if ctx.needs_input_grad[1]:
                grad_s = styles * (g @ q0)
            return grad_w, grad_s, None
        # create_graph pass (path-length regularisation differentiates G twice): everything is rebuilt from the INPUTS with differentiable tensor ops (q and d are
        # [O,I] / [N,O]: negligible work), so that the second differentiation sees the dependence of both gradients on weight and styles.  Saved intermediates
        # would come back as constants and silently drop those terms.
        q = weight.square().sum(dim=[2, 3])                              # [O, I]
        d = (styles.square() @ q.t() + ctx.eps).rsqrt()                  # [N, O]
        # d = (s^2 q^T + eps)^(-1/2)  =>  dd/d(s^2 q^T) = -d^3 / 2
        g = grad_d * (-0.5) * d.pow(3)  # [N, O]
        grad_w = grad_s = None
        if ctx.needs_input_grad[0]:
            grad_q = g.t() @ styles.square()  # [O, I]
            grad_w = 2 * weight * grad_q[:, :, None, None]
        if ctx.needs_input_grad[1]:
            grad_s = 2 * styles * (g @ q)  # [N, I]
        return grad_w, grad_s, None


def demod_coefs(weight, styles, eps=1e-8):
    """Demodulation coefficients [N, O] for weight [O, I, kh, kw] and styles [N, I]."""
    if weight.is_cuda and weight.dtype == torch.float32 and styles.dtype == torch.float32:
        return _DemodCoefsFn.apply(weight, styles, eps)
    return demod_coefs_ref(weight, styles, eps)


class _ScaleChannelsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, s):
        lib = custom_ops.get_native()
        xc = x.contiguous()
        sc = s.contiguous()
        n, c = xc.shape[:2]
        hw = xc.numel() // max(n * c, 1)
        y = torch.empty_like(xc)
        if xc.numel():
            with custom_ops.device_guard(xc):
                from . import amax as _amax
                custom_ops.check(_amax.launch_tracking(y, lambda: lib.sgv_scale_channels(xc.data_ptr(), sc.data_ptr(), y.data_ptr(), n, c, hw, _DTYPE_CODES[xc.dtype], _stream(xc))), lib)
        ctx.save_for_backward(x, s)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, s = ctx.saved_tensors
        return _scale_channels_backward(ctx, x, s, dy, None)


def _scale_channels_backward(ctx, x, s, dy, g_alias):
    """(dx, ds) of y = x * s[n, c] for the incoming dy; `g_alias`: a gradient that reached x through another consumer (summed into dx).  First-order pass with both
    gradients wanted: ONE streaming kernel (sgv_scale_dot_add_t: dx = dy * s (+ g_alias), ds = sum_px dy * x -- 3-4 tensor passes) instead of scale_channels +
    plane_dot (+ autograd's addition): 4 (+3)."""
    need_x, need_s = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
    if (need_x and need_s and not torch.is_grad_enabled() and dy.is_cuda and dy.dtype in _DTYPE_CODES and dy.dtype != torch.float64 and x.dtype == dy.dtype
            and s.dtype == torch.float32 and (g_alias is None or (g_alias.dtype == dy.dtype and g_alias.shape == dy.shape))):
        lib = custom_ops.get_native()
        dyc, xc, sc = dy.contiguous(), x.contiguous(), s.contiguous()
        ga = g_alias.contiguous() if g_alias is not None else None
        n, c = xc.shape[:2]
        hw = xc.numel() // max(n * c, 1)
        dx = torch.empty_like(dyc)
        from . import amax as _amax
        dot = _amax.zeros([n * c], dyc.device)
        with custom_ops.device_guard(dyc):
            custom_ops.check(lib.sgv_scale_dot_add_t(dyc.data_ptr(), xc.data_ptr(), sc.data_ptr(), ga.data_ptr() if ga is not None else None, dx.data_ptr(), dot.data_ptr(),
                                                     n * c, hw, _DTYPE_CODES[dyc.dtype], _stream(dyc)), lib)
        return dx, dot.reshape(n, c).to(s.dtype)
    dx = ds = None
    if need_x:
        dx = scale_channels(dy, s)
        if g_alias is not None:
            dx = dx + g_alias
    elif g_alias is not None:
        dx = g_alias
    if need_s:
        ds = plane_dot(dy, x).to(s.dtype)
    return dx, ds


class _ScaleChannelsAliasFn(torch.autograd.Function):
    """(x * s, x): the second result is x itself, for x's OTHER consumer (a synthesis block's output feeds the next block's up-sampling layer -- here -- and its
    own ToRGB, networks.py:239-262).  That consumer's gradient then arrives HERE as g_alias and is added in the store of this node's own gradient pass instead of
    by autograd's separate full-tensor addition (the discriminator's residual blocks use the same device: fused_fir_act.fir_down_with_input_alias)."""

    @staticmethod
    def forward(ctx, x, s):
        y = _ScaleChannelsFn.forward(ctx, x, s)
        ctx.set_materialize_grads(False)
        from . import amax as _amax
        return y, _amax.share(x.view_as(x), x)

    @staticmethod
    def backward(ctx, dy, g_alias=None):
        x, s = ctx.saved_tensors
        if dy is None:
            return g_alias, None
        return _scale_channels_backward(ctx, x, s, dy, g_alias)


def scale_channels_with_alias(x, s):
    """(scale_channels(x, s), x) -- hand the second result to x's other consumer (see _ScaleChannelsAliasFn)."""
    if x.is_cuda and x.ndim == 4 and x.dtype in (torch.float32, torch.float16, torch.bfloat16) and s.dtype == torch.float32 and x.is_contiguous() and x.numel() < 2 ** 31:
        return _ScaleChannelsAliasFn.apply(x, s)
    return scale_channels(x, s), x


class _PlaneDotFn(torch.autograd.Function):
    """out[n,c] = sum_{h,w} a[n,c,h,w] * b[n,c,h,w] in one pass over a and b (csrc/modulate.hip ``sgv_plane_dot``)."""

    @staticmethod
    def forward(ctx, a, b):
        lib = custom_ops.get_native()
        ac, bc = a.contiguous(), b.contiguous()
        n, c = ac.shape[:2]
        hw = ac.numel() // max(n * c, 1)
        out = _amax_mod().zeros([n, c], ac.device)
        if ac.numel():
            with custom_ops.device_guard(ac):
                custom_ops.check(lib.sgv_plane_dot(ac.data_ptr(), bc.data_ptr(), out.data_ptr(), n * c, hw, _DTYPE_CODES[ac.dtype], _stream(ac)), lib)
        ctx.save_for_backward(a, b)
        return out

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        da = db = None
        if ctx.needs_input_grad[0]:
            da = scale_channels(b, g.float())
        if ctx.needs_input_grad[1]:
            db = scale_channels(a, g.float())
        return da, db


def plane_dot(a, b):
    """sum over H,W of a * b -> fp32 [N, C]."""
    if a.is_cuda and a.ndim == 4 and a.dtype == b.dtype and a.dtype in (torch.float32, torch.float16, torch.bfloat16) and a.shape == b.shape \
            and a.numel() < 2 ** 31:
        return _PlaneDotFn.apply(a, b)
    return (a.float() * b.float()).sum(dim=[2, 3])


def scale_channels(x, s):
    """x[N,C,H,W] * s[N,C] broadcast over H,W; s is fp32 (computed in fp32, stored in x's dtype)."""
    if x.is_cuda and x.ndim == 4 and x.dtype in (torch.float32, torch.float16, torch.bfloat16) and s.dtype == torch.float32 \
            and x.is_contiguous() and x.numel() < 2 ** 31:
        return _ScaleChannelsFn.apply(x, s)
    return x * s.to(x.dtype).reshape(x.shape[0], x.shape[1], 1, 1)
