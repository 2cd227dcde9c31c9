"""Fused stride-1 3x3 layer:  y = clamp(act(dcoefs[n,o] * conv3x3(x * styles[n,i], W) + bias[o]) * gain).

The reference evaluates this as four separate full-tensor operations on the training path -- ``x * styles`` (networks.py:66),
the convolution (conv2d_resample.py:40-54 -> conv2d_gradfix.py:35), ``x * dcoefs`` (networks.py:70-71) and ``bias_act``
(networks.py:141-143; layers.py ``Conv2dLayer.forward`` for the un-modulated discriminator layers).  There is no single
reference op to mirror, so -- like ``fused_fir_act`` -- this is an internal fusion of that sequence:
``conv3x3_bias_act`` is DEFINED as the composition (`conv3x3_bias_act_composed`), and on the GPU it runs as

  forward   ONE kernel (``sgv_conv3x3_fused``, csrc/conv3x3_ws_kernel.h): the styles multiply the activations on their way into
            the matrix-core operand tiles, dcoefs / bias / activation / gain / clamp are applied to the accumulators before the
            only store.  Six full-tensor passes of the composition disappear.
  backward  ``sgv_act_grad_scale`` (activation gradient from the saved OUTPUT, times dcoefs, with the per-plane sums that give
            the bias and dcoefs gradients) -> data-gradient convolution -> ``sgv_scale_dot`` (input gradient and styles gradient
            in one pass) -> weight-gradient convolution on the re-scaled input.

Second order: when the node's gradient is itself differentiated (``create_graph=True``) its backward switches to differentiating
the composition on the saved inputs (one extra forward), so gradients of any order exist.  Passes that are known to be
differentiated twice (R1, path-length regularisation; training/loss.py) run under ``composition_only()`` and skip the fused
forward altogether; so do CPU tensors and shapes the kernel does not serve.  fp16 / bf16 activations (the mixed-precision blocks) take the same kernels with
16-bit tensor I/O: fp32 weight, scales, bias and accumulators.  Parity: forward within fused-multiply-add rounding of the composition (1e-6),
tests/test_fused_conv_gpu.py against the oracle composition.
"""

import contextlib
import os

import torch

from .. import custom_ops
from . import bias_act as _ba
from . import amax as _amax
from . import conv2d_gradfix as _cg
from . import modulation as _mod

# 0: never fuse; 1: only passes that record no autograd graph (the generator pass of the D phase, inference); 2: training too
mode = int(os.environ.get('SGV_FUSED_CONV', '2'))
_composition_depth = 0
accumulate_input_gradients = os.environ.get('SGV_ALIAS_ACC', '1') != '0'   # _FusedConvActFirFn: sum the layer input's two gradients in the data-gradient kernel's store


@contextlib.contextmanager
def composition_only():
    """Inside this context the layer is evaluated as its composition (needed wherever the result is differentiated twice)."""
    global _composition_depth
    _composition_depth += 1
    try:
        yield
    finally:
        _composition_depth -= 1


def conv3x3_bias_act_composed(x, weight, styles=None, dcoefs=None, bias=None, act='lrelu', alpha=None, gain=None, clamp=None):
    """The definition: scale -> conv -> scale -> bias_act, each differentiable to any order."""
    if styles is not None:
        x = _mod.scale_channels(x, styles)
    y = _cg.conv2d(x, _cg.cast_weight(weight, x), padding=1)
    if dcoefs is not None:
        y = _mod.scale_channels(y, dcoefs)
    return _ba.bias_act(y, bias.to(y.dtype) if bias is not None else None, act=act, alpha=alpha, gain=gain, clamp=clamp)


def _launch_fused(x, weight, styles, dcoefs, bias, act_idx, alpha, gain, clamp, mode=0, accumulate_into=None):
    """mode 0: forward (weight [O,I,3,3]); mode 1: data gradient of that layer (x is the output-side tensor, the result has I channels).
    ``accumulate_into``: an existing fp32 tensor of the result's shape that receives  += result  instead of a fresh output.
    x may be fp16 / bf16 (the mixed-precision blocks): the result has x's format, the weight / scales / bias stay fp32 (single bf16 operands, fp32 accumulate)."""
    _cg._selftest(x)
    lib = custom_ops.get_native()
    n, ci, h, w = x.shape
    co = weight.shape[0] if mode == 0 else weight.shape[1]
    dt = _cg._DT[x.dtype]
    assert accumulate_into is None or dt == 0
    y = accumulate_into if accumulate_into is not None else torch.empty([n, co, h, w], dtype=x.dtype, device=x.device)
    ws_bytes = int(lib.sgv_conv3x3_workspace_bytes(ci, co))
    ws = torch.empty([ws_bytes], dtype=torch.uint8, device=x.device)
    weight = weight.float()
    terms = _cg.native_conv_terms if dt == 0 else 1
    # terms = 4: the operand is x * styles, bounded by the product of the two tensors' bounds (csrc/sgv_split.h)
    p = custom_ops.Conv3x3Params(x.data_ptr(), weight.data_ptr(), y.data_ptr(), ws.data_ptr(), ws_bytes, n, ci, co, h, w, mode, terms,
                                 _amax.bound(x).data_ptr() if terms == 4 else None, _amax.bound(styles).data_ptr() if (terms == 4 and styles is not None) else None,
                                 _amax.bound(weight).data_ptr() if terms == 4 else None)
    old_bound = None
    if accumulate_into is not None:
        old_bound = _amax.cached(accumulate_into)
        _amax.invalidate(accumulate_into)      # written through its raw pointer below
    e = custom_ops.Conv3x3Epilogue(styles.data_ptr() if styles is not None else None, dcoefs.data_ptr() if dcoefs is not None else None,
                                   bias.data_ptr() if bias is not None else None, act_idx, alpha, gain, clamp, 1 if accumulate_into is not None else 0)
    with custom_ops.device_guard(x):
        # the store leaves max |stored value| behind (the producer / consumer kernel's epilogue): the bound of a fresh result -- or, with `accumulate_into`, of the
        # increment: |old + increment| <= bound(old) + bound(increment), one 1-element add instead of a pass over the sum (the skip GEMM's gradients read it)
        custom_ops.check(_amax.launch_tracking(y, lambda: lib.sgv_conv3x3_fused(p, e, dt, custom_ops.raw_stream(x))), lib)
    if accumulate_into is not None:
        inc = _amax.cached(y)
        if inc is not None and old_bound is not None:
            _amax.attach(y, old_bound + inc)
        else:
            _amax.invalidate(y)
    return y


class _FusedConvBiasActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, styles, dcoefs, bias, cfg):
        act, alpha, gain, clamp = cfg
        s = styles.contiguous() if styles is not None else None
        d = dcoefs.contiguous() if dcoefs is not None else None
        b = bias.contiguous().float() if bias is not None else None
        y = _launch_fused(x.contiguous(), weight.contiguous(), s, d, b, _ba.activation_funcs[act].cuda_idx, alpha, gain, clamp)
        ctx.cfg = cfg
        ctx.bias_dtype = bias.dtype if bias is not None else None
        ctx.save_for_backward(x, weight, styles, dcoefs, bias, y)   # the inputs themselves: the create_graph path below needs their history
        return y

    @staticmethod
    def backward(ctx, dy):
        act, alpha, gain, clamp = ctx.cfg
        x, weight, s, d, b, y = ctx.saved_tensors
        if torch.is_grad_enabled():
            # create_graph=True (somebody differentiates this gradient again, e.g. R1 without `composition_only()`): differentiate the
            # composition on the saved inputs instead -- one extra forward, gradients of any order.
            ins = [t for t, need in zip((x, weight, s, d, b), ctx.needs_input_grad[:5]) if need and t is not None]
            with torch.enable_grad():
                y2 = conv3x3_bias_act_composed(x, weight, styles=s, dcoefs=d, bias=b, act=act, alpha=alpha, gain=gain, clamp=(clamp if clamp >= 0 else None))
                grads = iter(torch.autograd.grad(y2, ins, dy, create_graph=True, allow_unused=True))
            return tuple(next(grads) if (need and t is not None) else None for t, need in zip((x, weight, s, d, b), ctx.needs_input_grad[:5])) + (None,)
        lib = custom_ops.get_native()
        dy, x, weight = dy.contiguous(), x.contiguous(), weight.contiguous()
        s = s.contiguous() if s is not None else None
        d = d.contiguous() if d is not None else None
        b = b.contiguous().float() if b is not None else None
        n, co, h, w = y.shape
        ci = x.shape[1]
        stream = custom_ops.raw_stream(dy)
        need_sums = (b is not None and ctx.needs_input_grad[4]) or (d is not None and ctx.needs_input_grad[3])
        sums = _amax.zeros([2, n * co], dy.device) if need_sums else None
        dzd = torch.empty_like(y)     # gradient w.r.t. the convolution result: bias_act gradient times dcoefs
        dt = _cg._DT[y.dtype]
        dy = dy.to(y.dtype)
        with custom_ops.device_guard(dy):   # (dzd feeds two convolutions: the kernel leaves its magnitude bound behind, ops/amax.py)
            custom_ops.check(_amax.launch_tracking(dzd, lambda: lib.sgv_act_grad_scale_t(
                dy.data_ptr(), y.data_ptr(), d.data_ptr() if d is not None else None, dzd.data_ptr(), sums.data_ptr() if sums is not None else None, n * co, h * w,
                _ba.activation_funcs[act].cuda_idx, alpha, gain, clamp, dt, stream)), lib)
        d_x = d_w = d_s = d_d = d_b = None
        if b is not None and ctx.needs_input_grad[4]:
            d_b = sums[0].reshape(n, co).sum(0).to(ctx.bias_dtype)
        if d is not None and ctx.needs_input_grad[3]:
            sg, sgv = sums[0].reshape(n, co), sums[1].reshape(n, co)
            d_d = (sgv - (b.reshape(1, co) * sg if b is not None else 0.0)) / d
        cfg = (False, (1, 1), (1, 1), (0, 0), (1, 1), 1)
        if ctx.needs_input_grad[0] or (s is not None and ctx.needs_input_grad[2]):
            tcfg = (True, (1, 1), (1, 1), (0, 0), (1, 1), 1)                                    # data gradient: the transposed form, same kernel family
            dxs = _cg._native_conv(dzd, weight, tcfg) if _cg._native_conv_ok(dzd, weight, tcfg) else _cg._aten_conv(dzd, weight.to(dzd.dtype), None, tcfg)
            if s is not None:
                d_x = torch.empty_like(dxs)
                dot = _amax.zeros([n * ci], dy.device)
                with custom_ops.device_guard(dy):
                    custom_ops.check(lib.sgv_scale_dot_t(dxs.data_ptr(), x.data_ptr(), s.data_ptr(), d_x.data_ptr(), dot.data_ptr(), n * ci, h * w, dt, stream), lib)
                d_s = dot.reshape(n, ci)
            else:
                d_x = dxs
        if ctx.needs_input_grad[1] and not _cg.weight_gradients_disabled:
            if s is not None and _cg.wrw_input_scale and _cg._native_wrw_kind(dzd, x, cfg, tuple(weight.shape)) == 's1':
                d_w = _cg._native_wrw(dzd, x, cfg, tuple(weight.shape), x_scale=s)   # x * styles is formed on the operand's way into LDS
            else:
                xs = _mod.scale_channels(x, s) if s is not None else x
                if _cg._native_wrw_ok(dzd, xs, cfg, tuple(weight.shape)):
                    d_w = _cg._native_wrw(dzd, xs, cfg, tuple(weight.shape))
                else:
                    _, d_w, _ = torch.ops.aten.convolution_backward(dzd, xs, weight.to(xs.dtype), None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1, [False, True, False])
            d_w = d_w.to(weight.dtype)
        return d_x, d_w, d_s, d_d, d_b, None


def _fusable(x, weight, styles, dcoefs, bias, act, alpha, gain, clamp):
    if act == 'linear' and clamp >= 0:   # the reference's linear + clamp gradient is NOT masked where the output saturated (bias_act.py:24 saves no y); keep that
        return False
    if mode == 0 or _composition_depth > 0 or _cg.native_conv_terms not in (1, 3, 4) or not _cg.enabled:
        return False
    if not (x.is_cuda and x.ndim == 4 and x.dtype in _cg._DT and weight.dtype == torch.float32 and tuple(weight.shape[2:]) == (3, 3)):
        return False
    if x.dtype != torch.float32 and not _cg.native_lowp:
        return False
    if act not in ('linear', 'lrelu') or not gain > 0 or (act == 'lrelu' and not 0 <= alpha <= 1):
        return False
    n, ci, h, w = x.shape
    co = weight.shape[0]
    if weight.shape[1] != ci or n * max(ci, co) > 65535 * 16:      # (planes: the element-wise backward kernels take them in slabs of 65,535; plane_dot up to 16 slabs)
        return False
    for t, c in ((styles, ci), (dcoefs, co)):
        if t is not None and (t.dtype != torch.float32 or tuple(t.shape) != (n, c)):
            return False
    if bias is not None and tuple(bias.shape) != (co,):
        return False
    needs_graph = torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (x, weight, styles, dcoefs, bias))
    if needs_graph and mode < 2:
        return False
    return bool(custom_ops.get_native().sgv_conv3x3_fused_supported(n, ci, co, h, w, _cg._DT[x.dtype]))


def conv3x3_bias_act(x, weight, styles=None, dcoefs=None, bias=None, act='lrelu', alpha=None, gain=None, clamp=None):
    """x [N,I,H,W]; weight [O,I,3,3] (correlation, padding 1); styles [N,I] / dcoefs [N,O] fp32 or None; bias [O] or None;
    act / alpha / gain / clamp as ``bias_act``."""
    _, alpha_f, gain_f, clamp_f = _ba._resolve(act, alpha, gain, clamp)
    if _fusable(x, weight, styles, dcoefs, bias, act, alpha_f, gain_f, clamp_f):
        return _FusedConvBiasActFn.apply(x, weight, styles, dcoefs, bias, (act, alpha_f, gain_f, clamp_f))
    return conv3x3_bias_act_composed(x, weight, styles=styles, dcoefs=dcoefs, bias=bias, act=act, alpha=alpha, gain=gain, clamp=clamp)


# ---------------------------------------------------------------------------------------------------------------------------------------
# Stride-1 layer whose output goes straight into the FIR in front of a down-sampling convolution (DiscriminatorBlock: conv0 -> conv1,
# networks.py:343-344; the FIR is conv2d_resample.py:113-126's).  Forward is the fused layer kernel followed by the FIR pass; what the pairing
# buys is the backward pass: the FIR's gradient (another FIR) and the layer's activation gradient are ONE kernel (sgv_upfirdn2d_fused mode 3)
# instead of upfirdn2d + sgv_act_grad_scale -- three tensor passes instead of five on the block's largest activations.

def conv3x3_bias_act_then_fir_composed(x, weight, bias, f, pads, act='lrelu', alpha=None, gain=None, clamp=None):
    from . import upfirdn2d as _ufd
    y = conv3x3_bias_act_composed(x, weight, bias=bias, act=act, alpha=alpha, gain=gain, clamp=clamp)
    return _ufd.upfirdn2d(y, f, padding=list(pads))


class _FusedConvActFirFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, f, cfg):
        from . import upfirdn2d as _ufd
        act, alpha, gain, clamp, pads = cfg
        b = bias.contiguous().float() if bias is not None else None
        y0 = _launch_fused(x.contiguous(), weight.contiguous(), None, None, b, _ba.activation_funcs[act].cuda_idx, alpha, gain, clamp)
        xb = _ufd.upfirdn2d(y0, f, padding=list(pads))
        ctx.set_materialize_grads(False)   # an unused output's gradient arrives as None, not as a tensor of zeros
        ctx.cfg = cfg
        ctx.bias_dtype = bias.dtype if bias is not None else None
        ctx.save_for_backward(x, weight, bias, f, y0)
        # second output: x itself, for the layer input's OTHER consumer (the residual block's skip branch).  Its gradient then arrives here
        # as g_alias and the data-gradient convolution adds into it in its store, instead of autograd adding two full tensors afterwards.
        return xb, _amax.share(x.view_as(x), x)

    @staticmethod
    def backward(ctx, g, g_alias=None):
        from . import fused_fir_act as _ffa
        from . import upfirdn2d as _ufd
        act, alpha, gain, clamp, pads = ctx.cfg
        x, weight, b, f, y0 = ctx.saved_tensors
        if g is None:   # only the alias was used downstream (also under create_graph: there is nothing of this layer to differentiate)
            return (g_alias if ctx.needs_input_grad[0] else None), None, None, None, None
        if torch.is_grad_enabled():
            ins = [t for t, need in zip((x, weight, b), ctx.needs_input_grad[:3]) if need and t is not None]
            with torch.enable_grad():
                y2 = conv3x3_bias_act_then_fir_composed(x, weight, b, f, pads, act=act, alpha=alpha, gain=gain, clamp=(clamp if clamp >= 0 else None))
                grads = iter(torch.autograd.grad(y2, ins, g, create_graph=True, allow_unused=True))
            out = [next(grads) if (need and t is not None) else None for t, need in zip((x, weight, b), ctx.needs_input_grad[:3])]
            if g_alias is not None and ctx.needs_input_grad[0]:
                out[0] = g_alias if out[0] is None else out[0] + g_alias
            return tuple(out) + (None, None)
        lib = custom_ops.get_native()
        g = g.contiguous()
        n, co, h, w = y0.shape
        fh, fw = f.shape
        # the gradient of upfirdn2d is upfirdn2d with the padding of upfirdn2d.py:251-261 and the filter flip inverted
        bpads = (fw - pads[0] - 1, w - g.shape[3] + pads[0], fh - pads[2] - 1, h - g.shape[2] + pads[2])
        dz = torch.empty_like(y0)
        dt = _cg._DT[y0.dtype]
        g = g.to(y0.dtype)
        sums = _amax.zeros([n * co], g.device)
        e = custom_ops.FirEpilogue(3, None, None, y0.data_ptr(), sums.data_ptr(), None, _ba.activation_funcs[act].cuda_idx, alpha, gain, clamp)
        with custom_ops.device_guard(g):
            rc = _amax.launch_tracking(dz, lambda: lib.sgv_upfirdn2d_fused(_ffa._ufd_params(g, f, dz, bpads, True, 1.0), e, dt, custom_ops.raw_stream(g)))
        need_db = b is not None and ctx.needs_input_grad[2]
        d_x = d_w = d_b = None
        if rc == 0:
            if need_db:
                d_b = sums.reshape(n, co).sum(0).to(ctx.bias_dtype)
        elif rc == -3:   # SGV_ERR_UNSUPPORTED (a geometry / width the lane-exchange kernel does not serve): the two passes
            gy = _ufd.upfirdn2d(g, f, padding=list(bpads), flip_filter=True)
            s2 = _amax.zeros([2, n * co], g.device) if need_db else None
            with custom_ops.device_guard(g):
                custom_ops.check(_amax.launch_tracking(dz, lambda: lib.sgv_act_grad_scale_t(
                    gy.data_ptr(), y0.data_ptr(), None, dz.data_ptr(), s2.data_ptr() if s2 is not None else None, n * co, h * w,
                    _ba.activation_funcs[act].cuda_idx, alpha, gain, clamp, dt, custom_ops.raw_stream(g))), lib)
            if need_db:
                d_b = s2[0].reshape(n, co).sum(0).to(ctx.bias_dtype)
        else:
            custom_ops.check(rc, lib)
        wc = weight.contiguous()
        cfg1 = (False, (1, 1), (1, 1), (0, 0), (1, 1), 1)
        if ctx.needs_input_grad[0]:
            tcfg = (True, (1, 1), (1, 1), (0, 0), (1, 1), 1)
            ci = x.shape[1]
            # In-place use of an incoming gradient: only a tensor that is plainly this node's to consume -- no autograd history (a recorded backward took
            # the create_graph branch above), not a view into somebody else's storage.  A hook or `retain_grad()` on the alias output would still see
            # the sum instead of the skip branch's gradient; SGV_ALIAS_ACC=0 restores the out-of-place add for such uses.
            if (accumulate_input_gradients and g_alias is not None and g_alias.is_contiguous() and g_alias.dtype == torch.float32 and dt == 0 and _cg.native_conv_terms in (1, 3, 4)
                    and not g_alias.requires_grad and g_alias._base is None
                    and lib.sgv_conv3x3_fused_supported(n, co, ci, h, w, 0)):
                # the data gradient lands IN the gradient the skip branch produced (one fp32 add per element in the convolution's store)
                d_x = _launch_fused(dz, wc, None, None, None, 1, 0.0, 1.0, -1.0, mode=1, accumulate_into=g_alias)
            else:
                d_x = _cg._native_conv(dz, wc, tcfg) if _cg._native_conv_ok(dz, wc, tcfg) else _cg._aten_conv(dz, wc.to(dz.dtype), None, tcfg)
                if g_alias is not None:
                    d_x = d_x + g_alias
        if ctx.needs_input_grad[1] and not _cg.weight_gradients_disabled:
            if _cg._native_wrw_ok(dz, x, cfg1, tuple(weight.shape)):
                d_w = _cg._native_wrw(dz, x, cfg1, tuple(weight.shape))
            else:
                _, d_w, _ = torch.ops.aten.convolution_backward(dz, x, weight.to(x.dtype), None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1, [False, True, False])
            d_w = d_w.to(weight.dtype)
        return d_x, d_w, d_b, None, None


def conv3x3_bias_act_then_fir(x, weight, bias, f, pads, act='lrelu', alpha=None, gain=None, clamp=None, with_input_alias=False):
    """upfirdn2d(conv3x3_bias_act(x, weight, bias=bias, ...), f, padding=pads) for an un-modulated layer; pads = (px0, px1, py0, py1).
    ``with_input_alias``: also return x for its other consumer (a residual block's skip branch) -- through the fused node, so that the two
    gradients of x meet inside the data-gradient convolution instead of in a separate addition."""
    _, alpha_f, gain_f, clamp_f = _ba._resolve(act, alpha, gain, clamp)
    pads = tuple(int(v) for v in pads)
    if (_fusable(x, weight, None, None, bias, act, alpha_f, gain_f, clamp_f) and f is not None and f.ndim == 2 and tuple(f.shape) == (4, 4) and f.is_cuda
            and f.dtype == torch.float32 and pads == (2, 2, 2, 2)):
        xb, x_alias = _FusedConvActFirFn.apply(x, weight, bias, f, (act, alpha_f, gain_f, clamp_f, pads))
        return (xb, x_alias) if with_input_alias else xb
    xb = conv3x3_bias_act_then_fir_composed(x, weight, bias, f, pads, act=act, alpha=alpha, gain=gain, clamp=clamp)
    return (xb, x) if with_input_alias else xb
