"""Fused down-sampling 3x3 layer tail:  y = clamp(act(conv3x3_stride2(xb, W) + bias) * gain) (+ residual).

``xb`` is the FIR-filtered (2H+1)x(2W+1) tensor that ``conv2d_resample`` hands to its strided convolution (conv2d_resample.py:113-126); the
reference then runs ``bias_act`` as a separate pass (layers.py ``Conv2dLayer.forward``) and, in the residual discriminator block, adds the skip
branch in a third one (``y.add_(x)``, networks.py:343-345).  As in ``fused_conv_act`` there is no reference op to mirror: the function is
DEFINED as that composition (`strided_conv3x3_bias_act_composed`) and, where the kernel serves the shape, evaluated as

  forward   ONE kernel (``sgv_conv3x3_s2_fused``, csrc/conv3x3s2_ws_kernel.h): bias / activation / gain / clamp are applied to the accumulators
            before the store; a residual is updated in place (one fp32 add per element, the reference's ``y.add_(x)``) and the activation output
            is stored next to it (the backward pass needs it).
  backward  ``sgv_act_grad_scale`` (activation gradient from the saved activation output + the bias-gradient sums) -> transposed convolution
            (data gradient) and stride-2 weight gradient on the hand-written kernels; the residual's gradient is dy itself.

Second order: as ``fused_conv_act`` -- a backward that is itself recorded differentiates the composition; passes known to be differentiated twice run
under ``fused_conv_act.composition_only()``.  Parity: tests/test_fused_conv_gpu.py against the oracle composition.
"""

import torch

from .. import custom_ops
from . import amax as _amax
from . import bias_act as _ba
from . import conv2d_gradfix as _cg
from . import fused_conv_act as _fca

_S2 = (False, (2, 2), (0, 0), (0, 0), (1, 1), 1)     # cfg of the strided convolution
_S2T = (True, (2, 2), (0, 0), (0, 0), (1, 1), 1)     # ... and of its data gradient


def strided_conv3x3_bias_act_composed(xb, weight, bias=None, act='lrelu', alpha=None, gain=None, clamp=None, residual=None):
    """The definition: strided convolution -> bias_act -> (+ residual), each differentiable to any order."""
    y = _cg.conv2d(xb, _cg.cast_weight(weight, xb), stride=2)
    y = _ba.bias_act(y, bias.to(y.dtype) if bias is not None else None, act=act, alpha=alpha, gain=gain, clamp=clamp)
    return residual.add_(y) if residual is not None else y       # the reference's in-place form (networks.py:345)


def _launch(xb, weight, bias, residual, want_act, act_idx, alpha, gain, clamp):
    """`residual` (dense fp32, or None) is updated in place and returned: the reference's `y.add_(x)`.  xb may be fp16 / bf16 (no residual then):
    the result has its format; weight and bias stay fp32."""
    _cg._selftest(xb)
    lib = custom_ops.get_native()
    n, ci, hb, wb = xb.shape
    co = weight.shape[0]
    hs, ws_ = (hb - 1) // 2, (wb - 1) // 2
    dt = _cg._DT[xb.dtype]
    assert residual is None or dt == 0
    weight = weight.float()
    y = residual if residual is not None else torch.empty([n, co, hs, ws_], dtype=xb.dtype, device=xb.device)
    a = torch.empty_like(y) if want_act else None
    wsb = int(lib.sgv_conv3x3_s2_workspace_bytes(n, ci, co, hs, ws_, 0))
    wsp = torch.empty([wsb], dtype=torch.uint8, device=xb.device)
    terms = _cg.native_conv_terms if dt == 0 else 1
    p = custom_ops.Conv3x3Params(xb.data_ptr(), weight.data_ptr(), y.data_ptr(), wsp.data_ptr(), wsb, n, ci, co, hs, ws_, 0, terms,
                                 _amax.bound(xb).data_ptr() if terms == 4 else None, None, _amax.bound(weight).data_ptr() if terms == 4 else None)
    if residual is not None:
        _amax.invalidate(residual)      # updated in place through its raw pointer
    e = custom_ops.Conv3x3S2Epilogue(bias.data_ptr() if bias is not None else None, a.data_ptr() if a is not None else None, act_idx, alpha, gain, clamp,
                                     1 if residual is not None else 0)
    with custom_ops.device_guard(xb):
        custom_ops.check(lib.sgv_conv3x3_s2_fused(p, e, dt, custom_ops.raw_stream(xb)), lib)
    return y, a


class _FusedDownFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xb, weight, bias, residual, cfg):
        act, alpha, gain, clamp = cfg
        b = bias.contiguous().float() if bias is not None else None
        need_graph = any(ctx.needs_input_grad[:3])
        y, a = _launch(xb.contiguous(), weight.contiguous(), b, residual, residual is not None and need_graph, _ba.activation_funcs[act].cuda_idx, alpha, gain, clamp)
        if residual is not None:
            ctx.mark_dirty(residual)
        ctx.cfg = cfg
        ctx.has_res = residual is not None
        ctx.bias_dtype = bias.dtype if bias is not None else None
        # with a residual only the activation output is kept (the sum is not needed by any gradient)
        ctx.save_for_backward(xb, weight, bias, a if residual is not None else y)
        return y

    @staticmethod
    def backward(ctx, dy):
        act, alpha, gain, clamp = ctx.cfg
        xb, weight, b, a = ctx.saved_tensors
        if torch.is_grad_enabled():
            # create_graph=True: differentiate the composition on the saved inputs (one extra forward, gradients of any order); the residual enters
            # the composition linearly, so its gradient is dy and a zero stand-in serves
            ins = [t for t, need in zip((xb, weight, b), ctx.needs_input_grad[:3]) if need and t is not None]
            with torch.enable_grad():
                y2 = strided_conv3x3_bias_act_composed(xb, weight, bias=b, act=act, alpha=alpha, gain=gain, clamp=(clamp if clamp >= 0 else None))
                grads = iter(torch.autograd.grad(y2, ins, dy, create_graph=True, allow_unused=True))
            out = tuple(next(grads) if (need and t is not None) else None for t, need in zip((xb, weight, b), ctx.needs_input_grad[:3]))
            return out + (dy if (ctx.has_res and ctx.needs_input_grad[3]) else None, None)
        lib = custom_ops.get_native()
        dy = dy.contiguous()
        d_x = d_w = d_b = None
        d_r = dy if (ctx.has_res and ctx.needs_input_grad[3]) else None
        if any(ctx.needs_input_grad[:3]):
            n, co, h, w = a.shape
            need_db = b is not None and ctx.needs_input_grad[2]
            sums = _amax.zeros([2, n * co], dy.device) if need_db else None
            dz = torch.empty_like(a)
            dy = dy.to(a.dtype)
            with custom_ops.device_guard(dy):
                custom_ops.check(_amax.launch_tracking(dz, lambda: lib.sgv_act_grad_scale_t(
                    dy.data_ptr(), a.data_ptr(), None, dz.data_ptr(), sums.data_ptr() if sums is not None else None, n * co, h * w,
                    _ba.activation_funcs[act].cuda_idx, alpha, gain, clamp, _cg._DT[a.dtype], custom_ops.raw_stream(dy))), lib)
            if need_db:
                d_b = sums[0].reshape(n, co).sum(0).to(ctx.bias_dtype)
            wc = weight.contiguous()
            if ctx.needs_input_grad[0]:
                d_x = _cg._native_conv(dz, wc, _S2T) if _cg._native_conv_ok(dz, wc, _S2T) else _cg._aten_conv(dz, wc.to(dz.dtype), None, _S2T)
            if ctx.needs_input_grad[1] and not _cg.weight_gradients_disabled:
                if _cg._native_wrw_ok(dz, xb, _S2, tuple(weight.shape)):
                    d_w = _cg._native_wrw(dz, xb, _S2, tuple(weight.shape))
                else:
                    _, d_w, _ = torch.ops.aten.convolution_backward(dz, xb, weight.to(xb.dtype), None, (2, 2), (0, 0), (1, 1), False, (0, 0), 1, [False, True, False])
                d_w = d_w.to(weight.dtype)
        return d_x, d_w, d_b, d_r, None


def _fusable(xb, weight, bias, residual, act, alpha, gain, clamp):
    if act == 'linear' and clamp >= 0:   # same reference quirk as fused_conv_act: linear + clamp has an unmasked gradient
        return False
    if _fca.mode == 0 or _fca._composition_depth > 0 or _cg.native_conv_terms not in (1, 3, 4) or not _cg.enabled or not _cg.native_conv_s2:
        return False
    if not (xb.is_cuda and xb.ndim == 4 and xb.dtype in _cg._DT and weight.dtype == torch.float32 and tuple(weight.shape[2:]) == (3, 3)):
        return False
    if xb.dtype != torch.float32 and (not _cg.native_lowp or residual is not None):   # the in-place residual add is an fp32 atomic
        return False
    if act not in ('linear', 'lrelu') or not gain > 0 or (act == 'lrelu' and not 0 <= alpha <= 1):
        return False
    n, ci, hb, wb = xb.shape
    co = weight.shape[0]
    if weight.shape[1] != ci or hb % 2 == 0 or wb % 2 == 0 or hb < 3 or wb < 3:
        return False
    hs, ws_ = (hb - 1) // 2, (wb - 1) // 2
    if bias is not None and tuple(bias.shape) != (co,):
        return False
    if residual is not None and (tuple(residual.shape) != (n, co, hs, ws_) or residual.dtype != torch.float32 or not residual.is_cuda or not residual.is_contiguous()
                                 or (residual.is_leaf and residual.requires_grad)):
        return False
    needs_graph = torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (xb, weight, bias, residual))
    if needs_graph and _fca.mode < 2:
        return False
    return bool(custom_ops.get_native().sgv_conv3x3_s2_fused_supported(n, ci, co, hs, ws_, _cg._DT[xb.dtype]))


def strided_conv3x3_bias_act(xb, weight, bias=None, act='lrelu', alpha=None, gain=None, clamp=None, residual=None):
    """xb [N,I,2H+1,2W+1]; weight [O,I,3,3] (correlation, stride 2, no padding); bias [O] or None; residual [N,O,H,W] or None (consumed: the
    composed form adds into it in place, as the reference does); act / alpha / gain / clamp as ``bias_act``."""
    _, alpha_f, gain_f, clamp_f = _ba._resolve(act, alpha, gain, clamp)
    if _fusable(xb, weight, bias, residual, act, alpha_f, gain_f, clamp_f):
        return _FusedDownFn.apply(xb, weight, bias, residual, (act, alpha_f, gain_f, clamp_f))
    return strided_conv3x3_bias_act_composed(xb, weight, bias=bias, act=act, alpha=alpha, gain=gain, clamp=clamp, residual=residual)
