"""conv2d / conv_transpose2d with well-behaved higher-order gradients on ROCm.

Boundary names of the reference's ``src/torch_utils/ops/conv2d_gradfix.py`` (``conv2d`` :35,
``conv_transpose2d`` :40, ``no_weight_gradients`` :26, module globals ``enabled`` :22 and
``weight_gradients_disabled`` :23), imported by name from ``loss.py``, ``training_loop.py`` and
``augment.py``.

Why it exists here: stock autograd differentiates a convolution's backward pass
(``_convolution_double_backward``) by re-expressing the weight-gradient as a convolution whose *kernel*
is the full-size output gradient (dilation = stride, batch folded into channels).  MIOpen has no fast
solver for a 256x256 "kernel" and falls back to im2col + one GEMM per sample: measured on MI355X, the
R1 phase (double backward through D, every 16th iteration) took ~10 s per iteration that way -- half of
the whole training time.  The reference solved the same problem on cuDNN with a custom op that is disabled
on torch >= 1.11 (conv2d_gradfix.py:53) and calls cuDNN-only ATen entry points (:143).  This version keeps
the idea and uses backend-neutral ATen ops: every derivative of a convolution is computed with ordinary
forward / backward-data / backward-weight convolutions of the ORIGINAL geometry
(``aten.convolution`` and ``aten.convolution_backward`` with an output mask), nested autograd Functions
make any order available, and ``no_weight_gradients()`` really skips the weight-gradient convolution
(used by the R1 / path-length passes, loss.py:111,162).  Where the hand-written 3x3 family serves the shape (``_native_conv_kind`` /
``_native_wrw_kind`` below: stride 1, stride 2 and transposed stride 2 on fp32 NCHW, csrc/conv3x3*.h, csrc/wrw*.h) each of those
convolutions is one of its kernels -- forward, data gradient, weight gradient and every higher-order term; the vendor library keeps the rest.
"""

import contextlib
import os

import torch

from .. import custom_ops
from . import amax as _amax

enabled = True                     # False -> plain torch.nn.functional calls
weight_gradients_disabled = False  # set inside no_weight_gradients()


@contextlib.contextmanager
def no_weight_gradients():
    global weight_gradients_disabled
    previous = weight_gradients_disabled
    weight_gradients_disabled = True
    try:
        yield
    finally:
        weight_gradients_disabled = previous


# Mixed precision (the reference's `num_fp16_res` blocks, networks.py:227,461; bf16 is this build's extension): activations in fp16 / bf16, fp32
# master weights.  The hand-written kernels take the 16-bit tensors together with the fp32 weight (every value becomes one bf16 operand, fp32
# accumulate) and return weight gradients in fp32 -- `cast_weight` therefore leaves the weight alone where they serve the call; the vendor
# fallback below receives `w.to(x.dtype)` like the reference's `weight.to(x.dtype)` (networks.py:67).
_DT = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}
native_lowp = os.environ.get('SGV_CONV_LOWP', '1') != '0'


def cast_weight(w, x):
    """The weight as conv2d / conv_transpose2d want it for input x: fp32 master weights stay fp32 next to 16-bit GPU activations (native mixed path)."""
    if native_lowp and enabled and x.is_cuda and x.dtype in (torch.float16, torch.bfloat16) and w.dtype == torch.float32 and native_conv_terms in (1, 3, 4):
        return w
    return w.to(x.dtype)


def _pair(v):
    return (int(v), int(v)) if isinstance(v, int) else tuple(int(i) for i in v)


small_image_channel_pad = os.environ.get('SGV_CONV_SMALL_PAD', '1') != '0'


def _pad_small_image_channels(x, w, stride, padding, dilation, groups):
    """3x3 / stride 1 / pad 1 on 4^2 ... 16^2 fp32 images whose input channel count is not a multiple of 64 while the output's is -- the discriminator's epilogue
    convolution, 512 + 1 minibatch-std channels at 4 x 4 (networks.py:518-576) -- gets zero channels up to the next multiple: the native small-image kernel
    (c_in % 16 == 0) then serves the convolution and, with c_in as its output-channel count (% 64 == 0), the data gradient too.  The copies are a few MB; autograd
    differentiates through the padding.  SGV_CONV_SMALL_PAD=0 leaves such layers to the vendor library."""
    if not (small_image_channel_pad and native_conv_terms in (1, 3, 4) and x.is_cuda and x.dtype == torch.float32 and w.dtype == torch.float32 and w.ndim == 4):
        return x, w
    ci, h, wd = x.shape[1], x.shape[2], x.shape[3]
    if not (tuple(w.shape[2:]) == (3, 3) and _pair(stride) == (1, 1) and _pair(padding) == (1, 1) and _pair(dilation) == (1, 1) and int(groups) == 1 and h == wd and h in (4, 8, 16)
            and w.shape[1] == ci and ci % 64 != 0 and ci > 64 and w.shape[0] % 64 == 0):
        return x, w
    pad = 64 - ci % 64
    xp, wp = torch.nn.functional.pad(x, (0, 0, 0, 0, 0, pad)), torch.nn.functional.pad(w, (0, 0, 0, 0, 0, pad))
    if native_conv_terms == 4:      # zero channels do not move a maximum: the padded tensors take their sources' magnitude bounds instead of a pass each
        _amax.same_values(wp, w)
        if _amax.cached(x) is not None:
            _amax.attach(xp, _amax.cached(x))
    return xp, wp


def conv2d(input, weight, bias=None, stride=1, padding=0, dilation=1, groups=1):
    if enabled and input.ndim == 4:
        input, weight = _pad_small_image_channels(input, weight, stride, padding, dilation, groups)
    if _use_custom(input):
        cfg = (False, _pair(stride), _pair(padding), (0, 0), _pair(dilation), int(groups))
        return _Conv.apply(input, weight, bias, cfg)
    if enabled and input.ndim == 4 and bias is None and not torch.is_grad_enabled():   # e.g. the generator pass of the D phase (loss.py:123)
        cfg = (False, _pair(stride), _pair(padding), (0, 0), _pair(dilation), int(groups))
        if _native_conv_ok(input, weight, cfg):
            return _native_conv(input, weight, cfg)
    return torch.nn.functional.conv2d(input=input, weight=weight.to(input.dtype), bias=bias, stride=stride, padding=padding, dilation=dilation, groups=groups)


def conv_transpose2d(input, weight, bias=None, stride=1, padding=0, output_padding=0, groups=1, dilation=1):
    if _use_custom(input):
        cfg = (True, _pair(stride), _pair(padding), _pair(output_padding), _pair(dilation), int(groups))
        return _Conv.apply(input, weight, bias, cfg)
    if enabled and input.ndim == 4 and bias is None and not torch.is_grad_enabled():
        cfg = (True, _pair(stride), _pair(padding), _pair(output_padding), _pair(dilation), int(groups))
        if _native_conv_ok(input, weight, cfg):
            return _native_conv(input, weight, cfg)
    return torch.nn.functional.conv_transpose2d(input=input, weight=weight.to(input.dtype), bias=bias, stride=stride, padding=padding,
                                                output_padding=output_padding, groups=groups, dilation=dilation)


def _use_custom(input):
    assert isinstance(input, torch.Tensor)
    return enabled and input.ndim == 4 and torch.is_grad_enabled()


def _aten_conv(x, w, b, cfg):
    transposed, stride, padding, output_padding, dilation, groups = cfg
    return torch.ops.aten.convolution(x, w, b, stride, padding, dilation, transposed, output_padding, groups)


def _data_grad_cfg(cfg, input_hw, output_hw, kernel_hw):
    """Geometry of the convolution that maps d(output) back to d(input): the opposite kind (conv <-> transposed
    conv) with the same stride/padding/dilation; a forward conv needs the output_padding that restores the
    exact input size."""
    transposed, stride, padding, _, dilation, groups = cfg
    if transposed:
        return (False, stride, padding, (0, 0), dilation, groups)
    out_pad = tuple(input_hw[i] - ((output_hw[i] - 1) * stride[i] - 2 * padding[i] + dilation[i] * (kernel_hw[i] - 1) + 1) for i in range(2))
    assert all(0 <= out_pad[i] < max(stride[i], dilation[i]) for i in range(2)), 'inconsistent convolution geometry'
    return (True, stride, padding, out_pad, dilation, groups)


class _Conv(torch.autograd.Function):
    """y = conv(x, w) (+ b) of either kind; cfg = (transposed, stride, padding, output_padding, dilation, groups)."""

    @staticmethod
    def forward(ctx, x, w, b, cfg):
        ctx.cfg = cfg
        ctx.has_bias = b is not None
        ctx.save_for_backward(x, w)
        if b is None and _native_conv_ok(x, w, cfg):
            return _native_conv(x, w, cfg)
        return _aten_conv(x, w.to(x.dtype), b, cfg)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            bcfg = _data_grad_cfg(ctx.cfg, x.shape[2:], dy.shape[2:], w.shape[2:])
            dx = _Conv.apply(dy, w, None, bcfg)
            assert dx.shape == x.shape
        if ctx.needs_input_grad[1] and not weight_gradients_disabled:
            dw = _ConvGradWeight.apply(dy, x, ctx.cfg, tuple(w.shape), w.dtype)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy.sum([0, 2, 3])
        return dx, dw, db, None


# The 3x3 family on NCHW fp32 tensors runs on the 16-bit matrix pipe with fp32 emulation (csrc/sgv_split.h).  terms = 4 (default since round 4): block-scaled
# 2-way fp16 split, 22 operand bits, fp32 accumulate -- fp32-grade (~1e-7 of the result's scale against float64, the vendor library's fp32 convolutions sit
# at 1.4-3.5e-7; tests/test_conv3x3_gpu.py), which is what the reference's `allow_tf32 = False` configuration computes (training_loop.py:129,141-142);
# 3 -> 2-way bf16 split (16 operand bits, 4.4e-6: inside north_star's 1e-3 but NOT fp32-grade); 1 -> plain bf16 products; 0 -> always the vendor library.
native_wrw_terms = int(os.environ.get('SGV_WRW_TERMS', '4'))
native_conv_terms = int(os.environ.get('SGV_CONV_TERMS', '4'))
native_conv_s2 = os.environ.get('SGV_CONV_S2', '1') != '0'         # stride-2 / transposed members (csrc/conv3x3s2_kernel.h)   # same switch for the forward / data-gradient kernel (csrc/conv3x3_kernel.h)


_selftested = set()      # device indices whose producer / consumer kernels passed the load-time self-test (ops/selftest.py)


def _selftest(t):
    """First native 3x3 launch of the process on t's device: the asm-load kernels prove themselves on exact integer data first (ops/selftest.py)."""
    if not t.is_cuda:
        return
    idx = t.device.index or 0
    if idx in _selftested:
        return
    from . import selftest
    if selftest.run(t.device) in ('ok', 'fallback', 'off'):     # ('running': the test's own launches; 'deferred': a hipGraph is being captured)
        _selftested.add(idx)


def _native_conv_kind(x, w, cfg):
    """Which hand-written kernel serves this convolution: 's1' (3x3 / stride 1 / pad 1, forward or data gradient,
    csrc/conv3x3_kernel.h), 's2' (3x3 / stride 2 / pad 0 between a (2H+1)x(2W+1) and an HxW tensor, strided or transposed,
    csrc/conv3x3s2_kernel.h) or None (vendor library)."""
    transposed, stride, padding, output_padding, dilation, groups = cfg
    if native_conv_terms not in (1, 3, 4) or groups != 1 or dilation != (1, 1) or output_padding != (0, 0):
        return None
    if w.ndim != 4 or tuple(w.shape[2:]) != (3, 3) or not (x.is_cuda and w.is_cuda) or x.dtype not in _DT or w.dtype not in (torch.float32, x.dtype):
        return None
    dt = _DT[x.dtype]
    if dt != 0 and not native_lowp:
        return None
    n, ci, h, wd = x.shape
    co = w.shape[1] if transposed else w.shape[0]
    if (w.shape[0] if transposed else w.shape[1]) != ci:
        return None
    lib = custom_ops.get_native()
    if stride == (1, 1) and padding == (1, 1):
        return 's1' if lib.sgv_conv3x3_supported(n, ci, co, h, wd, dt) else None
    if stride == (2, 2) and padding == (0, 0) and native_conv_s2:
        if transposed:
            return 's2' if lib.sgv_conv3x3_s2_supported_mode(n, ci, co, h, wd, 2, dt) else None
        if h % 2 == 1 and wd % 2 == 1 and h >= 3 and wd >= 3:
            return 's2' if lib.sgv_conv3x3_s2_supported_mode(n, ci, co, (h - 1) // 2, (wd - 1) // 2, 0, dt) else None
    return None


def _native_conv_ok(x, w, cfg):
    return _native_conv_kind(x, w, cfg) is not None


def _native_conv(x, w, cfg):
    _selftest(x)
    lib = custom_ops.get_native()
    kind = _native_conv_kind(x, w, cfg)
    transposed = cfg[0]
    dt = _DT[x.dtype]
    terms = native_conv_terms if dt == 0 else 1      # 16-bit tensors: every value is one bf16 operand
    xc, wc = x.contiguous(), w.float().contiguous()  # the kernels read fp32 weights (a 16-bit copy handed in by a caller is widened again: same bf16 operands)
    if terms == 4 and wc is not w:
        _amax.same_values(wc, w)      # (a transposed view made dense: the bound of the view -- taken on the parameter's own values -- serves the copy)
    if xc.data_ptr() % 16 != 0:   # a dense view at a storage offset that is not 16-byte aligned: the kernels load 16-byte vectors
        xc = xc.clone()
    n, ci, h, wd = xc.shape
    co = wc.shape[1] if transposed else wc.shape[0]
    if kind == 's1':
        y = torch.empty([n, co, h, wd], dtype=x.dtype, device=x.device)
        ws_bytes = int(lib.sgv_conv3x3_workspace_bytes(ci, co))
        ws = torch.empty([ws_bytes], dtype=torch.uint8, device=x.device)
        p = custom_ops.Conv3x3Params(xc.data_ptr(), wc.data_ptr(), y.data_ptr(), ws.data_ptr(), ws_bytes, n, ci, co, h, wd, 1 if transposed else 0, terms,
                                     _amax.bound(xc).data_ptr() if terms == 4 else None, None, _amax.bound(wc).data_ptr() if terms == 4 else None)
        fn = lib.sgv_conv3x3
    else:
        hs, wsm = (h, wd) if transposed else ((h - 1) // 2, (wd - 1) // 2)   # the small grid
        y = torch.empty([n, co, 2 * hs + 1, 2 * wsm + 1] if transposed else [n, co, hs, wsm], dtype=x.dtype, device=x.device)
        mode = 2 if transposed else 0
        ws_bytes = int(lib.sgv_conv3x3_s2_workspace_bytes(n, ci, co, hs, wsm, mode))
        ws = torch.empty([ws_bytes], dtype=torch.uint8, device=x.device)
        p = custom_ops.Conv3x3Params(xc.data_ptr(), wc.data_ptr(), y.data_ptr(), ws.data_ptr(), ws_bytes, n, ci, co, hs, wsm, mode, terms,
                                     _amax.bound(xc).data_ptr() if terms == 4 else None, None, _amax.bound(wc).data_ptr() if terms == 4 else None)
        fn = lib.sgv_conv3x3_s2
    with custom_ops.device_guard(xc):
        custom_ops.check(fn(p, dt, custom_ops.raw_stream(xc)), lib)
    return y


def _native_wrw_kind(dy, x, cfg, w_shape):
    """'s1': 3x3 / stride 1 / pad 1; 's2': 3x3 / stride 2 / pad 0 (strided or transposed layer); None: vendor library."""
    transposed, stride, padding, output_padding, dilation, groups = cfg
    if native_wrw_terms not in (1, 3, 4) or groups != 1 or dilation != (1, 1) or output_padding != (0, 0):
        return None
    if tuple(w_shape[2:]) != (3, 3) or not (dy.is_cuda and x.is_cuda) or x.dtype not in _DT or dy.dtype != x.dtype:
        return None
    dt = _DT[x.dtype]
    if dt != 0 and not native_lowp:
        return None
    lib = custom_ops.get_native()
    n, ci, h, w = x.shape
    if stride == (1, 1) and padding == (1, 1):
        if dy.shape[2:] != x.shape[2:]:
            return None
        # a transposed stride-1 layer (only met as a derivative of a convolution) has the same formula with x and dy swapped
        return 's1' if lib.sgv_conv3x3_wrw_supported(n, dy.shape[1] if not transposed else ci, ci if not transposed else dy.shape[1], h, w, dt) else None
    if stride == (2, 2) and padding == (0, 0) and native_conv_s2:
        small, big = (x, dy) if transposed else (dy, x)
        hs, ws = small.shape[2:]
        if tuple(big.shape[2:]) != (2 * hs + 1, 2 * ws + 1):
            return None
        return 's2' if lib.sgv_conv3x3_wrw_s2_supported(n, small.shape[1], big.shape[1], hs, ws, dt) else None
    return None


def _native_wrw_ok(dy, x, cfg, w_shape):
    return _native_wrw_kind(dy, x, cfg, w_shape) is not None


wrw_input_scale = os.environ.get('SGV_WRW_WS', '1') != '0' and os.environ.get('SGV_WRW_SCALE', '1') != '0'   # the stride-1 producer / consumer kernel can scale its input operand per (sample, channel)


def _wrw_bounds(terms, dyc, xc, scale=None):
    """(dy_amax, x_amax, x_amax2) of ConvWrwParams: device pointers to the operands' magnitude bounds for the block-scaled split (terms = 4)."""
    if terms != 4:
        return None, None, None
    return _amax.bound(dyc).data_ptr(), _amax.bound(xc).data_ptr(), (_amax.bound(scale).data_ptr() if scale is not None else None)


def _native_wrw(dy, x, cfg, w_shape, x_scale=None):
    """``x_scale`` ([N, Cin] fp32; stride-1 forward layers only, see ``wrw_input_scale``): the gradient is taken with x * x_scale[:, :, None, None]."""
    _selftest(x)
    lib = custom_ops.get_native()
    kind = _native_wrw_kind(dy, x, cfg, w_shape)
    dt = _DT[x.dtype]
    terms = native_wrw_terms if dt == 0 else 1
    dw = torch.empty(w_shape, dtype=torch.float32, device=x.device)    # fp32 whatever the tensors' format
    if kind == 's1' and x_scale is not None:
        assert not cfg[0] and wrw_input_scale
        dyc, xc, sc = dy.contiguous(), x.contiguous(), x_scale.contiguous()
        n, ci, h, w = xc.shape
        assert tuple(w_shape[:2]) == (dyc.shape[1], ci) and tuple(sc.shape) == (n, ci) and sc.dtype == torch.float32
        p = custom_ops.ConvWrwParams(dyc.data_ptr(), xc.data_ptr(), dw.data_ptr(), n, dyc.shape[1], ci, h, w, terms, *_wrw_bounds(terms, dyc, xc, sc))
        with custom_ops.device_guard(xc):
            custom_ops.check(lib.sgv_conv3x3_wrw_scaled(p, sc.data_ptr(), dt, custom_ops.raw_stream(xc)), lib)
        return dw
    assert x_scale is None
    if kind == 's1':
        dyc, xc = (x.contiguous(), dy.contiguous()) if cfg[0] else (dy.contiguous(), x.contiguous())   # weight is [dyc channels, xc channels, 3, 3]
        n, ci, h, w = xc.shape
        assert tuple(w_shape[:2]) == (dyc.shape[1], ci)
        p = custom_ops.ConvWrwParams(dyc.data_ptr(), xc.data_ptr(), dw.data_ptr(), n, dyc.shape[1], ci, h, w, terms, *_wrw_bounds(terms, dyc, xc))
        fn = lib.sgv_conv3x3_wrw
    else:   # the weight is [c_small, c_big, 3, 3] for both the strided ([c_out, c_in]) and the transposed ([c_in, c_out]) layer
        small, big = ((x, dy) if cfg[0] else (dy, x))
        dyc, xc = small.contiguous(), big.contiguous()
        n, cs, hs, ws = dyc.shape
        assert tuple(w_shape[:2]) == (cs, xc.shape[1])
        p = custom_ops.ConvWrwParams(dyc.data_ptr(), xc.data_ptr(), dw.data_ptr(), n, cs, xc.shape[1], hs, ws, terms, *_wrw_bounds(terms, dyc, xc))
        fn = lib.sgv_conv3x3_wrw_s2
    with custom_ops.device_guard(xc):
        custom_ops.check(fn(p, dt, custom_ops.raw_stream(xc)), lib)
    return dw


class _ConvGradWeight(torch.autograd.Function):
    """dw = d<dy, conv(x, w)>/dw, as one backward-weight convolution; bilinear in (dy, x), so its own
    derivatives are again plain convolutions."""

    @staticmethod
    def forward(ctx, dy, x, cfg, w_shape, w_dtype=None):
        transposed, stride, padding, output_padding, dilation, groups = cfg
        ctx.cfg = cfg
        ctx.save_for_backward(dy, x)
        w_dtype = x.dtype if w_dtype is None else w_dtype
        if _native_wrw_ok(dy, x, cfg, w_shape):
            return _native_wrw(dy, x, cfg, w_shape).to(w_dtype)
        w_like = x.new_empty(w_shape)  # only its shape/dtype are read when output_mask selects the weight gradient
        _, dw, _ = torch.ops.aten.convolution_backward(dy, x, w_like, None, stride, padding, dilation, transposed, output_padding, groups,
                                                       [False, True, False])
        return dw.to(w_dtype)

    @staticmethod
    def backward(ctx, d_dw):
        dy, x = ctx.saved_tensors
        g_dy = g_x = None
        if ctx.needs_input_grad[0]:   # dw is linear in dy: the forward convolution of x with the incoming gradient as weight
            g_dy = _Conv.apply(x, d_dw, None, ctx.cfg)
            assert g_dy.shape == dy.shape
        if ctx.needs_input_grad[1]:   # and linear in x: the data-gradient convolution of dy with that weight
            bcfg = _data_grad_cfg(ctx.cfg, x.shape[2:], dy.shape[2:], d_dw.shape[2:])
            g_x = _Conv.apply(dy, d_dw, None, bcfg)
            assert g_x.shape == x.shape
        return (g_dy, g_x) + (None,) * (len(ctx.needs_input_grad) - 2)
