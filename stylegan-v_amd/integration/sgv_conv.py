"""The 3x3 convolution kernels behind the reference's ``conv2d_gradfix`` (src/torch_utils/ops/conv2d_gradfix.py).

The reference funnels every convolution through ``conv2d`` / ``conv_transpose2d`` (:35-43) and builds gradients from three
primitives (:100-118 data gradient = the opposite kind of convolution, :140-170 ``Conv2dGradWeight``).  Reference-side
edit: ``Conv2d.forward`` (:112-114) tries ``conv3x3(input, weight, transpose, stride[0])`` before
``torch.nn.functional.conv2d`` / ``conv_transpose2d``; ``Conv2dGradWeight.forward`` (:146-152) tries
``conv3x3_weight_grad(...)`` before the cuDNN op.  Both return ``None`` for shapes the kernels do not serve (-> vendor
library).  Arithmetic: fp32 tensors and accumulation; ``terms = 4`` (default) = three fp16 MFMAs on hi/lo splits of the block-scaled operands --
fp32-grade, ~1e-7 vs float64, what the reference's ``allow_tf32 = False`` runs compute; each operand's magnitude bound is one ``sgv_absmax`` pass into a
1-element device tensor.  ``terms = 3`` = bf16 splits (~4e-6), ``terms = 1`` = plain bf16 products.
"""
import ctypes
import os

import torch

_LIB_PATH = os.environ.get('SGV_HIP_LIB') or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'csrc', 'libsgv_hip.so')
_lib = None
_i32, _i64, _vp = ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p


class _Conv(ctypes.Structure):   # struct sgv_conv3x3_params
    _fields_ = [('x', _vp), ('weight', _vp), ('y', _vp), ('workspace', _vp), ('workspace_bytes', _i64)] + \
               [(k, _i32) for k in ('n', 'c_in', 'c_out', 'h', 'w', 'mode', 'terms')] + [('x_amax', _vp), ('x_amax2', _vp), ('w_amax', _vp)]


class _Wrw(ctypes.Structure):    # struct sgv_conv_wrw_params
    _fields_ = [('dy', _vp), ('x', _vp), ('dw', _vp)] + [(k, _i32) for k in ('n', 'c_out', 'c_in', 'h', 'w', 'terms')] + \
               [('dy_amax', _vp), ('x_amax', _vp), ('x_amax2', _vp)]


def _get():
    global _lib
    if _lib is None:
        lib = ctypes.CDLL(_LIB_PATH)
        lib.sgv_last_error.restype = ctypes.c_char_p
        lib.sgv_conv3x3_workspace_bytes.restype = _i64
        lib.sgv_conv3x3_workspace_bytes.argtypes = [_i32, _i32]
        lib.sgv_conv3x3_s2_workspace_bytes.restype = _i64
        lib.sgv_conv3x3_s2_workspace_bytes.argtypes = [_i32] * 6
        for name in ('sgv_conv3x3', 'sgv_conv3x3_s2'):
            getattr(lib, name).argtypes = [ctypes.POINTER(_Conv), ctypes.c_int, _vp]
        for name in ('sgv_conv3x3_wrw', 'sgv_conv3x3_wrw_s2'):
            getattr(lib, name).argtypes = [ctypes.POINTER(_Wrw), ctypes.c_int, _vp]
        lib.sgv_absmax.argtypes = [_vp, _i64, ctypes.c_int, _vp, _i32, _vp]
        for name in ('sgv_conv3x3_supported', 'sgv_conv3x3_s2_supported', 'sgv_conv3x3_wrw_supported', 'sgv_conv3x3_wrw_s2_supported'):
            getattr(lib, name).argtypes = [_i32] * 5 + [ctypes.c_int]
        _lib = lib
    return _lib


def _run(fn, p, t):
    with torch.cuda.device_of(t):
        if fn(ctypes.byref(p), 0, _vp(torch.cuda.current_stream(t.device).cuda_stream)):
            raise RuntimeError(_get().sgv_last_error().decode(errors='replace'))


def _bound(t):
    """max |t| as a 1-element device tensor (the block scale of terms = 4 is derived from it on the device; nothing is read back)."""
    out = torch.empty([1], dtype=torch.float32, device=t.device)
    with torch.cuda.device_of(t):
        if _get().sgv_absmax(t.data_ptr(), t.numel(), 0, out.data_ptr(), 0, _vp(torch.cuda.current_stream(t.device).cuda_stream)):
            raise RuntimeError(_get().sgv_last_error().decode(errors='replace'))
    return out


def conv3x3(x, w, transposed, stride, terms=4):
    """3x3 fp32 NCHW, padding 1 at stride 1, padding 0 at stride 2; returns None when the shape is not served."""
    lib = _get()
    if x.dtype != torch.float32 or w.dtype != torch.float32 or tuple(w.shape[2:]) != (3, 3):
        return None
    x, w = x.contiguous(), w.contiguous()
    n, ci, h, wd = x.shape
    co = w.shape[1] if transposed else w.shape[0]
    if stride == 1:
        if not lib.sgv_conv3x3_supported(n, ci, co, h, wd, 0):
            return None
        y = x.new_empty([n, co, h, wd])
        nbytes, fn, mode, hs, ws = lib.sgv_conv3x3_workspace_bytes(ci, co), lib.sgv_conv3x3, int(transposed), h, wd
    else:
        hs, ws = (h, wd) if transposed else ((h - 1) // 2, (wd - 1) // 2)          # the small H x W grid
        if not transposed and (h % 2 == 0 or wd % 2 == 0):
            return None
        if not lib.sgv_conv3x3_s2_supported(n, ci, co, hs, ws, 0):
            return None
        y = x.new_empty([n, co, 2 * hs + 1, 2 * ws + 1] if transposed else [n, co, hs, ws])
        mode = 2 if transposed else 0
        nbytes, fn = lib.sgv_conv3x3_s2_workspace_bytes(n, ci, co, hs, ws, mode), lib.sgv_conv3x3_s2
    scratch = torch.empty([nbytes], dtype=torch.uint8, device=x.device)
    bx = _bound(x) if terms == 4 else None
    _run(fn, _Conv(x.data_ptr(), w.data_ptr(), y.data_ptr(), scratch.data_ptr(), nbytes, n, ci, co, hs, ws, mode, terms, bx.data_ptr() if bx is not None else None, None, None), x)
    return y


def conv3x3_weight_grad(small, big_or_x, w_shape, stride, terms=4):
    """stride 1: (dy, x) -> dw [c_out, c_in, 3, 3];  stride 2: (the H x W tensor, the (2H+1) x (2W+1) tensor) -> dw [c_small, c_big, 3, 3]."""
    lib = _get()
    small, big_or_x = small.contiguous(), big_or_x.contiguous()
    n, cs, h, wd = small.shape
    cb = big_or_x.shape[1]
    ok = (lib.sgv_conv3x3_wrw_supported if stride == 1 else lib.sgv_conv3x3_wrw_s2_supported)(n, cs, cb, h, wd, 0)
    if not ok or small.dtype != torch.float32:
        return None
    dw = small.new_empty(tuple(w_shape))
    bs, bb = (_bound(small), _bound(big_or_x)) if terms == 4 else (None, None)
    _run(lib.sgv_conv3x3_wrw if stride == 1 else lib.sgv_conv3x3_wrw_s2,
         _Wrw(small.data_ptr(), big_or_x.data_ptr(), dw.data_ptr(), n, cs, cb, h, wd, terms, bs.data_ptr() if bs is not None else None,
              bb.data_ptr() if bb is not None else None, None), small)
    return dw
