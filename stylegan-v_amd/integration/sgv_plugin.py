"""Drop-in for the reference's pybind plugin module (``_plugin`` in src/torch_utils/ops/upfirdn2d.py:24-31 and
src/torch_utils/ops/bias_act.py:38-47): the same two functions with the same positional signatures
(``upfirdn2d.cpp:16``, ``bias_act.cpp:32``), on top of the C ABI of include/sgv_ops.h.

Reference-side edit: in ``upfirdn2d.py:_init`` / ``bias_act.py:_init`` replace the ``custom_ops.get_plugin(...)`` call
(src/torch_utils/custom_ops.py:46) by ``from . import sgv_plugin as _plugin``.  The reference's autograd classes
(upfirdn2d.py:228-264, bias_act.py:145-206) then work unchanged: first and second order gradients reach the same kernels
through their ``grad = 1 / 2`` and swapped up/down forms.
"""
import ctypes
import os

import torch

_LIB_PATH = os.environ.get('SGV_HIP_LIB') or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'csrc', 'libsgv_hip.so')
_lib = None
_DT = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2, torch.float64: 3}
_i32, _i64, _f32, _vp = ctypes.c_int32, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p


class _UfdParams(ctypes.Structure):      # struct sgv_upfirdn2d_params (== upfirdn2d.h:14-40 with strides spelled out)
    _fields_ = [('x', _vp), ('f', _vp), ('y', _vp), ('up_x', _i32), ('up_y', _i32), ('down_x', _i32), ('down_y', _i32),
                ('pad_x0', _i32), ('pad_x1', _i32), ('pad_y0', _i32), ('pad_y1', _i32), ('flip', _i32), ('gain', _f32),
                ('in_w', _i32), ('in_h', _i32), ('in_c', _i32), ('in_n', _i32), ('in_sw', _i64), ('in_sh', _i64), ('in_sc', _i64), ('in_sn', _i64),
                ('f_w', _i32), ('f_h', _i32), ('f_sw', _i64), ('f_sh', _i64), ('out_w', _i32), ('out_h', _i32),
                ('out_sw', _i64), ('out_sh', _i64), ('out_sc', _i64), ('out_sn', _i64)]


class _BaParams(ctypes.Structure):       # struct sgv_bias_act_params (== bias_act.h:12-31)
    _fields_ = [(n, _vp) for n in ('x', 'b', 'xref', 'yref', 'dy', 'y')] + \
               [('grad', _i32), ('act', _i32), ('alpha', _f32), ('gain', _f32), ('clamp', _f32), ('size_x', _i32), ('size_b', _i32), ('step_b', _i32)]


def _get():
    global _lib
    if _lib is None:
        lib = ctypes.CDLL(_LIB_PATH)     # OSError if the library is missing: no silent fallback
        lib.sgv_last_error.restype = ctypes.c_char_p
        lib.sgv_upfirdn2d.argtypes = [ctypes.POINTER(_UfdParams), ctypes.c_int, _vp]
        lib.sgv_bias_act.argtypes = [ctypes.POINTER(_BaParams), ctypes.c_int, _vp]
        _lib = lib
    return _lib


def _check(rc):
    if rc:
        raise RuntimeError(_get().sgv_last_error().decode(errors='replace'))


def _stream(t):
    return _vp(torch.cuda.current_stream(t.device).cuda_stream)


def upfirdn2d(x, f, upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip, gain):
    """== ``upfirdn2d(x, f, upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip, gain)`` of upfirdn2d.cpp:16."""
    n, c, h, w = x.shape
    fh, fw = f.shape
    ow = (w * upx + padx0 + padx1 - fw + downx) // downx          # upfirdn2d.cpp:32-33
    oh = (h * upy + pady0 + pady1 - fh + downy) // downy
    if ow < 1 or oh < 1:
        raise RuntimeError('output must be at least 1x1')
    channels_last = c > 1 and x.stride(1) == 1 and x.is_contiguous(memory_format=torch.channels_last)
    y = torch.empty([n, c, oh, ow], dtype=x.dtype, device=x.device, memory_format=torch.channels_last if channels_last else torch.contiguous_format)
    f = f.float()
    p = _UfdParams(x.data_ptr(), f.data_ptr(), y.data_ptr(), upx, upy, downx, downy, padx0, padx1, pady0, pady1, int(flip), gain,
                   w, h, c, n, x.stride(3), x.stride(2), x.stride(1), x.stride(0), fw, fh, f.stride(1), f.stride(0),
                   ow, oh, y.stride(3), y.stride(2), y.stride(1), y.stride(0))
    with torch.cuda.device_of(x):
        _check(_get().sgv_upfirdn2d(ctypes.byref(p), _DT[x.dtype], _stream(x)))
    return y


def bias_act(x, b, xref, yref, dy, grad, dim, act, alpha, gain, clamp):
    """== ``bias_act(x, b, xref, yref, dy, grad, dim, act, alpha, gain, clamp)`` of bias_act.cpp:32 (empty tensor = absent)."""
    def ptr(t):
        return t.data_ptr() if t.numel() else None
    y = torch.empty_like(x)
    p = _BaParams(x.data_ptr(), ptr(b), ptr(xref), ptr(yref), ptr(dy), y.data_ptr(), grad, act, alpha, gain, clamp,
                  x.numel(), b.numel(), x.stride(dim) if b.numel() else 1)      # bias_act.cpp:73-75
    if x.numel():
        with torch.cuda.device_of(x):
            _check(_get().sgv_bias_act(ctypes.byref(p), _DT[x.dtype], _stream(x)))
    return y
