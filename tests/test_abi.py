"""The C-ABI library loads (no GPU needed) and exports exactly what include/sgv_ops.h declares."""
import ctypes
import os
import re

from stylegan_v_amd.torch_utils import custom_ops

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, 'include', 'sgv_ops.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(sgv_[a-z0-9_]+)\s*\(', text)))


def test_header_functions_all_exported_and_bound():
    lib = custom_ops.get_native()
    names = _declared_functions()
    assert len(names) >= 14
    for name in names:
        assert hasattr(lib, name), f'{name} declared in include/sgv_ops.h but not exported'
    assert sorted(custom_ops.ABI_SYMBOLS) == names, 'ctypes bindings out of sync with the header'


def test_version_and_error_string():
    lib = custom_ops.get_native()
    assert lib.sgv_version() == 100
    assert isinstance(lib.sgv_last_error(), bytes)
    assert lib.sgv_launch_count() >= 0


def test_argument_validation_without_gpu():
    """Precondition failures are reported through the ABI before any launch (upfirdn2d.cpp:19-36)."""
    lib = custom_ops.get_native()
    p = custom_ops.Upfirdn2dParams()
    assert lib.sgv_upfirdn2d(p, custom_ops.SGV_F32, None) == -1  # NULL pointers
    assert b'non-NULL' in lib.sgv_last_error()
    buf = (ctypes.c_float * 64)()
    addr = ctypes.addressof(buf)
    p.x = p.f = p.y = addr
    p.up_x = p.up_y = p.down_x = p.down_y = 1
    p.in_w, p.in_h, p.in_c, p.in_n = 4, 4, 1, 1
    p.f_w, p.f_h = 8, 8
    assert lib.sgv_upfirdn2d(p, custom_ops.SGV_F32, None) == -1  # output would be < 1x1
    assert b'at least 1x1' in lib.sgv_last_error()
    p.f_w = p.f_h = 2
    p.up_x = 0
    assert lib.sgv_upfirdn2d(p, custom_ops.SGV_F32, None) == -1
    assert b'upsampling factor' in lib.sgv_last_error()
    q = custom_ops.BiasActParams()
    q.x = q.y = addr
    q.size_x, q.act = 16, 11
    assert lib.sgv_bias_act(q, custom_ops.SGV_F32, None) == -3
    assert b'activation' in lib.sgv_last_error()
    assert lib.sgv_upfirdn2d(p, 17, None) == -3  # unknown dtype


def test_struct_layout_matches_header():
    # sizes a C compiler gives the structs of include/sgv_ops.h (LP64)
    assert ctypes.sizeof(custom_ops.Upfirdn2dParams) == 176
    assert ctypes.sizeof(custom_ops.BiasActParams) == 80
    assert ctypes.sizeof(custom_ops.TimeEncodeParams) == 96
    assert ctypes.sizeof(custom_ops.GemmParams) == 112
    assert ctypes.sizeof(custom_ops.ProfEntry) == 32
    assert ctypes.sizeof(custom_ops.PointwiseParams) == 56
    assert ctypes.sizeof(custom_ops.ConvWrwParams) == 48
    assert ctypes.sizeof(custom_ops.Conv3x3Params) == 72
