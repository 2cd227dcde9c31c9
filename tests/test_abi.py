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
    assert lib.sgv_version() == 105     # 1.05: sgv_ada_geometric_adjoint; 1.04: sgv_fc_grouped (the style affines of a synthesis pass as one launch); upfirdn2d kernel kinds 4 / 5 (2x LDS-tile kernels); 1.01: terms = 4 (block-scaled fp16 split), sgv_absmax, the *_amax fields of the convolution / GEMM parameter blocks; 1.02: sgv_ada_geometric; 1.03: sgv_prof_resume (per-launch timing inside graph replays), x_amax2 required with x_scale, 4x4 images
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
    assert ctypes.sizeof(custom_ops.GemmParams) == 144
    assert ctypes.sizeof(custom_ops.ProfEntry) == 32
    assert ctypes.sizeof(custom_ops.PointwiseParams) == 56
    assert ctypes.sizeof(custom_ops.ConvWrwParams) == 72
    assert ctypes.sizeof(custom_ops.Conv3x3Params) == 96


def test_convolution_family_shape_rules_and_validation_without_gpu():
    """Eligibility predicates, workspace sizes and precondition errors of the 3x3 convolution entry points are host-side."""
    lib = custom_ops.get_native()
    F32, F16 = custom_ops.SGV_F32, custom_ops.SGV_F16
    # stride 1: c_in % 16, c_out % 64, (W % 32 and H % 16) or whole 16x16 / 8x8 images in pairs / octets
    assert lib.sgv_conv3x3_supported(96, 64, 64, 256, 256, F32) == 1
    assert lib.sgv_conv3x3_supported(96, 16, 512, 32, 32, F32) == 1
    assert lib.sgv_conv3x3_supported(96, 512, 512, 16, 16, F32) == 1
    assert lib.sgv_conv3x3_supported(32, 512, 512, 8, 8, F32) == 1
    assert lib.sgv_conv3x3_supported(31, 512, 512, 16, 16, F32) == 1      # (round 5: a partly filled last tile)
    assert lib.sgv_conv3x3_supported(96, 512, 512, 4, 4, F32) == 1        # (round 5: 4 x 4 images, 32 per tile)
    assert lib.sgv_conv3x3_supported(96, 512, 512, 4, 8, F32) == 0
    assert lib.sgv_conv3x3_supported(96, 3, 64, 256, 256, F32) == 0
    assert lib.sgv_conv3x3_supported(96, 64, 48, 256, 256, F32) == 0
    # ... or c_out % 32 on the big-image (producer / consumer) kernel: a half-full last tile (the 32-channel layers of the 1024^2 synthesis network)
    assert lib.sgv_conv3x3_supported(16, 64, 32, 1024, 1024, F32) == 1
    assert lib.sgv_conv3x3_supported(96, 32, 96, 256, 256, custom_ops.SGV_BF16) == 1
    assert lib.sgv_conv3x3_supported(96, 512, 32, 16, 16, F32) == 0
    assert lib.sgv_conv3x3_fused_supported(16, 32, 32, 1024, 1024, F32) == 1
    assert lib.sgv_conv3x3_workspace_bytes(64, 32) == 64 * 64 * 9 * 4 + 16      # whole 64-row tiles + the weight bound of the block-scaled split
    # 16-bit tensors (fp32 weights and accumulate): the producer / consumer kernel of the big images only
    assert lib.sgv_conv3x3_supported(96, 64, 64, 256, 256, F16) == 1
    assert lib.sgv_conv3x3_supported(96, 64, 64, 256, 256, custom_ops.SGV_BF16) == 1
    assert lib.sgv_conv3x3_supported(96, 512, 512, 16, 16, F16) == 0
    assert lib.sgv_conv3x3_supported(96, 64, 64, 256, 256, custom_ops.SGV_F64) == 0
    assert lib.sgv_conv3x3_workspace_bytes(64, 128) == 64 * 128 * 9 * 4 + 16
    # stride 2 (h, w = the small grid): W % 32, H % 8
    assert lib.sgv_conv3x3_s2_supported(96, 64, 128, 128, 128, F32) == 1
    assert lib.sgv_conv3x3_s2_supported(96, 64, 128, 8, 32, F32) == 1
    assert lib.sgv_conv3x3_s2_supported(96, 64, 128, 16, 16, F32) == 0
    assert lib.sgv_conv3x3_s2_supported(96, 64, 128, 12, 32, F32) == 0
    assert lib.sgv_conv3x3_s2_supported_mode(16, 64, 32, 512, 512, 2, F32) == 1      # transposed form: c_out % 32 (half-full last tile)
    assert lib.sgv_conv3x3_s2_supported_mode(16, 64, 32, 512, 512, 0, F32) == 0      # the strided form keeps whole tiles
    ws_strided = lib.sgv_conv3x3_s2_workspace_bytes(4, 64, 128, 16, 32, 0)
    ws_transposed = lib.sgv_conv3x3_s2_workspace_bytes(4, 64, 128, 16, 32, 2)
    assert ws_strided == 64 * 128 * 10 * 4 + 16   # ten taps: the tap-pair layout of the strided kernel pads the ninth pair
    assert ws_transposed == ws_strided + 4 * (4 * 64 * (16 + 32) + 2 * 64 * 3 * 128)   # + edge lines + edge weights
    # weight gradients: channels % 64
    assert lib.sgv_conv3x3_wrw_supported(96, 64, 64, 256, 256, F32) == 1
    assert lib.sgv_conv3x3_wrw_supported(96, 64, 16, 256, 256, F32) == 0
    assert lib.sgv_conv3x3_wrw_supported(96, 64, 64, 48, 32, F32) == 0
    assert lib.sgv_conv3x3_wrw_s2_supported(96, 128, 64, 128, 128, F32) == 1
    assert lib.sgv_conv3x3_wrw_s2_supported(96, 128, 64, 128, 16, F32) == 0
    # preconditions are reported before anything is launched
    p = custom_ops.Conv3x3Params()
    assert lib.sgv_conv3x3(p, F32, None) == -1 and b'NULL' in lib.sgv_last_error()
    buf = (ctypes.c_float * 64)()
    addr = ctypes.addressof(buf)
    p.x = p.weight = p.y = p.workspace = addr
    p.n, p.c_in, p.c_out, p.h, p.w, p.mode, p.terms = 2, 64, 64, 20, 32, 0, 3
    assert lib.sgv_conv3x3(p, F32, None) == -3 and b'H % 16' in lib.sgv_last_error()
    p.h = 16
    p.terms = 2
    assert lib.sgv_conv3x3(p, F32, None) == -1 and b'terms' in lib.sgv_last_error()
    p.terms = 4                                    # the block-scaled split needs the operand's bound
    assert lib.sgv_conv3x3(p, F32, None) == -1 and b'x_amax' in lib.sgv_last_error()
    p.terms, p.workspace_bytes = 3, 16
    assert lib.sgv_conv3x3(p, F32, None) == -1 and b'workspace' in lib.sgv_last_error()
    p.mode, p.workspace_bytes = 1, 1 << 30
    assert lib.sgv_conv3x3_s2(p, F32, None) == -3 or b'mode' in lib.sgv_last_error()
    q = custom_ops.ConvWrwParams()
    assert lib.sgv_conv3x3_wrw(q, F32, None) == -1
    q.dy = q.x = q.dw = addr
    q.n, q.c_out, q.c_in, q.h, q.w, q.terms = 2, 64, 64, 32, 24, 3
    assert lib.sgv_conv3x3_wrw(q, F32, None) == -3 and b'W % 32' in lib.sgv_last_error()
    assert lib.sgv_conv3x3_wrw_s2(q, F32, None) == -3
    q.w, q.terms = 32, 4
    assert lib.sgv_conv3x3_wrw(q, F32, None) == -1 and b'dy_amax' in lib.sgv_last_error()
    assert lib.sgv_absmax(None, 4, F32, None, 0, None) == -1
    assert lib.sgv_ada_geometric(addr, addr, addr, addr, 1, 1, 8, 8, 8, 0, 0, 0, None) == -1 and b'margin' in lib.sgv_last_error()
    r = custom_ops.PointwiseParams()
    assert lib.sgv_pointwise_small(r, F32, None) == -1
    assert lib.sgv_bias_act_db(custom_ops.BiasActParams(), None, 1, F32, None) == -1 and b'db is NULL' in lib.sgv_last_error()


def test_profiling_families_match_the_header_enum():
    """sgv_prof_collect() writes SGV_K_COUNT entries into the array the Python side sizes from SGV_K_NAMES: the two lists must agree, name by name."""
    import os
    import re
    header = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'include', 'sgv_ops.h')).read()
    enum = re.findall(r'SGV_K_([A-Z0-9_]+)\s*=\s*(\d+)', header)
    count = [int(v) for k, v in enum if k == 'COUNT'][0]
    names = {int(v): k.lower() for k, v in enum if k != 'COUNT'}
    assert count == len(custom_ops.SGV_K_NAMES) == len(names)
    for idx, name in enumerate(custom_ops.SGV_K_NAMES):
        assert names[idx] == name, (idx, names[idx], name)
