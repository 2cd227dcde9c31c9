"""Parity tests proper: the HIP kernels, called through the C ABI, against the CPU oracle (bit-exact
where the arithmetic is exactly rounded) and against the reference's golden vectors.  GPU only."""
import numpy as np
import pytest
import torch

import oracle
from stylegan_v_amd.torch_utils import custom_ops
from stylegan_v_amd.torch_utils.ops import bias_act as ba
from stylegan_v_amd.torch_utils.ops import upfirdn2d as ufd
from util import Golden, assert_bit_equal, assert_close, dispatch_assert

pytestmark = pytest.mark.gpu
DEV = 'cuda'
UFD = Golden('upfirdn2d')
BA = Golden('bias_act')


def _gf(i, device='cpu'):
    return UFD.t(f'c{i}_f', device=device) if UFD.meta[i]['has_f'] else None


def _kind(x, f2d, up, down, padding):
    """Which kernel sgv_upfirdn2d picks (1 = row walker, 0 = generic) for a contiguous x."""
    lib = custom_ops.get_native()
    upx, upy = ufd._parse_scaling(up)
    dnx, dny = ufd._parse_scaling(down)
    px0, px1, py0, py1 = ufd._parse_padding(padding)
    n, c, h, w = x.shape
    fh, fw = f2d.shape
    p = custom_ops.Upfirdn2dParams()
    y = torch.empty([n, c, ufd.output_size(h, upy, dny, py0, py1, fh), ufd.output_size(w, upx, dnx, px0, px1, fw)], dtype=x.dtype, device=x.device)
    p.x, p.f, p.y = x.data_ptr(), f2d.data_ptr(), y.data_ptr()
    p.up_x, p.up_y, p.down_x, p.down_y = upx, upy, dnx, dny
    p.pad_x0, p.pad_x1, p.pad_y0, p.pad_y1 = px0, px1, py0, py1
    p.in_w, p.in_h, p.in_c, p.in_n = w, h, c, n
    p.in_sn, p.in_sc, p.in_sh, p.in_sw = x.stride()
    p.f_w, p.f_h = fw, fh
    p.f_sh, p.f_sw = f2d.stride()
    p.out_w, p.out_h = y.shape[3], y.shape[2]
    p.out_sn, p.out_sc, p.out_sh, p.out_sw = y.stride()
    return lib.sgv_upfirdn2d_kernel_kind(p, ufd._DTYPE_CODES[x.dtype])


def test_native_library_is_what_runs():
    before = custom_ops.launch_count()
    x = torch.randn([1, 2, 8, 8], device=DEV)
    ufd.upfirdn2d(x, ufd.setup_filter([1, 3, 3, 1], device=DEV), padding=1)
    ba.bias_act(x, torch.zeros([2], device=DEV), act='lrelu')
    assert custom_ops.native_loaded()
    assert custom_ops.launch_count() == before + 2


# ------------------------------------------------------------------------------------------------
# upfirdn2d


@pytest.mark.parametrize('i', range(len(UFD.meta)))
@pytest.mark.parametrize('dtype', [torch.float32, torch.float64, torch.float16, torch.bfloat16])
def test_upfirdn2d_golden_cases_bit_exact_vs_oracle(i, dtype):
    """Every golden configuration (hot-path shapes, random up/down/pad/crop/flip, separable), every
    dtype: HIP output == oracle output bit for bit (same tap order, explicit fma, RNE stores)."""
    m = UFD.meta[i]
    x = UFD.t(f'c{i}_x', dtype)
    kw = dict(up=m['up'], down=m['down'], padding=m['padding'], flip_filter=m['flip'], gain=m['gain'])
    f = _gf(i)
    y = ufd.upfirdn2d(x.to(DEV), None if f is None else f.to(DEV), **kw)
    assert_bit_equal(y, oracle.upfirdn2d(x, f, **kw), what=f'case {i} {dtype}')
    if dtype in (torch.float32, torch.float64):  # and within fp32 round-off of the reference's own output
        tol = 2e-5 if dtype == torch.float32 else 5e-6
        assert_close(y, UFD.t(f'c{i}_y'), atol=tol, rtol=tol, what=f'case {i} vs reference')


@pytest.mark.parametrize('i', range(len(UFD.meta)))
def test_upfirdn2d_gradients_vs_reference_autograd(i):
    """First derivative and the second-order term through the HIP op's autograd vs reference autograd."""
    m = UFD.meta[i]
    kw = dict(up=m['up'], down=m['down'], padding=m['padding'], flip_filter=m['flip'], gain=m['gain'])
    f = _gf(i, DEV)
    x = UFD.t(f'c{i}_x', device=DEV).requires_grad_(True)  # fp64
    dy = UFD.t(f'c{i}_dy', device=DEV).requires_grad_(True)
    y = ufd.upfirdn2d(x, f, **kw)
    (dx,) = torch.autograd.grad(y, x, dy, create_graph=True)
    assert_close(dx, UFD.t(f'c{i}_dx'), atol=5e-6, rtol=1e-6, what='dx')
    (ddy,) = torch.autograd.grad((dx * UFD.t(f'c{i}_v', device=DEV)).sum(), dy)
    assert_close(ddy, UFD.t(f'c{i}_ddy'), atol=5e-6, rtol=1e-6, what='ddy')


def test_upfirdn2d_gradgradcheck():
    f = ufd.setup_filter([1, 3, 3, 1], device=DEV)
    x = torch.randn([1, 2, 6, 6], dtype=torch.float64, device=DEV, requires_grad=True)
    for kw in (dict(up=2, padding=[2, 1, 2, 1], gain=4), dict(down=2, padding=1), dict(padding=[2, 2, 2, 2])):
        assert torch.autograd.gradcheck(lambda t: ufd.upfirdn2d(t, f, **kw), (x,))
        assert torch.autograd.gradgradcheck(lambda t: ufd.upfirdn2d(t, f, **kw), (x,))


ROWS_CONFIGS = [
    # (up, down, padding, gain): the hot-path calls and their backward counterparts
    (1, 1, [1, 1, 1, 1], 4), (1, 1, [2, 2, 2, 2], 4), (2, 1, [2, 1, 2, 1], 4), (1, 2, [1, 1, 1, 1], 1),
    (2, 1, [1, 2, 1, 2], 1), (2, 1, [3, 0, 2, 1], 1), (1, 2, [2, 0, 0, 2], 1), (1, 1, [0, 3, -1, 2], 1), (1, 2, [-1, 3, 2, -2], 2),
]
ROWS_SHAPES = [(1, 1, 5, 5), (1, 3, 4, 4), (2, 2, 8, 8), (1, 1, 6, 12), (3, 2, 7, 6), (2, 3, 9, 9), (1, 2, 16, 16), (3, 5, 17, 33), (1, 2, 64, 63), (2, 2, 37, 129), (1, 3, 19, 131), (1, 1, 65, 255),
               (2, 1, 31, 256), (1, 2, 33, 257), (1, 1, 12, 258), (1, 1, 9, 261), (1, 1, 40, 513), (1, 1, 7, 1025), (1, 1, 3, 300), (1, 1, 1, 140),
               # output rows of 40 / 72 / 192 / 200 / 320 (+1) / 1000 columns in the two FIR geometries: column blocks that do not fill the lanes of a plane
               # row -- the 16-bit tile kernel's eight-column lanes meet a window that overhangs the row's end by up to seven columns there
               (1, 2, 7, 41), (1, 1, 5, 73), (2, 1, 6, 193), (1, 1, 9, 201), (1, 1, 5, 322), (1, 1, 4, 1001), (1, 2, 7, 39), (1, 1, 5, 191), (1, 1, 6, 199),
               (1, 1, 5, 320)]


@pytest.mark.parametrize('cfg', ROWS_CONFIGS)
@pytest.mark.parametrize('dtype', [torch.float32, torch.float16, torch.bfloat16])
def test_upfirdn2d_row_walker_shapes_bit_exact(cfg, dtype):
    """The fast kernel across widths that straddle lane / wave / vector boundaries (incl. the
    out_w % 4 == 1 ragged column), odd heights, crops, both flips, filters smaller than 4x4."""
    up, down, padding, gain = cfg
    g = torch.Generator().manual_seed(sum(map(ord, str(cfg))) % 1000)
    filters = [ufd.setup_filter([1, 3, 3, 1]), torch.randn([4, 4], generator=g), torch.randn([3, 2], generator=g), torch.randn([1, 4], generator=g)]
    for shape in ROWS_SHAPES:
        x = torch.randn(shape, generator=g).to(dtype)
        for fi, f in enumerate(filters):
            for flip in (False, True):
                ow = ufd.output_size(shape[3], up, down, padding[0], padding[1], f.shape[1])
                oh = ufd.output_size(shape[2], up, down, padding[2], padding[3], f.shape[0])
                if ow < 1 or oh < 1:
                    continue
                xg, fg = x.to(DEV), f.to(DEV)
                kind = _kind(xg, fg, up, down, padding)
                hot = (up, down, padding[0], padding[2]) in ((1, 1, 1, 1), (1, 1, 2, 2), (2, 1, 2, 2), (1, 2, 1, 1))
                # 3 = LDS tile, 4 / 5 = 2x down / up LDS tile (any pad 0..3), 2 = lane-exchange, 1 = row walker; never the generic gather kernel (0)
                assert kind in ((2, 3, 4, 5) if hot else (1, 4, 5)), 'unexpected kernel selection'
                if down == 2 and up == 1 and 8 <= ow <= 128 and shape[3] >= 4 and min(padding[0], padding[2]) >= 0 and max(padding[0], padding[2]) <= 3:
                    dispatch_assert(kind == 4, '2x down-sampling with 8..128 output columns goes to the down2 tile kernel')
                if up == 2 and down == 1 and 9 <= ow <= 256 and shape[3] >= 4 and min(padding[0], padding[2]) >= 0 and max(padding[0], padding[2]) <= 3:
                    dispatch_assert(kind == 5, '2x up-sampling with 9..256 output columns goes to the up2 tile kernel')
                if up == 1 and down == 1 and hot and ow % 4 in (0, 1) and ow > 4 and shape[3] >= 4:
                    dispatch_assert(kind == 3, 'the FIR geometries with whole output quads go to the tile kernel')
                y = ufd.upfirdn2d(xg, fg, up=up, down=down, padding=padding, flip_filter=flip, gain=gain)
                ref = oracle.upfirdn2d(x, f, up=up, down=down, padding=padding, flip_filter=flip, gain=gain)
                assert_bit_equal(y, ref, what=f'{cfg} {shape} filter#{fi} flip={flip} {dtype} kind={kind}')


def test_upfirdn2d_tile_kernel_eight_columns_per_lane_form_in_a_child_process():
    """SGV_UFD_TILE_CPL8=1 (read once per process): the 16-bit tile kernel with eight output columns per lane, an opt-in form -- the width sweep above,
    bf16 and fp16, bit-exact against the oracle in a child interpreter that has the switch set."""
    import os
    import subprocess
    import sys
    if os.environ.get('SGV_UFD_TILE_CPL8') == '1':
        pytest.skip('already inside the child')
    env = dict(os.environ, SGV_UFD_TILE_CPL8='1')
    res = subprocess.run([sys.executable, '-m', 'pytest', os.path.abspath(__file__), '-q', '-x', '-m', 'gpu', '-k', 'row_walker_shapes', '-p', 'no:cacheprovider'],
                         env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-2000:]
    assert ' passed' in res.stdout


SYM6 = [0.015404109327027373, 0.0034907120842174702, -0.11799011114819057, -0.048311742585633, 0.4910559419267466, 0.787641141030194,
        0.3379294217276218, -0.07263752278646252, -0.021060292512300564, 0.04472490177066578, 0.0017677118642428036, -0.007800708325034148]


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16, torch.bfloat16])
def test_upfirdn2d_ada_separable_passes_bit_exact(dtype):
    """The ADA pipe's resampling (augment.py:289,300): separable 12-tap 'sym6' up x2 then, after the warp, down x2 with
    a crop (negative padding) and a flipped filter.  Each 1-D pass must select the row-walker kernel and match the oracle."""
    g = torch.Generator().manual_seed(21)
    f = torch.tensor(SYM6)
    for shape in ((2, 9, 76, 76), (1, 3, 33, 140), (1, 2, 268, 268)):
        x = torch.randn(shape, generator=g).to(dtype)
        xg, fg = x.to(DEV), f.to(DEV)
        # one-dimensional passes as the native layer sees them
        for up, down, pad, f2 in (((2, 1), 1, [6, 5, 0, 0], f.unsqueeze(0)), ((1, 2), 1, [0, 0, 6, 5], f.unsqueeze(1)),
                                  (1, (2, 1), [-1, -1, 0, 0], f.unsqueeze(0)), (1, (1, 2), [0, 0, -1, -1], f.unsqueeze(1)),
                                  (1, 1, [6, 5, 0, 0], f.unsqueeze(0)), (1, 1, [0, 0, 5, 6], f.unsqueeze(1))):
            assert _kind(xg, f2.to(DEV).contiguous(), up, down, pad) == 1
            for flip in (False, True):
                y = ufd.upfirdn2d(xg, f2.to(DEV), up=up, down=down, padding=pad, flip_filter=flip, gain=2)
                assert_bit_equal(y, oracle.upfirdn2d(x, f2, up=up, down=down, padding=pad, flip_filter=flip, gain=2), what=f'{shape} up={up} down={down} {dtype}')
        # and the public separable entry points (two launches each)
        before = custom_ops.launch_count()
        up = ufd.upsample2d(xg, fg, up=2, padding=2)
        assert custom_ops.launch_count() == before + 2
        assert_bit_equal(up, oracle.upfirdn2d(x, f, up=2, padding=[8, 7, 8, 7], gain=4), what='upsample2d sym6')
        dn = ufd.downsample2d(up, fg, down=2, padding=-4, flip_filter=True)
        ref_dn = oracle.upfirdn2d(up.cpu(), f, down=2, padding=[1, 1, 1, 1], flip_filter=True)
        assert_bit_equal(dn, ref_dn, what='downsample2d sym6')


def test_upfirdn2d_generic_layouts():
    """channels_last, sliced (non-dense) inputs and fp64 go to the generic kernel and still match."""
    g = torch.Generator().manual_seed(3)
    f = torch.randn([5, 3], generator=g)
    x = torch.randn([2, 6, 11, 13], generator=g)
    kw = dict(up=(2, 1), down=(1, 3), padding=[2, 3, 1, 0], gain=1.5)
    ref = oracle.upfirdn2d(x, f, **kw)
    ycl = ufd.upfirdn2d(x.to(DEV).contiguous(memory_format=torch.channels_last), f.to(DEV), **kw)
    assert ycl.is_contiguous(memory_format=torch.channels_last)
    assert_bit_equal(ycl.contiguous(), ref, what='channels_last')
    big = torch.randn([2, 6, 11, 26], generator=g)
    view = big.to(DEV)[:, :, :, ::2]
    assert_bit_equal(ufd.upfirdn2d(view, f.to(DEV), **kw), oracle.upfirdn2d(big[:, :, :, ::2].contiguous(), f, **kw), what='strided view')


def test_upfirdn2d_errors():
    x = torch.randn([1, 1, 4, 4], device=DEV)
    with pytest.raises(RuntimeError, match='at least 1x1'):
        ufd.upfirdn2d(x, torch.ones([6, 6], device=DEV))
    with pytest.raises(RuntimeError, match='same device'):
        ufd.upfirdn2d(x, torch.ones([2, 2]))
    with pytest.raises(RuntimeError, match='float32'):
        ufd.upfirdn2d(x, torch.ones([2, 2], device=DEV, dtype=torch.float64))
    with pytest.raises(AssertionError):
        ufd.upfirdn2d(x, None, up=0)


def test_upfirdn2d_full_size_properties():
    """BASELINE sizes ([32,64,257,257] -> [32,64,256,256], fp32): oracle on a slab of planes, plus
    size-independent properties: linearity and adjointness <A x, v> == <x, A^T v>."""
    g = torch.Generator(device=DEV).manual_seed(5)
    f = ufd.setup_filter([1, 3, 3, 1], device=DEV)
    x = torch.randn([32, 64, 257, 257], generator=g, device=DEV)
    kw = dict(padding=1, gain=4)
    y = ufd.upfirdn2d(x, f, **kw)
    assert y.shape == (32, 64, 256, 256)
    for n, c in ((0, 0), (17, 33), (31, 63)):
        ref = oracle.upfirdn2d(x[n:n + 1, c:c + 1].cpu(), f.cpu(), **kw)
        assert_bit_equal(y[n:n + 1, c:c + 1], ref, what=f'plane ({n},{c})')
    x2 = torch.randn(x.shape, generator=g, device=DEV)
    lin = ufd.upfirdn2d(x + 0.5 * x2, f, **kw) - (y + 0.5 * ufd.upfirdn2d(x2, f, **kw))
    assert lin.abs().max().item() < 1e-4
    del lin, x2
    xr = x.requires_grad_(True)
    v = torch.randn(y.shape, generator=g, device=DEV)
    yr = ufd.upfirdn2d(xr, f, **kw)
    (atv,) = torch.autograd.grad(yr, xr, v)
    lhs = (yr.detach().double() * v.double()).sum().item()
    rhs = (x.detach().double() * atv.double()).sum().item()
    assert abs(lhs - rhs) <= 1e-6 * max(abs(lhs), 1.0) + 1e-3 * np.sqrt(y.numel()) * 1e-3


# ------------------------------------------------------------------------------------------------
# bias_act

EXACT_ACTS = ('linear', 'relu', 'lrelu')


@pytest.mark.parametrize('i', range(len(BA.meta)))
@pytest.mark.parametrize('dtype', [torch.float32, torch.float64, torch.float16, torch.bfloat16])
def test_bias_act_forward_vs_oracle_and_reference(i, dtype):
    m = BA.meta[i]
    kw = dict(dim=m['dim'], act=m['act'], alpha=m['alpha'], gain=m['gain'], clamp=m['clamp'])
    x = BA.t(f'c{i}_x', dtype)
    b = BA.t(f'c{i}_b', dtype) if m['has_b'] else None
    y = ba.bias_act(x.to(DEV), None if b is None else b.to(DEV), **kw)
    ref = oracle.bias_act(x, b, **kw)
    if m['act'] in EXACT_ACTS:
        assert_bit_equal(y, ref, what=f"{m['act']} {dtype}")
    else:  # exp/log differ between OCML and glibc by a few ulp
        tol = {torch.float32: 2e-6, torch.float64: 1e-12, torch.float16: 2e-3, torch.bfloat16: 2e-2}[dtype]
        assert_close(y, ref, atol=tol, rtol=tol, what=f"{m['act']} {dtype}")
    if dtype == torch.float32:
        assert_close(y, BA.t(f'c{i}_y'), atol=1e-5, rtol=1e-5, what='vs reference')


@pytest.mark.parametrize('i', range(len(BA.meta)))
def test_bias_act_gradients_vs_reference_autograd(i):
    m = BA.meta[i]
    if m['act'] == 'linear' and m['clamp'] is not None:
        pytest.skip('reference quirk: native linear+clamp gradient is unmasked (see tests/test_oracle.py)')
    kw = dict(dim=m['dim'], act=m['act'], alpha=m['alpha'], gain=m['gain'], clamp=m['clamp'])
    x = BA.t(f'c{i}_x', device=DEV).requires_grad_(True)  # fp64
    b = BA.t(f'c{i}_b', device=DEV).requires_grad_(True) if m['has_b'] else None
    dy = BA.t(f'c{i}_dy', device=DEV).requires_grad_(True)
    y = ba.bias_act(x, b, **kw)
    ins = [x] + ([b] if b is not None else [])
    grads = torch.autograd.grad(y, ins, dy, create_graph=True)
    edge = ((y.detach().abs() - (m['clamp'] if m['clamp'] is not None else float('inf'))).abs() < 1e-6).cpu()
    assert_close(grads[0].detach().cpu()[~edge], BA.t(f'c{i}_dx')[~edge], atol=2e-6, rtol=2e-6, what='dx')
    if b is not None:
        assert_close(grads[1], BA.t(f'c{i}_db'), atol=2e-5, rtol=2e-6, what='db')
    second = torch.autograd.grad((grads[0] * BA.t(f'c{i}_w', device=DEV)).sum(), [dy, x], allow_unused=True)
    assert_close(second[0].cpu()[~edge], BA.t(f'c{i}_ddy')[~edge], atol=2e-6, rtol=2e-6, what='ddy')
    ddx = second[1] if second[1] is not None else torch.zeros_like(x)
    assert_close(ddx.cpu()[~edge], BA.t(f'c{i}_ddx')[~edge], atol=2e-6, rtol=2e-6, what='ddx')


def test_bias_act_layouts_tails_and_fc():
    g = torch.Generator().manual_seed(8)
    for shape, dim in (((3, 7, 5, 5), 1), ((5, 513), 1), ((2, 3, 1027), 2), ((1, 1, 1, 3), 1), ((4, 6, 8, 8), 1)):
        x = torch.randn(shape, generator=g)
        b = torch.randn([shape[dim]], generator=g)
        y = ba.bias_act(x.to(DEV), b.to(DEV), dim=dim, act='lrelu', clamp=0.9)
        assert_bit_equal(y, oracle.bias_act(x, b, dim=dim, act='lrelu', clamp=0.9), what=str(shape))
    x = torch.randn([2, 8, 6, 6], generator=g)
    b = torch.randn([8], generator=g)
    xcl = x.to(DEV).contiguous(memory_format=torch.channels_last)
    y = ba.bias_act(xcl, b.to(DEV), act='relu')
    assert y.is_contiguous(memory_format=torch.channels_last)
    assert_bit_equal(y.contiguous(), oracle.bias_act(x, b, act='relu'), what='channels_last')


def test_bias_act_full_size():
    """[32,64,256,256] fp32 (the largest hot-path call per 32 frames): a slab against the oracle,
    idempotence of relu, and the clamp bound."""
    g = torch.Generator(device=DEV).manual_seed(6)
    x = torch.randn([32, 64, 256, 256], generator=g, device=DEV)
    b = torch.randn([64], generator=g, device=DEV)
    y = ba.bias_act(x, b, act='lrelu', gain=np.sqrt(2), clamp=2.0)
    assert y.abs().max().item() <= 2.0
    ref = oracle.bias_act(x[5:6, 10:12].cpu().contiguous(), None, act='linear')  # layout check helper
    assert ref.shape == (1, 2, 256, 256)
    sl = oracle.bias_act(x[31:32].cpu(), b.cpu(), act='lrelu', gain=np.sqrt(2), clamp=2.0)
    assert_bit_equal(y[31:32], sl, what='slab')
    r = ba.bias_act(x, None, act='relu', gain=1)
    assert torch.equal(ba.bias_act(r, None, act='relu', gain=1), r)


def test_bias_act_gradgradcheck():
    x = torch.randn([2, 3, 4], dtype=torch.float64, device=DEV, requires_grad=True)
    b = torch.randn([3], dtype=torch.float64, device=DEV, requires_grad=True)
    for act in ('lrelu', 'tanh', 'swish', 'softplus'):
        fn = lambda xx, bb: ba.bias_act(xx, bb, act=act)  # noqa: E731
        assert torch.autograd.gradcheck(fn, (x, b))
        assert torch.autograd.gradgradcheck(fn, (x, b))


@pytest.mark.parametrize('dtype', [torch.float32, torch.float64])
def test_bias_act_linear_gain_clamp_is_twice_differentiable_through_the_fused_bias_gradient(dtype):
    """ToRGB with conv_clamp / linear Conv2dLayer: act='linear' + bias + gain/clamp saves no tensor at all, and the fused
    dx+db node is picked whenever the bias requires grad.  A create_graph pass (R1, path-length) used to die in its backward."""
    g = torch.Generator().manual_seed(21)
    x = torch.randn([2, 8, 6, 6], generator=g).to(dtype).to(DEV).requires_grad_(True)
    b = torch.randn([8], generator=g).to(dtype).to(DEV).requires_grad_(True)
    for gain, clamp in ((1.7, None), (1.0, 0.8), (0.5, 1.2)):
        y = ba.bias_act(x, b, act='linear', gain=gain, clamp=clamp)
        (gx,) = torch.autograd.grad(y.sum(), x, create_graph=True)       # needs_input_grad is (True, True): fused-db node
        pen = gx.square().sum() + (y * y).sum()
        dx2, db2 = torch.autograd.grad(pen, [x, b], allow_unused=True)
        yr = ba.bias_act(x.detach().double().cpu().requires_grad_(True), b.detach().double().cpu(), act='linear', gain=gain, clamp=clamp, impl='ref')
        assert_close(y, yr, atol=1e-5, rtol=1e-5, what='y')
        assert dx2 is not None and torch.isfinite(dx2).all()
    # explicit second-order value: d/d(dy) of <dx, v> = gain * v where the (native, unmasked) linear gradient applies
    dy = torch.randn(x.shape, generator=g).to(dtype).to(DEV).requires_grad_(True)
    y = ba.bias_act(x, b, act='linear', gain=1.7)
    gx, gb = torch.autograd.grad(y, [x, b], dy, create_graph=True)
    v = torch.randn(x.shape, generator=g).to(dtype).to(DEV)
    (ddy,) = torch.autograd.grad((gx * v).sum() + gb.sum(), dy)
    assert_close(ddy, 1.7 * v + 1.7, atol=1e-5, rtol=1e-5, what='ddy')


def test_bias_act_errors():
    x = torch.randn([2, 4, 3, 3], device=DEV)
    with pytest.raises(RuntimeError, match='wrong number of elements'):
        ba.bias_act(x, torch.zeros([5], device=DEV))
    with pytest.raises(RuntimeError, match='same dtype and device'):
        ba.bias_act(x, torch.zeros([4], device=DEV, dtype=torch.float64))
    with pytest.raises(KeyError):
        ba.bias_act(x, None, act='gelu')


def test_bias_act_and_conv_accept_dense_views_at_unaligned_storage_offsets():
    """x[1:] of an [N, 3] fp32 tensor is contiguous but starts 12 bytes into its storage: the reference op takes such views (bias_act.cpp:46-51
    only asks for dense), so the 16-byte-vector kernels re-materialise them instead of raising."""
    from stylegan_v_amd.torch_utils.ops import bias_act, conv2d_gradfix
    base = torch.randn(5, 3, device='cuda')
    x = base[1:]
    assert x.is_contiguous() and x.data_ptr() % 16 != 0
    b = torch.randn(3, device='cuda')
    y = bias_act.bias_act(x, b, act='lrelu')
    assert torch.allclose(y, bias_act.bias_act(x.clone(), b, act='lrelu'))
    flat = torch.randn(1 + 2 * 64 * 16 * 32, device='cuda')
    xc = flat[1:].view(2, 64, 16, 32)
    assert xc.is_contiguous() and xc.data_ptr() % 16 != 0
    w = torch.randn(64, 64, 3, 3, device='cuda') / 24
    assert torch.allclose(conv2d_gradfix.conv2d(xc, w, padding=1), conv2d_gradfix.conv2d(xc.clone(), w, padding=1))
