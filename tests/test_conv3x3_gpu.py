"""3x3 convolution family (stride 1, stride 2, transposed; csrc/conv3x3*_kernel.h) on the 16-bit matrix pipe with fp32 emulation vs fp64 -- every test under
both arithmetics: the block-scaled fp16 split (terms = 4, the default: fp32-grade, <= 5e-7) and the bf16 split (terms = 3: <= 1e-5)."""
import pytest
import torch
import torch.nn.functional as F

from stylegan_v_amd.torch_utils import custom_ops
from stylegan_v_amd.torch_utils.ops import conv2d_gradfix
from util import assert_close, dispatch_assert

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.fixture(params=[4, 3], ids=['f16split', 'bf16split'], autouse=True)
def arithmetic(request):
    saved = (conv2d_gradfix.native_conv_terms, conv2d_gradfix.native_wrw_terms)
    conv2d_gradfix.native_conv_terms = conv2d_gradfix.native_wrw_terms = request.param
    yield request.param
    conv2d_gradfix.native_conv_terms, conv2d_gradfix.native_wrw_terms = saved


def _tol():
    """rel-L2 / max-norm bound against float64: MIOpen-class (its fp32 convolutions: 1.4-3.5e-7) for the fp16 split, 1e-5 for the bf16 split."""
    return 5e-7 if conv2d_gradfix.native_conv_terms == 4 else 1e-5


def _rel(a, ref):
    a, ref = a.double().cpu(), ref.double().cpu()
    return ((a - ref).norm() / ref.norm()).item(), ((a - ref).abs().max() / ref.abs().max()).item()


@pytest.mark.parametrize('n,ci,co,h,w', [(2, 64, 64, 32, 32), (1, 16, 128, 16, 64), (3, 128, 64, 48, 32), (1, 80, 192, 32, 96), (5, 32, 64, 16, 32),
                                          (4, 64, 128, 16, 16), (6, 32, 64, 16, 16), (16, 48, 64, 8, 8), (8, 128, 192, 8, 8),
                                          (8, 512, 128, 8, 8), (24, 256, 64, 16, 16), (12, 384, 192, 16, 16),     # few tiles, many chunks: K shared out over 8 / 8 / 4 workgroups (atomics)
                                          (32, 64, 64, 4, 4), (96, 512, 512, 4, 4), (64, 576, 512, 4, 4), (8, 128, 64, 4, 4), (37, 48, 128, 4, 4),    # 4 x 4: 32 samples per tile (round 5)
                                          (3, 64, 64, 16, 16), (5, 32, 128, 8, 8), (13, 64, 64, 8, 8),     # a partly filled last tile
                                          (2, 32, 32, 32, 64), (1, 64, 32, 16, 32), (2, 48, 96, 32, 32), (3, 32, 32, 16, 32)])     # c_out % 32: a half-full last tile
@pytest.mark.parametrize('transposed', [False, True])
def test_conv3x3_matches_fp64(n, ci, co, h, w, transposed):
    g = torch.Generator().manual_seed(n + ci + co + h)
    x = (torch.randn([n, ci, h, w], generator=g) + 0.3).to(DEV)
    wt = (torch.randn([ci, co, 3, 3] if transposed else [co, ci, 3, 3], generator=g) / (3 * ci ** 0.5)).to(DEV)
    cfg = (transposed, (1, 1), (1, 1), (0, 0), (1, 1), 1)
    dispatch_assert(conv2d_gradfix._native_conv_ok(x, wt, cfg))
    custom_ops.prof_enable(16)
    y = (conv2d_gradfix.conv_transpose2d if transposed else conv2d_gradfix.conv2d)(x.requires_grad_(True), wt, padding=1)
    custom_ops.prof_disable()
    dispatch_assert(custom_ops.prof_collect()['conv3x3']['launches'] == 1)
    ref = (F.conv_transpose2d if transposed else F.conv2d)(x.detach().double().cpu(), wt.double().cpu(), padding=1)
    lib = (F.conv_transpose2d if transposed else F.conv2d)(x.detach(), wt, padding=1)
    l2, mx = _rel(y, ref)
    l2_lib, mx_lib = _rel(lib, ref)
    print(f'terms {conv2d_gradfix.native_conv_terms}: rel-L2 {l2:.2e} max {mx:.2e} | MIOpen fp32 rel-L2 {l2_lib:.2e} max {mx_lib:.2e}')
    assert l2 < _tol() and mx < 2 * _tol(), (l2, mx, l2_lib, mx_lib)
    # exact on small integers (hi/lo split is exact, products and sums fit fp32): catches any tap / channel / pixel mix-up
    xi = torch.randint(-3, 4, x.shape, generator=g).float().to(DEV)
    wi = torch.randint(-2, 3, wt.shape, generator=g).float().to(DEV)
    yi = conv2d_gradfix._native_conv(xi, wi, cfg)
    assert torch.equal(yi.cpu().double(), (F.conv_transpose2d if transposed else F.conv2d)(xi.double().cpu(), wi.double().cpu(), padding=1))


def test_conv3x3_gradients_first_and_second_order():
    g = torch.Generator().manual_seed(3)
    x = torch.randn([2, 64, 16, 32], generator=g).to(DEV).requires_grad_(True)
    w = (torch.randn([64, 64, 3, 3], generator=g) / 24).to(DEV).requires_grad_(True)
    xr, wr = x.detach().double().cpu().requires_grad_(True), w.detach().double().cpu().requires_grad_(True)

    def run(conv, xx, ww):
        y = conv(xx, ww, padding=1)
        gx, gw = torch.autograd.grad(y.tanh().sum(), [xx, ww], create_graph=True)
        g2 = torch.autograd.grad(gx.square().sum() + gw.square().sum(), [xx, ww])
        return y, gx, gw, g2[0], g2[1]
    custom_ops.prof_enable(256)
    got = run(conv2d_gradfix.conv2d, x, w)
    custom_ops.prof_disable()
    prof = custom_ops.prof_collect()
    dispatch_assert(prof['conv3x3']['launches'] >= 4)   # forward, dx, and the convolutions inside the double backward
    want = run(F.conv2d, xr, wr)
    for a, r, name in zip(got, want, ['y', 'dx', 'dw', 'd2x', 'd2w']):
        assert_close(a, r, atol=3e-5 * r.abs().max().item(), rtol=1e-4, what=name)


def test_conv3x3_unsupported_shapes_use_the_vendor_library():
    lib = custom_ops.get_native()
    assert lib.sgv_conv3x3_supported(3, 64, 64, 16, 16, 0) == 1   # (round 5: a last tile may be partly filled -- 16x16 images need not come in pairs)
    assert lib.sgv_conv3x3_supported(4, 64, 64, 8, 8, 0) == 1
    assert lib.sgv_conv3x3_supported(4, 64, 64, 4, 4, 0) == 1     # 4 x 4: 32 samples per tile
    assert lib.sgv_conv3x3_supported(4, 64, 64, 4, 8, 0) == 0     # square images only
    assert lib.sgv_conv3x3_supported(4, 64, 64, 2, 2, 0) == 0
    assert lib.sgv_conv3x3_supported(4, 3, 64, 32, 32, 0) == 0    # c_in % 16
    assert lib.sgv_conv3x3_supported(4, 64, 48, 32, 32, 0) == 0   # c_out % 32
    assert lib.sgv_conv3x3_supported(4, 64, 32, 32, 32, 0) == 1   # a half-full last tile on the producer / consumer kernel ...
    assert lib.sgv_conv3x3_supported(4, 64, 32, 16, 16, 0) == 0   # ... which the 16x16 / 8x8 form is not
    assert lib.sgv_conv3x3_supported(4, 64, 64, 24, 32, 0) == 0   # H % 16
    assert lib.sgv_conv3x3_supported(4, 64, 64, 32, 32, 2) == 1   # bf16 tensors: the producer / consumer kernel (tests/test_conv_lowp_gpu.py)
    assert lib.sgv_conv3x3_supported(4, 64, 64, 16, 16, 2) == 0   # ... which the 16x16 / 8x8 form is not
    assert lib.sgv_conv3x3_supported(4, 64, 64, 32, 32, 3) == 0   # fp64
    x = torch.randn([3, 64, 12, 12], device=DEV)
    w = torch.randn([64, 64, 3, 3], device=DEV, requires_grad=True)
    before = custom_ops.launch_count()
    y = conv2d_gradfix.conv2d(x, w, padding=1)
    assert custom_ops.launch_count() == before
    assert_close(y, F.conv2d(x, w, padding=1), atol=1e-4, rtol=1e-4)
    # strided / padded variants never take the native path
    assert not conv2d_gradfix._native_conv_ok(torch.randn([2, 64, 32, 32], device=DEV), w, (False, (2, 2), (1, 1), (0, 0), (1, 1), 1))
    assert not conv2d_gradfix._native_conv_ok(torch.randn([2, 64, 32, 32], device=DEV), w, (False, (1, 1), (0, 0), (0, 0), (1, 1), 1))


@pytest.mark.parametrize('n,cb,cs,h,w', [(2, 64, 64, 8, 32), (1, 16, 128, 16, 64), (3, 128, 64, 24, 32), (1, 32, 64, 8, 96), (2, 96, 32, 8, 32), (2, 32, 48, 16, 64)])
@pytest.mark.parametrize('transposed', [False, True])
def test_conv3x3_stride2_family_matches_fp64(n, cb, cs, h, w, transposed):
    """cb channels on the (2h+1)x(2w+1) side, cs on the h x w side: strided maps big -> small, transposed small -> big."""
    g = torch.Generator().manual_seed(n + cb + cs + h)
    hb, wb = 2 * h + 1, 2 * w + 1
    if transposed:
        x = (torch.randn([n, cs, h, w], generator=g) + 0.3).to(DEV)
        wt = (torch.randn([cs, cb, 3, 3], generator=g) / (3 * cs ** 0.5)).to(DEV)
        if cb % 32:
            pytest.skip('c_out % 32')      # (the transposed producer / consumer kernel takes a half-full last tile)
    else:
        x = (torch.randn([n, cb, hb, wb], generator=g) + 0.3).to(DEV)
        wt = (torch.randn([cs, cb, 3, 3], generator=g) / (3 * cb ** 0.5)).to(DEV)
        if cs % 64:
            pytest.skip('c_out % 64')
    cfg = (transposed, (2, 2), (0, 0), (0, 0), (1, 1), 1)
    dispatch_assert(conv2d_gradfix._native_conv_kind(x, wt, cfg) == 's2')
    op, ref_op = (conv2d_gradfix.conv_transpose2d, F.conv_transpose2d) if transposed else (conv2d_gradfix.conv2d, F.conv2d)
    custom_ops.prof_enable(16)
    y = op(x.requires_grad_(True), wt, stride=2)
    custom_ops.prof_disable()
    dispatch_assert(custom_ops.prof_collect()['conv3x3']['launches'] == 1)
    ref = ref_op(x.detach().double().cpu(), wt.double().cpu(), stride=2)
    assert y.shape == ref.shape
    l2, mx = _rel(y, ref)
    l2_lib, mx_lib = _rel(ref_op(x.detach(), wt, stride=2), ref)
    print(f'terms {conv2d_gradfix.native_conv_terms} s2: rel-L2 {l2:.2e} max {mx:.2e} | MIOpen fp32 rel-L2 {l2_lib:.2e} max {mx_lib:.2e}')
    assert l2 < _tol() and mx < 2 * _tol(), (l2, mx, l2_lib, mx_lib)
    xi = torch.randint(-3, 4, x.shape, generator=g).float().to(DEV)
    wi = torch.randint(-2, 3, wt.shape, generator=g).float().to(DEV)
    assert torch.equal(conv2d_gradfix._native_conv(xi, wi, cfg).cpu().double(), ref_op(xi.double().cpu(), wi.double().cpu(), stride=2))


@pytest.mark.parametrize('n,ci,co,h,w', [(4, 32, 64, 16, 16), (2, 64, 128, 8, 16), (4, 16, 64, 8, 8), (8, 48, 64, 4, 8), (6, 32, 64, 12, 16)])
def test_transposed_stride2_on_small_images_packs_samples(n, ci, co, h, w):
    """W = 16 / 8: the transposed kernel packs 2 / 4 samples into one 32-pixel tile row (each with its own zero halo); float64 reference,
    exact on small integers, samples must not leak into each other (a non-zero sample next to a zero one)."""
    g = torch.Generator().manual_seed(n + ci + co + h + w)
    x = (torch.randn([n, ci, h, w], generator=g) + 0.3).to(DEV)
    wt = (torch.randn([ci, co, 3, 3], generator=g) / (3 * ci ** 0.5)).to(DEV)
    cfg = (True, (2, 2), (0, 0), (0, 0), (1, 1), 1)
    dispatch_assert(conv2d_gradfix._native_conv_kind(x, wt, cfg) == 's2')
    y = conv2d_gradfix._native_conv(x, wt, cfg)
    ref = F.conv_transpose2d(x.double().cpu(), wt.double().cpu(), stride=2)
    assert y.shape == ref.shape
    l2, mx = _rel(y, ref)
    assert l2 < _tol() and mx < 2 * _tol(), (l2, mx)
    xi = torch.randint(-3, 4, x.shape, generator=g).float()
    xi[1::2] = 0                                                     # every second sample is zero: its output must be exactly zero
    wi = torch.randint(-2, 3, wt.shape, generator=g).float()
    yi = conv2d_gradfix._native_conv(xi.to(DEV), wi.to(DEV), cfg).cpu().double()
    assert torch.equal(yi, F.conv_transpose2d(xi.double(), wi.double(), stride=2))
    assert yi[1::2].abs().max() == 0
    # odd batch for W = 16 (or a batch that is no multiple of 4 for W = 8) stays with the vendor library
    assert conv2d_gradfix._native_conv_kind(x[:1], wt, cfg) is None


@pytest.mark.parametrize('n,ci,co,h,w', [(4, 32, 128, 16, 16), (2, 64, 256, 8, 16), (4, 16, 128, 8, 8), (8, 48, 128, 16, 8)])
def test_strided_stride2_on_small_images_packs_samples(n, ci, co, h, w):
    """W = 16 / 8 outputs: the tap-pair kernel packs 2 / 4 samples into one 32-pixel tile row; float64 reference, exact on small integers, no
    leakage between neighbouring samples."""
    g = torch.Generator().manual_seed(n + ci + co + h + w)
    x = (torch.randn([n, ci, 2 * h + 1, 2 * w + 1], generator=g) + 0.3).to(DEV)
    wt = (torch.randn([co, ci, 3, 3], generator=g) / (3 * ci ** 0.5)).to(DEV)
    cfg = (False, (2, 2), (0, 0), (0, 0), (1, 1), 1)
    dispatch_assert(conv2d_gradfix._native_conv_kind(x, wt, cfg) == 's2')
    y = conv2d_gradfix._native_conv(x, wt, cfg)
    ref = F.conv2d(x.double().cpu(), wt.double().cpu(), stride=2)
    assert y.shape == ref.shape
    l2, mx = _rel(y, ref)
    assert l2 < _tol() and mx < 2 * _tol(), (l2, mx)
    xi = torch.randint(-3, 4, x.shape, generator=g).float()
    xi[1::2] = 0
    wi = torch.randint(-2, 3, wt.shape, generator=g).float()
    yi = conv2d_gradfix._native_conv(xi.to(DEV), wi.to(DEV), cfg).cpu().double()
    assert torch.equal(yi, F.conv2d(xi.double(), wi.double(), stride=2))
    assert yi[1::2].abs().max() == 0
    assert conv2d_gradfix._native_conv_kind(x[:1], wt, cfg) is None


def test_conv3x3_stride2_gradients_first_and_second_order():
    g = torch.Generator().manual_seed(4)
    x = torch.randn([2, 64, 17, 65], generator=g).to(DEV).requires_grad_(True)
    w = (torch.randn([64, 64, 3, 3], generator=g) / 24).to(DEV).requires_grad_(True)
    xr, wr = x.detach().double().cpu().requires_grad_(True), w.detach().double().cpu().requires_grad_(True)

    def run(conv, xx, ww):
        y = conv(xx, ww, stride=2)
        gx, gw = torch.autograd.grad(y.tanh().sum(), [xx, ww], create_graph=True)
        g2 = torch.autograd.grad(gx.square().sum() + gw.square().sum(), [xx, ww])
        return y, gx, gw, g2[0], g2[1]
    custom_ops.prof_enable(256)
    got = run(conv2d_gradfix.conv2d, x, w)
    custom_ops.prof_disable()
    dispatch_assert(custom_ops.prof_collect()['conv3x3']['launches'] >= 4)
    want = run(F.conv2d, xr, wr)
    for a, r, name in zip(got, want, ['y', 'dx', 'dw', 'd2x', 'd2w']):
        assert_close(a, r, atol=3e-5 * r.abs().max().item(), rtol=1e-4, what=name)


@pytest.mark.parametrize('stride,transposed', [(1, False), (1, True), (2, False), (2, True)])
def test_conv3x3_family_vs_oracle(stride, transposed):
    """All four geometries and their weight gradients against oracle/oracle.py (float64 textbook definition)."""
    import oracle
    g = torch.Generator().manual_seed(10 + stride * 2 + transposed)
    h, w = (17, 65) if (stride == 2 and not transposed) else (16, 32) if stride == 1 else (8, 32)
    x = torch.randn([2, 64, h, w], generator=g)
    wt = torch.randn([64, 64, 3, 3], generator=g) / 24
    xg, wg = x.to(DEV), wt.to(DEV).requires_grad_(True)
    op = conv2d_gradfix.conv_transpose2d if transposed else conv2d_gradfix.conv2d
    custom_ops.prof_enable(64)
    y = op(xg, wg, stride=stride, padding=1 if stride == 1 else 0)
    dy = torch.randn(y.shape, generator=g)
    gw, = torch.autograd.grad(y, [wg], dy.to(DEV))
    custom_ops.prof_disable()
    prof = custom_ops.prof_collect()
    dispatch_assert(prof['conv3x3']['launches'] == 1 and prof['conv_wrw']['launches'] == 1)
    for got, ref, what in ((y, oracle.conv3x3(x.numpy(), wt.numpy(), stride=stride, transposed=transposed), 'y'),
                           (gw, oracle.conv3x3_weight_grad(dy.numpy(), x.numpy(), stride=stride, transposed=transposed), 'dw')):
        l2, mx = _rel(got.detach(), torch.as_tensor(ref))
        assert l2 < _tol() and mx < 2 * _tol(), (what, l2, mx)


def test_native_kernels_also_serve_no_grad_passes():
    """The D phase runs the generator under torch.no_grad() (loss.py:123 of the reference): those convolutions must not fall back."""
    x = torch.randn([2, 64, 16, 32], device=DEV)
    w = torch.randn([64, 64, 3, 3], device=DEV) / 24
    custom_ops.prof_enable(16)
    with torch.no_grad():
        y = conv2d_gradfix.conv2d(x, w, padding=1)
        yt = conv2d_gradfix.conv_transpose2d(x, w, stride=2)
    custom_ops.prof_disable()
    dispatch_assert(custom_ops.prof_collect()['conv3x3']['launches'] == 2)
    assert_close(y, F.conv2d(x.double(), w.double(), padding=1), atol=2e-5 * 3, rtol=1e-4)
    assert_close(yt, F.conv_transpose2d(x.double(), w.double(), stride=2), atol=2e-5 * 3, rtol=1e-4)


def test_small_images_with_odd_channel_counts_are_zero_padded_onto_the_native_kernel():
    """The discriminator's epilogue convolution (networks.py:518-576: 512 + 1 minibatch-std channels at 4 x 4) has c_in = 513: conv2d_gradfix pads x and w with zero
    channels up to a multiple of 64, so that the convolution AND its data gradient (c_out of the transposed form) run on conv3x3_small_kernel instead of the
    vendor library; gradients flow back through the padding (first and second order)."""
    g = torch.Generator().manual_seed(9)
    x = torch.randn([32, 513, 4, 4], generator=g).to(DEV).requires_grad_(True)
    w = (torch.randn([512, 513, 3, 3], generator=g) / 68).to(DEV).requires_grad_(True)
    before = custom_ops.kernel_variant_counts()['conv_small']
    y = conv2d_gradfix.conv2d(x, w, padding=1)
    dy = torch.randn(y.shape, generator=g).to(DEV)
    dx, dw = torch.autograd.grad(y, [x, w], dy, create_graph=True)
    dispatch_assert(custom_ops.kernel_variant_counts()['conv_small'] - before == 2, 'forward and data gradient must take the 4 x 4 kernel')
    (d2,) = torch.autograd.grad(dx.square().sum(), [w])
    xr, wr = x.detach().double().cpu().requires_grad_(True), w.detach().double().cpu().requires_grad_(True)
    yr = F.conv2d(xr, wr, padding=1)
    dxr, dwr = torch.autograd.grad(yr, [xr, wr], dy.double().cpu(), create_graph=True)
    (d2r,) = torch.autograd.grad(dxr.square().sum(), [wr])
    assert y.shape == yr.shape and dx.shape == xr.shape and dw.shape == wr.shape
    for got, ref, name in ((y, yr, 'y'), (dx, dxr, 'dx'), (dw, dwr, 'dw'), (d2, d2r, 'd2w')):
        l2, mx = _rel(got.detach(), ref.detach())
        assert l2 < 5e-6 and mx < 2e-5, (name, l2, mx)
