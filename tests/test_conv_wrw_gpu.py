"""3x3 convolution weight gradient on the bf16 matrix pipe with fp32 emulation (csrc/wrw_kernel.h) against fp64."""
import pytest
import torch
import torch.nn.functional as F

from stylegan_v_amd.torch_utils import custom_ops
from stylegan_v_amd.torch_utils.ops import conv2d_gradfix
from util import assert_close, dispatch_assert

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _ref_dw(dy, x):
    """fp64 on the host: dw[o,i,ky,kx] = sum dy[n,o,y,x] x[n,i,y+ky-1,x+kx-1]."""
    xd = x.double().cpu().requires_grad_(False)
    w = torch.zeros([dy.shape[1], x.shape[1], 3, 3], dtype=torch.float64, requires_grad=True)
    y = F.conv2d(xd, w, padding=1)
    return torch.autograd.grad(y, w, dy.double().cpu())[0]


def _native(dy, x, terms):
    old = conv2d_gradfix.native_wrw_terms
    conv2d_gradfix.native_wrw_terms = terms
    try:
        dispatch_assert(conv2d_gradfix._native_wrw_ok(dy, x, (False, (1, 1), (1, 1), (0, 0), (1, 1), 1), (dy.shape[1], x.shape[1], 3, 3)))
        return conv2d_gradfix._native_wrw(dy, x, (False, (1, 1), (1, 1), (0, 0), (1, 1), 1), (dy.shape[1], x.shape[1], 3, 3))
    finally:
        conv2d_gradfix.native_wrw_terms = old


def _rel(a, ref):
    a, ref = a.double().cpu(), ref.double().cpu()
    return ((a - ref).norm() / ref.norm()).item(), ((a - ref).abs().max() / ref.abs().max()).item()


@pytest.mark.parametrize('n,o,i,h,w', [(2, 64, 64, 32, 32), (1, 64, 128, 64, 64), (2, 128, 64, 64, 96), (3, 64, 64, 8, 32), (1, 192, 64, 96, 32), (2, 64, 64, 1, 32),
                                       # images 16 / 8 pixels wide: 2 / 4 samples share a 32-pixel row step (incl. batches that leave the last group short)
                                       (4, 64, 64, 16, 16), (3, 128, 64, 16, 16), (1, 64, 64, 16, 16), (8, 64, 128, 8, 8), (5, 64, 64, 8, 8), (2, 64, 64, 4, 8), (7, 64, 64, 32, 16)])
@pytest.mark.parametrize('terms', [4, 3])
def test_wrw_bf16x3_matches_fp64_as_well_as_the_vendor_fp32_kernel(n, o, i, h, w, terms):
    """terms = 4 (block-scaled fp16 split, the default): fp32-grade, <= 5e-7 of the result's scale; terms = 3 (bf16 split): <= 1e-5."""
    g = torch.Generator().manual_seed(n * 100 + o + i + h)
    dy = torch.randn([n, o, h, w], generator=g).to(DEV)
    x = (torch.randn([n, i, h, w], generator=g) * 1.5 + 0.25).to(DEV)
    ref = _ref_dw(dy, x)
    got = _native(dy, x, terms)
    assert got.shape == ref.shape and got.dtype == torch.float32
    l2, mx = _rel(got, ref)
    # vendor fp32 kernel on the same inputs, for scale
    w_like = x.new_empty(ref.shape)
    _, dw_lib, _ = torch.ops.aten.convolution_backward(dy, x, w_like, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1, [False, True, False])
    l2_lib, mx_lib = _rel(dw_lib, ref)
    print(f'terms {terms}: rel-L2 {l2:.2e} max {mx:.2e} | MIOpen fp32 rel-L2 {l2_lib:.2e} max {mx_lib:.2e}')
    # a weight gradient sums n * h * w products (up to 12,288 here) in fp32, three MFMA accumulations per 16 of them: measured 2.5-6.2e-7 with the fp16 split
    # (profiles/r04_conv_terms_accuracy.json) where the vendor's fp32 kernel has 2.0-2.8e-7 on the same inputs -- the summation, not the operands
    tol = 1e-6 if terms == 4 else 1e-5
    assert l2 < tol and mx < 2 * tol, (l2, mx, l2_lib, mx_lib)
    # asymmetric structure check: exact small integers survive the split exactly -> bit-exact result
    dyi = torch.randint(-3, 4, dy.shape, generator=g).float().to(DEV)
    xi = torch.randint(-3, 4, x.shape, generator=g).float().to(DEV)
    assert torch.equal(_native(dyi, xi, terms).cpu().double(), _ref_dw(dyi, xi))


def test_wrw_single_term_is_plain_bf16():
    g = torch.Generator().manual_seed(1)
    dy, x = torch.randn([2, 64, 32, 32], generator=g).to(DEV), torch.randn([2, 64, 32, 32], generator=g).to(DEV)
    ref = _ref_dw(dy, x)
    l2, _ = _rel(_native(dy, x, 1), ref)
    assert 1e-4 < l2 < 1e-2
    # exactly the product of bf16-rounded operands
    l2b, _ = _rel(_native(dy, x, 1), _ref_dw(dy.bfloat16().float(), x.bfloat16().float()))
    assert l2b < 1e-6


def test_conv2d_gradfix_uses_it_and_stays_twice_differentiable():
    g = torch.Generator().manual_seed(2)
    x = torch.randn([2, 64, 32, 32], generator=g).to(DEV).requires_grad_(True)
    w = (torch.randn([64, 64, 3, 3], generator=g) / 24).to(DEV).requires_grad_(True)
    custom_ops.prof_enable(64)
    y = conv2d_gradfix.conv2d(x, w, padding=1)
    gw, = torch.autograd.grad(y.sin().sum(), [w], create_graph=True)
    custom_ops.prof_disable()
    dispatch_assert(custom_ops.prof_collect()['conv_wrw']['launches'] == 1)
    xr, wr = x.detach().double().cpu().requires_grad_(True), w.detach().double().cpu().requires_grad_(True)
    gwr, = torch.autograd.grad(F.conv2d(xr, wr, padding=1).sin().sum(), [wr], create_graph=True)
    assert_close(gw, gwr, atol=2e-5 * gwr.abs().max().item(), rtol=1e-5, what='dw')
    # second order: d/dx and d/dw of |dw|^2 (goes through _ConvGradWeight.backward -> ordinary convolutions)
    g2 = torch.autograd.grad(gw.square().sum(), [x, w])
    g2r = torch.autograd.grad(gwr.square().sum(), [xr, wr])
    for a, r, name in zip(g2, g2r, 'xw'):
        assert_close(a, r, atol=1e-4 * r.abs().max().item(), rtol=1e-3, what='d2 ' + name)


def test_unsupported_shapes_fall_back_to_the_vendor_library():
    lib = custom_ops.get_native()
    assert lib.sgv_conv3x3_wrw_supported(4, 64, 64, 16, 16, 0) == 1    # 16 / 8 pixels wide: packed samples
    assert lib.sgv_conv3x3_wrw_supported(4, 64, 64, 4, 4, 0) == 0      # W < 8
    assert lib.sgv_conv3x3_wrw_supported(4, 64, 64, 24, 24, 0) == 0    # W not 8, 16 or a multiple of 32
    assert lib.sgv_conv3x3_wrw_supported(4, 64, 3, 32, 32, 0) == 0     # fromRGB-like channel counts
    assert lib.sgv_conv3x3_wrw_supported(4, 64, 64, 48, 32, 0) == 0    # H > 32 and not a multiple of 32
    assert lib.sgv_conv3x3_wrw_supported(4, 64, 64, 32, 32, 1) == 1    # fp16 tensors, fp32 gradient (tests/test_conv_lowp_gpu.py)
    assert lib.sgv_conv3x3_wrw_supported(4, 64, 64, 16, 16, 1) == 0    # ... not the packed-sample form
    assert lib.sgv_conv3x3_wrw_supported(4, 64, 64, 32, 32, 3) == 0    # fp64
    x = torch.randn([2, 64, 4, 4], device=DEV, requires_grad=True)
    w = torch.randn([64, 64, 3, 3], device=DEV, requires_grad=True)
    custom_ops.prof_enable(64)
    torch.autograd.grad(conv2d_gradfix.conv2d(x, w, padding=1).sum(), [w])
    custom_ops.prof_disable()
    assert custom_ops.prof_collect()['conv_wrw']['launches'] == 0
    p = custom_ops.ConvWrwParams(x.data_ptr(), x.data_ptr(), w.data_ptr(), 2, 64, 64, 4, 4, 3)
    assert lib.sgv_conv3x3_wrw(p, 0, None) == -3 and b'W % 32' in lib.sgv_last_error()


@pytest.mark.parametrize('n,cs,cb,h,w', [(2, 64, 64, 8, 32), (1, 128, 64, 32, 64), (3, 64, 128, 64, 32), (2, 64, 64, 1, 96),
                                          # small grid 16 / 8 pixels wide (big 33 / 17): 2 / 4 samples per row step, incl. a short last group
                                          (4, 64, 64, 16, 16), (3, 64, 128, 16, 16), (8, 128, 64, 8, 8), (5, 64, 64, 8, 8), (1, 64, 64, 4, 8)])
@pytest.mark.parametrize('transposed', [False, True])
@pytest.mark.parametrize('terms', [4, 3])
def test_wrw_stride2_family(n, cs, cb, h, w, transposed, terms, monkeypatch):
    """Weight gradient of the strided (big -> small) and of the transposed (small -> big) 3x3 layer."""
    monkeypatch.setattr(conv2d_gradfix, 'native_wrw_terms', terms)
    g = torch.Generator().manual_seed(n + cs + cb + h)
    small = torch.randn([n, cs, h, w], generator=g).to(DEV)
    big = (torch.randn([n, cb, 2 * h + 1, 2 * w + 1], generator=g) * 1.5 + 0.25).to(DEV)
    cfg = (transposed, (2, 2), (0, 0), (0, 0), (1, 1), 1)
    x, dy = (small, big) if transposed else (big, small)
    dispatch_assert(conv2d_gradfix._native_wrw_kind(dy, x, cfg, (cs, cb, 3, 3)) == 's2')
    got = conv2d_gradfix._native_wrw(dy, x, cfg, (cs, cb, 3, 3))
    wz = torch.zeros([cs, cb, 3, 3], dtype=torch.float64, requires_grad=True)
    xd, dyd = x.double().cpu(), dy.double().cpu()
    y = F.conv_transpose2d(xd, wz, stride=2) if transposed else F.conv2d(xd, wz, stride=2)
    ref = torch.autograd.grad(y, wz, dyd)[0]
    l2, mx = _rel(got, ref)
    print(f'terms {terms} wrw-s2 rel-L2 {l2:.2e} max {mx:.2e}')
    tol = 1e-6 if terms == 4 else 1e-5
    assert l2 < tol and mx < 2 * tol, (l2, mx)
    si = torch.randint(-3, 4, small.shape, generator=g).float().to(DEV)
    bi = torch.randint(-3, 4, big.shape, generator=g).float().to(DEV)
    xi, dyi = (si, bi) if transposed else (bi, si)
    yi = F.conv_transpose2d(xi.double().cpu(), wz, stride=2) if transposed else F.conv2d(xi.double().cpu(), wz, stride=2)
    assert torch.equal(conv2d_gradfix._native_wrw(dyi, xi, cfg, (cs, cb, 3, 3)).cpu().double(), torch.autograd.grad(yi, wz, dyi.double().cpu())[0])


def test_weight_gradient_with_input_scale_equals_scaling_first():
    """sgv_conv3x3_wrw_scaled: x * x_scale[n, i] formed on the operand's way into LDS == scaling x first (the same fp32 products enter the split)."""
    import torch
    from stylegan_v_amd.torch_utils.ops import conv2d_gradfix
    g = torch.Generator().manual_seed(3)
    n, co, ci, h, w = 3, 64, 128, 32, 64
    dy = torch.randn([n, co, h, w], generator=g).cuda()
    x = torch.randn([n, ci, h, w], generator=g).cuda()
    s = (torch.randn([n, ci], generator=g) * 0.5 + 1).cuda()
    cfg = (False, (1, 1), (1, 1), (0, 0), (1, 1), 1)
    dispatch_assert(conv2d_gradfix.wrw_input_scale and conv2d_gradfix._native_wrw_kind(dy, x, cfg, (co, ci, 3, 3)) == 's1', 'the producer / consumer weight-gradient kernel takes the input scale')
    a = conv2d_gradfix._native_wrw(dy, x, cfg, (co, ci, 3, 3), x_scale=s)
    b = conv2d_gradfix._native_wrw(dy, x * s[:, :, None, None], cfg, (co, ci, 3, 3))
    assert (a - b).abs().max().item() <= 2e-5 * b.abs().max().item()     # atomics: the accumulation order over workgroups differs run to run
    # packed-sample form (16-pixel images): every 8-pixel group takes the scale of its own sample
    n, h, w = 5, 16, 16
    dy, x = torch.randn([n, co, h, w], generator=g).cuda(), torch.randn([n, ci, h, w], generator=g).cuda()
    s = (torch.randn([n, ci], generator=g) * 0.5 + 1).cuda()
    a = conv2d_gradfix._native_wrw(dy, x, cfg, (co, ci, 3, 3), x_scale=s)
    b = conv2d_gradfix._native_wrw(dy, x * s[:, :, None, None], cfg, (co, ci, 3, 3))
    assert (a - b).abs().max().item() <= 2e-5 * b.abs().max().item()
