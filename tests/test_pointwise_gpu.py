"""ToRGB / fromRGB streaming 1x1 kernels (csrc/pointwise.hip) against torch's conv2d / matmul in fp64/fp32."""
import pytest
import torch
import torch.nn.functional as F

from stylegan_v_amd.torch_utils import custom_ops
from stylegan_v_amd.torch_utils.ops import pointwise, conv2d_resample
from util import assert_close

pytestmark = pytest.mark.gpu


def _ref(x, w):
    n = x.shape[0]
    y = torch.matmul(w.double().expand(n, -1, -1), x.double().reshape(n, x.shape[1], -1))
    return y.reshape(n, w.shape[1], *x.shape[2:])


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize('n,ci,co,h,w,shared', [
    (3, 64, 3, 32, 32, False), (2, 512, 3, 4, 4, False), (2, 37, 3, 10, 6, False), (4, 128, 3, 64, 64, True),
    (3, 3, 64, 32, 32, True), (2, 3, 128, 16, 16, False), (2, 1, 5, 8, 8, True), (2, 7, 1, 8, 8, False), (1, 4, 4, 6, 6, True),
    (2, 96, 2, 129, 4, False),
])
def test_pointwise_forward(dtype, n, ci, co, h, w, shared):
    torch.manual_seed(n * 1000 + ci * 10 + co)
    x = torch.randn(n, ci, h, w, device='cuda').to(dtype)
    wt = torch.randn(1 if shared else n, co, ci, device='cuda') / ci ** 0.5
    y = pointwise.pointwise_conv(x, wt)
    assert y.dtype == dtype and y.shape == (n, co, h, w)
    tol = {torch.float32: 2e-5, torch.float16: 2e-3, torch.bfloat16: 1.6e-2}[dtype]
    assert_close(y.double(), _ref(x, wt), atol=tol, rtol=tol)


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16])
@pytest.mark.parametrize('n,f,m,h,w', [(3, 3, 64, 32, 32), (2, 3, 512, 4, 4), (2, 1, 9, 130, 2), (2, 4, 4, 64, 64), (2, 3, 32, 128, 128)])
def test_outer(dtype, n, f, m, h, w):
    torch.manual_seed(f * 100 + m)
    a = torch.randn(n, f, h, w, device='cuda').to(dtype)
    b = torch.randn(n, m, h, w, device='cuda').to(dtype)
    out = pointwise.outer(a, b)
    ref = torch.matmul(a.double().reshape(n, f, -1), b.double().reshape(n, m, -1).transpose(1, 2))
    assert out.dtype == torch.float32
    scale = (h * w) ** 0.5
    assert_close(out.double() / scale, ref / scale, atol=2e-5 if dtype == torch.float32 else 1e-4, rtol=1e-5)


@pytest.mark.parametrize('n,ci,co,shared', [(2, 16, 3, False), (2, 3, 16, True), (2, 16, 3, True), (2, 3, 8, False)])
def test_pointwise_first_and_second_order_gradients(n, ci, co, shared):
    torch.manual_seed(7)
    x = torch.randn(n, ci, 6, 6, device='cuda', requires_grad=True)
    wt = (torch.randn(1 if shared else n, co, ci, device='cuda') / ci ** 0.5).requires_grad_(True)
    x2 = x.detach().clone().requires_grad_(True)
    w2 = wt.detach().clone().requires_grad_(True)

    def loss(fn, xx, ww):
        y = fn(xx, ww)
        g, gw = torch.autograd.grad((y * y.sin()).sum(), [xx, ww], create_graph=True)
        return y, g, gw, (g.square().sum() + gw.square().sum())

    y, g, gw, r = loss(pointwise.pointwise_conv, x, wt)
    yr, gr, gwr, rr = loss(pointwise.pointwise_conv_ref, x2, w2)
    assert_close(y, yr, atol=1e-5, rtol=1e-5)
    assert_close(g, gr, atol=1e-5, rtol=1e-5)
    assert_close(gw, gwr, atol=1e-4, rtol=1e-5)
    r.backward()
    rr.backward()
    assert_close(x.grad, x2.grad, atol=1e-4, rtol=1e-4)
    assert_close(wt.grad, w2.grad, atol=1e-3, rtol=1e-4)


def test_conv2d_resample_uses_stream_kernel_for_fromrgb_shape():
    from stylegan_v_amd.torch_utils import custom_ops
    torch.manual_seed(3)
    x = torch.randn(4, 3, 32, 32, device='cuda', requires_grad=True)
    w = torch.randn(24, 3, 1, 1, device='cuda', requires_grad=True)
    before = custom_ops.launch_count()
    y = conv2d_resample.conv2d_resample(x, w)
    assert custom_ops.launch_count() == before + 1
    yr = F.conv2d(x.detach(), w.detach())
    assert_close(y, yr, atol=1e-5, rtol=1e-5)
    gx, gw = torch.autograd.grad(y.square().sum(), [x, w])
    xr, wr = x.detach().clone().requires_grad_(True), w.detach().clone().requires_grad_(True)
    gxr, gwr = torch.autograd.grad(F.conv2d(xr, wr).square().sum(), [xr, wr])
    assert_close(gx, gxr, atol=1e-4, rtol=1e-4)
    assert_close(gw, gwr, atol=1e-3, rtol=1e-4)


@pytest.mark.parametrize('act,clamp,with_bias', [('lrelu', None, True), ('lrelu', 0.6, True), ('linear', None, True), ('lrelu', None, False)])
def test_fused_fromrgb_tail_is_bit_identical_to_the_two_pass_composition(act, clamp, with_bias):
    """sgv_pointwise_act = pointwise kernel + bias_act's own operations in its own order: one launch, bit-identical forward, the same first and
    second order gradients as the composition (layers.py Conv2dLayer.forward of the discriminator's fromRGB)."""
    from stylegan_v_amd.torch_utils import custom_ops
    from stylegan_v_amd.torch_utils.ops import bias_act
    torch.manual_seed(11)
    n, ci, co, h, w = 3, 3, 64, 32, 32
    mk = lambda *s: torch.randn(*s, device='cuda')   # noqa: E731
    x, wt, b = mk(n, ci, h, w).requires_grad_(True), (mk(1, co, ci) / ci ** 0.5).requires_grad_(True), (mk(co) * 0.5).requires_grad_(True) if with_bias else None
    gain = 2 ** 0.5 * 0.7

    def composed(xx, ww, bb):
        return bias_act.bias_act(pointwise.pointwise_conv(xx, ww), bb, act=act, gain=gain, clamp=clamp)

    def fused(xx, ww, bb):
        return pointwise.pointwise_conv_bias_act(xx, ww, bb, act=act, gain=gain, clamp=clamp)

    with torch.no_grad():
        before = custom_ops.launch_count()
        yf = fused(x, wt, b)
        assert custom_ops.launch_count() == before + 1
        assert torch.equal(yf, composed(x, wt, b))

    # first-order pass (no graph of the gradient): weight + bias gradient and input gradient kernels evaluate the activation derivative from
    # (dy, y) themselves -- two launches, no dz tensor
    ins = [t for t in (x, wt, b) if t is not None]
    dy = torch.randn(n, co, h, w, device='cuda')
    yf = fused(x, wt, b)
    before = custom_ops.launch_count()
    got = torch.autograd.grad(yf, ins, dy)
    assert custom_ops.launch_count() == before + 2
    want = torch.autograd.grad(composed(x, wt, b), ins, dy)
    for a, r, name in zip(got, want, 'xwb'):
        assert_close(a, r, atol=2e-4 if name != 'x' else 1e-5, rtol=1e-5, what='first-order d' + name)

    def r1(fn):
        ins = [t for t in (x, wt, b) if t is not None]
        y = fn(x, wt, b)
        g = torch.autograd.grad((y * y.sin()).sum(), ins, create_graph=True)
        gg = torch.autograd.grad(g[0].square().sum(), ins[1:], allow_unused=True)
        return list(g) + [t for t in gg if t is not None]
    for a, r in zip(r1(fused), r1(composed)):
        assert_close(a, r, atol=1e-4, rtol=1e-4)


def test_act_grad_scale_and_scale_dot_beyond_65535_planes():
    """More planes than blockIdx.y holds (192 frames of a 512-channel layer: the Dmain phase as one pass, larger per-GPU batches): the launch goes in slabs
    of 65,535 planes -- activation gradient, its two per-plane sums, input gradient and per-plane dot against the same formulas in torch."""
    lib = custom_ops.get_native()
    dev = torch.device('cuda')
    g = torch.Generator().manual_seed(3)
    planes, hw = 70001, 16
    dy, y = torch.randn([planes, hw], generator=g).to(dev), torch.randn([planes, hw], generator=g).to(dev)
    d = (torch.rand([planes], generator=g) + 0.5).to(dev)
    out = torch.empty_like(dy)
    sums = torch.zeros([2, planes], device=dev)
    alpha, gain = 0.2, 2 ** 0.5
    with torch.cuda.device(dev):
        custom_ops.check(lib.sgv_act_grad_scale(dy.data_ptr(), y.data_ptr(), d.data_ptr(), out.data_ptr(), sums.data_ptr(), planes, hw, 3, alpha, gain, -1.0,
                                                torch.cuda.current_stream().cuda_stream), lib)
    dz = torch.where(y > 0, dy, dy * alpha) * gain
    assert torch.allclose(out, dz * d[:, None], rtol=1e-6, atol=1e-6)
    assert torch.allclose(sums[0], dz.sum(1), rtol=1e-5, atol=1e-5) and torch.allclose(sums[1], (dy * y).sum(1), rtol=1e-5, atol=1e-5)
    a, b = torch.randn([planes, hw], generator=g).to(dev), torch.randn([planes, hw], generator=g).to(dev)
    o2, dot = torch.empty_like(a), torch.zeros([planes], device=dev)
    with torch.cuda.device(dev):
        custom_ops.check(lib.sgv_scale_dot(a.data_ptr(), b.data_ptr(), d.data_ptr(), o2.data_ptr(), dot.data_ptr(), planes, hw, torch.cuda.current_stream().cuda_stream), lib)
    assert torch.allclose(o2, a * d[:, None], rtol=1e-6, atol=1e-6) and torch.allclose(dot, (a * b).sum(1), rtol=1e-5, atol=1e-5)
