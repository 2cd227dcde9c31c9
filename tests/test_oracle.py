"""Pins the CPU oracle (oracle/) against golden vectors produced by the reference's own Python
implementations (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

import oracle
from util import Golden, assert_close

UFD = Golden('upfirdn2d')
BA = Golden('bias_act')


def _f(i):
    return UFD.t(f'c{i}_f') if UFD.meta[i]['has_f'] else None


@pytest.mark.parametrize('i', range(len(UFD.meta)))
def test_upfirdn2d_forward_matches_reference(i):
    m = UFD.meta[i]
    y = oracle.upfirdn2d(UFD.t(f'c{i}_x'), _f(i), up=m['up'], down=m['down'], padding=m['padding'], flip_filter=m['flip'], gain=m['gain'])
    # fp64 data, but `gain` crosses the native ABI as a C float (upfirdn2d.cpp:16) while the reference's
    # Python fallback folds it into the fp32 filter (upfirdn2d.py:190): agreement to fp32 epsilon of gain.
    assert_close(y, UFD.t(f'c{i}_y'), atol=5e-6, rtol=1e-6, what=f'case {i} fp64')
    # fp32 data path (fp32 accumulate): within fp32 round-off of the fp64 reference
    y32 = oracle.upfirdn2d(UFD.t(f'c{i}_x', torch.float32), _f(i), up=m['up'], down=m['down'], padding=m['padding'], flip_filter=m['flip'], gain=m['gain'])
    assert y32.dtype == torch.float32
    assert_close(y32, UFD.t(f'c{i}_y'), atol=2e-5, rtol=1e-5, what=f'case {i} fp32')


def _backward_cfg(m, x_shape, y_shape, f):
    """Gradient-as-upfirdn2d rule, upfirdn2d.py:245-261."""
    upx, upy = (m['up'], m['up']) if isinstance(m['up'], int) else m['up']
    dnx, dny = (m['down'], m['down']) if isinstance(m['down'], int) else m['down']
    p = m['padding']
    p = [p] * 4 if isinstance(p, int) else (p if len(p) == 4 else [p[0], p[0], p[1], p[1]])
    fw, fh = (1, 1) if f is None else (f.shape[-1], f.shape[0])
    ih, iw = x_shape[2:]
    oh, ow = y_shape[2:]
    pad = [fw - p[0] - 1, iw * upx - ow * dnx + p[0] - upx + 1, fh - p[2] - 1, ih * upy - oh * dny + p[2] - upy + 1]
    return dict(up=(dnx, dny), down=(upx, upy), padding=pad, flip_filter=not m['flip'], gain=m['gain'])


@pytest.mark.parametrize('i', range(len(UFD.meta)))
def test_upfirdn2d_gradients_match_reference_autograd(i):
    """d/dx and the second-order term d(dx.v)/d(dy), both expressed as oracle upfirdn2d calls."""
    m, f = UFD.meta[i], _f(i)
    x, y, dy, v = (UFD.t(f'c{i}_{k}') for k in ('x', 'y', 'dy', 'v'))
    bcfg = _backward_cfg(m, x.shape, y.shape, f)
    dx = oracle.upfirdn2d(dy, f, **bcfg)
    assert_close(dx, UFD.t(f'c{i}_dx'), atol=5e-6, rtol=1e-6, what=f'case {i} dx')
    # dx = B(dy) is linear in dy, so d(<B(dy), v>)/d(dy) = B^T(v) = forward op applied to v
    ddy = oracle.upfirdn2d(v, f, up=m['up'], down=m['down'], padding=m['padding'], flip_filter=m['flip'], gain=m['gain'])
    assert_close(ddy, UFD.t(f'c{i}_ddy'), atol=5e-6, rtol=1e-6, what=f'case {i} ddy')


def test_upfirdn2d_out_size_is_c_division():
    assert oracle.upfirdn2d_out_size(257, 1, 1, 1, 1, 4) == 256
    assert oracle.upfirdn2d_out_size(256, 1, 1, 2, 2, 4) == 257
    assert oracle.upfirdn2d_out_size(128, 2, 1, 2, 1, 4) == 256
    assert oracle.upfirdn2d_out_size(256, 1, 2, 1, 1, 4) == 128
    assert oracle.upfirdn2d_out_size(7, 1, 3, 0, 0, 2) == 2  # (7 - 2 + 3) / 3


def test_upfirdn2d_layouts_and_16bit():
    g = torch.Generator().manual_seed(0)
    x = torch.randn([2, 6, 9, 10], generator=g)
    f = torch.tensor([[1., 2., 1.], [2., 4., 2.], [1., 2., 1.]]) / 16
    y = oracle.upfirdn2d(x, f, up=2, padding=[2, 1, 1, 2])
    ycl = oracle.upfirdn2d(x.contiguous(memory_format=torch.channels_last), f, up=2, padding=[2, 1, 1, 2])
    assert ycl.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(y, ycl.contiguous())
    for dt, tol in ((torch.float16, 2e-3), (torch.bfloat16, 2e-2)):
        y16 = oracle.upfirdn2d(x.to(dt), f, up=2, padding=[2, 1, 1, 2])
        assert y16.dtype == dt
        assert_close(y16, oracle.upfirdn2d(x.to(dt).float(), f, up=2, padding=[2, 1, 1, 2]), atol=tol, rtol=tol, what=str(dt))
        # rounding of the stored result is round-to-nearest-even == torch's conversion
        exact = oracle.upfirdn2d(x.to(dt).float(), f, up=2, padding=[2, 1, 1, 2])
        assert torch.equal(y16, exact.to(dt))


@pytest.mark.parametrize('i', range(len(BA.meta)))
def test_bias_act_matches_reference(i):
    m = BA.meta[i]
    kw = dict(dim=m['dim'], act=m['act'], alpha=m['alpha'], gain=m['gain'], clamp=m['clamp'])
    x, dy, w = BA.t(f'c{i}_x'), BA.t(f'c{i}_dy'), BA.t(f'c{i}_w')
    b = BA.t(f'c{i}_b') if m['has_b'] else None
    y = oracle.bias_act(x, b, **kw)
    # alpha / gain / clamp cross the native ABI as C floats (bias_act.cpp:32): fp32-epsilon agreement on fp64 data
    assert_close(y, BA.t(f'c{i}_y'), atol=2e-6, rtol=2e-6, what='y')
    from stylegan_v_amd.torch_utils.ops.bias_act import activation_funcs
    spec = activation_funcs[m['act']]
    xref = x if ('x' in spec.ref or spec.has_2nd_grad) else None
    yref = y if 'y' in spec.ref else None
    # first derivative: grad=1 kernel form (bias_act.py:182)
    dx = oracle.bias_act(dy, b, grad=1, xref=xref, yref=yref, **kw)
    clamped_edge = ((y.abs() - (m['clamp'] if m['clamp'] is not None else float('inf'))).abs() < 1e-6)
    assert_close(dx[~clamped_edge], BA.t(f'c{i}_dx')[~clamped_edge], atol=2e-6, rtol=2e-6, what='dx')
    if m['act'] == 'linear' and m['clamp'] is not None:
        # Reference quirk kept on purpose: 'linear' saves no yref (bias_act.py:24 ref=''), so the native
        # gradient is NOT masked where the forward output saturated (bias_act.cu:141 sees yref=0), while
        # the reference's Python fallback (autograd of .clamp) masks it.  The oracle follows the kernel.
        assert_close(dx[clamped_edge], (dy * float(torch.tensor(m['gain'] or 1.0, dtype=torch.float32)))[clamped_edge], atol=1e-12, what='unmasked')
    elif m['has_b']:
        db = dx.sum([d for d in range(dx.ndim) if d != m['dim']])
        assert_close(db, BA.t(f'c{i}_db'), atol=2e-5, rtol=2e-6, what='db')
    # second order, w.r.t. dy: the same grad=1 form applied to w (bias_act.py:197-198)
    ddy = oracle.bias_act(w, b, grad=1, xref=xref, yref=yref, **kw)
    assert_close(ddy[~clamped_edge], BA.t(f'c{i}_ddy')[~clamped_edge], atol=2e-6, rtol=2e-6, what='ddy')
    # second order, w.r.t. x: grad=2 form (bias_act.py:200-201); zero for piecewise-linear activations
    if spec.has_2nd_grad:
        ddx = oracle.bias_act(w, b, grad=2, xref=xref, yref=yref, dy=dy, **kw)
        assert_close(ddx[~clamped_edge], BA.t(f'c{i}_ddx')[~clamped_edge], atol=2e-6, rtol=2e-6, what='ddx')
    else:
        assert BA.t(f'c{i}_ddx').abs().max() == 0


def test_bias_act_16bit_rounding():
    g = torch.Generator().manual_seed(1)
    x = torch.randn([3, 8, 5], generator=g)
    b = torch.randn([8], generator=g)
    for dt in (torch.float16, torch.bfloat16):
        y = oracle.bias_act(x.to(dt), b.to(dt), act='lrelu')
        exact = oracle.bias_act(x.to(dt).float(), b.to(dt).float(), act='lrelu')
        assert torch.equal(y, exact.to(dt))


@pytest.mark.parametrize('stride,transposed', [(1, False), (1, True), (2, False), (2, True)])
def test_conv3x3_oracle_vs_aten_float64(stride, transposed):
    """The reference's convolutions are ATen calls (conv2d_gradfix.py:35-43); the oracle's textbook restatement must agree with
    torch's CPU float64 kernels for the four geometries of the hot path, values and weight gradients."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(stride * 2 + transposed)
    n, k, m = 2, 5, 7
    h, w = (9, 11) if not (stride == 2 and not transposed) else (9, 13)
    x = torch.randn([n, k, h, w], generator=g, dtype=torch.float64)
    wt = torch.randn([k, m, 3, 3] if transposed else [m, k, 3, 3], generator=g, dtype=torch.float64, requires_grad=True)
    pad = 1 if stride == 1 else 0
    y = (F.conv_transpose2d if transposed else F.conv2d)(x, wt, stride=stride, padding=pad)
    got = oracle.conv3x3(x.numpy(), wt.detach().numpy(), stride=stride, transposed=transposed)
    assert got.shape == tuple(y.shape)
    assert np.abs(got - y.detach().numpy()).max() < 1e-12
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    dw, = torch.autograd.grad(y, wt, dy)
    got_dw = oracle.conv3x3_weight_grad(dy.numpy(), x.numpy(), stride=stride, transposed=transposed)
    assert np.abs(got_dw - dw.numpy()).max() < 1e-11


def test_differentiable_comparators_are_pinned():
    """oracle.dense / demod_coefs_torch / affine_resample (the float64 torch restatements the gradient and second-order GPU tests differentiate):
    dense against the reference's FullyConnectedLayer formula written out with its own operations (layers.py:126-137; the golden `networks.npz` pins the
    layers built from it end to end), the demodulation coefficients against the numpy restatement that materialises w[N,O,I,kh,kw] (networks.py:57-62),
    the resampler against the ATen entry points the reference calls (augment.py:295-297: affine_grid + grid_sample, bilinear, zeros, align_corners False)."""
    g = torch.Generator().manual_seed(5)
    x = torch.randn([7, 16], generator=g, dtype=torch.float64)
    w = torch.randn([5, 16], generator=g, dtype=torch.float64)
    b = torch.randn([5], generator=g, dtype=torch.float64)
    for act in ('linear', 'lrelu'):
        want = torch.addmm((b * 2.0).unsqueeze(0), x, (w * 0.3).t())
        if act == 'lrelu':
            want = torch.nn.functional.leaky_relu(want, 0.2) * np.sqrt(2)
        assert_close(oracle.dense(x, w, b, 0.3, 2.0, act), want, atol=1e-12, rtol=1e-12, what=f'dense {act}')
    xn = x * (x.square().mean(dim=1, keepdim=True) + 1e-8).rsqrt()
    assert_close(oracle.dense(x, w, None, 0.3, 1.0, 'linear', True), xn @ (w * 0.3).t(), atol=1e-12, rtol=1e-12, what='dense normalize')
    wt = torch.randn([6, 5, 3, 3], generator=g, dtype=torch.float64)
    s = torch.randn([4, 5], generator=g, dtype=torch.float64)
    assert_close(oracle.demod_coefs_torch(wt, s), oracle.modulated_demod_coefs(wt, s), atol=1e-12, rtol=1e-12, what='demod coefs')
    img = torch.randn([2, 3, 9, 11], generator=g, dtype=torch.float64)
    theta = torch.tensor([[0.9, 0.2, 0.05], [-0.15, 1.1, -0.1]], dtype=torch.float64).repeat(2, 1, 1) + 0.3 * torch.randn([2, 2, 3], generator=g, dtype=torch.float64)
    for out in ((8, 10), (13, 7)):
        grid = torch.nn.functional.affine_grid(theta, [2, 3, *out], align_corners=False)
        want = torch.nn.functional.grid_sample(img, grid, mode='bilinear', padding_mode='zeros', align_corners=False)
        assert_close(oracle.affine_resample(img, theta, out), want, atol=1e-12, rtol=1e-12, what=f'affine_resample {out}')
    xi = img.clone().requires_grad_(True)
    y = oracle.affine_resample(xi, theta, (8, 10))
    (gx,) = torch.autograd.grad(y.square().sum(), xi, create_graph=True)
    (g2,) = torch.autograd.grad(gx.square().sum(), xi)
    assert torch.isfinite(g2).all() and g2.abs().sum() > 0      # twice differentiable (torch's grid_sample is not)
