"""Fused stride-1 3x3 layer (ops/fused_conv_act.py -> sgv_conv3x3_fused, csrc/conv3x3_ws_kernel.h) against the ORACLE's composition
`oracle.conv3x3(x * s) * d + b -> oracle.bias_act`, forward and every first-order gradient, plus the second-order fallback."""
import numpy as np
import pytest
import torch

import oracle
from stylegan_v_amd.torch_utils import custom_ops
from stylegan_v_amd.torch_utils.ops import fused_conv_act
from util import dispatch_assert

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _oracle_layer(x, w, s, d, b, act, gain, clamp):
    """float64 composition; returns y and the pre-activation pieces the gradient formulas need."""
    xs = x.double() * (s.double()[:, :, None, None] if s is not None else 1.0)
    y0 = torch.from_numpy(oracle.conv3x3(xs.numpy(), w.double().numpy()))
    y1 = y0 * (d.double()[:, :, None, None] if d is not None else 1.0)
    y = oracle.bias_act(y1, b.double() if b is not None else None, act=act, gain=gain, clamp=clamp)
    return xs, y0, y1, y


def _oracle_grads(x, w, s, d, b, dy, act, gain, clamp, y_mask=None):
    """`y_mask`: the output whose sign / saturation pattern selects the activation derivative (bias_act.cu grad = 1 reads it from yref).
    Passing the kernel's own output removes the few elements whose pre-activation lies within rounding of the lrelu kink or the clamp
    bound from the comparison -- there the fp32 and fp64 evaluations legitimately take different branches."""
    xs, y0, y1, y = _oracle_layer(x, w, s, d, b, act, gain, clamp)
    dz = oracle.bias_act(dy.double(), b.double() if b is not None else None, act=act, gain=gain, clamp=clamp, grad=1, xref=y1, yref=y if y_mask is None else y_mask)
    db = dz.sum([0, 2, 3])
    dd = (dz * y0).sum([2, 3])
    dy0 = dz * (d.double()[:, :, None, None] if d is not None else 1.0)
    dxs = torch.from_numpy(oracle.conv3x3(dy0.numpy(), w.double().numpy(), transposed=True))   # conv_transpose2d with the same weight = the data gradient
    ds = (dxs * x.double()).sum([2, 3])
    dx = dxs * (s.double()[:, :, None, None] if s is not None else 1.0)
    dw = torch.from_numpy(oracle.conv3x3_weight_grad(dy0.numpy(), xs.numpy()))
    return y, dict(x=dx, w=dw, s=ds, d=dd, b=db)


def _rel(a, ref):
    return (a.double().cpu() - ref).abs().max().item() / max(ref.abs().max().item(), 1e-30)


@pytest.mark.parametrize('n,ci,co,h,w', [(2, 64, 64, 32, 32), (3, 48, 128, 16, 64), (1, 128, 64, 48, 32), (2, 32, 32, 32, 64), (1, 64, 96, 16, 32)])     # (the last two: c_out % 32)
@pytest.mark.parametrize('modulated,act,clamp', [(True, 'lrelu', None), (True, 'lrelu', 0.8), (False, 'lrelu', None), (True, 'linear', None), (False, 'linear', None)])
def test_fused_layer_forward_and_gradients_vs_oracle(n, ci, co, h, w, modulated, act, clamp):
    g = torch.Generator().manual_seed(n * 7 + ci + co + h)
    x = torch.randn([n, ci, h, w], generator=g)
    wt = torch.randn([co, ci, 3, 3], generator=g) / (3 * ci ** 0.5)
    s = (torch.randn([n, ci], generator=g) * 0.3 + 1) if modulated else None
    d = (torch.rand([n, co], generator=g) + 0.5) if modulated else None
    b = torch.randn([co], generator=g) * 0.5
    gain = 2 ** 0.5 if act == 'lrelu' else 1.3
    dy = torch.randn([n, co, h, w], generator=g)
    dev = lambda t: t.to(DEV).requires_grad_(True) if t is not None else None   # noqa: E731
    xg, wg, sg, dg, bg = dev(x), dev(wt), dev(s), dev(d), dev(b)
    custom_ops.prof_enable(64)
    y = fused_conv_act.conv3x3_bias_act(xg, wg, styles=sg, dcoefs=dg, bias=bg, act=act, gain=gain, clamp=clamp)
    custom_ops.prof_disable()
    prof = custom_ops.prof_collect()
    dispatch_assert(prof['conv3x3']['launches'] == 1 and prof['bias_act']['launches'] == 0 and prof['modulate']['launches'] == 0, 'the forward pass must be ONE kernel')
    y_ref, grads_ref = _oracle_grads(x, wt, s, d, b, dy, act, gain, clamp, y_mask=y.detach().double().cpu())
    # error scale = the un-clamped activation range: the convolution's rounding (4e-6 of ITS scale) passes straight through the epilogue,
    # while a clamp shrinks max |y|
    scale = oracle.bias_act(_oracle_layer(x, wt, s, d, b, act, gain, None)[2], b.double(), act=act, gain=gain).abs().max().item()
    err = (y.detach().double().cpu() - y_ref).abs().max().item() / scale
    assert err < 1e-5, f'forward: {err:.2e}'
    ins = [t for t in (xg, wg, sg, dg, bg) if t is not None]
    names = [k for k, t in zip('xwsdb', (xg, wg, sg, dg, bg)) if t is not None]
    got = torch.autograd.grad(y, ins, dy.to(DEV))
    for name, a in zip(names, got):
        err = _rel(a, grads_ref[name])
        assert err < 2e-5, f'd{name}: {err:.2e}'


def test_fused_layer_is_twice_differentiable_through_the_composition():
    g = torch.Generator().manual_seed(5)
    x = torch.randn([2, 64, 32, 32], generator=g).to(DEV).requires_grad_(True)
    wt = (torch.randn([64, 64, 3, 3], generator=g) / 24).to(DEV).requires_grad_(True)
    b = (torch.randn([64], generator=g) * 0.1).to(DEV).requires_grad_(True)

    def r1(fn):
        y = fn(x, wt, bias=b, act='lrelu')
        (gx,) = torch.autograd.grad(y.sum(), x, create_graph=True)
        return torch.autograd.grad(gx.square().sum(), [wt, b], allow_unused=True)
    got = r1(fused_conv_act.conv3x3_bias_act)
    with fused_conv_act.composition_only():
        want = r1(fused_conv_act.conv3x3_bias_act)
    for a, r in zip(got, want):
        assert (a is None) == (r is None)
        if a is not None:
            assert _rel(a, r.double().cpu()) < 1e-4


def test_fused_layer_beyond_65535_planes():
    """More than 65,535 (sample, channel) planes -- 43+ videos x 3 frames per GPU at 512 channels -- used to send the layer to the composition (a guard left from
    the days when the element-wise backward kernels put the planes on blockIdx.y in one launch; they take them in slabs now): the fused node serves them, and
    gives what the composition gives."""
    g = torch.Generator().manual_seed(8)
    n, c = 136, 512
    assert n * c > 65535
    x = torch.randn([n, c, 16, 32], generator=g).to(DEV).requires_grad_(True)
    wt = (torch.randn([c, c, 3, 3], generator=g) / (3 * c ** 0.5)).to(DEV).requires_grad_(True)
    b = (torch.randn([c], generator=g) * 0.1).to(DEV).requires_grad_(True)
    s = (torch.randn([n, c], generator=g) * 0.3 + 1).to(DEV).requires_grad_(True)
    d = (torch.rand([n, c], generator=g) + 0.5).to(DEV).requires_grad_(True)
    v = torch.randn([n, c, 16, 32], generator=g).to(DEV)

    def run():
        y = fused_conv_act.conv3x3_bias_act(x, wt, styles=s, dcoefs=d, bias=b, act='lrelu')
        return (y,) + torch.autograd.grad((y * v).sum(), [x, wt, b, s, d])
    before = custom_ops.kernel_variant_counts()
    got = run()
    after = custom_ops.kernel_variant_counts()
    dispatch_assert(after['conv_s1_ws_fused'] - before['conv_s1_ws_fused'] >= 1, 'the fused kernel did not serve the layer')
    with fused_conv_act.composition_only():
        want = run()
    for name, a, r in zip(('y', 'dx', 'dw', 'db', 'dstyles', 'ddcoefs'), got, want):
        assert _rel(a, r.double().cpu()) < 1e-4, name


def test_fused_layer_no_grad_pass_and_fallbacks():
    g = torch.Generator().manual_seed(6)
    x = torch.randn([2, 64, 32, 32], generator=g).to(DEV)
    wt = (torch.randn([64, 64, 3, 3], generator=g) / 24).to(DEV)
    s = (torch.randn([2, 64], generator=g) * 0.3 + 1).to(DEV)
    with torch.no_grad():
        before = custom_ops.launch_count()
        y = fused_conv_act.conv3x3_bias_act(x, wt, styles=s, act='lrelu')
        dispatch_assert(custom_ops.launch_count() - before == 1)          # the fused kernel (its weight preparation rides in the same accounted launch)
        with fused_conv_act.composition_only():
            yc = fused_conv_act.conv3x3_bias_act(x, wt, styles=s, act='lrelu')
    assert _rel(y, yc.double().cpu()) < 1e-5
    # shapes the fused kernel does not serve take the composition (16x16 images go to the small-image kernel + separate epilogue)
    x16 = torch.randn([2, 64, 16, 16], device=DEV)
    y16 = fused_conv_act.conv3x3_bias_act(x16, wt, act='lrelu')
    with fused_conv_act.composition_only():
        assert torch.equal(y16, fused_conv_act.conv3x3_bias_act(x16, wt, act='lrelu'))
    # linear + clamp keeps the reference's (unmasked) gradient semantics of bias_act.py:24 -> composition; tanh is not a fusable activation
    before = custom_ops.launch_count()
    fused_conv_act.conv3x3_bias_act(x, wt, act='linear', clamp=1.0)
    dispatch_assert(custom_ops.launch_count() - before == 2)              # convolution, then bias_act as its own pass
    yt = fused_conv_act.conv3x3_bias_act(x, wt, act='tanh')
    assert yt.abs().max() <= 1


# ---------------------------------------------------------------------------------------------------------------------------------------
# Fused down-sampling layer tail (ops/fused_down_act.py -> sgv_conv3x3_s2_fused, csrc/conv3x3s2_ws_kernel.h) against the oracle's composition
# `oracle.conv3x3(xb, w, stride=2) -> oracle.bias_act -> + residual`.

@pytest.mark.parametrize('n,ci,co,hs,ws', [(2, 32, 128, 8, 32), (1, 64, 256, 16, 64), (3, 16, 128, 24, 32), (4, 32, 128, 16, 16), (4, 16, 128, 8, 8)])
@pytest.mark.parametrize('act,clamp,with_res,with_bias', [('lrelu', None, True, True), ('lrelu', 0.7, True, True), ('lrelu', None, False, True),
                                                          ('linear', None, True, False), ('lrelu', 0.9, False, False)])
def test_fused_down_layer_forward_and_gradients_vs_oracle(n, ci, co, hs, ws, act, clamp, with_res, with_bias):
    from stylegan_v_amd.torch_utils.ops import fused_down_act
    g = torch.Generator().manual_seed(n * 11 + ci + co + hs)
    xb = torch.randn([n, ci, 2 * hs + 1, 2 * ws + 1], generator=g)
    wt = torch.randn([co, ci, 3, 3], generator=g) / (3 * ci ** 0.5)
    b = torch.randn([co], generator=g) * 0.5 if with_bias else None
    res = torch.randn([n, co, hs, ws], generator=g) if with_res else None
    gain = 0.5 ** 0.5 * (2 ** 0.5 if act == 'lrelu' else 1.0)
    dy = torch.randn([n, co, hs, ws], generator=g)
    dev = lambda t: t.to(DEV).requires_grad_(True) if t is not None else None   # noqa: E731
    xg, wg, bg, rg = dev(xb), dev(wt), dev(b), dev(res)
    custom_ops.prof_enable(64)
    y = fused_down_act.strided_conv3x3_bias_act(xg, wg, bias=bg, act=act, gain=gain, clamp=clamp, residual=rg.clone() if rg is not None else None)
    custom_ops.prof_disable()
    prof = custom_ops.prof_collect()
    dispatch_assert(prof['conv3x3']['launches'] == 1 and prof.get('bias_act', {}).get('launches', 0) == 0, 'the forward pass must be ONE kernel')

    y0 = torch.from_numpy(oracle.conv3x3(xb.double().numpy(), wt.double().numpy(), stride=2))
    a_ref = oracle.bias_act(y0, b.double() if b is not None else None, act=act, gain=gain, clamp=clamp)
    y_ref = a_ref + (res.double() if res is not None else 0.0)
    scale = oracle.bias_act(y0, b.double() if b is not None else None, act=act, gain=gain).abs().max().item()
    err = (y.detach().double().cpu() - y_ref).abs().max().item() / scale
    assert err < 1e-5, f'forward: {err:.2e}'

    ins = [t for t in (xg, wg, bg) if t is not None]
    names = [k for k, t in zip('xwb', (xg, wg, bg)) if t is not None]
    got = torch.autograd.grad(y, ins, dy.to(DEV))
    # the activation derivative is selected by the sign / saturation of the kernel's own activation output (see _oracle_grads above)
    # -- with a residual that is the kernel's act_out, which the same kernel without a residual returns as y (identical arithmetic)
    with torch.no_grad():
        a_gpu = fused_down_act.strided_conv3x3_bias_act(xg, wg, bias=bg, act=act, gain=gain, clamp=clamp).double().cpu()
    dz = oracle.bias_act(dy.double(), b.double() if b is not None else None, act=act, gain=gain, clamp=clamp, grad=1, xref=y0, yref=a_gpu)
    ref = dict(x=torch.from_numpy(oracle.conv3x3(dz.numpy(), wt.double().numpy(), stride=2, transposed=True)),
               w=torch.from_numpy(oracle.conv3x3_weight_grad(dz.numpy(), xb.double().numpy(), stride=2)), b=dz.sum([0, 2, 3]))
    for name, a in zip(names, got):
        err = _rel(a, ref[name])
        assert err < 3e-5, f'd{name}: {err:.2e}'
    if rg is not None:   # the residual's gradient is dy itself
        # (the residual handed to the op was a clone: differentiate through the clone)
        r2 = dev(res)
        y2 = fused_down_act.strided_conv3x3_bias_act(xg, wg, bias=bg, act=act, gain=gain, clamp=clamp, residual=r2 * 1.0)
        (gr,) = torch.autograd.grad(y2, [r2], dy.to(DEV))
        assert torch.equal(gr.cpu(), dy)


def test_fused_down_layer_second_order_and_fallbacks():
    from stylegan_v_amd.torch_utils.ops import fused_down_act
    g = torch.Generator().manual_seed(9)
    xb = torch.randn([2, 32, 17, 65], generator=g).to(DEV).requires_grad_(True)
    wt = (torch.randn([128, 32, 3, 3], generator=g) / 17).to(DEV).requires_grad_(True)
    b = (torch.randn([128], generator=g) * 0.1).to(DEV).requires_grad_(True)
    res = torch.randn([2, 128, 8, 32], generator=g).to(DEV)

    def r1(fn):
        y = fn(xb, wt, bias=b, act='lrelu', residual=res.clone())
        (gx,) = torch.autograd.grad(y.sum(), xb, create_graph=True)
        return torch.autograd.grad(gx.square().sum(), [wt, b], allow_unused=True)
    got = r1(fused_down_act.strided_conv3x3_bias_act)
    with fused_conv_act.composition_only():
        want = r1(fused_down_act.strided_conv3x3_bias_act)
    for a, r in zip(got, want):
        assert (a is None) == (r is None)
        if a is not None:
            assert _rel(a, r.double().cpu()) < 1e-4
    # 64 output channels are not served by the tap-pair kernel: composition (strided convolution, bias_act, add as separate launches)
    w64 = (torch.randn([64, 32, 3, 3], generator=g) / 17).to(DEV)
    with torch.no_grad():
        before = custom_ops.launch_count()
        y = fused_down_act.strided_conv3x3_bias_act(xb, w64, act='lrelu')
        dispatch_assert(custom_ops.launch_count() - before == 2)
        before = custom_ops.launch_count()
        yf = fused_down_act.strided_conv3x3_bias_act(xb, wt, bias=b, act='lrelu', residual=res.clone())
        dispatch_assert(custom_ops.launch_count() - before == 1)
        with fused_conv_act.composition_only():
            yc = fused_down_act.strided_conv3x3_bias_act(xb, wt, bias=b, act='lrelu', residual=res.clone())
    assert _rel(yf, yc.double().cpu()) < 1e-5 and y.shape == (2, 64, 8, 32)


# ---------------------------------------------------------------------------------------------------------------------------------------
# Stride-1 layer + the FIR in front of the next layer's strided convolution as one node (fused_conv_act.conv3x3_bias_act_then_fir): the backward
# pass runs the FIR's gradient and the activation gradient in ONE kernel (sgv_upfirdn2d_fused mode 3).

@pytest.mark.parametrize('n,ci,co,h,w', [(2, 64, 64, 32, 160), (1, 32, 64, 32, 64), (3, 16, 128, 16, 32)])
@pytest.mark.parametrize('act,clamp', [('lrelu', None), ('lrelu', 0.9), ('linear', None)])
def test_layer_then_fir_matches_the_composition(n, ci, co, h, w, act, clamp):
    from stylegan_v_amd.torch_utils.ops import upfirdn2d
    g = torch.Generator().manual_seed(n + ci + co + h + w)
    x = torch.randn([n, ci, h, w], generator=g).to(DEV).requires_grad_(True)
    wt = (torch.randn([co, ci, 3, 3], generator=g) / (3 * ci ** 0.5)).to(DEV).requires_grad_(True)
    b = (torch.randn([co], generator=g) * 0.3).to(DEV).requires_grad_(True)
    f = upfirdn2d.setup_filter([1, 3, 3, 1]).to(DEV)
    pads = (2, 2, 2, 2)
    gain = 2 ** 0.5 if act == 'lrelu' else 0.8
    y = fused_conv_act.conv3x3_bias_act_then_fir(x, wt, b, f, pads, act=act, gain=gain, clamp=clamp)
    with fused_conv_act.composition_only():
        yc = fused_conv_act.conv3x3_bias_act_then_fir(x, wt, b, f, pads, act=act, gain=gain, clamp=clamp)
    assert y.shape == (n, co, h + 1, w + 1) and _rel(y, yc.double().cpu()) < 1e-5
    dy = torch.randn(y.shape, generator=g).to(DEV)
    custom_ops.prof_enable(64)
    got = torch.autograd.grad(y, [x, wt, b], dy)
    custom_ops.prof_disable()
    prof = custom_ops.prof_collect()
    want = torch.autograd.grad(yc, [x, wt, b], dy)
    for a, r, name in zip(got, want, 'xwb'):
        # the activation mask comes from each side's own forward output: elements within rounding of the kink / clamp bound may differ
        err = _rel(a, r.double().cpu())
        assert err < 5e-4, f'd{name}: {err:.2e}'
    if w + 1 >= 129 or True:
        dispatch_assert(prof.get('modulate', {}).get('launches', 0) == 0, 'no separate activation-gradient pass')

    def r1(fn):
        yy = fn(x, wt, b, f, pads, act=act, gain=gain, clamp=clamp)
        (gx,) = torch.autograd.grad(yy.sum(), x, create_graph=True)
        return torch.autograd.grad(gx.square().sum(), [wt, b], allow_unused=True)
    got2 = r1(fused_conv_act.conv3x3_bias_act_then_fir)
    with fused_conv_act.composition_only():
        want2 = r1(fused_conv_act.conv3x3_bias_act_then_fir)
    for a, r in zip(got2, want2):
        assert (a is None) == (r is None)
        if a is not None:
            assert _rel(a, r.double().cpu()) < 1e-4


def test_layer_then_fir_sums_the_input_gradients_inside_the_data_gradient_kernel():
    """with_input_alias: the layer input's other consumer reads the node's second output; its gradient is added to by the data-gradient
    convolution's store (no separate full-tensor addition) and the total equals the composition's."""
    from stylegan_v_amd.torch_utils.ops import upfirdn2d
    g = torch.Generator().manual_seed(77)
    x = torch.randn([2, 64, 32, 64], generator=g).to(DEV).requires_grad_(True)
    wt = (torch.randn([64, 64, 3, 3], generator=g) / 24).to(DEV).requires_grad_(True)
    b = (torch.randn([64], generator=g) * 0.3).to(DEV).requires_grad_(True)
    f = upfirdn2d.setup_filter([1, 3, 3, 1]).to(DEV)
    dy = torch.randn([2, 64, 33, 65], generator=g).to(DEV)
    dz = torch.randn([2, 64, 32, 64], generator=g).to(DEV)

    def loss(alias):
        out = fused_conv_act.conv3x3_bias_act_then_fir(x, wt, b, f, (2, 2, 2, 2), act='lrelu', with_input_alias=alias)
        xb, xa = out if alias else (out, x)
        return (xb * dy).sum() + (xa * 3.0 * dz).sum()
    custom_ops.prof_enable(64)
    got = torch.autograd.grad(loss(True), [x, wt, b])
    custom_ops.prof_disable()
    dispatch_assert(custom_ops.prof_collect()['conv3x3']['launches'] == 2)          # forward and data gradient (the weight gradient is its own family)
    with fused_conv_act.composition_only():
        want = torch.autograd.grad(loss(False), [x, wt, b])
    for a, r, name in zip(got, want, 'xwb'):
        assert _rel(a, r.double().cpu()) < 5e-4, name
    # only the alias used downstream: its gradient passes through untouched
    xb, xa = fused_conv_act.conv3x3_bias_act_then_fir(x, wt, b, f, (2, 2, 2, 2), act='lrelu', with_input_alias=True)
    (gx,) = torch.autograd.grad((xa * dz).sum(), [x])
    assert torch.equal(gx, dz)
