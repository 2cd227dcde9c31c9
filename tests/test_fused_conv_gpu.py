"""Fused stride-1 3x3 layer (ops/fused_conv_act.py -> sgv_conv3x3_fused, csrc/conv3x3_ws_kernel.h) against the ORACLE's composition
`oracle.conv3x3(x * s) * d + b -> oracle.bias_act`, forward and every first-order gradient, plus the second-order fallback."""
import numpy as np
import pytest
import torch

import oracle
from stylegan_v_amd.torch_utils import custom_ops
from stylegan_v_amd.torch_utils.ops import fused_conv_act

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


@pytest.mark.parametrize('n,ci,co,h,w', [(2, 64, 64, 32, 32), (3, 48, 128, 16, 64), (1, 128, 64, 48, 32)])
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
    assert prof['conv3x3']['launches'] == 1 and prof['bias_act']['launches'] == 0 and prof['modulate']['launches'] == 0, 'the forward pass must be ONE kernel'
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


def test_fused_layer_no_grad_pass_and_fallbacks():
    g = torch.Generator().manual_seed(6)
    x = torch.randn([2, 64, 32, 32], generator=g).to(DEV)
    wt = (torch.randn([64, 64, 3, 3], generator=g) / 24).to(DEV)
    s = (torch.randn([2, 64], generator=g) * 0.3 + 1).to(DEV)
    with torch.no_grad():
        before = custom_ops.launch_count()
        y = fused_conv_act.conv3x3_bias_act(x, wt, styles=s, act='lrelu')
        assert custom_ops.launch_count() - before == 1          # the fused kernel (its weight preparation rides in the same accounted launch)
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
    assert custom_ops.launch_count() - before == 2              # convolution, then bias_act as its own pass
    yt = fused_conv_act.conv3x3_bias_act(x, wt, act='tanh')
    assert yt.abs().max() <= 1
