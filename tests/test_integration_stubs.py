"""INTEGRATION.md option B, executed: the reference-side bindings in stylegan-v_amd/integration/ (drop-in ``_plugin`` module
and the conv2d_gradfix hooks), which speak only ctypes + the C ABI of include/sgv_ops.h.

* CPU, build container (needs /root/reference): the REFERENCE's own autograd classes (`_upfirdn2d_cuda`,
  `_bias_act_cuda`) are run on top of `sgv_plugin` with a stand-in library whose two entry points decode the parameter
  structs with this package's independent ctypes mirror and hand the raw pointers to the C oracle.  Forward, backward and
  double backward must equal the reference's `impl='ref'` results: that executes the binding's marshalling (argument order,
  struct layout, "empty tensor = absent" convention) against the reference's real call sites.
* GPU: the same stub functions against libsgv_hip.so itself, with the reference's calling conventions, vs the oracle;
  the conv hooks vs the float64 oracle.
"""
import ctypes
import importlib
import os
import sys

import numpy as np
import pytest
import torch

import oracle
from stylegan_v_amd.integration import sgv_plugin, sgv_conv
from stylegan_v_amd.torch_utils import custom_ops
from util import assert_bit_equal, assert_close

REF = '/root/reference'


class _OracleBackedLib:
    """Looks like libsgv_hip.so to sgv_plugin (sgv_upfirdn2d / sgv_bias_act / sgv_last_error) but computes on HOST pointers with
    the C oracle.  The structs are re-read through custom_ops' ctypes declarations, not sgv_plugin's own."""

    def __init__(self):
        from oracle import oracle as omod
        self.olib = omod._get()
        self.calls = 0

    def sgv_last_error(self):
        return b'oracle-backed stand-in failed'

    def sgv_upfirdn2d(self, pref, dtype, stream):
        p = ctypes.cast(pref, ctypes.POINTER(custom_ops.Upfirdn2dParams)).contents
        self.calls += 1
        return self.olib.oracle_upfirdn2d(p.x, p.f, p.y, dtype, p.up_x, p.up_y, p.down_x, p.down_y, p.pad_x0, p.pad_x1, p.pad_y0, p.pad_y1, p.flip, p.gain,
                                          p.in_w, p.in_h, p.in_c, p.in_n, p.in_sw, p.in_sh, p.in_sc, p.in_sn, p.f_w, p.f_h, p.f_sw, p.f_sh,
                                          p.out_sw, p.out_sh, p.out_sc, p.out_sn)

    def sgv_bias_act(self, pref, dtype, stream):
        p = ctypes.cast(pref, ctypes.POINTER(custom_ops.BiasActParams)).contents
        self.calls += 1
        return self.olib.oracle_bias_act(p.x, p.b, p.xref, p.yref, p.dy, p.y, dtype, p.grad, p.act, p.alpha, p.gain, p.clamp, p.size_x, max(p.size_b, 1), p.step_b)


@pytest.fixture
def reference_ops(monkeypatch):
    if not os.path.isdir(REF):
        pytest.skip('reference checkout not present (GPU box)')
    monkeypatch.setattr(sys, 'dont_write_bytecode', True)
    for path in (os.path.join(REF, 'src'), REF):
        monkeypatch.syspath_prepend(path)
    saved = {k: v for k, v in sys.modules.items() if k == 'src' or k.startswith('src.')}
    for k in saved:
        del sys.modules[k]
    R_ufd = importlib.import_module('src.torch_utils.ops.upfirdn2d')
    R_ba = importlib.import_module('src.torch_utils.ops.bias_act')
    fake = _OracleBackedLib()
    monkeypatch.setattr(sgv_plugin, '_lib', fake)
    monkeypatch.setattr(sgv_plugin, '_stream', lambda t: None)
    # the maintainer's edit of upfirdn2d.py:_init / bias_act.py:_init, applied from outside
    for mod in (R_ufd, R_ba):
        monkeypatch.setattr(mod, '_plugin', sgv_plugin)
        monkeypatch.setattr(mod, '_inited', True)
    yield R_ufd, R_ba, fake
    for k in [k for k in sys.modules if k == 'src' or k.startswith('src.')]:
        del sys.modules[k]
    sys.modules.update(saved)


def test_reference_upfirdn2d_autograd_runs_on_the_stub(reference_ops):
    R_ufd, _, fake = reference_ops
    g = torch.Generator().manual_seed(0)
    f = R_ufd.setup_filter([1, 3, 3, 1])
    for up, down, pad, flip, gain in [(1, 1, [1, 1, 1, 1], False, 4), (2, 1, [2, 1, 2, 1], False, 4), (1, 2, [1, 1, 1, 1], True, 1), ((2, 1), (1, 2), [1, 0, 2, 1], False, 1.5)]:
        x = torch.randn([2, 3, 10, 12], generator=g, dtype=torch.float64)
        upx, upy = R_ufd._parse_scaling(up)
        dnx, dny = R_ufd._parse_scaling(down)

        def run(fn):
            xx = x.clone().requires_grad_(True)
            y = fn(xx)
            dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(1), dtype=torch.float64).requires_grad_(True)
            (dx,) = torch.autograd.grad(y, xx, dy, create_graph=True)
            (ddy,) = torch.autograd.grad(dx.square().sum(), dy)
            return y, dx, ddy
        calls = fake.calls
        got = run(lambda xx: R_ufd._upfirdn2d_cuda(up=up, down=down, padding=pad, flip_filter=flip, gain=gain).apply(xx, f))   # reference autograd class (upfirdn2d.py:213-264) -> stub
        assert fake.calls - calls == 3, 'forward, backward and double backward must each reach the plugin once'
        want = run(lambda xx: R_ufd.upfirdn2d(xx, f, up=up, down=down, padding=pad, flip_filter=flip, gain=gain, impl='ref'))
        for a, r, name in zip(got, want, ['y', 'dx', 'ddy']):
            assert_close(a, r, atol=1e-12, rtol=1e-12, what=f'upfirdn2d up={up} down={down} {name}')
    # separable filter: two plugin calls with sqrt(gain) each (upfirdn2d.py:239-240)
    x = torch.randn([1, 2, 9, 11], generator=g, dtype=torch.float64)
    f1 = torch.tensor([1., 3., 3., 1.]) / 8
    y = R_ufd._upfirdn2d_cuda(up=2, padding=[2, 1, 2, 1], gain=4).apply(x, f1)
    assert_close(y, R_ufd.upfirdn2d(x, f1, up=2, padding=[2, 1, 2, 1], gain=4, impl='ref'), atol=1e-12, rtol=1e-12)


@pytest.mark.parametrize('act', ['linear', 'relu', 'lrelu', 'tanh', 'sigmoid', 'elu', 'selu', 'softplus', 'swish'])
def test_reference_bias_act_autograd_runs_on_the_stub(reference_ops, act):
    _, R_ba, fake = reference_ops
    g = torch.Generator().manual_seed(3)
    for has_b, clamp, gain in [(True, None, None), (True, 0.7, 1.7), (False, 0.5, 0.9)]:
        x = torch.randn([2, 5, 6, 6], generator=g, dtype=torch.float64)
        b = torch.randn([5], generator=g, dtype=torch.float64) if has_b else None

        def run(impl):
            xx = x.clone().requires_grad_(True)
            bb = b.clone().requires_grad_(True) if b is not None else None
            if impl == 'stub':
                y = R_ba._bias_act_cuda(dim=1, act=act, alpha=None, gain=gain, clamp=clamp).apply(xx, bb if bb is not None else R_ba._null_tensor)   # bias_act.py:85-89
            else:
                y = R_ba.bias_act(xx, bb, act=act, gain=gain, clamp=clamp, impl='ref')
            dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(4), dtype=torch.float64).requires_grad_(True)
            ins = [xx] + ([bb] if bb is not None else [])
            grads = torch.autograd.grad(y, ins, dy, create_graph=True)
            second = torch.autograd.grad(grads[0].square().sum(), [dy, xx], allow_unused=True)
            return [y, *grads, second[0], second[1] if second[1] is not None else torch.zeros_like(xx)]
        calls = fake.calls
        got = run('stub')
        assert fake.calls > calls
        want = run('ref')
        names = ['y', 'dx'] + (['db'] if has_b else []) + ['ddy', 'ddx']
        for a, r, name in zip(got, want, names):
            if act == 'linear' and clamp is not None and name in ('dx', 'db', 'ddy'):
                continue   # (alpha, gain, clamp cross the plugin boundary as C floats, bias_act.cpp:32: 1e-6, not 1e-12)  reference behaviour: the native linear+clamp gradient is not masked (bias_act.py:24 saves no y), its Python fallback is
            assert_close(a, r, atol=1e-6, rtol=1e-6, what=f'bias_act {act} b={has_b} clamp={clamp} {name}')


def test_stub_signatures_match_the_reference_call_sites():
    """Static check that needs no library: positional arity of the reference's `_plugin.*` call sites == the stub's signatures."""
    if not os.path.isdir(REF):
        pytest.skip('reference checkout not present (GPU box)')
    import ast
    import inspect
    want = {'upfirdn2d': len(inspect.signature(sgv_plugin.upfirdn2d).parameters), 'bias_act': len(inspect.signature(sgv_plugin.bias_act).parameters)}
    seen = {'upfirdn2d': 0, 'bias_act': 0}
    for fname in ('upfirdn2d.py', 'bias_act.py'):
        tree = ast.parse(open(os.path.join(REF, 'src', 'torch_utils', 'ops', fname)).read())
        for node in ast.walk(tree):
            if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and isinstance(node.func.value, ast.Name) and node.func.value.id == '_plugin':
                assert len(node.args) == want[node.func.attr] and not node.keywords, f'{fname}:{node.lineno}'
                seen[node.func.attr] += 1
    assert seen == {'upfirdn2d': 3, 'bias_act': 3}


# ------------------------------------------------------------------------------------------------------------------
# GPU: the stubs on the real library

@pytest.mark.gpu
def test_plugin_stub_on_gpu_vs_oracle():
    g = torch.Generator().manual_seed(0)
    f = torch.tensor([1., 3., 3., 1.])
    f = (f[:, None] * f[None, :]) / 64
    before = custom_ops.launch_count()
    for dtype in (torch.float32, torch.float16, torch.bfloat16, torch.float64):
        x = torch.randn([2, 3, 33, 35], generator=g).to(dtype)
        for args in [(1, 1, 1, 1, 1, 1, 1, 1, False, 4.0), (2, 2, 1, 1, 2, 1, 2, 1, False, 4.0), (1, 1, 2, 2, 1, 1, 1, 1, True, 1.0), (1, 1, 1, 1, 2, 2, 2, 2, True, 4.0)]:
            y = sgv_plugin.upfirdn2d(x.cuda(), f.cuda(), *args)
            upx, upy, dnx, dny, px0, px1, py0, py1, flip, gain = args
            ref = oracle.upfirdn2d(x, f, up=(upx, upy), down=(dnx, dny), padding=[px0, px1, py0, py1], flip_filter=flip, gain=gain)
            assert_bit_equal(y.cpu(), ref, f'stub upfirdn2d {dtype} {args}')
        xc = x.cuda().contiguous(memory_format=torch.channels_last)
        y = sgv_plugin.upfirdn2d(xc, f.cuda(), 1, 1, 1, 1, 1, 1, 1, 1, False, 4.0)
        assert y.is_contiguous(memory_format=torch.channels_last)
        assert_bit_equal(y.cpu().contiguous(), oracle.upfirdn2d(x, f, padding=1, gain=4.0), 'stub upfirdn2d channels_last')
    null = torch.empty([0], device='cuda')
    x = torch.randn([2, 5, 6, 6], generator=g)
    b = torch.randn([5], generator=g)
    y = sgv_plugin.bias_act(x.cuda(), b.cuda(), null, null, null, 0, 1, 3, 0.2, 2 ** 0.5, 0.7)          # lrelu forward, as bias_act.py:153
    yr = oracle.bias_act(x, b, act='lrelu', clamp=0.7)
    assert_bit_equal(y.cpu(), yr, 'stub bias_act forward')
    dy = torch.randn(x.shape, generator=g)
    dx = sgv_plugin.bias_act(dy.cuda(), b.cuda(), x.cuda(), y, null, 1, 1, 3, 0.2, 2 ** 0.5, 0.7)       # grad 1, as bias_act.py:182
    assert_bit_equal(dx.cpu(), oracle.bias_act(dy, b, act='lrelu', clamp=0.7, grad=1, xref=x, yref=yr), 'stub bias_act grad 1')
    y0 = sgv_plugin.bias_act(x.cuda(), null, null, null, null, 0, 1, 1, 0.0, 1.5, -1.0)                # no bias: empty tensor = absent
    assert_bit_equal(y0.cpu(), oracle.bias_act(x, None, act='linear', gain=1.5), 'stub bias_act without bias')
    assert custom_ops.launch_count() - before >= 20, 'the stub did not reach the library this process has loaded'
    with pytest.raises(RuntimeError, match='output must be at least 1x1'):
        sgv_plugin.upfirdn2d(torch.zeros([1, 1, 2, 2], device='cuda'), f.cuda(), 1, 1, 1, 1, 0, 0, 0, 0, False, 1.0)


@pytest.mark.gpu
def test_conv_stub_on_gpu_vs_oracle():
    g = torch.Generator().manual_seed(1)
    x = torch.randn([2, 64, 16, 32], generator=g)
    w = torch.randn([64, 64, 3, 3], generator=g) / 24
    dev = lambda t: t.cuda()

    def close(a, ref, what):
        err = np.abs(a.double().cpu().numpy() - ref).max() / np.abs(ref).max()
        assert err < 1e-5, f'{what}: {err:.2e}'
    y = sgv_conv.conv3x3(dev(x), dev(w), False, 1)
    close(y, oracle.conv3x3(x.numpy(), w.numpy()), 'conv3x3')
    close(sgv_conv.conv3x3(dev(x), dev(w), True, 1), oracle.conv3x3(x.numpy(), w.numpy(), transposed=True), 'conv3x3 data gradient form')
    xb = torch.randn([2, 64, 17, 65], generator=g)
    close(sgv_conv.conv3x3(dev(xb), dev(w), False, 2), oracle.conv3x3(xb.numpy(), w.numpy(), stride=2), 'strided')
    yt = sgv_conv.conv3x3(dev(x[:, :, :8]), dev(w), True, 2)
    close(yt, oracle.conv3x3(x[:, :, :8].numpy(), w.numpy(), stride=2, transposed=True), 'transposed')
    dy = torch.randn(y.shape, generator=g)
    close(sgv_conv.conv3x3_weight_grad(dev(dy), dev(x), w.shape, 1), oracle.conv3x3_weight_grad(dy.numpy(), x.numpy()), 'weight gradient')
    dys = torch.randn([2, 64, 8, 32], generator=g)
    close(sgv_conv.conv3x3_weight_grad(dev(dys), dev(xb), w.shape, 2), oracle.conv3x3_weight_grad(dys.numpy(), xb.numpy(), stride=2), 'stride-2 weight gradient')
    assert sgv_conv.conv3x3(dev(x[:, :3]), dev(w[:, :3]), False, 1) is None, 'unsupported shapes must hand back to the vendor library'
    assert sgv_conv.conv3x3(dev(x).half(), dev(w).half(), False, 1) is None


@pytest.mark.gpu
def test_augment_stub_on_gpu_vs_the_reference_fixture():
    """integration/sgv_augment.py (the reference-side replacement of augment.py:284-303, ctypes only) against the reference's own evaluation of the block and of its
    first / second derivatives (tests/golden/ada_geometric.npz)."""
    from stylegan_v_amd.integration import sgv_augment
    from util import Golden, assert_close
    geo = Golden('ada_geometric')
    taps = geo.t('taps', device='cuda')
    theta, margin = geo.t('theta', device='cuda'), geo.meta['margin']
    before = custom_ops.launch_count()
    x = geo.t('x', device='cuda').requires_grad_(True)
    y = sgv_augment.ada_geometric(x, theta, taps, margin)
    assert_close(y.detach(), geo.t('y', device='cuda'), atol=3e-5, rtol=3e-5, what='stub forward vs the reference')
    (dx,) = torch.autograd.grad((y * geo.t('v', device='cuda')).sum(), x)
    assert_close(dx, geo.t('dx', device='cuda'), atol=5e-5, rtol=5e-5, what='stub backward vs the reference')
    x = geo.t('x', device='cuda').requires_grad_(True)
    y = sgv_augment.ada_geometric(x, theta, taps, margin)
    (g1,) = torch.autograd.grad((y ** 3).sum(), x, create_graph=True)
    (g2,) = torch.autograd.grad(g1.square().sum(), x)
    want1, want2 = geo.t('r1_g', device='cuda'), geo.t('r1_gg', device='cuda')
    assert_close(g1.detach(), want1, atol=1e-4 * want1.abs().max().item(), rtol=1e-4, what='stub first derivative of the cubic head')
    assert_close(g2, want2, atol=2e-4 * want2.abs().max().item(), rtol=2e-4, what='stub R1-shaped second derivative')
    assert custom_ops.launch_count() - before == 6, 'forward, adjoint; forward, adjoint, forward, adjoint: the stub reached the library this process has loaded'
