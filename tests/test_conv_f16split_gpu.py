"""The block-scaled fp16 split (terms = 4; csrc/sgv_split.h) -- what is specific to it: the magnitude-bound pass (sgv_absmax), dynamic range, loose
bounds, the bound cache of the host layer, and a side-by-side accuracy table (fp16 split | bf16 split | vendor fp32) against float64 for every member
of the family.  The geometry / indexing tests of the family run under both arithmetics in test_conv3x3_gpu.py, test_conv_wrw_gpu.py, test_fused_conv_gpu.py.

Reference: the reference computes these convolutions in strict fp32 (src/training/training_loop.py:129,141-142, `allow_tf32 = False`)."""
import json
import os

import pytest
import torch
import torch.nn.functional as F

from stylegan_v_amd.torch_utils import custom_ops
from stylegan_v_amd.torch_utils.ops import amax, conv2d_gradfix, fused_conv_act
from util import dispatch_assert

pytestmark = pytest.mark.gpu
DEV = 'cuda'
S1 = (False, (1, 1), (1, 1), (0, 0), (1, 1), 1)
S1T = (True, (1, 1), (1, 1), (0, 0), (1, 1), 1)
S2 = (False, (2, 2), (0, 0), (0, 0), (1, 1), 1)
S2T = (True, (2, 2), (0, 0), (0, 0), (1, 1), 1)


def _rel(a, ref):
    a, ref = a.double().cpu(), ref.double().cpu()
    return ((a - ref).norm() / ref.norm()).item(), ((a - ref).abs().max() / ref.abs().max()).item()


def _with_terms(terms, fn):
    saved = (conv2d_gradfix.native_conv_terms, conv2d_gradfix.native_wrw_terms)
    conv2d_gradfix.native_conv_terms = conv2d_gradfix.native_wrw_terms = terms
    try:
        return fn()
    finally:
        conv2d_gradfix.native_conv_terms, conv2d_gradfix.native_wrw_terms = saved


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16, torch.bfloat16])
def test_absmax_is_the_maximum_magnitude(dtype):
    lib = custom_ops.get_native()
    g = torch.Generator().manual_seed(1)
    code = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}[dtype]
    for numel in (1, 3, 255, 1024, 4097, 70001, 3_000_001):
        base = (torch.randn([numel + 1], generator=g) * 3).to(dtype).to(DEV)
        for t in (base[:-1], base[1:]):                    # 16-byte aligned start, and a view that is not
            out = torch.full([1], 123.0, device=DEV)
            custom_ops.check(lib.sgv_absmax(t.data_ptr(), t.numel(), code, out.data_ptr(), 0, custom_ops.raw_stream(t)), lib)
            want = t.float().abs().max().item()
            assert out.item() == want, (numel, dtype, t.data_ptr() % 16, out.item(), want, int(t.float().abs().argmax()))
    # accumulate keeps a larger previous bound; an empty tensor leaves zero; inf / NaN come out as non-finite bounds
    t = torch.randn([1000], generator=g).to(dtype).to(DEV)
    out = torch.full([1], 77.0, device=DEV)
    custom_ops.check(lib.sgv_absmax(t.data_ptr(), t.numel(), code, out.data_ptr(), 1, custom_ops.raw_stream(t)), lib)
    assert out.item() == 77.0
    custom_ops.check(lib.sgv_absmax(t.data_ptr(), 0, code, out.data_ptr(), 0, custom_ops.raw_stream(t)), lib)
    assert out.item() == 0.0
    for bad in (float('inf'), float('nan')):
        t2 = t.clone(); t2[500] = bad
        custom_ops.check(lib.sgv_absmax(t2.data_ptr(), t2.numel(), code, out.data_ptr(), 0, custom_ops.raw_stream(t2)), lib)
        assert not torch.isfinite(out).item()
    assert amax.bound(torch.zeros([64], device=DEV)).item() == 0.0


def test_bound_cache_follows_the_tensor():
    x = torch.randn([4, 64, 32, 32], device=DEV)
    b1 = amax.bound(x)
    assert amax.bound(x) is b1, 'a second request must not launch another pass'
    x.mul_(2)                                        # the version counter moved: the bound is recomputed
    b2 = amax.bound(x)
    assert b2 is not b1 and b2.item() == x.abs().max().item()
    amax.invalidate(x)
    assert amax.bound(x) is not b2
    nc = x.permute(0, 2, 3, 1)                       # any dense layout: the bound is over the elements
    assert amax.bound(nc).item() == x.abs().max().item()


def _family_cases():
    """(name, run(terms) -> result, float64 reference): one representative call of every member of the 3x3 family and of the tiled GEMM."""
    g = torch.Generator().manual_seed(7)
    def t(*shape, scale=1.0, shift=0.0):
        return (torch.randn(list(shape), generator=g) * scale + shift).to(DEV)
    cases = []
    x, w = t(2, 64, 32, 64, shift=0.3), t(128, 64, 3, 3, scale=1 / 24)
    cases.append(('s1 forward', lambda: conv2d_gradfix._native_conv(x, w, S1), F.conv2d(x.double().cpu(), w.double().cpu(), padding=1)))
    wt = t(64, 128, 3, 3, scale=1 / 24)
    cases.append(('s1 data gradient', lambda: conv2d_gradfix._native_conv(x, wt, S1T), F.conv_transpose2d(x.double().cpu(), wt.double().cpu(), padding=1)))
    xs, ws_ = t(4, 64, 16, 16, shift=0.3), t(64, 64, 3, 3, scale=1 / 24)
    cases.append(('16x16 images', lambda: conv2d_gradfix._native_conv(xs, ws_, S1), F.conv2d(xs.double().cpu(), ws_.double().cpu(), padding=1)))
    xb, wb = t(2, 64, 33, 65, shift=0.3), t(128, 64, 3, 3, scale=1 / 24)
    cases.append(('strided (tap pairs)', lambda: conv2d_gradfix._native_conv(xb, wb, S2), F.conv2d(xb.double().cpu(), wb.double().cpu(), stride=2)))
    wb64 = t(64, 64, 3, 3, scale=1 / 24)
    cases.append(('strided (64-channel tile)', lambda: conv2d_gradfix._native_conv(xb, wb64, S2), F.conv2d(xb.double().cpu(), wb64.double().cpu(), stride=2)))
    xt, wtt = t(2, 64, 16, 32, shift=0.3), t(64, 64, 3, 3, scale=1 / 24)
    cases.append(('transposed', lambda: conv2d_gradfix._native_conv(xt, wtt, S2T), F.conv_transpose2d(xt.double().cpu(), wtt.double().cpu(), stride=2)))
    xp = t(4, 64, 8, 16, shift=0.3)
    cases.append(('transposed, packed samples', lambda: conv2d_gradfix._native_conv(xp, wtt, S2T), F.conv_transpose2d(xp.double().cpu(), wtt.double().cpu(), stride=2)))
    xq, wq = t(4, 32, 33, 33, shift=0.3), t(128, 32, 3, 3, scale=1 / 17)
    cases.append(('strided, packed samples', lambda: conv2d_gradfix._native_conv(xq, wq, S2), F.conv2d(xq.double().cpu(), wq.double().cpu(), stride=2)))
    dy = t(2, 128, 32, 64)
    def ref_dw(dy_, x_, co, ci, **kw):
        wz = torch.zeros([co, ci, 3, 3], dtype=torch.float64, requires_grad=True)
        return torch.autograd.grad(F.conv2d(x_.double().cpu(), wz, **kw), wz, dy_.double().cpu())[0]
    cases.append(('s1 weight gradient', lambda: conv2d_gradfix._native_wrw(dy, x, S1, (128, 64, 3, 3)), ref_dw(dy, x, 128, 64, padding=1)))
    sc = (torch.rand([2, 64], generator=g) + 0.5).to(DEV)
    cases.append(('s1 weight gradient, input scale', lambda: conv2d_gradfix._native_wrw(dy, x, S1, (128, 64, 3, 3), x_scale=sc),
                  ref_dw(dy, x.double().cpu() * sc.double().cpu()[:, :, None, None], 128, 64, padding=1)))
    dys = t(2, 128, 16, 32)
    cases.append(('s2 weight gradient', lambda: conv2d_gradfix._native_wrw(dys, xb, S2, (128, 64, 3, 3)), ref_dw(dys, xb, 128, 64, stride=2)))
    dyp = t(4, 64, 16, 16)
    cases.append(('s1 weight gradient, packed samples', lambda: conv2d_gradfix._native_wrw(dyp, xs, S1, (64, 64, 3, 3)), ref_dw(dyp, xs, 64, 64, padding=1)))
    # a whole modulated layer: x * styles -> conv -> * dcoefs -> + bias -> lrelu * sqrt(2)
    st, dc, bi = (torch.rand([2, 64], generator=g) + 0.5).to(DEV), (torch.rand([2, 128], generator=g) + 0.5).to(DEV), t(128, scale=0.1)
    def layer_ref():
        y = F.conv2d(x.double().cpu() * st.double().cpu()[:, :, None, None], w.double().cpu(), padding=1) * dc.double().cpu()[:, :, None, None] + bi.double().cpu()[None, :, None, None]
        return F.leaky_relu(y, 0.2) * 2 ** 0.5
    cases.append(('fused modulated layer', lambda: fused_conv_act.conv3x3_bias_act(x, w, styles=st, dcoefs=dc, bias=bi, act='lrelu'), layer_ref()))
    # dense 1x1 skip product on the tiled GEMM
    from stylegan_v_amd.torch_utils.ops import gemm
    xg, wg = t(2, 256, 32, 32, shift=0.3), t(512, 256, 1, 1, scale=1 / 16)
    cases.append(('1x1 skip GEMM', lambda: gemm.conv1x1(xg, wg), F.conv2d(xg.double().cpu(), wg.double().cpu())))
    return cases


def test_every_member_is_fp32_grade_under_the_fp16_split():
    """rel-L2 <= 5e-7 against float64 for every member ('MIOpen-class'), next to the bf16 split and, where the vendor library has the
    call, its fp32 result -- written to $SGV_ERROR_TABLE_DIR/conv_terms_accuracy.json (-> profiles/r04_conv_terms_accuracy.json)."""
    table = {}
    for name, run, ref in _family_cases():
        row = {}
        for terms in (4, 3):
            before = custom_ops.launch_count()
            got = _with_terms(terms, run)
            dispatch_assert(custom_ops.launch_count() > before, f'{name}: no native launch')
            row[f'terms{terms}'] = _rel(got, ref)
        table[name] = row
        assert row['terms4'][0] < (1e-6 if 'weight gradient' in name else 5e-7) and row['terms4'][1] < 2e-6, (name, row)
        assert row['terms3'][0] < 1e-5, (name, row)
        assert row['terms4'][0] < row['terms3'][0] / 8, (name, row)       # the point of the exercise
    print(json.dumps(table, indent=1))
    out_dir = os.environ.get('SGV_ERROR_TABLE_DIR')
    if out_dir:
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, 'conv_terms_accuracy.json'), 'w') as fh:
            json.dump(dict(what='rel-L2, max-norm error against float64; terms4 = block-scaled fp16 split, terms3 = bf16 split', rows=table), fh, indent=1)


@pytest.mark.parametrize('x_scale,w_scale', [(1e-20, 1.0), (1e20, 1e-3), (3e-6, 7e4), (1.0, 1e-30), (6e4, 6e4)])
def test_dynamic_range_does_not_matter(x_scale, w_scale):
    """fp16 has 5 exponent bits: the block scale (a power of two per tensor, from its bound) must make the tensor's own magnitude irrelevant --
    gradients of 1e-20, weights of 1e-30, activations beyond fp16's maximum."""
    g = torch.Generator().manual_seed(11)
    x = ((torch.randn([2, 64, 16, 32], generator=g) + 0.3) * x_scale).to(DEV)
    w = (torch.randn([64, 64, 3, 3], generator=g) / 24 * w_scale).to(DEV)
    for cfg, ref_op, kw in ((S1, F.conv2d, dict(padding=1)), (S2T, F.conv_transpose2d, dict(stride=2))):
        y = _with_terms(4, lambda: conv2d_gradfix._native_conv(x, w, cfg))
        ref = ref_op(x.double().cpu(), w.double().cpu(), **kw)
        assert torch.isfinite(y).all()
        l2, mx = _rel(y, ref)
        assert l2 < 5e-7 and mx < 1.5e-6, (cfg, x_scale, w_scale, l2, mx)
    dy = (torch.randn([2, 64, 16, 32], generator=g) * x_scale).to(DEV)
    xx = (torch.randn([2, 64, 16, 32], generator=g) * w_scale + w_scale).to(DEV)
    dw = _with_terms(4, lambda: conv2d_gradfix._native_wrw(dy, xx, S1, (64, 64, 3, 3)))
    wz = torch.zeros([64, 64, 3, 3], dtype=torch.float64, requires_grad=True)
    ref = torch.autograd.grad(F.conv2d(xx.double().cpu(), wz, padding=1), wz, dy.double().cpu())[0]
    if ref.abs().max().item() > 1e-37:      # (a result below fp32's normal range cannot be compared relatively)
        l2, mx = _rel(dw, ref)
        assert l2 < 5e-7 and mx < 1.5e-6, (x_scale, w_scale, l2, mx)


def test_outliers_and_heavy_tails_keep_the_small_values():
    """One element 2^14 above the rest sets the block scale; the rest must keep fp32-grade relative precision (the split's absolute floor is 2^-39 of
    the bound, sgv_split.h), measured on the outputs that do not see the outlier."""
    g = torch.Generator().manual_seed(12)
    x = (torch.randn([2, 64, 32, 32], generator=g) * torch.exp(2 * torch.randn([2, 64, 1, 1], generator=g))).to(DEV)     # heavy-tailed channel scales
    x[0, 5, 3, 3] = 16384.0 * x.abs().max()
    w = (torch.randn([64, 64, 3, 3], generator=g) / 24).to(DEV)
    y = _with_terms(4, lambda: conv2d_gradfix._native_conv(x, w, S1))
    ref = F.conv2d(x.double().cpu(), w.double().cpu(), padding=1)
    clean = torch.ones_like(ref, dtype=torch.bool)
    clean[0, :, 2:5, 2:5] = False
    err = ((y.double().cpu() - ref)[clean].norm() / ref[clean].norm()).item()
    assert err < 1e-6, err            # measured 5.5e-7 (2.7e-7 without the outlier): an outlier 2^14 above everything else costs one bit, not the tensor
    assert _rel(y, ref)[0] < 1e-6


def test_a_loose_bound_costs_nothing_and_the_bound_may_be_a_product():
    """Any bound >= max |x| within ~2^10 keeps the accuracy (so a fused layer may bound x * styles by bound(x) * bound(styles)); checked through the C ABI
    with hand-made bounds."""
    lib = custom_ops.get_native()
    g = torch.Generator().manual_seed(13)
    x = (torch.randn([2, 64, 16, 32], generator=g) + 0.3).to(DEV)
    w = (torch.randn([64, 64, 3, 3], generator=g) / 24).to(DEV)
    ref = F.conv2d(x.double().cpu(), w.double().cpu(), padding=1)
    ws_bytes = int(lib.sgv_conv3x3_workspace_bytes(64, 64))
    for loosen in (1.0, 3.0, 1000.0):
        bound = (x.abs().max() * loosen).reshape(1).float()
        y = torch.empty_like(x)
        ws = torch.empty([ws_bytes], dtype=torch.uint8, device=DEV)
        p = custom_ops.Conv3x3Params(x.data_ptr(), w.data_ptr(), y.data_ptr(), ws.data_ptr(), ws_bytes, 2, 64, 64, 16, 32, 0, 4, bound.data_ptr(), None)
        custom_ops.check(lib.sgv_conv3x3(p, 0, custom_ops.raw_stream(x)), lib)
        assert _rel(y, ref)[0] < 5e-7, loosen
    # terms = 4 without a bound is an argument error, not a silent fallback
    p = custom_ops.Conv3x3Params(x.data_ptr(), w.data_ptr(), y.data_ptr(), ws.data_ptr(), ws_bytes, 2, 64, 64, 16, 32, 0, 4, None, None)
    assert lib.sgv_conv3x3(p, 0, custom_ops.raw_stream(x)) == -1 and b'x_amax' in lib.sgv_last_error()


def test_accumulating_store_replaces_the_cached_bound():
    """_FusedConvActFirFn's data gradient adds INTO the skip branch's gradient through its raw pointer; a bound cached on that tensor before must not
    survive (the next convolution upstream would scale by a stale, possibly too small bound).  The kernel's store leaves the increment's bound behind, and
    bound(old) + bound(increment) becomes the sum's bound without a pass over it; a target without a bound stays without one."""
    g = torch.Generator().manual_seed(14)
    dz = torch.randn([2, 64, 16, 32], generator=g).to(DEV)
    w = (torch.randn([64, 64, 3, 3], generator=g) / 24).to(DEV)
    acc = (torch.randn([2, 64, 16, 32], generator=g) * 1e-3).to(DEV)
    small = amax.bound(acc)
    want = acc.double().cpu() + F.conv_transpose2d(dz.double().cpu(), w.double().cpu(), padding=1)
    inc_max = F.conv_transpose2d(dz.double().cpu(), w.double().cpu(), padding=1).abs().max().item()
    out = _with_terms(4, lambda: fused_conv_act._launch_fused(dz, w, None, None, None, 1, 0.0, 1.0, -1.0, mode=1, accumulate_into=acc))
    assert out is acc and _rel(acc, want)[0] < 1e-6
    fresh = amax.cached(acc)
    assert fresh is not None and fresh is not small, 'the store left no bound behind'
    true_max = acc.abs().max().item()
    assert true_max <= fresh.item() <= small.item() + inc_max * (1 + 1e-5) and fresh.item() > 100 * small.item()
    # a fresh result carries the bound of what was stored
    y = _with_terms(4, lambda: fused_conv_act._launch_fused(dz, w, None, None, None, 1, 0.0, 1.0, -1.0, mode=1))
    assert amax.cached(y) is not None and amax.cached(y).item() == y.abs().max().item()
    # no bound on the target: none afterwards (the consumer's request runs the pass)
    acc2 = (torch.randn([2, 64, 16, 32], generator=g) * 1e-3).to(DEV)
    _with_terms(4, lambda: fused_conv_act._launch_fused(dz, w, None, None, None, 1, 0.0, 1.0, -1.0, mode=1, accumulate_into=acc2))
    assert amax.cached(acc2) is None and amax.bound(acc2).item() == acc2.abs().max().item()


def test_producers_leave_the_bound_of_their_output_behind():
    """sgv_amax_sink: the kernels that WRITE a convolution's input (LDS-tile FIR passes incl. fused modes 1 and 3, act_grad_scale, scale_channels, fromRGB)
    deliver max |output| as a by-product, so that no separate pass over the tensor is needed; an op that cannot serve the sink leaves it unserved."""
    from stylegan_v_amd.torch_utils.ops import bias_act, fused_fir_act, modulation, pointwise, upfirdn2d
    lib = custom_ops.get_native()
    g = torch.Generator().manual_seed(15)
    f = upfirdn2d.setup_filter([1, 3, 3, 1]).to(DEV)

    def bound_without_a_pass(t):
        cached = getattr(t, '_sgv_amax', None)
        assert cached is not None, 'the producer left no bound'
        return cached[2].item()

    def run():
        x = torch.randn([3, 8, 65, 65], generator=g).to(DEV)
        y = upfirdn2d.upfirdn2d(x, f, padding=1, gain=4)                                       # plain FIR, tile kernel
        assert bound_without_a_pass(y) == y.abs().max().item()
        y2 = upfirdn2d.upfirdn2d(x[:, :, :64, :64].contiguous(), f, padding=2)                 # 64 -> 65 columns: the odd output column
        assert bound_without_a_pass(y2) == y2.abs().max().item()
        sc, bi = (torch.rand([3, 8], generator=g) + 0.5).to(DEV), torch.randn([8], generator=g).to(DEV)
        y3 = fused_fir_act.fir_bias_act(x, f, scale=sc, bias=bi, padding=1, fir_gain=4, act='lrelu')   # fused mode 1
        assert bound_without_a_pass(y3) == y3.abs().max().item()
        y4 = modulation.scale_channels(x[:, :, :64, :64].contiguous(), sc)
        assert bound_without_a_pass(y4) == y4.abs().max().item()
        rgb, w, b = torch.randn([3, 3, 64, 64], generator=g).to(DEV), torch.randn([1, 32, 3], generator=g).to(DEV), torch.randn([32], generator=g).to(DEV)
        y5 = pointwise.pointwise_conv_bias_act(rgb, w, b, act='lrelu')                          # fromRGB as one kernel
        assert bound_without_a_pass(y5) == y5.abs().max().item()
        # the 2x up / down passes run on the lane-exchange kernel: its side output also covers the positions a lane computes past the right edge, so the bound
        # may sit above max |y| -- never above gain * sum|taps| * max |x|
        xmax = x.abs().max().item()
        up = upfirdn2d.upfirdn2d(x, f, up=2, padding=[2, 1, 2, 1], gain=4)
        assert up.abs().max().item() <= bound_without_a_pass(up) <= 4.001 * xmax
        xs = x[:, :, :64, :64].contiguous()
        dn = upfirdn2d.upfirdn2d(xs, f, down=2, padding=1)
        assert dn.abs().max().item() <= bound_without_a_pass(dn) <= 1.001 * xs.abs().max().item()
        # the gradient of "FIR, then bias + activation" (fused mode 2 on the lane-exchange kernel: what the layer in front's data- and weight-gradient
        # convolutions read) leaves its bound behind too
        xg = x.clone().requires_grad_(True)
        y6 = fused_fir_act.fir_bias_act(xg, f, scale=sc, bias=bi, padding=1, fir_gain=4, act='lrelu')
        (dx6,) = torch.autograd.grad(y6, xg, torch.randn(y6.shape, generator=g).to(DEV))
        assert dx6.abs().max().item() <= bound_without_a_pass(dx6) <= 1.5 * dx6.abs().max().item()
        # a FIR output whose kernel has no side output (the generic kernel: odd geometry) inherits gain * sum|taps| x its input's bound, if that is known
        odd = x[:, :, :33, :35].contiguous()
        amax.bound(odd)
        y7 = upfirdn2d.upfirdn2d(odd, f, up=3, padding=2, gain=9)
        assert y7.abs().max().item() <= amax.cached(y7).item() <= 9.001 * odd.abs().max().item()
    _with_terms(4, run)
    # with another arithmetic nothing is tracked
    x = torch.randn([1, 4, 33, 33], generator=g).to(DEV)
    y = _with_terms(3, lambda: upfirdn2d.upfirdn2d(x, f, padding=1))
    assert getattr(y, '_sgv_amax', None) is None
    # the C ABI contract: armed, served or not, disarmed either way
    out = torch.full([amax.SINK_SLOTS + 1], -1.0, device=DEV)
    t = torch.randn([4096], generator=g).to(DEV)
    lib.sgv_amax_sink(out.data_ptr())
    custom_ops.check(lib.sgv_absmax(t.data_ptr(), t.numel(), 0, torch.empty([1], device=DEV).data_ptr(), 0, custom_ops.raw_stream(t)), lib)    # not a producer
    assert lib.sgv_amax_sink_consumed() == 0 and bool((out == -1.0).all())
    y = torch.empty_like(t)
    s1 = torch.ones([1], device=DEV)
    custom_ops.check(lib.sgv_scale_channels(t.data_ptr(), s1.data_ptr(), y.data_ptr(), 1, 1, 4096, 0, custom_ops.raw_stream(t)), lib)          # the sink was disarmed by the call before
    assert lib.sgv_amax_sink_consumed() == 0 and bool((out == -1.0).all())
    out.zero_()
    lib.sgv_amax_sink(out.data_ptr())                                                                                                          # armed for THIS call
    custom_ops.check(lib.sgv_scale_channels(t.data_ptr(), s1.data_ptr(), y.data_ptr(), 1, 1, 4096, 0, custom_ops.raw_stream(t)), lib)
    assert lib.sgv_amax_sink_consumed() == 1 and out[0].item() == t.abs().max().item() and int((out[1:] > 0).sum()) >= 1


def test_scaled_operand_without_a_bound_of_the_scale_is_refused(monkeypatch):
    """terms = 4 with x_scale: the operand is x * x_scale, so a bound of x alone is not a bound of the operand (any |x_scale| > 2 would overflow the fp16
    split silently).  The C ABI refuses the call instead (the header used to advertise x_amax2 = NULL as allowed)."""
    if conv2d_gradfix.native_wrw_terms != 4 or not conv2d_gradfix.wrw_input_scale:
        pytest.skip('the default arithmetic is not the block-scaled fp16 split')
    g = torch.Generator().manual_seed(1)
    dy = torch.randn([2, 64, 32, 32], generator=g).to(DEV)
    x = torch.randn([2, 64, 32, 32], generator=g).to(DEV)
    s = (torch.rand([2, 64], generator=g) * 8).to(DEV)
    good = conv2d_gradfix._native_wrw(dy, x, S1, (64, 64, 3, 3), x_scale=s)
    ref = torch.nn.grad.conv2d_weight(x.double() * s.double()[:, :, None, None], (64, 64, 3, 3), dy.double(), padding=1)
    assert _rel(good, ref)[0] < 1e-6
    monkeypatch.setattr(conv2d_gradfix, '_wrw_bounds', lambda terms, dyc, xc, scale=None: (amax.bound(dyc).data_ptr(), amax.bound(xc).data_ptr(), None))
    with pytest.raises(RuntimeError, match='x_amax2'):
        conv2d_gradfix._native_wrw(dy, x, S1, (64, 64, 3, 3), x_scale=s)


def test_bounds_do_not_survive_a_graph_replay():
    """A replayed hipGraph rewrites tensors behind their version counters (the captured Adam step: the parameters).  `amax.graph_replayed()` -- called by
    TrainStep after every replay -- must make every bound taken before it stale."""
    w = torch.randn([64, 64, 3, 3], device=DEV)
    b0 = amax.bound(w)
    assert amax.cached(w) is b0
    w.data_ptr()                      # (no version bump: what a replay looks like to the host)
    with torch.no_grad():
        torch.cuda.current_stream().synchronize()
    amax.graph_replayed()
    assert amax.cached(w) is None, 'a bound taken before a replay must not be trusted after it'
    w.detach().mul_(4.0)              # the weights grew across a power of two
    amax.graph_replayed()
    b1 = amax.bound(w)
    assert b1.item() >= w.abs().max().item() and b1.item() > 2 * b0.item()
