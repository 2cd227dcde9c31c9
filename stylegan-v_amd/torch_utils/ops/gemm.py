"""Dense fp32 GEMMs on the gfx950 matrix cores (csrc/gemm.hip, ``sgv_gemm_f32``).

``linear(x, w, b)`` = ``x @ w.t() + b`` is the contraction of FullyConnectedLayer (reference:
src/training/layers.py:133-137); ``conv1x1(x, w, b)`` is a 1x1 convolution on contiguous NCHW seen as
a batched [Cout, Cin] x [Cin, H*W] product (reference: conv2d_resample.py:40-54 / F.conv2d).  Both use
v_mfma_f32_32x32x2_f32 (exact fp32).  Backward passes are the transposed GEMMs, through the same kernel.
Non-fp32 / CPU inputs use torch.
"""

import torch

from .. import custom_ops

enabled = True  # conv2d_resample routes whole-tile fp32 1x1 convolutions here
exact_fp32 = False  # True: every product on the exact-fp32 matrix pipe (default: split-bf16 products where the shape allows, as the 3x3 family)


def _launch(a, b, bias, c, m, n, k, lda, ldb, ldc, trans_b, batch=1, sa=0, sb=0, sc=0, bias_mode=0, k_split=1, residual=None):
    lib = custom_ops.get_native()
    p = custom_ops.GemmParams()
    p.a, p.b, p.c = a.data_ptr(), b.data_ptr(), c.data_ptr()
    p.bias = bias.data_ptr() if bias is not None else None
    p.residual = residual.data_ptr() if residual is not None else None
    p.m, p.n, p.k, p.lda, p.ldb, p.ldc = m, n, k, lda, ldb, ldc
    p.trans_b, p.batch, p.stride_a, p.stride_b, p.stride_c, p.bias_mode, p.k_split = int(trans_b), batch, sa, sb, sc, bias_mode, k_split
    # the arithmetic switch of the 3x3 family covers the dense products too: 0 -> the exact fp32 matrix pipe, 3 -> bf16 split, 4 (default) -> the
    # block-scaled fp16 split where the tile shape allows it (fp32-grade; bounds of both operands travel as device pointers), exact fp32 elsewhere
    terms = _gradfix().native_conv_terms
    p.exact_fp32 = 1 if (exact_fp32 or terms == 0) else (2 if terms == 4 else 0)
    if p.exact_fp32 == 2 and not (n % 128 == 0 and (k // max(k_split, 1)) % 32 == 0 and lda % 4 == 0):
        p.exact_fp32 = 1          # a shape the split members do not serve (csrc/gemm.hip): the exact pipe, no bounds needed
    if p.exact_fp32 == 2:
        from . import amax as _amax
        p.a_amax, p.b_amax = _amax.bound(a).data_ptr(), _amax.bound(b).data_ptr()
    with torch.cuda.device_of(c):
        from . import amax as _amax
        stream = torch.cuda.current_stream(c.device).cuda_stream
        if k_split == 1:      # the store knows the result: it leaves the magnitude bound of c behind (the next layer's convolutions need it)
            custom_ops.check(_amax.launch_tracking(c, lambda: lib.sgv_gemm_f32(p, stream)), lib)
        else:
            custom_ops.check(lib.sgv_gemm_f32(p, stream), lib)
    return c


def _gradfix():
    from . import conv2d_gradfix
    return conv2d_gradfix


def _native_ok(*tensors):
    return all(t is None or (t.is_cuda and t.dtype == torch.float32) for t in tensors)


def matmul_nt(a, b, bias=None):
    """a [M,K] @ b[N,K]^T (+ bias[N]) -> [M,N], no autograd.  Few output tiles and a long K (the [2112, 5632] x [5632, 512] products of the motion
    network's trajectory convolutions: 68 tiles on 256 CUs): K is cut into slices that run as separate workgroups and are summed afterwards."""
    a, b = a.contiguous(), b.contiguous()
    m, k = a.shape
    n = b.shape[0]
    tiles = ((m + 127) // 128) * ((n + 127) // 128)
    ks = 1
    while tiles * ks < 192 and ks < 8 and k % (ks * 2 * 32) == 0 and k // (ks * 2) >= 256:
        ks *= 2
    if ks > 1:
        per = torch.empty([ks, m, n], dtype=torch.float32, device=a.device)
        _launch(a, b, None, per, m, n, k, k, k, n, True, batch=1, sc=m * n, k_split=ks)
        c = per.sum(0)
        return c + bias.unsqueeze(0) if bias is not None else c
    c = torch.empty([m, n], dtype=torch.float32, device=a.device)
    return _launch(a, b, bias, c, m, n, k, k, k, n, True, bias_mode=1 if bias is not None else 0)


def matmul_nn(a, b):
    """a [M,K] @ b[K,N] -> [M,N], no autograd."""
    a, b = a.contiguous(), b.contiguous()
    m, k = a.shape
    n = b.shape[1]
    c = torch.empty([m, n], dtype=torch.float32, device=a.device)
    return _launch(a, b, None, c, m, n, k, k, n, n, False)


class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        ctx.has_b = b is not None
        return matmul_nt(x, w, b.contiguous() if b is not None else None)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = linear(dy, w.t())          # dy [M,N] @ w [N,K]
        if ctx.needs_input_grad[1]:
            dw = linear(dy.t(), x.t())      # dy^T [N,M] @ x [M,K]
        if ctx.has_b and ctx.needs_input_grad[2]:
            db = dy.sum(0)
        return dx, dw, db


def linear(x, w, b=None):
    """x [M,K], w [N,K] -> x @ w.t() + b."""
    if _native_ok(x, w, b) and x.ndim == 2 and w.ndim == 2:
        return _LinearFn.apply(x, w, b)
    y = x.matmul(w.t())
    return y + b.unsqueeze(0) if b is not None else y


class _Conv1x1Fn(torch.autograd.Function):
    """y = conv1x1(x, w) (+ b) (+ residual): `residual` [N,Cout,H,W] is added in the kernel's store (the other branch of a residual block)."""

    @staticmethod
    def forward(ctx, x, w, b, residual):
        xc = x.contiguous()
        w2 = w.reshape(w.shape[0], -1).contiguous()
        n, cin, h, wd = xc.shape
        cout = w2.shape[0]
        hw = h * wd
        y = torch.empty([n, cout, h, wd], dtype=torch.float32, device=x.device)
        _launch(w2, xc, b.contiguous() if b is not None else None, y, cout, hw, cin, cin, hw, hw, False,
                batch=n, sa=0, sb=cin * hw, sc=cout * hw, bias_mode=2 if b is not None else 0, residual=residual.contiguous() if residual is not None else None)
        ctx.save_for_backward(x, w)
        ctx.has_b = b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        w2 = w.reshape(w.shape[0], -1)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = conv1x1(dy, w2.t().reshape(w2.shape[1], w2.shape[0], 1, 1))
        if ctx.needs_input_grad[1] and not _gradfix().weight_gradients_disabled:  # same switch as conv2d_gradfix.py:23
            dw = conv1x1_weight_grad(dy, x).reshape(w.shape)
        if ctx.has_b and ctx.needs_input_grad[2]:
            db = dy.sum([0, 2, 3])
        return dx, dw, db, (dy if ctx.needs_input_grad[3] else None)


class _Conv1x1WeightGradFn(torch.autograd.Function):
    """dw[o,i] = sum_{n,p} dy[n,o,p] x[n,i,p]: one [Cout,HW] x [Cin,HW]^T product per sample (K = H*W, both operands k-contiguous
    in NCHW), summed over the batch.  Differentiable (R1 differentiates the discriminator's skip convolutions twice)."""

    @staticmethod
    def forward(ctx, dy, x):
        dyc, xc = dy.contiguous(), x.contiguous()
        n, cout, h, wd = dyc.shape
        cin, hw = xc.shape[1], h * wd
        # few output tiles, long K: cut K so that the launch has >= ~512 workgroups (a [128 x 64] gradient over 96 samples is 96 workgroups
        # of 16384 k-steps each otherwise: 1.5 ms where the operands stream in 0.3)
        tiles = ((cout + 127) // 128) * ((cin + 127) // 128)
        ks = 1
        while n * tiles * ks < 512 and ks < 16 and hw % (ks * 2 * 32) == 0 and n * ks * 2 <= 65535:
            ks *= 2
        if cin % 128 != 0 and cout % 128 == 0:
            # the kernel's branch-free path wants whole 128-column tiles (rows are free): compute the transpose, rows = cin
            per = torch.empty([n * ks, cin, cout], dtype=torch.float32, device=x.device)
            _launch(xc, dyc, None, per, cin, cout, hw, hw, hw, cout, True, batch=n, sa=cin * hw, sb=cout * hw, sc=cout * cin, k_split=ks)
            ctx.save_for_backward(dy, x)
            return per.sum(0).t()
        per = torch.empty([n * ks, cout, cin], dtype=torch.float32, device=x.device)
        _launch(dyc, xc, None, per, cout, cin, hw, hw, hw, cin, True, batch=n, sa=cout * hw, sb=cin * hw, sc=cout * cin, k_split=ks)
        ctx.save_for_backward(dy, x)
        return per.sum(0)

    @staticmethod
    def backward(ctx, g):  # g [Cout, Cin]
        dy, x = ctx.saved_tensors
        d_dy = d_x = None
        if ctx.needs_input_grad[0]:
            d_dy = conv1x1(x, g)
        if ctx.needs_input_grad[1]:
            d_x = conv1x1(dy, g.t())
        return d_dy, d_x


def conv1x1_weight_grad(dy, x):
    """dy [N,Cout,H,W], x [N,Cin,H,W] -> [Cout,Cin]."""
    if _native_ok(dy, x) and dy.ndim == 4:
        return _Conv1x1WeightGradFn.apply(dy, x)
    n = dy.shape[0]
    return torch.einsum('nop,nip->oi', dy.reshape(n, dy.shape[1], -1), x.reshape(n, x.shape[1], -1))


def is_full_tile_conv1x1(x, cout):
    """Shapes on which the MFMA kernel runs its branch-free path (and beats MIOpen's 1x1: 110-119 vs 76 TFLOP/s)."""
    return (x.ndim == 4 and x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and cout % 128 == 0 and (x.shape[2] * x.shape[3]) % 128 == 0
            and x.shape[1] % 16 == 0)


def conv1x1(x, w, b=None, residual=None):
    """1x1 convolution: x [N,Cin,H,W] contiguous, w [Cout,Cin,1,1] (or [Cout,Cin]) -> [N,Cout,H,W]; `residual` (same shape as the result,
    fp32) is added to it -- inside the kernel on the native path."""
    if _native_ok(x, w, b) and x.ndim == 4 and x.is_contiguous() and (residual is None or (residual.is_cuda and residual.dtype == torch.float32)):
        return _Conv1x1Fn.apply(x, w, b, residual)
    y = torch.nn.functional.conv2d(x, w.reshape(w.shape[0], -1, 1, 1), b)
    return y + residual if residual is not None else y
