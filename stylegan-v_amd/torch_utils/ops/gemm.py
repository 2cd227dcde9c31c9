"""Dense fp32 GEMMs on the gfx950 matrix cores (csrc/gemm.hip, ``sgv_gemm_f32``).

``linear(x, w, b)`` = ``x @ w.t() + b`` is the contraction of FullyConnectedLayer (reference:
src/training/layers.py:133-137); ``conv1x1(x, w, b)`` is a 1x1 convolution on contiguous NCHW seen as
a batched [Cout, Cin] x [Cin, H*W] product (reference: conv2d_resample.py:40-54 / F.conv2d).  Both use
v_mfma_f32_32x32x2_f32 (exact fp32).  Backward passes are the transposed GEMMs, through the same kernel.
Non-fp32 / CPU inputs use torch.
"""

import torch

from .. import custom_ops


def _launch(a, b, bias, c, m, n, k, lda, ldb, ldc, trans_b, batch=1, sa=0, sb=0, sc=0, bias_mode=0):
    lib = custom_ops.get_native()
    p = custom_ops.GemmParams()
    p.a, p.b, p.c = a.data_ptr(), b.data_ptr(), c.data_ptr()
    p.bias = bias.data_ptr() if bias is not None else None
    p.m, p.n, p.k, p.lda, p.ldb, p.ldc = m, n, k, lda, ldb, ldc
    p.trans_b, p.batch, p.stride_a, p.stride_b, p.stride_c, p.bias_mode = int(trans_b), batch, sa, sb, sc, bias_mode
    with torch.cuda.device_of(c):
        custom_ops.check(lib.sgv_gemm_f32(p, torch.cuda.current_stream(c.device).cuda_stream), lib)
    return c


def _native_ok(*tensors):
    return all(t is None or (t.is_cuda and t.dtype == torch.float32) for t in tensors)


def matmul_nt(a, b, bias=None):
    """a [M,K] @ b[N,K]^T (+ bias[N]) -> [M,N], no autograd."""
    a, b = a.contiguous(), b.contiguous()
    m, k = a.shape
    n = b.shape[0]
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
    @staticmethod
    def forward(ctx, x, w, b):
        xc = x.contiguous()
        w2 = w.reshape(w.shape[0], -1).contiguous()
        n, cin, h, wd = xc.shape
        cout = w2.shape[0]
        hw = h * wd
        y = torch.empty([n, cout, h, wd], dtype=torch.float32, device=x.device)
        _launch(w2, xc, b.contiguous() if b is not None else None, y, cout, hw, cin, cin, hw, hw, False,
                batch=n, sa=0, sb=cin * hw, sc=cout * hw, bias_mode=2 if b is not None else 0)
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
        if ctx.needs_input_grad[1]:
            n, cout = dy.shape[:2]
            dw = torch.einsum('nop,nip->oi', dy.reshape(n, cout, -1), x.reshape(n, x.shape[1], -1)).reshape(w.shape)
        if ctx.has_b and ctx.needs_input_grad[2]:
            db = dy.sum([0, 2, 3])
        return dx, dw, db


def conv1x1(x, w, b=None):
    """1x1 convolution: x [N,Cin,H,W] contiguous, w [Cout,Cin,1,1] (or [Cout,Cin]) -> [N,Cout,H,W]."""
    if _native_ok(x, w, b) and x.ndim == 4 and x.is_contiguous():
        return _Conv1x1Fn.apply(x, w, b)
    return torch.nn.functional.conv2d(x, w.reshape(w.shape[0], -1, 1, 1), b)
