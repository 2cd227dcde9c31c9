"""Dense layers of the hot path on the small-M matrix-core kernel (csrc/fc.hip, ``sgv_fc``).

``dense(x, weight, bias, ...)`` evaluates ``FullyConnectedLayer.forward`` of the reference (src/training/layers.py:108-138):

    y = act(normalize?(x) @ (weight * weight_gain).T + bias * bias_gain) * act_gain

as ONE kernel (the reference: scale the weight, scale the bias, addmm / matmul, bias_act -- four launches); ``normalize=True`` adds
the ``normalize_2nd_moment`` of the mapping network's input (layers.py:22-25, 85) as a prologue, so the z -> w chain is two launches.
Backward is two launches: the data gradient and the weight gradient both read dy and the saved output y and apply the activation
derivative on the fly; the bias gradient falls out of the weight-gradient launch.  A gradient that is itself differentiated
(``create_graph``: R1 through the discriminator's dense layers) switches to the plain-PyTorch composition, as do CPU tensors,
non-fp32 tensors and activations other than linear / lrelu.
"""

import math

import torch

from .. import custom_ops
from . import bias_act as _ba

enabled = True
large_m = 1024   # rows from which the tiled GEMM serves a dense layer (the 32 x 32-tile kernel of csrc/fc.hip is built for M = a few dozen)


def dense_ref(x, weight, bias=None, weight_gain=1.0, bias_gain=1.0, act='linear', normalize=False, act_gain=None):
    """The definition, in torch ops (layers.py:22-25, 126-137).  ``act_gain``: gain of the activation (default: the activation's own,
    sqrt(2) for lrelu; the trajectory convolutions of motion.py use a plain leaky relu, gain 1)."""
    if normalize:
        x = x * (x.square().mean(dim=1, keepdim=True) + 1e-8).rsqrt()
    w = weight.to(x.dtype) * weight_gain
    b = bias
    if b is not None:
        b = b.to(x.dtype)
        if bias_gain != 1:
            b = b * bias_gain
    if act == 'linear' and b is not None and act_gain in (None, 1):
        return torch.addmm(b.unsqueeze(0), x, w.t())
    return _ba.bias_act(x.matmul(w.t()), b, act=act, gain=act_gain)


def _launch(a, sam, sak, b, sbk, sbn, c, scm, scn, m, n, k, a_ref=None, bias=None, rowsum=None, normalize=False, act=1, alpha=0.0, gain=1.0, wgain=1.0,
            bgain=1.0, epilogue_act=False):
    lib = custom_ops.get_native()
    p = custom_ops.FcParams(a.data_ptr(), sam, sak, a_ref.data_ptr() if a_ref is not None else None, b.data_ptr(), sbk, sbn, c.data_ptr(), scm, scn,
                            bias.data_ptr() if bias is not None else None, rowsum.data_ptr() if rowsum is not None else None, m, n, k, int(normalize), act,
                            alpha, gain, wgain, bgain, int(epilogue_act), 1, 0, 0, 0, 0)
    with custom_ops.device_guard(c):
        custom_ops.check(lib.sgv_fc(p, custom_ops.raw_stream(c)), lib)


def _rows(x):
    """x as the kernel can address it without a copy: rows of contiguous floats at any 16-byte-multiple stride (`ws[:, i]` of the [N, num_ws, w_dim] style
    tensor: 26 affines per synthesis pass each made their own contiguous copy before), else a contiguous copy."""
    if x.dim() == 2 and x.stride(1) == 1 and x.stride(0) >= x.shape[1] and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0:
        return x
    return x.contiguous()


class _DenseFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, cfg):
        wgain, bgain, act, normalize, again = cfg
        spec = _ba.activation_funcs[act]
        xc, wc = _rows(x), weight.contiguous()
        bc = bias.contiguous().float() if bias is not None else None
        m, k = xc.shape
        n = wc.shape[0]
        y = torch.empty([m, n], dtype=torch.float32, device=x.device)
        _launch(xc, xc.stride(0), 1, wc, 1, k, y, n, 1, m, n, k, bias=bc, normalize=normalize, act=spec.cuda_idx, alpha=float(spec.def_alpha), gain=again,
                wgain=wgain, bgain=bgain, epilogue_act=True)
        ctx.cfg = cfg
        ctx.save_for_backward(x, weight, bias, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        wgain, bgain, act, normalize, again = ctx.cfg
        x, weight, bias, y = ctx.saved_tensors
        if torch.is_grad_enabled() or normalize:
            # create_graph (R1 / path length differentiate this gradient again), or the normalised mapping input (its Jacobian is not
            # worth a kernel: one [32, 512] tensor per iteration): differentiate the composition
            with torch.enable_grad():
                ins = [t for t, need in zip((x, weight, bias), ctx.needs_input_grad[:3]) if need and t is not None]
                xin = x if ctx.needs_input_grad[0] else x.detach()
                y2 = dense_ref(xin, weight, bias, wgain, bgain, act, normalize, again)
                grads = iter(torch.autograd.grad(y2, ins, dy, create_graph=torch.is_grad_enabled(), allow_unused=True))
            return tuple(next(grads) if (need and t is not None) else None for t, need in zip((x, weight, bias), ctx.needs_input_grad[:3])) + (None,)
        spec = _ba.activation_funcs[act]
        dyc, xc, wc = dy.contiguous(), _rows(x), weight.contiguous()
        m, k = xc.shape
        n = wc.shape[0]
        common = dict(a_ref=y, act=spec.cuda_idx, alpha=float(spec.def_alpha), gain=again, wgain=wgain)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty([m, k], dtype=torch.float32, device=dy.device)
            _launch(dyc, n, 1, wc, k, 1, dx, k, 1, m, k, n, **common)                       # dx[m,k] = sum_n dz[m,n] W[n,k]
        if ctx.needs_input_grad[1] or (bias is not None and ctx.needs_input_grad[2]):
            dw = torch.empty([n, k], dtype=torch.float32, device=dy.device)
            rowsum = torch.empty([n], dtype=torch.float32, device=dy.device) if bias is not None and ctx.needs_input_grad[2] else None
            _launch(dyc, 1, n, xc, xc.stride(0), 1, dw, k, 1, n, k, m, rowsum=rowsum, bgain=bgain, **common)   # dW[n,k] = sum_m dz[m,n] x[m,k]; db[n] = bg * sum_m dz[m,n]
            if rowsum is not None:
                db = rowsum.to(bias.dtype)
        return dx, dw, db, None


def dense(x, weight, bias=None, weight_gain=1.0, bias_gain=1.0, act='linear', normalize=False, act_gain=None):
    """x [M,K], weight [N,K], bias [N] or None -> [M,N]."""
    if enabled and x.is_cuda and x.ndim == 2 and x.dtype == torch.float32 and weight.dtype == torch.float32 and act in ('linear', 'lrelu') \
            and (bias is None or bias.dtype == torch.float32) and x.shape[0] <= 65535 * 32:
        again = float(_ba.activation_funcs[act].def_gain if act_gain is None else act_gain)
        if x.shape[0] >= large_m and not normalize and weight.shape[1] >= 256 and weight.shape[0] >= 128:
            # thousands of rows (the unfolded trajectories of the motion network, layers.py `EqLRConv1d.forward_nlc`: [32 * 76, 11 * 512] x [5632, 512]):
            # the 128 x 128-tile GEMM (csrc/gemm_kernel.h) + the fused bias / activation pass; both differentiable (twice)
            from . import gemm as _gemm
            b = bias * bias_gain if (bias is not None and bias_gain != 1) else bias
            return _ba.bias_act(_gemm.linear(x, weight * weight_gain), b, act=act, gain=again)
        return _DenseFn.apply(x, weight, bias, (float(weight_gain), float(bias_gain), act, bool(normalize), again))
    return dense_ref(x, weight, bias, weight_gain, bias_gain, act, normalize, act_gain)
