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


# ---------------------------------------------------------------------------------------------------------------------------------------
# The style affines of a synthesis pass as ONE launch forward and TWO backward (``sgv_fc_grouped``).  Every SynthesisLayer / ToRGBLayer owns
# ``affine = FullyConnectedLayer(w_dim, in_channels, bias_init=1)`` (networks.py:116,153) and calls it on its own column of ws: 21 dense launches per
# pass at FFS-256 (27 at 1024^2), 10-14 us each whatever the batch -- 1.6 ms of a 150-ms training iteration, 3.5 % of the 8-videos-per-GPU step.

grouped = True


def _fc_problem(a, sam, sak, b, sbk, sbn, c, scm, m, n, k, a_ref=None, bias=None, rowsum=None, act=1, alpha=0.0, gain=1.0, wgain=1.0, bgain=1.0, epilogue_act=False):
    return custom_ops.FcParams(a, sam, sak, a_ref, b, sbk, sbn, c, scm, 1, bias, rowsum, m, n, k, 0, act, alpha, gain, wgain, bgain, int(epilogue_act), 1, 0, 0, 0, 0)


def _launch_group(problems, device_tensor):
    lib = custom_ops.get_native()
    arr = (custom_ops.FcParams * len(problems))(*problems)
    with custom_ops.device_guard(device_tensor):
        custom_ops.check(lib.sgv_fc_grouped(arr, len(problems), custom_ops.raw_stream(device_tensor)), lib)


class _GroupedAffineFn(torch.autograd.Function):
    """styles_l = (ws[:, idx_l] @ (W_l * wg_l)^T + b_l * bg) * gain_l for every layer l, one launch; inputs: ws, W_0 .. W_{L-1}, b_0 .. b_{L-1}."""

    @staticmethod
    def forward(ctx, ws, cfg, *wb):
        idx, wgains, bgain, gains = cfg
        L = len(idx)
        weights, biases = wb[:L], wb[L:]
        wsc = ws if (ws.is_contiguous() and ws.data_ptr() % 16 == 0) else ws.contiguous()
        m, nws, k = wsc.shape
        outs = [torch.empty([m, w.shape[0]], dtype=torch.float32, device=ws.device) for w in weights]
        wcs = [w.contiguous() for w in weights]
        bcs = [b.contiguous() for b in biases]
        _launch_group([_fc_problem(wsc.data_ptr() + 4 * k * i, nws * k, 1, w.data_ptr(), 1, k, y.data_ptr(), w.shape[0], m, w.shape[0], k, bias=b.data_ptr(),
                                   gain=g, wgain=wg, bgain=bgain, epilogue_act=True) for i, w, b, y, wg, g in zip(idx, wcs, bcs, outs, wgains, gains)], wsc)
        ctx.cfg = cfg
        ctx.save_for_backward(ws, *wb)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *dys):
        idx, wgains, bgain, gains = ctx.cfg
        L = len(idx)
        ws = ctx.saved_tensors[0]
        weights, biases = ctx.saved_tensors[1:1 + L], ctx.saved_tensors[1 + L:]
        if torch.is_grad_enabled():
            # create_graph (path-length regularisation differentiates the styles twice): the composition, layer by layer
            with torch.enable_grad():
                ins = [t for t, need in zip((ws,) + tuple(weights) + tuple(biases), [ctx.needs_input_grad[0]] + list(ctx.needs_input_grad[2:])) if need]
                ys = [dense_ref(ws[:, i], w, b, wg, bgain, 'linear', False, g) for i, w, b, wg, g in zip(idx, weights, biases, wgains, gains)]
                live = [(y, dy) for y, dy in zip(ys, dys) if dy is not None]
                grads = iter(torch.autograd.grad([y for y, _ in live], ins, [dy for _, dy in live], create_graph=True, allow_unused=True))
            return (next(grads) if ctx.needs_input_grad[0] else None, None) + tuple(next(grads) if need else None for need in ctx.needs_input_grad[2:])
        m, nws, k = ws.shape
        wsc = ws if (ws.is_contiguous() and ws.data_ptr() % 16 == 0) else ws.contiguous()
        dyc = [dy.contiguous() if dy is not None else torch.zeros([m, w.shape[0]], dtype=torch.float32, device=ws.device) for dy, w in zip(dys, weights)]
        wcs = [w.contiguous() for w in weights]
        d_ws = None
        if ctx.needs_input_grad[0]:
            # Every layer writes its data gradient straight into its column of d(ws) (row stride nws * k).  A column of ws may feed two layers (each ToRGB shares its
            # w with the next block's conv0, networks.py:354-357): its second user ACCUMULATES, in a second launch behind the first.  (No index tensors, no host data:
            # the pass is replayed inside hipGraphs.)
            rounds, uses = [], {}
            for j, i in enumerate(idx):       # the k-th user of a column goes into launch k: no two problems of one launch touch the same column
                kth = uses.get(i, 0)
                uses[i] = kth + 1
                while len(rounds) <= kth:
                    rounds.append([])
                rounds[kth].append(j)
            d_ws = (torch.empty if len(uses) == nws else torch.zeros)([m, nws, k], dtype=torch.float32, device=ws.device)
            for kth, group in enumerate(rounds):
                acc = 1 if kth else 0
                if group:
                    probs = [_fc_problem(dyc[j].data_ptr(), wcs[j].shape[0], 1, wcs[j].data_ptr(), k, 1, d_ws.data_ptr() + 4 * k * idx[j], nws * k, m, k, wcs[j].shape[0],
                                         a_ref=dyc[j].data_ptr(), gain=gains[j], wgain=wgains[j]) for j in group]      # (a_ref = dy: a linear activation only applies the output gain there)
                    for pr in probs:
                        pr.accumulate = acc
                    _launch_group(probs, wsc)
        d_w = [None] * L
        d_b = [None] * L
        need_w = list(ctx.needs_input_grad[2:2 + L])
        need_b = list(ctx.needs_input_grad[2 + L:])
        if any(need_w) or any(need_b):
            d_w = [torch.empty_like(w) for w in wcs]
            d_b = [torch.empty([w.shape[0]], dtype=torch.float32, device=ws.device) for w in wcs]
            _launch_group([_fc_problem(dy.data_ptr(), 1, w.shape[0], wsc.data_ptr() + 4 * k * i, nws * k, 1, dw.data_ptr(), k, w.shape[0], k, m, a_ref=dy.data_ptr(),
                                       rowsum=db.data_ptr(), gain=g, wgain=wg, bgain=bgain) for i, dy, w, dw, db, wg, g in zip(idx, dyc, wcs, d_w, d_b, wgains, gains)], wsc)
        return (d_ws, None) + tuple(d_w) + tuple(d_b)


def grouped_affine(ws, idx, layers, gains=None):
    """[layer.forward(ws[:, i], gain=g) for layer, i, g in zip(layers, idx, gains)] for FullyConnectedLayer modules with linear activation and a bias, as one launch.
    ws [M, num_ws, K] fp32; returns a tuple of [M, out_features_l] tensors.  Falls back to the per-layer calls where the kernel does not apply."""
    gains = [1.0 if g is None else float(g) for g in (gains or [None] * len(layers))]
    ok = (grouped and enabled and ws.is_cuda and ws.ndim == 3 and ws.dtype == torch.float32 and len(layers) > 1 and ws.shape[0] <= 65535 * 32
          and all(l.activation == 'linear' and l.bias is not None and l.weight.dtype == torch.float32 and l.bias.dtype == torch.float32 and l.weight.shape[1] == ws.shape[2]
                  and l.bias_gain == layers[0].bias_gain for l in layers))
    if not ok:
        return tuple(l(ws[:, i], gain=(None if g == 1.0 else g)) for l, i, g in zip(layers, idx, gains))
    cfg = (tuple(int(i) for i in idx), tuple(float(l.weight_gain) for l in layers), float(layers[0].bias_gain), tuple(gains))
    return _GroupedAffineFn.apply(ws, cfg, *[l.weight for l in layers], *[l.bias for l in layers])
