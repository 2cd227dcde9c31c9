"""Fused element-wise tail of the continuous temporal positional encoder.

Reference: ``AlignedTimeEncoder.forward`` src/training/motion.py:201-212 (three raw phases, six
sin/cos, two lerps, a sum -- ~25 tiny PyTorch kernels).  GPU fp32 inputs run the single kernel
``sgv_time_encode`` (csrc/time_encode.hip); otherwise the same expression is evaluated with PyTorch
ops.  Backward is written with differentiable tensor ops (any order).
"""

import torch

from .. import custom_ops


def time_encode_ref(periods, phases, al, ar, freqs, phase_scales, t, t_left, t_right, alpha):
    """periods/phases [R, nf] (periods already tanh()+1), al/ar [R, 2nf], freqs/phase_scales [nf] or [1, nf],
    t/t_left/t_right/alpha [R].  Returns [R, 2nf]."""
    fr = freqs.reshape(1, -1)
    ps = phase_scales.reshape(1, -1)
    a = alpha.reshape(-1, 1)

    def pos(tau):
        raw = fr * periods * tau.reshape(-1, 1) + phases * ps
        return torch.cat([raw.sin(), raw.cos()], dim=1)

    remove = pos(t_left) * (1 - a) + pos(t_right) * a
    add = al * (1 - a) + ar * a
    return pos(t) - remove + add


class _TimeEncodeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, periods, phases, al, ar, freqs, phase_scales, t, t_left, t_right, alpha):
        lib = custom_ops.get_native()
        args = [v.contiguous() for v in (periods, phases, al, ar, freqs.reshape(-1), phase_scales.reshape(-1), t, t_left, t_right, alpha)]
        rows, nf = args[0].shape
        out = torch.empty([rows, 2 * nf], dtype=torch.float32, device=periods.device)
        p = custom_ops.TimeEncodeParams()
        (p.periods, p.phases, p.al, p.ar, p.freqs, p.phase_scales, p.t, p.t_left, p.t_right, p.alpha) = [v.data_ptr() for v in args]
        p.out, p.rows, p.nf = out.data_ptr(), rows, nf
        with torch.cuda.device_of(out):
            custom_ops.check(lib.sgv_time_encode(p, torch.cuda.current_stream(out.device).cuda_stream), lib)
        ctx.save_for_backward(periods, phases, freqs, phase_scales, t, t_left, t_right, alpha)
        return out

    @staticmethod
    def backward(ctx, g):
        periods, phases, freqs, phase_scales, t, t_left, t_right, alpha = ctx.saved_tensors
        nf = periods.shape[1]
        fr, ps, a = freqs.reshape(1, -1), phase_scales.reshape(1, -1), alpha.reshape(-1, 1)
        g_sin, g_cos = g[:, :nf], g[:, nf:]
        # d out / d raw(tau) = w(tau) * (g_sin*cos(raw) - g_cos*sin(raw)),  w = +1 (t), -(1-a) (t_left), -a (t_right)
        d_raw_sum = 0
        d_periods = 0
        for tau, wgt in ((t, 1.0), (t_left, -(1 - a)), (t_right, -a)):
            tau = tau.reshape(-1, 1)
            raw = fr * periods * tau + phases * ps
            d_raw = wgt * (g_sin * raw.cos() - g_cos * raw.sin())
            d_raw_sum = d_raw_sum + d_raw
            d_periods = d_periods + d_raw * fr * tau
        grads = [None] * 10
        if ctx.needs_input_grad[0]:
            grads[0] = d_periods
        if ctx.needs_input_grad[1]:
            grads[1] = d_raw_sum * ps
        if ctx.needs_input_grad[2]:
            grads[2] = g * (1 - a)
        if ctx.needs_input_grad[3]:
            grads[3] = g * a
        return tuple(grads)


def time_encode(periods, phases, al, ar, freqs, phase_scales, t, t_left, t_right, alpha):
    tensors = (periods, phases, al, ar, freqs, phase_scales, t, t_left, t_right, alpha)
    if periods.is_cuda and all(v.dtype == torch.float32 for v in tensors) and not any(v.requires_grad for v in (freqs, phase_scales, t, t_left, t_right, alpha)):
        return _TimeEncodeFn.apply(*tensors)
    return time_encode_ref(*tensors)
