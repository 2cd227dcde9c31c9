"""Continuous temporal positional encoder of the generator.

Mirrors ``src/training/motion.py`` of the reference: ``MotionMappingNetwork`` (:18-156) turns a noise
trajectory into per-frame motion codes, ``AlignedTimeEncoder`` (:160-214) turns a frame's time stamp and
its two neighbouring trajectory codes into the 2*dim sin/cos embedding
``pos(t) - lerp(pos(t_l), pos(t_r), a) + lerp(A_l, A_r, a)``.  The element-wise tail (three phases, six
sin/cos, two lerps) is one fused kernel (``csrc/time_encode.hip``); on the GPU the two trajectory convolutions (k = 11, valid) run
as dense-layer kernels on the unfolded [B, L, C] trajectory and the four prediction heads as two more (``csrc/fc.hip``: exact-fp32
MFMA, weight / bias gains, bias and leaky relu in the epilogue) -- no vendor convolution or GEMM is left in the encoder.  Only the configuration StyleGAN-V trains with is
built: ``gen_strategy='conv'`` and ``fourier=True`` (configs/model/stylegan-v.yaml:19-27).
"""

import contextlib
import threading
import math

import numpy as np
import torch

from ..torch_utils import misc
from ..torch_utils.ops import fc as _fc
from ..torch_utils.ops import time_encode as _te
from .layers import EqLRConv1d, FullyConnectedLayer


def construct_linspaced_frequencies(num_freqs, min_period_len, max_period_len):
    """2*pi / period for `num_freqs` periods spaced linearly in log2 from max down to min (ascending frequency)."""
    periods = 2.0 ** np.linspace(np.log2(min_period_len), np.log2(max_period_len), num_freqs)
    freqs = (2 * np.pi / periods)[::-1].copy().astype(np.float32)
    return torch.from_numpy(freqs).unsqueeze(0)


class AlignedTimeEncoder(torch.nn.Module):
    def __init__(self, latent_dim=512, cfg=None):
        super().__init__()
        self.cfg = cfg
        self.latent_dim = latent_dim
        freqs = construct_linspaced_frequencies(cfg.time_enc.dim, cfg.time_enc.min_period_len, cfg.time_enc.max_period_len)
        self.register_buffer('freqs', freqs)  # [1, nf]
        nf = freqs.shape[1]
        # bias-free predictors (a bias would let all videos share one motion mode)
        self.periods_predictor = FullyConnectedLayer(latent_dim, nf, activation='linear', bias=False)
        self.phase_predictor = FullyConnectedLayer(latent_dim, nf, activation='linear', bias=False)
        self.register_buffer('phase_scales', cfg.time_enc.max_period_len / (2 * np.pi / freqs))  # [1, nf], in [1, max/min]
        self.aligners_predictor = FullyConnectedLayer(latent_dim, nf * 2, activation='linear', bias=False)

    def get_dim(self):
        return self.freqs.shape[1] * 2

    def forward(self, t, motion_u_left, motion_u_right, interp_weights, t_left, t_right):
        b, f, udim = motion_u_left.shape
        misc.assert_shape(t, [b, f])
        misc.assert_shape(motion_u_right, [b, f, None])
        misc.assert_shape(interp_weights, [b, f, 1])
        assert t.shape == t_left.shape == t_right.shape
        ul = motion_u_left.reshape(b * f, udim)
        ur = motion_u_right.reshape(b * f, udim)
        # one GEMM for the three heads that read the left code (periods | phases | aligners), one for the right aligner
        heads = torch.cat([self.periods_predictor.weight, self.phase_predictor.weight, self.aligners_predictor.weight], dim=0)
        gain = self.periods_predictor.weight_gain  # identical for the three heads: lr_mul 1, same fan-in
        left = _fc.dense(ul, heads.to(ul.dtype), None, weight_gain=gain)   # one small-M MFMA kernel on the GPU (csrc/fc.hip)
        nf = self.freqs.shape[1]
        periods = left[:, :nf].tanh() + 1
        phases = left[:, nf:2 * nf]
        aligners_left = left[:, 2 * nf:]
        aligners_right = self.aligners_predictor(ur)
        return _te.time_encode(periods, phases, aligners_left, aligners_right, self.freqs, self.phase_scales,
                               t.reshape(-1).float(), t_left.reshape(-1).float(), t_right.reshape(-1).float(),
                               interp_weights.reshape(-1).float())


# innermost `frame_times_bounded_by` bound of the CALLING THREAD: a stack in thread-local storage, NOT module state -- it neither survives pickling nor
# reaches G_ema, and an evaluation / generation thread that uses a generator while a training phase holds a bound does not inherit it
# (it keeps the reference's t.max() sizing and is not clamped)
_tls = threading.local()


def _bound_stack():
    stack = getattr(_tls, 't_bound', None)
    if stack is None:
        stack = _tls.t_bound = []
    return stack


@contextlib.contextmanager
def frame_times_bounded_by(bound):
    """Promise that every frame time `t` handed to a MotionMappingNetwork inside the block is <= `bound` (a training loop that draws
    t < max_num_frames by construction).  `get_max_traj_len` then skips the reference's `t.max().item()` device->host read
    (src/training/motion.py:97-100); the trajectory gather clamps its index so that a broken promise cannot read out of bounds."""
    stack = _bound_stack()
    stack.append(float(bound))
    try:
        yield
    finally:
        stack.pop()


class MotionMappingNetwork(torch.nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        assert cfg.motion.gen_strategy == 'conv', 'only the conv trajectory generator (StyleGAN-V default) is built'
        assert cfg.motion.fourier, 'only the Fourier (AlignedTimeEncoder) head is built'
        self.time_encoder = AlignedTimeEncoder(cfg=cfg, latent_dim=cfg.motion.v_dim)
        k = cfg.motion.kernel_size
        self.conv = torch.nn.Sequential(
            EqLRConv1d(cfg.motion.z_dim + cfg.c_dim, cfg.motion.z_dim, k, padding=0, activation='lrelu', lr_multiplier=0.01),
            EqLRConv1d(cfg.motion.z_dim, cfg.motion.v_dim, k, padding=0, activation='lrelu', lr_multiplier=0.01),
        )
        self.num_additional_codes = (k - 1) * 2  # the two valid convolutions eat (k-1) codes each

    def get_max_traj_len(self, t):
        _t_bound = _bound_stack()
        if _t_bound:     # inside `frame_times_bounded_by`: no device->host read of t.max() (a pipeline stall per pass, illegal under hipGraph capture)
            max_t = max(self.cfg.sampling.max_num_frames - 1, _t_bound[-1])
        else:
            max_t = max(self.cfg.sampling.max_num_frames - 1, float(t.max().item()))
        return int(math.ceil(max_t / self.cfg.motion.motion_z_distance)) + 2

    def get_dim(self):
        return self.time_encoder.get_dim()

    def generate_motion_u_codes(self, c, t, motion_z=None):
        """Noise trajectory -> conv1d x2 -> for every frame the two trajectory codes that bracket its time stamp.

        c [B, c_dim], t [B, F] (fractional frame indices), motion_z optional [B, L, z_dim] noise to reuse."""
        b, f = t.shape
        dist = self.cfg.motion.motion_z_distance
        traj_len = self.get_max_traj_len(t) + self.num_additional_codes
        if motion_z is None:
            motion_z = torch.randn(b, traj_len, self.cfg.motion.z_dim, device=c.device)
        x = motion_z[:b, :traj_len, :self.cfg.motion.z_dim].to(c.device)
        if self.cfg.c_dim > 0:
            misc.assert_shape(c, [b, None])
            x = torch.cat([x, c.unsqueeze(1).expand(-1, traj_len, -1)], dim=2)
        if x.is_cuda and x.dtype == torch.float32 and _fc.enabled:
            trajs = x                          # [B, L, C] throughout: each valid conv1d + bias + lrelu is one dense-layer kernel on the unfolded trajectory
            for layer in self.conv:
                trajs = layer.forward_nlc(trajs)
        else:
            trajs = self.conv(x.permute(0, 2, 1)).permute(0, 2, 1)  # [B, L - 2(k-1), v_dim]

        left_idx = (t / dist).floor().long()
        if _bound_stack():   # nothing compared t with the promised bound on the host: keep both gathers inside the trajectory whatever t holds
            left_idx = left_idx.clamp(0, trajs.shape[1] - 2)
        rows = torch.arange(b, device=c.device).unsqueeze(1).expand(-1, f)
        u_left = trajs[rows, left_idx]
        u_right = trajs[rows, left_idx + 1]
        t_left = t - t % dist
        alpha = ((t % dist) / dist).unsqueeze(2).to(torch.float32)
        u = (u_left * (1 - alpha) + u_right * alpha).reshape(b * f, -1).to(torch.float32)
        return dict(motion_u_left=u_left, motion_u_right=u_right, t_left=t_left, t_right=t_left + dist, interp_weights=alpha,
                    motion_u=u, motion_z=motion_z)

    def forward(self, c, t, motion_z=None):
        assert len(c) == len(t) and t.ndim == 2
        info = self.generate_motion_u_codes(c, t, motion_z=motion_z)
        motion_v = self.time_encoder(t=t, motion_u_left=info['motion_u_left'], motion_u_right=info['motion_u_right'],
                                     t_left=info['t_left'], t_right=info['t_right'], interp_weights=info['interp_weights'])
        return dict(motion_v=motion_v, motion_z=info['motion_z'])
