"""Building blocks shared by the generator and the discriminator.

Module/parameter names mirror the reference's ``src/training/layers.py`` so that checkpoints map
one-to-one: ``MappingNetwork`` (:22), ``FullyConnectedLayer`` (:108), ``Conv2dLayer`` (:142),
``GenInput`` (:201), ``TemporalInput`` (:230), ``TemporalDifferenceEncoder`` (:255),
``FixedTimeEncoder`` (:301), ``EqLRConv1d`` (:331), ``sample_frames`` (:377),
``construct_log_spaced_freqs`` (:439).  Numerics follow SURVEY.md appendix E; all bias/activation
work goes through the fused ``bias_act`` kernel and every resampling step through ``upfirdn2d``.
"""

import math
import random

import numpy as np
import torch
import torch.nn.functional as F

from ..torch_utils import misc
from ..torch_utils.ops import bias_act, conv2d_gradfix, conv2d_resample, eqlr, fc, fused_conv_act, fused_down_act, pointwise, upfirdn2d


@misc.profiled_function
def normalize_2nd_moment(x, dim=1, eps=1e-8):
    return x * (x.square().mean(dim=dim, keepdim=True) + eps).rsqrt()


class FullyConnectedLayer(torch.nn.Module):
    """Equalised-learning-rate dense layer: weight stored as randn/lr_mul, used as W*lr_mul/sqrt(in)."""

    def __init__(self, in_features, out_features, bias=True, activation='linear', lr_multiplier=1, bias_init=0):
        super().__init__()
        self.activation = activation
        self.weight = torch.nn.Parameter(torch.randn([out_features, in_features]) / lr_multiplier)
        self.bias = torch.nn.Parameter(torch.full([out_features], float(bias_init))) if bias else None
        self.weight_gain = lr_multiplier / math.sqrt(in_features)
        self.bias_gain = lr_multiplier

    def forward(self, x, normalize_input=False, gain=None):
        """``normalize_input``: apply normalize_2nd_moment to x first (the mapping network's first layer; one kernel with the layer).
        ``gain``: the layer's output times a constant (ToRGB's ``affine(w) * weight_gain``, networks.py:314) -- rides on the kernel's activation gain instead of
        costing a launch here and another in the backward pass."""
        if x.ndim == 2:   # one kernel on the GPU (ops/fc.py -> csrc/fc.hip); the torch composition elsewhere
            act_gain = None if gain is None else float(bias_act.activation_funcs[self.activation].def_gain * gain)
            return fc.dense(x, self.weight, self.bias, weight_gain=self.weight_gain, bias_gain=self.bias_gain, act=self.activation, normalize=normalize_input, act_gain=act_gain)
        if gain is not None:
            return self.forward(x, normalize_input=normalize_input) * gain
        if normalize_input:
            x = normalize_2nd_moment(x)
        w = self.weight.to(x.dtype) * self.weight_gain
        b = self.bias
        if b is not None:
            b = b.to(x.dtype)
            if self.bias_gain != 1:
                b = b * self.bias_gain
        if self.activation == 'linear' and b is not None:
            return torch.addmm(b.unsqueeze(0), x, w.t())
        return bias_act.bias_act(x.matmul(w.t()), b, act=self.activation)

    def extra_repr(self):
        return f'in={self.weight.shape[1]}, out={self.weight.shape[0]}, act={self.activation}'


class MappingNetwork(torch.nn.Module):
    """z (and/or an embedded label c) -> w through ``num_layers`` dense+lrelu layers; tracks the running
    mean of w for truncation; optionally broadcasts to ``num_ws`` copies."""

    def __init__(self, z_dim, c_dim, w_dim, num_ws, num_layers=8, embed_features=None, layer_features=None,
                 activation='lrelu', lr_multiplier=0.01, w_avg_beta=0.995, cfg=None):
        super().__init__()
        self.cfg = cfg if cfg is not None else {}
        self.z_dim, self.c_dim, self.w_dim, self.num_ws = z_dim, c_dim, w_dim, num_ws
        self.num_layers, self.w_avg_beta = num_layers, w_avg_beta
        embed_features = 0 if c_dim == 0 else (w_dim if embed_features is None else embed_features)
        layer_features = w_dim if layer_features is None else layer_features
        widths = [z_dim + embed_features] + [layer_features] * (num_layers - 1) + [w_dim]
        if c_dim > 0:
            self.embed = FullyConnectedLayer(c_dim, embed_features)
        for idx, (fin, fout) in enumerate(zip(widths[:-1], widths[1:])):
            setattr(self, f'fc{idx}', FullyConnectedLayer(fin, fout, activation=activation, lr_multiplier=lr_multiplier))
        if num_ws is not None and w_avg_beta is not None:
            self.register_buffer('w_avg', torch.zeros([w_dim]))

    def forward(self, z, c, truncation_psi=1, truncation_cutoff=None, skip_w_avg_update=False):
        parts = []
        fold_norm = self.z_dim > 0 and self.c_dim == 0   # unconditional model: the input normalisation rides in fc0's kernel
        if self.z_dim > 0:
            misc.assert_shape(z, [None, self.z_dim])
            parts.append(z.to(torch.float32) if fold_norm else normalize_2nd_moment(z.to(torch.float32)))
        if self.c_dim > 0:
            misc.assert_shape(c, [None, self.c_dim])
            parts.append(normalize_2nd_moment(self.embed(c.to(torch.float32))))
        x = parts[0] if len(parts) == 1 else torch.cat(parts, dim=1)
        for idx in range(self.num_layers):
            x = getattr(self, f'fc{idx}')(x, normalize_input=(fold_norm and idx == 0))
        if self.w_avg_beta is not None and self.training and not skip_w_avg_update:
            self.w_avg.copy_(x.detach().mean(dim=0).lerp(self.w_avg, self.w_avg_beta))
        if self.num_ws is not None:
            x = x.unsqueeze(1).repeat([1, self.num_ws, 1])
        if truncation_psi != 1:
            assert self.w_avg_beta is not None
            if self.num_ws is None or truncation_cutoff is None:
                x = self.w_avg.lerp(x, truncation_psi)
            else:
                x[:, :truncation_cutoff] = self.w_avg.lerp(x[:, :truncation_cutoff], truncation_psi)
        return x


class Conv2dLayer(torch.nn.Module):
    """Non-modulated convolution (+ optional FIR resampling) followed by the fused bias/activation kernel."""

    def __init__(self, in_channels, out_channels, kernel_size, bias=True, activation='linear', up=1, down=1,
                 resample_filter=(1, 3, 3, 1), conv_clamp=None, channels_last=False, trainable=True, instance_norm=False,
                 lr_multiplier=1.0):
        super().__init__()
        self.activation, self.up, self.down, self.conv_clamp = activation, up, down, conv_clamp
        self.register_buffer('resample_filter', upfirdn2d.setup_filter(list(resample_filter)))
        self.padding = kernel_size // 2
        self.weight_gain = 1 / math.sqrt(in_channels * kernel_size ** 2)
        self.act_gain = bias_act.activation_funcs[activation].def_gain
        self.instance_norm, self.lr_multiplier = instance_norm, lr_multiplier
        fmt = torch.channels_last if channels_last else torch.contiguous_format
        weight = torch.randn([out_channels, in_channels, kernel_size, kernel_size]).to(memory_format=fmt)
        bias_t = torch.zeros([out_channels]) if bias else None
        if trainable:
            self.weight = torch.nn.Parameter(weight)
            self.bias = torch.nn.Parameter(bias_t) if bias_t is not None else None
        else:
            self.register_buffer('weight', weight)
            if bias_t is not None:
                self.register_buffer('bias', bias_t)
            else:
                self.bias = None

    def _scaled_parameters(self, x, gain):
        """(w, b, act_gain, clamp) of ``forward``: equalised-lr weight scale, and the gain of a linear un-clamped layer folded into the weights."""
        clamp = self.conv_clamp * gain if self.conv_clamp is not None else None
        act_gain = self.act_gain * gain
        fold = 1.0
        if self.activation == 'linear' and clamp is None and act_gain != 1:
            # (conv(x, w) + b) * g == conv(x, w * g) + b * g: the gain of a linear, un-clamped layer (the discriminator's skip branches, gain
            # sqrt(0.5)) rides on the weights instead of costing a pass over the output and another over its gradient
            fold, act_gain = act_gain, 1.0
        ws, bs = self.weight_gain * self.lr_multiplier * fold, self.lr_multiplier * fold
        self.__dict__['_eqlr_scales'] = (ws, bs)     # what the next eqlr.batched block of the enclosing network prepares for this layer
        w = eqlr.lookup(self.weight, ws)
        if w is None:
            w = self.weight * ws
        b = None
        if self.bias is not None:
            b = eqlr.lookup(self.bias, bs) if x.dtype == self.bias.dtype else None
            if b is None:
                b = self.bias.to(x.dtype)
                if bs != 1.0:     # (a product with 1.0 is the identity bit for bit: no launch for it)
                    b = b * bs
        return w, b, act_gain, clamp

    @staticmethod
    def _fused_dtype(x):
        """Tensor formats the fused layer kernels take: fp32, and fp16 / bf16 (mixed-precision blocks) unless SGV_CONV_LOWP=0."""
        return x.dtype == torch.float32 or (x.dtype in (torch.float16, torch.bfloat16) and conv2d_gradfix.native_lowp)

    def fusable_with_following_fir(self, x):
        """A plain stride-1 3x3 layer on a GPU tensor: ``forward_then_fir`` may pair it with the FIR pass of the next (down-sampling) layer."""
        return (self.up == 1 and self.down == 1 and self.padding == 1 and tuple(self.weight.shape[2:]) == (3, 3) and bool(fused_conv_act.mode) and x.is_cuda
                and self._fused_dtype(x) and self.activation in ('linear', 'lrelu') and not self.instance_norm)

    def accepts_prefiltered(self, x):
        """A down-sampling 3x3 layer that ``forward(..., prefiltered=True)`` serves (x: the tensor the FIR pass will be applied to)."""
        return (self.up == 1 and self.down == 2 and self.padding == 1 and tuple(self.weight.shape[2:]) == (3, 3) and bool(fused_conv_act.mode) and x.is_cuda
                and self._fused_dtype(x) and self.activation in ('linear', 'lrelu') and not self.instance_norm)

    def forward_then_fir(self, x, f, pads, gain=1, with_input_alias=False):
        """upfirdn2d(self(x, gain), f, padding=pads) -- this layer followed by the FIR pass in front of the next layer's strided convolution, as one
        autograd node whose backward pass runs the FIR's gradient and this layer's activation gradient in one kernel."""
        w, b, act_gain, clamp = self._scaled_parameters(x, gain)
        return fused_conv_act.conv3x3_bias_act_then_fir(x, w, b, f, pads, act=self.activation, gain=act_gain, clamp=clamp, with_input_alias=with_input_alias)

    def forward(self, x, gain=1, residual=None, prefiltered=False):
        """``residual`` (optional, same shape as the result): added to the layer's output -- the `y.add_(x)` of the residual discriminator block
        (networks.py:343-345) folded into the layer so that the down-sampling convolution can do it in its epilogue.
        ``prefiltered``: x already went through this down-sampling layer's FIR pass (3x3: ``forward_then_fir`` of the previous layer; the 1x1 skip
        branch with a residual: ``fused_fir_act.fir_down_with_input_alias``)."""
        w, b, act_gain, clamp = self._scaled_parameters(x, gain)
        if self.up == 1 and self.down == 1 and self.padding == 1 and tuple(w.shape[2:]) == (3, 3) and fused_conv_act.mode and x.is_cuda \
                and self._fused_dtype(x) and self.activation in ('linear', 'lrelu'):
            # stride-1 3x3 layer (DiscriminatorBlock conv0, epilogue conv): convolution + bias + activation as one kernel where served
            x = fused_conv_act.conv3x3_bias_act(x, w, bias=b, act=self.activation, gain=act_gain, clamp=clamp)
        elif self.up == 1 and self.down == 1 and tuple(w.shape[2:]) == (1, 1) and w.shape[1] <= 4 and w.shape[0] > w.shape[1] and x.is_cuda \
                and x.dtype == torch.float32 and x.is_contiguous() and pointwise.enabled:
            # fromRGB: 1x1 convolution from <= 4 channels + bias + activation as one streaming kernel
            x = pointwise.pointwise_conv_bias_act(x, w.reshape(1, w.shape[0], w.shape[1]), b, act=self.activation, gain=act_gain, clamp=clamp)
        elif residual is not None and self.up == 1 and self.down > 1 and tuple(w.shape[2:]) == (1, 1) and b is None and self.activation == 'linear' \
                and act_gain == 1 and clamp is None and x.is_cuda and self._fused_dtype(x):
            # skip branch of the residual block (gain already on the weights): FIR + decimate, then the 1x1 convolution whose store adds the
            # other branch's result
            x = conv2d_resample.downsampling_conv1x1(x, w, self.resample_filter, down=self.down, padding=self.padding, residual=residual, prefiltered=prefiltered)
            residual = None
        elif self.up == 1 and self.down == 2 and self.padding == 1 and tuple(w.shape[2:]) == (3, 3) and fused_conv_act.mode and x.is_cuda \
                and self._fused_dtype(x) and self.activation in ('linear', 'lrelu') and not self.instance_norm:
            # down-sampling 3x3 layer (DiscriminatorBlock conv1): FIR pass, then strided convolution + bias + activation (+ residual) as one kernel
            xb = x if prefiltered else conv2d_resample.downsampling_filter_pass(x, self.resample_filter, down=self.down, padding=self.padding)
            x = fused_down_act.strided_conv3x3_bias_act(xb, w, bias=b, act=self.activation, gain=act_gain, clamp=clamp, residual=residual)
            residual = None
        else:
            assert not prefiltered, 'prefiltered input is only understood by the fused down-sampling paths'
            x = conv2d_resample.conv2d_resample(x=x, w=conv2d_gradfix.cast_weight(w, x), f=self.resample_filter, up=self.up, down=self.down,
                                                padding=self.padding, flip_weight=(self.up == 1))
            if b is not None or self.activation != 'linear' or act_gain != 1 or clamp is not None:   # (a no-op bias_act hands its input back as-is)
                x = bias_act.bias_act(x, b, act=self.activation, gain=act_gain, clamp=clamp)
        if residual is not None:
            x = x + residual   # out of place: the residual may be an activation output that its own backward pass needs
        if self.instance_norm:
            x = (x - x.mean(dim=(2, 3), keepdim=True)) / (x.std(dim=(2, 3), keepdim=True) + 1e-8)
        return x


class TemporalInput(torch.nn.Module):
    """4x4 synthesis input: a learned constant concatenated with the motion code broadcast over the 4x4 grid."""

    def __init__(self, cfg, channel_dim, motion_v_dim):
        super().__init__()
        self.cfg = cfg
        self.motion_v_dim = motion_v_dim
        self.const = torch.nn.Parameter(torch.randn(1, channel_dim, 4, 4))

    def get_dim(self):
        return self.motion_v_dim + self.const.shape[1]

    def forward(self, motion_v):
        n = motion_v.shape[0]
        side = self.const.shape[2:]
        return torch.cat([self.const.expand(n, -1, -1, -1), motion_v[:, :, None, None].expand(-1, -1, *side)], dim=1)


class GenInput(torch.nn.Module):
    def __init__(self, cfg, channel_dim, motion_v_dim=None):
        super().__init__()
        self.cfg = cfg
        if cfg.input.type == 'const':
            self.input = torch.nn.Parameter(torch.randn([channel_dim, 4, 4]))
            self.total_dim = channel_dim
        elif cfg.input.type == 'temporal':
            self.input = TemporalInput(cfg, channel_dim, motion_v_dim=motion_v_dim)
            self.total_dim = self.input.get_dim()
        else:
            raise NotImplementedError(f'Unknown input type: {cfg.input.type}')

    def forward(self, batch_size, motion_v=None, dtype=None, memory_format=None):
        if self.cfg.input.type == 'const':
            x = self.input.to(dtype=dtype, memory_format=memory_format)
            return x.unsqueeze(0).repeat([batch_size, 1, 1, 1])
        return self.input(motion_v=motion_v)


def construct_log_spaced_freqs(max_num_frames, skip_small_t_freqs=0):
    """pi * 2^k / T for k = 0..log2(T)-1, T = max_num_frames rounded up to a power of two."""
    time_resolution = 2 ** math.ceil(math.log2(max_num_frames))
    num = int(math.ceil(math.log2(time_resolution)))
    powers = 2.0 ** torch.arange(num - skip_small_t_freqs, dtype=torch.float32)
    return (powers * math.pi / time_resolution).unsqueeze(0)


class FixedTimeEncoder(torch.nn.Module):
    def __init__(self, max_num_frames, skip_small_t_freqs=0):
        super().__init__()
        assert max_num_frames >= 1
        self.register_buffer('fourier_coefs', construct_log_spaced_freqs(max_num_frames, skip_small_t_freqs=skip_small_t_freqs))

    def get_dim(self):
        return self.fourier_coefs.shape[1] * 2

    def forward(self, t):
        assert t.ndim == 2
        raw = self.fourier_coefs * t.reshape(-1, 1).float()
        return torch.cat([raw.sin(), raw.cos()], dim=1)


class TemporalDifferenceEncoder(torch.nn.Module):
    """Discriminator-side embedding of the time gaps between the frames of a clip:
    learned table lookup of round(dt) concatenated with fixed Fourier features of dt."""

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        if cfg.sampling.num_frames_per_video > 1:
            self.d = 256
            self.const_embed = torch.nn.Embedding(cfg.sampling.max_num_frames, self.d)
            self.time_encoder = FixedTimeEncoder(cfg.sampling.max_num_frames, skip_small_t_freqs=cfg.get('skip_small_t_freqs', 0))

    def get_dim(self):
        nf = self.cfg.sampling.num_frames_per_video
        if nf == 1:
            return 1
        per_diff = self.d + self.time_encoder.get_dim()
        return per_diff if self.cfg.sampling.type == 'uniform' else per_diff * (nf - 1)

    def forward(self, t):
        nf = self.cfg.sampling.num_frames_per_video
        misc.assert_shape(t, [None, nf])
        if nf == 1:
            return torch.zeros(len(t), 1, device=t.device)
        diffs = (t[:, 1] - t[:, 0]) if self.cfg.sampling.type == 'uniform' else (t[:, 1:] - t[:, :-1]).reshape(-1)
        table = self.const_embed(diffs.float().round().long())
        fourier = self.time_encoder(diffs.unsqueeze(1))
        return torch.cat([table, fourier], dim=1).reshape(t.shape[0], -1)


class EqLRConv1d(torch.nn.Module):
    """Equalised-lr 1-D convolution of the motion trajectory network (lrelu slope 0.2, no sqrt(2) gain)."""

    def __init__(self, in_features, out_features, kernel_size, padding=0, stride=1, activation='linear', lr_multiplier=1.0,
                 bias=True, bias_init=0.0):
        super().__init__()
        assert activation in ('lrelu', 'linear')
        self.activation, self.padding, self.stride = activation, padding, stride
        self.weight = torch.nn.Parameter(torch.randn([out_features, in_features, kernel_size]) / lr_multiplier)
        self.bias = torch.nn.Parameter(torch.full([out_features], float(bias_init))) if bias else None
        self.weight_gain = lr_multiplier / math.sqrt(in_features * kernel_size)
        self.bias_gain = lr_multiplier

    def forward(self, x):
        assert x.ndim == 3
        w = self.weight.to(x.dtype) * self.weight_gain
        b = self.bias
        if b is not None:
            b = b.to(x.dtype)
            if self.bias_gain != 1:
                b = b * self.bias_gain
        y = F.conv1d(x, w, bias=None, stride=self.stride, padding=self.padding)
        # bias + leaky-relu through the fused kernel (gain 1: the reference applies F.leaky_relu without sqrt(2))
        return bias_act.bias_act(y, b, dim=1, act=self.activation, gain=1)

    def forward_nlc(self, x):
        """The same layer on a [B, L, C] trajectory (the layout the motion network holds it in, motion.py:84-110), without the two
        permutes around F.conv1d: the valid convolution is the product of the unfolded trajectory [B * Lout, k * C] with the weight
        re-laid as [O, k * C], i.e. ONE dense-layer kernel (ops/fc.py -> csrc/fc.hip, exact-fp32 MFMA) with the weight gain, bias gain,
        bias and leaky relu in its epilogue; gradients through the same kernel's data- / weight-gradient forms.  Returns [B, Lout, O]."""
        assert x.ndim == 3 and self.stride == 1 and self.padding == 0
        bsz, length, ch = x.shape
        o, c, k = self.weight.shape
        assert c == ch
        lout = length - k + 1
        cols = x.unfold(1, k, 1).permute(0, 1, 3, 2).reshape(bsz * lout, k * ch)      # [B, Lout, C, k] view -> [B * Lout, (k, C)]
        w2 = self.weight.permute(0, 2, 1).reshape(o, k * ch)                          # [O, (k, C)]
        y = fc.dense(cols, w2.to(x.dtype), self.bias, weight_gain=self.weight_gain, bias_gain=self.bias_gain, act=self.activation, act_gain=1)
        return y.reshape(bsz, lout, o)


# ------------------------------------------------------------------------------------------------
# Host-side frame-index sampling (layers.py:377-435 contract): which frames of a video a clip uses.


def sample_frames(cfg, total_video_len, **kwargs):
    if cfg['type'] == 'random':
        return random_frame_sampling(cfg, total_video_len, **kwargs)
    if cfg['type'] == 'uniform':
        return uniform_frame_sampling(cfg, total_video_len, **kwargs)
    raise NotImplementedError(cfg['type'])


def random_frame_sampling(cfg, total_video_len, use_fractional_t=False):
    nf = cfg['num_frames_per_video']
    lo, hi = nf - 1, min(total_video_len - 1, cfg.get('max_dist', float('inf')))
    dists = cfg.get('total_dists')
    choices = [d for d in dists if lo <= d <= hi] if isinstance(dists, (list, tuple)) else range(lo, hi)
    span = random.choice(choices)
    offset = random.random() * (total_video_len - span - 1) if use_fractional_t else random.randint(0, total_video_len - span - 1)
    idx = [offset]
    if nf > 1:
        idx.append(offset + span)
    if nf > 2:
        idx.extend(offset + d for d in random.sample(range(1, span), k=nf - 2))
    return np.array(sorted(idx))


def uniform_frame_sampling(cfg, total_video_len, use_fractional_t=False):
    nf = cfg['num_frames_per_video']
    dists = cfg.get('dists_between_frames')
    if isinstance(dists, (list, tuple)):
        valid = [d for d in dists if d <= cfg['max_dist_between_frames'] and d * nf - d + 1 <= total_video_len]
        d = random.choice(valid)
    else:
        d = random.randint(1, min(cfg.get('max_dist', float('inf')), total_video_len // nf))
    total = d * nf - d + 1
    offset = random.random() * (total_video_len - total) if use_fractional_t else random.randint(0, total_video_len - total)
    return offset + np.arange(nf) * d
