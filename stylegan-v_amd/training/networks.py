"""StyleGAN-V generator and discriminator on top of the native op layer.

Module tree and parameter names mirror the reference's ``src/training/networks.py`` (checkpoints map
one-to-one): ``modulated_conv2d`` (:29-86), ``SynthesisLayer`` (:90), ``ToRGBLayer`` (:148),
``SynthesisBlock`` (:167), ``SynthesisNetwork`` (:270), ``Generator`` (:370), ``DiscriminatorBlock``
(:405), ``MinibatchStdLayer`` (:492), ``DiscriminatorEpilogue`` (:518), ``Discriminator`` (:580).
Architecture/numerics per SURVEY.md appendix E.

What is different underneath (MI355X-first, same results to fp32 round-off):
  * weight demodulation never builds w[N,O,I,kh,kw]: d = rsqrt(s^2 . sum_k W^2 + eps) comes from
    ``ops.modulation.demod_coefs`` (csrc/modulate.hip, wave-shuffle reduction);
  * the per-sample channel scalings x*s and y*d are single streaming kernels (``scale_channels``);
  * every FIR resampling / bias+activation step is a hand-written gfx950 kernel (upfirdn2d, bias_act);
  * low-precision blocks can run bf16 as well as the reference's fp16 (``lowp_dtype``).
  * 3x3 convolutions (forward, data and weight gradients, stride 1 and the stride-2 pairs around the FIR) run on the
    hand-written matrix-core kernels of csrc/conv3x3*.h / wrw_kernel.h; a stride-1 SynthesisLayer is ONE kernel forward
    (``ops.fused_conv_act``: styles in the operand load, dcoefs / bias / lrelu / gain / clamp in the accumulator store), an
    up-sampling one is transposed convolution + one FIR kernel with the same epilogue (``ops.fused_fir_act``);
  * on the GPU, eval-mode synthesis uses the same activation-scaling formulation as training (``prefer_native_inference``)
    instead of the reference's per-sample grouped convolution -- the same function, without per-sample weight tensors.
Shapes the kernels do not serve (4x4 layers, odd channel counts, 16-bit tensors) fall back to MIOpen, as the reference leaves
every convolution to cuDNN.
"""

import contextlib
import math
import os
import threading

import numpy as np
import torch

from ..torch_utils import misc
from ..torch_utils.ops import bias_act, conv2d_gradfix, conv2d_resample, eqlr, fc, fma, fused_conv_act, fused_fir_act, modulation, pointwise, upfirdn2d
from .layers import Conv2dLayer, FullyConnectedLayer, GenInput, MappingNetwork, TemporalDifferenceEncoder
from .motion import MotionMappingNetwork

deferred_rgb = os.environ.get('SGV_DEFER_RGB', '1') != '0'      # a block's ToRGB runs inside the next block (see SynthesisNetwork.forward)


prefer_native_inference = True   # GPU eval-mode synthesis: scale-conv-scale through the native kernels instead of the grouped convolution
alias_in_fir = os.environ.get('SGV_ALIAS_IN_FIR', '0') != '0'       # which of the block input's two consumers receives the other's gradient (see DiscriminatorBlock.forward)
residual_in_skip = os.environ.get('SGV_RES_IN_SKIP', '1') != '0'   # where the residual discriminator block forms its sum (see DiscriminatorBlock.forward)


@misc.profiled_function
def modulated_conv2d(x, weight, styles, noise=None, up=1, down=1, padding=0, resample_filter=None, demodulate=True,
                     flip_weight=True, fused_modconv=True):
    """y[n,o] = d[n,o] * conv(x[n,i] * s[n,i], W[o,i])  with  d = rsqrt(sum_{i,k}(W*s)^2 + 1e-8)  (+ noise).

    x [N,I,H,W], weight [O,I,kh,kw], styles [N,I], noise broadcastable to the output or None.
    ``fused_modconv=True`` evaluates the same function as one grouped convolution with per-sample
    weights (the reference's inference path, networks.py:77-86); ``False`` scales activations before
    and after a shared-weight convolution (training path, networks.py:65-74).
    """
    n = x.shape[0]
    oc, ic, kh, kw = weight.shape
    misc.assert_shape(x, [n, ic, None, None])
    misc.assert_shape(styles, [n, ic])

    # fp16 range guard of the reference (networks.py:50-52); bf16/fp32 have the exponent range and skip it.
    if x.dtype == torch.float16 and demodulate:
        weight = weight * (1 / math.sqrt(ic * kh * kw) / weight.norm(float('inf'), dim=[1, 2, 3], keepdim=True))
        styles = styles / styles.norm(float('inf'), dim=1, keepdim=True)

    dcoefs = modulation.demod_coefs(weight, styles) if demodulate else None  # [N, O], fp32

    if not fused_modconv:
        x = modulation.scale_channels(x, styles)
        x = conv2d_resample.conv2d_resample(x=x, w=conv2d_gradfix.cast_weight(weight, x), f=resample_filter, up=up, down=down, padding=padding,
                                            flip_weight=flip_weight)
        if demodulate and noise is not None:
            return fma.fma(x, dcoefs.to(x.dtype).reshape(n, -1, 1, 1), noise.to(x.dtype))
        if demodulate:
            return modulation.scale_channels(x, dcoefs)
        if noise is not None:
            return x.add_(noise.to(x.dtype))
        return x

    # Grouped-convolution formulation: per-sample weights [N*O, I, kh, kw], batch folded into channels.
    w = weight.unsqueeze(0) * styles.reshape(n, 1, ic, 1, 1)
    if demodulate:
        w = w * dcoefs.reshape(n, oc, 1, 1, 1)
    x = x.reshape(1, n * ic, *x.shape[2:])
    x = conv2d_resample.conv2d_resample(x=x, w=w.reshape(n * oc, ic, kh, kw).to(x.dtype), f=resample_filter, up=up, down=down,
                                        padding=padding, groups=n, flip_weight=flip_weight)
    x = x.reshape(n, oc, *x.shape[2:])
    if noise is not None:
        x = x.add_(noise)
    return x


class SynthesisLayer(torch.nn.Module):
    def __init__(self, in_channels, out_channels, w_dim, resolution, kernel_size=3, up=1, activation='lrelu',
                 resample_filter=(1, 3, 3, 1), conv_clamp=None, channels_last=False, cfg=None):
        super().__init__()
        self.cfg = cfg
        self.resolution, self.up, self.activation, self.conv_clamp = resolution, up, activation, conv_clamp
        self.register_buffer('resample_filter', upfirdn2d.setup_filter(list(resample_filter)))
        self.padding = kernel_size // 2
        self.act_gain = bias_act.activation_funcs[activation].def_gain
        self.affine = FullyConnectedLayer(w_dim, in_channels, bias_init=1)
        fmt = torch.channels_last if channels_last else torch.contiguous_format
        self.weight = torch.nn.Parameter(torch.randn([out_channels, in_channels, kernel_size, kernel_size]).to(memory_format=fmt))
        if cfg.use_noise:
            self.register_buffer('noise_const', torch.randn([resolution, resolution]))
            self.noise_strength = torch.nn.Parameter(torch.zeros([]))
        self.bias = torch.nn.Parameter(torch.zeros([out_channels]))

    def forward(self, x, w, noise_mode='random', fused_modconv=True, gain=1, styles=None, alias=False):
        """``styles``: ``self.affine(w)`` already evaluated by the caller (SynthesisNetwork.forward runs the affines of a whole pass as one launch).
        ``alias=True``: return (y, x_alias) -- x_alias is x for its other consumer (the previous block's ToRGB), whose gradient is then summed inside this layer's
        own backward pass (ops/modulation.py ``scale_channels_with_alias``); a path that cannot do that returns x itself."""
        if alias is True:
            box = [x]
            y = self.forward(x, w, noise_mode=noise_mode, fused_modconv=fused_modconv, gain=gain, styles=styles, alias=box)
            return y, box[0]
        assert noise_mode in ('random', 'const', 'none')
        misc.assert_shape(x, [None, self.weight.shape[1], self.resolution // self.up, self.resolution // self.up])
        if styles is None:
            styles = self.affine(w)
        noise = None
        if self.cfg.use_noise and noise_mode == 'random':
            noise = torch.randn([x.shape[0], 1, self.resolution, self.resolution], device=x.device) * self.noise_strength
        if self.cfg.use_noise and noise_mode == 'const':
            noise = self.noise_const * self.noise_strength
        clamp = self.conv_clamp * gain if self.conv_clamp is not None else None
        if self.up == 2 and not fused_modconv and noise is None and fused_fir_act.enabled and x.is_cuda and self.activation in ('linear', 'lrelu'):
            # Up-sampling layer, training path: x*s -> transposed conv -> [FIR * dcoefs + bias -> act -> clamp] where the
            # bracket is one kernel (ops/fused_fir_act.py) instead of upfirdn2d + scale + bias_act.
            weight, s = self.weight, styles
            if x.dtype == torch.float16:  # same fp16 range guard as modulated_conv2d
                weight = weight * (1 / math.sqrt(weight[0].numel()) / weight.norm(float('inf'), dim=[1, 2, 3], keepdim=True))
                s = s / s.norm(float('inf'), dim=1, keepdim=True)
            dcoefs = modulation.demod_coefs(weight, s)
            if isinstance(alias, list):
                x, alias[0] = modulation.scale_channels_with_alias(x, s)
            else:
                x = modulation.scale_channels(x, s)
            x, fir_pad = conv2d_resample.upsampling_conv_parts(x, conv2d_gradfix.cast_weight(weight, x), self.resample_filter, up=self.up, padding=self.padding,
                                                               flip_weight=False)
            return fused_fir_act.fir_bias_act(x, self.resample_filter, scale=dcoefs, bias=self.bias, padding=fir_pad, fir_gain=self.up ** 2,
                                              act=self.activation, gain=self.act_gain * gain, clamp=clamp)
        if self.up == 1 and not fused_modconv and noise is None and fused_conv_act.mode and x.is_cuda \
                and (x.dtype == torch.float32 or (x.dtype in (torch.float16, torch.bfloat16) and conv2d_gradfix.native_lowp)) \
                and self.activation in ('linear', 'lrelu') and tuple(self.weight.shape[2:]) == (3, 3) and self.padding == 1:
            # Stride-1 layer, training formulation: [x*s -> conv3x3 -> *dcoefs + bias -> act -> clamp] as one kernel where the shape is
            # served (ops/fused_conv_act.py), as the four-op composition otherwise.
            weight, s = self.weight, styles
            if x.dtype == torch.float16:  # same fp16 range guard as modulated_conv2d (the data gradient is stored before it meets the styles)
                weight = weight * (1 / math.sqrt(weight[0].numel()) / weight.norm(float('inf'), dim=[1, 2, 3], keepdim=True))
                s = s / s.norm(float('inf'), dim=1, keepdim=True)
            dcoefs = modulation.demod_coefs(weight, s)
            return fused_conv_act.conv3x3_bias_act(x, weight, styles=s, dcoefs=dcoefs, bias=self.bias, act=self.activation,
                                                   gain=self.act_gain * gain, clamp=clamp)
        x = modulated_conv2d(x=x, weight=self.weight, styles=styles, noise=noise, up=self.up, padding=self.padding,
                             resample_filter=self.resample_filter, flip_weight=(self.up == 1), fused_modconv=fused_modconv)
        return bias_act.bias_act(x, self.bias.to(x.dtype), act=self.activation, gain=self.act_gain * gain, clamp=clamp)


class ToRGBLayer(torch.nn.Module):
    def __init__(self, in_channels, out_channels, w_dim, kernel_size=1, conv_clamp=None, channels_last=False):
        super().__init__()
        self.conv_clamp = conv_clamp
        self.affine = FullyConnectedLayer(w_dim, in_channels, bias_init=1)
        fmt = torch.channels_last if channels_last else torch.contiguous_format
        self.weight = torch.nn.Parameter(torch.randn([out_channels, in_channels, kernel_size, kernel_size]).to(memory_format=fmt))
        self.bias = torch.nn.Parameter(torch.zeros([out_channels]))
        self.weight_gain = 1 / math.sqrt(in_channels * kernel_size ** 2)

    def forward(self, x, w, fused_modconv=True, styles=None):
        if styles is None:
            styles = self.affine(w, gain=self.weight_gain)      # (networks.py:314: affine(w) * weight_gain; the factor rides on the dense kernel's output gain)
        oc, ic, kh, kw = self.weight.shape
        if kh == 1 and kw == 1 and oc <= 4 and pointwise.enabled and x.is_cuda and x.is_contiguous():
            # y[n,o] = sum_i x[n,i] * (W[o,i] * s[n,i]): the style goes into per-sample weights [N,3,I] and the whole layer is
            # one pass over x (csrc/pointwise.hip) instead of x*s followed by a C_out=3 convolution.  Same function for
            # either value of `fused_modconv` (no demodulation here).
            w_ps = self.weight.reshape(1, oc, ic) * styles.reshape(-1, 1, ic)
            x = pointwise.pointwise_conv(x, w_ps)
        else:
            x = modulated_conv2d(x=x, weight=self.weight, styles=styles, demodulate=False, fused_modconv=fused_modconv)
        return bias_act.bias_act(x, self.bias.to(x.dtype), clamp=self.conv_clamp)


class SynthesisBlock(torch.nn.Module):
    def __init__(self, in_channels, out_channels, w_dim, motion_v_dim, resolution, img_channels, is_last, architecture='skip',
                 resample_filter=(1, 3, 3, 1), conv_clamp=None, use_fp16=False, fp16_channels_last=False, lowp_dtype=torch.float16,
                 cfg=None, **layer_kwargs):
        assert architecture in ('orig', 'skip', 'resnet')
        super().__init__()
        self.cfg = cfg
        self.in_channels, self.w_dim, self.resolution, self.img_channels = in_channels, w_dim, resolution, img_channels
        self.is_last, self.architecture, self.use_fp16, self.lowp_dtype = is_last, architecture, use_fp16, lowp_dtype
        self.channels_last = use_fp16 and fp16_channels_last
        self.register_buffer('resample_filter', upfirdn2d.setup_filter(list(resample_filter)))
        self.num_conv = self.num_torgb = 0
        common = dict(w_dim=w_dim, resolution=resolution, conv_clamp=conv_clamp, channels_last=self.channels_last, kernel_size=3, cfg=cfg)
        if in_channels == 0:
            self.input = GenInput(cfg, out_channels, motion_v_dim=motion_v_dim)
            conv1_in = self.input.total_dim
        else:
            self.conv0 = SynthesisLayer(in_channels, out_channels, up=2, resample_filter=resample_filter, **common, **layer_kwargs)
            self.num_conv += 1
            conv1_in = out_channels
        self.conv1 = SynthesisLayer(conv1_in, out_channels, **common, **layer_kwargs)
        self.num_conv += 1
        if is_last or architecture == 'skip':
            self.torgb = ToRGBLayer(out_channels, img_channels, w_dim=w_dim, conv_clamp=conv_clamp, channels_last=self.channels_last)
            self.num_torgb += 1
        if in_channels != 0 and architecture == 'resnet':
            self.skip = Conv2dLayer(in_channels, out_channels, kernel_size=1, bias=False, up=2, resample_filter=resample_filter,
                                    channels_last=self.channels_last)

    def affine_layers(self):
        """The layers that own a style affine, in call order (one column of this block's ws each)."""
        return ([self.conv0] if self.in_channels != 0 else []) + [self.conv1] + ([self.torgb] if (self.is_last or self.architecture == 'skip') else [])

    def forward(self, x, img, ws, motion_v=None, force_fp32=False, fused_modconv=None, styles=None, pending_rgb=None, defer_rgb=False, **layer_kwargs):
        """``defer_rgb`` / ``pending_rgb`` (SynthesisNetwork.forward, skip architecture): a block may leave its ToRGB to the NEXT block, which runs it on the alias of x
        that its own conv0 hands back -- the two gradients of x (from ToRGB and from the next block's conv0) are then summed inside conv0's backward pass instead of by a
        separate full-tensor addition.  With defer_rgb the block returns (x, img, pending) where img still lacks this block's RGB contribution."""
        s_iter = iter(styles) if styles is not None else iter(lambda: None, 0)      # (styles: one tensor per affine_layers() entry, or None: every layer runs its own)
        if isinstance(ws, (tuple, list)):
            # the per-layer latents already split by the caller (SynthesisNetwork.forward unbinds `ws` ONCE: one stack in the backward pass instead of a
            # cat + zero-fill + copy + add per block, 20 launches per generator backward)
            assert len(ws) == self.num_conv + self.num_torgb
            w_iter, ws = iter(ws), ws[0]
        else:
            misc.assert_shape(ws, [None, self.num_conv + self.num_torgb, self.w_dim])
            w_iter = iter(ws.unbind(dim=1))
        dtype = self.lowp_dtype if self.use_fp16 and not force_fp32 else torch.float32
        fmt = torch.channels_last if self.channels_last and not force_fp32 else torch.contiguous_format
        if fused_modconv is None:  # reference policy (networks.py:230-232)
            fused_modconv = (not self.training) and (dtype == torch.float32 or (isinstance(x, torch.Tensor) and int(x.shape[0]) == 1))
            if fused_modconv and prefer_native_inference and ws.is_cuda and dtype == torch.float32:
                fused_modconv = False

        if self.in_channels == 0:
            x = self.input(ws.shape[0], motion_v=motion_v, dtype=dtype, memory_format=fmt)
            x = self.conv1(x, next(w_iter), fused_modconv=fused_modconv, styles=next(s_iter), **layer_kwargs)
        else:
            misc.assert_shape(x, [None, self.in_channels, self.resolution // 2, self.resolution // 2])
            x = x.to(dtype=dtype, memory_format=fmt)
            if self.architecture == 'resnet':
                y = self.skip(x, gain=math.sqrt(0.5))
                x = self.conv0(x, next(w_iter), fused_modconv=fused_modconv, styles=next(s_iter), **layer_kwargs)
                x = self.conv1(x, next(w_iter), fused_modconv=fused_modconv, gain=math.sqrt(0.5), styles=next(s_iter), **layer_kwargs)
                x = y.add_(x)
            else:
                if pending_rgb is not None:
                    x, x_prev = self.conv0(x, next(w_iter), fused_modconv=fused_modconv, styles=next(s_iter), alias=True, **layer_kwargs)
                    img = pending_rgb(x_prev, img)      # the previous block's ToRGB, on the alias of its x
                else:
                    x = self.conv0(x, next(w_iter), fused_modconv=fused_modconv, styles=next(s_iter), **layer_kwargs)
                x = self.conv1(x, next(w_iter), fused_modconv=fused_modconv, styles=next(s_iter), **layer_kwargs)

        if img is not None:
            misc.assert_shape(img, [None, self.img_channels, self.resolution // 2, self.resolution // 2])
            img = upfirdn2d.upsample2d(img, self.resample_filter)
        if self.is_last or self.architecture == 'skip':
            w_rgb, s_rgb = next(w_iter), next(s_iter)

            def rgb(x_in, img_in):
                y = self.torgb(x_in, w_rgb, fused_modconv=fused_modconv, styles=s_rgb).to(dtype=torch.float32, memory_format=torch.contiguous_format)
                return img_in.add_(y) if img_in is not None else y
            if defer_rgb:
                assert x.dtype == dtype
                return x, img, rgb
            img = rgb(x, img)
        assert x.dtype == dtype
        assert img is None or img.dtype == torch.float32
        return (x, img, None) if defer_rgb else (x, img)


class SynthesisNetwork(torch.nn.Module):
    def __init__(self, w_dim, img_resolution, img_channels, channel_base=32768, channel_max=512, num_fp16_res=0, cfg=None, **block_kwargs):
        assert img_resolution >= 4 and img_resolution & (img_resolution - 1) == 0
        super().__init__()
        self.w_dim, self.cfg, self.img_resolution, self.img_channels = w_dim, cfg, img_resolution, img_channels
        self.img_resolution_log2 = int(np.log2(img_resolution))
        self.block_resolutions = [2 ** i for i in range(2, self.img_resolution_log2 + 1)]
        channels = {res: min(channel_base // res, channel_max) for res in self.block_resolutions}
        fp16_resolution = max(2 ** (self.img_resolution_log2 + 1 - num_fp16_res), 8)

        if cfg.motion.v_dim > 0:
            self.motion_encoder = MotionMappingNetwork(cfg)
            self.motion_v_dim = self.motion_encoder.get_dim()
        else:
            self.motion_encoder, self.motion_v_dim = None, 0

        self.num_ws = 0
        for res in self.block_resolutions:
            block = SynthesisBlock(channels[res // 2] if res > 4 else 0, channels[res],
                                   w_dim=w_dim + (self.motion_v_dim if cfg.time_enc.cond_type == 'concat_w' else 0),
                                   motion_v_dim=self.motion_v_dim, resolution=res, img_channels=img_channels,
                                   is_last=(res == img_resolution), use_fp16=(res >= fp16_resolution), cfg=cfg, **block_kwargs)
            self.num_ws += block.num_conv + (block.num_torgb if res == img_resolution else 0)
            setattr(self, f'b{res}', block)

    def forward(self, ws, t=None, c=None, motion_z=None, motion_v=None, **block_kwargs):
        assert len(ws) == len(c) == len(t), f'Wrong shape: {ws.shape}, {c.shape}, {t.shape}'
        assert t.ndim == 2, f'Wrong shape: {t.shape}'
        misc.assert_shape(ws, [None, self.num_ws, self.w_dim])
        frames = t.shape[1]
        cond = self.cfg.time_enc.cond_type
        if self.motion_encoder is None:
            motion_v = None
        elif motion_v is None:
            motion_v = self.motion_encoder(c, t, motion_z=motion_z)['motion_v']  # [B*F, motion_v_dim]
        ws = ws.repeat_interleave(frames, dim=0)  # every frame of a video shares its w: [B*F, num_ws, w_dim]
        if motion_v is not None and cond == 'concat_w':
            ws = torch.cat([ws, motion_v.unsqueeze(1).expand(-1, self.num_ws, -1)], dim=2)
        elif motion_v is not None and cond == 'sum_w':
            ws = ws + motion_v.unsqueeze(1)
        ws = ws.to(torch.float32)

        x = img = None
        w_idx = 0
        all_w = ws.unbind(dim=1)
        # The style affines of the whole pass (every SynthesisLayer / ToRGBLayer: `self.affine(w)`, networks.py:116,153) as ONE launch, and two in the backward
        # pass (ops/fc.py `grouped_affine` -> sgv_fc_grouped): 21 launches of 10-14 us each at FFS-256 otherwise, whatever the batch.
        all_styles = None
        if fc.grouped and ws.is_cuda and ws.dtype == torch.float32:
            layers, cols, gains, k = [], [], [], 0
            for res in self.block_resolutions:
                block = getattr(self, f'b{res}')
                for j, layer in enumerate(block.affine_layers()):
                    layers.append(layer.affine)
                    cols.append(k + j)
                    gains.append(layer.weight_gain if isinstance(layer, ToRGBLayer) else None)
                k += block.num_conv
            all_styles = fc.grouped_affine(ws, cols, layers, gains)
        s_idx = 0
        # Training pass on the GPU, skip architecture: every block but the last leaves its ToRGB to the next block (SynthesisBlock.forward `defer_rgb`): x_k's two
        # gradients meet inside the next block's conv0 backward instead of in a separate addition (6 full-tensor additions per generator backward, 0.9 ms).
        defer = (deferred_rgb and ws.is_cuda and torch.is_grad_enabled() and ws.requires_grad and block_kwargs.get('fused_modconv') in (None, False) and self.training
                 and all(getattr(self, f'b{r}').architecture == 'skip' and not getattr(self, f'b{r}').use_fp16 for r in self.block_resolutions)
                 and not block_kwargs.get('force_fp32', False) and not self.cfg.use_noise)
        pending = None
        for res in self.block_resolutions:
            block = getattr(self, f'b{res}')
            # each ToRGB shares its w with the next block's conv0: advance by num_conv only (networks.py:354-357)
            cur_ws = all_w[w_idx:w_idx + block.num_conv + block.num_torgb]
            w_idx += block.num_conv
            n_aff = block.num_conv + block.num_torgb
            cur_styles = all_styles[s_idx:s_idx + n_aff] if all_styles is not None else None
            s_idx += n_aff
            if defer:
                is_final = res == self.block_resolutions[-1]
                out = block(x, img, cur_ws, motion_v=motion_v if cond == 'concat_const' else None, styles=cur_styles, pending_rgb=pending, defer_rgb=not is_final, **block_kwargs)
                x, img, pending = out[0], out[1], (out[2] if len(out) == 3 else None)
            else:
                x, img = block(x, img, cur_ws, motion_v=motion_v if cond == 'concat_const' else None, styles=cur_styles, **block_kwargs)
        return img


class Generator(torch.nn.Module):
    def __init__(self, c_dim, w_dim, img_resolution, img_channels, mapping_kwargs=None, synthesis_kwargs=None, cfg=None):
        super().__init__()
        self.cfg = cfg
        self.sampling_dict = dict(cfg.sampling)
        self.z_dim, self.c_dim, self.w_dim = cfg.z_dim, c_dim, w_dim
        self.img_resolution, self.img_channels = img_resolution, img_channels
        self.synthesis = SynthesisNetwork(w_dim=w_dim, img_resolution=img_resolution, img_channels=img_channels, cfg=cfg, **(synthesis_kwargs or {}))
        self.num_ws = self.synthesis.num_ws
        self.mapping = MappingNetwork(z_dim=self.z_dim, c_dim=c_dim, w_dim=w_dim, num_ws=self.num_ws, **(mapping_kwargs or {}))

    def forward(self, z, c, t, truncation_psi=1, truncation_cutoff=None, **synthesis_kwargs):
        assert len(z) == len(c) == len(t), f'Wrong shape: {z.shape}, {c.shape}, {t.shape}'
        assert t.ndim == 2, f'Wrong shape: {t.shape}'
        ws = self.mapping(z, c, truncation_psi=truncation_psi, truncation_cutoff=truncation_cutoff)
        return self.synthesis(ws, t=t, c=c, **synthesis_kwargs)  # [B*F, C, H, W]


# ------------------------------------------------------------------------------------------------


class DiscriminatorBlock(torch.nn.Module):
    def __init__(self, in_channels, tmp_channels, out_channels, resolution, img_channels, first_layer_idx, architecture='resnet',
                 activation='lrelu', resample_filter=(1, 3, 3, 1), conv_clamp=None, use_fp16=False, fp16_channels_last=False,
                 freeze_layers=0, lowp_dtype=torch.float16, cfg=None):
        assert architecture in ('orig', 'skip', 'resnet')
        super().__init__()
        self.cfg = cfg
        self.in_channels, self.resolution, self.img_channels = in_channels, resolution, img_channels
        self.first_layer_idx, self.architecture, self.use_fp16, self.lowp_dtype = first_layer_idx, architecture, use_fp16, lowp_dtype
        self.channels_last = use_fp16 and fp16_channels_last
        self.register_buffer('resample_filter', upfirdn2d.setup_filter(list(resample_filter)))
        self.num_layers = 0

        def next_trainable():  # Freeze-D bookkeeping: layers below `freeze_layers` become buffers
            trainable = (self.first_layer_idx + self.num_layers) >= freeze_layers
            self.num_layers += 1
            return trainable

        conv0_in = in_channels if in_channels > 0 else tmp_channels
        common = dict(conv_clamp=conv_clamp, channels_last=self.channels_last)
        if in_channels == 0 or architecture == 'skip':
            self.fromrgb = Conv2dLayer(img_channels, tmp_channels, kernel_size=1, activation=activation, trainable=next_trainable(), **common)
        self.conv0 = Conv2dLayer(conv0_in, tmp_channels, kernel_size=3, activation=activation, trainable=next_trainable(), **common)
        self.conv1 = Conv2dLayer(tmp_channels, out_channels, kernel_size=3, activation=activation, down=2, trainable=next_trainable(),
                                 resample_filter=resample_filter, **common)
        if architecture == 'resnet':
            self.skip = Conv2dLayer(conv0_in, out_channels, kernel_size=1, bias=False, down=2, trainable=next_trainable(),
                                    resample_filter=resample_filter, channels_last=self.channels_last)

    def forward(self, x, img, force_fp32=False):
        dtype = self.lowp_dtype if self.use_fp16 and not force_fp32 else torch.float32
        fmt = torch.channels_last if self.channels_last and not force_fp32 else torch.contiguous_format
        if x is not None:
            misc.assert_shape(x, [None, self.in_channels, self.resolution, self.resolution])
            x = x.to(dtype=dtype, memory_format=fmt)
        if self.in_channels == 0 or self.architecture == 'skip':
            misc.assert_shape(img, [None, self.img_channels, self.resolution, self.resolution])
            img = img.to(dtype=dtype, memory_format=fmt)
            y = self.fromrgb(img)
            x = x + y if x is not None else y
            img = upfirdn2d.downsample2d(img, self.resample_filter) if self.architecture == 'skip' else None
        if self.architecture == 'resnet':
            # `y = skip(x); x = conv1(conv0(x)); x = y.add_(x)` of the reference, evaluated in the other order so that the sum is formed in the
            # store of the skip branch's 1x1 convolution (its GEMM has the spare load slots; the 3x3 kernel's MFMA waves do not)
            if residual_in_skip:
                sk = self.skip
                if (self.conv0.fusable_with_following_fir(x) and self.conv1.accepts_prefiltered(x) and tuple(sk.weight.shape[2:]) == (1, 1) and sk.down == 2
                        and sk.up == 1 and sk.bias is None and sk.activation == 'linear' and sk.conv_clamp is None):
                    # x has two consumers.  The skip branch's FIR + decimate node hands x on to conv0 as its second output, so conv0's data
                    # gradient comes back to that node and is added in the store of its own gradient pass (a streaming 2x up-sampling FIR);
                    # conv0 and the FIR pass in front of conv1's strided convolution are one node (one-kernel FIR + activation gradient).
                    f1 = self.conv1.resample_filter
                    pads1 = conv2d_resample.downsampling_pads(f1, self.conv1.down, self.conv1.padding)
                    if alias_in_fir:
                        xd, xc = fused_fir_act.fir_down_with_input_alias(x, sk.resample_filter, sk.down, conv2d_resample.downsampling_pads(sk.resample_filter, sk.down, sk.padding))
                        xb = self.conv0.forward_then_fir(xc, f1, pads1)
                        y = self.conv1(xb, gain=math.sqrt(0.5), prefiltered=True)
                        x = sk(xd, gain=math.sqrt(0.5), residual=y, prefiltered=True)
                    else:   # the other way round: conv0's node hands x to the skip branch and adds into its gradient with atomics in the data-gradient kernel
                        xb, xs_ = self.conv0.forward_then_fir(x, f1, pads1, with_input_alias=True)
                        y = self.conv1(xb, gain=math.sqrt(0.5), prefiltered=True)
                        x = sk(xs_, gain=math.sqrt(0.5), residual=y)
                else:
                    y = self.conv1(self.conv0(x), gain=math.sqrt(0.5))
                    x = sk(x, gain=math.sqrt(0.5), residual=y)
            else:   # the sum in the strided 3x3 kernel's store instead (one atomic add per element into the skip branch's result)
                y = self.skip(x, gain=math.sqrt(0.5))
                x = self.conv1(self.conv0(x), gain=math.sqrt(0.5), residual=y)
        else:
            x = self.conv1(self.conv0(x))
        assert x.dtype == dtype
        return x, img


_mbstd_tls = threading.local()


@contextlib.contextmanager
def minibatch_std_segments(segments):
    """Every MinibatchStdLayer called by THIS thread inside the block treats its batch as `segments` independent batches back to back
    (thread-local: a concurrent evaluation pass through the same discriminator is not affected)."""
    prev = getattr(_mbstd_tls, 'segments', None)
    _mbstd_tls.segments = int(segments)
    try:
        yield
    finally:
        _mbstd_tls.segments = prev


class MinibatchStdLayer(torch.nn.Module):
    """Appends, per group of `group_size` samples, the channel/pixel-averaged std over the group as extra feature map(s)."""

    def __init__(self, group_size, num_channels=1):
        super().__init__()
        self.group_size, self.num_channels = group_size, num_channels
        self.segments = 1   # > 1: the batch is that many independent batches back to back; groups never straddle them.  A pass that needs it for ONE call
                            # (loss.py: generated + real clips in one discriminator pass) uses `minibatch_std_segments`, which is thread-local

    def forward(self, x):
        n, c, h, w = x.shape
        f = self.num_channels
        seg = getattr(_mbstd_tls, 'segments', None) or getattr(self, 'segments', 1)     # (modules unpickled from before the attribute existed)
        s = seg if seg > 1 and n % seg == 0 else 1
        if s > 1:
            ns = n // s
            g = min(self.group_size, ns) if self.group_size is not None else ns
            y = x.reshape(s, g, -1, f, c // f, h, w)       # [S, G, ns/G, F, c, H, W]: sample s*ns + r*(ns/G) + m is member r of group m of segment s, as below
            y = y - y.mean(dim=1, keepdim=True)
            y = (y.square().mean(dim=1) + 1e-8).sqrt()     # [S, ns/G, F, c, H, W]
            y = y.mean(dim=[3, 4, 5]).reshape(s, -1, f, 1, 1)
            return torch.cat([x, y.repeat(1, g, 1, h, w).reshape(n, f, h, w)], dim=1)
        g = min(self.group_size, n) if self.group_size is not None else n
        y = x.reshape(g, -1, f, c // f, h, w)          # [G, n/G, F, c, H, W]
        y = y - y.mean(dim=0)
        y = (y.square().mean(dim=0) + 1e-8).sqrt()     # std over the group
        y = y.mean(dim=[2, 3, 4]).reshape(-1, f, 1, 1)  # [n/G, F, 1, 1]
        return torch.cat([x, y.repeat(g, 1, h, w)], dim=1)


class DiscriminatorEpilogue(torch.nn.Module):
    def __init__(self, in_channels, cmap_dim, resolution, img_channels, architecture='resnet', mbstd_group_size=4, mbstd_num_channels=1,
                 activation='lrelu', conv_clamp=None, cfg=None):
        assert architecture in ('orig', 'skip', 'resnet')
        super().__init__()
        self.cfg = cfg
        self.in_channels, self.cmap_dim, self.resolution, self.img_channels, self.architecture = in_channels, cmap_dim, resolution, img_channels, architecture
        if architecture == 'skip':
            self.fromrgb = Conv2dLayer(img_channels, in_channels, kernel_size=1, activation=activation)
        self.mbstd = MinibatchStdLayer(group_size=mbstd_group_size, num_channels=mbstd_num_channels) if mbstd_num_channels > 0 else None
        self.conv = Conv2dLayer(in_channels + mbstd_num_channels, in_channels, kernel_size=3, activation=activation, conv_clamp=conv_clamp)
        self.fc = FullyConnectedLayer(in_channels * resolution ** 2, in_channels, activation=activation)
        self.out = FullyConnectedLayer(in_channels, 1 if cmap_dim == 0 else cmap_dim)

    def forward(self, x, img, cmap, force_fp32=False):
        misc.assert_shape(x, [None, self.in_channels, self.resolution, self.resolution])
        x = x.to(dtype=torch.float32, memory_format=torch.contiguous_format)
        if self.architecture == 'skip':
            misc.assert_shape(img, [None, self.img_channels, self.resolution, self.resolution])
            x = x + self.fromrgb(img.to(dtype=torch.float32, memory_format=torch.contiguous_format))
        if self.mbstd is not None:
            x = self.mbstd(x)
        x = self.out(self.fc(self.conv(x).flatten(1)))
        if self.cmap_dim > 0:  # projection discriminator on the conditioning embedding
            misc.assert_shape(cmap, [None, self.cmap_dim])
            x = (x * cmap).sum(dim=1, keepdim=True) * (1 / math.sqrt(self.cmap_dim))
        return x


class Discriminator(torch.nn.Module):
    def __init__(self, c_dim, img_resolution, img_channels, architecture='resnet', channel_base=32768, channel_max=512, num_fp16_res=0,
                 conv_clamp=None, cmap_dim=None, block_kwargs=None, mapping_kwargs=None, epilogue_kwargs=None, cfg=None):
        super().__init__()
        self.cfg = cfg
        self.c_dim, self.img_resolution, self.img_channels = c_dim, img_resolution, img_channels
        self.img_resolution_log2 = int(np.log2(img_resolution))
        self.block_resolutions = [2 ** i for i in range(self.img_resolution_log2, 2, -1)]
        channels = {res: min(channel_base // res, channel_max) for res in self.block_resolutions + [4]}
        fp16_resolution = max(2 ** (self.img_resolution_log2 + 1 - num_fp16_res), 8)
        frames = cfg.sampling.num_frames_per_video
        if cmap_dim is None:
            cmap_dim = channels[4]
        self.time_encoder = TemporalDifferenceEncoder(cfg) if frames > 1 else None
        if self.time_encoder is not None:
            assert self.time_encoder.get_dim() > 0
        if c_dim == 0 and self.time_encoder is None:
            cmap_dim = 0
        total_c_dim = c_dim + (0 if self.time_encoder is None else self.time_encoder.get_dim())
        common = dict(img_channels=img_channels, architecture=architecture, conv_clamp=conv_clamp)
        layer_idx = 0
        for res in self.block_resolutions:
            in_ch = channels[res] if res < img_resolution else 0
            out_ch = channels[res // 2]
            # frames are concatenated along channels at `concat_res`: halve the producer's width, widen the consumer
            if res // 2 == cfg.concat_res:
                out_ch //= cfg.num_frames_div_factor
            if res == cfg.concat_res:
                in_ch = (in_ch // cfg.num_frames_div_factor) * frames
            block = DiscriminatorBlock(in_ch, channels[res], out_ch, resolution=res, first_layer_idx=layer_idx, use_fp16=(res >= fp16_resolution),
                                       cfg=cfg, **(block_kwargs or {}), **common)
            setattr(self, f'b{res}', block)
            layer_idx += block.num_layers
        if c_dim > 0 or self.time_encoder is not None:
            self.mapping = MappingNetwork(z_dim=0, c_dim=total_c_dim, w_dim=cmap_dim, num_ws=None, w_avg_beta=None, **(mapping_kwargs or {}))
        self.b4 = DiscriminatorEpilogue(channels[4], cmap_dim=cmap_dim, resolution=4, cfg=cfg, **(epilogue_kwargs or {}), **common)

    def forward(self, img, c, t, **block_kwargs):
        frames = self.cfg.sampling.num_frames_per_video
        assert len(img) == t.shape[0] * t.shape[1], f'Wrong shape: {img.shape}, {t.shape}'
        assert t.ndim == 2, f'Wrong shape: {t.shape}'
        if self.time_encoder is not None:
            c = torch.cat([c, self.time_encoder(t.reshape(-1, frames))], dim=1)
            if self.cfg.dummy_c:
                c = c * 0.0
        x = None
        with eqlr.batched(self, Conv2dLayer):   # the equalised-lr products of all convolution layers as one launch (and one for their gradients)
            for res in self.block_resolutions:
                if res == self.cfg.concat_res:
                    x = x.reshape(-1, frames * x.shape[1], *x.shape[2:])  # [B*F, C, h, w] -> [B, F*C, h, w]
                x, img = getattr(self, f'b{res}')(x, img, **block_kwargs)
            cmap = self.mapping(None, c) if c.shape[1] > 0 else None
            return {'image_logits': self.b4(x, img, cmap).squeeze(1)}
