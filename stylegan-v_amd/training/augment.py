"""Adaptive discriminator augmentation: the `bgc` pipeline (pixel blitting + general geometric + colour transforms) the reference
trains with by default (`aug=ada`, `augpipe=bgc`: src/train.py:44,238-277).

Function mirrored: ``AugmentPipe.forward`` of src/training/augment.py:172-375 -- same parameter distributions, same order of
composition, same execution: ONE inverse homogeneous 2-D transform per sample (x-flip, 90-degree rotation, integer translation,
isotropic scale, pre-rotation, anisotropic scale, post-rotation, fractional translation; :185-268) applied as reflect-pad ->
2x up-sampling with the 12-tap `sym6` low-pass -> bilinear resampling -> 2x down-sampling (:270-300), then ONE 4x4 colour matrix
per sample (brightness, contrast, luma flip, hue, saturation; :306-368).  Like the reference, the geometric path runs whenever any
geometric augmentation is configured, even at p = 0.  The image-space filter / noise / cutout stages of the larger
pipelines (`bgcf...`) are not part of `bgc` and are not built.

Organisation (not the reference's): every augmentation is one row of a table -- (probability multiplier, how to draw its parameter,
how the parameter becomes a matrix) -- and ``forward`` folds the rows; the resampling step goes through ``affine_resample`` below,
which on the GPU is a hand-written gather kernel with its adjoint (csrc/resample.hip) instead of affine_grid + grid_sample, and stays
differentiable to any order (R1 runs through this path on real images).

Where the parameters live (round 4): the per-sample parameters are a few hundred bytes, but drawing and composing them on the device costs ~260 launches per call
(`torch.full` per constant matrix entry, a `bmm` per composition, one RNG launch per draw: 790 launches and 2.6 ms of device time per training iteration,
profiles/r04_c11_*), and the padding margin has to come back to the host for the tensor sizes.  With ``host_params`` (default outside hipGraph capture) the same
table is folded on the HOST with CPU tensors -- same distributions, same order of composition, the margin is a host value, no device -> host read -- and the
three results (theta [n,2,3], colour matrix [n,3,3] + offset [n,3]) go to the device in pinned, non-blocking copies.  ``p`` stays a device buffer that ADA adapts on
the device; its host mirror is refreshed only when the buffer has changed (once per `ada_interval` iterations: the one read the reference's loop also does,
training_loop.py:407-410).  Under capture (``static_margin``) everything stays on the device as before.
"""

import math

import numpy as np
import torch

from ..torch_utils.ops import pointwise, resample, upfirdn2d

SYM6 = [0.015404109327027373, 0.0034907120842174702, -0.11799011114819057, -0.048311742585633, 0.4910559419267466, 0.787641141030194,
        0.3379294217276218, -0.07263752278646252, -0.021060292512300564, 0.04472490177066578, 0.0017677118642428036, -0.007800708325034148]

BGC = dict(xflip=1, rotate90=1, xint=1, scale=1, rotate=1, aniso=1, xfrac=1, brightness=1, contrast=1, lumaflip=1, hue=1, saturation=1)


def _mat(rows, like):
    """[B, r, c] matrix from a nested list whose entries are python numbers or [B] tensors."""
    flat = [e for row in rows for e in row]
    ref = next((e for e in flat if isinstance(e, torch.Tensor)), None)
    b = ref.shape[0] if ref is not None else 1
    cols = [e if isinstance(e, torch.Tensor) else torch.full([b], float(e), device=like.device, dtype=torch.float32) for e in flat]
    return torch.stack(cols, dim=-1).reshape(b, len(rows), len(rows[0]))


def _translate(tx, ty, like):
    return _mat([[1, 0, tx], [0, 1, ty], [0, 0, 1]], like)


def _scale(sx, sy, like):
    return _mat([[sx, 0, 0], [0, sy, 0], [0, 0, 1]], like)


def _rotate(theta, like):
    c, s = torch.cos(theta), torch.sin(theta)
    return _mat([[c, -s, 0], [s, c, 0], [0, 0, 1]], like)


class AugmentPipe(torch.nn.Module):
    def __init__(self, xflip=0, rotate90=0, xint=0, xint_max=0.125, scale=0, rotate=0, aniso=0, xfrac=0, scale_std=0.2, rotate_max=1,
                 aniso_std=0.2, xfrac_std=0.125, brightness=0, contrast=0, lumaflip=0, hue=0, saturation=0, brightness_std=0.2,
                 contrast_std=0.5, hue_max=1, saturation_std=1, imgfilter=0, noise=0, cutout=0, **_unused):
        super().__init__()
        if imgfilter or noise or cutout:
            raise NotImplementedError('image-space filtering / noise / cutout are outside the bgc pipeline built here')
        self.register_buffer('p', torch.ones([]))   # overall probability multiplier, adapted by ADA (training_loop.py:407-410)
        self.mult = dict(xflip=float(xflip), rotate90=float(rotate90), xint=float(xint), scale=float(scale), rotate=float(rotate), aniso=float(aniso),
                         xfrac=float(xfrac), brightness=float(brightness), contrast=float(contrast), lumaflip=float(lumaflip), hue=float(hue),
                         saturation=float(saturation))
        self.xint_max, self.scale_std, self.rotate_max, self.aniso_std, self.xfrac_std = xint_max, scale_std, rotate_max, aniso_std, xfrac_std
        self.brightness_std, self.contrast_std, self.hue_max, self.saturation_std = brightness_std, contrast_std, hue_max, saturation_std
        self.register_buffer('Hz_geom', upfirdn2d.setup_filter(SYM6))
        self._hz_taps = self.Hz_geom.tolist()
        # True: reflect-pad by the worst-case margin (w - 1, h - 1: what the margin below is clamped to anyway) instead of reading the batch's
        # margin back from the device.  Same result (extra padding is never sampled), no device -> host sync, static shapes: what hipGraph
        # capture needs.  The price is a padded image of (3w - 2) x (3h - 2) instead of typically (w + 12) x (h + 12).
        self.static_margin = False
        # True: draw / compose the per-sample parameters on the host and upload the results (module docstring).  Only takes effect for CUDA images outside capture.
        self.host_params = True
        self.fused_geometric = True   # the geometric execution as one kernel per direction (ops/resample.py ada_geometric)
        self._p_host = None      # (id of the buffer's storage, its version, 0-d CPU tensor)
        self._slots, self._slot_phase, self._slot_cursor = {}, None, 0      # parameter slots of captured phases (begin_phase)
        self._const = {}

    def _c(self, key, device, make):
        """Small constant tensors, created once per device (not inside every forward: a host -> device copy is illegal under stream capture)."""
        k = (key, str(device))
        if k not in self._const:
            self._const[k] = make().to(device)
        return self._const[k]

    def _param_device(self, images):
        if self.host_params and images.is_cuda and not self.static_margin and not torch.cuda.is_current_stream_capturing():
            return torch.device('cpu')
        return images.device

    def _p_on(self, device):
        """`p` where the parameters are drawn: the buffer itself, or its host mirror (re-read only after the buffer changed)."""
        if self.p.device == device:
            return self.p
        key = (self.p.data_ptr(), self.p._version)
        if self._p_host is None or self._p_host[0] != key:
            self._p_host = (key, self.p.detach().to(device))       # one device -> host read per change of p
        return self._p_host[1]

    @staticmethod
    def _upload(t, device):
        """A small host tensor to the device without blocking the host (pinned staging block from the caching host allocator)."""
        if t.device == device:
            return t
        buf = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        buf.copy_(t)
        return buf.to(device, non_blocking=True)

    # -- parameter draws ------------------------------------------------------------------------------------------------------
    def _draw(self, name, shape, kind, neutral, device, pct, prob=None):
        """One augmentation parameter per sample: drawn from `kind`, replaced by `neutral` with probability 1 - mult * p
        (or 1 - prob); with `pct` (the reference's debug_percentile) the draw is the given percentile for every sample."""
        if kind == 'flip':        # uniform over {0, 1}
            v = torch.floor(torch.rand(shape, device=device) * 2) if pct is None else torch.full(shape, math.floor(pct * 2), device=device, dtype=torch.float32)
        elif kind == 'quarter':   # uniform over {0, 1, 2, 3}
            v = torch.floor(torch.rand(shape, device=device) * 4) if pct is None else torch.full(shape, math.floor(pct * 4), device=device, dtype=torch.float32)
        elif kind == 'uniform':   # uniform over [-1, 1]
            v = torch.rand(shape, device=device) * 2 - 1 if pct is None else torch.full(shape, pct * 2 - 1, device=device, dtype=torch.float32)
        else:                     # standard normal
            v = torch.randn(shape, device=device) if pct is None else torch.full(shape, float(torch.erfinv(torch.tensor(pct * 2 - 1.0))), device=device)
        if pct is not None:
            return v
        gate_shape = [shape[0]] + [1] * (len(shape) - 1)
        keep = torch.rand(gate_shape, device=device) < (self.mult[name] * self._p_on(device) if prob is None else prob)
        return torch.where(keep, v, torch.full_like(v, neutral))

    def forward(self, images, debug_percentile=None):
        assert isinstance(images, torch.Tensor) and images.ndim == 4
        n, ch, h, w = images.shape
        prm = self._take_slot(images, debug_percentile)
        if prm is None:
            prm = self._fold_parameters(n, ch, h, w, self._param_device(images), debug_percentile)
            prm = {k: (self._upload(v, images.device) if isinstance(v, torch.Tensor) else v) for k, v in prm.items()}
        if prm['theta'] is not None:
            images = self._execute(images, prm['theta'], prm['margin'])
        if prm['colour'] == 'rgb':
            # RGB, or F frames of one video folded into 3F channels (loss.py:58-66): the same matrix for every frame
            f, cw, cb = ch // 3, prm['cw'], prm['cb']
            if images.is_cuda and images.dtype == torch.float32:
                # 3 -> 3 channels with per-sample weights: the streaming kernel ToRGB uses (csrc/pointwise.hip) instead of a batched 3x3 GEMM
                # (rocBLAS: 0.42 ms per call on 150 MB, 10x its HBM time; profiles/r04_c8_ada_step_kernel_stats.csv)
                y = pointwise.pointwise_conv(images.reshape(n * f, 3, h, w), cw)
                flat = (y + cb.reshape(n * f, 3, 1, 1)) if y.requires_grad else y.add_(cb.reshape(n * f, 3, 1, 1))
            else:
                flat = cw @ images.reshape(n * f, 3, h * w) + cb
            images = flat.reshape(n, ch, h, w)
        elif prm['colour'] == 'l':
            cm = prm['cw']
            images = (images.reshape(n, ch, h * w) * cm[:, :, :3].sum(dim=2, keepdim=True) + cm[:, :, 3:]).reshape(n, ch, h, w)
        return images

    # -- parameter slots: what a captured phase reads instead of drawing ----------------------------------------------------------------
    def begin_phase(self, name):
        """A captured phase is about to run (its eager warm-up, its capture or a replay).  Every call of the pipe inside the phase takes its parameters from
        persistent device tensors -- one "slot" per call, in call order -- which are filled HERE from a host-side draw and pinned, non-blocking copies on the current
        stream: the replayed graph holds the two image kernels of a call and none of the ~260 launches of the device-side parameter table (790 per iteration,
        13 ms of a 153 ms captured iteration: profiles/r06_c35_ada_in_step.txt).  Same distributions, same order of composition; `p` through its host mirror."""
        self._slot_phase, self._slot_cursor = name, 0
        for key, slot in self._slots.items():
            if key[0] == name:
                self._fill_slot(slot)

    def rewind(self):
        """The phase's function runs again from its first call (warm-up passes, then the capture)."""
        self._slot_cursor = 0

    def end_phase(self):
        self._slot_phase = None

    def _take_slot(self, images, pct):
        if self._slot_phase is None or pct is not None or images.shape[1] % 3 != 0:
            return None
        key = (self._slot_phase, self._slot_cursor)
        self._slot_cursor += 1
        slot = self._slots.get(key)
        if slot is None:
            if images.is_cuda and torch.cuda.is_current_stream_capturing():
                raise RuntimeError('AugmentPipe: call %d of phase %r first seen under stream capture (the eager warm-up passes create the slots)' % key[::-1])
            n, ch, h, w = images.shape
            host = self._fold_parameters(n, ch, h, w, torch.device('cpu'), None)
            slot = self._slots[key] = dict(shape=tuple(images.shape), margin=host['margin'], colour=host['colour'],
                                           **{k: (None if host[k] is None else torch.empty(host[k].shape, dtype=host[k].dtype, device=images.device)) for k in ('theta', 'cw', 'cb')})
            self._fill_slot(slot, host)
        if slot['shape'] != tuple(images.shape):
            raise RuntimeError('AugmentPipe: call %d of phase %r changed shape: %s, captured with %s' % (key[1], key[0], tuple(images.shape), slot['shape']))
        return slot

    def _fill_slot(self, slot, host=None):
        if host is None:
            host = self._fold_parameters(*slot['shape'], torch.device('cpu'), None)
        assert host['margin'] == slot['margin'] and host['colour'] == slot['colour']
        for k in ('theta', 'cw', 'cb'):
            if slot[k] is not None:
                if slot[k].is_cuda:
                    buf = torch.empty(host[k].shape, dtype=host[k].dtype, pin_memory=True)
                    buf.copy_(host[k])
                    slot[k].copy_(buf, non_blocking=True)
                else:
                    slot[k].copy_(host[k])

    def _fold_parameters(self, n, ch, h, w, dev, pct):
        """The parameter table folded on `dev`: -> dict(theta [n,2,3] | None, margin (mx0, mx1, my0, my1), colour 'rgb' | 'l' | None, cw, cb), tensors on `dev`."""
        on = self.mult
        like = torch.empty(0, device=dev)

        # ---- inverse geometric transform G_inv (maps output pixels to input pixels), composed left to right (augment.py:185-268) ----
        g_inv, geometric = None, False

        def push(m):
            nonlocal g_inv, geometric
            g_inv = m if g_inv is None else g_inv @ m
            geometric = True
        if on['xflip'] > 0:
            i = self._draw('xflip', [n], 'flip', 0, dev, pct)
            push(_scale(1 / (1 - 2 * i), 1, like))
        if on['rotate90'] > 0:
            i = self._draw('rotate90', [n], 'quarter', 0, dev, pct)
            push(_rotate(math.pi / 2 * i, like))                       # inverse of a rotation by -pi/2 * i
        if on['xint'] > 0:
            t = self._draw('xint', [n, 2], 'uniform', 0, dev, pct) * self.xint_max
            push(_translate(-torch.round(t[:, 0] * w), -torch.round(t[:, 1] * h), like))
        if on['scale'] > 0:
            s = torch.exp2(self._draw('scale', [n], 'normal', 0, dev, pct) * self.scale_std)
            push(_scale(1 / s, 1 / s, like))
        p_rot = 1 - torch.sqrt((1 - on['rotate'] * self._p_on(dev)).clamp(0, 1))   # P(pre OR post) = p
        if on['rotate'] > 0:
            th = self._draw('rotate', [n], 'uniform', 0, dev, pct, prob=p_rot) * math.pi * self.rotate_max
            push(_rotate(th, like))                                    # inverse of a rotation by -theta
        if on['aniso'] > 0:
            s = torch.exp2(self._draw('aniso', [n], 'normal', 0, dev, pct) * self.aniso_std)
            push(_scale(1 / s, s, like))
        if on['rotate'] > 0:
            th = self._draw('rotate', [n], 'uniform', 0, dev, None if pct is None else 0.5, prob=p_rot) * math.pi * self.rotate_max   # debug mode: no post-rotation
            push(_rotate(th, like))
        if on['xfrac'] > 0:
            t = self._draw('xfrac', [n, 2], 'normal', 0, dev, pct) * self.xfrac_std
            push(_translate(-t[:, 0] * w, -t[:, 1] * h, like))

        theta, margin = self._theta(g_inv, h, w) if geometric else (None, None)

        # ---- colour transform C (augment.py:306-368) ----
        eye4 = self._c('eye4', dev, lambda: torch.eye(4))
        c_mat, coloured = eye4.unsqueeze(0), False
        vv = self._c('vv', dev, lambda: (torch.tensor([1., 1., 1., 0.]) / math.sqrt(3)).outer(torch.tensor([1., 1., 1., 0.]) / math.sqrt(3)))
        if on['brightness'] > 0:
            b = self._draw('brightness', [n], 'normal', 0, dev, pct) * self.brightness_std
            c_mat, coloured = _mat([[1, 0, 0, b], [0, 1, 0, b], [0, 0, 1, b], [0, 0, 0, 1]], like) @ c_mat, True
        if on['contrast'] > 0:
            c = torch.exp2(self._draw('contrast', [n], 'normal', 0, dev, pct) * self.contrast_std)
            c_mat, coloured = _mat([[c, 0, 0, 0], [0, c, 0, 0], [0, 0, c, 0], [0, 0, 0, 1]], like) @ c_mat, True
        if on['lumaflip'] > 0:
            i = self._draw('lumaflip', [n, 1, 1], 'flip', 0, dev, pct)
            c_mat, coloured = (eye4 - 2 * vv * i) @ c_mat, True          # Householder reflection about the luma axis
        if on['hue'] > 0 and ch > 1:
            th = self._draw('hue', [n], 'uniform', 0, dev, pct) * math.pi * self.hue_max
            k = self._c('cross', dev, lambda: torch.tensor([[0., -1, 1, 0], [1, 0, -1, 0], [-1, 1, 0, 0], [0, 0, 0, 0]]) / math.sqrt(3))   # cross-product matrix of the luma axis
            cth, sth = torch.cos(th).reshape(n, 1, 1), torch.sin(th).reshape(n, 1, 1)
            rot = vv * (1 - cth) + self._c('d1110', dev, lambda: torch.diag(torch.tensor([1., 1., 1., 0.]))) * cth + k * sth     # Rodrigues about v
            rot = rot + self._c('d0001', dev, lambda: torch.diag(torch.tensor([0., 0., 0., 1.])))
            c_mat, coloured = rot @ c_mat, True
        if on['saturation'] > 0 and ch > 1:
            s = torch.exp2(self._draw('saturation', [n, 1, 1], 'normal', 0, dev, pct) * self.saturation_std)
            c_mat, coloured = (vv + (eye4 - vv) * s) @ c_mat, True
        colour, cw, cb = None, None, None
        if coloured:
            c_mat = c_mat.expand(n, 4, 4)
            if ch % 3 == 0:
                f = ch // 3
                cm = c_mat.repeat_interleave(f, dim=0) if f > 1 else c_mat
                colour, cw, cb = 'rgb', cm[:, :3, :3].contiguous(), cm[:, :3, 3:].contiguous()
            elif ch == 1:
                colour, cw = 'l', c_mat[:, :3, :].mean(dim=1, keepdim=True).contiguous()
            else:
                raise ValueError('Image must be RGB (3 channels) or L (1 channel)')
        return dict(theta=theta, margin=margin, colour=colour, cw=cw, cb=cb)

    # -- geometric execution (augment.py:270-300) -----------------------------------------------------------------------------------
    def _theta(self, g_inv, h, w):
        """Margin of the reflect padding and the map the resampling step gets (augment.py:272-296), on the parameters' device (the host with `host_params` and for
        the slots of a captured phase: the margin read below is free there)."""
        dev = g_inv.device
        cx, cy = (w - 1) / 2, (h - 1) / 2
        pad = self.Hz_geom.shape[0] // 4
        if self.static_margin:
            mx0, my0, mx1, my1 = w - 1, h - 1, w - 1, h - 1                   # the clamp bound of the reference's margin (:281-282)
        else:
            corners = self._c(('corners', w, h), dev, lambda: torch.tensor([[-cx, -cy, 1], [cx, -cy, 1], [cx, cy, 1], [-cx, cy, 1]], dtype=torch.float32))
            cp = g_inv @ corners.t()                                          # [n, 3, 4]: where the output corners come from
            ext = cp[:, :2, :].permute(1, 0, 2).flatten(1)                    # [2, n * 4]
            margin = torch.cat([-ext, ext]).max(dim=1).values                 # [x0, y0, x1, y1]
            margin = margin + self._c(('moff', w, h, pad), dev, lambda: torch.tensor([pad * 2 - cx, pad * 2 - cy] * 2, dtype=torch.float32))
            margin = margin.clamp(min=0).minimum(self._c(('mmax', w, h), dev, lambda: torch.tensor([w - 1, h - 1] * 2, dtype=torch.float32)))
            mx0, my0, mx1, my1 = (int(v) for v in margin.ceil().tolist())     # one device -> host read per call, as in the reference (:283)
        # the padded, then 2x up-sampled image -> the (size + 2 pad) * 2 resampled image
        wu, hu = (w + mx0 + mx1) * 2, (h + my0 + my1) * 2
        g_inv = _translate((mx0 - mx1) / 2, (my0 - my1) / 2, g_inv) @ g_inv
        s2, s2i = _scale(2, 2, g_inv), _scale(0.5, 0.5, g_inv)
        g_inv = s2 @ g_inv @ s2i
        g_inv = _translate(-0.5, -0.5, g_inv) @ g_inv @ _translate(0.5, 0.5, g_inv)
        out_h, out_w = (h + pad * 2) * 2, (w + pad * 2) * 2
        g_inv = _scale(2 / wu, 2 / hu, g_inv) @ g_inv @ _scale(out_w / 2, out_h / 2, g_inv)
        return g_inv[:, :2, :].contiguous(), (mx0, mx1, my0, my1)

    def _execute(self, images, theta, margin):
        n, ch, h, w = images.shape
        pad = self.Hz_geom.shape[0] // 4
        mx0, mx1, my0, my1 = margin
        if self.fused_geometric and resample.ada_geometric_fused_ok(images, self.Hz_geom):
            # reflect pad -> up -> resample -> down as ONE kernel: no padded / up-sampled / resampled image in memory (csrc/resample.hip); calls that
            # will be differentiated (the generator's phase, R1) get the node whose backward is the block's adjoint as one kernel
            return resample.ada_geometric(images, theta, self.Hz_geom, margin, f_host=self._filter_taps())
        images = torch.nn.functional.pad(images, [mx0, mx1, my0, my1], mode='reflect')
        images = upfirdn2d.upsample2d(images, self.Hz_geom, up=2)
        images = resample.affine_resample(images, theta, ((h + pad * 2) * 2, (w + pad * 2) * 2))
        return upfirdn2d.downsample2d(images, self.Hz_geom, down=2, padding=-pad * 2, flip_filter=True)

    def _resample(self, images, g_inv):
        """Geometric execution from the inverse maps (pixel coordinates about the image centre)."""
        theta, margin = self._theta(g_inv, images.shape[2], images.shape[3])
        return self._execute(images, self._upload(theta, images.device), margin)

    def _filter_taps(self):
        """`Hz_geom` as host floats (launch arguments of the fused kernel); a constant of the pipeline, read at construction."""
        return self._hz_taps


def ada_update(augment_pipe, sign_real_mean, batch_size, interval=4, target=0.6, kimg=500):
    """p <- max(p + sign(E[sign(D(real))] - target) * batch * interval / (kimg * 1000), 0), all on the device (training_loop.py:407-410)."""
    step = torch.sign(sign_real_mean - target) * (batch_size * interval) / (kimg * 1000)
    augment_pipe.p.copy_((augment_pipe.p + step).clamp(min=0))
