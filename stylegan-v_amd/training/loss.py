"""Non-saturating logistic GAN loss with R1 and (optional) path-length regularisation.

Mirror of the reference's ``StyleGAN2Loss`` (src/training/loss.py:25-173): same phases
(``Gmain``, ``Greg``, ``Gboth``, ``Dmain``, ``Dreg``, ``Dboth``), same gradient-synchronisation
gating through ``misc.ddp_sync``, same video-consistent augmentation view.  Statistics reporting
(``training_stats``) is observability and out of scope: losses are returned to the caller instead.

Reference behaviours kept on purpose: path-length regularisation only works with one
frame per video -- with F > 1 the reference's penalty shapes do not broadcast (loss.py:117) -- and the
StyleGAN-V config disables it (``pl_weight: 0``).  The same ``RuntimeError`` surfaces here.
"""

import contextlib
import os

import numpy as np
import torch
import torch.nn.functional as F

from ..torch_utils import misc
from ..torch_utils.ops import conv2d_gradfix, fused_conv_act


class StyleGAN2Loss:
    def __init__(self, cfg, device, G_mapping, G_synthesis, D, augment_pipe=None, style_mixing_prob=0.0, r1_gamma=10,
                 pl_batch_shrink=2, pl_decay=0.01, pl_weight=0.0):
        self.cfg, self.device = cfg, device
        self.G_mapping, self.G_synthesis, self.D = G_mapping, G_synthesis, D
        self.augment_pipe = augment_pipe
        self.style_mixing_prob, self.r1_gamma = style_mixing_prob, r1_gamma
        self.pl_batch_shrink, self.pl_decay, self.pl_weight = pl_batch_shrink, pl_decay, pl_weight
        self.pl_mean = torch.zeros([], device=device)
        # same key, same place, same default as the reference (loss.py:58: cfg.model.loss_kwargs.get('video_consistent_aug', False))
        loss_kwargs = getattr(getattr(cfg, 'model', None), 'loss_kwargs', None)
        self.video_consistent_aug = bool(loss_kwargs.get('video_consistent_aug', False)) if hasattr(loss_kwargs, 'get') else False
        self.frames = cfg.sampling.num_frames_per_video
        # `d_concat` (on by default since round 4; SGV_D_CONCAT=0 restores the reference's two passes): the Dmain phase runs the discriminator ONCE on
        # [generated clips, real clips] instead of twice (src/training/loss.py:122-151) --
        # the same two loss terms, the same gradients (their sum), half the launches of the phase, twice the batch for the small-resolution layers, and no
        # second accumulation pass over every parameter gradient.  The only cross-sample operation of D, the minibatch-std layer, keeps its groups inside
        # each half (networks.minibatch_std_segments).  Equivalence: tests/test_dmain_concat.py (CPU), tests/test_ddp_gloo.py (under DDP); the whole -m gpu
        # suite ran with it (tools/gpu_recipes/r04_call6.sh).  Measured: 8 videos/GPU with graphs +7.4 %, 32 videos/GPU +2.4 % (profiles/r03_d_concat_ab.log).
        # Cost: about twice the peak activation memory of the phase (see the fallback below); with aug=ada the augmentation parameters of both halves come
        # from ONE draw of 2B samples, so the random stream differs from the two-pass schedule's (same distributions; AugmentPipe.host_params also draws on
        # the host generator when the step is eager and on the device generator under capture: eager and captured runs of one seed are not sample-identical).
        self.d_concat = os.environ.get('SGV_D_CONCAT', '1') != '0'

    def run_G(self, z, c, t, sync):
        with misc.ddp_sync(self.G_mapping, sync):
            ws = self.G_mapping(z, c)
            if self.style_mixing_prob > 0:
                cutoff = torch.empty([], dtype=torch.int64, device=ws.device).random_(1, ws.shape[1])
                cutoff = torch.where(torch.rand([], device=ws.device) < self.style_mixing_prob, cutoff, torch.full_like(cutoff, ws.shape[1]))
                ws[:, cutoff:] = self.G_mapping(torch.randn_like(z), c, skip_w_avg_update=True)[:, cutoff:]
        with misc.ddp_sync(self.G_synthesis, sync):
            img = self.G_synthesis(ws, t=t, c=c)
        return img, ws

    @staticmethod
    def _mbstd_segments(segments):
        from .networks import minibatch_std_segments
        return minibatch_std_segments(segments)     # thread-local: scoped to this pass of this thread

    def run_D(self, img, c, t, sync):
        if self.augment_pipe is not None:
            if self.video_consistent_aug:  # one transform per video: fold the frames into channels
                nf, ch, h, w = img.shape
                img = self.augment_pipe(img.reshape(nf // self.frames, self.frames * ch, h, w)).reshape(nf, ch, h, w)
            else:
                img = self.augment_pipe(img)
        with misc.ddp_sync(self.D, sync):
            return self.D(img, c, t)

    def accumulate_gradients(self, phase, real_img, real_c, real_t, gen_z, gen_c, gen_t, sync, gain):
        """Runs forward+backward of one phase, accumulating into .grad.  Returns a dict of scalar losses."""
        assert phase in ('Gmain', 'Greg', 'Gboth', 'Dmain', 'Dreg', 'Dboth')
        do_Gmain = phase in ('Gmain', 'Gboth')
        do_Dmain = phase in ('Dmain', 'Dboth')
        do_Gpl = phase in ('Greg', 'Gboth') and self.pl_weight != 0
        do_Dr1 = phase in ('Dreg', 'Dboth') and self.r1_gamma != 0
        out = {}
        real_img = real_img.reshape(-1, *real_img.shape[2:])  # [B, F, C, H, W] -> [B*F, C, H, W]

        if do_Gmain:  # maximise logits of generated clips
            gen_img, _ = self.run_G(gen_z, gen_c, gen_t, sync=(sync and not do_Gpl))
            logits = self.run_D(gen_img, gen_c, gen_t, sync=False)['image_logits']
            loss = F.softplus(-logits)
            loss.mean().mul(gain).backward()
            out['G/loss'] = loss.detach().mean()

        if do_Gpl:  # path-length regularisation (second-order through G)
            bs = gen_z.shape[0] // self.pl_batch_shrink
            with fused_conv_act.composition_only():   # differentiated twice below
                gen_img, gen_ws = self.run_G(gen_z[:bs], gen_c[:bs], gen_t[:bs], sync=sync)
            pl_noise = torch.randn_like(gen_img) / np.sqrt(gen_img.shape[2] * gen_img.shape[3])
            with conv2d_gradfix.no_weight_gradients():
                (pl_grads,) = torch.autograd.grad(outputs=[(gen_img * pl_noise).sum()], inputs=[gen_ws], create_graph=True, only_inputs=True)
            pl_lengths = pl_grads.square().sum(2).mean(1).sqrt()
            pl_mean = self.pl_mean.lerp(pl_lengths.mean(), self.pl_decay)
            self.pl_mean.copy_(pl_mean.detach())
            loss_pl = (pl_lengths - pl_mean).square() * self.pl_weight
            (gen_img[:, 0, 0, 0] * 0 + loss_pl).mean().mul(gain).backward()
            out['G/reg'] = loss_pl.detach().mean()

        if do_Dmain and not do_Dr1 and self.d_concat and len(gen_z) == len(real_c):
            with torch.no_grad():
                gen_img, _ = self.run_G(gen_z, gen_c, gen_t, sync=False)
            # One pass holds the autograd state of BOTH halves at once -- about twice the peak activation memory of the reference's schedule (loss.py:122-151
            # runs and differentiates the generated half before the real half exists).  A configuration that fits under two passes must not fail under one:
            # an out-of-memory error in the forward pass (nothing has been accumulated yet) switches this loss back to two passes for good.
            #  The cache is emptied BEHIND the except block -- inside it the exception's traceback still holds the failed pass's activations.  Only the
            # forward pass is guarded: an out-of-memory error in the backward pass of the doubled batch has already accumulated part of the gradients and propagates.
            # The switch is per rank (no collective in a failure path): ranks then differ in the minibatch-std grouping of this phase, not in collective counts.
            logits, oom = None, False
            try:
                with self._mbstd_segments(2):
                    logits = self.run_D(torch.cat([gen_img, real_img.detach()]), torch.cat([gen_c, real_c]), torch.cat([gen_t, real_t]), sync=sync)['image_logits']
            except torch.cuda.OutOfMemoryError:
                if torch.cuda.is_current_stream_capturing():
                    raise
                oom = True
            if oom:
                logits = None
                self.d_concat = False
                torch.cuda.empty_cache()
                print('[sgv] Dmain as one discriminator pass over [generated, real] ran out of memory: two passes from here on (SGV_D_CONCAT=0)', flush=True)
            if logits is not None:
                logits_gen, logits_real = logits[:len(gen_z)], logits[len(gen_z):]
                loss_Dgen, loss_Dreal = F.softplus(logits_gen), F.softplus(-logits_real)
                out['signs_real'] = logits_real.detach().sign().mean()
                out['D/loss'] = (loss_Dgen + loss_Dreal).detach().mean()
                (loss_Dgen.mean() + loss_Dreal.mean()).mul(gain).backward()
                return out

        loss_Dgen = 0
        if do_Dmain:  # minimise logits of generated clips (G frozen)
            with torch.no_grad():
                gen_img, _ = self.run_G(gen_z, gen_c, gen_t, sync=False)
            logits = self.run_D(gen_img, gen_c, gen_t, sync=False)['image_logits']  # synced by the real pass below
            loss_Dgen = F.softplus(logits)
            loss_Dgen.mean().mul(gain).backward()

        if do_Dmain or do_Dr1:  # maximise logits of real clips and/or R1 penalty on them
            real_tmp = real_img.detach().requires_grad_(do_Dr1)
            with (fused_conv_act.composition_only() if do_Dr1 else contextlib.nullcontext()):   # R1 differentiates this pass twice
                logits = self.run_D(real_tmp, real_c, real_t, sync=sync)['image_logits']
            loss_Dreal = 0
            out['signs_real'] = logits.detach().sign().mean()   # what the reference reports as 'Loss/signs/real' (loss.py:147): the ADA feedback signal
            if do_Dmain:
                loss_Dreal = F.softplus(-logits)
                out['D/loss'] = (loss_Dgen + loss_Dreal).detach().mean()
            loss_r1 = 0
            if do_Dr1:
                with conv2d_gradfix.no_weight_gradients():
                    (r1_grads,) = torch.autograd.grad(outputs=[logits.sum()], inputs=[real_tmp], create_graph=True, only_inputs=True)
                r1_penalty = r1_grads.square().sum([1, 2, 3])
                loss_r1 = (r1_penalty * (self.r1_gamma / 2)).reshape(-1, len(real_tmp) // len(logits)).mean(dim=1)  # per video
                out['D/reg'] = loss_r1.detach().mean()
                out['r1_penalty'] = r1_penalty.detach().mean()
            (logits * 0 + loss_Dreal + loss_r1).mean().mul(gain).backward()
        return out
