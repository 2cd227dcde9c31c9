"""One G+D training iteration of StyleGAN-V on one rank, plus the data-parallel wrapping.

This is the benchmark driver's view of the reference training loop (src/training/training_loop.py):
model construction (:163-165), DDP wrap of G.mapping / G.synthesis / D with ``broadcast_buffers=False``
(:215-232), the lazy-regularisation phase list with rescaled Adam hyper-parameters (:238-252), and the
per-iteration body (:330-404): fresh latents and frame times per phase, zero_grad -> accumulate_gradients
-> nan_to_num on every gradient -> Adam step, then the G_ema update.  Dataset I/O, snapshots, metrics,
logging and the launch machinery of the reference are out of scope; real frames are synthetic
(uniform uint8 noise scaled to [-1, 1]) and frame times are drawn like ``layers.sample_frames``.

Multi-GPU: one process per GPU; the only collective is DDP's bucketed gradient all-reduce over
RCCL/xGMI (gloo on CPU for tests), gated per phase exactly as the reference gates it (loss.py:45-69).
"""

import copy
import math

import numpy as np
import torch

from ..torch_utils import misc
from . import config as cfgs
from . import motion
from .loss import StyleGAN2Loss
from .networks import Discriminator, Generator


class _HipGraph:
    """One captured hipGraph (torch.cuda.CUDAGraph).  `capture(fn)` records fn's launches without executing them and returns fn's result (static
    output tensors); tensors fn allocates -- the gradients a backward pass binds to `p.grad` -- live in the graph's private pool and are rewritten in
    place by every `replay()`."""

    def __init__(self, pool=None):
        self.graph, self._pool = torch.cuda.CUDAGraph(), pool

    capture_hook = None      # optional context-manager factory entered around every capture (bench.py: per-launch event nodes for the kernels it times)

    def capture(self, fn):
        # thread_local: the RCCL watchdog thread of an initialised process group queries events while this thread captures, which the
        # default (global) capture mode treats as an error in the OTHER thread (observed: segmentation fault in capture_end)
        kw = dict(pool=self._pool) if self._pool is not None else {}
        from ..torch_utils.ops import amax as _amax      # (the magnitude-bound slots of the capture come from arenas the graph itself clears: ops/amax.py)
        import contextlib
        hook = _HipGraph.capture_hook() if _HipGraph.capture_hook is not None else contextlib.nullcontext()
        with _amax.capture_scope(), torch.cuda.graph(self.graph, capture_error_mode='thread_local', **kw), hook:
            return fn()

    def pool(self):
        return self.graph.pool()

    def replay(self):
        self.graph.replay()


class _EmulatedGraph:
    """Host-side stand-in for `_HipGraph` (``use_graphs='emulate'``; any device): keeps the one property of a replayed graph that the surrounding
    Python has to get right -- a replay rewrites the tensors that were bound to `p.grad` AT CAPTURE TIME and never touches the `p.grad` attributes
    themselves, whatever eager phases did to them in between (`opt.zero_grad(set_to_none=True)` of a reg phase un-binds or re-binds them).  It is
    what lets the world-size-2 gloo test (tests/test_ddp_gloo.py) run the graph schedule's gradient all-reduce on CPU.
    `capture(fn, executes=True)` runs fn once with the training state put back afterwards (a real capture executes nothing) and remembers the
    gradient tensors fn left bound; `replay()` runs fn again on the side, copies the fresh gradients into those tensors and restores the
    attributes.  A graph captured with `reads_grads_of=<grad graph>` (the update graph) runs fn with `p.grad` bound to that graph's tensors."""

    def __init__(self, params, state_fn, reads_grads_of=None):
        self.params, self.state_fn, self.src = list(params), state_fn, reads_grads_of
        self.fn, self.bufs = None, None

    def capture(self, fn):
        self.fn = fn
        if self.src is not None:          # update graph: nothing to discover
            return None
        saved = [t.clone() for t in self.state_fn()]
        out = fn()
        with torch.no_grad():
            for t, s0 in zip(self.state_fn(), saved):
                t.copy_(s0)
        self.bufs = [p.grad for p in self.params]     # bound like after a real capture
        return out

    def pool(self):
        return None

    def replay(self):
        bound = [p.grad for p in self.params]
        if self.src is not None:
            for p, g in zip(self.params, self.src.bufs):
                p.grad = g
            self.fn()
        else:
            self.out = self.fn()
            with torch.no_grad():
                for p, buf in zip(self.params, self.bufs):
                    if buf is not None:
                        buf.copy_(p.grad)
        for p, g in zip(self.params, bound):
            p.grad = g


def build_models(g_kwargs, d_kwargs, device, seed=0):
    torch.manual_seed(seed)
    G = Generator(**g_kwargs).train().requires_grad_(False).to(device)
    D = Discriminator(**d_kwargs).train().requires_grad_(False).to(device)
    G_ema = copy.deepcopy(G).eval()
    return G, D, G_ema


def sample_frame_times(sampling, batch, generator=None, device='cpu'):
    """Sorted fractional frame positions [batch, F] within a clip of `max_num_frames` frames, distance between first
    and last frame drawn from `total_dists` capped by `max_dist` (layers.random_frame_sampling with use_fractional_t)."""
    nf, total = sampling.num_frames_per_video, sampling.max_num_frames
    hi = min(total - 1, sampling.get('max_dist', 10 ** 9))
    dists = [d for d in (sampling.get('total_dists') or range(nf - 1, hi)) if nf - 1 <= d <= hi] or [max(nf - 1, 1)]
    g = generator
    span = torch.tensor(dists, dtype=torch.float32)[torch.randint(len(dists), [batch], generator=g)]
    offset = torch.rand([batch], generator=g) * (total - span - 1)
    cols = [offset]
    if nf > 1:
        cols.append(offset + span)
    for _ in range(nf - 2):
        cols.append(offset + 1 + torch.rand([batch], generator=g) * (span - 1).clamp(min=0))
    times = torch.stack(cols, dim=1).sort(dim=1).values
    if torch.device(device).type == 'cuda':      # pinned + non-blocking: a pageable upload waits for the stream to drain (a host <-> device sync per phase)
        return times.pin_memory().to(device, non_blocking=True)
    return times.to(device)


class TrainStep:
    """Holds G, D, G_ema, optimisers and the loss; ``step()`` runs one iteration of the phase schedule."""

    def __init__(self, g_kwargs, d_kwargs, train_cfg, device, batch_gpu, world_size=1, rank=0, seed=0, ddp=None, bucket_cap_mb=25, use_graphs=False, augment='noaug',
                 ada_target=0.6, ada_interval=4, ada_kimg=500, ddp_manual=None):
        self.device, self.batch_gpu, self.world_size, self.rank = torch.device(device), batch_gpu, world_size, rank
        self.train_cfg = train_cfg
        self.G, self.D, self.G_ema = build_models(g_kwargs, d_kwargs, device, seed=seed)
        self.sampling = g_kwargs['cfg'].sampling
        self.frames = self.sampling.num_frames_per_video
        self.res, self.img_channels = g_kwargs['img_resolution'], g_kwargs['img_channels']
        self.z_dim = self.G.z_dim
        self.gen = torch.Generator().manual_seed(seed * world_size + rank)  # rank-specific latents (training_loop.py:137-139)
        # the synthetic clips and the latents are drawn ON the device (the reference draws z there, training_loop.py:339; its clips arrive from pinned, prefetched
        # loader batches): a CPU draw + pageable upload of 19 MB per iteration is 59 ms of host time and a host <-> device sync at every iteration's start
        self.dev_gen = torch.Generator(device=device).manual_seed(seed * world_size + rank) if torch.device(device).type == 'cuda' else self.gen
        self.batch_size = batch_gpu * world_size

        # Multi-GPU: DDP wrappers only add the gradient all-reduce; parameters stay shared with the raw modules.  With hipGraph replay the
        # wrappers are left out and the gradients of the phase's module are averaged by ONE flat all-reduce behind its backward pass
        # (`_allreduce_gradients`; same mean as DDP's buckets, not overlapped with the backward -- ~1.5 ms against a 58 ms step): torch's DDP
        # reducer under stream capture crashed `capture_end` on ROCm 7.2 / torch 2.10 even with its synchronisation switched off.
        self.ddp = (world_size > 1) if ddp is None else ddp
        self.emulate_graphs = use_graphs == 'emulate'
        self.use_graphs = self.emulate_graphs or (bool(use_graphs) and self.device.type == 'cuda')
        self.ddp_manual = self.ddp and (self.use_graphs if ddp_manual is None else bool(ddp_manual))
        modules = dict(G_mapping=self.G.mapping, G_synthesis=self.G.synthesis, D=self.D)
        if self.ddp and not self.ddp_manual:
            ids = [self.device.index] if self.device.type == 'cuda' else None
            for name, mod in list(modules.items()):
                mod.requires_grad_(True)
                modules[name] = torch.nn.parallel.DistributedDataParallel(mod, device_ids=ids, broadcast_buffers=False, bucket_cap_mb=bucket_cap_mb)
                mod.requires_grad_(False)
        if self.ddp:
            with torch.no_grad():
                for mod in ((self.G, self.D, self.G_ema) if self.ddp_manual else (self.G_ema,)):   # DDP's constructor broadcasts G / D itself
                    for p in misc.params_and_buffers(mod):
                        torch.distributed.broadcast(p, src=0)
        # Discriminator augmentation (train.py:238-277, training_loop.py:197-205): 'ada' = the bgc pipeline with p starting at 0 and adapted
        # every `ada_interval` iterations from the sign of D's outputs on real clips; one transform per video (configs/model/stylegan-v.yaml:58).
        self.augment_pipe, self.ada = None, None
        self._ada_cfg = dict(target=ada_target, interval=ada_interval, kimg=ada_kimg)
        self.loss = StyleGAN2Loss(cfg=g_kwargs['cfg'], device=self.device, r1_gamma=train_cfg.r1_gamma, pl_weight=train_cfg.pl_weight, **modules)
        self.set_augment(augment)

        # Phase list with lazy regularisation (training_loop.py:238-252): reg every `interval` iterations with
        # lr and betas rescaled by c = interval / (interval + 1).
        self.phases = []
        # torch's fused multi-tensor Adam on the GPU (same update rule as the reference's torch.optim.Adam, training_loop.py:245-251; one launch per
        # parameter group instead of ~12 multi-tensor passes)
        adam_kw = dict(fused=True) if self.device.type == 'cuda' else {}
        if self.emulate_graphs:
            adam_kw = {}
        for name, module, interval in (('G', self.G, train_cfg.G_reg_interval), ('D', self.D, train_cfg.D_reg_interval)):
            if interval is None:
                opt = torch.optim.Adam(module.parameters(), lr=train_cfg.lr, betas=tuple(train_cfg.betas), eps=1e-8, **adam_kw)
                self.phases.append(dict(name=name + 'both', module=module, opt=opt, interval=1))
            else:
                ratio = interval / (interval + 1)
                opt = torch.optim.Adam(module.parameters(), lr=train_cfg.lr * ratio, betas=tuple(b ** ratio for b in train_cfg.betas), eps=1e-8, **adam_kw)
                self.phases.append(dict(name=name + 'main', module=module, opt=opt, interval=1))
                self.phases.append(dict(name=name + 'reg', module=module, opt=opt, interval=interval))
        self.cur_nimg = 0
        self.batch_idx = 0
        self.last_losses = {}
        # Frame times are drawn inside [0, max_num_frames - 1) here (sample_frame_times), so the training passes promise that bound to the
        # motion encoder (motion.frame_times_bounded_by) instead of reading t.max() back from the device.  The promise is scoped to
        # `_run_phase`: G / G_ema used outside of it (evaluation, long-video generation) keep the reference's t.max() behaviour.
        self._t_bound = float(self.sampling.max_num_frames - 1)
        # hipGraph replay of the two every-iteration phases (small per-GPU batches are launch-bound).  Works with DDP (the graph's backward
        # runs un-synchronised, one flat RCCL all-reduce of the gradients follows the replay) and with ADA (the pipe pads by its static
        # worst-case margin while graphs are on, so that nothing is read back to the host).
        self._graphs = {}
        if self.use_graphs and not self.emulate_graphs:
            for phase in self.phases:
                for group in phase['opt'].param_groups:
                    group['capturable'] = True
        if self.use_graphs:
            if self.augment_pipe is not None:
                self.augment_pipe.static_margin = True

    def set_augment(self, augment):
        """'noaug' or 'ada' (see __init__); can be switched on an existing instance (bench.py times both on the same models)."""
        assert augment in ('noaug', 'ada')
        if augment == 'ada':
            from .augment import AugmentPipe, BGC
            self.augment_pipe = AugmentPipe(**BGC).train().requires_grad_(False).to(self.device)
            self.augment_pipe.p.copy_(torch.zeros([]))
            self.ada = dict(self._ada_cfg, acc=torch.zeros([2], device=self.device))
            self.augment_pipe.static_margin = bool(getattr(self, 'use_graphs', False))   # no device -> host read of the padding under capture
            self._graphs = {}         # phases captured without the pipe are stale
        else:
            if getattr(self, 'augment_pipe', None) is not None:
                self._graphs = {}
            self.augment_pipe, self.ada = None, None
        self.loss.augment_pipe = self.augment_pipe
        self.loss.video_consistent_aug = augment == 'ada'

    # -- synthetic inputs -------------------------------------------------------------------------
    def reseed_inputs(self, seed):
        """Restart the streams the latents, frame times and synthetic clips are drawn from (two runs that are to see the same inputs)."""
        self.gen = torch.Generator().manual_seed(seed)
        self.dev_gen = torch.Generator(device=self.device).manual_seed(seed) if self.device.type == 'cuda' else self.gen

    def synthetic_real_batch(self):
        """uint8-like noise frames scaled as training_loop.py:335: [batch_gpu, F, C, H, W] in [-1, 1]."""
        raw = torch.randint(0, 256, [self.batch_gpu, self.frames, self.img_channels, self.res, self.res], generator=self.dev_gen, dtype=torch.uint8, device=self.device)
        return raw.to(torch.float32) / 127.5 - 1

    def _latents(self):
        z = torch.randn([self.batch_gpu, self.z_dim], generator=self.dev_gen, device=self.device)
        c = torch.zeros([self.batch_gpu, 0], device=self.device)
        t = sample_frame_times(self.sampling, self.batch_gpu, generator=self.gen, device=self.device)
        return z, c, t

    # -- one phase: zero_grad -> accumulate_gradients -> nan_to_num -> Adam (training_loop.py:351-389) -----------------
    def _phase_gradients(self, phase, sync, real_img, real_c, real_t, gen_z, gen_c, gen_t):
        if self.augment_pipe is not None:
            self.augment_pipe.rewind()               # (a captured phase's function runs several times -- warm-up, capture -- over the same parameter slots)
        phase['opt'].zero_grad(set_to_none=True)
        phase['module'].requires_grad_(True)
        with motion.frame_times_bounded_by(self._t_bound):
            losses = self.loss.accumulate_gradients(phase=phase['name'], real_img=real_img, real_c=real_c, real_t=real_t, gen_z=gen_z,
                                                    gen_c=gen_c, gen_t=gen_t, sync=sync, gain=phase['interval'])
        phase['module'].requires_grad_(False)
        return losses

    def _phase_update(self, phase):
        grads = [p.grad for p in phase['module'].parameters() if p.grad is not None]
        if grads:
            sanitize_gradients_(grads)
        phase['opt'].step()

    def _run_phase(self, phase, real_img, real_c, real_t, gen_z, gen_c, gen_t):
        losses = self._phase_gradients(phase, True, real_img, real_c, real_t, gen_z, gen_c, gen_t)
        if self.ddp_manual:
            self._allreduce_gradients(phase)
        self._phase_update(phase)
        return losses

    def _allreduce_gradients(self, phase, grads=None):
        """Average the phase module's gradients over the ranks as ONE flat all-reduce (what DDP's buckets do during backward in the eager path).
        Used behind a replayed hipGraph, whose backward pass ran with DDP's own synchronisation off.  `grads`: the tensors to reduce -- a replayed
        phase passes the gradient buffers of its capture (`entry['grads']`): `p.grad` only points at them until the next eager phase on the same
        optimiser re-binds it (Greg / Dreg call `zero_grad(set_to_none=True)`), while the captured update graph keeps reading the buffers."""
        if grads is None:
            grads = [p.grad for p in phase['module'].parameters() if p.grad is not None]
        if not grads:
            return
        flat = torch.cat([g.reshape(-1) for g in grads])
        torch.distributed.all_reduce(flat)
        flat.mul_(1.0 / self.world_size)
        torch._foreach_copy_(grads, [v.view_as(g) for v, g in zip(flat.split([g.numel() for g in grads]), grads)])

    # -- the same phase as hipGraph launches --------------------------------------------------------------------------------------
    def _training_state(self, phase):
        """Every tensor a run of `phase` may write: parameters and buffers of G and D (the generator's w_avg moves in every G pass) and the
        phase's optimiser state."""
        tensors = [t.detach() for mod in (self.G, self.D) for t in misc.params_and_buffers(mod)]
        opt_state = [(p, k, v) for p, st in phase['opt'].state.items() for k, v in st.items() if isinstance(v, torch.Tensor)]
        return tensors, opt_state

    def _run_phase_graph(self, phase, real_img, real_c, real_t, gen_z, gen_c, gen_t):
        """A phase as two hipGraphs: (A) zero_grad + forward + backward, (B) gradient sanitising + Adam; with several ranks the flat gradient
        all-reduce runs eagerly between them (no collective is captured).
        First call of a phase: eager warm-up runs on a side stream (library initialisation, allocator warm-up, Adam state allocation) -- on
        the LIVE models, so parameters, buffers and optimiser state are put back afterwards: the warm-up must not count as training -- then the
        two captures (which execute nothing) and one replay of each, which is this iteration's update.  Later calls copy the inputs into the
        captured buffers and replay.  Every kernel of the native library launches on torch's current stream without allocating or
        synchronising, so it is capture-safe as is."""
        name = phase['name']
        if self.augment_pipe is not None:
            self.augment_pipe.begin_phase(name)      # the augmentation parameters of this run of the phase: drawn on the host, copied into the tensors the graph reads
        try:
            return self._run_phase_graph_body(phase, name, real_img, real_c, real_t, gen_z, gen_c, gen_t)
        finally:
            if self.augment_pipe is not None:
                self.augment_pipe.end_phase()

    def _run_phase_graph_body(self, phase, name, real_img, real_c, real_t, gen_z, gen_c, gen_t):
        entry = self._graphs.get(name)
        if entry is None:
            static = dict(real_img=real_img.clone(), real_c=real_c.clone(), real_t=real_t.clone(), gen_z=gen_z.clone(), gen_c=gen_c.clone(), gen_t=gen_t.clone())
            if not self.emulate_graphs:
                tensors, opt_before = self._training_state(phase)
                saved = [t.clone() for t in tensors]
                saved_opt = {(id(p), k): v.clone() for p, k, v in opt_before}
                side = torch.cuda.Stream(device=self.device)
                side.wait_stream(torch.cuda.current_stream(self.device))
                with torch.cuda.stream(side):
                    for _ in range(2):
                        self._phase_gradients(phase, False, **static)
                        self._phase_update(phase)
                    with torch.no_grad():
                        for t, s0 in zip(tensors, saved):
                            t.copy_(s0)
                        for p, k, v in self._training_state(phase)[1]:      # in place: the capture below records these tensors' addresses
                            s0 = saved_opt.get((id(p), k))
                            v.copy_(s0) if s0 is not None else v.zero_()
                torch.cuda.current_stream(self.device).wait_stream(side)
            params = list(phase['module'].parameters())
            if self.emulate_graphs:
                g_grad = _EmulatedGraph(params, lambda: self._training_state(phase)[0])
                g_upd = _EmulatedGraph(params, None, reads_grads_of=g_grad)
            else:
                g_grad = _HipGraph()
            out = g_grad.capture(lambda: self._phase_gradients(phase, False, **static))
            # the gradient tensors of THIS capture: what every later replay rewrites, what the update graph reads, and therefore what the
            # all-reduce between the two must reduce (`p.grad` stops pointing at them after the first eager reg phase)
            grads = [p.grad for p in params if p.grad is not None]
            if not self.emulate_graphs:
                g_upd = _HipGraph(pool=g_grad.pool())
            g_upd.capture(lambda: self._phase_update(phase))
            entry = self._graphs[name] = dict(grad=g_grad, update=g_upd, static=static, out=out, grads=grads)
        else:
            for key, val in (('real_img', real_img), ('real_c', real_c), ('real_t', real_t), ('gen_z', gen_z), ('gen_c', gen_c), ('gen_t', gen_t)):
                entry['static'][key].copy_(val)
        entry['grad'].replay()
        if self.ddp_manual:
            self._allreduce_gradients(phase, grads=entry['grads'])
        entry['update'].replay()
        # the replays rewrote the phase's parameters (and every captured activation) behind the tensors' version counters: no magnitude bound taken
        # before this point may be trusted by a later eager phase (Greg / Dreg run eagerly between captured main phases)
        from ..torch_utils.ops import amax as _amax
        _amax.graph_replayed()
        out = getattr(entry['grad'], 'out', None) or entry['out']      # (the emulated graph returns each replay's own result)
        return {k: v.clone() for k, v in out.items()}

    # -- one iteration ----------------------------------------------------------------------------
    def step(self, real_img=None, real_t=None):
        if real_img is None:
            real_img = self.synthetic_real_batch()
        if real_t is None:
            real_t = sample_frame_times(self.sampling, self.batch_gpu, generator=self.gen, device=self.device)
        real_c = torch.zeros([self.batch_gpu, 0], device=self.device)
        ran = []
        for phase in self.phases:
            if self.batch_idx % phase['interval'] != 0:
                continue
            gen_z, gen_c, gen_t = self._latents()
            if self.use_graphs and phase['name'] in ('Gmain', 'Dmain'):
                losses = self._run_phase_graph(phase, real_img, real_c, real_t, gen_z, gen_c, gen_t)
            else:
                losses = self._run_phase(phase, real_img, real_c, real_t, gen_z, gen_c, gen_t)
            self.last_losses.update(losses)
            ran.append(phase['name'])
            if self.ada is not None and 'signs_real' in losses:
                self.ada['acc'] += torch.stack([losses['signs_real'], torch.ones_like(losses['signs_real'])])
        if self.ada is not None and self.batch_idx % self.ada['interval'] == 0:   # training_loop.py:407-410, entirely on the device
            from .augment import ada_update
            mean_sign = self.ada['acc'][0] / self.ada['acc'][1].clamp(min=1)
            if self.ddp:
                torch.distributed.all_reduce(mean_sign)
                mean_sign = mean_sign / self.world_size
            ada_update(self.augment_pipe, mean_sign, self.batch_size, self.ada['interval'], self.ada['target'], self.ada['kimg'])
            self.ada['acc'].zero_()

        # G_ema (training_loop.py:392-400)
        ema_nimg = self.train_cfg.ema_kimg * 1000
        if self.train_cfg.ema_rampup is not None:
            ema_nimg = min(ema_nimg, self.cur_nimg * self.train_cfg.ema_rampup)
        beta = 0.5 ** (self.batch_size / max(ema_nimg, 1e-8))
        with torch.no_grad():
            ema_params = list(self.G_ema.parameters())
            torch._foreach_lerp_(ema_params, [p.detach() for p in self.G.parameters()], 1 - beta)  # p_ema.lerp(p, 1-beta) == p.lerp(p_ema, beta)
            torch._foreach_copy_(list(self.G_ema.buffers()), list(self.G.buffers()))
        self.cur_nimg += self.batch_size * self.frames
        self.batch_idx += 1
        return ran


def sanitize_gradients_(grads):
    """`misc.nan_to_num(param.grad, nan=0, posinf=1e5, neginf=-1e5, out=param.grad)` for every gradient of the phase
    (training_loop.py:384-386) -- one multi-tensor launch per 96 gradients instead of one launch per parameter."""
    with torch.no_grad():
        misc.nan_to_num_list_(grads, nan=0.0, posinf=1e5, neginf=-1e5)


def smoke_step(device):
    """Tiny G+D: one full iteration (Gmain, Greg no-op, Dmain, Dreg incl. R1 double-backward) on `device`."""
    g_kwargs, d_kwargs = cfgs.small_test_model_kwargs(res=32)
    train_cfg = cfgs.Config(r1_gamma=1.0, lr=0.0025, betas=(0.0, 0.99), ema_kimg=1.0, ema_rampup=0.05, G_reg_interval=4, D_reg_interval=16, pl_weight=0.0)
    ts = TrainStep(g_kwargs, d_kwargs, train_cfg, device=device, batch_gpu=4, world_size=1)
    ran = ts.step()
    assert ran == ['Gmain', 'Greg', 'Dmain', 'Dreg'], ran
    for name, p in list(ts.G.named_parameters()) + list(ts.D.named_parameters()):
        assert torch.isfinite(p).all(), name
    assert 'r1_penalty' in ts.last_losses and math.isfinite(float(ts.last_losses['r1_penalty']))
    return ts
