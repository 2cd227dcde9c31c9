"""Data-parallel path on CPU: two processes (gloo) each run the training step's phases on half of a batch; the
all-reduced gradients must equal the single-process gradients of the whole batch, gradient synchronisation must be
gated per phase like the reference gates it (misc.ddp_sync / loss.py:45-69), and ranks must stay consistent."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _make(world, rank, batch_gpu, ddp, ddp_manual=None, use_graphs=False, reg_intervals=(4, 16)):
    from stylegan_v_amd.training import config as cfgs
    from stylegan_v_amd.training.train_step import TrainStep
    g_kwargs, d_kwargs = cfgs.small_test_model_kwargs(res=32)
    # D's minibatch-std groups must not straddle ranks for the equivalence: group size 2 with 2 videos per rank
    train_cfg = cfgs.Config(r1_gamma=1.0, lr=0.0025, betas=(0.0, 0.99), ema_kimg=1.0, ema_rampup=0.05, G_reg_interval=reg_intervals[0],
                            D_reg_interval=reg_intervals[1], pl_weight=0.0)
    return TrainStep(g_kwargs, d_kwargs, train_cfg, device='cpu', batch_gpu=batch_gpu, world_size=world, rank=rank, seed=0, ddp=ddp, ddp_manual=ddp_manual,
                     use_graphs=use_graphs)


def _phase_grads(ts, phase, real_img, real_t, z, t):
    c = torch.zeros([z.shape[0], 0])
    mod = ts.G if phase.startswith('G') else ts.D
    for p in list(ts.G.parameters()) + list(ts.D.parameters()):
        p.grad = None
    mod.requires_grad_(True)
    ts.loss.accumulate_gradients(phase=phase, real_img=real_img, real_c=c, real_t=real_t, gen_z=z, gen_c=c, gen_t=t, sync=True, gain=1)
    mod.requires_grad_(False)
    return {n: p.grad.clone() for n, p in mod.named_parameters() if p.grad is not None}


def _inputs():
    g = torch.Generator().manual_seed(123)
    B, F = 4, 3
    real = torch.rand([B, F, 3, 32, 32], generator=g) * 2 - 1
    real_t = torch.sort(torch.rand([B, F], generator=g) * 30, dim=1).values
    z = torch.randn([B, 32], generator=g)
    t = torch.sort(torch.rand([B, F], generator=g) * 30, dim=1).values
    return real, real_t, z, t


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        ts = _make(world, rank, batch_gpu=2, ddp=True)
        ts.loss.d_concat = False          # the reference's two discriminator passes first (src/training/loss.py:122-151); the one-pass form is compared with them below
        real, real_t, z, t = _inputs()
        sl = slice(rank * 2, rank * 2 + 2)
        # motion noise is drawn inside G: fix it per video so both runs see the same trajectories
        torch.manual_seed(1000)
        res = {}
        for phase in ('Gmain', 'Dmain', 'Dreg'):
            torch.manual_seed(77 + rank)  # rank-local RNG stream for the in-forward randn
            res[phase] = _phase_grads(ts, phase, real[sl], real_t[sl], z[sl], t[sl])
        # Dmain as ONE discriminator pass over [generated, real] clips (loss.d_concat): a single synchronised backward must leave the same
        # all-reduced gradients as the reference's two passes (the first un-synchronised, the second synchronised)
        ts.loss.d_concat = True
        torch.manual_seed(77 + rank)
        one_pass = _phase_grads(ts, 'Dmain', real[sl], real_t[sl], z[sl], t[sl])
        ts.loss.d_concat = False
        assert set(one_pass) == set(res['Dmain'])
        for name, g1 in one_pass.items():
            want = res['Dmain'][name]
            assert (g1 - want).abs().max().item() <= 2e-5 * (want.abs().max().item() + 1e-6), f'Dmain as one pass differs under DDP: {name}'
        # the wrapper-free form used under hipGraph replay (one flat all-reduce behind the backward pass) gives the same averaged gradients
        tm = _make(world, rank, batch_gpu=2, ddp=True, ddp_manual=True)
        assert tm.ddp_manual and not isinstance(tm.loss.D, torch.nn.parallel.DistributedDataParallel)
        tm.loss.d_concat = False
        tm.G.load_state_dict(ts.G.state_dict()); tm.D.load_state_dict(ts.D.state_dict())
        for phase in ('Gmain', 'Dmain', 'Dreg'):
            torch.manual_seed(77 + rank)
            _phase_grads(tm, phase, real[sl], real_t[sl], z[sl], t[sl])
            tm._allreduce_gradients(next(ph for ph in tm.phases if ph['name'] == phase))
            mod = tm.G if phase.startswith('G') else tm.D
            for name, p in mod.named_parameters():
                if p.grad is not None:
                    want = res[phase][name]
                    assert (p.grad - want).abs().max().item() <= 1e-6 * (want.abs().max().item() + 1e-6), f'manual all-reduce differs from DDP: {phase} {name}'
        assert tm.step() == ['Gmain', 'Greg', 'Dmain', 'Dreg']
        # gating: a D-only phase must leave G's DDP wrappers un-synchronised and vice versa (no hang == correct gating)
        ran = ts.step()
        assert ran == ['Gmain', 'Greg', 'Dmain', 'Dreg']
        from stylegan_v_amd.torch_utils import misc
        misc.check_ddp_consistency(ts.G, ignore_regex=r'.*\.w_avg')
        misc.check_ddp_consistency(ts.D)
        torch.save(res, os.path.join(out_dir, f'rank{rank}.pt'))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_ddp_two_ranks_match_single_process(tmp_path, monkeypatch):
    # Reference semantics to make the comparison exact: the generator draws motion noise with torch.randn inside
    # forward; patch it to a deterministic function of the video's latent so single- and multi-process runs agree.
    import stylegan_v_amd.training.motion as motion
    port = _free_port()
    ctx = mp.get_context('spawn')
    procs = [ctx.Process(target=_worker_entry, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=540)
        assert p.exitcode == 0, f'rank process failed with exit code {p.exitcode}'
    r0 = torch.load(tmp_path / 'rank0.pt')
    r1 = torch.load(tmp_path / 'rank1.pt')

    _patch_motion_noise(motion)
    ts = _make(1, 0, batch_gpu=4, ddp=False)
    ts.loss.d_concat = False
    real, real_t, z, t = _inputs()
    # D's minibatch-std layer groups sample k with sample k + N/G (networks.py:506): order the single-process batch so
    # that its groups are exactly the per-rank groups {0,1} and {2,3}.
    perm = [0, 2, 1, 3]
    real, real_t, z, t = real[perm], real_t[perm], z[perm], t[perm]
    for phase in ('Gmain', 'Dmain', 'Dreg'):
        single = _phase_grads(ts, phase, real, real_t, z, t)
        assert set(single) == set(r0[phase]) == set(r1[phase])
        for name, g in single.items():
            assert torch.allclose(r0[phase][name], r1[phase][name], atol=0, rtol=0), f'{phase} {name}: ranks disagree after all-reduce'
            err = (r0[phase][name] - g).abs().max().item()
            scale = g.abs().max().item() + 1e-6
            assert err <= 2e-4 * scale + 1e-6, f'{phase} {name}: DDP grad differs from the single-process grad by {err:.3e} (scale {scale:.3e})'


def _patch_motion_noise(motion):
    """Deterministic motion noise: seeded per video by the first frame time (same in every process)."""
    orig = motion.MotionMappingNetwork.generate_motion_u_codes

    def patched(self, c, t, motion_z=None):
        if motion_z is None:
            traj_len = self.get_max_traj_len(t) + self.num_additional_codes
            rows = []
            for i in range(t.shape[0]):
                g = torch.Generator().manual_seed(int(t[i, 0].item() * 1e6) % (2 ** 31))
                rows.append(torch.randn([64, self.cfg.motion.z_dim], generator=g))
            motion_z = torch.stack(rows)[:, :max(traj_len, 1)]
            assert motion_z.shape[1] >= traj_len
        return orig(self, c, t, motion_z=motion_z)
    motion.MotionMappingNetwork.generate_motion_u_codes = patched


def _worker_entry(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import stylegan_v_amd.training.motion as motion
    _patch_motion_noise(motion)
    _worker(rank, world, port, out_dir)


def _graph_worker(rank, world, port, out_dir):
    """The hipGraph schedule under DDP (TrainStep._run_phase_graph with the host-side graph stand-in `use_graphs='emulate'`): Gmain / Dmain replayed,
    Greg / Dreg eager on the SAME optimisers every second iteration -- so from iteration 1 on `p.grad` no longer points at the captured gradient
    buffers.  Ranks must stay bit-consistent and follow the eager manual-all-reduce schedule."""
    sys.path.insert(0, ROOT)
    import stylegan_v_amd.training.motion as motion
    _patch_motion_noise(motion)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from stylegan_v_amd.torch_utils import misc
        tg = _make(world, rank, batch_gpu=2, ddp=True, use_graphs='emulate', reg_intervals=(2, 2))
        te = _make(world, rank, batch_gpu=2, ddp=True, ddp_manual=True, reg_intervals=(2, 2))
        assert tg.use_graphs and tg.ddp_manual and te.ddp_manual and not te.use_graphs
        te.G.load_state_dict(tg.G.state_dict()); te.D.load_state_dict(tg.D.state_dict()); te.G_ema.load_state_dict(tg.G_ema.state_dict())
        g = torch.Generator().manual_seed(500 + rank)       # rank-specific data: un-reduced gradients would differ between the ranks
        schedule = []
        for it in range(5):
            real = torch.rand([2, 3, 3, 32, 32], generator=g) * 2 - 1
            real_t = torch.sort(torch.rand([2, 3], generator=g) * 30, dim=1).values
            schedule.append(tuple(tg.step(real.clone(), real_t.clone())))
            assert tuple(te.step(real.clone(), real_t.clone())) == schedule[-1]
            misc.check_ddp_consistency(tg.G, ignore_regex=r'.*\.w_avg')
            misc.check_ddp_consistency(tg.D)
            for (name, a), (_, b) in zip(list(tg.G.named_parameters()) + list(tg.D.named_parameters()), list(te.G.named_parameters()) + list(te.D.named_parameters())):
                err = (a - b).abs().max().item()
                assert err <= 1e-5 * (b.abs().max().item() + 1e-3), f'iteration {it}: {name} differs from the eager schedule by {err:.3e}'
        assert schedule == [('Gmain', 'Greg', 'Dmain', 'Dreg'), ('Gmain', 'Dmain')] * 2 + [('Gmain', 'Greg', 'Dmain', 'Dreg')]
        assert set(tg._graphs) == {'Gmain', 'Dmain'}
        # the property the emulation exists for: after an eager reg phase the live `p.grad` tensors are NOT the captured buffers
        stale = sum(int(p.grad is not buf) for p, buf in zip(tg.D.parameters(), tg._graphs['Dmain']['grad'].bufs) if buf is not None)
        assert stale > 0, 'the schedule never re-bound p.grad: the test does not exercise the captured-buffer all-reduce'
        open(os.path.join(out_dir, f'graph_rank{rank}.ok'), 'w').write('ok')
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_graph_schedule_allreduces_the_captured_gradient_buffers(tmp_path):
    port = _free_port()
    ctx = mp.get_context('spawn')
    procs = [ctx.Process(target=_graph_worker, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=540)
        assert p.exitcode == 0, f'rank process failed with exit code {p.exitcode}'
    assert (tmp_path / 'graph_rank0.ok').exists() and (tmp_path / 'graph_rank1.ok').exists()


def _bench_worker(rank, world, port, out_dir, fail):
    """bench.py's own N > 1 headline path (build_headline_step + timed_steps) on two gloo ranks, with the host-side stand-in of a replayed graph."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import bench

        def make_step(graphs):
            ts_ = _make(world, rank, batch_gpu=2, ddp=True, use_graphs=graphs, reg_intervals=(2, 2))
            if graphs and fail:      # a capture that the runtime refuses: raised inside the first step, in front of the step's first collective, on every rank
                def refuse(*a, **k):
                    raise RuntimeError('capture refused (injected)')
                ts_._run_phase_graph = refuse
            return ts_
        ts, mode = bench.build_headline_step(make_step, world, rank, 'emulate', warmup=1, dev_sync=lambda: None)
        if not fail:
            assert mode == 'emulated capture' and ts.use_graphs and ts.ddp_manual, (mode, ts.use_graphs, ts.ddp_manual)
            assert set(ts._graphs) == {'Gmain', 'Dmain'}, 'the warm-up must hold the captures: none inside the timed region'
        else:
            assert mode.startswith('eager (capture failed') and not ts.use_graphs and not ts.ddp_manual, mode     # EVERY rank fell back, not only the one that failed
        ts.batch_idx = 0
        seconds, phases, mine = bench.timed_steps(ts, 3, world, lambda: None, torch.device('cpu'))
        assert phases == {'Gmain': 3, 'Dmain': 3, 'Greg': 2, 'Dreg': 2}, phases
        assert seconds >= float(mine.item()) - 1e-9       # MAX over ranks
        both = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(both, torch.tensor([seconds], dtype=torch.float64))
        assert float(both[0]) == float(both[1]), 'every rank must report the same (max) time'
        # the ranks trained ONE model: parameters identical after the collective-synchronised steps
        flat = torch.cat([p.detach().reshape(-1) for m in (ts.G, ts.D) for p in m.parameters()])
        peers = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(peers, flat)
        assert torch.equal(peers[0], peers[1]), 'ranks diverged'
        open(os.path.join(out_dir, f'bench_rank{rank}.ok'), 'w').write(mode)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize('fail', [False, True])
def test_bench_multi_gpu_headline_path_runs_the_same_step_on_every_rank(tmp_path, fail):
    """`bench.py --gpus N`: the captured step with the manual flat all-reduce at every N (here: its host-side stand-in on two gloo ranks), and -- when a rank's capture
    is refused -- the agreed fallback of ALL ranks to the eager DDP step, reported in `headline_mode`."""
    port = _free_port()
    ctx = mp.get_context('spawn')
    procs = [ctx.Process(target=_bench_worker, args=(r, 2, port, str(tmp_path), fail)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=840)
        assert p.exitcode == 0, f'rank process failed with exit code {p.exitcode}'
    modes = {(tmp_path / f'bench_rank{r}.ok').read_text() for r in range(2)}
    assert len(modes) == 1
