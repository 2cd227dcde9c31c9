"""SGV_D_CONCAT (loss.py `d_concat`, on by default since round 4): the Dmain phase as ONE discriminator pass over [generated, real] clips -- same losses, same gradients
as the reference's two passes (src/training/loss.py:122-151); the minibatch-std layer keeps its groups inside each half."""
import torch

from stylegan_v_amd.training import config as cfgs
from stylegan_v_amd.training.networks import MinibatchStdLayer, _mbstd_tls, minibatch_std_segments
from stylegan_v_amd.training.train_step import TrainStep


def test_minibatch_std_segments_equal_separate_batches():
    g = torch.Generator().manual_seed(0)
    for n, group, f in ((8, 4, 1), (12, 4, 2), (4, 4, 1), (6, None, 1)):
        layer = MinibatchStdLayer(group, num_channels=f)
        a, b = torch.randn([n, 6, 4, 4], generator=g), torch.randn([n, 6, 4, 4], generator=g)
        want = torch.cat([layer(a), layer(b)])
        layer.segments = 2
        got = layer(torch.cat([a, b]))
        assert torch.allclose(got, want, rtol=0, atol=1e-6)
        layer.segments = 1
        assert torch.equal(layer(a), want[:n])
        with minibatch_std_segments(2):          # the per-pass, thread-local form loss.py uses
            assert torch.equal(layer(torch.cat([a, b])), got)
            seen = []
            import threading
            th = threading.Thread(target=lambda: seen.append(torch.equal(layer(a), want[:n])))   # another thread is not affected
            th.start(); th.join()
            assert seen == [True]
        assert torch.equal(layer(a), want[:n])


def test_dmain_as_one_pass_has_the_two_pass_gradients():
    g_kwargs, d_kwargs = cfgs.small_test_model_kwargs(res=32)
    train_cfg = cfgs.Config(r1_gamma=1.0, lr=0.0025, betas=(0.0, 0.99), ema_kimg=1.0, ema_rampup=0.05, G_reg_interval=4, D_reg_interval=16, pl_weight=0.0)
    torch.manual_seed(0)
    ts = TrainStep(g_kwargs, d_kwargs, train_cfg, device='cpu', batch_gpu=8, world_size=1, ddp=False)
    ts.D.requires_grad_(True)
    b, fr = 8, ts.frames
    g = torch.Generator().manual_seed(1)
    real = torch.randn([b, fr, 3, 32, 32], generator=g)
    c = torch.zeros([b, 0])
    real_t = torch.sort(torch.randint(0, 16, [b, fr], generator=g).float(), dim=1).values
    gen_t = torch.sort(torch.randint(0, 16, [b, fr], generator=g).float(), dim=1).values
    gen_z = torch.randn([b, ts.G.z_dim], generator=g)

    def run(concat):
        ts.loss.d_concat = concat
        for p in ts.D.parameters():
            p.grad = None
        torch.manual_seed(5)     # the generator draws its motion codes inside the pass
        out = ts.loss.accumulate_gradients('Dmain', real, c, real_t, gen_z, c, gen_t, sync=True, gain=1)
        return out, [p.grad.clone() for p in ts.D.parameters()]
    want, gw = run(False)
    got, gg = run(True)
    assert getattr(_mbstd_tls, 'segments', None) is None and all(m.segments == 1 for m in ts.D.modules() if isinstance(m, MinibatchStdLayer)), 'the segment hint must not outlive the pass'
    assert torch.allclose(got['D/loss'], want['D/loss'], rtol=1e-6) and torch.equal(got['signs_real'], want['signs_real'])
    for a, r in zip(gg, gw):
        assert torch.allclose(a, r, rtol=1e-4, atol=1e-5 * max(r.abs().max().item(), 1e-8))
    # the regularised iteration (Dboth / Dreg) keeps the reference's two passes
    ts.loss.d_concat = True
    out = ts.loss.accumulate_gradients('Dreg', real, c, real_t, gen_z, c, gen_t, sync=True, gain=16)
    assert 'r1_penalty' in out


def test_frame_time_bound_is_scoped_to_the_calling_thread():
    """motion.frame_times_bounded_by (the training passes' promise that replaces the reference's t.max().item() read, motion.py:97-100) lives in
    thread-local storage: a generation thread using a generator meanwhile keeps the reference's sizing from t.max()."""
    import threading
    from stylegan_v_amd.training import motion
    g_kwargs, _ = cfgs.small_test_model_kwargs(res=32)
    torch.manual_seed(0)
    from stylegan_v_amd.training.networks import Generator
    G = Generator(**g_kwargs).eval()
    enc = G.synthesis.motion_encoder
    far = torch.tensor([[0.0, 100.0, 300.0]])
    want = enc.get_max_traj_len(far)
    seen = {}
    with motion.frame_times_bounded_by(31.0):
        inside = enc.get_max_traj_len(far)                      # this thread: sized from the promised bound, not from t
        th = threading.Thread(target=lambda: seen.setdefault('other', enc.get_max_traj_len(far)))
        th.start(); th.join()
    assert seen['other'] == want and inside < want
    assert enc.get_max_traj_len(far) == want
