#!/usr/bin/env python3
"""The REFERENCE's own training step on the host cores -- the `cpu_baseline.kind = "reference"` leg (north_star: "the reference's pure-Python fallback ops timed on
the node's host cores").

Only where /root/reference exists (the build container; the checkout cannot travel to the GPU box, where bench.py reports `kind: "port"`).  Nothing of the
reference is copied: its modules are IMPORTED (with the 40-line omegaconf stand-in of tests/golden/_shims) and driven the way src/training/training_loop.py:365-389
drives them: per phase `zero_grad` -> `StyleGAN2Loss.accumulate_gradients` (src/training/loss.py:74-173) -> `nan_to_num` of the gradients -> `Adam.step`, with the
lazy-regularisation schedule of training_loop.py:186-202 (Dreg every 16th iteration; Greg is a no-op at the config's pl_weight = 0).  On CPU tensors the reference's
ops take their pure-Python path (`_upfirdn2d_ref` upfirdn2d.py:161-208, `_bias_act_ref` bias_act.py:86-123, ATen convolutions).

    python tools/cpu_reference_step.py [--json profiles/r06_cpu_baseline_reference_vs_port.json]

times the reference and this repo's port (bench.cpu_baseline's step) in ONE process on the same cores, same protocol: 2 warm-up main iterations, >= 3 timed
(median), one R1 iteration weighted 1/16."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'


def available():
    return os.path.isdir(os.path.join(REF, 'src', 'training'))


def make_reference_step(res=256, frames=3):
    """-> one(batch_idx) -> (seconds, [phase names]) for ONE video of `frames` frames: the reference's modules, loss and update order."""
    import numpy as np  # noqa: F401
    import torch
    os.environ.setdefault('PYTHONDONTWRITEBYTECODE', '1')
    sys.dont_write_bytecode = True
    for p in (os.path.join(ROOT, 'tests', 'golden', '_shims'), REF, os.path.join(REF, 'src')):
        if p not in sys.path:
            sys.path.insert(0, p)
    from omegaconf import OmegaConf
    from training.networks import Generator, Discriminator
    from training.loss import StyleGAN2Loss
    assert res == 256, 'FFS 256^2 (BASELINE config 3) is the configuration this leg restates'
    sampling = dict(type='random', num_frames_per_video=frames, max_num_frames=1024, total_dists=[1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048], max_dist=32)
    gcfg = OmegaConf.create(dict(sampling=sampling, use_noise=False, input=dict(type='temporal'), w_dim=512, z_dim=512, c_dim=0,
                                 motion=dict(z_dim=512, v_dim=512, motion_z_distance=16, gen_strategy='conv', kernel_size=11, use_fractional_t=True, fourier=True),
                                 time_enc=dict(cond_type='concat_const', dim=256, min_period_len=16, max_period_len=1024, phase_dropout_std=1.0)))
    dcfg = OmegaConf.create(dict(sampling=sampling, concat_res=16, num_frames_div_factor=2, dummy_c=False))
    torch.manual_seed(0)
    G = Generator(c_dim=0, w_dim=512, img_resolution=res, img_channels=3, mapping_kwargs=dict(num_layers=2, cfg=gcfg),
                  synthesis_kwargs=dict(channel_base=16384, channel_max=512, num_fp16_res=0, conv_clamp=None), cfg=gcfg)
    D = Discriminator(c_dim=0, img_resolution=res, img_channels=3, channel_base=16384, channel_max=512, num_fp16_res=0, conv_clamp=None,
                      mapping_kwargs=dict(num_layers=2), epilogue_kwargs=dict(mbstd_group_size=1), cfg=dcfg)
    assert sum(p.numel() for p in G.parameters()) == 32105941 and sum(p.numel() for p in D.parameters()) == 25333568      # the benchmark's models
    G.train().requires_grad_(False)
    D.train().requires_grad_(False)
    lcfg = OmegaConf.create(dict(sampling=sampling, model=dict(loss_kwargs=dict())))
    loss = StyleGAN2Loss(cfg=lcfg, device=torch.device('cpu'), G_mapping=G.mapping, G_synthesis=G.synthesis, D=D, augment_pipe=None,
                         style_mixing_prob=0.0, r1_gamma=1.0, pl_weight=0.0)
    # training_loop.py:186-202: lazy regularisation rescales lr / betas by mb_ratio = interval / (interval + 1); G has no reg phase at pl_weight 0 ('Gboth' -> here 'Gmain')
    opt_g = torch.optim.Adam(G.parameters(), lr=0.002, betas=(0.0, 0.99), eps=1e-8)
    mb = 16 / 17
    opt_d = torch.optim.Adam(D.parameters(), lr=0.002 * mb, betas=(0.0 ** mb, 0.99 ** mb), eps=1e-8)
    gen = torch.Generator().manual_seed(1)

    def one(batch_idx):
        t1 = time.time()
        real = torch.rand([1, frames, 3, res, res], generator=gen) * 2 - 1
        c = torch.zeros([1, 0])
        real_t = torch.sort(torch.rand([1, frames], generator=gen) * 32, dim=1).values
        z = torch.randn([1, 512], generator=gen)
        gen_t = torch.sort(torch.rand([1, frames], generator=gen) * 32, dim=1).values
        phases = [('Gmain', G, opt_g, 1), ('Dmain', D, opt_d, 1)] + ([('Dreg', D, opt_d, 16)] if batch_idx % 16 == 0 else [])
        for name, module, opt, interval in phases:
            opt.zero_grad(set_to_none=True)
            module.requires_grad_(True)
            loss.accumulate_gradients(phase=name, real_img=real, real_c=c, real_t=real_t, gen_z=z, gen_c=c, gen_t=gen_t, sync=True, gain=interval)
            module.requires_grad_(False)
            for p in module.parameters():       # training_loop.py:384-386
                if p.grad is not None:
                    torch.nan_to_num(p.grad, nan=0, posinf=1e5, neginf=-1e5, out=p.grad)
            opt.step()
        return time.time() - t1, [p[0] for p in phases]
    return one


def time_step(one, warmups=2, timed=3, with_reg=True):
    t_warm = [one(1)[0] for _ in range(warmups)]
    reps = [one(1) for _ in range(timed)]
    ts = sorted(r[0] for r in reps)
    t_main = ts[len(ts) // 2]
    t_reg, ph_reg = one(16) if with_reg else (None, None)
    per_iter = t_main + (max(t_reg - t_main, 0.0) / 16.0 if t_reg is not None else 0.0)
    return dict(seconds_warmup_iterations=[round(v, 3) for v in t_warm], main_iteration_samples=[round(r[0], 3) for r in reps], seconds_main_iteration_median=t_main,
                phases_main=reps[0][1], seconds_reg_iteration=t_reg, phases_reg=ph_reg, seconds_per_iteration=per_iter)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--json', default=None)
    ap.add_argument('--threads', type=int, default=0)
    ap.add_argument('--frames', type=int, default=3)
    args = ap.parse_args()
    import torch
    sys.path.insert(0, ROOT)
    threads = args.threads or len(os.sched_getaffinity(0))
    torch.set_num_threads(threads)
    out = dict(host_cpus=threads, frames=args.frames, what='one video x %d frames at 256^2, fp32, FFS-256 models (32.1 M + 25.3 M parameters), %d threads; 2 warm-up + 3 timed main iterations '
               '(median) + one R1 iteration weighted 1/16' % (args.frames, threads))
    assert available(), 'no reference checkout here'
    r = time_step(make_reference_step(256, args.frames))
    r['value_img_s'] = args.frames / r['seconds_per_iteration']
    out['reference'] = r
    print('reference:', json.dumps(r), flush=True)
    # the port: the step bench.cpu_baseline times on the GPU box
    from stylegan_v_amd.training import config as cfgs
    from stylegan_v_amd.training.train_step import TrainStep
    g_kwargs, d_kwargs, train_cfg = cfgs.model_kwargs(resolution=256, batch_size=1, num_gpus=1, fp32=True, num_frames_per_video=args.frames)
    ts = TrainStep(g_kwargs, d_kwargs, train_cfg, device='cpu', batch_gpu=1, world_size=1, ddp=False)

    def port_one(batch_idx):
        ts.batch_idx = batch_idx
        t1 = time.time()
        phases = ts.step()
        return time.time() - t1, phases
    p = time_step(port_one)
    p['value_img_s'] = args.frames / p['seconds_per_iteration']
    out['port'] = p
    out['port_over_reference'] = p['value_img_s'] / r['value_img_s']
    print('port:', json.dumps(p), flush=True)
    print('port / reference (img/s): %.3f' % out['port_over_reference'])
    if args.json:
        with open(args.json, 'w') as fh:
            json.dump(out, fh, indent=1)


if __name__ == '__main__':
    main()
