"""GPU time of each phase of the train step (Gmain, Dmain, Dreg / R1) at the benchmark configuration, plus the native-kernel families inside Dreg.
`python tools/phase_times.py [videos]`"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stylegan_v_amd
from stylegan_v_amd.torch_utils import custom_ops
from stylegan_v_amd.training import config as cfgs
from stylegan_v_amd.training.train_step import TrainStep, sample_frame_times

videos = int(sys.argv[1]) if len(sys.argv) > 1 else 32
stylegan_v_amd.configure_miopen(immediate=True)
dev = torch.device('cuda', 0)
g_kwargs, d_kwargs, train_cfg = cfgs.model_kwargs(resolution=256, batch_size=videos, num_gpus=1, fp32=True, num_frames_per_video=3, lowp_dtype=None)
ts = TrainStep(g_kwargs, d_kwargs, train_cfg, device=dev, batch_gpu=videos)
for _ in range(3):
    ts.batch_idx = 0
    ts.step()
torch.cuda.synchronize()
real_img = ts.synthetic_real_batch()
real_t = sample_frame_times(ts.sampling, ts.batch_gpu, generator=ts.gen, device=dev)
real_c = torch.zeros([ts.batch_gpu, 0], device=dev)
for phase in ts.phases:
    times = []
    fam = None
    for rep in range(4):
        gen_z, gen_c, gen_t = ts._latents()
        if rep == 3:
            custom_ops.prof_enable(1 << 16)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        ts._run_phase(phase, real_img, real_c, real_t, gen_z, gen_c, gen_t)
        b.record(); torch.cuda.synchronize()
        if rep == 3:
            custom_ops.prof_disable(); fam = custom_ops.prof_collect()
        else:
            times.append(a.elapsed_time(b))
    native = sum(v['ms'] for v in fam.values())
    print(f"{phase['name']:6s} interval {phase['interval']:2d}: {min(times):7.1f} ms   (native kernels {native:6.1f} ms: " +
          ', '.join(f"{k} {v['ms']:.1f}" for k, v in sorted(fam.items(), key=lambda kv: -kv[1]['ms']) if v['ms'] > 0.5) + ')')
