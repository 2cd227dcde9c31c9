"""Host time to enqueue one training step (no synchronisation inside the loop) against its GPU time: how far the eager step is from being launch-bound."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stylegan_v_amd
from stylegan_v_amd.training import config as cfgs
from stylegan_v_amd.training.train_step import TrainStep

videos = int(sys.argv[1]) if len(sys.argv) > 1 else 32
stylegan_v_amd.configure_miopen(immediate=True)
dev = torch.device('cuda', 0)
g_kwargs, d_kwargs, train_cfg = cfgs.model_kwargs(resolution=256, batch_size=videos, num_gpus=1, fp32=True, num_frames_per_video=3, lowp_dtype=None)
ts = TrainStep(g_kwargs, d_kwargs, train_cfg, device=dev, batch_gpu=videos)
for _ in range(4):
    ts.step()
torch.cuda.synchronize()
ts.batch_idx = 1          # no regularisation phases in the sample
n = 8
t0 = time.perf_counter()
for _ in range(n):
    ts.step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f'{videos} videos/GPU: host enqueue {1e3 * (t1 - t0) / n:.1f} ms per step, wall incl. GPU drain {1e3 * (t2 - t0) / n:.1f} ms per step')
