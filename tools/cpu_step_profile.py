#!/usr/bin/env python3
"""Host-side cost of the eager training step: cProfile of a few iterations of the benchmark's step (32 videos x 3 frames, fp32) with the device running
asynchronously -- which Python / torch / ctypes frames the ~2,000 launches per iteration spend their host time in.  The eager step is host-bound for part
of every iteration (profiles/r05_bench_step_kernel_stats_final.csv: 144.7 ms of kernels per 158.6-ms iteration; the captured step runs 149.7 ms).

    python tools/cpu_step_profile.py [steps] [batch_gpu] > profile.txt"""
import cProfile
import io
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import stylegan_v_amd  # noqa: E402
from stylegan_v_amd.training import config as cfgs  # noqa: E402
from stylegan_v_amd.training.train_step import TrainStep  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    stylegan_v_amd.configure_miopen(immediate=True)
    dev = torch.device('cuda', 0)
    g_kwargs, d_kwargs, train_cfg = cfgs.model_kwargs(resolution=256, batch_size=batch, num_gpus=1, fp32=True, num_frames_per_video=3)
    ts = TrainStep(g_kwargs, d_kwargs, train_cfg, device=dev, batch_gpu=batch, world_size=1, rank=0, augment='noaug')
    for _ in range(4):
        ts.step()
    torch.cuda.synchronize()
    ts.batch_idx = 1        # plain iterations (no reg phases)
    t0 = time.perf_counter()
    for _ in range(steps):
        ts.step()
    t_host = time.perf_counter() - t0          # host time to ENQUEUE the steps (returns before the device is done if the host is ahead)
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f'{steps} plain iterations: host enqueue {1e3 * t_host / steps:.1f} ms / iteration, until the device is idle {1e3 * t_all / steps:.1f} ms / iteration')
    ts.batch_idx = 1
    prof = cProfile.Profile()
    prof.enable()
    for _ in range(steps):
        ts.step()
    prof.disable()
    torch.cuda.synchronize()
    for key in ('tottime', 'cumulative'):
        buf = io.StringIO()
        pstats.Stats(prof, stream=buf).strip_dirs().sort_stats(key).print_stats(45)
        print(buf.getvalue()[:9000])


if __name__ == '__main__':
    main()
