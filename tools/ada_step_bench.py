#!/usr/bin/env python3
"""The eager training step with aug=ada and with aug=noaug on one box, one process per configuration (SGV_ADA_ADJOINT=0: the differentiated calls of the
geometric block run the four-pass composition, as in rounds 4-5).

    python tools/ada_step_bench.py [--steps 8] [--aug ada|noaug] [--graphs 0|1]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stylegan_v_amd.training import config as cfgs  # noqa: E402
from stylegan_v_amd.training.train_step import TrainStep  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=8)
    ap.add_argument('--aug', type=str, default='ada')
    ap.add_argument('--graphs', type=int, default=0)
    ap.add_argument('--p', type=float, default=-1.0, help='>= 0: fix the augmentation probability at this value before the timed steps')
    args = ap.parse_args()
    g_kwargs, d_kwargs, train_cfg = cfgs.model_kwargs(resolution=256, batch_size=32, num_gpus=1, fp32=True, num_frames_per_video=3)
    ts = TrainStep(g_kwargs, d_kwargs, train_cfg, device='cuda', batch_gpu=32, world_size=1, use_graphs=bool(args.graphs), augment=args.aug)
    ts.batch_idx = 1
    for _ in range(4):
        ts.step()
    if args.p >= 0 and ts.augment_pipe is not None:
        ts.augment_pipe.p.copy_(torch.tensor(args.p))
    ts.batch_idx = 1          # (no R1 iteration inside the window: main iterations only)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        ts.step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t1) / args.steps
    print(f'aug={args.aug} graphs={args.graphs} SGV_ADA_ADJOINT={os.environ.get("SGV_ADA_ADJOINT", "1")} p={float(ts.augment_pipe.p) if ts.augment_pipe is not None else None}: '
          f'{dt*1e3:.2f} ms per main iteration, {96/dt:.1f} img/s')


if __name__ == '__main__':
    main()
