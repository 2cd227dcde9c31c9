#!/usr/bin/env python3
"""Eager DDP path of TrainStep on an RCCL group of ONE rank (what a 1-GPU box can exercise of the N > 1 launch): DistributedDataParallel wrappers around
G.mapping / G.synthesis / D over `nccl`, the default one-pass Dmain, three iterations, finite losses, parameters moved.

    MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 python tools/ddp_one_rank_check.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from stylegan_v_amd.training import config as cfgs
    from stylegan_v_amd.training.train_step import TrainStep
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29517')
    os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
    torch.cuda.set_device(0)
    torch.distributed.init_process_group('nccl', rank=0, world_size=1)
    g_kwargs, d_kwargs, train_cfg = cfgs.model_kwargs(resolution=256, batch_size=4, num_gpus=1, fp32=True, num_frames_per_video=3)
    for aug in ('noaug', 'ada'):
        ts = TrainStep(g_kwargs, d_kwargs, train_cfg, device=torch.device('cuda', 0), batch_gpu=4, world_size=1, rank=0, ddp=True, augment=aug)
        assert ts.ddp and not ts.ddp_manual
        before = [p.detach().clone() for p in list(ts.D.parameters())[:4]]
        for i in range(3):
            phases = ts.step()
        torch.cuda.synchronize()
        losses = {k: float(v) for k, v in ts.last_losses.items()}
        assert all(v == v and abs(v) < 1e6 for v in losses.values()), losses
        moved = max((a - b.detach()).abs().max().item() for a, b in zip(before, list(ts.D.parameters())[:4]))
        assert moved > 0
        print(f'aug={aug}: DDP({type(ts.loss.D).__name__}) over nccl, 3 iterations, phases of the last {phases}, losses', {k: round(v, 4) for k, v in losses.items()}, flush=True)
    torch.distributed.destroy_process_group()
    print('one-rank RCCL DDP check ok')


if __name__ == '__main__':
    main()
