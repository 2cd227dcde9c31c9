"""Lab (round 5, GPU call 25): in ONE process, blocks of 12 main iterations replayed from graphs alternate with blocks of 12 eager main iterations of a second TrainStep
(same configuration, own models); per block the median device time and the card's engine clock / socket power.  Question: when the captured step runs in the fast state
(141 ms, ~2.08 GHz), does the eager step between two such blocks run at that clock too?

    python tools/eager_vs_graph_state_lab.py
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
import stylegan_v_amd
from stylegan_v_amd.torch_utils import custom_ops
from stylegan_v_amd.training import config as cfgs
from stylegan_v_amd.training import train_step as tsmod


def main():
    device = torch.device('cuda', 0)
    custom_ops.get_native()
    stylegan_v_amd.configure_miopen(immediate=True)
    g_kwargs, d_kwargs, train_cfg = cfgs.model_kwargs(resolution=256, batch_size=32, num_gpus=1, fp32=True, num_frames_per_video=3, lowp_dtype=None)
    tg = tsmod.TrainStep(g_kwargs, d_kwargs, train_cfg, device=device, batch_gpu=32, world_size=1, rank=0, use_graphs=True, augment='noaug')
    te = tsmod.TrainStep(g_kwargs, d_kwargs, train_cfg, device=device, batch_gpu=32, world_size=1, rank=0, use_graphs=False, augment='noaug')
    for ts in (tg, te):
        ts.batch_idx = 0
        ts.step(); ts.step(); ts.step()
    torch.cuda.synchronize()
    sampler = bench.PowerSampler(0, period=0.01)
    t0 = time.perf_counter()

    def block(label, ts, k=12):
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(k + 1)]
        with sampler:
            marks[0].record()
            for i in range(k):
                ts.batch_idx = 1
                ts.step()
                marks[i + 1].record()
            torch.cuda.synchronize()
        ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(k))
        p = sampler.summary() or {}
        print(f'[{time.perf_counter() - t0:5.1f} s] {label:9s} median {ms[k // 2]:7.2f} ms (min {ms[0]:.2f})  sclk {p.get("sclk_MHz")} MHz ({p.get("sclk_MHz_min")}..{p.get("sclk_MHz_max")}), '
              f'{p.get("socket_W")} W, {p.get("hotspot_C")} C', flush=True)

    for r in range(8):
        block('captured', tg)
        block('eager', te)


if __name__ == '__main__':
    main()
