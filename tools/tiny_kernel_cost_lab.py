"""Lab (round 5, GPU call 28): what does ONE more tiny launch between the step's heavy kernels cost the captured step?

Call 27 / the third record: 240 one-thread timestamp kernels around 120 heavy launches made the iteration that carried them 11 ms slower (46 us per tiny kernel, ten times
its own duration): idle dips next to power-limited kernels change the clocks the chip runs at.  If that holds for ANY tiny launch, the ~800 small element-wise / fill /
reduction launches a main iteration still carries cost far more than their own microseconds.

Two graph sets of the same step in one process: A as it is, B with one extra one-element `add_` launched behind every convolution / dense layer's forward (forward hooks:
about 2 x the number of layers per phase, since G and D run in both phases).  Blocks of 12 replays alternate A, B, A, B ...; per block the median device time and the card's
clock / power; at the end the difference per extra launch.

    python tools/tiny_kernel_cost_lab.py [extra launches per hook = 1]
"""
import os
import statistics
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
    per_hook = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    device = torch.device('cuda', 0)
    custom_ops.get_native()
    stylegan_v_amd.configure_miopen(immediate=True)
    g_kwargs, d_kwargs, train_cfg = cfgs.model_kwargs(resolution=256, batch_size=32, num_gpus=1, fp32=True, num_frames_per_video=3, lowp_dtype=None)
    ts = tsmod.TrainStep(g_kwargs, d_kwargs, train_cfg, device=device, batch_gpu=32, world_size=1, rank=0, use_graphs=True, augment='noaug')
    dummy = torch.zeros(1, device=device)
    state = dict(on=False, count=0)

    def hook(module, inputs, output):
        if state['on']:
            for _ in range(per_hook):
                dummy.add_(1.0)
                state['count'] += 1

    n_hooks = 0
    for net in (ts.G, ts.D):
        for m in net.modules():
            if type(m).__name__ in ('Conv2dLayer', 'SynthesisLayer', 'ToRGBLayer', 'FullyConnectedLayer'):
                m.register_forward_hook(hook)
                n_hooks += 1
    sets = []
    for s, on in enumerate((False, True)):
        ts._graphs = {}
        state['on'] = on
        ts.batch_idx = 0 if s == 0 else 1
        c0 = state['count']
        ts.step()
        torch.cuda.synchronize()
        # the capturing iteration ran the hooks in its two eager warm-up passes and in the capture of each phase: a third of the count is what a replay carries
        extra = (state['count'] - c0) // 3
        sets.append(ts._graphs)
    state['on'] = False
    print(f'{n_hooks} hooked modules; set B carries {extra} extra one-element launches per main iteration', flush=True)
    sampler = bench.PowerSampler(0, period=0.01)
    t0 = time.perf_counter()
    med = {0: [], 1: []}
    for r in range(8):
        for s in (0, 1):
            ts._graphs = sets[s]
            k = 12
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
            med[s].append(ms[k // 2])
            print(f'[{time.perf_counter() - t0:5.1f} s] set {"AB"[s]}: median {ms[k // 2]:7.2f} ms (min {ms[0]:.2f})  sclk {p.get("sclk_MHz")} MHz, {p.get("socket_W")} W', flush=True)
    a, b = statistics.median(med[0]), statistics.median(med[1])
    print(f'median of block medians: A {a:.2f} ms, B {b:.2f} ms: {1e3 * (b - a) / max(extra, 1):.1f} us per extra launch ({extra} of them)', flush=True)
    pairs = [y - x for x, y in zip(med[0], med[1])]
    print('B - A per adjacent pair of blocks (ms): ' + ' '.join(f'{d:.2f}' for d in pairs), flush=True)


if __name__ == '__main__':
    main()
