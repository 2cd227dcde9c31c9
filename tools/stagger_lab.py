"""Lab (round 5, GPU call 26): does the stride-1 convolution run at a higher clock when its workgroups do NOT run their chunks in phase?

The step's matrix-bound kernels hold 1.95 GHz (eager) to 2.33 GHz (some replays) at LOWER power the higher the clock (profiles/r05_c25_*): the chip is not at its
package power cap, something reacts to how the current is drawn.  All 256 workgroups of the persistent convolution start together and run chunk after chunk (barrier, 216 MFMAs per consumer wave, barrier) in lockstep.  SGV_CONV_STAGGER=k delays workgroup b by (b % 16) * k * ~2 us at its start.  Back-to-back launches of one layer, time per launch and the card's clock / power.

    SGV_CONV_STAGGER=k python tools/stagger_lab.py

Result (profiles/r05_c26_stagger.log): in phase, the kernel runs AT THE PACKAGE POWER CAP -- 1,399-1,400 W over 1,200 back-to-back launches, engine clock 1.65-1.68 GHz,
395-424 TFLOP/s -- and staggering only adds its delay (388 / 380 / 374 / 351 TFLOP/s for k = 1 / 2 / 4 / 8).  The stagger parameter was removed from the kernel again; without
it this script is the power / clock probe of the stride-1 convolution (the variable is ignored).
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
import stylegan_v_amd  # noqa: F401
from stylegan_v_amd.torch_utils import custom_ops
from stylegan_v_amd.torch_utils.ops import conv2d_gradfix


def main():
    dev = torch.device('cuda', 0)
    custom_ops.get_native()
    torch.manual_seed(0)
    sampler = bench.PowerSampler(0, period=0.01)
    for (n, c, res) in ((96, 128, 128), (96, 256, 64), (192, 128, 128)):
        x = torch.randn(n, c, res, res, device=dev)
        w = torch.randn(c, c, 3, 3, device=dev) * 0.05
        with torch.no_grad():
            for _ in range(20):
                y = conv2d_gradfix.conv2d(x, w, padding=1)
            torch.cuda.synchronize()
            reps = 1200
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with sampler:
                a.record()
                for _ in range(reps):
                    y = conv2d_gradfix.conv2d(x, w, padding=1)
                b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / reps
        p = sampler.summary() or {}
        tf = 2.0 * n * res * res * c * c * 9 / (ms * 1e-3) / 1e12
        print(f'stagger {os.environ.get("SGV_CONV_STAGGER", "0")}: [{n} x {c} x {res}^2] {ms:.4f} ms per launch = {tf:.1f} TFLOP/s; sclk {p.get("sclk_MHz")} MHz ({p.get("sclk_MHz_min")}..{p.get("sclk_MHz_max")}), '
              f'{p.get("socket_W")} W, {p.get("hotspot_C")} C', flush=True)


if __name__ == '__main__':
    main()
