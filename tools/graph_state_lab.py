"""Lab (round 5, GPU call 22): the two speeds of the captured step over time.  Two clean graph sets A, B of the FFS-256 step in one process; then
    (1) 160 main iterations of A back to back, (2) 80 alternating A, B, (3) 80 of B, (4) 80 of A with a 20-ms host sleep every 8 iterations --
per-iteration device times (events) and the card's clock / power per phase.  Question: does the chip leave the slow state (149 ms, 1.95 GHz, 1,215 W) on its own, with time,
or with a change of what is replayed?

    python tools/graph_state_lab.py
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
    ts = tsmod.TrainStep(g_kwargs, d_kwargs, train_cfg, device=device, batch_gpu=32, world_size=1, rank=0, use_graphs=True, augment='noaug')
    sets = []
    for s in range(2):
        ts._graphs = {}
        ts.batch_idx = 0 if s == 0 else 1
        ts.step()
        torch.cuda.synchronize()
        sets.append(ts._graphs)
    sampler = bench.PowerSampler(0, period=0.01)
    # the other clock domains of the same card: pp_dpm_* list the levels of a domain, the current one starred; freq2_input is the memory clock
    import glob
    card_dir = None
    for key, (path, _) in sampler.files.items():
        card_dir = path.split('/hwmon/')[0]
    dpm = {name: os.path.join(card_dir, name) for name in ('pp_dpm_sclk', 'pp_dpm_mclk', 'pp_dpm_fclk', 'pp_dpm_socclk', 'pp_dpm_vclk', 'pp_dpm_dclk')
           if card_dir and os.path.exists(os.path.join(card_dir, name))}
    extra = {}
    if card_dir:
        for hw in glob.glob(os.path.join(card_dir, 'hwmon', 'hwmon*')):
            for name in ('freq2_input', 'in2_input', 'temp3_input', 'power1_cap'):
                if os.path.exists(os.path.join(hw, name)):
                    extra[name] = os.path.join(hw, name)

    def domains():
        out = []
        for name, path in dpm.items():
            try:
                cur = [ln.strip() for ln in open(path).read().splitlines() if '*' in ln]
                out.append(f'{name[7:]} {cur[0] if cur else "?"}')
            except Exception as e:      # noqa
                out.append(f'{name[7:]} unreadable')
        for name, path in extra.items():
            try:
                out.append(f'{name} {open(path).read().strip()}')
            except Exception:      # noqa
                pass
        return '; '.join(out)
    t_start = time.perf_counter()

    def run(label, order, sleep_every=0):
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(len(order) + 1)]
        seen = []
        with sampler:
            marks[0].record()
            for i, s in enumerate(order):
                ts._graphs = sets[s]
                ts.batch_idx = 1
                ts.step()
                marks[i + 1].record()
                if i % 20 == 10:
                    torch.cuda.current_stream().synchronize() if False else None
                    seen.append(domains())      # read while the device is busy with the iterations already queued
                if sleep_every and (i + 1) % sleep_every == 0:
                    torch.cuda.synchronize()
                    time.sleep(0.02)
            torch.cuda.synchronize()
        ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(len(order))]
        p = sampler.summary() or {}
        print(f'[{time.perf_counter() - t_start:6.1f} s] {label}: median {sorted(ms)[len(ms) // 2]:.1f} ms; sclk {p.get("sclk_MHz")} ({p.get("sclk_MHz_min")}..{p.get("sclk_MHz_max")}) MHz, '
              f'{p.get("socket_W")} W, {p.get("hotspot_C")} C', flush=True)
        print('    ' + ' '.join(f'{v:.0f}' for v in ms), flush=True)
        for d in sorted(set(seen)):
            print(f'    domains ({seen.count(d)} of {len(seen)} reads): {d}', flush=True)

    n_long = int(sys.argv[1]) if len(sys.argv) > 1 else 160
    run(f'A x {n_long}', [0] * n_long)
    if len(sys.argv) <= 1:
        run('A, B alternating x 80', [0, 1] * 40)
        run('B x 80', [1] * 80)
        run('A x 80, 20 ms of idle every 8 iterations', [0] * 80, sleep_every=8)
        run('A x 80', [0] * 80)


if __name__ == '__main__':
    main()
