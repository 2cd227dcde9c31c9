"""Lab (round 5, GPU calls 17-18): why does one captured set of the train step replay in 141-144 ms per main iteration and another -- same process, same
code, other buffers -- in 149-152 ms?

Captures several graph sets of the FFS-256 step in ONE process, every native launch of each bracketed by the timestamp kernels of csrc/sgv_runtime.hip, and
prints per set: the device time of a main iteration (events around 6 replays, median), the sum of the native kernels' own durations by family, and the rest
(torch's element-wise kernels + whatever lies between the nodes).  If the slow sets have slower KERNELS, the cause is where their buffers lie; if the kernels
agree and the rest differs, it is the runtime's node dispatch.

    python tools/graph_sets_lab.py [sets=4] [stamped=1]
"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import contextlib

import torch

import stylegan_v_amd
from stylegan_v_amd.torch_utils import custom_ops
from stylegan_v_amd.training import config as cfgs
from stylegan_v_amd.training import train_step as tsmod


import bench      # (the PCI-matched hwmon sampler)


class Sampler(bench.PowerSampler):
    def __init__(self):
        super().__init__(0, period=0.005)

    def summary(self):
        d = super().summary()
        if not d:
            return 'no hwmon files for this device'
        return f"{d['samples']} samples of {d.get('card')}: " + ', '.join(f"{k} {d[k]:.0f} ({d[k + '_min']:.0f}..{d[k + '_max']:.0f})" for k in ('sclk_MHz', 'socket_W', 'hotspot_C') if k in d)


SAMPLER = None


def main():
    global SAMPLER
    SAMPLER = Sampler()      # (needs the device)
    n_sets = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    stamped = (sys.argv[2] if len(sys.argv) > 2 else '1') == '1'
    device = torch.device('cuda', 0)
    custom_ops.get_native()
    stylegan_v_amd.configure_miopen(immediate=True)
    g_kwargs, d_kwargs, train_cfg = cfgs.model_kwargs(resolution=256, batch_size=32, num_gpus=1, fp32=True, num_frames_per_video=3, lowp_dtype=None)
    custom_ops.prof_families(None)
    custom_ops.prof_enable(1 << 15)
    custom_ops.prof_disable()

    @contextlib.contextmanager
    def hook():
        custom_ops.prof_resume()
        try:
            yield
        finally:
            custom_ops.prof_disable()

    ts = tsmod.TrainStep(g_kwargs, d_kwargs, train_cfg, device=device, batch_gpu=32, world_size=1, rank=0, use_graphs=True, augment='noaug')

    def timed_replays(k=6):
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(k + 1)]
        with SAMPLER:
            marks[0].record()
            for i in range(k):
                ts.batch_idx = 1
                ts.step()
                marks[i + 1].record()
            torch.cuda.synchronize()
        return [marks[i].elapsed_time(marks[i + 1]) for i in range(k)]

    sets = []
    for s in range(n_sets):
        ts._graphs = {}
        tsmod._HipGraph.capture_hook = hook if stamped else None
        ts.batch_idx = 0 if s == 0 else 1          # the first iteration also runs the eager regularisation phases once (allocator, MIOpen)
        ts.step()
        torch.cuda.synchronize()
        tsmod._HipGraph.capture_hook = None
        ms = timed_replays()
        line = f'set {s}: main iteration {statistics.median(ms):8.3f} ms (min {min(ms):.3f}, max {max(ms):.3f})'
        if stamped:
            recs = custom_ops.prof_collect_records(1 << 15, with_variant=False)
            fam = {}
            for name, t, _, _ in recs:
                fam[name] = fam.get(name, 0.0) + t
            total = sum(fam.values())
            line += f'; {len(recs)} native launches, own time {total:8.3f} ms, rest {statistics.median(ms) - total:7.3f} ms; ' + \
                    ', '.join(f'{k} {v:.2f}' for k, v in sorted(fam.items(), key=lambda kv: -kv[1])[:8])
        line += ' | ' + SAMPLER.summary()
        print(line, flush=True)
        sets.append(ts._graphs)
    # second round: every set again, in order and then reversed (is the speed a property of the set, or of the moment?)
    for order in (range(n_sets), reversed(range(n_sets))):
        out = []
        for s in order:
            ts._graphs = sets[s]
            ms = timed_replays(4)
            out.append(f'set {s}: {statistics.median(ms):.3f} [{SAMPLER.summary()}]')
        print('again: ' + ', '.join(out), flush=True)
    try:
        free, total = torch.cuda.mem_get_info()
        print(f'memory: {torch.cuda.memory_reserved() / 2**30:.1f} GiB reserved by torch, {(total - free) / 2**30:.1f} GiB in use on the device')
    except Exception as e:      # noqa
        print('mem_get_info failed:', e)


if __name__ == '__main__':
    main()
