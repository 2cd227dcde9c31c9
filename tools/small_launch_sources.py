"""Which Python lines issue the small launches of a main iteration?  One eager main iteration (Gmain + Dmain) of the FFS-256 step under torch.profiler with stacks; every
aten operator that launched a device kernel shorter than 12 us is charged to the innermost stack frame inside this package.  Native launches (ctypes calls into
libsgv_hip.so) are not seen by the profiler; tools/captured_census.py counts those.

    python tools/small_launch_sources.py > table.txt
"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.profiler as tprof

import stylegan_v_amd
from stylegan_v_amd.torch_utils import custom_ops
from stylegan_v_amd.training import config as cfgs
from stylegan_v_amd.training import train_step as tsmod

device = torch.device('cuda', 0)
custom_ops.get_native()
stylegan_v_amd.configure_miopen(immediate=True)
g_kwargs, d_kwargs, train_cfg = cfgs.model_kwargs(resolution=256, batch_size=32, num_gpus=1, fp32=True, num_frames_per_video=3, lowp_dtype=None)
ts = tsmod.TrainStep(g_kwargs, d_kwargs, train_cfg, device=device, batch_gpu=32, world_size=1, rank=0, use_graphs=False, augment='noaug')
ts.batch_idx = 0
ts.step(); ts.batch_idx = 1; ts.step()
torch.cuda.synchronize()
with tprof.profile(activities=[tprof.ProfilerActivity.CPU, tprof.ProfilerActivity.CUDA], with_stack=True, record_shapes=True,
                   experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
    ts.batch_idx = 1
    ts.step()
    torch.cuda.synchronize()

PKG = 'stylegan'
table = collections.defaultdict(lambda: [0, 0.0, collections.Counter()])
n_small = 0
for ev in prof.events():
    if ev.device_type != torch.autograd.DeviceType.CPU or not ev.kernels:
        continue
    dur = sum(k.duration for k in ev.kernels)
    if dur >= 12.0 * len(ev.kernels):
        continue
    if any(c.kernels for c in ev.cpu_children):       # charge the innermost operator only
        continue
    frame = '(no frame inside the package: autograd engine / optimizer)'
    node, stack = ev, []
    while node is not None and not stack:
        stack = list(node.stack or [])
        node = node.cpu_parent
    hits = [fr.strip() for fr in stack if PKG in fr and 'site-packages' not in fr and 'dist-packages' not in fr]
    if hits:
        frame = hits[0] + ('  <-  ' + hits[1] if len(hits) > 1 else '')
    key = (frame[-200:], ev.name)
    table[key][0] += len(ev.kernels)
    table[key][1] += dur
    table[key][2][str(ev.input_shapes)[:60]] += 1
    n_small += len(ev.kernels)
print(f'{n_small} small device launches issued by aten operators in one eager main iteration')
for (frame, name), (n, us, shapes) in sorted(table.items(), key=lambda kv: -kv[1][0])[:90]:
    print(f'{n:5d}  {us / n:5.1f} us  {name:28s} {frame}   {shapes.most_common(2)}')
