"""Kernel census of ONE captured main iteration (Gmain + Dmain replayed), for `rocprofv3 --kernel-trace`: captures the step, then a marker (a cumsum over 54,321 elements),
then 4 replayed main iterations, then the marker again.  tools/captured_census_report.py splits the trace at the markers.

    rocprofv3 --kernel-trace --output-format csv -d OUT -- python tools/captured_census.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import stylegan_v_amd
from stylegan_v_amd.torch_utils import custom_ops
from stylegan_v_amd.training import config as cfgs
from stylegan_v_amd.training import train_step as tsmod

device = torch.device('cuda', 0)
custom_ops.get_native()
stylegan_v_amd.configure_miopen(immediate=True)
g_kwargs, d_kwargs, train_cfg = cfgs.model_kwargs(resolution=256, batch_size=32, num_gpus=1, fp32=True, num_frames_per_video=3, lowp_dtype=None)
ts = tsmod.TrainStep(g_kwargs, d_kwargs, train_cfg, device=device, batch_gpu=32, world_size=1, rank=0, use_graphs=True, augment='noaug')
ts.batch_idx = 0
ts.step(); ts.batch_idx = 1; ts.step()
torch.cuda.synchronize()
marker = torch.ones(54321, device=device)
marker.cumsum(0); torch.cuda.synchronize()
for _ in range(4):
    ts.batch_idx = 1
    ts.step()
torch.cuda.synchronize()
marker.cumsum(0); torch.cuda.synchronize()
print('done')
