#!/usr/bin/env python3
"""BASELINE.json configs[1]: FFS 256x256 generator-only forward, 32 videos x 3 frames, one MI355X.
Mirrors the reference's src/scripts/profile_model.py:46-80 (5 warm-up + 25 timed iterations at batch 32) and
reports frames/s for the training-mode path (scale-conv-scale) and the eval-mode path (grouped conv).
Environment: B (videos), F (frames per video), RES (resolution), LOWP.  BASELINE.json configs[4]'s per-GPU share (SkyTimelapse 1024x1024
synthesis, 16-frame clips, 8 videos over 8 GPUs) is  RES=1024 F=16 B=1."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stylegan_v_amd  # noqa: E402
from stylegan_v_amd.torch_utils import custom_ops  # noqa: E402
from stylegan_v_amd.training import config as cfgs  # noqa: E402
from stylegan_v_amd.training.networks import Generator  # noqa: E402

stylegan_v_amd.configure_miopen()
dev = torch.device('cuda')
B, F, RES = int(os.environ.get('B', 32)), int(os.environ.get('F', 3)), int(os.environ.get('RES', 256))
ITERS = int(os.environ.get('ITERS', 25))
lowp = {'none': None, 'bf16': torch.bfloat16, 'fp16': torch.float16}[os.environ.get('LOWP', 'none')]
g_kwargs, _, _ = cfgs.model_kwargs(resolution=RES, batch_size=B, num_frames_per_video=F, fp32=lowp is None, lowp_dtype=lowp)
G = Generator(**g_kwargs).to(dev).requires_grad_(False)
z, c = torch.randn([B, 512], device=dev), torch.zeros([B, 0], device=dev)
t = torch.sort(torch.rand([B, F], device=dev) * 100, dim=1).values
out = {}
for mode in ('train', 'eval'):
    G.train(mode == 'train')
    with torch.no_grad():
        for _ in range(5):
            G(z, c, t)
        torch.cuda.synchronize()
        n0 = custom_ops.launch_count()
        t0 = time.perf_counter()
        for _ in range(ITERS):
            G(z, c, t)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / ITERS
    out[mode] = dict(ms_per_forward=1e3 * dt, frames_per_s=B * F / dt, native_launches=(custom_ops.launch_count() - n0) // ITERS)
print(json.dumps(dict(workload=f'G forward {RES}^2, {B} videos x {F} frames', lowp=os.environ.get('LOWP', 'none'), **out)))
