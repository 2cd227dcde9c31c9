#!/usr/bin/env python3
"""Probe MIOpen on the hot-path convolution shapes: first-call latency (solver selection + kernel build on a
cold cache) and steady-state TFLOP/s, forward and backward.  Flushes a line per shape so a timeout still tells."""
import os
import sys
import time

os.environ.setdefault('MIOPEN_FIND_MODE', '2')
import torch

torch.backends.miopen.immediate = os.environ.get('SGV_MIOPEN_IMMEDIATE', '1') == '1'
N = int(sys.argv[1]) if len(sys.argv) > 1 else 96
dev = 'cuda'
shapes = [  # (label, Cin, Cout, H, k, stride, transposed)
    ('G b256 conv1 3x3', 64, 64, 256, 3, 1, False), ('G b256 conv0 up (convT s2)', 128, 64, 128, 3, 2, True),
    ('G b128 conv1 3x3', 128, 128, 128, 3, 1, False), ('G b64 conv1 3x3', 256, 256, 64, 3, 1, False),
    ('G b32 conv1 3x3', 512, 512, 32, 3, 1, False), ('D b256 conv1 3x3 s2', 64, 128, 257, 3, 2, False),
    ('D b256 fromrgb 1x1', 3, 64, 256, 1, 1, False), ('G torgb 1x1', 64, 3, 256, 1, 1, False), ('D b64 skip 1x1', 256, 512, 32, 1, 1, False),
]
for label, cin, cout, h, k, stride, transposed in shapes:
    x = torch.randn([N, cin, h, h], device=dev, requires_grad=True)
    w = torch.randn([cin, cout, k, k] if transposed else [cout, cin, k, k], device=dev, requires_grad=True)
    pad = 0 if (stride == 2 and not transposed) else k // 2
    fn = (lambda: torch.nn.functional.conv_transpose2d(x, w, stride=stride, padding=0)) if transposed else \
         (lambda: torch.nn.functional.conv2d(x, w, stride=stride, padding=pad))
    t0 = time.time(); y = fn(); torch.cuda.synchronize(); t_first_f = time.time() - t0
    g = torch.randn_like(y)
    t0 = time.time(); torch.autograd.grad(y, [x, w], g); torch.cuda.synchronize(); t_first_b = time.time() - t0
    flops = 2.0 * y.numel() * cin * k * k if not transposed else 2.0 * x.numel() * cout * k * k
    reps = 5
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record()
    for _ in range(reps):
        y = fn()
    e1.record()
    for _ in range(reps):
        y = fn(); torch.autograd.grad(y, [x, w], g)
    e2.record(); torch.cuda.synchronize()
    tf = e0.elapsed_time(e1) / reps
    tfb = e1.elapsed_time(e2) / reps
    print(f'{label:30s} first fwd {t_first_f:6.2f}s bwd {t_first_b:6.2f}s | fwd {tf:8.3f} ms {flops/tf/1e9:7.1f} TF | fwd+bwd {tfb:8.3f} ms {3*flops/tfb/1e9:7.1f} TF', flush=True)
