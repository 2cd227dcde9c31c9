#!/usr/bin/env python3
"""Forward / adjoint time of the ADA geometric kernels against the reflect margin handed to them (identity maps): where does the static worst-case margin cost time?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stylegan_v_amd.torch_utils.ops import resample  # noqa: E402
from stylegan_v_amd.training.augment import AugmentPipe, BGC  # noqa: E402


def main():
    dev = torch.device('cuda')
    pipe = AugmentPipe(**BGC).to(dev)
    x = torch.randn([32, 9, 256, 256], device=dev)
    v = torch.randn_like(x)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for m in (6, 7, 8, 16, 32, 64, 128, 200, 255):
        pipe.static_margin = False
        g_inv = torch.eye(3).repeat(32, 1, 1)
        # theta for a forced margin m on every side: _theta's bookkeeping with the margin overridden
        import stylegan_v_amd.training.augment as A
        w = h = 256
        pad = 3
        wu, hu = (w + 2 * m) * 2, (h + 2 * m) * 2
        g = g_inv
        s2, s2i = A._scale(2, 2, g), A._scale(0.5, 0.5, g)
        g = s2 @ g @ s2i
        g = A._translate(-0.5, -0.5, g) @ g @ A._translate(0.5, 0.5, g)
        out_h, out_w = (h + pad * 2) * 2, (w + pad * 2) * 2
        g = A._scale(2 / wu, 2 / hu, g) @ g @ A._scale(out_w / 2, out_h / 2, g)
        theta = g[:, :2, :].contiguous().to(dev)
        xg = x.clone().requires_grad_(True)
        res = []
        for label, fn in (('forward', lambda: resample.ada_geometric(x, theta, pipe.Hz_geom, (m, m, m, m), f_host=pipe._filter_taps())),
                          ('adjoint', None)):
            if fn is None:
                y = resample.ada_geometric(xg, theta, pipe.Hz_geom, (m, m, m, m), f_host=pipe._filter_taps())
                fn = lambda: torch.autograd.grad(y, xg, v, retain_graph=True)     # noqa: E731
            for _ in range(6):
                fn()
            ts = []
            for _ in range(3):
                e0.record()
                for _ in range(8):
                    fn()
                e1.record()
                e1.synchronize()
                ts.append(e0.elapsed_time(e1) / 8)
            res.append(sorted(ts)[1] * 1e3)
        print(f'margin {m:4d}: forward {res[0]:7.1f} us   adjoint {res[1]:7.1f} us')


if __name__ == '__main__':
    main()
