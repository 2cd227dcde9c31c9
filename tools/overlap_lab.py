"""Lab (round 5, GPU call 24): do an MFMA-bound and an HBM-bound kernel of the step run CONCURRENTLY when they are launched on two streams?

The step is a chain of ~90 ms of matrix-bound convolutions and ~35 ms of HBM-bound passes (FIR, modulation, layer tails) per main iteration; branches that are independent
in the graph (the discriminator's skip path, toRGB) could overlap the two kinds.  The stride-1 convolution is a persistent kernel (one workgroup per CU, most of the LDS, 256
registers per consumer wave), so whether anything can run next to it is a question for the hardware.  Measured here: T(conv), T(FIR x k), T(both on one stream),
T(conv on stream 1 || FIR x k on stream 2), for the 128^2 x 128-channel layer at 96 images.

    python tools/overlap_lab.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import stylegan_v_amd  # noqa: F401
from stylegan_v_amd.torch_utils import custom_ops
from stylegan_v_amd.torch_utils.ops import conv2d_gradfix, upfirdn2d


def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    dev = torch.device('cuda', 0)
    custom_ops.get_native()
    torch.manual_seed(0)
    for (n, c, res, k_fir) in ((96, 128, 128, 8), (96, 256, 64, 8), (96, 64, 256, 4)):
        x = torch.randn(n, c, res, res, device=dev)
        w = torch.randn(c, c, 3, 3, device=dev) * 0.05
        y_fir_in = torch.randn(n, c, res, res, device=dev)
        f = upfirdn2d.setup_filter([1, 3, 3, 1], device=dev)
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        with torch.no_grad():
            conv = lambda: conv2d_gradfix.conv2d(x, w, padding=1)
            fir = lambda: [upfirdn2d.filter2d(y_fir_in, f, padding=[1, 2, 1, 2]) for _ in range(k_fir)]
            n0 = custom_ops.launch_count(); conv(); fir(); torch.cuda.synchronize()
            assert custom_ops.launch_count() - n0 >= 1 + k_fir, 'native kernels were not used'

            def serial():
                conv(); fir()

            def parallel():
                cur = torch.cuda.current_stream()
                s1.wait_stream(cur); s2.wait_stream(cur)
                with torch.cuda.stream(s1):
                    conv()
                with torch.cuda.stream(s2):
                    fir()
                cur.wait_stream(s1); cur.wait_stream(s2)

            def parallel_fir_first():
                cur = torch.cuda.current_stream()
                s1.wait_stream(cur); s2.wait_stream(cur)
                with torch.cuda.stream(s2):
                    fir()
                with torch.cuda.stream(s1):
                    conv()
                cur.wait_stream(s1); cur.wait_stream(s2)

            t_c, t_f, t_s, t_p, t_q = timed(conv), timed(fir), timed(serial), timed(parallel), timed(parallel_fir_first)
        print(f'[{n} x {c} x {res}^2] conv {t_c:.3f} ms, FIR x {k_fir} {t_f:.3f} ms, one stream {t_s:.3f} ms, two streams {t_p:.3f} ms (FIR launched first: {t_q:.3f}); '
              f'hidden {100 * (t_s - min(t_p, t_q)) / min(t_c, t_f):.0f} % of the shorter one', flush=True)


if __name__ == '__main__':
    main()
