"""Weight gradient of the < 32-pixel layers at the benchmark batch: own packed-sample kernel vs the vendor library (ms per call)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stylegan_v_amd  # noqa
from stylegan_v_amd.torch_utils.ops import conv2d_gradfix as cg

def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it

N = 96
for c, r in [(512, 16), (512, 8)]:
    x = torch.randn([N, c, r, r], device='cuda'); dy = torch.randn([N, c, r, r], device='cuda')
    cfg = (False, (1, 1), (1, 1), (0, 0), (1, 1), 1)
    ws = (c, c, 3, 3)
    own = t(lambda: cg._native_wrw(dy, x, cfg, ws)) if cg._native_wrw_ok(dy, x, cfg, ws) else float('nan')
    wl = x.new_empty(ws)
    lib = t(lambda: torch.ops.aten.convolution_backward(dy, x, wl, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1, [False, True, False]))
    gf = 2.0 * N * r * r * c * c * 9 / 1e9
    print(f's1 {c}ch {r}x{r}: own {own:.3f} ms ({gf / own:.0f} TF/s)  vendor {lib:.3f} ms ({gf / lib:.0f} TF/s)')
    for transposed in (False, True):
        hs = r // 2
        small = torch.randn([N, c, hs, hs], device='cuda'); big = torch.randn([N, c, r + 1, r + 1], device='cuda')
        cfg2 = (transposed, (2, 2), (0, 0), (0, 0), (1, 1), 1)
        xx, dd = (small, big) if transposed else (big, small)
        own = t(lambda: cg._native_wrw(dd, xx, cfg2, ws)) if cg._native_wrw_ok(dd, xx, cfg2, ws) else float('nan')
        lib = t(lambda: torch.ops.aten.convolution_backward(dd, xx, wl, None, (2, 2), (0, 0), (1, 1), transposed, (0, 0), 1, [False, True, False]))
        gf = 2.0 * N * hs * hs * c * c * 9 / 1e9
        print(f's2{"T" if transposed else " "} {c}ch {r + 1}->{hs}: own {own:.3f} ms  vendor {lib:.3f} ms ({gf / lib:.0f} TF/s)')
