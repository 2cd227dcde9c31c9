"""Dense-layer kernel (csrc/fc.hip) on the shapes of the train step, through ops.fc._launch (no autograd), GPU microseconds per call.
Environment: SGV_FC_UNROLL=1|4, SGV_FC_WAVES=0 (follow K) | 16."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stylegan_v_amd  # noqa
from stylegan_v_amd.torch_utils.ops import fc

def t(fn, it=40):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3

out = []
for name, m, k, n in [('epilogue 8192->512 x32', 32, 8192, 512), ('affine 512->512 x96', 96, 512, 512), ('mapping 512->512 x32', 32, 512, 512),
                      ('conv1d cols 5632->512 x2048', 2048, 5632, 512), ('temporal 256->512 x768', 768, 256, 512)]:
    x = torch.randn([m, k], device='cuda'); w = torch.randn([n, k], device='cuda'); b = torch.randn([n], device='cuda')
    y = torch.empty([m, n], device='cuda'); dy = torch.randn([m, n], device='cuda'); dx = torch.empty([m, k], device='cuda'); dw = torch.empty([n, k], device='cuda')
    rs = torch.empty([n], device='cuda')
    f = t(lambda: fc._launch(x, k, 1, w, 1, k, y, n, 1, m, n, k, bias=b, act=3, alpha=0.2, gain=1.4, wgain=0.1, epilogue_act=True))
    common = dict(a_ref=y, act=3, alpha=0.2, gain=1.4, wgain=0.1)
    g = t(lambda: fc._launch(dy, n, 1, w, k, 1, dx, k, 1, m, k, n, **common))
    h = t(lambda: fc._launch(dy, 1, n, x, k, 1, dw, k, 1, n, k, m, rowsum=rs, bgain=1.0, **common))
    out.append(f'{name:30s} fwd {f:7.1f}  dx {g:7.1f}  dw {h:7.1f} us')
print(f"UNROLL={os.environ.get('SGV_FC_UNROLL', '4')} WAVES={os.environ.get('SGV_FC_WAVES', '0')}")
print('\n'.join(out))

# the trajectory convolutions as dense layers: 32 x 32-tile kernel vs tiled GEMM route (fc.large_m)
for m in (2432, 2112):
    k, n = 5632, 512
    x = torch.randn([m, k], device='cuda', requires_grad=True); w = torch.randn([n, k], device='cuda', requires_grad=True); b = torch.randn([n], device='cuda', requires_grad=True)
    dy = torch.randn([m, n], device='cuda')
    for lm in (1 << 30, 1024):
        fc.large_m = lm
        def run():
            y = fc.dense(x, w, b, weight_gain=0.01, bias_gain=0.01, act='lrelu', act_gain=1)
            torch.autograd.grad(y, [x, w, b], dy)
        def fwd():
            with torch.no_grad(): fc.dense(x, w, b, weight_gain=0.01, bias_gain=0.01, act='lrelu', act_gain=1)
        print(f'conv1d as dense, M = {m}: route {"fc kernel " if lm > 4096 else "tiled GEMM"}  fwd {t(fwd):7.1f} us   fwd + bwd {t(run):7.1f} us')
