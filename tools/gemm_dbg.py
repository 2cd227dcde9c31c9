import sys, torch
sys.path.insert(0, '.')
from stylegan_v_amd.torch_utils.ops import gemm
DEV = 'cuda'
def check(m, n, k, name):
    g = torch.Generator().manual_seed(m + n + k)
    a = torch.randn([m, k], generator=g).to(DEV)
    b = torch.randn([n, k], generator=g).to(DEV)
    c = gemm.matmul_nt(a, b)
    ref = a.double() @ b.double().t()
    err = (c.double() - ref).abs()
    scale = ref.abs().max().item()
    tm, tn = (m + 127) // 128, n // 128
    pad = torch.zeros([tm * 128, n], dtype=torch.float64, device=DEV); pad[:m] = err
    tiles = pad.reshape(tm, 128, tn, 128).amax(dim=(1, 3)) / scale
    bad = (tiles > 1e-4).nonzero()
    print(name, (m, n, k), 'max rel', (err.max() / scale).item(), 'bad tiles', bad.shape[0], 'of', tm * tn, bad[:12].tolist())
    if bad.shape[0]:
        i, j = bad[0].tolist()
        e = pad[i*128:(i+1)*128, j*128:(j+1)*128] / scale
        rows = (e.amax(dim=1) > 1e-4).nonzero().flatten().tolist(); cols = (e.amax(dim=0) > 1e-4).nonzero().flatten().tolist()
        print('   first bad tile rows', rows[:20], 'n', len(rows), 'cols', cols[:20], 'n', len(cols))
for shp in ((2432, 512, 5632), (2432, 5632, 512), (1024, 256, 128), (1024, 128, 256), (128, 256, 1024), (256, 512, 64), (256, 1024, 32), (256, 1024, 96), (256, 2048, 512), (256, 4096, 512), (256, 5632, 512), (128, 5632, 32)):
    check(*shp, 'nt')
