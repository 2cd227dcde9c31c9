import sys, torch
sys.path.insert(0, '.')
from stylegan_v_amd.torch_utils.ops import fused_conv_act, conv2d_gradfix, bias_act
import oracle
DEV='cuda'
S1 = (False, (1, 1), (1, 1), (0, 0), (1, 1), 1)
for N, c, r in ((2,64,256),(8,64,256),(96,64,256),(96,512,32)):
    g = torch.Generator(device=DEV).manual_seed(1)
    x = torch.randn([N, c, r, r], generator=g, device=DEV) + 0.25
    w = torch.randn([c, c, 3, 3], generator=g, device=DEV) / (3 * c ** 0.5)
    b = torch.randn([c], generator=g, device=DEV) * 0.5
    for name, bb in (('bias', b), ('nobias', None)):
        y = fused_conv_act._launch_fused(x, w, None, None, bb, 3, 0.2, 2 ** 0.5, -1.0)
        plain = conv2d_gradfix._native_conv(x, w, S1)
        ref = bias_act.bias_act(plain, bb, act='lrelu')
        d = (y - ref).abs()
        per_frame = d.flatten(1).max(1).values
        print(N, c, r, name, 'max diff', d.max().item(), 'ref max', ref.abs().max().item(), 'bad frames', int((per_frame > 1e-3).sum()), 'first bad', (per_frame > 1e-3).nonzero().flatten()[:8].tolist())
        if d.max() > 1e-3:
            bad = (d > 1e-3)
            print('   bad channels', bad.any(dim=(0,2,3)).nonzero().flatten()[:16].tolist(), 'bad rows', bad.any(dim=(0,1,3)).nonzero().flatten()[:16].tolist(), 'frac', bad.float().mean().item())
            print('   y sample', y[0,0,0,:6].tolist(), 'ref', ref[0,0,0,:6].tolist(), 'plain', plain[0,0,0,:6].tolist())
    # oracle on a slab of frame 0
    ref0 = oracle.bias_act(torch.from_numpy(oracle.conv3x3(x[0:1,:,0:7].double().cpu().numpy(), w.double().cpu().numpy()))[:,:,0:6], b.double().cpu(), act='lrelu', alpha=0.2, gain=2**0.5)
    y = fused_conv_act._launch_fused(x, w, None, None, b, 3, 0.2, 2 ** 0.5, -1.0)
    print('   oracle slab err', ((y[0:1,:,0:6].double().cpu()-ref0).abs().max()/ref0.abs().max()).item())
