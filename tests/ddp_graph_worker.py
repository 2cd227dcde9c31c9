"""Child process of test_train_step_graphs_with_ddp_and_ada_on_an_nccl_group_of_one (tests/test_extras_gpu.py): config 4's regime -- DDP over an
RCCL process group, hipGraph replay and aug=ada together -- in its own process, so that a crash inside graph capture cannot take the test session
down with it.  Prints OK on success."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', sys.argv[1] if len(sys.argv) > 1 else '29533')

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import stylegan_v_amd  # noqa: E402
from stylegan_v_amd.torch_utils import custom_ops  # noqa: E402
from stylegan_v_amd.training import config as cfgs  # noqa: E402
from stylegan_v_amd.training.train_step import TrainStep  # noqa: E402


def make(**kw):
    g_kwargs, d_kwargs = cfgs.small_test_model_kwargs(res=32)
    train_cfg = cfgs.Config(r1_gamma=1.0, lr=0.0025, betas=(0.0, 0.99), ema_kimg=1.0, ema_rampup=0.05, G_reg_interval=4, D_reg_interval=16, pl_weight=0.0)
    return TrainStep(g_kwargs, d_kwargs, train_cfg, device='cuda', batch_gpu=4, world_size=1, **kw)


def main():
    stylegan_v_amd.configure_miopen()
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    ts = make(ddp=True, use_graphs=True, augment='ada')
    assert ts.ddp and ts.ddp_manual and ts.use_graphs and ts.augment_pipe is not None and ts.augment_pipe.static_margin
    assert ts.step() == ['Gmain', 'Greg', 'Dmain', 'Dreg']
    torch.cuda.synchronize()
    for phase in ts.phases:
        for st in phase['opt'].state.values():
            assert float(st['step']) in (1.0, 2.0), 'capture iteration = one optimiser step per phase run'
    launches = custom_ops.launch_count()
    ts.augment_pipe.p.fill_(0.3)        # augmentations really drawn inside the replayed graphs
    before = {k: v.detach().clone() for k, v in ts.D.named_parameters()}
    for _ in range(3):
        assert ts.step() == ['Gmain', 'Dmain']
    torch.cuda.synchronize()
    assert set(ts._graphs) == {'Gmain', 'Dmain'}
    assert custom_ops.launch_count() == launches, 'replayed phases must not launch native kernels from the host'
    assert sum(int(not torch.equal(v, before[k])) for k, v in ts.D.named_parameters()) > 10
    for name, p in list(ts.G.named_parameters()) + list(ts.D.named_parameters()):
        assert torch.isfinite(p).all(), name
    assert torch.isfinite(ts.last_losses['D/loss']) and 'signs_real' in ts.last_losses
    # the eager DDP schedule on the same kind of instance reaches the same loss range (same models, same data distribution)
    te = make(ddp=True, use_graphs=False, augment='ada')
    te.step(); te.step()
    assert abs(float(te.last_losses['D/loss']) - float(ts.last_losses['D/loss'])) < 1.0
    dist.destroy_process_group()
    print('OK')


if __name__ == '__main__':
    main()
