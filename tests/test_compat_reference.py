"""Drop-in check against the real reference checkout (build container only: skipped where /root/reference is absent).
The reference's own Generator/Discriminator are imported UNMODIFIED with this package's op layer aliased under
`src.torch_utils.ops.*`; their outputs must equal the golden fixtures (which were produced with the reference's ops)."""
import os
import subprocess
import sys

import pytest

REF = '/root/reference'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import os, sys, json
sys.dont_write_bytecode = True
ROOT, REF = sys.argv[1], sys.argv[2]
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden', '_shims'), REF, os.path.join(REF, 'src')]
import torch
import stylegan_v_amd.compat as compat
mapping = compat.install()
from src.torch_utils.ops import upfirdn2d as U, bias_act as B
import stylegan_v_amd.torch_utils.ops.upfirdn2d as mine
assert U is mine, 'alias not in effect'
from training.networks import Generator, Discriminator      # reference modules
import training.networks as RN
assert RN.upfirdn2d is mine and RN.bias_act is B
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import make_golden
from util import Golden, assert_close
NET = Golden('networks')
G, D = make_golden.build_small_models()
G.load_state_dict({k[2:]: NET.t(k) for k in NET.keys('G.')})
D.load_state_dict({k[2:]: NET.t(k) for k in NET.keys('D.')})
z, t, mz = NET.t('z'), NET.t('t'), NET.t('motion_z')
c = torch.zeros([z.shape[0], 0])
G.train(); D.train()
img = G.synthesis(G.mapping(z, c, skip_w_avg_update=True), t=t, c=c, motion_z=mz)
assert_close(img, NET.t('img_train'), atol=1e-5, rtol=1e-5, what='reference G on this op layer')
assert_close(D(NET.t('img_train'), c, t)['image_logits'], NET.t('logits_fake'), atol=1e-5, rtol=1e-5, what='reference D on this op layer')
print('COMPAT-OK', len(mapping))
'''


@pytest.mark.skipif(not os.path.isdir(REF), reason='reference checkout not present (GPU box)')
def test_reference_modules_run_unmodified_on_this_op_layer():
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1')
    res = subprocess.run([sys.executable, '-c', SCRIPT, ROOT, REF], capture_output=True, text=True, env=env, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert 'COMPAT-OK' in res.stdout
