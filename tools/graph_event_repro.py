"""Per-launch event nodes inside a torch.cuda.graph capture: one upfirdn2d call + one 3x3 convolution captured with sgv_prof_resume active, three replays, the records."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stylegan_v_amd.torch_utils import custom_ops
from stylegan_v_amd.torch_utils.ops import upfirdn2d, conv2d_gradfix
dev = torch.device('cuda')
x = torch.randn([8, 64, 129, 129], device=dev)
w = torch.randn([64, 64, 3, 3], device=dev) / 24
f = upfirdn2d.setup_filter([1, 3, 3, 1], device=dev)
for _ in range(2):
    y = conv2d_gradfix.conv2d(upfirdn2d.upfirdn2d(x, f, padding=1), w, padding=1)
torch.cuda.synchronize()
custom_ops.prof_families(('conv3x3_s1', 'upfirdn2d_lanes'))
custom_ops.prof_enable(64)
custom_ops.prof_disable()
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        y = conv2d_gradfix.conv2d(upfirdn2d.upfirdn2d(x, f, padding=1), w, padding=1)
torch.cuda.current_stream().wait_stream(s)
with torch.no_grad(), torch.cuda.graph(g, capture_error_mode='thread_local'):
    custom_ops.prof_resume()
    y = conv2d_gradfix.conv2d(upfirdn2d.upfirdn2d(x, f, padding=1), w, padding=1)
    custom_ops.prof_disable()
for r in range(3):
    g.replay()
    torch.cuda.synchronize()
print('records', custom_ops.prof_collect_records(64, with_variant=True))
