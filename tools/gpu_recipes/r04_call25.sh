# round 4, GPU call 25: load-time self-test of the asm-load kernels: its test, its cost at process start, the step (kernel sources unchanged)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_extras_gpu.py -x -q -m gpu -k "selftest or capture_iteration or path_length" 2>&1 | tail -3
python - <<'PY'
import time, torch
from stylegan_v_amd.torch_utils.ops import selftest
torch.zeros(1, device='cuda'); torch.cuda.synchronize()
t = time.perf_counter(); r = selftest.run(torch.device('cuda', 0)); torch.cuda.synchronize()
print('selftest', r, round(time.perf_counter() - t, 2), 's')
PY
OFF="--cpu-seconds 0 --strict-steps 0 --bf16-steps 0 --pl-steps 0 --lowp-steps 0 --split3-steps 0 --ada-steps 0"
timeout 300 python bench.py $OFF --steps 20 --warmup 5 2> gpurun_out/r04_c25_bench.err | tail -1 > gpurun_out/r04_c25_bench.json; grep "warm-up iteration 0" gpurun_out/r04_c25_bench.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/r04_c25_bench.json'))
print('value', round(d['value'], 1), 'ms', round(d['ms_per_step'], 2), 'no_prof', round(d['value_no_prof'], 1), 'traffic', d['roofline']['traffic'])
PY
