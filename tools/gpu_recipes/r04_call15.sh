# round 4, GPU call 15: bound side outputs of the lane-exchange FIR kernel and of the stride-1 convolution's store (accumulate: bound(old) + bound(increment)):
# tests, the remaining separate bound passes, the step
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_conv_f16split_gpu.py tests/test_ops_gpu.py tests/test_fused_conv_gpu.py tests/test_extras_gpu.py tests/test_networks.py -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/r04_c15_tests.log
OFF="--cpu-seconds 0 --strict-steps 0 --bf16-steps 0 --pl-steps 0 --lowp-steps 0 --split3-steps 0 --ada-steps 0"
SGV_AMAX_TRACE=1 timeout 300 python bench.py $OFF --no-prof --steps 16 --warmup 16 > /dev/null 2> gpurun_out/r04_c15_trace.err; echo "rc=$?"
grep -A24 "amax trace" gpurun_out/r04_c15_trace.err | cut -c1-160
timeout 400 python bench.py $OFF --steps 20 --warmup 5 > gpurun_out/r04_c15_bench.json 2> gpurun_out/r04_c15_bench.err; echo "rc=$?"
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r04_c15_bench.json') if l.startswith('{')][-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'no_prof', d.get('value_no_prof'))
k = d['kernels_by_variant']
print({n: round(k[n]['ms_per_step'], 2) for n in ('absmax', 'ufd_lanes_fused2', 'conv_s1_ws_accumulate', 'ufd_lanes_seg', 'ufd_lanes') if n in k})
PY
