# round 4, GPU call 24: dense layers read row-strided inputs (ws[:, i]) in place: tests, the step
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_fc_gpu.py tests/test_networks.py -x -q -m gpu 2>&1 | tail -2
OFF="--cpu-seconds 0 --strict-steps 0 --bf16-steps 0 --pl-steps 0 --lowp-steps 0 --split3-steps 0 --ada-steps 0"
timeout 300 python bench.py $OFF --steps 20 --warmup 5 > gpurun_out/r04_c24_bench.json 2> gpurun_out/r04_c24_bench.err
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r04_c24_bench.json') if l.startswith('{')][-1])
print('value', round(d['value'], 1), 'ms', round(d['ms_per_step'], 2), 'no_prof', round(d['value_no_prof'], 1), 'traffic', d['roofline']['traffic'])
PY
