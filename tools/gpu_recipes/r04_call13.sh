# round 4, GPU call 13: the geometric execution of the ADA pipeline as one forward kernel: parity tests, then the ada companion next to the plain step
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_augment.py -x -q -m gpu 2>&1 | tail -15 | tee gpurun_out/r04_c13_tests.log
OFF="--cpu-seconds 0 --strict-steps 0 --bf16-steps 0 --pl-steps 0 --lowp-steps 0 --split3-steps 0"
timeout 400 python bench.py $OFF --ada-steps 12 --steps 12 --warmup 4 > gpurun_out/r04_c13_bench.json 2> gpurun_out/r04_c13_bench.err; echo "rc=$?"
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r04_c13_bench.json') if l.startswith('{')][-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'no_prof', d.get('value_no_prof'), 'ada', d['aug_ada'])
PY
