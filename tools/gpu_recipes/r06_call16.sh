# round 6, call 16: grouped style affines (sgv_fc_grouped): parity suites, captured census, bench at 32 and 8 videos
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_c16
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_fc_gpu.py tests/test_networks.py tests/test_extras_gpu.py tests/test_abi.py -x -q -m gpu 2>&1 | tail -12 > $OUT/pytest.log; tail -5 $OUT/pytest.log
OFF="--strict-steps 0 --bf16-steps 0 --pl-steps 0 --ada-steps 0 --lowp-steps 0 --split3-steps 0 --graph-steps 0"
timeout 600 python bench.py --cpu-seconds 0 $OFF --steps 20 --warmup 5 > $OUT/bench_b32.json 2> $OUT/bench_b32.err
timeout 600 python bench.py --batch-gpu 8 --cpu-seconds 0 $OFF --steps 20 --warmup 5 > $OUT/bench_b8.json 2> $OUT/bench_b8.err
python - <<'PY'
import json
for f in ('b32','b8'):
    d=json.loads(open(f'gpurun_out/r06_c16/bench_{f}.json').read().strip().splitlines()[-1])
    print(f, 'value', d['value'], 'ms', d['ms_per_step'], 'eager', d.get('value_eager'), 'mode', d['config'].get('headline_mode'), 'power', d.get('power'))
PY
