# round 6, call 18: zero arenas for the small accumulation buffers: parity suites (whole -m gpu), census of a captured main iteration, captured bench at 32 / 8 videos
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_c18
mkdir -p $OUT
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 > $OUT/pytest.log; tail -3 $OUT/pytest.log
( cd /tmp && SGV_SELFTEST=0 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/census -- python $GRAFT_REPO_ROOT/tools/captured_census.py > /tmp/census.log 2>&1 ); tail -1 /tmp/census.log
f=$(find /tmp/census -name "*kernel_trace.csv" | head -1); python tools/captured_census_report.py $f > $OUT/captured_census.txt 2>&1; head -16 $OUT/captured_census.txt | cut -c1-190
OFF="--strict-steps 0 --bf16-steps 0 --pl-steps 0 --ada-steps 0 --lowp-steps 0 --split3-steps 0 --graph-steps 0"
timeout 600 python bench.py --cpu-seconds 0 $OFF --steps 20 --warmup 5 > $OUT/bench_b32.json 2> $OUT/bench_b32.err
timeout 600 python bench.py --batch-gpu 8 --cpu-seconds 0 $OFF --steps 20 --warmup 5 > $OUT/bench_b8.json 2> $OUT/bench_b8.err
python - <<'PY'
import json
for f in ('b32','b8'):
    d=json.loads(open(f'gpurun_out/r06_c18/bench_{f}.json').read().strip().splitlines()[-1])
    print(f, 'value', d['value'], 'ms', d['ms_per_step'], 'eager', d.get('value_eager'), 'mode', d['config'].get('headline_mode'), 'ufd', d['roofline_upfirdn2d']['frac'], 'power', d.get('power'))
PY
