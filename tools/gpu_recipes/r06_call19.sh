# round 6, call 19: x * styles backward as one pass + the deferred ToRGB (its gradient summed inside the next block's conv0 backward), NaN-preserving fp16 operand
# clamp: whole -m gpu suite, then A / B of the eager + captured step with SGV_DEFER_RGB=0 / 1 on one box
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_c19
mkdir -p $OUT
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > $OUT/pytest.log; tail -4 $OUT/pytest.log
OFF="--strict-steps 0 --bf16-steps 0 --pl-steps 0 --ada-steps 0 --lowp-steps 0 --split3-steps 0 --graph-steps 0"
for v in 1 0 1 0; do
  SGV_DEFER_RGB=$v timeout 600 python bench.py --cpu-seconds 0 $OFF --steps 20 --warmup 5 > $OUT/bench_defer$v.json 2> $OUT/bench_defer$v.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r06_c19/bench_defer$v.json').read().strip().splitlines()[-1])
dd=json.load(open('bench_detail.json'))
kv=dd['kernels_by_variant']
print('defer=$v value', d['value'], 'ms', d['ms_per_step'], 'eager', d.get('value_eager'), 'mode', d['config'].get('headline_mode'), 'sclk', d['power']['sclk_MHz'], 'modulate', round(kv['modulate']['ms_per_step'],3), 'step_ms', dd['step_ms'][3:8])
PY
done
