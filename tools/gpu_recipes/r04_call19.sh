# round 4, GPU call 19: dense-layer kernel with all k steps of a wave in flight for K <= 512 (one latency per launch): A/B in the step, 32 and 8 videos
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_fc_gpu.py -x -q -m gpu 2>&1 | tail -2
OFF="--cpu-seconds 0 --strict-steps 0 --bf16-steps 0 --pl-steps 0 --lowp-steps 0 --split3-steps 0 --ada-steps 0"
for u in 1 auto; do
  for b in 32; do
    if [ $u = 1 ]; then export SGV_FC_UNROLL=1; else unset SGV_FC_UNROLL; fi
    timeout 300 python bench.py $OFF --batch-gpu $b --steps 20 --warmup 5 > gpurun_out/r04_c19_bench_u${u}_b$b.json 2> gpurun_out/r04_c19_bench_u${u}_b$b.err
    python - $u $b <<'PY'
import json, sys
d = json.loads([l for l in open('gpurun_out/r04_c19_bench_u%s_b%s.json' % (sys.argv[1], sys.argv[2])) if l.startswith('{')][-1])
k = d['kernels_by_variant']['fc']
print('unroll', sys.argv[1], 'batch', sys.argv[2], 'value', round(d['value'], 1), 'ms', round(d['ms_per_step'], 2), 'no_prof', round(d['value_no_prof'], 1), 'fc ms', round(k['ms_per_step'], 2), 'avg us', round(k['avg_us'], 1))
PY
  done
done
