# round 4, GPU call 27: producers of the stride-1 weight-gradient kernel write channel rows c and c + 4 from one 8-lane group (no 2-way LDS store conflict): tests, A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_conv_wrw_gpu.py tests/test_conv_bench_shapes_gpu.py -x -q -m gpu 2>&1 | tail -2
OFF="--cpu-seconds 0 --strict-steps 0 --bf16-steps 0 --pl-steps 0 --lowp-steps 0 --split3-steps 0 --ada-steps 0"
for perm in 0 1 0 1; do
  SGV_WRW_LANE_PERM=$perm timeout 300 python bench.py $OFF --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r04_c27_bench_perm$perm.json
  python - $perm <<'PY'
import json, sys
d = json.load(open('gpurun_out/r04_c27_bench_perm%s.json' % sys.argv[1]))
k = d['kernels_by_variant']
print('perm', sys.argv[1], 'value', round(d['value'], 1), 'no_prof', round(d['value_no_prof'], 1), {n: (round(k[n]['ms_per_step'], 2), round(k[n]['frac_of_ceiling'], 3)) for n in ('wrw_s1_ws', 'wrw_s1_ws_scaled', 'wrw_s1_ws_packed')})
PY
done
