# round 3, GPU call 12: 16-bit tensor I/O of the stride-1 conv / weight-gradient kernels: parity tests, network tests, --lowp bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_conv_lowp_gpu.py tests/test_networks.py tests/test_conv2d_gradfix.py -m gpu -q -s --timeout 600 > gpurun_out/r03_t12.log 2>&1; echo "pytest rc=$?"; grep -v amdgpu.ids gpurun_out/r03_t12.log | grep -E "passed|failed|FAILED|AssertionError|error vs|worst" | cut -c1-260 | tail -60
for lp in bf16 fp16; do
timeout 400 python bench.py --cpu-seconds 0 --lowp $lp --strict-steps 0 --bf16-steps 0 --pl-steps 0 2> gpurun_out/r03_b12_$lp.err | tail -1 > gpurun_out/r03_b12_$lp.json; echo "bench $lp rc=$?"; cut -c1-200 gpurun_out/r03_b12_$lp.json
python - $lp <<'PY'
import json, sys
d = json.load(open('gpurun_out/r03_b12_%s.json' % sys.argv[1]))
print('value', d['value'], 'no_prof', d.get('value_no_prof'), 'ada', d.get('value_aug_ada'), d['dtype'])
print('variants', {k: v for k, v in d.get('kernel_variants', {}).items() if v})
for k, v in d['kernels'].items():
    print('%-18s %5d launches %8.2f ms  %s' % (k, v['launches'], v['ms_total'], ' '.join('%s=%.1f' % (a, v[a]) for a in ('GBps', 'TFLOPs') if a in v)))
PY
done
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof12 -o lowp -- python $GRAFT_REPO_ROOT/bench.py --cpu-seconds 0 --lowp bf16 --steps 4 --warmup 2 --strict-steps 0 --bf16-steps 0 --pl-steps 0 --ada-steps 0 --clean-steps 0 > /dev/null 2>&1
f=$(find /tmp/prof12 -name "*kernel_stats.csv" | head -1); echo "stats: $f"; head -40 "$f" | cut -c1-200 > $GRAFT_REPO_ROOT/gpurun_out/r03_lowp_kernel_stats.csv; head -30 $GRAFT_REPO_ROOT/gpurun_out/r03_lowp_kernel_stats.csv
