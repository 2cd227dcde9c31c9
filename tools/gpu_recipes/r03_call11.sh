# round 3, GPU call 11: tile kernel with NT / F44 template parameters: op tests, lab, bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_fused_bench_shapes_gpu.py tests/test_extras_gpu.py tests/test_fused_conv_gpu.py tests/test_fc_gpu.py -m gpu -q --timeout 600 > gpurun_out/r03_t11.log 2>&1; echo "pytest rc=$?"; grep -v amdgpu.ids gpurun_out/r03_t11.log | grep -E "passed|failed|FAILED|AssertionError" | cut -c1-260 | tail -20
{
for cfg in "32 257 1" "96 257 1" "96 256 2"; do
    echo "== N IH pad: $cfg"
    timeout 120 tools/ufd_lab $cfg 2>&1 | grep -E "libsgv|copy2 ntl1|V6 LDS tile, loads up front, 16 rows NT0|V7|mismatch"
done
} > gpurun_out/r03_ufd_lab11.log 2>&1
cat gpurun_out/r03_ufd_lab11.log
timeout 500 python bench.py --cpu-seconds 0 2> gpurun_out/r03_b11.err | tail -1 > gpurun_out/r03_b11.json; echo "bench rc=$?"; cut -c1-200 gpurun_out/r03_b11.json
python - <<'PY'
import json
d = json.load(open('gpurun_out/r03_b11.json'))
print('value', d['value'], 'no_prof', d.get('value_no_prof'), 'strict', d.get('value_strict_fp32'), 'ada', d.get('value_aug_ada'), 'bf16p', d.get('value_bf16_products'), 'pl_f1', d.get('value_pl_f1'))
print('roofline', d['roofline']['frac'], 'ufd', d['roofline_upfirdn2d']['frac'], d['roofline_upfirdn2d']['achieved'])
for r in d['upfirdn2d_by_size'][:12]:
    print('  ufd %9.1f MB x%4d  %8.1f us  %7.1f GB/s  %5.1f%% of family time' % (r['algorithmic_MB'], r['launches'], r['avg_us'], r['GBps'], 100 * r['share_of_family_time']))
for k, v in d['kernels'].items():
    print('%-18s %5d launches %8.2f ms  %s' % (k, v['launches'], v['ms_total'], ' '.join('%s=%.1f' % (a, v[a]) for a in ('GBps', 'TFLOPs') if a in v)))
PY
