# round 3, GPU call 3: upfirdn2d_tile_kernel in the product (op tests bit-exact vs the oracle, fused bench-shape tests), the rest of the suite
# with the DDP + graphs test in a child process, lab line of the product, bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_fused_bench_shapes_gpu.py -m gpu -q --timeout 600 > gpurun_out/r03_t3a.log 2>&1; echo "ops + fused bench shapes rc=$?"; grep -v amdgpu.ids gpurun_out/r03_t3a.log | tail -30
timeout 900 python -m pytest tests -m gpu -q --timeout 600 --deselect tests/test_ops_gpu.py --deselect tests/test_fused_bench_shapes_gpu.py > gpurun_out/r03_t3b.log 2>&1; echo "rest rc=$?"; grep -v amdgpu.ids gpurun_out/r03_t3b.log | tail -60
{
for cfg in "32 257 1" "96 257 1" "96 256 2"; do
    echo "== N IH pad: $cfg"
    timeout 120 tools/ufd_lab $cfg 2>&1 | grep -E "libsgv|copy2 ntl1|V6 LDS tile, loads up front, 16 rows NT0|mismatch"
done
} > gpurun_out/r03_ufd_lab3.log 2>&1
cat gpurun_out/r03_ufd_lab3.log
timeout 500 python bench.py --cpu-seconds 0 2> gpurun_out/r03_b3.err | tail -1 > gpurun_out/r03_b3.json; echo "bench rc=$?"; cut -c1-200 gpurun_out/r03_b3.json
python - <<'PY'
import json
d = json.load(open('gpurun_out/r03_b3.json'))
print('value', d['value'], 'no_prof', d.get('value_no_prof'), 'strict', d.get('value_strict_fp32'), 'ada', d.get('value_aug_ada'), 'bf16p', d.get('value_bf16_products'), 'pl_f1', d.get('value_pl_f1'))
print('roofline', d['roofline']['frac'], 'ufd', d['roofline_upfirdn2d']['frac'], d['roofline_upfirdn2d']['achieved'])
for r in d['upfirdn2d_by_size']:
    print('  ufd %9.1f MB x%4d  %8.1f us  %7.1f GB/s  %5.1f%% of family time' % (r['algorithmic_MB'], r['launches'], r['avg_us'], r['GBps'], 100 * r['share_of_family_time']))
for k, v in d['kernels'].items():
    print('%-18s %5d launches %8.2f ms  %s' % (k, v['launches'], v['ms_total'], ' '.join('%s=%.1f' % (a, v[a]) for a in ('GBps', 'TFLOPs') if a in v)))
PY
timeout 300 python bench.py --cpu-seconds 0 --batch-gpu 8 --steps 20 --warmup 3 --strict-steps 0 --ada-steps 0 --bf16-steps 0 --pl-steps 0 --graphs 2> gpurun_out/r03_b3_b8g.err | tail -1 > gpurun_out/r03_b3_b8g.json; echo "batch8 graphs rc=$?"; cut -c1-200 gpurun_out/r03_b3_b8g.json
