# round 3, GPU call 5: GEMM debug, network-golden error report (default + strict), WIDE tile kernel in the lab harness, g1024 workload
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python tools/gemm_dbg.py 2>&1 | grep -v amdgpu.ids | tail -30
timeout 900 python -m pytest tests/test_fused_bench_shapes_gpu.py tests/test_networks.py tests/test_extras_gpu.py tests/test_fc_gpu.py -m gpu -q --timeout 600 -s > gpurun_out/r03_t5.log 2>&1; echo "pytest rc=$?"; grep -v amdgpu.ids gpurun_out/r03_t5.log | grep -E "passed|failed|FAILED|AssertionError|networks_full|outside their bound|max error" | cut -c1-260 | tail -60
{
for cfg in "32 257 1" "96 257 1" "96 256 2"; do
    echo "== N IH pad: $cfg"
    timeout 120 tools/ufd_lab $cfg 2>&1 | grep -E "libsgv|copy2 ntl1|V6 LDS tile, loads up front, 16 rows NT0|mismatch"
done
} > gpurun_out/r03_ufd_lab5.log 2>&1
cat gpurun_out/r03_ufd_lab5.log
timeout 300 python bench.py --workload g1024 --steps 10 --warmup 3 --cpu-seconds 20 2> gpurun_out/r03_g1024.err | tail -1 > gpurun_out/r03_g1024.json; echo "g1024 rc=$?"; cut -c1-400 gpurun_out/r03_g1024.json; tail -3 gpurun_out/r03_g1024.err
python - <<'PY'
import json
try:
    d = json.load(open('gpurun_out/r03_g1024.json'))
    print('g1024 value', d['value'], 'no_prof', d['value_no_prof'], 'roofline', d['roofline'] and d['roofline']['frac'], 'cpu', d['cpu_baseline'])
    for r in d['upfirdn2d_by_size']:
        print('  ufd %9.1f MB x%4d  %8.1f us  %7.1f GB/s  %5.1f%%' % (r['algorithmic_MB'], r['launches'], r['avg_us'], r['GBps'], 100 * r['share_of_family_time']))
    for k, v in d['kernels'].items():
        print('%-18s %5d launches %8.2f ms  %s' % (k, v['launches'], v['ms_total'], ' '.join('%s=%.1f' % (a, v[a]) for a in ('GBps', 'TFLOPs') if a in v)))
except Exception as e:
    print('g1024 parse failed', e)
PY
