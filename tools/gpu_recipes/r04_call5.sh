# round 4, GPU call 5: bounds as by-products of the producing kernels (sgv_amax_sink) + the streaming absmax kernel: tests, then the step with
# terms = 4 against terms = 3 on one box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04_tables
export TMPDIR=/tmp SGV_ERROR_TABLE_DIR=$GRAFT_REPO_ROOT/gpurun_out/r04_tables
python -c "from stylegan_v_amd.torch_utils import custom_ops as c; import sys; sys.exit(0 if c.is_built() else 1)" || { echo "in-tree library is stale: stop"; exit 1; }
timeout 420 python -m pytest tests/test_conv_f16split_gpu.py tests/test_conv_wrw_gpu.py tests/test_extras_gpu.py tests/test_fused_conv_gpu.py tests/test_pointwise_gpu.py -m gpu -q --timeout 200 > gpurun_out/r04_c5_tests.log 2>&1; echo "tests rc=$?"
grep -v amdgpu.ids gpurun_out/r04_c5_tests.log | grep -E "passed|failed|FAILED|Error|error" | cut -c1-300 | tail -25
B="python bench.py --cpu-seconds 0 --strict-steps 0 --bf16-steps 0 --pl-steps 0 --ada-steps 0 --lowp-steps 0 --steps 12 --warmup 3"
timeout 240 $B > gpurun_out/r04_c5_bench_terms4.json 2> gpurun_out/r04_c5_bench_terms4.err; echo "bench4 rc=$?"
SGV_CONV_TERMS=3 SGV_WRW_TERMS=3 timeout 240 $B > gpurun_out/r04_c5_bench_terms3.json 2> gpurun_out/r04_c5_bench_terms3.err; echo "bench3 rc=$?"
python - <<'PY'
import json
for t in (4, 3):
    try:
        d = json.loads(open(f'gpurun_out/r04_c5_bench_terms{t}.json').read().strip().splitlines()[-1])
    except Exception as e:
        print('terms', t, 'no line', e); continue
    print('terms', t, 'value', round(d['value'], 1), 'no_prof', round(d['value_no_prof'] or 0, 1), 'ms', round(d['ms_per_step'], 1), 'roofline', {k: d['roofline'].get(k) for k in ('achieved', 'frac', 'avg_launch_us')})
    kv = d.get('kernels_by_variant') or {}
    for name, v in list(kv.items())[:16]:
        print('   %-28s %6d launches %8.2f ms/step %8.1f us %s' % (name, v['launches'], v['ms_per_step'], v['avg_us'], ('%.0f TF' % v['TFLOPs']) if 'TFLOPs' in v else ('%.0f GB/s' % v.get('GBps', 0))))
    k = d.get('kernels') or {}
    if 'absmax' in k: print('   absmax family', k['absmax'])
PY
