# round 3, GPU call 14: 16-bit tensor I/O stage 2 (stride-2 family + fused layer nodes): parity tests, network tests, --lowp bench + kernel table
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_conv_lowp_gpu.py tests/test_networks.py tests/test_fused_conv_gpu.py -m gpu -q -s --timeout 900 > gpurun_out/r03_t14.log 2>&1; echo "pytest rc=$?"; grep -v amdgpu.ids gpurun_out/r03_t14.log | grep -E "passed|failed|FAILED|Error|rel-L2|differ" | cut -c1-300 | tail -40
for lp in bf16 fp16; do
timeout 400 python bench.py --cpu-seconds 0 --lowp $lp --strict-steps 0 --bf16-steps 0 --pl-steps 0 2> gpurun_out/r03_b14_$lp.err | tail -1 > gpurun_out/r03_b14_$lp.json; echo "bench $lp rc=$?"; cut -c1-200 gpurun_out/r03_b14_$lp.json; tail -3 gpurun_out/r03_b14_$lp.err
python - $lp <<'PY'
import json, sys
d = json.load(open('gpurun_out/r03_b14_%s.json' % sys.argv[1]))
print('value', d['value'], 'no_prof', d.get('value_no_prof'), 'ada', d.get('value_aug_ada'), d['dtype'])
for k, v in d['kernels'].items():
    print('%-18s %5d launches %8.2f ms  %s' % (k, v['launches'], v['ms_total'], ' '.join('%s=%.1f' % (a, v[a]) for a in ('GBps', 'TFLOPs') if a in v)))
PY
done
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof14 -- python $GRAFT_REPO_ROOT/bench.py --cpu-seconds 0 --lowp bf16 --steps 8 --warmup 2 --strict-steps 0 --bf16-steps 0 --pl-steps 0 --ada-steps 0 --clean-steps 0 --no-prof > /tmp/prof14.log 2>&1
cd $GRAFT_REPO_ROOT
find /tmp/prof14 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r03_lowp2_bf16_kernel_stats.csv
grep '"metric"' /tmp/prof14.log | cut -c1-160
head -32 gpurun_out/r03_lowp2_bf16_kernel_stats.csv | cut -c1-180
