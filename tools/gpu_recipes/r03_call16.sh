# round 3, GPU call 16: default bench line with the mixed-precision companion and the per-variant table
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python bench.py --cpu-seconds 0 2> gpurun_out/r03_b16.err | tail -1 > gpurun_out/r03_b16.json; echo "bench rc=$?"; tail -3 gpurun_out/r03_b16.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/r03_b16.json'))
print('value', d['value'], 'no_prof', d.get('value_no_prof'), 'strict', d.get('value_strict_fp32'), 'ada', d.get('value_aug_ada'), 'bf16p', d.get('value_bf16_products'), 'pl_f1', d.get('value_pl_f1'), 'lowp', d.get('value_lowp_bf16'))
print('roofline', d['roofline']['frac'], 'ufd', d['roofline_upfirdn2d']['frac'])
for k, v in d['kernels_by_variant'].items():
    print('%-28s %5d calls %7.2f ms/step %8.1f us  %5.1f%%  %s' % (k, v['launches'], v['ms_per_step'], v['avg_us'], 100 * v['share_of_native_time'], ' '.join('%s=%.3g' % (a, v[a]) for a in ('TFLOPs', 'frac_of_ceiling', 'GBps') if a in v)))
PY
