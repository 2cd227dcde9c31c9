# round 4, GPU call 22: events inside the timed region for the two roofline families only, full tables in a pass of their own: value against value_no_prof
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
OFF="--cpu-seconds 0 --strict-steps 0 --bf16-steps 0 --pl-steps 0 --lowp-steps 0 --split3-steps 0 --ada-steps 0"
timeout 400 python bench.py $OFF --steps 20 --warmup 5 > gpurun_out/r04_c22_bench.json 2> gpurun_out/r04_c22_bench.err; echo "rc=$?"
tail -3 gpurun_out/r04_c22_bench.err
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r04_c22_bench.json') if l.startswith('{')][-1])
print('value', round(d['value'], 1), 'ms', round(d['ms_per_step'], 2), 'no_prof', round(d['value_no_prof'], 1))
r = d['roofline']; print('roofline', r['kernel'][:40], round(r['achieved'], 1), round(r['frac'], 3), r['launches'], round(r['avg_launch_us'], 1))
u = d['roofline_upfirdn2d']; print('ufd', round(u['achieved']), round(u['frac'], 3), u['launches'])
f = d['roofline_conv_family']; print('family', round(f['achieved'], 1), round(f['frac'], 3), f['launches'])
k = d['kernels_by_variant']; print('variants', len(k), {n: round(k[n]['ms_per_step'], 2) for n in list(k)[:6]})
print('kernels', {n: (e['launches'], round(e['ms_total'], 1)) for n, e in d['kernels'].items()})
PY
