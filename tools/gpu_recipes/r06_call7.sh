# round 6, call 7: mode 2 (act-gradient prologue) on the tile kernel: fused / network parity suites, then the driver's bench command
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_c7
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_extras_gpu.py tests/test_fused_bench_shapes_gpu.py tests/test_fused_conv_gpu.py tests/test_networks.py -x -q -m gpu 2>&1 | tail -15 > $OUT/pytest.log; cat $OUT/pytest.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err; echo "bench rc=$?"
cp bench_detail.json $OUT/bench_driver_cmd_detail.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_c7/bench_driver_cmd.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'eager', d.get('value_eager'), 'ada', d.get('value_aug_ada'), 'power', d.get('power'))
print('roofline', d['roofline']['frac'], 'ufd', d['roofline_upfirdn2d'])
dd=json.load(open('gpurun_out/r06_c7/bench_driver_cmd_detail.json'))
for k,v in sorted(dd['kernels_by_variant'].items(), key=lambda kv:-kv[1].get('ms_per_step',0)):
    if k.startswith('ufd') or k in ('modulate','bias_act','absmax','fc'): print(k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items()})
PY
