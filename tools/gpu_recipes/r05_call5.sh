# round 5, GPU call 5: 4 x 4 layers on conv3x3_small_kernel (+ partly filled tiles, zero-padded channels), whole-tensor gradient sketches of the full-size golden,
# FIR tap sums remembered on the caller's filter (the aug=ada regression of call 4), then the driver's bench command
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
SGV_ERROR_TABLE_DIR=gpurun_out timeout 1500 python -m pytest tests/ -q -m gpu --maxfail=40 -rf -s 2>&1 > gpurun_out/r05_c5_pytest_full.log; grep "networks_full on cuda\] whole\|fp16 golden\] [0-9i]" gpurun_out/r05_c5_pytest_full.log | cut -c1-330; grep -v "^\[\|^\.\|^$\|^terms" gpurun_out/r05_c5_pytest_full.log | tail -40
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_c5_bench.json 2> gpurun_out/r05_c5_bench.err; echo "bench rc=$?"
cp bench_detail.json gpurun_out/r05_c5_bench_detail.json
python - <<'PY'
import json
d = json.load(open('gpurun_out/r05_c5_bench_detail.json'))
print({k: round(v, 1) for k, v in d.items() if k.startswith('value') and isinstance(v, float)}, 'ms', round(d['ms_per_step'], 2), 'roofline', round(d['roofline']['frac'], 3), 'ufd', round(d['roofline_upfirdn2d']['frac'], 3))
for k, v in d['kernels_by_variant'].items():
    if k in ('conv_small', 'gemm_bf16x3_stream', 'conv1x1_wstat', 'gemm_bf16x3', 'convT_ws', 'conv_s1_ws_fused'): print(k, v['launches'], round(v['ms_per_step'], 2), round(v.get('TFLOPs', 0)), round(v.get('GBps', 0)))
PY
