# round 3, GPU call 30: 16-bit FIR tile kernel with eight columns per lane: op tests (bit-exact vs the oracle, all dtypes), mixed-precision bench A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_fused_bench_shapes_gpu.py tests/test_conv_lowp_gpu.py -m gpu -q -x --timeout 600 2>&1 | grep -v amdgpu.ids | tail -4
B="python bench.py --cpu-seconds 0 --strict-steps 0 --bf16-steps 0 --pl-steps 0 --ada-steps 0 --lowp-steps 0 --clean-steps 0 --steps 10 --warmup 3 --lowp bf16"
for sw in 1 0 1 0; do
SGV_UFD_TILE_CPL8=$sw timeout 200 $B 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
kv=d['kernels_by_variant']
print('cpl8=$sw', round(d['value'],1), round(d['ms_per_step'],2), {k: (round(v['ms_per_step'],2), round(v['GBps'])) for k,v in kv.items() if k.startswith('ufd')})"
done | tee gpurun_out/r03_ufd_tile_cpl8_ab.log
