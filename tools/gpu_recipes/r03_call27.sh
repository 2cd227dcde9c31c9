# round 3, GPU call 27: convT edge strips with the input channels split over the waves of a workgroup: tests + bench A/B (SGV_CONVT_EDGE_KSPLIT)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_conv3x3_gpu.py tests/test_conv_lowp_gpu.py tests/test_conv_bench_shapes_gpu.py tests/test_extras_gpu.py -m gpu -q -x --timeout 300 2>&1 | grep -v amdgpu.ids | tail -4
B="python bench.py --cpu-seconds 0 --strict-steps 0 --bf16-steps 0 --pl-steps 0 --ada-steps 0 --lowp-steps 0 --clean-steps 0 --steps 10 --warmup 3"
for sw in 1 0 1 0; do
SGV_CONVT_EDGE_KSPLIT=$sw timeout 200 $B 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
kv=d['kernels_by_variant']
print('ksplit=$sw', round(d['value'],1), round(d['ms_per_step'],2), 'convT_ws us', round(kv['convT_ws']['avg_us'],1), 'packed', round(kv['convT_ws_packed']['avg_us'],1), 'gemm ms/step', round(kv.get('gemm_bf16x3_stream',{}).get('ms_per_step',0)+kv.get('gemm_bf16x3',{}).get('ms_per_step',0),2))"
done | tee gpurun_out/r03_convT_edge_ksplit_ab.log
