# round 3, GPU call 23: persistent bf16x3 GEMM (gemm_bf16x3_stream_kernel): its tests, per-shape table against the one-workgroup-per-tile member
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_extras_gpu.py -m gpu -q -x -k "gemm" --timeout 200 2>&1 | grep -v amdgpu.ids | tail -5
{
echo "== persistent (default)"; timeout 200 python tools/gemm_bench.py 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for r in d[0]['rows']: print('%-55s %-16s %.3f ms %7.1f TF/s %7.1f GB/s' % (r['layer'], r['op'], r['ms'], r['TFLOPs'], r['GBps']))
print({k:v for k,v in d[0]['variants'].items() if v})"
echo "== SGV_GEMM_STREAM=0"; SGV_GEMM_STREAM=0 timeout 200 python tools/gemm_bench.py 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for r in d[0]['rows']: print('%-55s %-16s %.3f ms %7.1f TF/s %7.1f GB/s' % (r['layer'], r['op'], r['ms'], r['TFLOPs'], r['GBps']))
print({k:v for k,v in d[0]['variants'].items() if v})"
} > gpurun_out/r03_gemm_stream_ab.log 2>&1
cat gpurun_out/r03_gemm_stream_ab.log
