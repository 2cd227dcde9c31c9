# round 3, GPU call 32 (second attempt: the first one shipped a tree whose library did not compile -- every test then tried to rebuild it; recipes now stop
# when the in-tree library is stale): tile FIR kernel with packed FMAs and wave-uniform clamp shifts: op tests (incl. the new widths), FIR variants in the bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "from stylegan_v_amd.torch_utils import custom_ops as c; import sys; sys.exit(0 if c.is_built() else 1)" || { echo "in-tree library is stale: stop"; exit 1; }
timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -q -x --timeout 120 2>&1 | grep -v amdgpu.ids | tail -4
timeout 200 python -m pytest tests/test_fused_bench_shapes_gpu.py tests/test_extras_gpu.py -m gpu -q -x --timeout 120 2>&1 | grep -v amdgpu.ids | tail -3
B="python bench.py --cpu-seconds 0 --strict-steps 0 --bf16-steps 0 --pl-steps 0 --ada-steps 0 --lowp-steps 0 --clean-steps 0 --steps 8 --warmup 3"
for mode in "--lowp bf16" ""; do
  timeout 150 $B $mode 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
kv=d['kernels_by_variant']
print('$mode', round(d['value'],1), {k: (round(v['ms_per_step'],2), round(v['GBps'])) for k,v in kv.items() if k.startswith('ufd_tile')})"
done | tee gpurun_out/r03_ufd_tile_pkfma.log
