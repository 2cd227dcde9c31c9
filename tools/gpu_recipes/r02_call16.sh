# stride-2 weight gradient, producer / consumer form: tests + bench A/B against the 4-wave kernel
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_conv_wrw_gpu.py tests/test_conv_bench_shapes_gpu.py tests/test_fused_conv_gpu.py tests/test_networks.py -x -q -m gpu 2>&1 | tail -4
timeout 200 python tools/wrw_small_bench.py 2>&1 | grep -v amdgpu.ids
for i in 1 2; do
  SGV_WRW_S2_WS=0 timeout 300 python bench.py --steps 12 --warmup 4 --cpu-seconds 0 --strict-steps 0 --ada-steps 0 --bf16-steps 0 --no-prof 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('4-wave s2 wrw', d['value'], d['ms_per_step'])"
  timeout 300 python bench.py --steps 12 --warmup 4 --cpu-seconds 0 --strict-steps 0 --ada-steps 0 --bf16-steps 0 --no-prof 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ws s2 wrw    ', d['value'], d['ms_per_step'])"
done
