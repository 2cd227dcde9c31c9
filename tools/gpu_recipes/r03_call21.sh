# round 3, GPU call 21: convolution / fused / network test files under the fallback switches, failing test names kept
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
{
timeout 300 python -m pytest tests/test_conv3x3_gpu.py tests/test_conv_wrw_gpu.py -m gpu -q --timeout 200 2>&1 | grep -v amdgpu.ids | tail -3
for sw in SGV_S2_WS=0 SGV_CONV_WS=0 SGV_WRW_WS=0 SGV_WRW_S2_WS=0 SGV_FUSED_CONV=0 SGV_UFD_TILE=0 SGV_CONV_LOWP=0 "SGV_CONV_TERMS=0 SGV_WRW_TERMS=0"; do
  echo "== $sw"
  files="tests/test_conv3x3_gpu.py tests/test_conv_wrw_gpu.py tests/test_fused_conv_gpu.py tests/test_conv_lowp_gpu.py tests/test_networks.py"
  if [ "$sw" = "SGV_UFD_TILE=0" ]; then files="$files tests/test_ops_gpu.py tests/test_fused_bench_shapes_gpu.py"; fi
  env $sw timeout 600 python -m pytest $files -m gpu -q --timeout 500 2>&1 | grep -v amdgpu.ids | grep -E "^FAILED|^E  .*(Error|assert)|passed|failed" | cut -c1-230 | head -40
done
} > gpurun_out/r03_fallback_switches_tests.log 2>&1
cat gpurun_out/r03_fallback_switches_tests.log
