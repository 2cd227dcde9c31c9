# round 3, GPU call 20: the whole -m gpu suite, smoke(), then the suite's convolution / fused / network files under the fallback switches
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > gpurun_out/r03_t20_full.log 2>&1; echo "pytest rc=$?"; grep -v amdgpu.ids gpurun_out/r03_t20_full.log | grep -E "passed|failed|FAILED|Error" | cut -c1-260 | tail -15
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2 | tee gpurun_out/r03_smoke20.log
{
for sw in SGV_S2_WS=0 SGV_CONV_WS=0 SGV_WRW_WS=0 SGV_WRW_S2_WS=0 SGV_FUSED_CONV=0 SGV_UFD_TILE=0 SGV_CONV_LOWP=0 SGV_CONV_TERMS=1; do
  echo "== $sw"
  env $sw timeout 600 python -m pytest tests/test_conv3x3_gpu.py tests/test_conv_wrw_gpu.py tests/test_fused_conv_gpu.py tests/test_conv_lowp_gpu.py tests/test_ops_gpu.py tests/test_networks.py -m gpu -q --timeout 500 -rs 2>&1 | grep -v amdgpu.ids | grep -E "passed|failed|FAILED|SKIPPED" | cut -c1-200 | sort | uniq -c | sort -rn | head -12
done
} > gpurun_out/r03_fallback_switches_tests.log 2>&1
tail -60 gpurun_out/r03_fallback_switches_tests.log
