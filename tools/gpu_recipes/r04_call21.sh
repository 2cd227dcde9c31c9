# round 4, GPU call 21: what fails under SGV_CONVT_EDGE_MFMA=0 and under the one-role-per-wave kernels (terms 3)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T="tests/test_fused_conv_gpu.py tests/test_networks.py tests/test_conv3x3_gpu.py tests/test_conv_wrw_gpu.py"
{
echo "== SGV_CONVT_EDGE_MFMA=0"
SGV_CONVT_EDGE_MFMA=0 timeout 600 python -m pytest $T -m gpu -q --timeout 300 -p no:cacheprovider 2>&1 | grep -E "^FAILED|^ERROR|Error|assert " | cut -c1-300 | head -40
echo "== one-role kernels, terms 3"
SGV_CONV_TERMS=3 SGV_WRW_TERMS=3 SGV_CONV_WS=0 SGV_S2_WS=0 SGV_WRW_WS=0 SGV_WRW_S2_WS=0 timeout 600 python -m pytest $T -m gpu -q --timeout 300 -p no:cacheprovider 2>&1 | grep -E "^FAILED|^ERROR|Error|assert " | cut -c1-300 | head -40
} 2>&1 | tee gpurun_out/r04_c21_fallback_failures.log
