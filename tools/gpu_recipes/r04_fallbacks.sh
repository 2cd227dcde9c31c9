# round 4: the switchable paths still pass the convolution / fused-layer / network / ADA tests (each line: one environment, the last line of pytest)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T="tests/test_fused_conv_gpu.py tests/test_networks.py tests/test_conv3x3_gpu.py tests/test_conv_wrw_gpu.py"
{
for e in "SGV_CONV_TERMS=3 SGV_WRW_TERMS=3" "SGV_CONV_TERMS=3 SGV_WRW_TERMS=3 SGV_CONV_WS=0 SGV_S2_WS=0 SGV_WRW_WS=0 SGV_WRW_S2_WS=0" "SGV_CONV_TERMS=1 SGV_WRW_TERMS=1" "SGV_FUSED_CONV=0" "SGV_RES_IN_SKIP=0" "SGV_ALIAS_ACC=0" \
         "SGV_CONVT_EDGE_MFMA=0" "SGV_GEMM_STREAM=0" "SGV_GEMM_TERMS=0" "SGV_UFD_TILE=0" "SGV_EQLR_BATCH=0"; do
  echo "== $e: $(env $e timeout 600 python -m pytest $T -m gpu -q --timeout 300 -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -1)"
done
echo "== SGV_D_CONCAT=0: $(SGV_D_CONCAT=0 timeout 600 python -m pytest tests/test_dmain_concat.py tests/test_extras_gpu.py -m gpu -q --timeout 300 2>&1 | tail -1)"
echo "== SGV_RESAMPLE_ADJOINT=scatter: $(SGV_RESAMPLE_ADJOINT=scatter timeout 600 python -m pytest tests/test_augment.py -m gpu -q --timeout 300 2>&1 | tail -1)"
} 2>&1 | tee gpurun_out/r04_fallback_switches_tests.log
