# round 2: the switchable fallback paths still pass the conv / fused / network tests
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T="tests/test_conv3x3_gpu.py tests/test_fused_conv_gpu.py tests/test_conv_bench_shapes_gpu.py tests/test_networks.py tests/test_conv_wrw_gpu.py"
for e in "SGV_S2_WS=0" "SGV_CONV_WS=0" "SGV_WRW_WS=0" "SGV_FUSED_CONV=0" "SGV_RES_IN_SKIP=0" "SGV_ALIAS_ACC=0" "SGV_CONVT_EDGE_MFMA=0" "SGV_GEMM_FULLNK=0"; do
  echo "== $e"; env $e timeout 600 python -m pytest $T -m gpu -q --timeout 300 2>&1 | grep -v amdgpu.ids | tail -2
done 2>&1 | tee gpurun_out/r02_fallbacks.log
timeout 300 python bench.py --steps 10 --warmup 3 --cpu-seconds 0 --strict-steps 0 --ada-steps 0 --bf16-steps 0 2>/dev/null | tail -1 | cut -c1-160
