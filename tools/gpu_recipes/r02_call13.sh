# round 2, GPU call 13: fused down-sampling layer tail: tests + networks + bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_fused_conv_gpu.py tests/test_networks.py tests/test_abi.py -m gpu -q --timeout 300 -x > gpurun_out/r02_t13.log 2>&1; echo "pytest rc=$?"; grep -v amdgpu.ids gpurun_out/r02_t13.log | tail -30
timeout 300 python bench.py --steps 10 --warmup 3 --cpu-seconds 0 --strict-steps 0 --ada-steps 0 --bf16-steps 0 2> gpurun_out/r02_b13.err | tail -1 > gpurun_out/r02_b13.json; echo "bench rc=$?"; cut -c1-200 gpurun_out/r02_b13.json
SGV_FUSED_CONV=0 timeout 300 python bench.py --steps 10 --warmup 3 --cpu-seconds 0 --strict-steps 0 --ada-steps 0 --bf16-steps 0 2> gpurun_out/r02_b13b.err | tail -1 > gpurun_out/r02_b13b.json; echo "bench (no fusion) rc=$?"; cut -c1-200 gpurun_out/r02_b13b.json
