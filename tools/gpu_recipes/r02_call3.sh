# round 2, GPU call 3: producer pipeline rework (DMA wave + asm loads with counted waits): lab, fused tests, bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 240 tools/conv_lab 5 ws > gpurun_out/r02_conv_lab_ws3.log 2>&1; echo "lab rc=$?"; grep "check ws\|ablation\|4-wave" gpurun_out/r02_conv_lab_ws3.log
timeout 600 python -m pytest tests/test_fused_conv_gpu.py tests/test_conv_bench_shapes_gpu.py tests/test_conv3x3_gpu.py tests/test_networks.py -m gpu -q --timeout 300 > gpurun_out/r02_t3.log 2>&1; echo "pytest rc=$?"; grep -v amdgpu.ids gpurun_out/r02_t3.log | tail -8
timeout 300 python bench.py --steps 10 --warmup 3 --cpu-seconds 0 --strict-steps 0 2> gpurun_out/r02_b3.err | tail -1 > gpurun_out/r02_b3.json; echo "bench rc=$?"; cut -c1-200 gpurun_out/r02_b3.json
