# round 2, GPU call 8: transposed-convolution edges through batched sgv_fc; producer / consumer strided kernel (lab + tests + bench)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 200 tools/conv_s2_lab 5 0 > gpurun_out/r02_conv_s2_lab_base.log 2>&1; echo "lab base rc=$?"; grep -v "^$" gpurun_out/r02_conv_s2_lab_base.log | cut -c1-170
timeout 200 tools/conv_s2_lab 5 1 > gpurun_out/r02_conv_s2_lab_ws.log 2>&1; echo "lab ws rc=$?"; grep -v "^$" gpurun_out/r02_conv_s2_lab_ws.log | cut -c1-170
timeout 900 python -m pytest tests/test_fc_gpu.py tests/test_conv3x3_gpu.py tests/test_conv_bench_shapes_gpu.py tests/test_networks.py -m gpu -q --timeout 300 -x > gpurun_out/r02_t8.log 2>&1; echo "pytest rc=$?"; grep -v amdgpu.ids gpurun_out/r02_t8.log | tail -12
timeout 300 python bench.py --steps 10 --warmup 3 --cpu-seconds 0 --strict-steps 0 --ada-steps 0 2> gpurun_out/r02_b8.err | tail -1 > gpurun_out/r02_b8.json; echo "bench rc=$?"; cut -c1-200 gpurun_out/r02_b8.json
