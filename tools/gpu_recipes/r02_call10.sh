# round 2, GPU call 10: tap-pair strided kernel + batched-fc transposed edges in the library: tests + bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_fc_gpu.py tests/test_conv3x3_gpu.py tests/test_conv_bench_shapes_gpu.py tests/test_networks.py tests/test_abi.py -m gpu -q --timeout 300 -x > gpurun_out/r02_t10.log 2>&1; echo "pytest rc=$?"; grep -v amdgpu.ids gpurun_out/r02_t10.log | tail -12
timeout 300 python bench.py --steps 10 --warmup 3 --cpu-seconds 0 --strict-steps 0 --ada-steps 0 2> gpurun_out/r02_b10.err | tail -1 > gpurun_out/r02_b10.json; echo "bench rc=$?"; cut -c1-200 gpurun_out/r02_b10.json
SGV_S2_WS=0 SGV_CONVT_EDGE_FC=0 timeout 300 python bench.py --steps 10 --warmup 3 --cpu-seconds 0 --strict-steps 0 --ada-steps 0 2> gpurun_out/r02_b10b.err | tail -1 > gpurun_out/r02_b10b.json; echo "bench (old s2 + old edges) rc=$?"; cut -c1-200 gpurun_out/r02_b10b.json
