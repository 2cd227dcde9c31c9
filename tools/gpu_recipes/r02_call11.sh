# round 2, GPU call 11: producer / consumer transposed kernel + MFMA edge kernel in the lab (kind 1)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
echo "=== checks (with edges)"; timeout 120 tools/conv_s2_lab 1 1 0 1 2>&1 | grep "check transposed"
echo "=== ws with edges"; timeout 120 tools/conv_s2_lab 5 1 0 1 2>&1 | grep "transposed .*terms=3" | grep -v check | cut -c1-150
echo "=== ws NO_EDGE"; NO_EDGE=1 timeout 120 tools/conv_s2_lab 5 1 0 1 2>&1 | grep "transposed .*terms=3" | grep -v check | cut -c1-150
} > gpurun_out/r02_convT_lab_edges.log 2>&1
cat gpurun_out/r02_convT_lab_edges.log
timeout 600 python -m pytest tests/test_conv3x3_gpu.py tests/test_conv_bench_shapes_gpu.py -m gpu -q --timeout 300 -x 2>&1 | grep -v amdgpu.ids | tail -3
timeout 300 python bench.py --steps 10 --warmup 3 --cpu-seconds 0 --strict-steps 0 --ada-steps 0 --bf16-steps 4 2> gpurun_out/r02_b11.err | tail -1 > gpurun_out/r02_b11.json; echo "bench rc=$?"; cut -c1-200 gpurun_out/r02_b11.json; python -c "
import json; d=json.load(open('gpurun_out/r02_b11.json')); print('bf16 products:', d.get('value_bf16_products'))"
