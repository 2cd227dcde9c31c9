# round 3, GPU call 37: half-full last output-channel tile (c_out % 32) in the stride-1 and transposed producer / consumer kernels: parity tests, the 1024^2 synthesis line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "from stylegan_v_amd.torch_utils import custom_ops as c; import sys; sys.exit(0 if c.is_built() else 1)" || { echo "in-tree library is stale: stop"; exit 1; }
timeout 300 python -m pytest tests/test_conv3x3_gpu.py tests/test_fused_conv_gpu.py tests/test_conv_lowp_gpu.py tests/test_conv_bench_shapes_gpu.py -m gpu -q -x --timeout 200 2>&1 | grep -v amdgpu.ids | tail -5
timeout 150 python bench.py --workload g1024 --cpu-seconds 0 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('g1024', round(d['value'],1), 'frames/s', round(d['ms_per_step'],2), 'ms', 'no_prof', round(d['value_no_prof'],1))"
SGV_CONV_M32=0 timeout 150 python bench.py --workload g1024 --cpu-seconds 0 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('g1024 SGV_CONV_M32=0', round(d['value'],1), 'frames/s', round(d['ms_per_step'],2), 'ms')"
