# round 3, GPU call 42: the slab form of act_grad_scale / scale_dot (> 65,535 planes) + the modulation / fused-layer tests around it; Dmain as one pass at 32 videos per GPU
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "from stylegan_v_amd.torch_utils import custom_ops as c; import sys; sys.exit(0 if c.is_built() else 1)" || { echo "in-tree library is stale: stop"; exit 1; }
timeout 60 python -m pytest tests/test_pointwise_gpu.py tests/test_fused_conv_gpu.py -m gpu -q -x --timeout 50 -k "planes or fused_layer_forward" 2>&1 | grep -v amdgpu.ids | tail -2
SGV_D_CONCAT=1 timeout 75 python bench.py --cpu-seconds 0 --strict-steps 0 --bf16-steps 0 --pl-steps 0 --ada-steps 0 --lowp-steps 0 --clean-steps 0 --no-prof --steps 10 --warmup 3 2> gpurun_out/r03_d_concat_32.err | tail -1 > gpurun_out/r03_d_concat_32.json
tail -2 gpurun_out/r03_d_concat_32.err | cut -c1-200
python -c "
import json
d=json.load(open('gpurun_out/r03_d_concat_32.json')); print('d_concat=1  32 videos/GPU', round(d['value'],1), round(d['ms_per_step'],2))" | tee -a gpurun_out/r03_d_concat_ab.log
