# round 3, GPU call 40: Dmain as one discriminator pass (SGV_D_CONCAT=1, off by default): bench A/B at 32 and 8 videos per GPU
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "from stylegan_v_amd.torch_utils import custom_ops as c; import sys; sys.exit(0 if c.is_built() else 1)" || { echo "in-tree library is stale: stop"; exit 1; }
B="python bench.py --cpu-seconds 0 --strict-steps 0 --bf16-steps 0 --pl-steps 0 --ada-steps 0 --lowp-steps 0 --clean-steps 0 --no-prof --steps 10 --warmup 3"
for sw in 1 0 1 0; do
  SGV_D_CONCAT=$sw timeout 60 $B 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('d_concat=$sw  32 videos/GPU', round(d['value'],1), round(d['ms_per_step'],2))"
done | tee gpurun_out/r03_d_concat_ab.log
for sw in 1 0; do
  SGV_D_CONCAT=$sw timeout 60 $B --batch-gpu 8 --graphs 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('d_concat=$sw   8 videos/GPU, graphs', round(d['value'],1), round(d['ms_per_step'],2))"
done | tee -a gpurun_out/r03_d_concat_ab.log
