# round 3, GPU call 39: the mixed-precision step as its own run (bf16 and fp16 tensors in the four highest resolutions), final tree
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "from stylegan_v_amd.torch_utils import custom_ops as c; import sys; sys.exit(0 if c.is_built() else 1)" || { echo "in-tree library is stale: stop"; exit 1; }
for lp in bf16 fp16; do
  timeout 100 python bench.py --lowp $lp --cpu-seconds 0 --steps 16 --warmup 3 2>/dev/null | tail -1 > gpurun_out/r03_bench_lowp_$lp.json
  python -c "
import json; d=json.load(open('gpurun_out/r03_bench_lowp_$lp.json')); print('$lp', round(d['value'],1), round(d['ms_per_step'],2), 'no_prof', round(d['value_no_prof'],1), d['dtype'][:80])"
done
