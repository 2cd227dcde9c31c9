# round 4, GPU call 28: the step at 16 and 64 videos per GPU (the sweep around the headline's 32 and config 4's 8), final sources
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
OFF="--cpu-seconds 0 --strict-steps 0 --bf16-steps 0 --pl-steps 0 --lowp-steps 0 --split3-steps 0 --ada-steps 0"
for b in 16 64; do
  timeout 400 python bench.py $OFF --batch-gpu $b --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r04_c28_bench_b$b.json
  python - $b <<'PY'
import json, sys
d = json.load(open('gpurun_out/r04_c28_bench_b%s.json' % sys.argv[1]))
print('videos/GPU', sys.argv[1], 'value', round(d['value'], 1), 'ms', round(d['ms_per_step'], 2), 'no_prof', round(d['value_no_prof'], 1), 'roofline frac', round(d['roofline']['frac'], 3), 'ufd', round(d['config']['upfirdn2d_in_step_frac'], 3))
PY
done
python - <<'PY'
import torch
print('peak memory not tracked here; device memory in use after the runs:', torch.cuda.mem_get_info())
PY
