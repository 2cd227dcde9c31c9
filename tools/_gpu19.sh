cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
for lp in none bf16; do
timeout 200 python bench.py --steps 16 --warmup 2 --cpu-seconds 0 --batch-gpu 8 --lowp $lp 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$lp batch-gpu 8:', round(d['value'],1), 'img/s', round(d['ms_per_step'],1), 'ms/step', {k: round(v['ms_total']/16,2) for k,v in d['kernels'].items()})"
done | tee gpurun_out/bench19.log
