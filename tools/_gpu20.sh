cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -6 | tee gpurun_out/t20.log
timeout 200 python bench.py --steps 16 --warmup 2 --cpu-seconds 0 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('fp32 32/gpu:', round(d['value'],1), 'img/s', round(d['ms_per_step'],1), 'ms/step', {k: (round(v['ms_total']/16,2), round(v.get('GBps',0))) for k,v in d['kernels'].items()})"
timeout 200 python bench.py --steps 16 --warmup 2 --cpu-seconds 0 --batch-gpu 8 --lowp bf16 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bf16 8/gpu:', round(d['value'],1), 'img/s', round(d['ms_per_step'],1), 'ms/step')"
