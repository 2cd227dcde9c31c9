# round 3, GPU call 45: the same line with the Dmain phase as one pass (SGV_D_CONCAT=1, off by default)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
SGV_D_CONCAT=1 timeout 35 python bench.py --lowp bf16 --batch-gpu 8 --graphs --cpu-seconds 0 --steps 10 --warmup 3 --clean-steps 0 2>/dev/null | tail -1 > gpurun_out/r03_bench_batch8_bf16_hipgraph_dconcat.json
python -c "
import json; d=json.load(open('gpurun_out/r03_bench_batch8_bf16_hipgraph_dconcat.json')); print('8 videos/GPU, bf16, graphs, D_CONCAT:', round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms')"
