# round 3, GPU call 44 (the round's last GPU seconds): BASELINE configs[3]'s per-GPU workload -- 8 videos per GPU, bf16 tensors, hipGraph replay -- on one GPU
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 40 python bench.py --lowp bf16 --batch-gpu 8 --graphs --cpu-seconds 0 --steps 10 --warmup 3 --clean-steps 0 2>/dev/null | tail -1 > gpurun_out/r03_bench_batch8_bf16_hipgraph.json
python -c "
import json; d=json.load(open('gpurun_out/r03_bench_batch8_bf16_hipgraph.json')); print('8 videos/GPU, bf16, graphs:', round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms')"
