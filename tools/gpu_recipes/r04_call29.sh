# round 4, GPU call 29: fused stride-1 layer beyond 65,535 (sample, channel) planes (a stale host-side guard): test, then 64 videos per GPU again
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_fused_conv_gpu.py -x -q -m gpu -k "beyond or no_grad_pass" 2>&1 | tail -3
OFF="--cpu-seconds 0 --strict-steps 0 --bf16-steps 0 --pl-steps 0 --lowp-steps 0 --split3-steps 0 --ada-steps 0"
timeout 400 python bench.py $OFF --batch-gpu 64 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r04_c29_bench_b64.json
python - <<'PY'
import json
d = json.load(open('gpurun_out/r04_c29_bench_b64.json'))
k = d['kernels_by_variant']
print('videos/GPU 64 value', round(d['value'], 1), 'ms', round(d['ms_per_step'], 2), {n: round(k[n]['ms_per_step'], 1) for n in ('conv_s1_ws', 'conv_s1_ws_fused', 'conv_s1_ws_accumulate', 'bias_act') if n in k})
PY
