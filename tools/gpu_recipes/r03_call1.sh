# round 3, GPU call 1: whole GPU suite on the new tree (fused bench-shape tests, FFS-256 module golden, graphs + DDP + ADA), the new fir_asm
# kernel with / without the LDS hand-off in the lab harness, bench line, torch-op table of the step
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/r03_t1.log 2>&1; echo "pytest rc=$?"; grep -v amdgpu.ids gpurun_out/r03_t1.log | tail -40
{
for cfg in "32 257 1" "96 257 1" "96 256 2"; do
  for lds in 0 1; do
    echo "== N IH pad: $cfg  SGV_FIR_LDS=$lds"
    SGV_FIR_LDS=$lds timeout 120 tools/ufd_lab $cfg 2>&1 | grep -E "libsgv|copy2 ntl1|V4 asm loads, counted waits PF4 strip 16|mismatch"
  done
  echo "== N IH pad: $cfg  SGV_FIR_ASM=0 (lanes kernel)"
  SGV_FIR_ASM=0 timeout 120 tools/ufd_lab $cfg 2>&1 | grep -E "libsgv|mismatch"
done
} > gpurun_out/r03_ufd_lab1.log 2>&1
cat gpurun_out/r03_ufd_lab1.log
timeout 400 python bench.py --cpu-seconds 0 2> gpurun_out/r03_b1.err | tail -1 > gpurun_out/r03_b1.json; echo "bench rc=$?"; cut -c1-300 gpurun_out/r03_b1.json
python - <<'PY'
import json
d = json.load(open('gpurun_out/r03_b1.json'))
print('value', d['value'], 'strict', d.get('value_strict_fp32'), 'ada', d.get('value_aug_ada'), 'bf16p', d.get('value_bf16_products'))
print('roofline', d['roofline']['frac'], 'ufd', d['roofline_upfirdn2d']['frac'], d['roofline_upfirdn2d']['achieved'])
for k, v in d['kernels'].items():
    print('%-18s %5d launches %8.2f ms  %s' % (k, v['launches'], v['ms_total'], ' '.join('%s=%.1f' % (a, v[a]) for a in ('GBps', 'TFLOPs') if a in v)))
PY
SGV_TORCH_PROFILE=gpurun_out/r03_torch_ops.txt timeout 300 python bench.py --cpu-seconds 0 --steps 4 --warmup 2 --strict-steps 0 --ada-steps 0 --bf16-steps 0 > /dev/null 2> gpurun_out/r03_b1p.err; echo "torch profile rc=$?"
timeout 300 python bench.py --cpu-seconds 0 --batch-gpu 8 --steps 20 --warmup 3 --strict-steps 0 --ada-steps 0 --bf16-steps 0 --no-prof 2> gpurun_out/r03_b1_b8.err | tail -1 > gpurun_out/r03_b1_b8.json; echo "batch8 eager rc=$?"; cut -c1-200 gpurun_out/r03_b1_b8.json
timeout 300 python bench.py --cpu-seconds 0 --batch-gpu 8 --steps 20 --warmup 3 --strict-steps 0 --ada-steps 0 --bf16-steps 0 --graphs 2> gpurun_out/r03_b1_b8g.err | tail -1 > gpurun_out/r03_b1_b8g.json; echo "batch8 graphs rc=$?"; cut -c1-200 gpurun_out/r03_b1_b8g.json
timeout 300 python bench.py --cpu-seconds 0 --batch-gpu 8 --steps 20 --warmup 3 --strict-steps 0 --ada-steps 0 --bf16-steps 0 --graphs --aug ada 2> gpurun_out/r03_b1_b8ga.err | tail -1 > gpurun_out/r03_b1_b8ga.json; echo "batch8 graphs+ada rc=$?"; cut -c1-200 gpurun_out/r03_b1_b8ga.json
timeout 300 python bench.py --cpu-seconds 0 --batch-gpu 8 --steps 20 --warmup 3 --strict-steps 0 --ada-steps 0 --bf16-steps 0 --no-prof --aug ada 2> gpurun_out/r03_b1_b8a.err | tail -1 > gpurun_out/r03_b1_b8a.json; echo "batch8 eager+ada rc=$?"; cut -c1-200 gpurun_out/r03_b1_b8a.json
tail -5 gpurun_out/r03_b1_b8g.err gpurun_out/r03_b1_b8ga.err
