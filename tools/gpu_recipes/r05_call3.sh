# round 5, GPU call 3: whole -m gpu suite on the round's changes so far (fir_asm FIR kernel removed, W-stationary 1x1 member, fp16 tensors as fp16 MFMA operands + the
# reference-generated fp16 golden), then the mixed-precision step with fp16 and bf16 tensors
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
SGV_ERROR_TABLE_DIR=gpurun_out timeout 1500 python -m pytest tests/ -q -m gpu --maxfail=12 -rf -s 2>&1 | grep -v "^\[" | tail -60 > gpurun_out/r05_c3_pytest_tail.log; tail -40 gpurun_out/r05_c3_pytest_tail.log
SGV_ERROR_TABLE_DIR=gpurun_out timeout 600 python -m pytest tests/test_networks.py tests/test_conv_lowp_gpu.py -q -m gpu -s -k "fp16 or 16bit" 2>&1 | grep "^\[" | cut -c1-220 > gpurun_out/r05_c3_fp16_prints.log; tail -45 gpurun_out/r05_c3_fp16_prints.log
OFF="--cpu-seconds 0 --strict-steps 0 --bf16-steps 0 --pl-steps 0 --ada-steps 0 --lowp-steps 0 --split3-steps 0"
for lp in fp16 bf16; do
  timeout 300 python bench.py $OFF --lowp $lp --steps 10 --warmup 4 > gpurun_out/r05_c3_bench_lowp_$lp.json 2> gpurun_out/r05_c3_bench_lowp_$lp.err; echo "lowp $lp rc=$?"
  cp bench_detail.json gpurun_out/r05_c3_bench_detail_lowp_$lp.json
  python - gpurun_out/r05_c3_bench_lowp_$lp.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print(d['dtype'], 'value', d['value'], 'ms', d['ms_per_step'], 'no_prof', d.get('value_no_prof'))
PY
done
