# round 2, final validation: full GPU suite, smoke, default bench (all companions), hipGraph bench, 8-videos-per-GPU line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/r02_final_tests.log 2>&1; echo "pytest rc=$?"; grep -v amdgpu.ids gpurun_out/r02_final_tests.log | tail -4
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python bench.py --steps 20 --warmup 5 2> gpurun_out/r02_bench_final.err | tail -1 > gpurun_out/r02_bench_final.json; echo "bench rc=$?"; python - <<'PY'
import json
d=json.load(open('gpurun_out/r02_bench_final.json'))
print({k:d[k] for k in ('value','ms_per_step','value_strict_fp32','value_aug_ada','value_bf16_products')})
print('roofline', {k:d['roofline'][k] for k in ('kernel','achieved','peak','frac','traffic')})
print('ufd', {k:d['roofline_upfirdn2d'][k] for k in ('achieved','frac','traffic')})
print('cpu', d['cpu_baseline'])
PY
timeout 300 python bench.py --steps 10 --warmup 3 --cpu-seconds 0 --strict-steps 0 --ada-steps 0 --bf16-steps 0 --graphs 2>/dev/null | tail -1 > gpurun_out/r02_bench_final_hipgraph.json; cut -c1-160 gpurun_out/r02_bench_final_hipgraph.json
timeout 300 python bench.py --steps 10 --warmup 3 --cpu-seconds 0 --strict-steps 0 --ada-steps 0 --bf16-steps 0 --batch-gpu 8 2>/dev/null | tail -1 > gpurun_out/r02_bench_final_batch8.json; cut -c1-160 gpurun_out/r02_bench_final_batch8.json
timeout 300 python bench.py --steps 10 --warmup 3 --cpu-seconds 0 --strict-steps 0 --ada-steps 0 --bf16-steps 0 --batch-gpu 8 --graphs 2>/dev/null | tail -1 > gpurun_out/r02_bench_final_batch8_hipgraph.json; cut -c1-160 gpurun_out/r02_bench_final_batch8_hipgraph.json
