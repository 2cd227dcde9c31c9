# compiler-inserted vmcnt(0) removed from the producers (wrw x-scale load, conv3x3 epilogue vectors as counted asm loads): lab, tests, bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for a in 0 6; do echo "== wrw s1 lab, WRW_ABL=$a"; WRW_ABL=$a timeout 200 tools/wrw_lab 10 2>&1 | grep -E "ws views=1 terms=3 rows=(64|32) grid" | grep -v "256\^2   ws views=1 terms=3 rows=32\|128\^2  ws views=1 terms=3 rows=32\|64\^2   ws views=1 terms=3 rows=32"; done
echo "== conv lab"; timeout 300 tools/conv_lab 10 2>&1 | grep -E "ws |EPI|PRO" | tail -24
timeout 900 python -m pytest tests/test_conv_wrw_gpu.py tests/test_conv_bench_shapes_gpu.py tests/test_fused_conv_gpu.py tests/test_conv3x3_gpu.py tests/test_networks.py -x -q -m gpu 2>&1 | tail -3
for i in 1 2; do
  timeout 300 python bench.py --steps 12 --warmup 4 --cpu-seconds 0 --strict-steps 0 --ada-steps 0 --bf16-steps 0 --no-prof 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'])"
done
