# round 6, call 2: the product tile kernel with the XCD-aware tile order, non-temporal interior loads, aligned 257-wide stores and the 64-register bound,
# each switched off in turn (lab builds -DSGV_TILE_LAB_OFF=2/4/8), against V6; then the upfirdn2d GPU tests and ops_bench through the C ABI
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_c2
mkdir -p $OUT
export TMPDIR=/tmp
for a in "32 257 1" "32 256 2" "96 257 1" "96 256 2"; do
  for v in 0 2 4 8; do
    echo "== $a  LAB_OFF=$v" >> $OUT/ufd_lab6.log
    if [ $v = 0 ]; then timeout 300 tools/ufd_lab6_off$v $a 2>&1 | grep -v "^copy" >> $OUT/ufd_lab6.log; else UFD_PRODUCT_ONLY=1 timeout 300 tools/ufd_lab6_off$v $a 2>&1 | grep V7 >> $OUT/ufd_lab6.log; fi
  done
done
cat $OUT/ufd_lab6.log | cut -c1-200
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu 2>&1 | tail -5 > $OUT/pytest_ops_gpu.log; cat $OUT/pytest_ops_gpu.log
timeout 300 python tools/ops_bench.py --frames 32 --only upfirdn2d --json $OUT/ops_bench_upfirdn2d_n32.json > $OUT/ops_bench_upfirdn2d_n32.log 2> $OUT/ops_bench.err
SGV_TILE_XCD=0 timeout 300 python tools/ops_bench.py --frames 32 --only upfirdn2d > $OUT/ops_bench_upfirdn2d_n32_xcd0.log 2>> $OUT/ops_bench.err
head -12 $OUT/ops_bench_upfirdn2d_n32.log; echo; head -12 $OUT/ops_bench_upfirdn2d_n32_xcd0.log
