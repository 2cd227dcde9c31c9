# round 6, call 1: diagnostics for the FIR tile kernel -- copy ceilings (aligned / misaligned 16-byte requests), V6 / product / span-load variants, XCD-aware tile order;
# one launch per event bracket vs four; the same binary under the kernel tracer; the product through tools/ops_bench.py
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_c1
mkdir -p $OUT
export TMPDIR=/tmp
for a in "32 257 1" "32 256 2" "96 257 1"; do
  echo "== $a" >> $OUT/ufd_lab6.log
  timeout 300 tools/ufd_lab6 $a >> $OUT/ufd_lab6.log 2>&1
done
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/trace -- $GRAFT_REPO_ROOT/tools/ufd_lab6 32 257 1 > /dev/null 2>&1 )
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/ufd_lab6_n32_257_kernel_stats.csv
rm -rf $OUT/trace
timeout 300 python tools/ops_bench.py --frames 32 --only upfirdn2d --json $OUT/ops_bench_upfirdn2d_n32.json > $OUT/ops_bench_upfirdn2d_n32.log 2> $OUT/ops_bench.err
cat $OUT/ufd_lab6.log | cut -c1-200
head -8 $OUT/ops_bench_upfirdn2d_n32.log
