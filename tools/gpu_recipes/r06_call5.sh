# round 6, call 5: tile kernel with the XCD-aware order + non-temporal interior loads in the product: ops_bench (settled / cold), the driver's bench command
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_c5
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python tools/ops_bench.py --frames 32 --only upfirdn2d --json $OUT/ops_bench_upfirdn2d_n32.json > $OUT/ops_bench_upfirdn2d_n32.log 2> $OUT/ops_bench.err
cat $OUT/ops_bench_upfirdn2d_n32.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err; echo "bench rc=$?"
cp bench_detail.json $OUT/bench_driver_cmd_detail.json
tail -1 $OUT/bench_driver_cmd.json | cut -c1-3000
grep -i "ufd\|upfirdn" $OUT/bench_driver_cmd.err | head -40 | cut -c1-250
