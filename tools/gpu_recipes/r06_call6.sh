# round 6, call 6: the 2x down- / up-sampling LDS-tile kernels: parity (ops suite incl. the width / filter / flip sweep, fused suites), ops_bench with the tile forms on and off
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_c6
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_extras_gpu.py tests/test_fused_bench_shapes_gpu.py tests/test_fused_conv_gpu.py -x -q -m gpu 2>&1 | tail -15 > $OUT/pytest.log; cat $OUT/pytest.log
timeout 300 python tools/ops_bench.py --frames 32 --only upfirdn2d --json $OUT/ops_bench_upfirdn2d_n32.json > $OUT/ops_bench_upfirdn2d_n32.log 2> $OUT/ops_bench.err
SGV_UFD_TILE2X=0 timeout 300 python tools/ops_bench.py --frames 32 --only upfirdn2d > $OUT/ops_bench_upfirdn2d_n32_lanes.log 2>> $OUT/ops_bench.err
paste -d'\n' $OUT/ops_bench_upfirdn2d_n32.log $OUT/ops_bench_upfirdn2d_n32_lanes.log | grep "down2\|up2" | cut -c1-150
tail -3 $OUT/ops_bench.err
