# round 6, call 12: tile kernel with the 4k+1-th column as its own small phase, hoisted scale / bias loads, reciprocal act-gradient math, one bound atomic per workgroup:
# parity suites, then the family at 96 frames (bound armed / not; mode 2 tile vs lane-exchange)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_c12
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_extras_gpu.py tests/test_fused_bench_shapes_gpu.py tests/test_fused_conv_gpu.py -x -q -m gpu 2>&1 | tail -15 > $OUT/pytest.log; tail -4 $OUT/pytest.log
timeout 600 python tools/fir_bench.py --frames 96 --widths 256,128 --amax 1 --rounds 3 > $OUT/fir_bench_n96_amax.log 2> $OUT/err.log
timeout 600 python tools/fir_bench.py --frames 96 --widths 256,128 --rounds 3 > $OUT/fir_bench_n96.log 2>> $OUT/err.log
SGV_UFD_TILE_EPI2=0 timeout 600 python tools/fir_bench.py --frames 96 --widths 256,128 --rounds 3 --only mode2 > $OUT/fir_bench_n96_mode2_lanes.log 2>> $OUT/err.log
for f in n96_amax n96 n96_mode2_lanes; do echo "== $f"; cat $OUT/fir_bench_$f.log | cut -c1-150; done
tail -3 $OUT/err.log
