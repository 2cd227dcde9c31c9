# round 6, call 11: does the bound side output (one atomic max per wave) explain the gap between the tile kernels in the step and alone?
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_c11
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python tools/fir_bench.py --frames 96 --widths 256,128 --amax 1 --rounds 3 > $OUT/fir_bench_n96_amax_default.log 2> $OUT/err.log
SGV_UFD_TILE2X=0 SGV_UFD_TILE_EPI2=0 timeout 600 python tools/fir_bench.py --frames 96 --widths 256,128 --amax 1 --rounds 3 > $OUT/fir_bench_n96_amax_lanes.log 2>> $OUT/err.log
timeout 600 python tools/fir_bench.py --frames 96 --widths 256,128 --rounds 3 > $OUT/fir_bench_n96_noamax_default.log 2>> $OUT/err.log
for f in n96_amax_default n96_amax_lanes n96_noamax_default; do echo "== $f"; cat $OUT/fir_bench_$f.log | cut -c1-150; done
tail -3 $OUT/err.log
