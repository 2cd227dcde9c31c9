# round 6, call 9: the FIR family behind matrix-bound kernels (the step's regime) vs back to back, tile forms vs lane-exchange forms
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_c9
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python tools/fir_bench.py --frames 96 --widths 256,128 --interleave 2 --rounds 3 > $OUT/fir_bench_n96_inter_default.log 2> $OUT/err.log
SGV_UFD_TILE2X=0 SGV_UFD_TILE_EPI2=0 timeout 600 python tools/fir_bench.py --frames 96 --widths 256,128 --interleave 2 --rounds 3 > $OUT/fir_bench_n96_inter_lanes.log 2>> $OUT/err.log
timeout 600 python tools/fir_bench.py --frames 32 > $OUT/fir_bench_n32_default.log 2>> $OUT/err.log
SGV_UFD_TILE2X=0 SGV_UFD_TILE_EPI2=0 timeout 600 python tools/fir_bench.py --frames 32 > $OUT/fir_bench_n32_lanes.log 2>> $OUT/err.log
for f in n96_inter_default n96_inter_lanes n32_default n32_lanes; do echo "== $f"; cat $OUT/fir_bench_$f.log | cut -c1-150; done
tail -5 $OUT/err.log
