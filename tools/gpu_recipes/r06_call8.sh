# round 6, call 8: the FIR family at the step's own sizes (96 frames), settled protocol: tile forms vs the lane-exchange forms for the 2x geometries and mode 2
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_c8
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python tools/fir_bench.py --frames 96 > $OUT/fir_bench_n96_default.log 2> $OUT/err.log
SGV_UFD_TILE2X=0 SGV_UFD_TILE_EPI2=0 timeout 600 python tools/fir_bench.py --frames 96 > $OUT/fir_bench_n96_lanes.log 2>> $OUT/err.log
SGV_TILE_XCD=0 timeout 600 python tools/fir_bench.py --frames 96 > $OUT/fir_bench_n96_xcd0.log 2>> $OUT/err.log
timeout 600 python tools/fir_bench.py --frames 32 > $OUT/fir_bench_n32_default.log 2>> $OUT/err.log
SGV_UFD_TILE2X=0 SGV_UFD_TILE_EPI2=0 timeout 600 python tools/fir_bench.py --frames 32 > $OUT/fir_bench_n32_lanes.log 2>> $OUT/err.log
for f in n96_default n96_lanes n96_xcd0 n32_default n32_lanes; do echo "== $f"; cat $OUT/fir_bench_$f.log | cut -c1-140; done
tail -5 $OUT/err.log
