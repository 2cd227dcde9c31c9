# round 6, call 17: FIR bounds inherited instead of armed; census of one captured main iteration, the Python sources of the small aten launches, the separate bound passes
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_c17
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_networks.py tests/test_extras_gpu.py tests/test_conv_f16split_gpu.py tests/test_conv3x3_gpu.py tests/test_fused_conv_gpu.py -x -q -m gpu 2>&1 | tail -6 > $OUT/pytest.log; tail -3 $OUT/pytest.log
( cd /tmp && SGV_SELFTEST=0 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/census -- python $GRAFT_REPO_ROOT/tools/captured_census.py > /tmp/census.log 2>&1 ); tail -2 /tmp/census.log
f=$(find /tmp/census -name "*kernel_trace.csv" | head -1); python tools/captured_census_report.py $f > $OUT/captured_census.txt 2>&1; head -45 $OUT/captured_census.txt | cut -c1-190
SGV_SELFTEST=0 timeout 300 python tools/small_launch_sources.py 2>/dev/null > $OUT/small_launch_sources.txt; head -40 $OUT/small_launch_sources.txt | cut -c1-230
SGV_AMAX_TRACE=1 SGV_SELFTEST=0 timeout 300 python bench.py --eager --steps 3 --warmup 1 --cpu-seconds 0 --no-prof --strict-steps 0 --bf16-steps 0 --pl-steps 0 --ada-steps 0 --lowp-steps 0 --split3-steps 0 --graph-steps 0 2>&1 | grep -A45 "amax trace" > $OUT/amax_trace.txt; head -45 $OUT/amax_trace.txt | cut -c1-200
