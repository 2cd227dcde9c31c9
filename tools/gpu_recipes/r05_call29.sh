# round 5, GPU call 29: kernel census of one captured main iteration of the current step (which small launches are left)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp SGV_SELFTEST=0
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/census -- python $GRAFT_REPO_ROOT/tools/captured_census.py > /tmp/census.log 2>&1 ); tail -2 /tmp/census.log
f=$(find /tmp/census -name "*kernel_trace.csv" | head -1); python tools/captured_census_report.py $f > gpurun_out/r05_c29_captured_census.txt 2>&1; cut -c1-200 gpurun_out/r05_c29_captured_census.txt | head -80
