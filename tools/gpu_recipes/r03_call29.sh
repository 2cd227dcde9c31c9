# round 3, GPU call 29: kernel stats of the mixed-precision (bf16) step
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --cpu-seconds 0 --strict-steps 0 --bf16-steps 0 --pl-steps 0 --ada-steps 0 --lowp-steps 0 --clean-steps 0 --lowp bf16"
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof29 -- $B --steps 16 --warmup 2 --no-prof > /tmp/prof29.log 2>&1
cd $GRAFT_REPO_ROOT
find /tmp/prof29 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r03_bench_step_lowp_bf16_kernel_stats.csv
grep '"metric"' /tmp/prof29.log | cut -c1-200
