# round 3, GPU call 19: kernel-time table of the default (fp32) step: what is left outside the hand-written kernels
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof19 -- python $GRAFT_REPO_ROOT/bench.py --cpu-seconds 0 --steps 8 --warmup 2 --strict-steps 0 --bf16-steps 0 --pl-steps 0 --ada-steps 0 --lowp-steps 0 --clean-steps 0 --no-prof > /tmp/prof19.log 2>&1
cd $GRAFT_REPO_ROOT
find /tmp/prof19 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r03_fp32_kernel_stats_c19.csv
grep '"metric"' /tmp/prof19.log | cut -c1-160
SGV_TORCH_PROFILE=gpurun_out/r03_torch_profile_c19.txt timeout 300 python bench.py --cpu-seconds 0 --steps 4 --warmup 2 --strict-steps 0 --bf16-steps 0 --pl-steps 0 --ada-steps 0 --lowp-steps 0 --clean-steps 0 --no-prof 2>/dev/null | cut -c1-100
