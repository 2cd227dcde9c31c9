# round 3, GPU call 13: kernel-time table of the --lowp bf16 / fp16 step (what is left on vendor kernels and torch element-wise ops)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
for lp in bf16 fp16; do
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof13_$lp -- python $GRAFT_REPO_ROOT/bench.py --cpu-seconds 0 --lowp $lp --steps 8 --warmup 2 --strict-steps 0 --bf16-steps 0 --pl-steps 0 --ada-steps 0 --clean-steps 0 --no-prof > /tmp/prof13_$lp.log 2>&1
cd $GRAFT_REPO_ROOT
find /tmp/prof13_$lp -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r03_lowp_${lp}_kernel_stats.csv
grep '"metric"' /tmp/prof13_$lp.log | cut -c1-160
head -45 gpurun_out/r03_lowp_${lp}_kernel_stats.csv | cut -c1-220
done
