# rocprof kernel stats of a few bench steps (no PMC): quick per-kernel check after a kernel change
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_q -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --cpu-seconds 0 --strict-steps 0 --ada-steps 0 --bf16-steps 0 --no-prof > /tmp/prof_q.log 2>&1
cd $GRAFT_REPO_ROOT
find /tmp/prof_q -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r02_quick_kernel_stats.csv
grep '"metric"' /tmp/prof_q.log | cut -c1-200
grep -E "fc_kernel|wrw|gemm_f32" gpurun_out/r02_quick_kernel_stats.csv | cut -c1-200
