# round 5, GPU call 10: why the captured-step companion of the default line (608.6 img/s) is slower than the standalone --graphs run (641.2): same process state or not
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
OFF="--cpu-seconds 0 --strict-steps 0 --bf16-steps 0 --pl-steps 0 --ada-steps 0 --lowp-steps 0 --split3-steps 0"
timeout 300 python bench.py $OFF --graph-steps 20 --steps 20 --warmup 5 > gpurun_out/r05_c10_companion_only.json 2> gpurun_out/r05_c10_companion_only.err; tail -c 400 gpurun_out/r05_c10_companion_only.json; echo
timeout 300 python bench.py $OFF --graph-steps 0 --graphs --steps 20 --warmup 5 > gpurun_out/r05_c10_graphs.json 2> gpurun_out/r05_c10_graphs.err; tail -c 300 gpurun_out/r05_c10_graphs.json; echo
timeout 300 python bench.py $OFF --graph-steps 20 --no-prof --clean-steps 0 --steps 20 --warmup 5 > gpurun_out/r05_c10_companion_noprof.json 2> gpurun_out/r05_c10_companion_noprof.err; tail -c 400 gpurun_out/r05_c10_companion_noprof.json; echo
