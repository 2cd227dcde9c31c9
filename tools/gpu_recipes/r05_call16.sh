# round 5, GPU call 16: the captured headline with the timing kernels in a second graph set that only the last timed iteration replays; full default line (companions on) + the eager headline
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
S=$(date +%s); timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_c16_default.json 2> gpurun_out/r05_c16_default.err; echo "rc=$?"; echo "wall $(( $(date +%s) - S )) s"; wc -c gpurun_out/r05_c16_default.json; cat gpurun_out/r05_c16_default.json; cp bench_detail.json gpurun_out/r05_c16_default_detail.json
