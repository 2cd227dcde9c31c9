# round 5, GPU call 14: per-launch timing inside captured graphs by device-clock timestamp kernels: the minimal repro, then the captured headline vs the eager one
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 120 python tools/graph_event_repro.py 2>&1 | tail -3 | cut -c1-400
bash tools/gpu_recipes/r05_call11.sh
