set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/miopen_cache
export TMPDIR=/tmp
export MIOPEN_USER_DB_PATH=$GRAFT_REPO_ROOT/gpurun_out/miopen_cache
export MIOPEN_CUSTOM_CACHE_DIR=$GRAFT_REPO_ROOT/gpurun_out/miopen_cache
timeout 200 python -m pytest tests/test_networks.py -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/t3.log
timeout 420 python bench.py --steps 2 --warmup 2 --cpu-seconds 0 > gpurun_out/bench3.json 2> gpurun_out/bench3.err; tail -8 gpurun_out/bench3.err; cat gpurun_out/bench3.json
du -sh gpurun_out/miopen_cache; ls gpurun_out/miopen_cache | head
