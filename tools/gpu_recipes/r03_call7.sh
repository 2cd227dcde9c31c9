# round 3, GPU call 7: where does the product tile kernel lose against the lab kernel?  V7 = the product kernel launched directly from the lab harness
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
{
for cfg in "32 257 1" "96 257 1" "96 256 2"; do
    echo "== N IH pad: $cfg"
    timeout 120 tools/ufd_lab $cfg 2>&1 | grep -E "libsgv|copy2 ntl1|V6 LDS tile, loads up front, 16 rows NT0|V7|mismatch"
done
} > gpurun_out/r03_ufd_lab7.log 2>&1
cat gpurun_out/r03_ufd_lab7.log
timeout 300 python -m pytest tests/test_fc_gpu.py -m gpu -q 2>&1 | tail -2
