# round 3: bisect the product tile kernel against the lab kernel, two interleaved passes per binary (boxes differ by a few per cent; A/B only inside one call)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
for pass in 1 2; do
for e in _exp0 _exp8 _exp16 _exp24; do
    echo "== pass $pass binary ufd_lab$e, N=96 257->256"
    timeout 120 tools/ufd_lab$e 96 257 1 2>&1 | grep -E "V6 LDS tile, loads up front, 16 rows NT0|V7 product|mismatch"
done
done
} > gpurun_out/r03_ufd_lab10.log 2>&1
cat gpurun_out/r03_ufd_lab10.log
