# round 3, GPU call 17: are the tile-end store bursts what the stride-2 kernels wait on?  pairs (ws=2) and convT (ws=1) with ABL 0 / 7 / 8 / 10 and staggered starts
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
for abl in 0 8 7 10; do
  echo "=== pairs (strided) ABL $abl"; timeout 120 tools/conv_s2_lab 5 2 $abl 0 2>&1 | grep "strided .*terms=3" | grep -v check | cut -c1-120
  echo "=== convT ABL $abl (NO_EDGE)"; NO_EDGE=1 timeout 120 tools/conv_s2_lab 5 1 $abl 1 2>&1 | grep "transposed .*terms=3" | grep -v check | cut -c1-120
done
for st in 1 2 4 8; do
  echo "=== pairs STAGGER $st"; STAGGER=$st timeout 120 tools/conv_s2_lab 5 2 0 0 2>&1 | grep "strided .*terms=3" | grep -v check | cut -c1-120
  echo "=== convT STAGGER $st (NO_EDGE)"; NO_EDGE=1 STAGGER=$st timeout 120 tools/conv_s2_lab 5 1 0 1 2>&1 | grep "transposed .*terms=3" | grep -v check | cut -c1-120
done
} > gpurun_out/r03_s2_lab_store_burst.log 2>&1
cat gpurun_out/r03_s2_lab_store_burst.log
