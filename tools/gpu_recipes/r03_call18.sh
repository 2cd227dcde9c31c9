# round 3, GPU call 18: convT epilogue with one 8-byte store per (even, odd) column pair vs one dword store per accumulator
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
echo "=== checks (paired stores, with edges)"; timeout 120 tools/conv_s2_lab 1 1 0 1 2>&1 | grep "check transposed"
echo "=== convT paired stores (NO_EDGE)"; NO_EDGE=1 timeout 120 tools/conv_s2_lab 5 1 0 1 2>&1 | grep "transposed .*terms=" | grep -v check | cut -c1-120
echo "=== convT one store per accumulator (NO_EDGE)"; NO_EDGE=1 STAGGER=-1 timeout 120 tools/conv_s2_lab 5 1 0 1 2>&1 | grep "transposed .*terms=" | grep -v check | cut -c1-120
echo "=== convT consumers only, paired"; NO_EDGE=1 timeout 120 tools/conv_s2_lab 5 1 7 1 2>&1 | grep "transposed .*terms=3" | grep -v check | cut -c1-120
echo "=== convT with edges, paired"; timeout 120 tools/conv_s2_lab 5 1 0 1 2>&1 | grep "transposed .*terms=3" | grep -v check | cut -c1-120
} > gpurun_out/r03_convT_lab_paired_stores.log 2>&1
cat gpurun_out/r03_convT_lab_paired_stores.log
