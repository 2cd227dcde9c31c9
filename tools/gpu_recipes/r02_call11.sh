# round 2, GPU call 11: producer / consumer transposed kernel in the lab (kind 1), with ablations
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
echo "=== checks (with edges)"; timeout 120 tools/conv_s2_lab 1 1 0 1 2>&1 | grep "check transposed"
for v in "0 0" "1 0" "1 6" "1 7"; do
  echo "=== ws/abl $v (NO_EDGE)"; NO_EDGE=1 timeout 120 tools/conv_s2_lab 5 $v 1 2>&1 | grep "transposed .*terms" | grep -v check | cut -c1-150
done
} > gpurun_out/r02_convT_lab_ws.log 2>&1
cat gpurun_out/r02_convT_lab_ws.log
