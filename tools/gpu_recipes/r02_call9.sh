# round 2, GPU call 9+: ablations of the producer / consumer strided kernels (env WS: 1 | 2, ABLS: ablation ids, ORDERS: tile order)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for o in ${ORDERS:-0}; do for a in ${ABLS:-0}; do
  echo "=== ws ${WS:-1} ablation $a order $o"; timeout 120 tools/conv_s2_lab 5 ${WS:-1} $a 0 $o 2>&1 | grep "strided .*terms" | cut -c1-150
done; done > gpurun_out/r02_conv_s2_lab_ws_ablations.log 2>&1
cat gpurun_out/r02_conv_s2_lab_ws_ablations.log
