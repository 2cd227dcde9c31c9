# round 3, GPU call 24: ablation table of the current stride-1 producer / consumer kernel on all four layer shapes (tools/conv_lab 5 ws abl [index])
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 6 7; do timeout 120 tools/conv_lab 5 ws abl $i 2>&1 | grep "ws ablation\|fault\|error\|coredump" ; done >> gpurun_out/r03_conv_lab_ws_ablations.log
cat gpurun_out/r03_conv_lab_ws_ablations.log
