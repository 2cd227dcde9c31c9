# round 3, GPU call 26: which producer stream costs the consumers their cycles -- the weight DMA or the x loads (lab ablations 8 / 9)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 0 5 8 9; do timeout 120 tools/conv_lab 5 ws abl $i 2>&1 | grep "ws ablation\|fault\|error\|coredump" ; done | tee -a gpurun_out/r03_conv_lab_ws_ablations.log
