# round 3, GPU call 25: row-reuse consumer order (ORD = 1) of the stride-1 kernel against the tap-major order, lab
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 tools/conv_lab 5 ws 2>&1 | grep -v amdgpu.ids | grep "check ws\|ms" > gpurun_out/r03_conv_lab_ws_row_reuse.log
cat gpurun_out/r03_conv_lab_ws_row_reuse.log
for i in 0 1 6; do timeout 120 tools/conv_lab 5 ws abl $i 2>&1 | grep "ws ablation\|fault\|error\|coredump" ; done
