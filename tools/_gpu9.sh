cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for st in 8 16; do echo "== SGV_LANES_STRIP=$st"; SGV_LANES_STRIP=$st timeout 120 ./tools/ufd_lab 32 2>&1 | grep -E "libsgv|FIR"; done | tee gpurun_out/ufd_lab3.log
timeout 120 ./tools/ufd_lab 32 2>&1 | tee -a gpurun_out/ufd_lab3.log
timeout 120 ./tools/ufd_lab 32 256 2 2>&1 | grep -E "libsgv|FIR" | tee -a gpurun_out/ufd_lab3.log
