# round 2, GPU call 2: ws ablations / priorities, SQ counters on the lab, fixed tests
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 240 tools/conv_lab 5 ws > gpurun_out/r02_conv_lab_ws2.log 2>&1; echo "lab rc=$?"; grep "ablation\|4-wave" gpurun_out/r02_conv_lab_ws2.log
cd /tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*\|GRBM_[A-Z_]*" | sort -u | tr '\n' ' ' > $GRAFT_REPO_ROOT/gpurun_out/r02_counters_available.txt
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_lab_a -- $GRAFT_REPO_ROOT/tools/conv_lab 2 ws > $GRAFT_REPO_ROOT/gpurun_out/pmc_lab_a.log 2>&1; echo "pmc a rc=$?"
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_lab_b -- $GRAFT_REPO_ROOT/tools/conv_lab 2 ws > $GRAFT_REPO_ROOT/gpurun_out/pmc_lab_b.log 2>&1; echo "pmc b rc=$?"
cd $GRAFT_REPO_ROOT
python tools/pmc_kernel_table.py gpurun_out/pmc_lab_a gpurun_out/pmc_lab_b > gpurun_out/r02_pmc_conv_lab_table.txt 2>&1; head -40 gpurun_out/r02_pmc_conv_lab_table.txt | cut -c1-400
tail -3 gpurun_out/pmc_lab_b.log
rm -rf gpurun_out/pmc_lab_a gpurun_out/pmc_lab_b
timeout 600 python -m pytest tests/test_fused_conv_gpu.py tests/test_networks.py -m gpu -q --timeout 300 > gpurun_out/r02_t2.log 2>&1; echo "pytest rc=$?"; grep -v amdgpu.ids gpurun_out/r02_t2.log | tail -15
