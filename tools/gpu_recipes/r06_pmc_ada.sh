cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c45
export TMPDIR=/tmp
rm -rf /tmp/pmc_ada1 /tmp/pmc_ada2
( cd /tmp && timeout 400 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE SQ_WAVES --kernel-trace --output-format csv -d /tmp/pmc_ada1 -- python $GRAFT_REPO_ROOT/tools/ada_bench.py --static 1 --rounds 1 > /tmp/pmc_ada1.log 2>&1; echo "pass1 rc=$?" )
( cd /tmp && timeout 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d /tmp/pmc_ada2 -- python $GRAFT_REPO_ROOT/tools/ada_bench.py --static 1 --rounds 1 > /tmp/pmc_ada2.log 2>&1; echo "pass2 rc=$?" )
python tools/pmc_kernel_table.py /tmp/pmc_ada1 /tmp/pmc_ada2 | grep -i "kernel \|ada_geo" > gpurun_out/c45/r06_pmc_ada_kernels.txt; cat gpurun_out/c45/r06_pmc_ada_kernels.txt | cut -c1-400
tail -3 /tmp/pmc_ada1.log
