# round 6, call 15 (the "counter proof" branch): which queue the stride-2 members of the 3x3 family wait on.  Three separate --pmc passes of the
# eager step (2 iterations; --pmc with --kernel-trace only), per-kernel means; the stride-1 kernel (0.47 of its ceiling) is the control in every table.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_c15
mkdir -p $OUT
export TMPDIR=/tmp
OFF="--strict-steps 0 --bf16-steps 0 --pl-steps 0 --ada-steps 0 --lowp-steps 0 --split3-steps 0 --graph-steps 0"
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > $OUT/sq_counters_available.txt; wc -l $OUT/sq_counters_available.txt
pass() {
  name=$1; shift
  rm -rf /tmp/pmc_$name
  ( cd /tmp && timeout 400 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmc_$name -- python $GRAFT_REPO_ROOT/bench.py --eager --steps 2 --warmup 1 --cpu-seconds 0 --no-prof $OFF > /tmp/pmc_$name.log 2>&1; echo "pmc $name rc=$?" )
  python tools/pmc_kernel_table.py /tmp/pmc_$name > $OUT/pmc_$name.txt 2>/dev/null
  grep -E "^kernel|conv3x3_ws_kernel|s2_pairs|convT3x3_s2_ws|wrw3x3_s2|wrw3x3_ws" $OUT/pmc_$name.txt | cut -c1-260
}
pass A SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE
pass B SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VMEM
pass C SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT
pass D SQ_WAVES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_FLAT SQ_ACTIVE_INST_FLAT SQ_WAIT_INST_ANY SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL
