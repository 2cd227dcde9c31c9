# round 4, GPU call 26: SQ counters of the step's kernels on the final sources (own passes, --pmc with --kernel-trace only): matrix-pipe busy share, LDS
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
OFF="--cpu-seconds 0 --strict-steps 0 --bf16-steps 0 --pl-steps 0 --lowp-steps 0 --split3-steps 0 --ada-steps 0 --no-prof"
B="python $GRAFT_REPO_ROOT/bench.py $OFF --steps 2 --warmup 1"
rm -rf /tmp/pmc26
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA --kernel-trace --output-format csv -d /tmp/pmc26/mfma -- $B > /tmp/pmc26_mfma.log 2>&1; echo "pmc MFMA rc=$?" )
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d /tmp/pmc26/lds -- $B > /tmp/pmc26_lds.log 2>&1; echo "pmc LDS rc=$?"; tail -2 /tmp/pmc26_lds.log )
python tools/pmc_kernel_table.py /tmp/pmc26/mfma > gpurun_out/r04_pmc_bench_step_MFMA_table.txt
python tools/pmc_kernel_table.py /tmp/pmc26/lds > gpurun_out/r04_pmc_bench_step_LDS_table.txt 2>/dev/null
grep -E "^kernel|conv3x3_ws|s2_pairs|convT3x3_s2_ws|wrw3x3|conv3x3_small|gemm_bf16x3_stream" gpurun_out/r04_pmc_bench_step_MFMA_table.txt | cut -c1-260
grep -E "^kernel|conv3x3_ws|s2_pairs|convT3x3_s2_ws|wrw3x3" gpurun_out/r04_pmc_bench_step_LDS_table.txt | cut -c1-260
