set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
RND=${PMC_ROUND:-r03}
for tgt in ${@:-conv_one wgrad_group_one}; do
  for c in SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/${RND}_pmc/$tgt/$c -- python3 $R/tools/$tgt.py > /dev/null 2>&1
  done
done
find $R/gpurun_out/${RND}_pmc -name "*counter_collection.csv" | wc -l
