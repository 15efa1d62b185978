#!/bin/bash
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/convpmc
mkdir -p $O; cd /tmp && export TMPDIR=/tmp
for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_WAVES SQ_LEVEL_WAVES SQ_ACCUM_PREV_HIRES"; do
  n=$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --output-format csv -d $O/$n -o p -- python3 $R/scripts/conv_micro.py > $O/$n.log 2>&1 || echo "fail $c"
done
ls $O
