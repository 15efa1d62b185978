#!/bin/bash
# SQ counter passes over the fused gdMlp front half alone (scripts/pig_time.py): where do its waves wait?
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pig_pmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $O/p$i -o p -- python3 $R/scripts/pig_time.py > $O/p$i.log 2>&1 || { tail -5 $O/p$i.log; continue; }
done
python3 - <<PY
import csv,glob,collections
tot=collections.defaultdict(float); n=collections.defaultdict(int)
for f in glob.glob("$O/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "pi_gate_x6_kernel" in r["Kernel_Name"]:
            tot[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]]+=1
for k in sorted(tot): print(f"{k:36s} {tot[k]/max(n[k],1):16.0f}   (per launch, {n[k]} launches)")
PY
