#!/bin/bash
# SQ / GRBM counter passes over the fused gdMlp kernel alone (scripts/gdx_micro.py): clock held under load, matrix-pipe busy share,
# VALU / MFMA co-execution, where the waves wait.  Usage: gpurun -- 'bash scripts/gdx_pmc.sh [tag]'
set -e
TAG=${1:-gdx_pmc}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -o p -- python3 $R/scripts/gdx_micro.py 3 > $O/p$i.log 2>&1 || { tail -5 $O/p$i.log; continue; }
done
python3 - <<PY
import csv,glob,collections
tot=collections.defaultdict(float); n=collections.defaultdict(int)
for f in glob.glob("$O/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gdmlp_x6_kernel" in r["Kernel_Name"]:
            key=(r["Kernel_Name"].split("(")[0][-32:], r["Counter_Name"])
            tot[key]+=float(r["Counter_Value"]); n[key]+=1
dur=collections.defaultdict(list)
for f in glob.glob("$O/p*/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gdmlp_x6_kernel" in r["Kernel_Name"]:
            dur[r["Kernel_Name"].split("(")[0][-32:]].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k in sorted(dur): print(f"{k:34s} duration under counters: median {sorted(dur[k])[len(dur[k])//2]:.1f} us ({len(dur[k])} launches)")
for k in sorted(tot): print(f"{k[0]:34s} {k[1]:34s} {tot[k]/max(n[k],1):18.0f}   (per launch, {n[k]} launches)")
PY
