#!/bin/bash
# SQ / GRBM counter passes over one kernel of a micro-benchmark script: clock held, matrix-pipe and VALU busy shares, co-execution, waits, LDS.
#   gpurun -- 'bash scripts/kernel_pmc.sh scripts/ss2d_front_micro.py ss2d_front_x6_kernel tag'
set -e
SCRIPT=$1; KERN=$2; TAG=${3:-kpmc}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -o p -- python3 $R/$SCRIPT 3 > $O/p$i.log 2>&1 || { tail -5 $O/p$i.log; continue; }
done
python3 - <<PY
import csv,glob,collections,re
tot=collections.defaultdict(float); n=collections.defaultdict(int); grid={}
for f in glob.glob("$O/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "$KERN" in r["Kernel_Name"]:
            m=re.sub(r"\(.*","",r["Kernel_Name"]).replace("void (anonymous namespace)::","")+" grid "+r.get("Grid_Size","?")
            tot[(m, r["Counter_Name"])]+=float(r["Counter_Value"]); n[(m, r["Counter_Name"])]+=1
dur=collections.defaultdict(list)
for f in glob.glob("$O/p*/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "$KERN" in r["Kernel_Name"]:
            m=re.sub(r"\(.*","",r["Kernel_Name"]).replace("void (anonymous namespace)::","")+" grid "+str(int(r.get("Grid_Size_X","0"))*int(r.get("Grid_Size_Y","1"))*int(r.get("Grid_Size_Z","1")))
            dur[m].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k in sorted(dur): print("%-60s median %8.1f us (%d launches)" % (k, sorted(dur[k])[len(dur[k])//2], len(dur[k])))
for k in sorted(tot): print("%-60s %-32s %16.0f (%d)" % (k[0], k[1], tot[k]/n[k], n[k]))
PY
