#!/bin/bash
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/scanpmc
mkdir -p $O; cd /tmp && export TMPDIR=/tmp
for full in 0 1; do
  export BEM_SCAN_FULL=$full
  rocprofv3 --kernel-trace --output-format csv -d $O/kt$full -o kt -- python3 $R/scripts/scan_micro.py > $O/kt$full.log 2>&1
  for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQ_INSTS_LDS SQ_INST_CYCLES_VMEM"; do
    n=$(echo $c | tr ' ' '_')
    rocprofv3 --pmc $c --output-format csv -d $O/p${full}_$n -o p -- python3 $R/scripts/scan_micro.py > $O/p${full}_$n.log 2>&1 || echo "fail $c"
  done
done
ls $O
