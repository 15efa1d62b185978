#!/bin/bash
# kernel trace of the eval bench: bash scripts/kt_eval.sh <tag> [ENV=VAL ...]
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=$1; shift
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$TAG -o kt -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/kt_$TAG.log 2>&1
python3 $R/profiles/summarize.py $(find $O/kt_$TAG -name "*kernel_trace.csv" | head -1) 60 > $O/kt_${TAG}_summary.txt
