#!/bin/bash
# One gpurun call: training-step bench line + kernel trace.  Usage: gpurun -- 'bash scripts/train_profile.sh'
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --config train --steps 3 --warmup 2 --no-cpu-baseline > $O/bench_train.json 2> $O/bench_train.err || { tail -30 $O/bench_train.err; exit 1; }
cat $O/bench_train.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_train -o kt -- python3 $R/bench.py --config train --steps 2 --warmup 1 --no-cpu-baseline > $O/kt_train.log 2>&1
python3 $R/profiles/summarize.py $(find $O/kt_train -name "*kernel_trace.csv" | head -1) 40 > $O/kt_train_summary.txt
cat $O/kt_train_summary.txt
