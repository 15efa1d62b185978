#!/bin/bash
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/kt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/kt.log 2>&1
tail -1 $O/kt.log | cut -c1-200
