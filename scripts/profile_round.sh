#!/bin/bash
# One gpurun call: bench line + kernel trace + the two PMC passes.  Usage: gpurun -- 'bash scripts/profile_round.sh'
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -o f -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -o w -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/pmc_w.log 2>&1
find $O -name "*.csv" | head -20
cat $O/bench.json
