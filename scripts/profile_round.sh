#!/bin/bash
# One gpurun call: the round's evidence for profiles/.  Usage: gpurun --timeout 1200 -- 'bash scripts/profile_round.sh r03 <commit>'
#   bench lines (eval, train) | kernel trace + stats of both | PMC passes of the eval step: FETCH_SIZE, WRITE_SIZE (separate runs, per
#   MI355X_MICROARCH.md) and SQ_VALU_MFMA_BUSY_CYCLES + SQ_BUSY_CYCLES + GRBM_GUI_ACTIVE (matrix-pipe busy share of the GEMM / conv kernels)
# Counter passes never combine with trace domains other than the kernel trace (gpurun refuses that).  The program after `--` is python3 itself.
set -e
TAG=${1:-r03}; COMMIT=${2:-unknown}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err
python3 $R/bench.py --config train --steps 5 --warmup 2 > $O/bench_train.json 2> $O/bench_train.err
python3 $R/bench.py --config train1 --steps 5 --warmup 2 > $O/bench_train1.json 2> $O/bench_train1.err
BEM_STAGE1_GRAPH=0 python3 $R/bench.py --config train1 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_train1_launched.json 2> $O/bench_train1_launched.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/kt.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_train -o kt -- python3 $R/bench.py --config train --steps 3 --warmup 1 --no-cpu-baseline > $O/kt_train.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -o f -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -o w -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/pmc_w.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_m -o m -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/pmc_m.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_ft -o f -- python3 $R/bench.py --config train --steps 1 --warmup 1 --no-cpu-baseline > $O/pmc_ft.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_wt -o w -- python3 $R/bench.py --config train --steps 1 --warmup 1 --no-cpu-baseline > $O/pmc_wt.log 2>&1
python3 $R/profiles/summarize.py $(find $O/kt -name "*kernel_trace.csv" | head -1) 45 > $O/kernel_trace_summary.txt
python3 $R/profiles/summarize.py $(find $O/kt_train -name "*kernel_trace.csv" | head -1) 45 > $O/train_kernel_trace_summary.txt
python3 $R/profiles/make_traffic.py $(find $O/pmc_f -name "*counter_collection.csv" | head -1) $(find $O/pmc_w -name "*counter_collection.csv" | head -1) $O/traffic.json 2 $COMMIT
python3 $R/profiles/make_traffic.py $(find $O/pmc_ft -name "*counter_collection.csv" | head -1) $(find $O/pmc_wt -name "*counter_collection.csv" | head -1) $O/traffic_train.json 2 $COMMIT
python3 $R/profiles/mfma_busy.py $(find $O/pmc_m -name "*counter_collection.csv" | head -1) $O/mfma_busy.json
cat $O/bench.json; cat $O/bench_train.json; cat $O/bench_train1.json; head -12 $O/kernel_trace_summary.txt
