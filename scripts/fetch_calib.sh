#!/bin/bash
# Counter calibration run (scripts/probes/fetch_calib.hip): FETCH_SIZE and WRITE_SIZE in separate passes, factor = bytes moved / counter bytes.
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/fetch_calib; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f -o f -- $R/scripts/probes/fetch_calib > $O/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w -o w -- $R/scripts/probes/fetch_calib > $O/w.log 2>&1
python3 - <<PY
import csv,glob,collections,re
B=float(1<<30)
for tag,ctr in (("f","FETCH_SIZE"),("w","WRITE_SIZE")):
    agg=collections.defaultdict(list)
    for f in glob.glob("$O/%s/**/*counter_collection.csv"%tag, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"]==ctr: agg[re.sub(r"\(.*","",r["Kernel_Name"])].append(float(r["Counter_Value"])*1024)
    for k,v in sorted(agg.items()):
        m=sorted(v)[len(v)//2]
        print("%-12s %-40s counter %14.0f bytes   factor (true / counter) %.3f" % (ctr,k,m,B/m if m else float('nan')))
PY
