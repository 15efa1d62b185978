#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 counter passes (--pmc FETCH_SIZE, --pmc WRITE_SIZE; separate runs).

  python profiles/make_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>

Units and the gfx950 correction follow MI355X_MICROARCH.md: both counters are in KiB; FETCH_SIZE under-reports wide
streaming reads by 2x on gfx950 and is doubled; WRITE_SIZE is exact.  hbm_bytes_per_launch = 2 * FETCH + WRITE."""
import collections
import csv
import json
import re
import sys


def per_kernel(path, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        n = re.sub(r"\(.*$", "", n).strip()
        agg[n][0] += 1
        agg[n][1] += float(r["Counter_Value"]) * 1024.0
    return agg


f, w = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
out = {"_note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) of `python bench.py --steps 1 --warmup 1 "
                "--no-cpu-baseline`, averaged per launch, KiB -> bytes; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts "
                "64 B per 128 B request on wide streaming reads); WRITE_SIZE as is. hbm_bytes_per_launch = 2*FETCH + WRITE.",
       "kernels": {}}
for k in sorted(f, key=lambda k: -(2 * f[k][1] + w.get(k, [0, 0.0])[1])):
    n = f[k][0]
    fb, wb = f[k][1] / n, (w[k][1] / w[k][0] if k in w and w[k][0] else 0.0)
    out["kernels"][k] = {"launches": n, "fetch_size_bytes_raw": fb, "write_size_bytes": wb, "hbm_bytes_per_launch": 2 * fb + wb}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(f"{len(out['kernels'])} kernels -> {sys.argv[3]}")
