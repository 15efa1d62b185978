#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 counter passes (--pmc FETCH_SIZE, --pmc WRITE_SIZE; separate runs).

  python profiles/make_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> [steps] [commit]

Units and the gfx950 correction follow MI355X_MICROARCH.md (HBM section): both counters are in KiB; FETCH_SIZE counts 64 B per 128-B request and
is doubled; WRITE_SIZE is exact.  The guide calibrates that factor for 16-byte-per-lane streaming reads only and asks for a calibration of
other access widths: profiles/r03_fetch_write_counter_calibration.txt (scripts/probes/fetch_calib.hip, scripts/fetch_calib.sh) streams 1 GiB
with coalesced 4-, 8- and 16-byte loads and with the 128-byte-per-half-wave row pattern of the NCHW kernels -- FETCH_SIZE reports exactly
half of the bytes in every case (factor 2.000), WRITE_SIZE exactly the bytes for 4-, 8- and 16-byte stores (factor 1.000).  Every global
load of this repo's kernels is one of those coalesced forms, so one factor applies to all kernels:  hbm_bytes = 2 * FETCH_SIZE + WRITE_SIZE."""
import collections
import csv
import json
import re
import sys

FETCH_FACTOR = 2.0


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
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
rows, total = [], 0.0
for k in sorted(f, key=lambda k: -(FETCH_FACTOR * f[k][1] + w.get(k, [0, 0.0])[1])):
    n = f[k][0]
    fb, wb = f[k][1] / n, (w[k][1] / w[k][0] if k in w and w[k][0] else 0.0)
    rows.append({"kernel": k, "launches": n, "fetch_size_bytes_raw": fb, "write_size_bytes": wb, "fetch_factor": FETCH_FACTOR,
                 "bytes_per_launch": FETCH_FACTOR * fb + wb})
    total += n * (FETCH_FACTOR * fb + wb)
out = {"_note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) of `python bench.py --steps 1 --warmup 1 --no-cpu-baseline`, "
                "averaged per launch, KiB -> bytes; FETCH_SIZE x 2 (calibrated for every access width used, see the docstring of "
                "profiles/make_traffic.py); WRITE_SIZE as is.  bytes_per_launch = 2 * FETCH + WRITE.  The process runs the step `steps` times "
                "(warm-up + timed) after building the nets: total_bytes includes the one-off weight packing of the first step.",
       "commit": sys.argv[5] if len(sys.argv) > 5 else None, "steps_in_process": steps, "total_bytes": total, "bytes_per_step": total / steps,
       "kernels": rows}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(f"{len(rows)} kernels, {total / steps / 1e9:.1f} GB per step -> {sys.argv[3]}")
