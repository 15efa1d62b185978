#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per kernel and launch shape, count / avg / total."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
agg = collections.defaultdict(lambda: [0, 0.0])
tot = 0.0
for r in rows:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    n = n.split("(")[0][:34]
    key = (n, int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), r["Grid_Size_Y"], r["Grid_Size_Z"])
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg[key][0] += 1
    agg[key][1] += d
    tot += d
print(f"total kernel time {tot / 1e3:.2f} ms over {len(rows)} launches")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{k[0]:34s} grid({k[1]},{k[2]},{k[3]}) x{v[0]:4d}  avg {v[1] / v[0]:9.1f} us  total {v[1] / 1e3:8.2f} ms  {100 * v[1] / tot:5.1f}%")
