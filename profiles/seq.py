#!/usr/bin/env python3
"""Print the launch sequence (last third of a kernel trace) with durations, optionally filtered by substring."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
seq = rows[2 * len(rows) // 3:]
lo, hi = int(sys.argv[2]), int(sys.argv[3])
for r in seq[lo:hi]:
    nm = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:30]
    g = (int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print(f"{nm:30s} {str(g):16s} {d:8.1f} us")
