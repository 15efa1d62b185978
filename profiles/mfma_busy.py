#!/usr/bin/env python3
"""Matrix-pipe busy share per kernel from one rocprofv3 counter pass (--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE).

  python profiles/mfma_busy.py <counter_collection.csv> <out.json>

Per kernel (summed over its launches): mfma_busy_cycles (SQ_VALU_MFMA_BUSY_CYCLES counts cycles, summed over every SIMD of the chip:
32 per v_mfma_f32_32x32x16_bf16, 64 per v_mfma_f32_32x32x2_f32 -- MI355X_MICROARCH.md cycle-constants table) and
    mfma_util = mfma_busy_cycles / (1024 SIMDs x kernel cycles),   kernel cycles = GRBM_GUI_ACTIVE / 8
(rocprofv3 reports GRBM_GUI_ACTIVE as the sum over the 8 XCDs, same guide, 'DVFS give-back').  ROCm 7.2 ships no gfx950 derived-metric
definitions (MfmaUtil falls back to gfx94x formulas), hence the explicit formula.  Calibration: the f32-MFMA im2col convolution
conv2d_mfma_kernel<4,4,2,*> reads 0.42-0.44 here and 50-53 TFLOP/s of 157 by its launch times (clock below 2.4 GHz under load)."""
import collections
import csv
import json
import re
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    n = re.sub(r"\(.*$", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")).strip()
    agg[n][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_BUSY_CYCLES":
        cnt[n] += 1
out = {"_note": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE of `python bench.py --steps 1 --warmup 1 --no-cpu-baseline`; "
                "sums over all launches of a kernel; mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 * GRBM_GUI_ACTIVE / 8)", "kernels": {}}
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)):
    mf, sq = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), v.get("SQ_BUSY_CYCLES", 0.0)
    out["kernels"][k] = {"launches": cnt[k], "mfma_busy_cycles": mf, "sq_busy_cycles": sq, "gui_active": v.get("GRBM_GUI_ACTIVE", 0.0),
                         "mfma_util": (mf / (1024.0 * v.get("GRBM_GUI_ACTIVE", 0.0) / 8.0)) if v.get("GRBM_GUI_ACTIVE", 0.0) else None}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(f"{len(out['kernels'])} kernels -> {sys.argv[2]}")
