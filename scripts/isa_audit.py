#!/usr/bin/env python3
"""ISA audit: per kernel, count loads that are immediately followed by a full s_waitcnt vmcnt(0)
(= serialized memory latency), branches and MFMAs.  Usage: isa_audit.py file.hip"""
import re, subprocess, sys, tempfile, os
src = sys.argv[1]
out = tempfile.mktemp(suffix=".s")
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-fast-math", "-S", "--cuda-device-only", "-o", out, src],
               check=True, stderr=subprocess.DEVNULL, cwd="/tmp")
name, lines = None, []
kern = {}
for ln in open(out):
    m = re.match(r"^(_Z\w+):", ln)
    if m:
        name = m.group(1); kern[name] = []
    if name:
        kern[name].append(ln)
        if "s_endpgm" in ln:
            name = None
for k, ls in kern.items():
    ins = [l.split(";")[0].strip() for l in ls if l.strip() and not l.strip().startswith((";", "."))]
    ser = 0
    for i, l in enumerate(ins):
        if l.startswith(("global_load", "buffer_load")):
            nxt = [x for x in ins[i + 1:i + 4]]
            if any(x.startswith("s_waitcnt vmcnt(0)") for x in nxt):
                ser += 1
    nl = sum(l.startswith(("global_load", "buffer_load")) for l in ins)
    print(f"{k[:70]:70s} instr {len(ins):6d} loads {nl:4d} serialized {ser:4d} branches {sum(l.startswith('s_cbranch') for l in ins):4d} "
          f"mfma {sum(l.startswith('v_mfma') for l in ins):4d} scratch {sum('scratch_' in l for l in ins):4d}")
