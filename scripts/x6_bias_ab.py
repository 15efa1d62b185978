#!/usr/bin/env python3
"""A/B for DESIGN.md section 6.4: does reading the GEMM bias back from LDS as one float4 per lane (the round-1 epilogue) ever
return a wrong value?  Every output of full-size launches (config-5 level-0 plane, 224x320 = 71680 pixels; the shapes of
project_in / out_proj / project_out) is compared with an f32 reference while a second stream keeps the memory system busy.
Run once per library:  BEM_HIP_LIB=.../libbem_hip_dbg.so python scripts/x6_bias_ab.py   (LDS float4 form)
                       python scripts/x6_bias_ab.py                                      (shipped form: bias in registers)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bayesian-enhancement-model_amd"))
import torch  # noqa: E402
from bem import native, ops  # noqa: E402

print("library:", native.LIB_PATH)
torch.manual_seed(0)
side = torch.cuda.Stream()
big = torch.randn(64, 320, 128, 128, device="cuda")            # 1.3 GB: the background stream's depthwise convolution input
dww = torch.randn(320, 1, 3, 3, device="cuda")
total_bad = 0
for K, M, use_res in ((40, 320, False), (40, 40, True), (160, 40, True), (320, 160, False), (80, 80, True)):
    x = torch.randn(4, K, 224, 320, device="cuda")
    w = torch.randn(M, K, device="cuda") * K ** -0.5
    b = torch.randn(M, device="cuda")
    r = torch.randn(4, M, 224, 320, device="cuda") if use_res else None
    ref = torch.einsum("mk,bkhw->bmhw", w.double(), x.double()).float() + b[None, :, None, None] + (r if use_res else 0)
    Wp = ops.pack_pw_weight(w, x6=True)
    bad = 0
    for rep in range(25):
        with torch.cuda.stream(side):
            ops.dwconv3x3(big, dww, None, mode=2)
        y = ops.pw_gemm(x, Wp, M, bias=b, res=r)
        bad += int(((y - ref).abs() > 1e-3).sum())
    torch.cuda.synchronize()
    total_bad += bad
    print(f"K={K} M={M} res={use_res}: {bad} outputs off by more than 1e-3 in 25 launches of {y.numel()} outputs")
print("TOTAL BAD", total_bad)
