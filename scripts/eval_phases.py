#!/usr/bin/env python3
"""GPU time of the phases of one eval step (8 images x 8 samples at 256x256): Stage I, decomp(image), Stage II by U-Net stage, finalisation.
HIP events on the launch stream around each phase (BEM_DECOMP_OVERLAP=0 so that the phases are sequential); 5 timed steps after 2 warm-ups,
no host synchronisation between the steps.
  gpurun -- python scripts/eval_phases.py"""
import os
import sys

os.environ["BEM_DECOMP_OVERLAP"] = "0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bayesian-enhancement-model_amd"))
import torch  # noqa: E402

from bem.pipeline import BEMPipeline, build_nets, synthetic_pair  # noqa: E402

dev = torch.device("cuda", 0)
net1, net2 = build_nets(device=dev)
pipe = BEMPipeline(net1, net2, 16, 0.1)
lq, gt = synthetic_pair((8, 3, 256, 256), device=dev)
marks = []


def mark(name):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    marks.append((name, e))


def hook(mod, name, pre):
    if pre:
        mod.register_forward_pre_hook(lambda m, a: mark(name))
    else:
        mod.register_forward_hook(lambda m, a, o: mark(name))


hook(net1, "start stage1", True); hook(net1, "stage1", False)
for br in ("Q1", "Q2"):
    hook(getattr(net2, f"first_conv_{br}"), f"{br} prep (decomp + copies)", True)
    enc, dec = getattr(net2, f"encoders_{br}"), getattr(net2, f"decoders_{br}")
    hook(enc[0], f"{br} first_conv", True); hook(enc[0], f"{br} enc level0 (2 VSS blocks)", False)
    hook(enc[1], f"{br} down0", True); hook(enc[1], f"{br} enc level1 (2 VSS blocks)", False)
    hook(dec[0], f"{br} .. before dec level1", True)
    hook(dec[0]["block"], f"{br} dec1 up+fuse", True); hook(dec[0]["block"], f"{br} dec level1 (2 VSS blocks)", False)
    hook(dec[1]["block"], f"{br} dec0 up+fuse", True); hook(dec[1]["block"], f"{br} dec level0 (2 VSS blocks)", False)
hook(net2.bottleneck_block, "down1 + bottleneck fuse", True); hook(net2.bottleneck_block, "bottleneck level2 (2 VSS blocks)", False)
tot, runs = {}, []
for it in range(7):                      # no synchronisation between steps: the host runs ahead of the GPU as it does in bench.py
    marks = []
    mark("begin")
    pipe.enhance(lq, gt, 8, gt_mean=True, seed=1000 + it, sync=False)
    mark("end (proj, iwt+hamilton, finalize, select)")
    runs.append(marks)
torch.cuda.synchronize()
for marks in runs[2:]:
    for (n0, e0), (n1, e1) in zip(marks[:-1], marks[1:]):
        tot[n1] = tot.get(n1, 0.0) + e0.elapsed_time(e1) / 5
s = sum(tot.values())
for k, v in tot.items():
    print(f"{k:45s} {v:8.3f} ms  {100 * v / s:5.1f} %")
print(f"{'step':45s} {s:8.3f} ms")
