#!/usr/bin/env python3
"""Is the eval step bound by the host's launch rate?  K steps are enqueued without synchronisation: the time until the host has issued the
last launch against the time until the GPU has finished.  gpurun -- python scripts/host_vs_gpu.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bayesian-enhancement-model_amd"))
import torch  # noqa: E402

from bem.pipeline import BEMPipeline, build_nets, synthetic_pair  # noqa: E402

dev = torch.device("cuda", 0)
net1, net2 = build_nets(device=dev)
pipe = BEMPipeline(net1, net2, 16, 0.1)
lq, gt = synthetic_pair((8, 3, 256, 256), device=dev)
for i in range(3):
    pipe.enhance(lq, gt, 8, gt_mean=True, seed=i, sync=False)
torch.cuda.synchronize()
K = 10
t0 = time.perf_counter()
for i in range(K):
    pipe.enhance(lq, gt, 8, gt_mean=True, seed=10 + i, sync=False)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host issue time {1e3 * (t1 - t0) / K:.2f} ms per step; GPU done after {1e3 * (t2 - t0) / K:.2f} ms per step")
# the host alone: the same step with a tiny input keeps the launch count and shrinks the GPU work
lq2, gt2 = synthetic_pair((8, 3, 64, 64), device=dev)
for i in range(2):
    pipe.enhance(lq2, gt2, 8, gt_mean=True, seed=i, sync=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(K):
    pipe.enhance(lq2, gt2, 8, gt_mean=True, seed=10 + i, sync=False)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"64x64 input: host issue time {1e3 * (t1 - t0) / K:.2f} ms per step; GPU done after {1e3 * (t2 - t0) / K:.2f} ms per step")
