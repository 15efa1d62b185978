"""Throughput at the BASELINE config-5 geometry on one GPU: 400x600 images (padded to 448x640), N = 16 samples."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "bayesian-enhancement-model_amd"))
import torch
from bem.pipeline import BEMPipeline, build_nets, synthetic_pair
net1, net2 = build_nets(device="cuda")
pipe = BEMPipeline(net1, net2)
B, N = int(sys.argv[1]) if len(sys.argv) > 1 else 4, 16
lq, gt = synthetic_pair((B, 3, 400, 600), device="cuda")
for i in range(2):
    pipe.enhance(lq, gt, N, seed=i, sync=False)
torch.cuda.synchronize(); t0 = time.perf_counter()
K = 3
for i in range(K):
    r = pipe.enhance(lq, gt, N, seed=10 + i, sync=False)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
print(f"config-5 geometry: {B} images x {N} samples at 400x600: {dt*1e3:.1f} ms/step = {B/dt:.1f} img/s, peak memory {torch.cuda.max_memory_allocated()/2**30:.1f} GiB, best {r['best'].tolist()}")
