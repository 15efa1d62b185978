"""Which torch-side ops (copies, cats, ...) does one eval step still launch?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "bayesian-enhancement-model_amd"))
import torch
from torch.profiler import profile, ProfilerActivity
from bem.pipeline import BEMPipeline, build_nets, synthetic_pair
net1, net2 = build_nets(device="cuda")
pipe = BEMPipeline(net1, net2)
lq, gt = synthetic_pair((8, 3, 256, 256), device="cuda")
for i in range(2):
    pipe.enhance(lq, gt, 8, seed=i, sync=False)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=False) as prof:
    pipe.enhance(lq, gt, 8, seed=5, sync=False)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="count", row_limit=25, max_name_column_width=50))
