import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "bayesian-enhancement-model_amd"))
import torch
from oracle import bem_oracle as O
from bem.pipeline import BEMPipeline, build_nets, synthetic_pair
net1, net2 = build_nets(device="cuda")
sd1 = {k: v.detach().cpu() for k, v in net1.state_dict().items()}
sd2 = {k: v.detach().cpu() for k, v in net2.state_dict().items()}
for shape in ((1, 3, 400, 600), (1, 3, 256, 256)):
    lq, gt = synthetic_pair(shape, seed=5)
    hd, wd = -(-shape[2] // 64) * 4, -(-shape[3] // 64) * 4
    noise = torch.randn(1, 3, hd, wd, generator=torch.Generator().manual_seed(9))
    ref = O.eval_mc_ref(sd1, sd2, lq, gt, 1, deterministic=True, gt_mean=True, noise_list=[noise], scan=O.selective_scan_c)
    out = BEMPipeline(net1, net2).enhance(lq.cuda(), gt.cuda(), 1, gt_mean=True, deterministic=True, noise=noise.cuda())
    a = out["final"][0].permute(1, 2, 0).cpu(); b = torch.from_numpy(ref["finals"][0])
    d = (a - b).abs()
    print(shape, "max", d.max().item(), "mean", d.mean().item(), "p99.9", d.flatten().kthvalue(int(d.numel()*0.999)).values.item(),
          "psnr", float(out["psnr"][0]), ref["psnr"][0], "cond diff", (out["conds"].cpu() - ref["conds"][0]).abs().max().item(),
          "raw diff", (out["raw"].cpu()[..., :shape[2], :shape[3]].clamp(0,1) - ref["preds"][0]).abs().max().item())
    idx = d.argmax(); print("  argmax at", divmod(idx.item() // 3, shape[3]), "channel", idx.item() % 3)
