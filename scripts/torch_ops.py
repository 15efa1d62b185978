"""Which torch-side ops (copies, cats, fills ...) does one eval step still launch, and from where?
   python scripts/torch_ops.py   -> table by op, then the Python call sites of every aten::copy_ / cat / fill / index op of one step.
   python scripts/torch_ops.py train1   -> the same for one Stage-I training step launched kernel by kernel (in its captured form every such op is a graph node)."""
import collections, os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "bayesian-enhancement-model_amd"))
import torch
from torch.profiler import profile, ProfilerActivity
from torch.utils._python_dispatch import TorchDispatchMode
from bem.pipeline import BEMPipeline, build_nets, synthetic_pair
if len(sys.argv) > 1 and sys.argv[1] == "train1":
    os.environ["BEM_STAGE1_GRAPH"] = "0"
    from basicsr.models import build_model
    from basicsr.utils.options import parse as parse_opt
    from bem import ops
    opt = parse_opt(os.path.join(ROOT, "bayesian-enhancement-model_amd", "Options", "CG_UNet_LOLv1.yml"), is_train=True)
    opt["dist"], opt["rank"], opt["world_size"] = False, 0, 1
    torch.manual_seed(100)
    model = build_model(opt)
    lq, gt = synthetic_pair((8, 3, 128, 128), device="cuda")
    batch = dict(lq_down=ops.resize_down(lq, 16), gt=gt, gt_down=ops.resize_down(gt, 16), mask=(torch.rand(8, 8, 8) < 0.4).float().cuda())
    it = [0]

    class _P:
        def enhance(self, *a, **k):
            it[0] += 1
            model.update_learning_rate(it[0], warmup_iter=-1)
            model.feed_train_data(batch)
            model.optimize_parameters(it[0])
    pipe = _P()
else:
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


class Sites(TorchDispatchMode):
    """every aten op that launches device work, keyed by the innermost frame inside this repo"""
    def __init__(self):
        super().__init__()
        self.sites = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(s in name for s in ("view", "reshape", "empty", "as_strided", "detach", "alias", "_unsafe_view", "expand", "permute", "transpose",
                                       "select", "slice", "unsqueeze", "squeeze", "t.default", "size", "stride", "is_", "sym_", "_local_scalar")):
            fr = [f for f in traceback.extract_stack() if "bayesian-enhancement-model_amd" in f.filename]
            where = f"{os.path.basename(fr[-1].filename)}:{fr[-1].lineno}" if fr else "?"
            self.sites[(name, where)] += 1
        return func(*args, **(kwargs or {}))


with Sites() as s:
    pipe.enhance(lq, gt, 8, seed=6, sync=False)
    torch.cuda.synchronize()
for (name, where), c in s.sites.most_common(60):
    print(f"{c:5d}  {name:40s} {where}")
