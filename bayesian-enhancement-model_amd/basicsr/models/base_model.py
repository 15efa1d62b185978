"""Model wrapper base (basicsr/models/base_model.py): device placement, checkpoint saving / loading (:236-343), training-state
save / resume (:345-394), schedulers (:124-168), learning-rate update with linear warm-up (:209-230), loss-dict reduction (:396-421).
File formats are the reference's: ``<label>_<iter>.pth`` = {param_key: state_dict on CPU}, ``<iter>.state`` = {epoch, iter,
optimizers: [state_dict], schedulers: [state_dict], best_metric, ...}; both load with ``torch.load(weights_only=True)``."""
import os
import time
from collections import OrderedDict

import torch

from basicsr.models import lr_scheduler


class _LazyLog(OrderedDict):
    """log_dict whose values are device scalars until somebody reads them (the reference calls .item() inside the step, one host
    synchronisation per iteration; here the step stays asynchronous and the read pays it)."""

    def __getitem__(self, k):
        v = super().__getitem__(k)
        if torch.is_tensor(v):
            v = float(v.mean().item())
            super().__setitem__(k, v)
        return v

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def values(self):
        return [self[k] for k in self.keys()]


class BaseModel:
    def __init__(self, opt):
        self.opt = opt
        self.device = torch.device("cuda" if opt.get("num_gpu", 1) != 0 and torch.cuda.is_available() else "cpu")
        self.is_train = opt.get("is_train", False)
        self.schedulers, self.optimizers = [], []
        self.log_dict = OrderedDict()

    def model_to_device(self, net):
        return net.to(self.device)

    def sync_gradients(self, optimizer):
        """opt['dist']: average the gradients over the ranks before clipping (what the reference gets from DistributedDataParallel,
        base_model.py:97-100); without an initialised process group a distributed option file is an error, not N silent replicas."""
        if not self.opt.get("dist"):
            return
        if not (torch.distributed.is_available() and torch.distributed.is_initialized()):
            raise RuntimeError("opt['dist'] is set but torch.distributed is not initialised (launch with --launcher pytorch)")
        optimizer.all_reduce_grads()

    def get_bare_model(self, net):
        return net.module if hasattr(net, "module") else net

    def load_network(self, net, load_path, strict=True, param_key="params"):
        ck = torch.load(load_path, map_location="cpu", weights_only=True)
        sd = ck[param_key] if param_key is not None and param_key in ck else ck
        sd = {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}
        net.load_state_dict(sd, strict=strict)

    def save(self, epoch, current_iter):
        """Save networks and training state (overridden by the model classes)."""

    def save_network(self, net, net_label, current_iter, param_key="params"):
        """``<models>/<net_label>_<iter>.pth`` (iter -1 -> 'latest'); net / param_key may be lists of equal length (base_model.py:236-280)."""
        if self.opt.get("rank", 0) != 0:                       # @master_only in the reference (base_model.py:235)
            return None
        it = "latest" if current_iter == -1 else current_iter
        save_path = os.path.join(self.opt["path"]["models"], f"{net_label}_{it}.pth")
        nets = net if isinstance(net, list) else [net]
        keys = param_key if isinstance(param_key, list) else [param_key]
        assert len(nets) == len(keys), "The lengths of net and param_key should be the same."
        save_dict = {}
        for n, k in zip(nets, keys):
            sd = OrderedDict()
            for name, t in self.get_bare_model(n).state_dict().items():
                sd[name[7:] if name.startswith("module.") else name] = t.detach().cpu()
            save_dict[k] = sd
        os.makedirs(os.path.dirname(save_path), exist_ok=True)
        retry, err = 3, None
        while retry > 0:                                       # "avoid occasional writing errors" (:265-278)
            try:
                torch.save(save_dict, save_path)
                return save_path
            except OSError as e:
                err = e
                time.sleep(1)
            retry -= 1
        raise IOError(f"Cannot save {save_path}.") from err

    def save_training_state(self, epoch, current_iter, **kwargs):
        """``<training_states>/<iter>.state`` with every optimizer's and scheduler's state_dict (base_model.py:345-376); rank 0 only."""
        if self.opt.get("rank", 0) != 0 or current_iter == -1:
            return None
        state = {"epoch": epoch, "iter": current_iter, "optimizers": [], "schedulers": []}
        state.update(kwargs)
        state["optimizers"] = [o.state_dict() for o in self.optimizers]
        state["schedulers"] = [s.state_dict() for s in self.schedulers]
        state["best_metric"] = kwargs["best_metric"]           # the reference requires it (:370)
        save_path = os.path.join(self.opt["path"]["training_states"], f"{current_iter}.state")
        os.makedirs(os.path.dirname(save_path), exist_ok=True)
        torch.save(state, save_path)
        return save_path

    def resume_training(self, resume_state):
        """Reload optimizers and schedulers (base_model.py:379-394)."""
        ro, rs = resume_state["optimizers"], resume_state["schedulers"]
        assert len(ro) == len(self.optimizers), "Wrong lengths of optimizers"
        assert len(rs) == len(self.schedulers), "Wrong lengths of schedulers"
        for o, sd in zip(self.optimizers, ro):
            o.load_state_dict(sd)
        for s, sd in zip(self.schedulers, rs):
            s.load_state_dict(sd)

    # -- training-side helpers -------------------------------------------------------------------------------------
    def setup_schedulers(self):
        sch = dict(self.opt["train"]["scheduler"])
        kind = sch.pop("type")
        if kind == "CosineAnnealingRestartCyclicLR":
            for o in self.optimizers:
                self.schedulers.append(lr_scheduler.CosineAnnealingRestartCyclicLR(o, **sch))
        elif kind == "TrueCosineAnnealingLR":
            for o in self.optimizers:
                self.schedulers.append(torch.optim.lr_scheduler.CosineAnnealingLR(o, **sch))
        else:
            raise NotImplementedError(f"Scheduler {kind} is not implemented yet.")

    def _get_init_lr(self):
        return [[g["initial_lr"] for g in o.param_groups] for o in self.optimizers]

    def _set_lr(self, lr_groups_l):
        for o, lrs in zip(self.optimizers, lr_groups_l):
            for g, lr in zip(o.param_groups, lrs):
                g["lr"] = lr

    def update_learning_rate(self, current_iter, warmup_iter=-1):
        if current_iter > 1:
            for s in self.schedulers:
                s.step()
        if current_iter < warmup_iter:
            self._set_lr([[v / warmup_iter * current_iter for v in g] for g in self._get_init_lr()])

    def get_current_learning_rate(self):
        return [g["lr"] for g in self.optimizers[0].param_groups]

    def get_current_log(self):
        return self.log_dict

    def reduce_loss_dict(self, loss_dict):
        """Average over ranks when distributed (one reduce of the stacked scalars to rank 0, base_model.py:405-415)."""
        with torch.no_grad():
            if self.opt.get("dist"):
                keys = list(loss_dict)
                losses = torch.stack([loss_dict[k].reshape(()) for k in keys], 0)
                torch.distributed.reduce(losses, dst=0)
                if self.opt.get("rank", 0) == 0:
                    losses /= self.opt["world_size"]
                loss_dict = {k: v for k, v in zip(keys, losses)}
            return _LazyLog((k, v.detach()) for k, v in loss_dict.items())
