"""Resume helpers of the training driver (basicsr/utils/misc.py:94-124, basicsr/train.py:74-94)."""
import os
import os.path as osp

import torch


def check_resume(opt, resume_iter):
    """When resuming, point every ``pretrain_network_*`` at ``<models>/net_*_<iter>.pth`` and read 'params' (misc.py:94-124)."""
    if opt["path"].get("resume_state"):
        networks = [k for k in opt.keys() if k.startswith("network_")]
        if any(opt["path"].get(f"pretrain_{n}") is not None for n in networks):
            print("pretrain_network path will be ignored during resuming.")
        for n in networks:
            name = f"pretrain_{n}"
            base = n.replace("network_", "")
            ign = opt["path"].get("ignore_resume_networks")
            if ign is None or n not in ign:
                opt["path"][name] = osp.join(opt["path"]["models"], f"net_{base}_{resume_iter}.pth")
                print(f"Set {name} to {opt['path'][name]}")
        for k in [k for k in opt["path"].keys() if k.startswith("param_key")]:
            if opt["path"][k] == "params_ema":
                opt["path"][k] = "params"
                print(f"Set {k} to params")


def load_resume_state(opt, experiments_root="experiments"):
    """``--auto_resume``: the newest ``<iter>.state`` under experiments/<name>/training_states; else ``path.resume_state`` (train.py:74-94).
    The state file is read with the non-executing loader (it holds tensors, numbers, lists and dicts only)."""
    path = None
    if opt.get("auto_resume"):
        d = osp.join(experiments_root, opt["name"], "training_states")
        if osp.isdir(d):
            its = [float(f[:-len(".state")]) for f in os.listdir(d) if f.endswith(".state")]
            if its:
                path = osp.join(d, f"{max(its):.0f}.state")
                opt["path"]["resume_state"] = path
    elif opt["path"].get("resume_state"):
        path = opt["path"]["resume_state"]
    if path is None:
        return None
    dev = f"cuda:{torch.cuda.current_device()}" if torch.cuda.is_available() else "cpu"
    state = torch.load(path, map_location=dev, weights_only=True)
    check_resume(opt, state["iter"])
    return state
