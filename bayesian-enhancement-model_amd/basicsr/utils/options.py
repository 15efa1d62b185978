"""Option-file parsing with the reference's result schema (basicsr/utils/options.py:13-35,220-278)."""
from collections import OrderedDict
from os import path as osp

import yaml


def ordered_yaml():
    """yaml Loader/Dumper that keep mapping order (safe loader: option files are plain data)."""
    Loader, Dumper = yaml.SafeLoader, yaml.SafeDumper

    class _L(Loader):
        pass

    class _D(Dumper):
        pass
    tag = yaml.resolver.BaseResolver.DEFAULT_MAPPING_TAG
    _D.add_representer(OrderedDict, lambda d, data: d.represent_dict(data.items()))
    _L.add_constructor(tag, lambda l, node: OrderedDict(l.construct_pairs(node)))
    return _L, _D


def parse(opt_path, is_train=True):
    with open(opt_path, "r") as f:
        Loader, _ = ordered_yaml()
        opt = yaml.load(f, Loader=Loader)
    opt["is_train"] = is_train
    opt["name"] = osp.basename(opt_path).split(".")[0]
    for phase, dataset in opt.get("datasets", {}).items():
        dataset["phase"] = phase.split("_")[0]
        if "scale" in opt:
            dataset["scale"] = opt["scale"]
        for k in ("dataroot_gt", "dataroot_lq"):
            if dataset.get(k) is not None:
                dataset[k] = osp.expanduser(dataset[k])
    for key, val in opt["path"].items():
        if val is not None and ("resume_state" in key or "pretrain_network" in key):
            opt["path"][key] = osp.expanduser(val)
    root = osp.abspath(osp.join(__file__, osp.pardir, osp.pardir, osp.pardir))
    opt["path"]["root"] = root
    if is_train:
        exp = osp.join(root, "experiments", opt["name"])
        opt["path"].update(experiments_root=exp, models=osp.join(exp, "models"),
                           training_states=osp.join(exp, "training_states"), log=exp,
                           visualization=osp.join(exp, "visualization"))
        if "debug" in opt["name"]:
            if "val" in opt:
                opt["val"]["val_freq"] = 8
            opt["logger"]["print_freq"] = 1
            opt["logger"]["save_checkpoint_freq"] = 8
    else:
        res = osp.join(root, "results", opt["name"])
        opt["path"].update(results_root=res, log=res, visualization=osp.join(res, "visualization"))
    return opt
