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


def _postprocess_yml_value(value):
    """options.py:75-96: '~' / 'none' -> None, booleans, ints, floats, lists, else the string."""
    v = value.strip()
    if v in ("~",) or v.lower() == "none":
        return None
    if v.lower() == "true":
        return True
    if v.lower() == "false":
        return False
    if v.startswith("!!float"):
        return float(v.replace("!!float", ""))
    if v.lstrip("-").isdigit():
        return int(v)
    try:
        return float(v)
    except ValueError:
        pass
    if v.startswith("["):
        return yaml.safe_load(v)
    return v


def parse_options(root_path, is_train=True, argv=None):
    """Command line + option file of the training / test drivers (basicsr/utils/options.py:99-200): same flags (--opt --launcher
    --auto_resume --debug --local_rank --force_yml), same distributed / seed / path / debug handling, returns (opt, args).
    ``--force_yml a:b=v`` walks the keys instead of exec'ing a string.  Two extra flags select the tensor dataset shim of basicsr.data
    (the file-backed datasets are outside the HIP path): ``--synthetic N`` and ``--pairs file.pt``."""
    import argparse
    import random

    import torch
    p = argparse.ArgumentParser()
    p.add_argument("--opt", required=True, help="Path to option YAML file.")
    p.add_argument("--launcher", choices=["none", "pytorch", "slurm"], default="none", help="job launcher")
    p.add_argument("--auto_resume", action="store_true")
    p.add_argument("--debug", action="store_true")
    p.add_argument("--local_rank", type=int, default=0)
    p.add_argument("--force_yml", nargs="+", default=None, help="Force to update yml files. Examples: train:ema_decay=0.999")
    p.add_argument("--synthetic", type=int, default=0, help="train / validate on N seeded synthetic pairs (basicsr.data shim)")
    p.add_argument("--pairs", default=None, help="train / validate on a .pt file of dict(lq, gt) tensors (basicsr.data shim)")
    args = p.parse_args(argv)
    with open(args.opt, "r") as f:
        opt = yaml.load(f, Loader=ordered_yaml()[0])
    if args.launcher == "none":
        opt["dist"] = False
        print("Disable distributed.", flush=True)
        opt["rank"], opt["world_size"] = 0, 1
    else:
        if args.launcher == "slurm":
            raise NotImplementedError("launcher slurm: start the ranks with torch.distributed.run and --launcher pytorch")
        import os

        import torch.distributed as dist
        opt["dist"] = True
        local = int(os.environ.get("LOCAL_RANK", args.local_rank))
        if torch.cuda.is_available():
            torch.cuda.set_device(local)
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str((opt.get("dist_params") or {}).get("port", 29500)))
            backend = (opt.get("dist_params") or {}).get("backend", "nccl")     # "nccl" is RCCL on ROCm
            dist.init_process_group(backend if torch.cuda.is_available() else "gloo")
        opt["rank"], opt["world_size"] = dist.get_rank(), dist.get_world_size()
    seed = opt.get("manual_seed")
    if seed is None:
        seed = random.randint(1, 10000)
        opt["manual_seed"] = seed
    random.seed(seed + opt["rank"])
    torch.manual_seed(seed + opt["rank"])
    if args.force_yml is not None:
        for entry in args.force_yml:
            keys, value = entry.split("=")
            node, path = opt, [k.strip() for k in keys.strip().split(":")]
            for k in path[:-1]:
                node = node[k]
            if path[-1] not in node:
                raise KeyError(f"--force_yml {keys}: creating new keys is not supported")
            node[path[-1]] = _postprocess_yml_value(value)
    opt["auto_resume"], opt["is_train"] = args.auto_resume, is_train
    if args.debug and not opt["name"].startswith("debug"):
        opt["name"] = "debug_" + opt["name"]
    if opt.get("num_gpu") == "auto":
        opt["num_gpu"] = torch.cuda.device_count()
    for phase, dataset in opt.get("datasets", {}).items():
        dataset["phase"] = phase.split("_")[0]
        if "scale" in opt:
            dataset["scale"] = opt["scale"]
        for k in ("dataroot_gt", "dataroot_lq"):
            if dataset.get(k) is not None:
                dataset[k] = osp.expanduser(dataset[k])
        if args.pairs:
            dataset["type"], dataset["pairs"] = "TensorPairs", args.pairs
        elif args.synthetic:
            dataset["type"], dataset["num_images"] = "Synthetic", args.synthetic
    for key, val in opt["path"].items():
        if val is not None and ("resume_state" in key or "pretrain_network" in key):
            opt["path"][key] = osp.expanduser(val)
    if is_train:
        exp = osp.join(opt["path"].get("experiments_root") or osp.join(root_path, "experiments"), opt["name"])
        opt["path"].update(experiments_root=exp, models=osp.join(exp, "models"), training_states=osp.join(exp, "training_states"), log=exp,
                           visualization=osp.join(exp, "visualization"))
        if "debug" in opt["name"]:
            if "val" in opt:
                opt["val"]["val_freq"] = 8
            opt["logger"]["print_freq"] = 1
            opt["logger"]["save_checkpoint_freq"] = 8
    else:
        res = osp.join(opt["path"].get("results_root") or osp.join(root_path, "results"), opt["name"])
        opt["path"].update(results_root=res, log=res, visualization=osp.join(res, "visualization"))
    return opt, args
