#!/usr/bin/env python3
"""Training driver with the loop shape of basicsr/train.py:97-262:

    parse_options -> load_resume_state (:74-94) -> make dirs -> dataloaders -> build_model -> [resume_training] ->
    for epoch: for batch:  current_iter += 1 ; update_learning_rate ; feed_train_data ; optimize_parameters ;
                           log every print_freq ; save every save_checkpoint_freq ; validate every val_freq (+ save_best) ->
    save 'latest'.

  python basicsr/train.py --opt Options/DecompDualBranch2DDWavelet_4.yml --synthetic 64 [--auto_resume] [--debug]
      [--force_yml train:total_iter=200 logger:save_checkpoint_freq=50] [--launcher pytorch]

What is the hot path runs on the HIP kernels (``optimize_parameters`` of the two model classes: forward, backward, clip, AdamW with no
host synchronisation inside the step; validation through the inference kernels).  What is control plane stays small and host-side:
the message logger is ``print`` (TensorBoard / wandb are not dependencies), the datasets are the tensor shim of ``basicsr.data`` (the
file-backed datasets are host I/O outside the path: pass ``--synthetic N`` or ``--pairs file.pt``), there are no prefetcher threads
(the batches already live on the device).  ``--launcher pytorch`` (under ``python -m torch.distributed.run``) trains data-parallel:
one process per GPU, the flat gradient buffer averaged by one RCCL all-reduce per step (bem.train.BemAdamW.all_reduce_grads)."""
import datetime
import math
import os
import os.path as osp
import shutil
import sys
import time

_PKG = osp.abspath(osp.join(osp.dirname(osp.abspath(__file__)), osp.pardir))
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)

import torch  # noqa: E402

from basicsr.data import build_dataloader, build_dataset  # noqa: E402
from basicsr.models import build_model  # noqa: E402
from basicsr.utils.misc import check_resume, load_resume_state  # noqa: E402,F401
from basicsr.utils.options import parse_options  # noqa: E402


def create_train_val_dataloader(opt, device):
    """train.py:34-71: the train loader, the val loaders, total epochs / iterations."""
    train_loader, val_loaders, total_epochs, total_iters = None, [], 0, 0
    for phase, dataset_opt in opt["datasets"].items():
        dataset_opt["model_type"] = opt["model_type"]
        if phase == "train":
            ratio = dataset_opt.get("dataset_enlarge_ratio", 1)
            if opt["model_type"] == "ConditionGenerator":
                dataset_opt.setdefault("mask_ratio", 0.4)          # Dataset_PairedImage_Mask's MIM mask (paired_image_dataset.py:374-383)
            train_set = build_dataset(dataset_opt)
            train_loader = build_dataloader(train_set, dataset_opt, num_gpu=opt["num_gpu"], dist=opt["dist"], seed=opt["manual_seed"],
                                            device=device, rank=opt["rank"], world=opt["world_size"], train=True)
            per_epoch = math.ceil(len(train_set) * ratio / (dataset_opt["batch_size_per_gpu"] * opt["world_size"]))
            total_iters = int(opt["train"]["total_iter"])
            total_epochs = math.ceil(total_iters / max(per_epoch, 1))
            print(f"Training statistics:\n\tNumber of train images: {len(train_set)}\n\tDataset enlarge ratio: {ratio}"
                  f"\n\tBatch size per gpu: {dataset_opt['batch_size_per_gpu']}\n\tWorld size (gpu number): {opt['world_size']}"
                  f"\n\tRequire iter number per epoch: {per_epoch}\n\tTotal epochs: {total_epochs}; iters: {total_iters}.", flush=True)
        elif phase.split("_")[0] == "val":
            val_set = build_dataset(dataset_opt)
            val_loaders.append(build_dataloader(val_set, dataset_opt, seed=opt["manual_seed"], device=device, train=False))
        else:
            raise ValueError(f"Dataset phase {phase} is not recognized.")
    return train_loader, val_loaders, total_epochs, total_iters


def train_pipeline(root_path, argv=None):
    opt, args = parse_options(root_path, is_train=True, argv=argv)
    opt["root_path"] = root_path
    device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")

    resume_state = load_resume_state(opt, experiments_root=osp.dirname(opt["path"]["experiments_root"]))
    if resume_state is None and opt["rank"] == 0:
        # make_exp_dirs (utils/misc.py:54-71): an existing experiment directory is set aside, not overwritten
        exp = opt["path"]["experiments_root"]
        if osp.exists(exp):
            shutil.move(exp, exp + "_archived_" + time.strftime("%Y%m%d_%H%M%S"))
        for k in ("experiments_root", "models", "training_states", "visualization"):
            os.makedirs(opt["path"][k], exist_ok=True)
    if opt["rank"] == 0:
        os.makedirs(opt["path"]["experiments_root"], exist_ok=True)
        shutil.copy(args.opt, opt["path"]["experiments_root"])          # copy_opt_file

    train_loader, val_loaders, total_epochs, total_iters = create_train_val_dataloader(opt, device)
    model = build_model(opt)
    if resume_state:
        model.resume_training(resume_state)
        print(f"Resuming training from epoch: {resume_state['epoch']}, iter: {resume_state['iter']}.", flush=True)
        start_epoch, current_iter, best_metric = resume_state["epoch"], resume_state["iter"], resume_state["best_metric"]
    else:
        start_epoch, current_iter, best_metric = 0, 0, {"iter": 0}
        if opt.get("val") is not None:
            for k in (opt["val"].get("metrics") or {"psnr": None}):
                best_metric[k] = 0

    print(f"Start training from epoch: {start_epoch}, iter: {current_iter}", flush=True)
    start_time, t_iter = time.time(), time.time()
    log = opt["logger"]
    epoch = start_epoch
    for epoch in range(start_epoch, total_epochs + 1):
        train_loader.set_epoch(epoch)
        batches = iter(train_loader)
        if resume_state and epoch == start_epoch:
            # a resumed epoch continues behind the batches the saved run had consumed (the loader's order is a function of seed + epoch)
            for _ in range(current_iter - epoch * len(train_loader)):
                next(batches, None)
        for train_data in batches:
            current_iter += 1
            if current_iter > total_iters:
                break
            model.update_learning_rate(current_iter, warmup_iter=opt["train"].get("warmup_iter", -1))
            model.current_iter_hint = current_iter
            model.feed_train_data(train_data)
            model.optimize_parameters(current_iter)
            if current_iter % log["print_freq"] == 0 and opt["rank"] == 0:
                dt, t_iter = (time.time() - t_iter) / log["print_freq"], time.time()
                msg = f"[epoch:{epoch:3d}, iter:{current_iter:8,d}, lr:({', '.join(f'{v:.3e}' for v in model.get_current_learning_rate())})] [time (iter): {dt:.3f}] "
                msg += " ".join(f"{k}: {v:.4e}" for k, v in model.get_current_log().items())
                print(msg, flush=True)
            if current_iter % int(log["save_checkpoint_freq"]) == 0:
                print("Saving models and training states.", flush=True)
                model.save(epoch, current_iter, best_metric=best_metric)
            if opt.get("val") is not None and val_loaders and opt["val"].get("val_freq") and current_iter % int(opt["val"]["val_freq"]) == 0:
                cur = sum(model.validation(vl, current_iter, None, opt["val"].get("save_img", False), opt["val"].get("rgb2bgr", True),
                                           opt["val"].get("use_image", True)) for vl in val_loaders) / len(val_loaders)
                if best_metric.get("psnr", 0) < cur:
                    best_metric["psnr"], best_metric["iter"] = cur, current_iter
                    model.save_best(best_metric)
        if current_iter > total_iters:
            break
    print(f"End of training. Time consumed: {datetime.timedelta(seconds=int(time.time() - start_time))}", flush=True)
    print("Save the latest model.", flush=True)
    model.save(epoch=-1, current_iter=-1, best_metric=best_metric)
    if opt.get("val") is not None:
        for vl in val_loaders:
            model.validation(vl, current_iter, None, opt["val"].get("save_img", False), opt["val"].get("rgb2bgr", True), opt["val"].get("use_image", True))
    return model, dict(iter=min(current_iter, total_iters), best_metric=best_metric)


if __name__ == "__main__":
    train_pipeline(_PKG)
