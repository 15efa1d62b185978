#!/usr/bin/env python3
"""Two-stage N-sample Bayesian enhancement of a folder of images -- the driver of Enhancement/eval.py with the reference's command
line (eval.py:30-60) on the device-resident pipeline (bem.pipeline.BEMPipeline):

  python Enhancement/eval.py --opt Options/CG_UNet_LOLv1.yml --cond_opt Options/DecompDualBranch2DDWavelet_4.yml \
      --weights cg.pth --cond_weights stage2.pth --input_dir data/LOLv1/Test/input --target_dir data/LOLv1/Test/target \
      --dataset LOLv1 --GT_mean --num_samples 16 [--no_ref clip] [--psnr_weight 0.5] [--Monte_Carlo] [--deterministic]

What differs from the reference script, none of it in the results: all N samples of an image go through Stage I and Stage II as one
batch (``--parallel_num`` is accepted and ignored), conditions never leave the GPU, selection metrics run on the device, and images
are read / written with PIL (cv2, skimage, natsort, lpips, torchmetrics are not dependencies).  ``--no_ref clip`` uses the scorer
returned by ``make_clip_scorer`` -- the deterministic stand-in of bem.scorers unless a CLIP-IQA module is importable; ``--no_ref
niqe | uiqm_uciqe`` and ``--lpips`` need host-side metric packages that are outside this path and raise a clear error.
Output: ``<result_dir>/<dataset>/<image>.png`` (the selected candidate) and ``result.txt`` with the reference's summary lines."""
import argparse
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
if PKG not in sys.path:
    sys.path.insert(0, PKG)


def get_parser():
    p = argparse.ArgumentParser(description="Image Enhancement")
    p.add_argument("--result_dir", default="./results/", type=str, help="Directory for results")
    p.add_argument("--input_dir", default="", type=str, help="Directory for inputs")
    p.add_argument("--target_dir", default="", type=str, help="Directory for targets")
    p.add_argument("--opt", type=str, default="youryaml.yaml", help="Path to option YAML file.")
    p.add_argument("--cond_opt", type=str, default="your condition_yaml.yaml", help="Path to option YAML file.")
    p.add_argument("--weights", default="yourweight.pth", type=str, help="Path to weights")
    p.add_argument("--cond_weights", default="yourweight.pth", type=str, help="Path to weights")
    p.add_argument("--dataset", default="yourdataset", type=str, help="Name of dataset")
    p.add_argument("--GT_mean", action="store_true", help="Use the mean of GT to rectify the output of the model")
    p.add_argument("--num_samples", default=200, type=int, help="Number of random samples")
    p.add_argument("--Monte_Carlo", action="store_true", help="also report the average of the random samples")
    p.add_argument("--psnr_weight", default=1.0, type=float, help="Balance between PSNR and SSIM")
    p.add_argument("--no_ref", default="", type=str, choices=["", "clip", "niqe", "uiqm_uciqe"], help="no reference image quality evaluator")
    p.add_argument("--uiqm_weight", default=1.0, type=float, help="Balance between UIQM and UICIQE")
    p.add_argument("--lpips", action="store_true", help="True to compute LPIPS")
    p.add_argument("--deterministic", action="store_true", help="Use deterministic mode")
    p.add_argument("--parallel_num", default=1, type=int, help="accepted for compatibility: all samples of an image run as one batch")
    p.add_argument("--seed", default=287128, type=int, help="fix random seed to reproduce consistent results")
    p.add_argument("--clip_prompts", nargs="+", default=["brightness", "noisiness", "quality"])
    return p


def make_clip_scorer(prompts):
    from bem.scorers import ClipStandIn
    return ClipStandIn(prompts)


def load_rgb(path):
    from PIL import Image
    return np.asarray(Image.open(path).convert("RGB"), dtype=np.float32) / 255.0          # utils.load_img + /255 (eval.py:166-170)


def save_rgb(path, img01_hw3):
    from PIL import Image
    Image.fromarray(np.rint(np.clip(img01_hw3, 0, 1) * 255.0).astype(np.uint8)).save(path)   # img_as_ubyte


def load_params(net, path):
    ck = torch.load(path, map_location="cpu", weights_only=True)
    sd = ck["params"] if "params" in ck else ck
    net.load_state_dict({(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()})      # eval.py:98-100 (strict)


def main(argv=None):
    args = get_parser().parse_args(argv)
    if args.no_ref in ("niqe", "uiqm_uciqe") or args.lpips:
        raise SystemExit("--no_ref niqe / uiqm_uciqe and --lpips are host-side metric packages outside the HIP path (SURVEY.md section 2 rows 16, 1)")
    from basicsr.bayesian import set_prediction_type
    from basicsr.models import build_model
    from basicsr.utils.options import parse
    from bem.pipeline import BEMPipeline
    from bem.scorers import FullReference
    # one process per GPU under `python -m torch.distributed.run --nproc-per-node N Enhancement/eval.py ...`: the driver feeds ONE image at a
    # time (eval.py:160-222), so the N Bayesian samples of that image are what gets sharded (sample-major, bem.dist); every rank holds the
    # gathered candidates, rank 0 writes the files
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if not dist.is_initialized():
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", torch.cuda.current_device()))
    torch.manual_seed(args.seed)
    np.random.seed(args.seed)
    opt, cond_opt = parse(args.opt, is_train=False), parse(args.cond_opt, is_train=False)
    net = build_model(opt).net_g
    set_prediction_type(net, deterministic=args.deterministic)
    cond_net = build_model(cond_opt).net_g
    load_params(net, args.weights)
    load_params(cond_net, args.cond_weights)
    net, cond_net = net.cuda().eval(), cond_net.cuda().eval()
    # scale from the Stage-I option file, noise level from the Stage-II one, default 0 (eval.py:174-176,207)
    pipe = BEMPipeline(net, cond_net, opt.get("condition", {}).get("scale_down", 16), cond_opt.get("condition", {}).get("noise_level", 0))
    result_dir = os.path.join(args.result_dir, args.dataset)
    os.makedirs(result_dir, exist_ok=True)
    names = sorted(f for f in os.listdir(args.input_dir) if f.lower().endswith((".png", ".jpg", ".jpeg", ".bmp")))
    scorer = make_clip_scorer(args.clip_prompts) if args.no_ref == "clip" else (FullReference(args.psnr_weight) if args.target_dir else None)
    psnr, ssim, mc_psnr, mc_ssim = [], [], [], []
    t0 = time.perf_counter()
    with torch.inference_mode():
        for i, name in enumerate(names):
            img = torch.from_numpy(load_rgb(os.path.join(args.input_dir, name))).permute(2, 0, 1)[None].cuda()
            tgt = None
            if args.target_dir:
                tgt = torch.from_numpy(load_rgb(os.path.join(args.target_dir, name))).permute(2, 0, 1)[None].cuda()
            r = pipe.enhance(img, tgt, args.num_samples, gt_mean=args.GT_mean, deterministic=args.deterministic, scorer=scorer,
                             monte_carlo=args.Monte_Carlo, seed=args.seed + i, shard=(rank, world) if world > 1 else None)
            best = r["best_images"]
            if tgt is not None:
                from bem import ops
                _, p = ops.candidate_finalize(best.contiguous(), tgt.contiguous(), 1, best.shape[2], best.shape[3], False)
                psnr.append(float(p[0]))
                ssim.append(float(ops.ssim(best.contiguous(), tgt.contiguous(), 1)[0]))
                if args.Monte_Carlo:
                    mc_psnr.append(float(r["mc_psnr"][0])); mc_ssim.append(float(r["mc_ssim"][0]))
            if rank == 0:
                save_rgb(os.path.join(result_dir, os.path.splitext(name)[0] + ".png"), best[0].permute(1, 2, 0).cpu().numpy())
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return dict(psnr=psnr, ssim=ssim, mc_psnr=mc_psnr, mc_ssim=mc_ssim, result_dir=result_dir)
    print(f"running time: {time.perf_counter() - t0:.4f} sec")
    with open(os.path.join(result_dir, "result.txt"), "w") as f:
        if args.target_dir:
            for label, vals, unit in (("Best_PSNR", psnr, " dB"), ("Best_SSIM", ssim, "")):
                line = f"{label}: {np.mean(vals):.4f}{unit}"
                print(line); f.write(line + " \n")
            if args.Monte_Carlo:
                for label, vals, unit in (("MC_PSNR", mc_psnr, " dB"), ("MC_SSIM", mc_ssim, "")):
                    line = f"{label}: {np.mean(vals):.4f}{unit}"
                    print(line); f.write(line + " \n")
    return dict(psnr=psnr, ssim=ssim, mc_psnr=mc_psnr, mc_ssim=mc_ssim, result_dir=result_dir)


if __name__ == "__main__":
    main()
