#!/usr/bin/env python3
"""bench.py -- images/s of the two-stage N = 8 Bayesian enhancement eval at 256x256 (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One step = one pass of the hot path over one batch of synthetic input PER RANK:
  8 images (256x256, LOL-like dark) x 8 Stage-I weight samples -> 64 conditions -> 64 Stage-II forwards
  -> GT-mean + PSNR per candidate -> per-image selection; for N > 1 the candidates of all ranks are
  all-gathered over RCCL (xGMI) so every rank holds every image's candidates (weak scaling).
Inputs are resident in HBM before the timed region; weights are seeded random init of the full architecture
(n_feat 40, blocks [2,2,2], shipped QD model4 decomposition weights).  f32 tensors throughout; the pointwise GEMMs evaluate
their f32 products as six bf16-limb products with f32 accumulation (pw_gemm_x6.hip: error below torch's own f32 GEMM).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "bayesian-enhancement-model_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md)
MFMA_F32_PEAK_TFLOPS = 157.3   # dense f32 MFMA peak (MI355X_MICROARCH.md)


def cpu_baseline(n_samples_timed=2, num_samples=8):
    """The CPU oracle ("port": this repo's restatement of the reference's PyTorch-CPU path, including the
    per-time-step Python loop of selective_scan_torch) on a bounded sample of the same workload."""
    from oracle import bem_oracle as O
    from bem.pipeline import build_nets, synthetic_pair
    net1, net2 = build_nets(device="cpu")
    sd1 = {k: v.detach() for k, v in net1.state_dict().items()}
    sd2 = {k: v.detach() for k, v in net2.state_dict().items()}
    lq, gt = synthetic_pair((1, 3, 256, 256))
    cores = torch.get_num_threads()
    t0 = time.perf_counter()
    O.eval_mc_ref(sd1, sd2, lq, gt, n_samples_timed, gt_mean=True, scan=O.selective_scan_ref,
                  generator=torch.Generator().manual_seed(0))
    dt = time.perf_counter() - t0
    per_image = dt * num_samples / n_samples_timed
    return {"value": 1.0 / per_image, "unit": "img/s", "cores": cores, "kind": "port",
            "sample": f"1 image x {n_samples_timed} of {num_samples} samples at 256x256 ({dt:.1f} s CPU), scaled to {num_samples} samples/image"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--images", type=int, default=8, help="images per rank per step")
    ap.add_argument("--samples", type=int, default=8, help="Bayesian samples per image")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-kernel", default="pw_x6_stream<2>", help="kernel whose launches are timed with HIP events for the roofline entry (bem.ops._KEYS)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from bem import native, ops
    native.lib()      # fail loudly without the HIP library
    from bem.dist import gather_candidates
    from bem.pipeline import BEMPipeline, build_nets, synthetic_pair

    net1, net2 = build_nets(device=dev)
    pipe = BEMPipeline(net1, net2, 16, 0.1)
    B, N, S = args.images, args.samples, args.size
    lq, gt = synthetic_pair((B, 3, S, S), seed=287128 + rank, device=dev)     # every rank its own images

    def step(i):
        r = pipe.enhance(lq, gt, N, gt_mean=True, seed=1000 + i, sync=False)   # selection on the device, no host sync per step
        if world > 1:
            return gather_candidates(r["final"], r["psnr"], world)
        return r["final"], r["psnr"]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    ops.profile_start(args.profile_kernel)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    barrier()
    dt = time.perf_counter() - t0
    prof = ops.profile_stop()
    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())

    if rank == 0:
        imgs = world * B * args.steps
        out = {
            "metric": f"images/sec (whole node) CG_UNet N={N} Bayesian eval @{S}x{S}",
            "value": imgs / dt, "unit": "img/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "arithmetic": "f32 storage and accumulation; pointwise GEMM products as exact 3-limb bf16 expansions (6 MFMA products, error <= torch f32 GEMM)",
            "config": {"workload": f"CG_UNet_LOLv1.yml + DecompDualBranch2DDWavelet_4.yml eval, batch={B} {S}x{S}, N={N} Bayesian samples + GT_mean per GPU",
                       "images_per_gpu": B, "samples_per_image": N, "parallelism": f"image-sharded x{world}, RCCL all-gather of candidates"},
        }
        # roofline of the dominant kernel, from HIP events recorded around its launches in the timed steps
        if prof and prof["launches"]:
            avg_ms = prof["ms"] / prof["launches"]
            if prof["bound"] == "mfma":
                ach = prof["flops"] / prof["launches"] / (avg_ms * 1e-3) / 1e12
                out["roofline"] = {"kernel": prof["kernel"], "bound": "mfma", "achieved": ach, "peak": MFMA_F32_PEAK_TFLOPS,
                                   "unit": "TFLOP/s", "frac": ach / MFMA_F32_PEAK_TFLOPS, "traffic": None,
                                   "launches": prof["launches"], "avg_launch_us": avg_ms * 1e3,
                                   "hbm_GBps_algorithmic": prof["bytes"] / prof["launches"] / (avg_ms * 1e-3) / 1e9}
            else:
                ach = prof["bytes"] / prof["launches"] / (avg_ms * 1e-3) / 1e9
                out["roofline"] = {"kernel": prof["kernel"], "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": ach / HBM_PEAK_GBS, "traffic": None, "launches": prof["launches"], "avg_launch_us": avg_ms * 1e3}
        # HBM traffic of that kernel from the PMC counters: rocprofv3 cannot be nested inside this process, so the per-launch
        # figure is taken from the committed counter passes of this same command (profiles/r01_traffic.json, see its _note)
        try:
            tr = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))["kernels"]
            key = out["roofline"]["kernel"].replace("_kernel<", "_kernel<")
            if key in tr and world == 1 and (B, N, S) == (8, 8, 256):
                out["roofline"]["traffic"] = tr[key]["hbm_bytes_per_launch"]
                out["roofline"]["algorithmic_bytes_per_launch"] = prof["bytes"] / prof["launches"]
        except (OSError, KeyError, ValueError):
            pass
        # whole-path figure on SURVEY.md section 8d's algorithmic bytes: N * (32360 * Hp*Wp/4 + 11.1e6) per image
        bytes_img = N * (32360.0 * S * S / 4 + 11.1e6)
        out["path_hbm_roofline_frac"] = (out["value"] / world) * bytes_img / (HBM_PEAK_GBS * 1e9)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(2, N)
            out["speedup_vs_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
