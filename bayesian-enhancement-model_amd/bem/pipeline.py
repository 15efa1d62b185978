"""The two-stage N-sample Bayesian enhancement loop of Enhancement/eval.py:146-297, kept on the device.

Differences from the reference driver that do not change results:
  * all (image, sample) pairs of a call go through Stage-I and Stage-II as ONE batch -- every pair has
    its own Bayesian weight sample (per-batch-element weights in the GEMM / depthwise kernels), which is
    what the reference's B=1 loop draws (eval.py:199-211);
  * Stage-I outputs never leave the GPU (the reference copies every sample D2H and back, eval.py:211,219);
  * decomp(image) is evaluated once per image and shared by its N samples (it does not depend on the sample) -- and, depending on nothing
    Stage I produces, on a second HIP stream beside it: Stage I is ~160 kernels on H/16 x W/16 planes that occupy a few CUs each, the
    decomposition's full-resolution kernels fill the rest of the chip meanwhile (BEM_DECOMP_OVERLAP=0: one stream).
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional

import torch

from . import ops
from .modules import SampleCtx, sampling


class BEMPipeline:
    """net1: Stage-I Bayesian ``Network`` (after convert2bnn), net2: Stage-II ``DecompDualBranchDDWavelet``."""

    def __init__(self, net1, net2, scale_down: int = 16, noise_level: float = 0.1):
        self.net1, self.net2 = net1, net2
        self.scale, self.noise_level = scale_down, noise_level
        self._side = None          # second stream for decomp(image), created on first use

    @torch.no_grad()
    def candidates(self, imgs, targets, num_samples: int, gt_mean: bool, deterministic: bool = False,
                   eps: Optional[Dict[str, torch.Tensor]] = None, noise: Optional[torch.Tensor] = None,
                   img_down: Optional[torch.Tensor] = None, seed: int = 0, rank: int = 0, sample_offset: int = 0,
                   total_samples: Optional[int] = None):
        """imgs (B,3,h,w) in [0,1] on the GPU, targets (B,3,h,w) or None.
        Returns dict(conds (B*N,3,hd,wd), raw (B*N,3,Hp,Wp), final (B*N,3,h,w), psnr (B*N)); row = image*N + sample.
        ``sample_offset`` / ``total_samples``: this call draws samples [offset, offset + num_samples) of ``total_samples`` per image
        (sample-major multi-GPU sharding, bem.dist): injected ``eps`` / ``noise`` are given for ALL (image, sample) rows and sliced here."""
        from basicsr.bayesian import set_prediction_type
        B, _, h, w = imgs.shape
        N = 1 if deterministic else num_samples
        NT = total_samples or N
        if not deterministic and (eps is not None or noise is not None) and NT != N:
            rows = (torch.arange(B)[:, None] * NT + sample_offset + torch.arange(N)[None, :]).reshape(-1).to(imgs.device)
            if eps is not None:
                eps = {k_: v.index_select(0, rows) for k_, v in eps.items()}
            if noise is not None:
                noise = noise.index_select(0, rows)
        f = 4 * self.scale
        Hp = (((h + f) // f) * f) if h % f else h                                # _padimg_np, eval.py:146-153
        Wp = (((w + f) // f) * f) if w % f else w
        pad = ops.pad_reflect(imgs.contiguous(), Hp, Wp)
        if img_down is None:
            img_down = ops.resize_down(pad, self.scale)
        hd, wd = img_down.shape[-2:]
        x1 = img_down[:, None].expand(B, N, 3, hd, wd).reshape(B * N, 3, hd, wd)  # (B*N,3,hd,wd): row = image*N + sample
        hoist = hasattr(self.net2, "forward_decomposed")
        d_img, main = None, torch.cuda.current_stream()
        if hoist and os.environ.get("BEM_DECOMP_OVERLAP", "1") != "0":
            if self._side is None or self._side.device != pad.device:
                self._side = torch.cuda.Stream(device=pad.device)
            self._side.wait_stream(main)                                         # pad is ready
            with torch.cuda.stream(self._side):
                d_img = self.net2.decompose(pad, 0)                              # once per image, beside Stage I
            pad.record_stream(self._side)
        set_prediction_type(self.net1, deterministic)
        with sampling(None if deterministic else SampleCtx(B * N, eps, seed, rank=rank)) as ctx:
            pred = self.net1(x1)[-1]
        tmean = ops.plane_mean(targets.contiguous()) if (gt_mean and targets is not None) else None
        if noise is None and self.noise_level:
            # torch.randn_like of eval.py:209: its own key space (bit 62 of the stream id), per rank and per forward
            noise = ops.randn(tuple(pred.shape), pred.device, seed, (1 << 62) | (rank << 44) | SampleCtx._epoch)
        conds = ops.cond_postproc(pred, tmean, noise if self.noise_level else None, N, self.noise_level)
        cond_up = ops.bilinear_up(conds, self.scale)                             # (B*N,3,Hp,Wp)
        if hoist:
            if d_img is None:
                d_img = self.net2.decompose(pad, 0)                              # once per image
            else:
                main.wait_stream(self._side)
                d_img.record_stream(main)                                        # allocated on the side stream, consumed (and freed) on this one
            d_cond = self.net2.decompose(cond_up, 0)
            raw = self.net2.forward_decomposed(d_img, d_cond, None if N == 1 else N)
        else:
            # Stage-II archs without a separable decomposition stage (DecompDualBranch, the *2 / *DD / SingleBranch siblings): the
            # reference's own call per candidate, net(cat(img, cond)) (eval.py:211-213), as one batch of B*N rows
            x2 = torch.empty(B * N, 6, Hp, Wp, device=pad.device, dtype=pad.dtype)
            ops.copy_channels(pad[:, None].expand(B, N, 3, Hp, Wp).reshape(B * N, 3, Hp, Wp).contiguous(), x2, 0)
            ops.copy_channels(cond_up, x2, 3)
            raw = self.net2(x2)[-1]
        final, psnr = ops.candidate_finalize(raw, None if targets is None else targets.contiguous(), N, h, w,
                                             bool(gt_mean and targets is not None))
        return dict(conds=conds, raw=raw, final=final, psnr=psnr, N=N)

    @staticmethod
    def select(psnr_rows: List[float]) -> int:
        """eval.py:284-285 with psnr_weight = 1: index of the first maximum of psnr / max(psnr)."""
        m = max(psnr_rows)
        if m == 0:                 # no usable score (no targets): the first candidate, like the reference's empty-score fallback
            return 0
        rel = [p / m for p in psnr_rows]
        return rel.index(max(rel))

    @torch.no_grad()
    def enhance(self, imgs, targets, num_samples, gt_mean=True, deterministic=False, sync=True, scorer=None, monte_carlo=False, shard=None, **kw):
        """candidates + per-image selection.  Default selection = the full-reference PSNR rule (first maximum of psnr / max(psnr),
        eval.py:284-285) on the device; ``scorer`` (bem.scorers) switches to the PSNR/SSIM-weighted rule or a no-reference
        scorer (eval.py:268-281).  ``monte_carlo``: also returns the Monte-Carlo mean prediction (eval.py:224-225,308-314) with its
        PSNR / SSIM.  With ``sync=False`` nothing is copied to the host."""
        if shard is not None and shard[1] > 1 and not deterministic:
            # sample-major multi-GPU form (bem.dist): this rank draws its block of the N samples of every image, one RCCL all-gather per
            # tensor brings candidates, scores (and raw outputs for the Monte-Carlo mean) back into the unsharded (image, sample) order
            from . import dist as bdist
            rank, world = shard
            kw.pop("rank", None)
            lo, hi = bdist.shard_samples(num_samples, rank, world)
            B = imgs.shape[0]
            r = self.candidates(imgs, targets, max(hi - lo, 1), gt_mean, False, rank=rank, sample_offset=lo, total_samples=num_samples, **kw)
            for k_ in ("final", "psnr", "conds") + (("raw",) if monte_carlo else ()):
                t = r[k_] if hi > lo else r[k_][:0]
                r[k_] = bdist.gather_samples(t.contiguous(), B, num_samples, rank, world)
            if not monte_carlo:
                r.pop("raw")
            r["N"] = num_samples
        else:
            r = self.candidates(imgs, targets, num_samples, gt_mean, deterministic, **kw)
        N = r["N"]
        h, w = imgs.shape[-2:]
        if scorer is not None:
            sel = scorer.select(r["final"], targets, N, psnr=r["psnr"])
            best, img = sel["best"], sel["best_images"]
            bp = ops.select_scores(None, r["psnr"], N, rule="max")[1] if targets is None else _gather(r["psnr"], best, N)
            r.update(scores=sel["s1"], scores2=sel["s2"])
        else:
            best, bp, img = ops.select_best(r["final"], r["psnr"], N)
            if targets is None:
                best = torch.zeros_like(best)          # no reference: the reference keeps the first sample (eval.py:291-293)
                img = r["final"][::N].contiguous()
        r.update(best=best, best_images=img, best_psnr=bp)
        if monte_carlo:
            mc = ops.mc_mean(r["raw"], None if targets is None else targets.contiguous(), N, h, w, bool(gt_mean and targets is not None))
            r["mc"] = mc
            if targets is not None:
                _, r["mc_psnr"] = ops.candidate_finalize(mc, targets.contiguous(), 1, h, w, False)
                r["mc_ssim"] = ops.ssim(mc, targets.contiguous(), 1) if min(h, w) > 10 else None
        if sync:
            r["best"], r["best_psnr"] = best.cpu().tolist(), bp.cpu().tolist()
        return r


def _gather(v, best, N):
    """v (B*N), best (B) -> v[b*N + best[b]] on the device (index arithmetic only)."""
    B = best.numel()
    return v.view(B, N).gather(1, best.long().view(B, 1)).view(B)


# ------------------------------------------------------------------------------------------------
# small shared helpers (bench.py, smoke(), tests)
# ------------------------------------------------------------------------------------------------
def synthetic_pair(shape, seed=287128, device="cpu"):
    """LOL-like dark input and bright target: lq = 0.25 U[0,1), gt = clamp(3.5 lq + 0.05 N)."""
    g = torch.Generator().manual_seed(seed)
    lq = 0.25 * torch.rand(shape, generator=g)
    gt = (3.5 * lq + 0.05 * torch.randn(shape, generator=g)).clamp(0, 1)
    return lq.to(device), gt.to(device)


def build_nets(n_feat=40, num_blocks=(2, 2, 2), seed=100, device="cuda", stage2="DecompDualBranchDDWavelet", decomp="model4"):
    """Seeded random-init Stage-I (Bayesian) and Stage-II nets through the registry, like eval.py:84-85."""
    from basicsr.archs import build_network
    from basicsr.bayesian import convert2bnn_selective
    torch.manual_seed(seed)
    common = dict(n_feat=n_feat, d_state=[1, 1, 1], ssm_ratio=1, mlp_ratio=4, mlp_type="gdmlp", use_pixelshuffle=True,
                  drop_path=0.0, sam=False, stage=1, num_blocks=list(num_blocks))
    net1 = build_network(dict(type="Network", in_channels=3, out_channels=3, **common))
    convert2bnn_selective(net1, {"sigma_init": 0.05, "decay": 0.998, "pretrain": False})
    net2 = build_network(dict(type=stage2, in_channels=6, out_channels=3, decomp_model=decomp, **common))
    return net1.to(device).eval(), net2.to(device).eval()
