"""Candidate scoring and selection of Enhancement/eval.py:229-297 as a pluggable interface.

A scorer maps the finished candidates of a batch -- ``final`` (B*N,3,h,w) in [0,1], row = image*N + sample, and the targets
(B,3,h,w) or None -- to one or two score vectors plus the selection rule the reference applies to them:

  FullReference(psnr_weight)   PSNR (Enhancement/utils.py:5-9) and, for psnr_weight < 1, SSIM (utils.py:12-57), rule
                               ``index(max(w * psnr / max(psnr) + (1 - w) * ssim / max(ssim)))``            (eval.py:284-285)
  NoReference(fn, 'max'|'min') any image-quality function returning one score per candidate; ``index(max)`` for CLIP-IQA
                               (eval.py:271), ``index(min)`` for NIQE (eval.py:273-274)
  ClipStandIn()                a deterministic stand-in for torchmetrics' CLIPImageQualityAssessment, whose weights cannot be
                               fetched here.  It follows eval.py:229-243's post-processing (prompt scores
                               'brightness' x 0.7, 'noisiness' x 1, 'quality', averaged) on closed-form per-image statistics;
                               its VALUES are parity-unpinned, the ordering logic applied to them is the reference's.
All scores stay on the device; ``select`` runs bem_select_scores_f32 (first index on ties, float64 comparisons)."""
from __future__ import annotations

from typing import Callable, Optional

import torch

from . import ops


class Scorer:
    rule = "weighted"      # 'weighted' | 'max' | 'min'
    weight = 1.0

    def scores(self, final, targets, samples_per_image, psnr=None):
        """-> (s1 (Bn), s2 (Bn) | None)"""
        raise NotImplementedError

    def select(self, final, targets, samples_per_image, psnr=None):
        """-> dict(best (B) int32, best_images (B,3,h,w), s1, s2, best_s1, best_s2)"""
        s1, s2 = self.scores(final, targets, samples_per_image, psnr)
        best, b1, b2, img = ops.select_scores(final, s1.contiguous(), samples_per_image, None if s2 is None else s2.contiguous(), self.weight, self.rule)
        return dict(best=best, best_images=img, s1=s1, s2=s2, best_s1=b1, best_s2=b2)


class FullReference(Scorer):
    def __init__(self, psnr_weight: float = 1.0):
        self.weight = float(psnr_weight)

    def scores(self, final, targets, samples_per_image, psnr=None):
        if targets is None:
            raise ValueError("FullReference scorer needs targets")
        if psnr is None:
            raise ValueError("FullReference scorer: pass the PSNR vector computed by candidate_finalize")
        # the reference always evaluates both metrics (eval.py:263-264); SSIM is skipped here only when it cannot change the choice
        s2 = ops.ssim(final, targets.contiguous(), samples_per_image) if self.weight != 1.0 else None
        return psnr, s2


class NoReference(Scorer):
    def __init__(self, fn: Callable[[torch.Tensor], torch.Tensor], rule: str = "max"):
        if rule not in ("max", "min"):
            raise ValueError("NoReference: rule must be 'max' or 'min'")
        self.fn, self.rule = fn, rule

    def scores(self, final, targets, samples_per_image, psnr=None):
        s = self.fn(final)
        if s.shape != (final.shape[0],):
            raise ValueError("NoReference scorer function must return one score per candidate")
        return s.float().contiguous(), None


class ClipStandIn(NoReference):
    """eval.py:229-243 with closed-form 'prompt' scores in [0,1] instead of CLIP similarities (deterministic, weights-free)."""

    def __init__(self, prompts=("brightness", "noisiness", "quality")):
        self.prompts = tuple(prompts)
        super().__init__(self._score, "max")

    def _score(self, final):
        Bn = final.shape[0]
        m = ops.plane_mean(final.contiguous())                              # (Bn,3) channel means, HIP reduction
        lum = 0.299 * m[:, 0] + 0.587 * m[:, 1] + 0.114 * m[:, 2]
        sq = ops.plane_mean((final * final).contiguous())
        var = (sq - m * m).clamp_min(0).mean(dim=1)
        vals = {"brightness": lum.clamp(0, 1) * 0.7,                         # eval.py:236-238: brightness scaled down by 0.7
                "noisiness": (1.0 - 4.0 * var).clamp(0, 1) * 1.0,
                "quality": (1.0 - (lum - 0.45).abs() * 2.0).clamp(0, 1)}
        unknown = [p for p in self.prompts if p not in vals]
        if unknown:
            raise ValueError(f"ClipStandIn: unknown prompts {unknown}")
        return torch.stack([vals[p] for p in self.prompts]).mean(dim=0).reshape(Bn)
