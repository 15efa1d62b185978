"""Multi-GPU layer: one process per GPU, (image, sample) pairs sharded image-major across ranks, ONE exchange step -- an RCCL
all-gather (torch.distributed backend "nccl" is RCCL on ROCm) over xGMI -- then the per-image selection of eval.py:268-297 on the
gathered set.  No collective touches the data path before that: Stage-I / Stage-II are independent per (image, sample) pair and
the weights are replicated (2.8 M + 2.7 M parameters).

Two forms of the exchange (BASELINE north star asks for the first; SURVEY 5 / 8e prices both):
  candidates : every rank gathers every candidate (B*N/W, 3, h, w) and its score, selects on the full set
               (config 5: 369 MB per rank, ~17 ms on a ring, per-link bound);
  scores     : ranks select locally (image-major sharding keeps an image's N candidates on one rank), gather the winners and
               the score table only (N x less payload).
Both return identical (best index, best image) for every image -- tests/test_dist_cpu.py.

Sample-major sharding (``shard_samples`` / ``gather_samples`` / ``enhance_sample_sharded``) is the split for B < world -- the way the
eval driver is actually called (Enhancement/eval.py:160-222 feeds ONE image at a time): the N samples of every image are block-partitioned
over the ranks, each rank recomputes decomp(image) for its samples, and the same all-gather brings the candidates back into the
unsharded (image, sample) order, so the first-maximum selection sees exactly the list the single-GPU run sees.
The reference has no collective on this path: nothing to mirror, only results to match."""
from __future__ import annotations

from typing import Callable, List, Optional

import torch
import torch.distributed as dist


def shard_images(n_images: int, rank: int, world: int):
    """Image-major block partition: keeps all N candidates of an image on one rank (selection is local,
    decomp(image) is reused by its N samples).  Returns (start, stop) of this rank's images."""
    per, rem = divmod(n_images, world)
    start = rank * per + min(rank, rem)
    return start, start + per + (1 if rank < rem else 0)


def _all_gather_rows(t: torch.Tensor, world: int) -> torch.Tensor:
    """(n, ...) on every rank (equal n) -> (world * n, ...) rank-major.  nccl: one all_gather_into_tensor; gloo (CPU tests): list form."""
    t = t.contiguous()
    out = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), device=t.device, dtype=t.dtype)
    if dist.get_backend() == "gloo":
        dist.all_gather(list(out.chunk(world)), t)
    else:
        dist.all_gather_into_tensor(out, t)
    return out


def gather_candidates(final: torch.Tensor, score: torch.Tensor, world: int):
    """final (Bn,3,h,w), score (Bn) on every rank (equal shapes) -> (world*Bn,3,h,w), (world*Bn) on every rank, rank-major."""
    if world == 1:
        return final, score
    return _all_gather_rows(final, world), _all_gather_rows(score, world)


def gather_ragged(final: torch.Tensor, score: torch.Tensor, counts, world: int):
    """Uneven shards (n_images % world != 0): pad to the largest shard, gather, drop the padding."""
    if world == 1:
        return final, score
    m = max(counts)
    pad = m - final.shape[0]
    if pad:
        final = torch.cat([final, final.new_zeros((pad,) + tuple(final.shape[1:]))])
        score = torch.cat([score, score.new_full((pad,), float("-inf"))])
    f, s = gather_candidates(final, score, world)
    keep = torch.cat([torch.arange(r * m, r * m + c) for r, c in enumerate(counts)]).to(f.device)
    return f.index_select(0, keep), s.index_select(0, keep)


def first_best(score_rows: List[List[float]], rule: str = "max") -> List[int]:
    """Python ``list.index(max(...))`` / ``index(min(...))`` per image (eval.py:268-285): FIRST occurrence on ties."""
    pick = max if rule == "max" else min
    return [row.index(pick(row)) for row in score_rows]


def select_best(score: torch.Tensor, samples_per_image: int):
    """eval.py:284-285 with psnr_weight = 1: index of the first maximum of psnr / max(psnr) per image."""
    best = []
    for row in score.view(-1, samples_per_image).cpu().tolist():
        m = max(row)
        if m == 0:               # no usable score (e.g. all-zero PSNR without targets): the reference's first candidate
            best.append(0)
            continue
        rel = [v / m for v in row]
        best.append(rel.index(max(rel)))
    return best


def _select_device(final, score, N):
    """Selection without leaving the device: the HIP kernel on GPU tensors, a torch restatement of the same first-maximum rule on
    CPU tensors (gloo tests).  Returns (best (B) int64, best images (B,3,h,w))."""
    B = final.shape[0] // N
    if final.is_cuda:
        from . import ops
        best, _, img = ops.select_best(final.contiguous(), score.contiguous(), N)
        return best.long(), img
    s = score.view(B, N)
    m = s.max(dim=1, keepdim=True).values
    rel = torch.where(m != 0, s / torch.where(m != 0, m, torch.ones_like(m)), torch.zeros_like(s))
    top = rel.max(dim=1, keepdim=True).values
    best = (rel == top).float().argmax(dim=1)                      # first index of the maximum
    img = final.view(B, N, *final.shape[1:])[torch.arange(B), best]
    return best, img


def exchange_and_select(final: torch.Tensor, score: torch.Tensor, samples_per_image: int, world: int, mode: str = "candidates",
                        counts: Optional[List[int]] = None):
    """The exchange step + selection.  final (Bn_local,3,h,w), score (Bn_local), rows = image*N + sample (image-major shard).
    Returns (best images (B_total,3,h,w), best index (B_total)) on every rank, images in global order.
    counts: images per rank when the shards are ragged (None = equal)."""
    N = samples_per_image
    if world == 1:
        best, img = _select_device(final, score, N)
        return img, best
    if mode == "candidates":
        if counts is None:
            f, s = gather_candidates(final, score, world)
        else:
            f, s = gather_ragged(final, score, [c * N for c in counts], world)
        best, img = _select_device(f, s, N)
        return img, best
    if mode != "scores":
        raise ValueError(f"exchange_and_select: unknown mode {mode}")
    best, img = _select_device(final, score, N)
    if counts is None:
        return _all_gather_rows(img, world), _all_gather_rows(best, world)
    m = max(counts)
    pad = m - img.shape[0]
    if pad:
        img = torch.cat([img, img.new_zeros((pad,) + tuple(img.shape[1:]))])
        best = torch.cat([best, best.new_zeros(pad)])
    gi, gb = _all_gather_rows(img, world), _all_gather_rows(best, world)
    keep = torch.cat([torch.arange(r * m, r * m + c) for r, c in enumerate(counts)]).to(gi.device)
    return gi.index_select(0, keep), gb.index_select(0, keep)


def enhance_sharded(enhance: Callable, imgs: torch.Tensor, targets: Optional[torch.Tensor], num_samples: int, rank: int, world: int,
                    mode: str = "candidates", **kw):
    """The whole multi-GPU eval step for a GLOBAL batch present on every rank: image-major shard -> local ``enhance`` (any callable
    with BEMPipeline.enhance's contract: returns dict(final (b*N,3,h,w), psnr (b*N), N)) -> exchange -> selection.
    Returns (best images (B,3,h,w), best index (B)) -- equal on every rank and equal to the unsharded result."""
    B = imgs.shape[0]
    lo, hi = shard_images(B, rank, world)
    counts = [shard_images(B, r, world)[1] - shard_images(B, r, world)[0] for r in range(world)]
    if hi > lo:
        r = enhance(imgs[lo:hi], None if targets is None else targets[lo:hi], num_samples, **kw)
        final, score, N = r["final"], r["psnr"], r["N"]
    else:                                             # more ranks than images: an empty shard still joins the collective
        N = kw.get("N_hint", num_samples)
        final = imgs.new_zeros((0, 3) + tuple(imgs.shape[2:]))
        score = imgs.new_zeros((0,))
    ragged = len(set(counts)) > 1
    return exchange_and_select(final, score, N, world, mode, counts if ragged else None)


# ------------------------------------------------------------------------------------------------------------------
# sample-major sharding: the N samples of each image are split over the ranks (B < world, e.g. the eval driver's B = 1)
# ------------------------------------------------------------------------------------------------------------------
def shard_samples(num_samples: int, rank: int, world: int):
    """Block partition of the sample indices 0..N-1: (start, stop) of this rank.  Blocks are ordered by rank, so "first maximum over
    the gathered list" is the first maximum of the unsharded sample order."""
    return shard_images(num_samples, rank, world)


def gather_samples(t: torch.Tensor, n_images: int, num_samples: int, rank: int, world: int) -> torch.Tensor:
    """t (B * n_local, ...) with rows = image * n_local + local sample (this rank's block of samples of EVERY image)
    -> (B * N, ...) with rows = image * N + sample on every rank.  One all-gather of the (padded) blocks, then an index permutation."""
    if world == 1:
        return t
    counts = [shard_samples(num_samples, r, world)[1] - shard_samples(num_samples, r, world)[0] for r in range(world)]
    m, n_loc = max(counts), counts[rank]
    B = n_images
    blk = t.reshape((B, n_loc) + tuple(t.shape[1:]))
    if n_loc < m:
        blk = torch.cat([blk, blk.new_zeros((B, m - n_loc) + tuple(t.shape[1:]))], 1)
    g = _all_gather_rows(blk.reshape((B * m,) + tuple(t.shape[1:])), world)                 # rows = (rank, image, slot)
    idx = [(r * B + b) * m + sl for b in range(B) for r in range(world) for sl in range(counts[r])]
    return g.index_select(0, torch.tensor(idx, device=g.device))


def enhance_sample_sharded(candidates: Callable, imgs: torch.Tensor, targets: Optional[torch.Tensor], num_samples: int, rank: int, world: int,
                           mode: str = "candidates", **kw):
    """The multi-GPU eval step for FEW images (B < world): every rank draws its block of the N samples of every image.
    ``candidates(imgs, targets, n_local, sample_offset=lo, total_samples=N, **kw)`` has BEMPipeline.candidates' contract and returns
    dict(final (B*n_local,3,h,w), psnr (B*n_local)).  Returns (best images (B,3,h,w), best index (B)), identical on every rank and
    identical to the unsharded selection (first maximum, eval.py:284-285).
    mode 'candidates': all-gather of every candidate + score; 'scores': all-gather of the score table only, the winners are
    contributed by their owners through one all-reduce of (B,3,h,w) (every other rank adds exact zeros)."""
    N, B = num_samples, imgs.shape[0]
    lo, hi = shard_samples(N, rank, world)
    if hi > lo:
        r = candidates(imgs, targets, hi - lo, sample_offset=lo, total_samples=N, **kw)
        final, score = r["final"], r["psnr"]
    else:                                                 # more ranks than samples: an empty block still joins the collectives
        h, w = kw.get("hw_hint", tuple(imgs.shape[2:]))
        final, score = imgs.new_zeros((0, 3, h, w)), imgs.new_zeros((0,))
    if world == 1:
        best, img = _select_device(final, score, N)
        return img, best
    gs = gather_samples(score, B, N, rank, world)
    if mode == "candidates":
        best, img = _select_device(gather_samples(final, B, N, rank, world), gs, N)
        return img, best
    if mode != "scores":
        raise ValueError(f"enhance_sample_sharded: unknown mode {mode}")
    best, _ = _select_device(gs.new_zeros((B * N, 1)), gs, N)             # selection needs the scores only
    mine = (best >= lo) & (best < hi)
    loc = (best - lo).clamp(0, max(hi - lo - 1, 0))
    if hi > lo:
        pick = final.reshape((B, hi - lo) + tuple(final.shape[1:]))[torch.arange(B, device=final.device), loc]
        img = pick * mine.view(B, 1, 1, 1).to(pick.dtype)
    else:
        img = final.new_zeros((B,) + tuple(final.shape[1:]))
    dist.all_reduce(img)
    return img, best
