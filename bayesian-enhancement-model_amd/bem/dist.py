"""Multi-GPU layer: one process per GPU, (image, sample) pairs sharded image-major across ranks, one
exchange step -- an RCCL all-gather (torch.distributed backend "nccl" is RCCL on ROCm) of the finished
candidates and their scores, so that every rank can run the per-image selection (eval.py:268-297) on
all candidates.  No collective touches the data path before that: Stage-I / Stage-II are embarrassingly
parallel over (image, sample) pairs and the weights are replicated."""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_images(n_images: int, rank: int, world: int):
    """Image-major block partition: keeps all N candidates of an image on one rank (selection is local,
    decomp(image) is reused by its N samples).  Returns (start, stop) of this rank's images."""
    per, rem = divmod(n_images, world)
    start = rank * per + min(rank, rem)
    return start, start + per + (1 if rank < rem else 0)


def gather_candidates(final: torch.Tensor, score: torch.Tensor, world: int):
    """final (Bn,3,h,w), score (Bn) on every rank (equal shapes) -> (world*Bn,3,h,w), (world*Bn) on every
    rank, rank-major.  Works with backend nccl (GPU tensors) and gloo (CPU tensors, used by the CPU tests)."""
    if world == 1:
        return final, score
    out_f = torch.empty((world * final.shape[0],) + tuple(final.shape[1:]), device=final.device, dtype=final.dtype)
    out_s = torch.empty(world * score.shape[0], device=score.device, dtype=score.dtype)
    if dist.get_backend() == "gloo":
        fl = list(out_f.chunk(world))
        sl = list(out_s.chunk(world))
        dist.all_gather(fl, final.contiguous())
        dist.all_gather(sl, score.contiguous())
    else:
        dist.all_gather_into_tensor(out_f, final.contiguous())
        dist.all_gather_into_tensor(out_s, score.contiguous())
    return out_f, out_s


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


def select_best(score: torch.Tensor, samples_per_image: int):
    """First maximum per image (list.index(max) semantics of eval.py:285)."""
    s = score.view(-1, samples_per_image).cpu().tolist()
    best = []
    for row in s:
        m = max(row)
        rel = [v / m for v in row]
        best.append(rel.index(max(rel)))
    return best
