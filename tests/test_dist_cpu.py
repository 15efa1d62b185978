"""CPU, world_size 2, gloo: the multi-GPU layer (image sharding + candidate all-gather + selection)."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import PKG  # noqa: F401


def _worker(rank, world, port, n_images, N, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bem.dist import gather_candidates, gather_ragged, select_best, shard_images
    g = torch.Generator().manual_seed(0)
    allf = torch.rand(n_images * N, 3, 5, 4, generator=g)                 # every rank knows the full truth
    alls = torch.rand(n_images * N, generator=g)
    counts = []
    for r in range(world):
        a, b = shard_images(n_images, r, world)
        counts.append((b - a) * N)
    a, b = shard_images(n_images, rank, world)
    f, s = allf[a * N:b * N].clone(), alls[a * N:b * N].clone()
    if len(set(counts)) == 1:
        gf, gs = gather_candidates(f, s, world)
    else:
        gf, gs = gather_ragged(f, s, counts, world)
    ok = torch.equal(gf, allf) and torch.equal(gs, alls)
    best = select_best(gs, N)
    ok = ok and best == [int(torch.argmax(alls[i * N:(i + 1) * N])) for i in range(n_images)]
    ret[rank] = ok
    dist.destroy_process_group()


@pytest.mark.parametrize("n_images,N", [(4, 3), (5, 2)])
def test_gather_and_select_world2(n_images, N):
    world = 2
    port = 29600 + n_images
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, n_images, N, ret), nprocs=world, join=True)
    assert all(ret[r] for r in range(world))


def test_shard_images_partition():
    from bem.dist import shard_images
    for n, w in ((8, 8), (8, 4), (64, 8), (5, 2), (3, 8)):
        spans = [shard_images(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
        assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1


# ------------------------------------------------------------------------------------------------------------------
# the real multi-GPU path: shard -> per-rank enhance -> exchange (both forms) -> selection == the unsharded result
# ------------------------------------------------------------------------------------------------------------------
def _stub_enhance(imgs, targets, num_samples, **kw):
    """A deterministic stand-in with BEMPipeline.enhance's contract: candidate n of an image is a fixed function of the image
    (so every rank can reproduce any other rank's candidates), score = PSNR against the target; ties are forced on image 1."""
    b = imgs.shape[0]
    N = num_samples
    scale = torch.linspace(0.6, 1.4, N).view(1, N, 1, 1, 1)
    final = (imgs[:, None] * scale).clamp(0, 1)
    mse = ((final - targets[:, None]) ** 2).mean(dim=(2, 3, 4))
    psnr = 10 * torch.log10(1 / mse)
    psnr[imgs[:, 0, 0, 0] > 0.5] = 20.0                              # all-equal scores -> first index must win
    return dict(final=final.reshape(b * N, *imgs.shape[1:]), psnr=psnr.reshape(b * N), N=N)


def _sharded_worker(rank, world, port, n_images, N, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bem.dist import enhance_sharded
    g = torch.Generator().manual_seed(3)
    imgs = torch.rand(n_images, 3, 6, 5, generator=g) * 0.5
    imgs[-1, 0, 0, 0] = 0.9                                          # the tie image
    tg = (imgs * 1.2).clamp(0, 1)
    full = _stub_enhance(imgs, tg, N)
    s = full["psnr"].view(n_images, N)
    ref_best = [row.index(max(row)) for row in s.tolist()]
    ref_img = full["final"].view(n_images, N, 3, 6, 5)[torch.arange(n_images), torch.tensor(ref_best)]
    ok = True
    for mode in ("candidates", "scores"):
        img, best = enhance_sharded(_stub_enhance, imgs, tg, N, rank, world, mode=mode)
        ok = ok and best.tolist() == ref_best and torch.equal(img, ref_img)
    ret[rank] = ok
    dist.destroy_process_group()


@pytest.mark.parametrize("n_images,N", [(4, 3), (5, 4), (1, 2)])
def test_enhance_sharded_equals_unsharded_world2(n_images, N):
    """Even shards, ragged shards, and fewer images than ranks (an empty shard joins the collectives)."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_sharded_worker, args=(world, 29650 + n_images, n_images, N, ret), nprocs=world, join=True)
    assert all(ret[r] for r in range(world))


def test_select_guards_all_zero_scores():
    from bem.dist import select_best
    assert select_best(torch.zeros(6), 3) == [0, 0]
    # ties -> first index; a negative maximum flips the ratio order exactly as psnr / max(psnr) does in eval.py:284
    assert select_best(torch.tensor([1.0, 3.0, 3.0, -2.0, -1.0, -1.0]), 3) == [1, 0]


# ------------------------------------------------------------------------------------------------------------------
# sample-major sharding (the eval driver's B = 1: the N samples of one image split over the ranks)
# ------------------------------------------------------------------------------------------------------------------
def _stub_candidates(imgs, targets, n_local, sample_offset=0, total_samples=None, **kw):
    """BEMPipeline.candidates' contract on a closed form: candidate n of an image depends on (image, n) only, so the samples a rank
    draws for its block [offset, offset + n_local) are the same tensors the unsharded call produces for those indices."""
    N = total_samples or n_local
    b = imgs.shape[0]
    scale = torch.linspace(0.6, 1.4, N)[sample_offset:sample_offset + n_local].view(1, n_local, 1, 1, 1)
    final = (imgs[:, None] * scale).clamp(0, 1)
    mse = ((final - targets[:, None]) ** 2).mean(dim=(2, 3, 4))
    psnr = 10 * torch.log10(1 / mse)
    psnr[imgs[:, 0, 0, 0] > 0.5] = 20.0                              # an image whose candidates all tie: the first index must win
    return dict(final=final.reshape(b * n_local, *imgs.shape[1:]), psnr=psnr.reshape(b * n_local))


def _sample_worker(rank, world, port, n_images, N, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bem.dist import enhance_sample_sharded, gather_samples, shard_samples
    g = torch.Generator().manual_seed(11)
    imgs = torch.rand(n_images, 3, 6, 5, generator=g) * 0.5
    imgs[-1, 0, 0, 0] = 0.9                                          # the tie image
    tg = (imgs * 1.2).clamp(0, 1)
    full = _stub_candidates(imgs, tg, N)
    s = full["psnr"].view(n_images, N)
    ref_best = [row.index(max(row)) for row in s.tolist()]
    ref_img = full["final"].view(n_images, N, 3, 6, 5)[torch.arange(n_images), torch.tensor(ref_best)]
    ok = True
    # the gather puts every (image, sample) row where the unsharded call has it
    lo, hi = shard_samples(N, rank, world)
    mine = _stub_candidates(imgs, tg, hi - lo, sample_offset=lo, total_samples=N) if hi > lo else dict(final=imgs.new_zeros((0, 3, 6, 5)))
    ok = ok and torch.equal(gather_samples(mine["final"], n_images, N, rank, world), full["final"])
    for mode in ("candidates", "scores"):
        img, best = enhance_sample_sharded(_stub_candidates, imgs, tg, N, rank, world, mode=mode)
        ok = ok and best.tolist() == ref_best and torch.equal(img, ref_img)
    ret[rank] = ok
    dist.destroy_process_group()


@pytest.mark.parametrize("n_images,N", [(1, 5), (2, 4), (1, 1)])
def test_sample_major_sharding_equals_unsharded_world2(n_images, N):
    """B = 1 with N = 5 (ragged blocks 3 + 2), B = 2 with even blocks, and N = 1 < world (an empty block joins the collectives):
    selected image and index equal the unsharded first-maximum rule (eval.py:284-285), ties included."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_sample_worker, args=(world, 29700 + 10 * n_images + N, n_images, N, ret), nprocs=world, join=True)
    assert all(ret[r] for r in range(world))
