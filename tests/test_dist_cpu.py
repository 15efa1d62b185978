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
