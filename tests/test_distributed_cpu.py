"""World-size-2 gloo tests (CPU) of the sharded-scoring plumbing: contiguous shard bounds, the all-gather of per-rank
logits (equal and ragged shards), rank-order concatenation.  The per-rank scorer here is a stand-in row function -
the collective and shard arithmetic are what is under test; the GPU path itself is covered by the -m gpu suite."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from leclip_amd import parallel


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _score(x):  # deterministic per-row "logits": depends only on the row, like the real scorer
    return torch.stack([x.flatten(1).sum(1) * (c + 1) for c in range(5)], dim=1)


def _worker(rank, world, port, n_global, out_q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    r, w, _ = parallel.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    g = torch.Generator().manual_seed(5)
    images = torch.randn(n_global, 3, 4, 4, generator=g)
    sc = parallel.ShardedScorer(_score, n_out=5)
    lo, hi = parallel.shard_bounds(n_global, rank, world)
    gathered = sc.score_local(images[lo:hi].contiguous(), n_global)
    gathered2 = sc.score_global(images)
    ref = _score(images)
    ok = torch.allclose(gathered, ref) and torch.equal(gathered, gathered2) and gathered.shape == (n_global, 5)
    out_q.put((rank, bool(ok), (lo, hi)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_global", [8, 7, 2, 1])
def test_sharded_scoring_world2(n_global):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_global, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    (lo0, hi0), (lo1, hi1) = res[0][2], res[1][2]
    assert lo0 == 0 and hi0 == lo1 and hi1 == n_global and (hi0 - lo0) - (hi1 - lo1) in (0, 1)


def test_shard_bounds_cover_exactly():
    for n in range(0, 40):
        for w in (1, 2, 3, 8):
            spans = [parallel.shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_single_process_is_identity():
    x = torch.randn(6, 3, 4, 4)
    sc = parallel.ShardedScorer(_score)
    assert sc.world == 1 and torch.equal(sc.score_local(x), _score(x)) and torch.equal(parallel.all_gather_rows(_score(x)), _score(x))
