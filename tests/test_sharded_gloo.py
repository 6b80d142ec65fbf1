"""Entity-sharded scorer (r-tucker_amd/sharded.py) under the gloo backend, world_size 2,
on CPU.  The shard-local compute is injected from the oracle (tests may use oracle/ as a
stand-in so the HOST logic -- row partition, padding, in-place all-gather slot, layout
conversion -- runs without a GPU); the GPU tests cover the real local kernel."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import gen
from oracle import score_oracle as orc


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _oracle_local(core, R, S, O_loc, h, r, out, **kw):
    out.copy_(orc.score_ref(core, R, S, O_loc, h, r))
    return out


def _worker(rank, world, port, n_ent, n_rel, B, rank3, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from r_tucker_amd.sharded import ShardedEntityScorer
        core, R, S, O = [torch.from_numpy(x) for x in gen.make_params(n_ent, n_rel, rank3, 5)]
        h, r = [torch.from_numpy(x) for x in gen.make_queries(n_ent, n_rel, B, 5)]
        sc = ShardedEntityScorer(n_ent, local_score=_oracle_local)
        O_loc = sc.local_block(O)
        assert O_loc.shape == (sc.shards.n_loc, rank3[2])
        g = sc.score_gathered(core, R, S, O_loc, h, r)
        P = sc.scores_rowmajor(g)
        ref = orc.score_ref(core, R, S, O, h, r)
        ok = P.shape == ref.shape and torch.allclose(P, ref, atol=1e-6)
        # shard-wise view addresses the same numbers without the copy
        v = sc.view_BPn(g)
        lo, hi = sc.shards.bounds(world - 1)
        ok = ok and torch.equal(v[:, world - 1, : hi - lo], P[:, lo:hi])
        q.put((rank, bool(ok), float((P - ref).abs().max())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_ent", [64, 101])      # even and ragged (last shard padded)
def test_sharded_scorer_world2_gloo(n_ent):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(rk, 2, port, n_ent, 5, 9, (3, 8, 8), q)) for rk in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=5) for _ in range(2))
    assert [r[1] for r in res] == [True, True], res


def test_entity_shards_partition():
    from r_tucker_amd.sharded import EntityShards
    for n, w in [(40943, 8), (14951, 8), (1_000_000, 8), (7, 8), (64, 1), (101, 2)]:
        sh = EntityShards(n, w)
        cover = []
        for rk in range(w):
            lo, hi = sh.bounds(rk)
            assert 0 <= lo <= hi <= n and hi - lo <= sh.n_loc
            cover += list(range(lo, hi)) if n < 2000 else []
        if n < 2000:
            assert cover == list(range(n))
        assert sh.bounds(w - 1)[1] == n or sh.n_loc * (w - 1) >= n
        assert sh.n_loc * w >= n
    full = torch.arange(22, dtype=torch.float32).reshape(11, 2)
    sh = EntityShards(11, 4)
    blocks = [sh.take(full, rk) for rk in range(4)]
    assert all(b.shape == (3, 2) for b in blocks)
    assert torch.equal(torch.cat(blocks)[:11], full) and torch.all(torch.cat(blocks)[11:] == 0)
