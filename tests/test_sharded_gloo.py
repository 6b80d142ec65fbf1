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


def _split_worker(rank, world, port, n_ent, B, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from r_tucker_amd.sharded import ShardedEntityScorer
        n_rel, rank3 = 5, (3, 8, 8)
        core, R, S, O = [torch.from_numpy(x) for x in gen.make_params(n_ent, n_rel, rank3, 11)]
        h, r = [torch.from_numpy(x) for x in gen.make_queries(n_ent, n_rel, B, 11)]
        calls = []

        def qv(core_, R_, S_, hh, rr, **kw):          # stage 1 of a SLICE of the batch
            calls.append(int(hh.numel()))
            return orc.query_vectors_ref(core_, R_, S_, hh, rr)

        def score_from_v(v, O_loc, out, **kw):        # stage 2 on the gathered vectors
            out.copy_(torch.sigmoid(v @ O_loc.T))

        sc = ShardedEntityScorer(n_ent, local_score=_oracle_local, stage1="split", query_vectors_fn=qv,
                                 score_from_v_fn=score_from_v)
        P = sc.score(core, R, S, sc.local_block(O), h, r)
        ref = orc.score_ref(core, R, S, O, h, r)
        B_loc = -(-B // world)
        want = max(0, min(B, (rank + 1) * B_loc) - min(B, rank * B_loc))
        ok = torch.allclose(P, ref, atol=1e-6) and calls == ([want] if want else [])
        # the "auto" choice: small relation rank -> replicated (no query-vector exchange)
        sc2 = ShardedEntityScorer(n_ent, local_score=_oracle_local, stage1="auto", query_vectors_fn=qv,
                                  score_from_v_fn=score_from_v)
        n_before = len(calls)
        P2 = sc2.score(core, R, S, sc2.local_block(O), h, r)
        ok = ok and torch.allclose(P2, ref, atol=1e-6) and len(calls) == n_before
        q.put((rank, bool(ok), float((P - ref).abs().max())))
    finally:
        dist.destroy_process_group()


def _relation_worker(rank, world, port, n_ent, B, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from r_tucker_amd.sharded import ShardedEntityScorer
        n_rel, rank3 = 5, (3, 8, 8)
        core, R, S, O = [torch.from_numpy(x) for x in gen.make_params(n_ent, n_rel, rank3, 13)]
        h, r = [torch.from_numpy(x) for x in gen.make_queries(n_ent, n_rel, B, 13)]
        tables = torch.einsum("ua,abc->ubc", R, core)
        mine = []

        def qv_part(core_, R_, S_, hh, rr, tables_, part, n_parts, out):     # the C ABI's contract on the CPU
            sel = (rr % n_parts) == part
            mine.append(int(sel.sum()))
            if sel.any():
                out[sel] = torch.einsum("db,dbc->dc", S_[hh[sel]], tables_[rr[sel]])
            return out

        def score_from_v(v, O_loc, out, **kw):
            out.copy_(torch.sigmoid(v @ O_loc.T))

        sc = ShardedEntityScorer(n_ent, local_score=_oracle_local, stage1="relation", score_from_v_fn=score_from_v)
        sc.query_vectors_part_fn = qv_part
        P = sc.score(core, R, S, sc.local_block(O), h, r, tables=tables)
        ref = orc.score_ref(core, R, S, O, h, r)
        ok = torch.allclose(P, ref, atol=1e-5) and mine == [int(((r % world) == rank).sum())]
        q.put((rank, bool(ok), float((P - ref).abs().max())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_ent,B", [(64, 9), (37, 1)])
def test_sharded_stage1_by_relation_world2_gloo(n_ent, B):
    """Stage 1 split over the ranks by relation id (rank p: relations congruent to p), one all-reduce(SUM) of the
    zero-initialised B x c vectors, stage 2 on the local entity shard: same matrix as the oracle on one device."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_relation_worker, args=(rk, 2, port, n_ent, B, q)) for rk in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=5) for _ in range(2))
    assert [r[1] for r in res] == [True, True], res


@pytest.mark.parametrize("n_ent,B", [(64, 9), (101, 2), (37, 1)])   # ragged batch slices, an empty slice, ragged shards
def test_sharded_stage1_split_world2_gloo(n_ent, B):
    """Stage 1 split over the ranks (each contracts ceil(B/P) queries, one all-gather of the B x c vectors),
    stage 2 on the local entity shard, scores all-gathered: same matrix as the oracle on one device."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_split_worker, args=(rk, 2, port, n_ent, B, q)) for rk in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=5) for _ in range(2))
    assert [r[1] for r in res] == [True, True], res


class _Flt:
    """The three arrays of evaluation.DeviceFilter, on the CPU."""
    def __init__(self, slot_of_item, pair_ptr, pair_obj):
        self.slot_of_item, self.pair_ptr, self.pair_obj = slot_of_item, pair_ptr, pair_obj


def _cpu_target_scores(block, obj, col0):
    j = obj - col0
    own = (j >= 0) & (j < block.shape[1])
    pt = torch.full((block.shape[0],), float("-inf"))
    pt[own] = block[own.nonzero().view(-1), j[own]]
    return pt


def _cpu_rank_counts(block, obj, col0, pt, flt, item_ids, want_bce):
    """Plain restatement of what rtk_filtered_rank_partial_f32 counts, for one column block."""
    B, n = block.shape
    counts = torch.zeros(B, dtype=torch.int32)
    for d in range(B):
        p = block[d].clone()
        t = int(obj[d]) - col0
        if flt is not None:
            s = int(flt.slot_of_item[item_ids[d]])
            for g in flt.pair_obj[flt.pair_ptr[s]:flt.pair_ptr[s + 1]].tolist():
                if 0 <= g - col0 < n and g - col0 != t:
                    p[g - col0] = 0.0
        before = torch.arange(n) < t
        counts[d] = int((p > pt[d]).sum() + ((p == pt[d]) & before).sum())
    return (counts, torch.zeros(B, dtype=torch.float64)) if want_bce else counts


def _rank_worker(rank, world, port, n_ent, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from r_tucker_amd.sharded import ShardedEntityScorer
        n_rel, B, rank3 = 5, 24, (3, 8, 8)
        core, R, S, O = [torch.from_numpy(x) for x in gen.make_params(n_ent, n_rel, rank3, 9)]
        O = (O * 8).round() / 8                       # coarse values -> exact ties between scores
        S = (S * 4).round() / 4
        h, r = [torch.from_numpy(x) for x in gen.make_queries(n_ent, n_rel, B, 9)]
        rng = np.random.default_rng(9)
        obj = torch.from_numpy(rng.integers(0, n_ent, B))
        # one filter list per query: the queried object plus up to 5 other known-true objects
        lists = [sorted(set([int(obj[d])] + rng.integers(0, n_ent, rng.integers(0, 6)).tolist())) for d in range(B)]
        ptr = torch.tensor(np.concatenate([[0], np.cumsum([len(x) for x in lists])]), dtype=torch.int64)
        flt = _Flt(torch.arange(B), ptr, torch.tensor([x for l in lists for x in l], dtype=torch.int64))
        sc = ShardedEntityScorer(n_ent, local_score=_oracle_local)
        ranks = sc.filtered_ranks(core, R, S, sc.local_block(O), h, r, obj, flt, torch.arange(B),
                                  target_scores_fn=_cpu_target_scores, rank_counts_fn=_cpu_rank_counts)
        P = orc.score_ref(core, R, S, O, h, r)
        targets = torch.zeros_like(P)
        for d, l in enumerate(lists):
            targets[d, l] = 1.0
        ref = orc.filter_and_rank_stable(P, targets, obj)
        q.put((rank, bool(torch.equal(ranks.long(), ref.long())), int((ranks.long() - ref.long()).abs().max())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_ent", [64, 101])
def test_sharded_filtered_ranks_world2_gloo(n_ent):
    """Ranking without the gather: per-block counts + two small all-reduces give the ranks the
    oracle computes on the full score matrix (stable tie order, filtered objects zeroed)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_worker, args=(rk, 2, port, n_ent, q)) for rk in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=5) for _ in range(2))
    assert [r[1] for r in res] == [True, True], res


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
