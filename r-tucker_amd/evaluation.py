"""Evaluation on the device: the reference's ``evaluate()`` (train.py:94-125) with the
score path and the ranking tail both in HIP.

``evaluate(model, dataset)`` returns the same dictionary the reference returns --
``{"mrr", "hits@1", "hits@3", "hits@10"}`` averaged over the queries, plus the mean BCE
loss -- but never builds the dense (B, N) target matrix on the host and never sorts:
the filter lists of every (subject, relation) pair are uploaded once as a CSR and
``rtk_filtered_rank_f32`` counts the rank of the queried object in one pass over the
score row.  Tie order is that of a stable descending sort (torch's CUDA sort /
``sort(stable=True)``); the reference's default CPU sort is unstable, so on exactly tied
scores its own ranks are implementation-defined.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from .tucker import SFTucker, Tucker


class DeviceFilter:
    """The (subject, relation) -> known-true-objects CSR of a ``KG_dataset`` on the device."""

    def __init__(self, dataset, device):
        self.device = torch.device(device)
        # every object of a pair once (the dense targets of Dataset.py:43-50 are idempotent under
        # repeated triples; the kernels that walk the CSR are not)
        ptr, obj = np.asarray(dataset._ptr, dtype=np.int64), np.asarray(dataset._obj, dtype=np.int64)
        seg = np.repeat(np.arange(len(ptr) - 1, dtype=np.int64), np.diff(ptr))
        key = np.unique(seg * (int(obj.max()) + 1 if len(obj) else 1) + obj)
        m = int(obj.max()) + 1 if len(obj) else 1
        ptr = np.concatenate([[0], np.cumsum(np.bincount(key // m, minlength=len(ptr) - 1))]).astype(np.int64)
        obj = (key % m).astype(np.int64)
        self.pair_ptr = torch.as_tensor(ptr, device=self.device)
        self.pair_obj = torch.as_tensor(obj, device=self.device)
        f = np.asarray(dataset.features, dtype=np.int64)
        # pair slot of every item, vectorised (a Python loop over the items cost more than the evaluation pass itself):
        # the pairs' (s, r) keys sorted once, the items' keys located by binary search
        if hasattr(dataset, "_pairs"):
            pairs = np.asarray(dataset._pairs, dtype=np.int64).reshape(-1, 2)
        else:                                    # (a dataset that only keeps the (s, r) -> slot dictionary)
            pairs = np.zeros((len(dataset._pair_slot), 2), dtype=np.int64)
            for (s_, r_), i in dataset._pair_slot.items():
                pairs[i] = (s_, r_)
        nr = int(max(pairs[:, 1].max() if len(pairs) else 0, f[:, 1].max() if len(f) else 0)) + 1
        pk = pairs[:, 0] * nr + pairs[:, 1]
        order = np.argsort(pk, kind="stable")
        ik = f[:, 0] * nr + f[:, 1]
        pos = np.searchsorted(pk[order], ik)
        if len(f) and (pos.max(initial=0) >= len(pk) or not np.array_equal(pk[order][np.minimum(pos, len(pk) - 1)], ik)):
            raise KeyError("an item's (subject, relation) pair is not among the dataset's pairs")
        slots = order[pos] if len(f) else np.zeros(0, np.int64)
        self.slot_of_item = torch.as_tensor(slots, device=self.device)
        self.features = torch.as_tensor(f, device=self.device)
        # contiguous columns: the evaluation loop hands the kernels pointer offsets into them
        self.subj = self.features[:, 0].contiguous()
        self.rel = self.features[:, 1].contiguous()
        self.obj = self.features[:, 2].contiguous() if f.shape[1] > 2 else None
        self._plans = {}

    @classmethod
    def of(cls, dataset, device):
        """The filter of ``dataset`` on ``device``, built once per dataset object (``evaluate()`` runs twice per epoch)."""
        device = torch.device(device)
        if device.type == "cuda" and device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        cache = dataset.__dict__.setdefault("_device_filters", {})
        if device not in cache:
            cache[device] = cls(dataset, device)
        return cache[device]


def filtered_ranks(P: torch.Tensor, obj_idx: torch.Tensor, flt: DeviceFilter = None, item_ids: torch.Tensor = None,
                   want_bce: bool = False):
    """Ranks (int32, B) of ``obj_idx`` in the rows of ``P`` after filtering; optionally the per-row BCE sums."""
    lib = _lib.load()
    if not P.is_cuda or P.dtype != torch.float32 or P.dim() != 2 or P.stride(1) != 1:
        raise RuntimeError("P must be a float32 (B, N) GPU tensor with unit column stride")
    B, N = P.shape
    dev = P.device
    obj = obj_idx.to(device=dev, dtype=torch.int64).contiguous().view(-1)
    if obj.numel() != B:
        raise RuntimeError("obj_idx must have one entry per row of P")
    ranks = torch.empty(B, dtype=torch.int32, device=dev)
    bce = torch.empty(B, dtype=torch.float64, device=dev) if want_bce else None
    slot = ptr = objs = None
    if flt is not None:
        slot = flt.slot_of_item[item_ids.to(dev)].contiguous()
        ptr, objs = flt.pair_ptr, flt.pair_obj
    with torch.cuda.device(dev):
        sp = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(lib.rtk_filtered_rank_f32(P.data_ptr(), B, N, P.stride(0), obj.data_ptr(),
                                             slot.data_ptr() if slot is not None else None,
                                             ptr.data_ptr() if ptr is not None else None,
                                             objs.data_ptr() if objs is not None else None,
                                             ranks.data_ptr(), bce.data_ptr() if want_bce else None, sp),
                   "rtk_filtered_rank_f32")
    return (ranks, bce) if want_bce else ranks


def target_scores_block(P: torch.Tensor, obj_idx: torch.Tensor, col0: int) -> torch.Tensor:
    """Scores of the queried objects that fall into this column block, -inf for the others
    (step 1 of the sharded ranking; all-reduce MAX over the ranks completes it)."""
    lib = _lib.load()
    B, n_loc = P.shape
    dev = P.device
    obj = obj_idx.to(device=dev, dtype=torch.int64).contiguous().view(-1)
    pt = torch.empty(B, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        sp = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(lib.rtk_target_scores_f32(P.data_ptr(), B, n_loc, P.stride(0), int(col0), obj.data_ptr(),
                                             pt.data_ptr(), sp), "rtk_target_scores_f32")
    return pt


def rank_counts_block(P: torch.Tensor, obj_idx: torch.Tensor, col0: int, target_scores: torch.Tensor,
                      flt: DeviceFilter = None, item_ids: torch.Tensor = None, want_bce: bool = False):
    """This column block's share of the rank count (int32, B) and, optionally, of the BCE row sums
    (step 2 of the sharded ranking; all-reduce SUM, then rank = 1 + count)."""
    lib = _lib.load()
    if not P.is_cuda or P.dtype != torch.float32 or P.dim() != 2 or P.stride(1) != 1:
        raise RuntimeError("P must be a float32 (B, n_local) GPU tensor with unit column stride")
    B, n_loc = P.shape
    dev = P.device
    obj = obj_idx.to(device=dev, dtype=torch.int64).contiguous().view(-1)
    counts = torch.empty(B, dtype=torch.int32, device=dev)
    bce = torch.empty(B, dtype=torch.float64, device=dev) if want_bce else None
    slot = ptr = objs = None
    if flt is not None:
        slot = flt.slot_of_item[item_ids.to(dev)].contiguous()
        ptr, objs = flt.pair_ptr, flt.pair_obj
    pt = target_scores.to(device=dev, dtype=torch.float32).contiguous()
    with torch.cuda.device(dev):
        sp = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(lib.rtk_filtered_rank_partial_f32(P.data_ptr(), B, n_loc, P.stride(0), int(col0), pt.data_ptr(),
                                                     obj.data_ptr(),
                                                     slot.data_ptr() if slot is not None else None,
                                                     ptr.data_ptr() if ptr is not None else None,
                                                     objs.data_ptr() if objs is not None else None,
                                                     counts.data_ptr(), bce.data_ptr() if want_bce else None, sp),
                   "rtk_filtered_rank_partial_f32")
    return (counts, bce) if want_bce else counts


def metrics_from_ranks(ranks: torch.Tensor):
    """Batch SUMS like ``src/utils/metrics.py`` (mrr = sum 1/rank, hits@k = #(rank <= k))."""
    r = ranks.to(torch.float64)
    return {"mrr": (1.0 / r).sum(), "hits@1": (ranks <= 1).sum(), "hits@3": (ranks <= 3).sum(),
            "hits@10": (ranks <= 10).sum()}


class _EvalPlan:
    """Buffers of one evaluation pass for fixed (batch size, entity count, operand dtype): the score matrix of ONE
    batch (every batch is consumed by the ranking kernel on the same stream before the next one overwrites it),
    ranks, BCE row sums, packed query planes, running sums."""

    def __init__(self, B, N, c, n_rel, dtype, device):
        from .ops import alloc_scores
        lib = _lib.load()
        bf16 = dtype == torch.bfloat16
        self.dcode = _lib.RTK_BF16 if bf16 else _lib.RTK_F32
        self.P = alloc_scores(B, N, device)
        self.ld = self.P.stride(0) if B > 1 else N
        self.ranks = torch.empty(B, dtype=torch.int32, device=device)
        self.bce = torch.empty(B, dtype=torch.float64, device=device)
        self.acc = torch.zeros(5, dtype=torch.float64, device=device)
        self.qp = torch.empty(lib.rtk_packed_query_bytes(self.dcode, B, c), dtype=torch.uint8, device=device)
        self.ws_bytes = lib.rtk_from_tables_workspace_bytes(B, n_rel)


@torch.no_grad()
def evaluate(model, dataset, batch_size=512, device=None, flt: DeviceFilter = None):
    """The reference's ``evaluate`` loop on the device.  ``dataset``: a test-mode ``KG_dataset``.
    Returns (metrics dict averaged over queries, mean BCE loss) like train.py:123-125.

    Per batch: stage 1 against the relation tables of the frozen parameters, the score kernel, the filtered-rank
    kernel over the score matrix, the metric sums -- four calls into the C ABI on buffers that are allocated once per
    (dataset, batch size) and reused; the filter CSR is built once per dataset object.  Out-of-range ids are reported
    at the pass's own synchronisation point (reference: IndexError)."""
    from . import ops
    device = torch.device(device) if device is not None else next(model.parameters()).device
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    flt = flt or DeviceFilter.of(dataset, device)
    model.eval()          # train.py:96; also drops the relation-table cache: the tables are rebuilt once, below
    sym = hasattr(model, "E")
    core, R = model.core.data, model.R.weight.data
    S = model.E.weight.data if sym else model.S.weight.data
    O = S if sym else model.O.weight.data
    n = len(dataset)
    a, b, c = core.shape
    if (b != c or core.dtype not in (torch.float32, torch.bfloat16) or c > 512 or not core.is_cuda
            or any(t.dtype != core.dtype or t.device != core.device or not t.is_contiguous() for t in (R, S, O))):
        return _evaluate_generic(model, dataset, batch_size, device, flt)      # (raises what the closure raises)
    lib = _lib.load()
    N, n_rel = O.shape[0], R.shape[0]
    bf16 = core.dtype == torch.bfloat16
    key = (batch_size, N, c, n_rel, core.dtype)
    plan = flt._plans.get(key)
    if plan is None:
        plan = flt._plans[key] = _EvalPlan(min(batch_size, max(n, 1)), N, c, n_rel, core.dtype, device)
    tables = model._cached_tables(core, R) if hasattr(model, "_cached_tables") else None
    if tables is None:
        tables = ops.relation_tables(core, R)
    ft = lib.rtk_query_vectors_from_tables_bf16 if bf16 else lib.rtk_query_vectors_from_tables_f32
    sp_fn = lib.rtk_score_packed_bf16 if bf16 else lib.rtk_score_packed_f32
    sflags = _lib.RTK_SCORE_SIGMOID | (_lib.RTK_SCORE_SIGMOID_FAST if ops.DEFAULT_SIGMOID == "fast" else 0)
    with torch.cuda.device(device):
        sp = torch.cuda.current_stream(device).cuda_stream
        ws = ops._workspace(device, sp, plan.ws_bytes)
        plan.acc.zero_()
        P, qp, ranks, bce, acc = (plan.P.data_ptr(), plan.qp.data_ptr(), plan.ranks.data_ptr(), plan.bce.data_ptr(),
                                  plan.acc.data_ptr())
        tp, Sp, Op, wsp, wsn = tables.data_ptr(), S.data_ptr(), O.data_ptr(), ws.data_ptr(), ws.numel()
        hp, rp, op, slp = flt.subj.data_ptr(), flt.rel.data_ptr(), flt.obj.data_ptr(), flt.slot_of_item.data_ptr()
        ptr, objs = flt.pair_ptr.data_ptr(), flt.pair_obj.data_ptr()
        n_batches = 0
        for lo in range(0, n, batch_size):
            nb = min(batch_size, n - lo)
            o8 = 8 * lo
            _lib.check(ft(tp, n_rel, b, c, Sp, S.shape[0], rp + o8, hp + o8, nb, None, qp, wsp, wsn, sp),
                       "rtk_query_vectors_from_tables")
            _lib.check(sp_fn(qp, nb, c, Op, N, P, plan.ld, sflags, sp), "rtk_score_packed")
            _lib.check(lib.rtk_filtered_rank_f32(P, nb, N, plan.ld, op + o8, slp + o8, ptr, objs, ranks, bce, sp),
                       "rtk_filtered_rank_f32")
            # the reference averages the per-batch MEAN losses (train.py:113,125)
            _lib.check(lib.rtk_rank_metrics_scaled_f64(ranks, bce, nb, 1.0 / (nb * N), acc, sp), "rtk_rank_metrics_scaled_f64")
            n_batches += 1
    acc = plan.acc.cpu().tolist()
    ops.check_device_errors(device)
    sums = {"mrr": acc[0] / n, "hits@1": acc[1] / n, "hits@3": acc[2] / n, "hits@10": acc[3] / n}
    return sums, acc[4] / max(n_batches, 1)


@torch.no_grad()
def _evaluate_generic(model, dataset, batch_size, device, flt):
    """The same loop through the model's closure (any operand the closure accepts; shapes the fast path does not
    cover end in the closure's own error)."""
    from .ops import check_device_errors, index_check
    if hasattr(model, "E"):
        T = SFTucker(model.core.data, [model.R.weight], num_shared_factors=2, shared_factor=model.E.weight)
    else:
        T = Tucker(model.core.data, [model.R.weight, model.S.weight, model.O.weight])
    n = len(dataset)
    lib = _lib.load()
    acc = torch.zeros(5, dtype=torch.float64, device=device)   # running sums on the device: one sync at the end
    loss_terms = 0
    n_batches = 0
    # out-of-range ids: checked once, at the loop's own synchronisation point below (reference: IndexError)
    with index_check("deferred"):
        for lo in range(0, n, batch_size):
            hi = min(lo + batch_size, n)
            ids = torch.arange(lo, hi, device=device)
            f = flt.features[lo:hi]
            P = model(f[:, 0], f[:, 1])(T)       # eval mode + no_grad: relation tables built once, reused by every batch
            ranks, bce = filtered_ranks(P, f[:, 2], flt, ids, want_bce=True)
            # the reference averages the per-batch MEAN losses (train.py:113,125): scale this batch's row sums
            bce.mul_(1.0 / (P.shape[0] * P.shape[1]))
            with torch.cuda.device(device):
                _lib.check(lib.rtk_rank_metrics_f64(ranks.data_ptr(), bce.data_ptr(), hi - lo, acc.data_ptr(),
                                                    torch.cuda.current_stream(device).cuda_stream), "rtk_rank_metrics_f64")
            n_batches += 1
    acc = acc.cpu().tolist()
    check_device_errors(device)
    sums = {"mrr": acc[0] / n, "hits@1": acc[1] / n, "hits@3": acc[2] / n, "hits@10": acc[3] / n}
    return sums, acc[4] / n_batches
