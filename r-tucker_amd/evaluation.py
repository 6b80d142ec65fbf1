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
        f = dataset.features
        slots = np.fromiter((dataset._pair_slot[(int(s), int(r))] for s, r in f[:, :2]), dtype=np.int64, count=len(f))
        self.slot_of_item = torch.as_tensor(slots, device=self.device)
        self.features = torch.as_tensor(np.asarray(f, dtype=np.int64), device=self.device)


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


@torch.no_grad()
def evaluate(model, dataset, batch_size=512, device=None, flt: DeviceFilter = None):
    """The reference's ``evaluate`` loop on the device.  ``dataset``: a test-mode ``KG_dataset``.
    Returns (metrics dict averaged over queries, mean BCE loss) like train.py:123-125."""
    from .ops import check_device_errors, index_check
    device = torch.device(device) if device is not None else next(model.parameters()).device
    flt = flt or DeviceFilter(dataset, device)
    model.eval()          # train.py:96; also drops the relation-table cache: the tables are rebuilt once, below
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
