"""CPU oracle for R-TuckER's 1-vs-all Tucker scoring path.

*** TEST INFRASTRUCTURE -- NOT PRODUCT CODE. ***
Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module, and there only as the checker / the timed
CPU baseline.  Nothing under ``r-tucker_amd/`` imports it; the product path
fails loudly when the HIP library is missing instead of falling back to this.

Parity status: PINNED.  ``score_ref`` / ``filter_and_rank`` are checked against
golden vectors produced by the reference's own ``score_fn``,
``filter_predictions`` and ``metrics`` (imported from the reference in the build
container by ``tests/golden/make_golden.py``; fixtures under ``tests/golden/``).

Every function cites the reference lines it restates (paths relative to the
reference checkout).  Two flavours are provided:

* ``*_ref``   -- the reference's op sequence on torch CPU ops in the operands'
                 dtype (fp32): gathers -> einsum -> bmm -> matmul -> sigmoid.
                 This is what "the reference CPU path" means, and it is what
                 ``bench.py`` times as ``cpu_baseline`` (kind "port").
* ``*_exact`` -- the same mathematics in numpy float64 (einsum contraction in
                 one go); used to measure *both* the fp32 CPU path's and the
                 HIP path's rounding error against a common, higher-precision
                 value, so a tolerance can be stated rather than guessed.
"""
from __future__ import annotations

import numpy as np
import torch


# --------------------------------------------------------------------------
# containers: what ``score_fn(T)`` reads from ``T``
# --------------------------------------------------------------------------
class TuckerBag:
    """Attribute bag with the fields ``score_fn`` reads from ``tucker_riemopt.Tucker``.

    Reference use: ``train.py:41`` builds ``Tucker(core, [R, S, O])``;
    ``src/model/asymmetric/R_TuckER.py:43-47`` reads ``.core`` and ``.factors``.
    """

    def __init__(self, core, factors):
        self.core = core
        self.factors = list(factors)


class SFTuckerBag:
    """Attribute bag for ``tucker_riemopt.SFTucker`` (``train.py:39``;
    ``src/model/symmetric/R_TuckER.py:40-44`` reads ``.core``,
    ``.regular_factors`` and ``.shared_factor``)."""

    def __init__(self, core, regular_factors, num_shared_factors, shared_factor):
        self.core = core
        self.regular_factors = list(regular_factors)
        self.num_shared_factors = num_shared_factors
        self.shared_factor = shared_factor


# --------------------------------------------------------------------------
# the five-op sequence, reference order, torch CPU
# --------------------------------------------------------------------------
def logits_ref(core, R, S, O, subject_idx, relation_idx):
    """Logits of the 1-vs-all score, reference op order.

    ``src/model/asymmetric/R_TuckER.py:43-47`` (and ``symmetric/...:40-44`` with
    ``S is O``):  Rb = R[r]; Sb = S[h]; W = einsum("abc,da->dbc", G, Rb);
    v = bmm(Sb.view(-1,1,b), W).view(-1, b);  Z = v @ O.T
    The ``.view(-1, b)`` of a ``(B,1,c)`` result is what forces ``b == c``
    (SURVEY.md A10): reproduced, so a ``b != c`` core raises here as well.
    """
    relations = R[relation_idx, :]
    subjects = S[subject_idx, :]
    preds = torch.einsum("abc,da->dbc", core, relations)
    preds = torch.bmm(subjects.view(-1, 1, subjects.shape[1]), preds).view(-1, subjects.shape[1])
    return preds @ O.T


def score_ref(core, R, S, O, subject_idx, relation_idx):
    """``sigmoid(logits)`` -- ``src/model/asymmetric/R_TuckER.py:48``."""
    return torch.sigmoid(logits_ref(core, R, S, O, subject_idx, relation_idx))


def score_fn_ref(T, subject_idx, relation_idx):
    """``score_fn(T)`` for either container kind (asymmetric: ``R_TuckER.py:42-48``;
    symmetric: ``symmetric/R_TuckER.py:39-45``)."""
    if hasattr(T, "shared_factor"):
        return score_ref(T.core, T.regular_factors[0], T.shared_factor, T.shared_factor,
                         subject_idx, relation_idx)
    return score_ref(T.core, T.factors[0], T.factors[1], T.factors[2], subject_idx, relation_idx)


def query_vectors_ref(core, R, S, subject_idx, relation_idx):
    """Only the core-contraction half: ``v[d,:] = Sb[d] . (G x_0 Rb[d])``
    (``R_TuckER.py:43-46``).  Returned so tests can check the two HIP stages
    separately."""
    relations = R[relation_idx, :]
    subjects = S[subject_idx, :]
    W = torch.einsum("abc,da->dbc", core, relations)
    return torch.bmm(subjects.view(-1, 1, subjects.shape[1]), W).view(-1, subjects.shape[1])


# --------------------------------------------------------------------------
# float64 restatement (error yard-stick)
# --------------------------------------------------------------------------
def _np64(x):
    if isinstance(x, torch.Tensor):
        x = x.detach().to(torch.float64).cpu().numpy()
    return np.asarray(x, dtype=np.float64)


def query_vectors_exact(core, R, S, subject_idx, relation_idx):
    G, Rn, Sn = _np64(core), _np64(R), _np64(S)
    h = np.asarray(subject_idx).astype(np.int64)
    r = np.asarray(relation_idx).astype(np.int64)
    return np.einsum("abc,da,db->dc", G, Rn[r], Sn[h], optimize=True)


def logits_exact(core, R, S, O, subject_idx, relation_idx):
    v = query_vectors_exact(core, R, S, subject_idx, relation_idx)
    return v @ _np64(O).T


def score_exact(core, R, S, O, subject_idx, relation_idx):
    z = logits_exact(core, R, S, O, subject_idx, relation_idx)
    return 1.0 / (1.0 + np.exp(-z))


# --------------------------------------------------------------------------
# gradients of the path (what the optimizer differentiates, train.py:79-82)
# --------------------------------------------------------------------------
def score_grads_ref(core, R, S, O, subject_idx, relation_idx, weight, shared=False):
    """d(sum(P * weight)) / d{core, R, S, O} through the reference op sequence
    with torch autograd on CPU.  ``shared=True``: S and O are the same matrix
    (symmetric model) and one gradient is returned for it."""
    core = core.detach().clone().requires_grad_(True)
    R = R.detach().clone().requires_grad_(True)
    S = S.detach().clone().requires_grad_(True)
    if shared:
        P = score_ref(core, R, S, S, subject_idx, relation_idx)
        (P * weight).sum().backward()
        return core.grad, R.grad, S.grad
    O = O.detach().clone().requires_grad_(True)
    P = score_ref(core, R, S, O, subject_idx, relation_idx)
    (P * weight).sum().backward()
    return core.grad, R.grad, S.grad, O.grad


# --------------------------------------------------------------------------
# the eval tail: filtered ranking (parity harness for MRR)
# --------------------------------------------------------------------------
def filter_predictions_ref(predictions, targets, filt):
    """``src/utils/utils.py:15-22``: keep the score of the queried object, zero the
    scores (and targets) of every other known-true object, in place; afterwards
    each target row holds exactly one 1."""
    keep = predictions.gather(1, filt)
    predictions[targets == 1] = 0
    targets[targets == 1] = 0
    predictions.scatter_(1, filt, keep)
    targets.scatter_(1, filt, torch.ones(keep.shape, device=targets.device, dtype=targets.dtype))
    return predictions, targets


def ranks_ref(predictions, targets):
    """Per-query rank as ``src/utils/metrics.py:5-8`` computes it: descending sort
    of the scores, gather the targets, position of the first 1 (+1)."""
    _, idx = torch.sort(predictions, dim=1, descending=True)
    targets_sorted = targets.gather(1, idx)
    return targets_sorted.argmax(dim=1) + 1, targets_sorted


def metrics_ref(predictions, targets):
    """``src/utils/metrics.py:4-22``: batch *sums* of 1/rank and clipped hits@k."""
    ranks, targets_sorted = ranks_ref(predictions, targets)
    out = {"mrr": torch.sum(1 / ranks)}
    for k in (1, 3, 10):
        hits_k = targets_sorted[:, :k].sum(dim=1).float()
        hits_k[hits_k > 1] = 1
        out[f"hits@{k}"] = hits_k.sum()
    return out


def filter_and_rank(predictions, targets, object_idx):
    """filter_predictions + ranks, on copies (the reference mutates in place,
    ``train.py:115-117``).  Returns (ranks[int64 B], metrics dict of batch sums)."""
    p = predictions.clone()
    t = targets.clone()
    p, t = filter_predictions_ref(p, t, object_idx.reshape(-1, 1))
    ranks, _ = ranks_ref(p, t)
    return ranks, metrics_ref(p, t)


def ranks_stable_ref(predictions, targets):
    """Ranks under a STABLE descending sort (tied scores keep index order) -- the order torch's
    CUDA sort gives the reference; ``metrics.py:5`` itself calls ``torch.sort`` with the default
    ``stable=False``, whose CPU tie order is unspecified."""
    _, idx = torch.sort(predictions, dim=1, descending=True, stable=True)
    return targets.gather(1, idx).argmax(dim=1) + 1


def filter_and_rank_stable(predictions, targets, object_idx):
    p, t = predictions.clone(), targets.clone()
    p, t = filter_predictions_ref(p, t, object_idx.reshape(-1, 1))
    return ranks_stable_ref(p, t)


def bce_mean_ref(predictions, targets):
    """``nn.BCELoss(reduction="mean")(predictions, targets)`` -- train.py:113,136."""
    return torch.nn.functional.binary_cross_entropy(predictions, targets, reduction="mean")


def bce_loss_grads_ref(core, R, S, O, subject_idx, relation_idx, targets, shared=False):
    """Training loss term of train.py:79 (``criterion(score_fn(T), targets)`` with
    ``nn.BCELoss()``, train.py:136) and its gradients w.r.t. the Tucker operands, by torch autograd
    on CPU through the reference op sequence.  ``targets``: the dense label-smoothed matrix the
    reference's Dataset builds (Dataset.py:43-53).  Returns (loss, g_core, g_R, g_S[, g_O])."""
    core = core.detach().clone().requires_grad_(True)
    R = R.detach().clone().requires_grad_(True)
    S = S.detach().clone().requires_grad_(True)
    if shared:
        loss = bce_mean_ref(score_ref(core, R, S, S, subject_idx, relation_idx), targets)
        loss.backward()
        return loss.detach(), core.grad, R.grad, S.grad
    O = O.detach().clone().requires_grad_(True)
    loss = bce_mean_ref(score_ref(core, R, S, O, subject_idx, relation_idx), targets)
    loss.backward()
    return loss.detach(), core.grad, R.grad, S.grad, O.grad
