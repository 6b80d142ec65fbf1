"""Training loss without dense targets (rtk_bce_rows_f32 / rtk_bce_grad_f32 behind
r_tucker_amd.bce_loss_1vN) against the oracle: nn.BCELoss on the reference op sequence with the
dense label-smoothed targets the reference's Dataset builds (train.py:79,136; Dataset.py:43-53)."""
import os

import numpy as np
import pytest
import torch

import gen
from oracle import score_oracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def rt():
    assert torch.cuda.is_available()
    import r_tucker_amd
    r_tucker_amd._lib.load()
    return r_tucker_amd


class _Pairs:
    """The attributes DeviceFilter reads from a KG_dataset, for a synthetic (pair -> objects) table."""
    def __init__(self, pairs, lists, n_ent, eps):
        self._pair_slot = {p: i for i, p in enumerate(pairs)}
        self._ptr = np.concatenate([[0], np.cumsum([len(x) for x in lists])]).astype(np.int64)
        self._obj = np.asarray([x for l in lists for x in l], dtype=np.int64)
        self.features = np.asarray(pairs, dtype=np.int64)
        self.n_ent, self.label_smoothing = n_ent, eps

    def dense(self, ids):
        t = torch.zeros((len(ids), self.n_ent))
        for row, i in enumerate(ids):
            t[row, self._obj[self._ptr[i]:self._ptr[i + 1]]] = 1
        return (1 - self.label_smoothing) * t + self.label_smoothing / self.n_ent


@pytest.mark.parametrize("mode", ["asym", "sym"])
@pytest.mark.parametrize("eps", [0.0, 0.1])
def test_loss_and_gradients_against_oracle(rt, mode, eps):
    n_ent, n_rel, B, rank = 3001, 7, 48, (5, 32, 32)      # odd N: scalar path of the in-place gradient
    core, R, S, O = gen.make_params(n_ent, n_rel, rank, 41, shared=(mode == "sym"))
    rng = np.random.default_rng(41)
    pairs = [(int(s), int(r)) for s, r in zip(rng.permutation(n_ent)[:200], rng.integers(0, n_rel, 200))]
    lists = [rng.integers(0, n_ent, rng.integers(1, 9)).tolist() for _ in pairs]
    lists[3] = lists[3] + lists[3]                         # repeated triples: every object must count once
    ds = _Pairs(pairs, lists, n_ent, eps)
    ids = rng.permutation(200)[:B]
    h = torch.from_numpy(ds.features[ids, 0].copy())
    r = torch.from_numpy(ds.features[ids, 1].copy())
    tc, tR, tS, tO = [torch.from_numpy(x) for x in (core, R, S, O)]
    ref = orc.bce_loss_grads_ref(tc, tR, tS, tO, h, r, ds.dense(ids), shared=(mode == "sym"))

    flt = rt.DeviceFilter(ds, "cuda")
    dc, dR, dS = [x.clone().cuda().requires_grad_(True) for x in (tc, tR, tS)]
    dO = dS if mode == "sym" else tO.clone().cuda().requires_grad_(True)
    loss = rt.bce_loss_1vN(dc, dR, dS, dO, h.cuda(), r.cuda(), flt, torch.from_numpy(ids).cuda(), label_smoothing=eps)
    assert abs(loss.item() - ref[0].item()) <= 2e-6 * max(1.0, abs(ref[0].item()))
    (loss * 3.0).backward()                                # a non-unit upstream gradient
    got = [dc.grad, dR.grad, dS.grad] + ([] if mode == "sym" else [dO.grad])
    for g, e in zip(got, ref[1:]):
        e = 3.0 * e
        assert g.shape == e.shape
        assert (g.cpu() - e).abs().max().item() <= 2e-4 * e.abs().max().item() + 1e-9


def test_wn18rr_train_batch_loss(rt):
    """A real batch of the WN18RR train split (label smoothing 0.1 as in the README recipe)."""
    from r_tucker_amd.data import Data, KG_dataset
    data = Data(os.path.join(ROOT, "data", "WN18RR") + "/", reverse=True)
    train = KG_dataset(data, data.train_data, label_smoothing=0.1)
    n_ent, n_rel, rank = len(data.entities), len(data.relations), (6, 40, 40)
    core, R, S, O = gen.make_params(n_ent, n_rel, rank, 7)
    ids = np.arange(1000, 1128)
    f = train.features[ids]
    targets = train.dense_targets(ids)
    h, r = torch.from_numpy(f[:, 0].copy()), torch.from_numpy(f[:, 1].copy())
    ref = orc.bce_mean_ref(orc.score_ref(*[torch.from_numpy(x) for x in (core, R, S, O)], h, r), targets).item()
    flt = rt.DeviceFilter(train, "cuda")
    with torch.no_grad():
        loss = rt.bce_loss_1vN(*[torch.from_numpy(x).cuda() for x in (core, R, S, O)], h.cuda(), r.cuda(), flt,
                               torch.from_numpy(ids).cuda(), label_smoothing=0.1)
    assert abs(loss.item() - ref) <= 2e-6 * max(1.0, abs(ref))
