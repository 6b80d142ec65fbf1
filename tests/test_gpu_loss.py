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


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("mode", ["asym", "sym"])
@pytest.mark.parametrize("eps", [0.0, 0.1])
def test_loss_and_gradients_against_oracle(rt, mode, eps, fused, monkeypatch):
    """fused: BCE terms + the logit gradient's base from the score kernel's epilogue (rtk_score_packed_bce_f32 +
    rtk_bce_patch_pos_f32); not fused: scores, then rtk_bce_rows_f32 / rtk_bce_grad_f32 over them."""
    from r_tucker_amd import ops
    monkeypatch.setattr(ops, "FUSED_BCE", fused)
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


@pytest.mark.parametrize("scale", [1.0, 40.0])
def test_fused_loss_equals_the_three_pass_form(rt, scale, monkeypatch):
    """Same batch through both forms, ragged sizes (B = 70: last query tile of 6 rows; N = 3003), and with logits
    scaled until scores saturate to exactly 1.0f / 0.0f (zero logit gradient there, BCE log clamped at -100): the
    loss agrees to rounding, every gradient to the accuracy of the backward GEMMs."""
    from r_tucker_amd import ops
    # (entity rank 224 > 208: the three-pass form's scores come from the same split kernel as the fused form's,
    # so which entries saturate is decided by the same bits in both)
    n_ent, n_rel, B, rank = 3003, 7, 70, (3, 224, 224)
    core, R, S, O = gen.make_params(n_ent, n_rel, rank, 43)
    core = core * scale
    rng = np.random.default_rng(43)
    pairs = [(int(s), int(r)) for s, r in zip(rng.permutation(n_ent)[:120], rng.integers(0, n_rel, 120))]
    lists = [rng.integers(0, n_ent, rng.integers(1, 12)).tolist() for _ in pairs]
    ds = _Pairs(pairs, lists, n_ent, 0.1)
    ids = rng.permutation(120)[:B]
    h = torch.from_numpy(ds.features[ids, 0].copy()).cuda()
    r = torch.from_numpy(ds.features[ids, 1].copy()).cuda()
    flt = rt.DeviceFilter(ds, "cuda")
    out = {}
    for fused in (True, False):
        monkeypatch.setattr(ops, "FUSED_BCE", fused)
        ps = [torch.from_numpy(x).cuda().requires_grad_(True) for x in (core, R, S, O)]
        loss = rt.bce_loss_1vN(*ps, h, r, flt, torch.from_numpy(ids).cuda(), label_smoothing=0.1)
        (loss * 0.5).backward()
        out[fused] = (loss.item(), [p.grad.clone() for p in ps])
    if scale > 1:
        with torch.no_grad():
            P = rt.score_1vN(*[torch.from_numpy(x).cuda() for x in (core, R, S, O)], h, r)
        assert int((P == 1.0).sum()) > 0 and int((P == 0.0).sum()) >= 0        # the saturated branch is exercised
    assert abs(out[True][0] - out[False][0]) <= 2e-6 * max(1.0, abs(out[False][0]))
    for a, b in zip(out[True][1], out[False][1]):
        assert (a - b).abs().max().item() <= 1e-5 * b.abs().max().item() + 1e-12
