#!/usr/bin/env python3
"""Gradient fixture at the TRAINING shape of WN18RR: the Riemannian gradient
differentiates loss_fn at the doubled-rank construct, i.e. core (20,400,400), factors n x 400 / n x 20,
40 943 entities, one batch of (s, r) pairs (train.py:78-82).

    python tests/golden/make_golden_train_grad.py [--reference /root/reference]

Runs the REFERENCE's own closure (src/model/asymmetric/R_TuckER.py:41-50) under torch autograd with
``nn.BCELoss(reduction="mean")`` (train.py:136) against label-smoothed multi-hot targets built the way
``src/data/Dataset.py:43-53`` builds them, and stores DATA only: the loss, the full relation-factor
gradient, strided samples and float64 column sums of the large gradients.  Inputs are regenerated from
seeds by ``r_tucker_amd.synthetic``; the batch is ragged (B = 500, not a multiple of 32) and contains
repeated subjects and relations, so the row scatter of the backward is exercised.
Build container only (needs /root/reference).
"""
from __future__ import annotations

import argparse
import json
import os
import sys

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen  # noqa: E402
from make_golden import install_container_stub  # noqa: E402

N_ENT, N_REL, B, RANK, SEED, EPS = 40943, 22, 500, (20, 400, 400), 2026, 0.1


def make_batch():
    """(h, r, lists): queries with repeated subjects / relations and 1-6 known objects each."""
    rng = np.random.default_rng(SEED + 5)
    h = rng.integers(0, N_ENT, size=B, dtype=np.int64)
    h[1::7] = h[0]                       # one subject many times
    h[3::11] = h[2]
    r = rng.integers(0, N_REL, size=B, dtype=np.int64)
    lists = [sorted(set(rng.integers(0, N_ENT, rng.integers(1, 7)).tolist())) for _ in range(B)]
    return h, r, lists


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    args = ap.parse_args()
    sys.path.insert(0, args.reference)
    Tucker, _ = install_container_stub()
    import torch
    from src.model.asymmetric.R_TuckER import R_TuckER as RefAsym

    core, R, S, O = gen.make_params(N_ENT, N_REL, RANK, SEED)
    h, r, lists = make_batch()
    targets = torch.zeros((B, N_ENT), dtype=torch.float32)
    for d, l in enumerate(lists):
        targets[d, l] = 1.0
    targets = (1.0 - EPS) * targets + (1.0 / targets.shape[1]) * EPS      # Dataset.py:51-52 (vec * (1-eps) + eps / N)
    model = RefAsym((N_ENT, N_REL), RANK)
    leaves = [torch.from_numpy(x).clone().requires_grad_(True) for x in (core, R, S, O)]
    T = Tucker(leaves[0], leaves[1:])
    loss = torch.nn.BCELoss(reduction="mean")(model(torch.from_numpy(h), torch.from_numpy(r))(T), targets)
    loss.backward()
    g_core, g_R, g_S, g_O = [x.grad.numpy() for x in leaves]
    idx_core = np.arange(4096, dtype=np.int64) * (g_core.size // 4096) + 5
    idx_O = np.arange(4096, dtype=np.int64) * (g_O.size // 4096) + 11
    rows_S = np.unique(h)[:: max(1, len(np.unique(h)) // 24)][:24]
    rows_S = np.unique(np.concatenate([rows_S, [h[0], h[2]]]))            # the repeated subjects
    np.savez_compressed(os.path.join(HERE, "wn18rr_train_grad.npz"),
                        loss=np.float64(loss.item()), g_R=g_R,
                        core_idx=idx_core, g_core_sample=g_core.reshape(-1)[idx_core],
                        g_core_colsum=g_core.astype(np.float64).sum(axis=(0, 1)),
                        O_idx=idx_O, g_O_sample=g_O.reshape(-1)[idx_O],
                        g_O_colsum=g_O.astype(np.float64).sum(axis=0),
                        S_rows=rows_S, g_S_rows=g_S[rows_S],
                        g_S_colsum=g_S.astype(np.float64).sum(axis=0),
                        g_S_nonzero_rows=np.int64((np.abs(g_S).sum(axis=1) > 0).sum()))
    with open(os.path.join(HERE, "meta.json")) as f:
        meta = json.load(f)
    meta["cases"]["wn18rr_train_grad"] = dict(n_ent=N_ENT, n_rel=N_REL, batch=B, rank=RANK, seed=SEED, eps=EPS,
                                             inputs_sha256=gen.digest(core, R, S, O, h, r),
                                             lists_sha256=gen.digest(np.asarray([x for l in lists for x in l], dtype=np.int64)),
                                             torch=torch.__version__)
    with open(os.path.join(HERE, "meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("wrote wn18rr_train_grad.npz; loss", loss.item())


if __name__ == "__main__":
    main()
