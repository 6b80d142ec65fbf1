#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE's own
scoring closure, ``filter_predictions`` and ``metrics`` on seeded inputs.

Run in the build container only (it needs /root/reference, which does not exist
on the GPU box):

    python tests/golden/make_golden.py [--reference /root/reference]

What is imported from the reference (read-only, no bytecode written):
  src/model/asymmetric/R_TuckER.py, src/model/symmetric/R_TuckER.py  (score_fn)
  src/utils/utils.py::filter_predictions, src/utils/metrics.py::metrics
  src/data/Data.py::Data, src/data/Dataset.py::KG_dataset             (WN18RR ids)

``tucker_riemopt`` (pinned 1.0.1, poetry.lock:235-236) is not installed and not
vendored; the scoring path only reads attributes of its container objects
(asymmetric/R_TuckER.py:43-47, symmetric/R_TuckER.py:40-44), so an in-memory
module holding two attribute-bag classes satisfies the ``from tucker_riemopt
import Tucker`` line.  No arithmetic of that package is involved in any vector
written here.

Outputs are DATA only: inputs are regenerated from seeds by tests/golden/gen.py
(sha256 of the inputs is stored and verified by the tests); outputs are the
reference's logits/probabilities/ranks.  No reference source text is stored.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen  # noqa: E402


def install_container_stub():
    m = types.ModuleType("tucker_riemopt")

    class Tucker:
        def __init__(self, core, factors):
            self.core, self.factors = core, factors

    class SFTucker:
        def __init__(self, core, regular_factors, num_shared_factors, shared_factor):
            self.core = core
            self.regular_factors = regular_factors
            self.num_shared_factors = num_shared_factors
            self.shared_factor = shared_factor

    m.Tucker, m.SFTucker = Tucker, SFTucker
    sys.modules["tucker_riemopt"] = m
    return Tucker, SFTucker


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    args = ap.parse_args()
    sys.path.insert(0, args.reference)
    Tucker, SFTucker = install_container_stub()

    import torch
    from src.model.asymmetric.R_TuckER import R_TuckER as RefAsym
    from src.model.symmetric.R_TuckER import R_TuckER as RefSym
    from src.utils.utils import filter_predictions
    from src.utils.metrics import metrics
    from src.data.Data import Data
    from src.data.Dataset import KG_dataset

    torch.manual_seed(0)
    meta = {"torch": torch.__version__, "numpy": np.__version__,
            "threads": torch.get_num_threads(), "cases": {}}

    def ref_score(mode, core, R, S, O, h, r, want_logits=True):
        """Run the reference closure.  Returns (logits, probabilities, model, T).
        The logits are the tensor the closure itself hands to ``torch.sigmoid``
        (asymmetric/R_TuckER.py:48), captured by wrapping ``torch.sigmoid`` for
        the duration of the call -- ``torch.logit(P)`` would be lossy."""
        n_ent, n_rel = S.shape[0], R.shape[0]
        rank = tuple(core.shape)
        Model = RefSym if mode == "sym" else RefAsym
        model = Model((n_ent, n_rel), rank)
        with torch.no_grad():
            model.core.copy_(torch.from_numpy(core))
            model.R.weight.copy_(torch.from_numpy(R))
            if mode == "sym":
                model.E.weight.copy_(torch.from_numpy(S))
                T = SFTucker(model.core.data, [model.R.weight], num_shared_factors=2,
                             shared_factor=model.E.weight)
            else:
                model.S.weight.copy_(torch.from_numpy(S))
                model.O.weight.copy_(torch.from_numpy(O))
                T = Tucker(model.core.data, [model.R.weight, model.S.weight, model.O.weight])
        score_fn = model(torch.from_numpy(h), torch.from_numpy(r))
        captured = {}
        real_sigmoid = torch.sigmoid

        def spy(x):
            captured["z"] = x.detach().clone()
            return real_sigmoid(x)

        torch.sigmoid = spy
        try:
            with torch.no_grad():
                P = score_fn(T)
        finally:
            torch.sigmoid = real_sigmoid
        return captured["z"].numpy(), P.numpy(), model, T

    # ---- (1) tiny exact cases + (5) gradient fixture ------------------------
    for mode in ("asym", "sym"):
        n_ent, n_rel, B, rank, seed = 50, 4, 7, (3, 5, 5), 0
        core, R, S, O = gen.make_params(n_ent, n_rel, rank, seed, shared=(mode == "sym"))
        h, r = gen.make_queries(n_ent, n_rel, B, seed)
        z, P, model, T = ref_score(mode, core, R, S, O, h, r)
        # gradient of sum(P*w) w.r.t. every operand, through the reference closure
        wrng = np.random.default_rng(seed + 7)
        w = wrng.standard_normal((B, n_ent)).astype(np.float32)
        score_fn = model(torch.from_numpy(h), torch.from_numpy(r))
        if mode == "sym":
            leaves = [model.core.detach().clone().requires_grad_(True),
                      model.R.weight.detach().clone().requires_grad_(True),
                      model.E.weight.detach().clone().requires_grad_(True)]
            Tg = SFTucker(leaves[0], [leaves[1]], num_shared_factors=2, shared_factor=leaves[2])
        else:
            leaves = [model.core.detach().clone().requires_grad_(True),
                      model.R.weight.detach().clone().requires_grad_(True),
                      model.S.weight.detach().clone().requires_grad_(True),
                      model.O.weight.detach().clone().requires_grad_(True)]
            Tg = Tucker(leaves[0], leaves[1:])
        (score_fn(Tg) * torch.from_numpy(w)).sum().backward()
        grads = {f"grad{i}": g.grad.numpy() for i, g in enumerate(leaves)}
        np.savez_compressed(os.path.join(HERE, f"tiny_{mode}.npz"),
                            logits=z, probs=P, w=w, h=h, r=r, **grads)
        meta["cases"][f"tiny_{mode}"] = dict(n_ent=n_ent, n_rel=n_rel, batch=B, rank=rank, seed=seed,
                                             inputs_sha256=gen.digest(core, R, S, O, h, r),
                                             state_dict_keys=list(model.state_dict().keys()))

    # ---- (2) medium, full tensors ------------------------------------------
    for mode in ("asym", "sym"):
        n_ent, n_rel, B, rank, seed = 2000, 22, 64, (10, 200, 200), 322
        core, R, S, O = gen.make_params(n_ent, n_rel, rank, seed, shared=(mode == "sym"))
        h, r = gen.make_queries(n_ent, n_rel, B, seed)
        z, P, _, _ = ref_score(mode, core, R, S, O, h, r)
        np.savez_compressed(os.path.join(HERE, f"medium_{mode}.npz"), logits=z, probs=P)
        meta["cases"][f"medium_{mode}"] = dict(n_ent=n_ent, n_rel=n_rel, batch=B, rank=rank, seed=seed,
                                               inputs_sha256=gen.digest(core, R, S, O, h, r))

    # ---- (3) WN18RR-shaped, model rank and the doubled (train-time) rank -----
    for name, rank in (("wn18rr_shape", (10, 200, 200)), ("wn18rr_shape_2x", (20, 400, 400))):
        n_ent, n_rel, B, seed = 40943, 22, 512, 322
        core, R, S, O = gen.make_params(n_ent, n_rel, rank, seed)
        h, r = gen.make_queries(n_ent, n_rel, B, seed)
        z, P, _, _ = ref_score("asym", core, R, S, O, h, r)
        flat_idx = np.arange(4096, dtype=np.int64) * ((B * n_ent) // 4096) + 17
        np.savez_compressed(os.path.join(HERE, f"{name}.npz"),
                            sample_idx=flat_idx,
                            logits_sample=z.reshape(-1)[flat_idx], probs_sample=P.reshape(-1)[flat_idx],
                            row_max=z.max(axis=1), row_argmax=z.argmax(axis=1).astype(np.int64),
                            row_sum_probs=P.astype(np.float64).sum(axis=1),
                            col_sum_probs=P.astype(np.float64).sum(axis=0))
        meta["cases"][name] = dict(n_ent=n_ent, n_rel=n_rel, batch=B, rank=rank, seed=seed,
                                   inputs_sha256=gen.digest(core, R, S, O, h, r))

    # ---- (4) b != c must raise ---------------------------------------------
    core, R, S, O = gen.make_params(20, 3, (3, 5, 7), 1)
    h, r = gen.make_queries(20, 3, 2, 1)
    try:
        ref_score("asym", core, R, S, O, h, r)
        raised = None
    except Exception as e:  # noqa: BLE001
        raised = type(e).__name__
    meta["cases"]["b_ne_c"] = dict(rank=(3, 5, 7), raises=raised)

    # ---- (6)/(7) filtered ranking on real WN18RR queries ---------------------
    data_dir = os.path.join(args.reference, "data", "WN18RR") + "/"
    data = Data(data_dir, reverse=True)
    n_ent, n_rel = len(data.entities), len(data.relations)
    splits = {}
    for split, triples in (("valid", data.valid_data), ("test", data.test_data)):
        ds = KG_dataset(data, triples, test_set=True)
        feats = np.asarray(ds.data_index, dtype=np.int64)
        splits[split] = (ds, feats)
    train_ds = KG_dataset(data, data.train_data, label_smoothing=0.1)
    meta["wn18rr"] = dict(n_ent=n_ent, n_rel=n_rel,
                          n_train_pairs=len(train_ds), n_valid=len(splits["valid"][0]),
                          n_test=len(splits["test"][0]),
                          entities_sha256=gen.digest(np.frombuffer("\n".join(data.entities).encode(), dtype=np.uint8)),
                          relations=data.relations)

    rank, seed = (10, 200, 200), 322
    rank_out = {}
    # "planted" = structured stand-in for a trained checkpoint (gen.make_planted_params):
    # all train triples + every second valid/test triple are planted.
    planted = np.concatenate([np.asarray(train_ds.data_index, dtype=np.int64),
                              splits["valid"][1][::2], splits["test"][1][::2]])
    meta["wn18rr"]["planted_sha256"] = gen.digest(planted)
    variants = (("spread", lambda: gen.make_params(n_ent, n_rel, rank, seed, logit_std=3.0)),
                ("saturated", lambda: gen.make_params(n_ent, n_rel, rank, seed, logit_std=24.0)),
                ("planted", lambda: gen.make_planted_params(planted, n_ent, n_rel, rank, seed, gain=8.0)),
                ("planted_sat", lambda: gen.make_planted_params(planted, n_ent, n_rel, rank, seed, gain=40.0)))
    for tag, make in variants:
        core, R, S, O = make()
        for split in ("valid", "test"):
            ds, feats = splits[split]
            sums = {"mrr": 0.0, "hits@1": 0.0, "hits@3": 0.0, "hits@10": 0.0}
            all_ranks = []
            n_sat = 0
            for lo in range(0, len(ds), 512):
                hi = min(lo + 512, len(ds))
                items = [ds[i] for i in range(lo, hi)]
                f = torch.stack([it[0] for it in items])
                t = torch.stack([it[1] for it in items])
                _, P, _, _ = ref_score("asym", core, R, S, O, f[:, 0].numpy(), f[:, 1].numpy())
                n_sat += int((P == 1.0).sum())
                P = torch.from_numpy(P)
                fp, ft = filter_predictions(P, t, f[:, 2].reshape(-1, 1))
                m = metrics(fp, ft)
                _, idx = torch.sort(fp, dim=1, descending=True)
                all_ranks.append((ft.gather(1, idx).argmax(dim=1) + 1).numpy())
                for k in sums:
                    sums[k] += float(m[k])
            rank_out[f"{tag}_{split}_ranks"] = np.concatenate(all_ranks).astype(np.int32)
            meta["cases"][f"rank_{tag}_{split}"] = dict(
                rank=rank, seed=seed, n=len(ds), sums=sums, scores_equal_to_one=n_sat,
                mrr=sums["mrr"] / len(ds), inputs_sha256=gen.digest(core, R, S, O))
    # the query ids themselves (so the product's own loader can be checked against them)
    rank_out["valid_features"] = splits["valid"][1].astype(np.int32)
    rank_out["test_features"] = splits["test"][1].astype(np.int32)
    # a small dense slice of targets to pin the filter vocabulary: first 64 test queries, CSR
    ds, feats = splits["test"]
    rows, cols = [], []
    for i in range(64):
        nz = torch.nonzero(ds[i][1]).reshape(-1).numpy()
        rows += [i] * len(nz)
        cols += nz.tolist()
    rank_out["test64_target_rows"] = np.asarray(rows, dtype=np.int32)
    rank_out["test64_target_cols"] = np.asarray(cols, dtype=np.int32)
    # label-smoothed train targets for the first 8 (s,r) pairs (Dataset.py:49-52)
    tr_feats = np.asarray([train_ds[i][0].numpy() for i in range(8)], dtype=np.int32)
    rank_out["train8_features"] = tr_feats
    rank_out["train8_target_sum"] = np.asarray([float(train_ds[i][1].double().sum()) for i in range(8)])
    rank_out["train8_target_max"] = np.asarray([float(train_ds[i][1].max()) for i in range(8)])
    rank_out["train8_target_min"] = np.asarray([float(train_ds[i][1].min()) for i in range(8)])
    np.savez_compressed(os.path.join(HERE, "wn18rr_rank.npz"), **rank_out)

    # ---- tie semantics of filter_predictions + metrics on an engineered case --
    P = torch.tensor([[0.9, 0.9, 0.9, 0.1, 1.0, 1.0],
                      [0.5, 0.5, 0.5, 0.5, 0.5, 0.5],
                      [1.0, 0.2, 1.0, 1.0, 0.0, 0.3]], dtype=torch.float32)
    t = torch.tensor([[0, 1, 1, 0, 0, 1],
                      [1, 0, 0, 0, 0, 1],
                      [0, 0, 1, 1, 0, 0]], dtype=torch.float32)
    o = torch.tensor([[2], [5], [3]])
    P_in, t_in = P.clone(), t.clone()
    fp, ft = filter_predictions(P, t, o)
    m = metrics(fp, ft)
    _, idx = torch.sort(fp, dim=1, descending=True)
    np.savez_compressed(os.path.join(HERE, "ties.npz"), P=P_in.numpy(), t=t_in.numpy(), o=o.numpy(),
                        fp=fp.numpy(), ft=ft.numpy(), ranks=(ft.gather(1, idx).argmax(dim=1) + 1).numpy(),
                        sums=np.asarray([float(m[k]) for k in ("mrr", "hits@1", "hits@3", "hits@10")]))

    with open(os.path.join(HERE, "meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("wrote fixtures to", HERE)


if __name__ == "__main__":
    main()
