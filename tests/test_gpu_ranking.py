"""WN18RR filtered-ranking / MRR parity: scores from the HIP path, fed through the
oracle's restatement of filter_predictions + metrics, against per-query ranks and
metric sums the REFERENCE produced from its own scores (tests/golden/wn18rr_rank.npz).

Bar (north_star): MRR within +-0.001 of the reference.  Per-query ranks are also
compared: fp32 sigmoid outputs tie massively near 1.0 and ties are broken by sort
order, so a 1-ulp difference can move a rank, and torch's default CPU sort is unstable (its tie
order varies with the host's thread count); we require >= 99 % of ranks identical (>= 97 % on
the deliberately saturated fixtures) and report the rest.
"""
import os

import numpy as np
import pytest
import torch

import gen
from oracle import score_oracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def wn():
    from r_tucker_amd.data import Data, KG_dataset
    data = Data(os.path.join(ROOT, "data", "WN18RR") + "/", reverse=True)
    ds = {"valid": KG_dataset(data, data.valid_data, test_set=True),
          "test": KG_dataset(data, data.test_data, test_set=True),
          "train": KG_dataset(data, data.train_data, label_smoothing=0.1)}
    return data, ds


@pytest.mark.parametrize("sigmoid_mode", ["fast", "exact"])
@pytest.mark.parametrize("variant", ["spread", "saturated", "planted", "planted_sat"])
def test_filtered_mrr_matches_reference(wn, golden, golden_meta, variant, sigmoid_mode, monkeypatch):
    import r_tucker_amd as rt
    from r_tucker_amd import ops
    monkeypatch.setattr(ops, "DEFAULT_SIGMOID", sigmoid_mode)
    data, ds = wn
    n_ent, n_rel, rank, seed = len(data.entities), len(data.relations), (10, 200, 200), 322
    if variant.startswith("planted"):
        planted = np.concatenate([np.asarray(ds["train"].data_index, dtype=np.int64),
                                  ds["valid"].features[::2], ds["test"].features[::2]])
        params = gen.make_planted_params(planted, n_ent, n_rel, rank, seed, gain=8.0 if variant == "planted" else 40.0)
    else:
        params = gen.make_params(n_ent, n_rel, rank, seed, logit_std=3.0 if variant == "spread" else 24.0)
    model = rt.AsymmetricR_TuckER((n_ent, n_rel), rank)
    model.init({"core": torch.from_numpy(params[0]), "R.weight": torch.from_numpy(params[1]),
                "S.weight": torch.from_numpy(params[2]), "O.weight": torch.from_numpy(params[3])})
    model.cuda().eval()
    T = rt.Tucker(model.core.data, [model.R.weight, model.S.weight, model.O.weight])
    g = golden("wn18rr_rank")
    for split in ("valid", "test"):
        case = golden_meta["cases"][f"rank_{variant}_{split}"]
        assert gen.digest(*params) == case["inputs_sha256"]
        d = ds[split]
        sums = {"mrr": 0.0, "hits@1": 0.0, "hits@3": 0.0, "hits@10": 0.0}
        ranks = []
        for lo in range(0, len(d), 512):
            ids = np.arange(lo, min(lo + 512, len(d)))
            f = torch.from_numpy(d.features[ids]).cuda()
            with torch.no_grad():
                P = model(f[:, 0], f[:, 1])(T)                 # HIP path
            rk, m = orc.filter_and_rank(P.cpu(), d.dense_targets(ids), f[:, 2].cpu())   # checker
            ranks.append(rk.numpy())
            for k in sums:
                sums[k] += float(m[k])
        ranks = np.concatenate(ranks)
        same = float((ranks == g[f"{variant}_{split}_ranks"]).mean())
        mrr, ref_mrr = sums["mrr"] / len(d), case["mrr"]
        print(f"\n{variant}/{split} [{sigmoid_mode}]: MRR hip {mrr:.6f} ref {ref_mrr:.6f}  identical ranks {same:.4%}  "
              f"hits@1 {sums['hits@1']:.0f}/{case['sums']['hits@1']:.0f}")
        assert abs(mrr - ref_mrr) <= 1e-3
        # exactly tied scores (fp32 sigmoid saturates at 1.0) are ordered by torch's UNSTABLE CPU sort: the tie order --
        # hence the rank of a tied target -- depends on the host (thread count), both in the golden run and here
        assert same >= (0.97 if "sat" in variant else 0.99)
        for k in ("hits@1", "hits@3", "hits@10"):
            assert abs(sums[k] - case["sums"][k]) <= 0.002 * len(d) + 1
