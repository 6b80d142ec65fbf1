"""The oracle (oracle/score_oracle.py) against the golden vectors produced by the
reference's own score_fn / filter_predictions / metrics (tests/golden/make_golden.py).
CPU only.  This is what "parity pinned" rests on."""
import numpy as np
import pytest
import torch

import gen
from oracle import score_oracle as orc


def _params(meta_case, shared=False, **kw):
    c = meta_case
    core, R, S, O = gen.make_params(c["n_ent"], c["n_rel"], tuple(c["rank"]), c["seed"], shared=shared, **kw)
    h, r = gen.make_queries(c["n_ent"], c["n_rel"], c["batch"], c["seed"])
    assert gen.digest(core, R, S, O, h, r) == c["inputs_sha256"], "seeded inputs differ from the fixture's"
    t = [torch.from_numpy(x) for x in (core, R, S, O)]
    return t, torch.from_numpy(h), torch.from_numpy(r)


@pytest.mark.parametrize("mode", ["asym", "sym"])
@pytest.mark.parametrize("size", ["tiny", "medium"])
def test_scores_match_reference_bitwise_or_close(golden, golden_meta, mode, size):
    name = f"{size}_{mode}"
    (core, R, S, O), h, r = _params(golden_meta["cases"][name], shared=(mode == "sym"))
    g = golden(name)
    z = orc.logits_ref(core, R, S, O, h, r).numpy()
    p = orc.score_ref(core, R, S, O, h, r).numpy()
    # same op sequence, same library, same thread count -> expected identical; allow
    # a few ulp in case the BLAS picks another blocking on another host.
    np.testing.assert_allclose(z, g["logits"], rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(p, g["probs"], rtol=0, atol=5e-7)
    # and the float64 restatement bounds both
    ze = orc.logits_exact(core, R, S, O, h, r)
    assert np.max(np.abs(ze - g["logits"]) / (1 + np.abs(ze))) < 2e-5
    assert np.max(np.abs(orc.score_exact(core, R, S, O, h, r) - g["probs"])) < 5e-6


@pytest.mark.parametrize("mode", ["asym", "sym"])
def test_container_entry_point(golden, golden_meta, mode):
    (core, R, S, O), h, r = _params(golden_meta["cases"][f"tiny_{mode}"], shared=(mode == "sym"))
    T = orc.SFTuckerBag(core, [R], 2, S) if mode == "sym" else orc.TuckerBag(core, [R, S, O])
    np.testing.assert_allclose(orc.score_fn_ref(T, h, r).numpy(), golden(f"tiny_{mode}")["probs"], atol=5e-7)


@pytest.mark.parametrize("mode", ["asym", "sym"])
def test_gradients_match_reference(golden, golden_meta, mode):
    (core, R, S, O), h, r = _params(golden_meta["cases"][f"tiny_{mode}"], shared=(mode == "sym"))
    g = golden(f"tiny_{mode}")
    grads = orc.score_grads_ref(core, R, S, O, h, r, torch.from_numpy(g["w"]), shared=(mode == "sym"))
    for i, gr in enumerate(grads):
        np.testing.assert_allclose(gr.numpy(), g[f"grad{i}"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("name", ["wn18rr_shape", "wn18rr_shape_2x"])
def test_full_size_samples(golden, golden_meta, name):
    (core, R, S, O), h, r = _params(golden_meta["cases"][name])
    g = golden(name)
    z = orc.logits_ref(core, R, S, O, h, r)
    p = torch.sigmoid(z).numpy()
    z = z.numpy()
    np.testing.assert_allclose(z.reshape(-1)[g["sample_idx"]], g["logits_sample"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(p.reshape(-1)[g["sample_idx"]], g["probs_sample"], atol=2e-6)
    np.testing.assert_allclose(z.max(axis=1), g["row_max"], rtol=1e-5, atol=1e-5)
    assert (z.argmax(axis=1) == g["row_argmax"]).mean() > 0.99
    np.testing.assert_allclose(p.astype(np.float64).sum(axis=1), g["row_sum_probs"], rtol=1e-6)
    np.testing.assert_allclose(p.astype(np.float64).sum(axis=0), g["col_sum_probs"], rtol=1e-5, atol=1e-5)


def test_b_ne_c_raises_like_reference(golden_meta):
    assert golden_meta["cases"]["b_ne_c"]["raises"] == "RuntimeError"
    core, R, S, O = [torch.from_numpy(x) for x in gen.make_params(20, 3, (3, 5, 7), 1)]
    h, r = [torch.from_numpy(x) for x in gen.make_queries(20, 3, 2, 1)]
    with pytest.raises(RuntimeError):
        orc.score_ref(core, R, S, O, h, r)


def test_filter_and_rank_tie_semantics(golden):
    g = golden("ties")
    P, t, o = torch.from_numpy(g["P"]), torch.from_numpy(g["t"]), torch.from_numpy(g["o"])
    fp, ft = orc.filter_predictions_ref(P.clone(), t.clone(), o)
    np.testing.assert_array_equal(fp.numpy(), g["fp"])
    np.testing.assert_array_equal(ft.numpy(), g["ft"])
    ranks, m = orc.filter_and_rank(P, t, o.reshape(-1))
    np.testing.assert_array_equal(ranks.numpy(), g["ranks"])
    np.testing.assert_allclose([float(m[k]) for k in ("mrr", "hits@1", "hits@3", "hits@10")], g["sums"])
    # filter_and_rank works on copies; the reference mutates (utils.py:19-21)
    np.testing.assert_array_equal(P.numpy(), g["P"])
