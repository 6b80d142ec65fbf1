"""Host data path (r-tucker_amd/data.py) against ids/targets captured from the
reference's Data + KG_dataset (tests/golden/wn18rr_rank.npz, meta.json).  CPU only."""
import os

import numpy as np
import pytest
import torch

import gen
from r_tucker_amd.data import Data, KG_dataset

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def wn():
    data = Data(os.path.join(ROOT, "data", "WN18RR") + "/", reverse=True)
    return data


def test_vocabulary_matches_reference(wn, golden_meta):
    m = golden_meta["wn18rr"]
    assert len(wn.entities) == m["n_ent"] == 40943
    assert wn.relations == m["relations"] and len(wn.relations) == 22
    assert gen.digest(np.frombuffer("\n".join(wn.entities).encode(), dtype=np.uint8)) == m["entities_sha256"]


def test_query_ids_and_filter_targets_match_reference(wn, golden, golden_meta):
    g = golden("wn18rr_rank")
    m = golden_meta["wn18rr"]
    valid = KG_dataset(wn, wn.valid_data, test_set=True)
    test = KG_dataset(wn, wn.test_data, test_set=True)
    train = KG_dataset(wn, wn.train_data, label_smoothing=0.1)
    assert (len(train), len(valid), len(test)) == (m["n_train_pairs"], m["n_valid"], m["n_test"])
    np.testing.assert_array_equal(valid.features, g["valid_features"])
    np.testing.assert_array_equal(test.features, g["test_features"])
    t = test.dense_targets(np.arange(64))
    rows, cols = torch.nonzero(t, as_tuple=True)
    np.testing.assert_array_equal(rows.numpy(), g["test64_target_rows"])
    np.testing.assert_array_equal(cols.numpy(), g["test64_target_cols"])
    f0, t0 = test[5]
    np.testing.assert_array_equal(f0.numpy(), g["test_features"][5])
    np.testing.assert_array_equal(t0.numpy(), t[5].numpy())
    # train items: (s, r) pairs in first-appearance order, label-smoothed targets (Dataset.py:49-52)
    np.testing.assert_array_equal(train.features[:8], g["train8_features"])
    tt = train.dense_targets(np.arange(8))
    np.testing.assert_allclose(tt.double().sum(1).numpy(), g["train8_target_sum"], rtol=1e-6)
    np.testing.assert_allclose(tt.max(1).values.numpy(), g["train8_target_max"], rtol=1e-6)
    np.testing.assert_allclose(tt.min(1).values.numpy(), g["train8_target_min"], rtol=1e-6)
    # the planted-triple list used by the ranking fixtures is reproducible from this loader
    planted = np.concatenate([np.asarray(train.data_index, dtype=np.int64), valid.features[::2], test.features[::2]])
    assert gen.digest(planted) == m["planted_sha256"]
