"""The committed trained-model fixture (tests/golden/wn18rr_trained_q8.npz, written by tools/pack_checkpoint_q8.py): what
tests/test_gpu_trained_checkpoint.py loads on a clean clone.  CPU checks of the container itself."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
PATH = os.path.join(ROOT, "tests", "golden", "wn18rr_trained_q8.npz")


def test_fixture_is_small_and_complete():
    assert os.path.getsize(PATH) <= 20 * 1024 * 1024          # a <= 20 MB trained model in the tree
    z = np.load(PATH, allow_pickle=False)
    assert z["core"].shape == (10, 200, 200) and z["core"].dtype == np.float32
    assert z["R"].shape == (22, 10)
    for n in ("S", "O"):
        assert z[n + "_q8"].shape == (40943, 200) and z[n + "_q8"].dtype == np.int8
        assert z[n + "_scale"].shape == (40943,) and np.all(z[n + "_scale"] > 0)
    assert int(z["epoch"]) == 500
    assert np.isfinite(z["core"]).all() and np.linalg.norm(z["core"]) > 1e4      # a TRAINED core, not the initial one


def test_quantisation_round_trip_and_conditioning():
    from pack_checkpoint_q8 import dequantise, quantise
    z = np.load(PATH, allow_pickle=False)
    S = dequantise(z["S_q8"], z["S_scale"])
    q, s = quantise(S)
    assert np.array_equal(q, z["S_q8"]) and np.allclose(s, z["S_scale"], rtol=1e-6)      # a fixed point of the packer
    # the factors came from orthonormal columns: after the int8 rounding the Gram matrix is still well conditioned
    # (the test re-orthonormalises them by QR before use)
    g = S.astype(np.float64).T @ S.astype(np.float64)
    w = np.linalg.eigvalsh(g)
    assert w[0] > 0.5 and w[-1] < 2.0
