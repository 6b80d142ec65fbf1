"""Deterministic synthetic parameters / queries shared by the golden-vector
generator, the parity tests, ``smoke()`` and ``bench.py``.

numpy's PCG64 ``Generator`` is used (not torch's) so the same seed gives the
same bytes wherever this image runs; every fixture also stores a sha256 of the
inputs it was computed from and the tests verify it before comparing outputs.

Scaling: factor entries ~ N(0,1); core entries ~ N(0, s^2)
with s = logit_std / sqrt(a*b*c), so that logits z = sum G.R.S.O have standard
deviation ~ logit_std and sigmoid(z) spreads over (0,1).  (The reference's own
``init()`` -- ``src/model/asymmetric/R_TuckER.py:27-39`` -- gives scores
0.5 +- 4e-5, useless as a parity vector.)
"""
from __future__ import annotations

import hashlib

import numpy as np


def make_params(n_ent, n_rel, rank, seed, shared=False, logit_std=3.0):
    a, b, c = rank
    rng = np.random.default_rng(seed)
    core = (rng.standard_normal((a, b, c)) * (logit_std / np.sqrt(a * b * c))).astype(np.float32)
    R = rng.standard_normal((n_rel, a)).astype(np.float32)
    S = rng.standard_normal((n_ent, b)).astype(np.float32)
    O = S if shared else rng.standard_normal((n_ent, c)).astype(np.float32)
    return core, R, S, O


def make_queries(n_ent, n_rel, batch, seed):
    rng = np.random.default_rng(seed + 1_000_003)
    h = rng.integers(0, n_ent, size=batch, dtype=np.int64)
    r = rng.integers(0, n_rel, size=batch, dtype=np.int64)
    return h, r


def digest(*arrays):
    m = hashlib.sha256()
    for x in arrays:
        x = np.ascontiguousarray(x)
        m.update(str(x.dtype).encode())
        m.update(str(x.shape).encode())
        m.update(x.tobytes())
    return m.hexdigest()


def make_planted_params(triples, n_ent, n_rel, rank, seed, gain=8.0):
    """Parameters with *structure*: a stand-in for a trained checkpoint (the
    reference ships none, ``.gitignore:7``), so that filtered ranks are small and
    MRR is O(0.1-1) instead of the ~1/N of random parameters.

    core, R, O are random as in ``make_params``; each subject row is then set to
    ``S[s] = beta * sum_i O[o_i] . W_{r_i}^T`` over the planted triples ``(s, r_i,
    o_i)``, where ``W_r = G x_0 R[r]`` -- so ``S[s] . W_r . O[o]^T`` is large exactly
    for planted ``(s, r, o)``.  ``beta`` is chosen so a singly-planted triple gets a
    logit of about ``gain``.  Pure numpy, float64 accumulation, deterministic.
    """
    a, b, c = rank
    core, R, _, O = make_params(n_ent, n_rel, rank, seed)
    W = np.einsum("abc,ra->rbc", core.astype(np.float64), R.astype(np.float64))  # (nR, b, c)
    t = np.asarray(triples, dtype=np.int64)
    S = np.zeros((n_ent, b), dtype=np.float64)
    O64 = O.astype(np.float64)
    for r in range(n_rel):
        sel = t[t[:, 1] == r]
        if len(sel) == 0:
            continue
        contrib = O64[sel[:, 2]] @ W[r].T            # (n_r, b)
        np.add.at(S, sel[:, 0], contrib)
    # expected logit of a singly planted triple: O[o] W^T W O[o]^T ~ c * b * var(W)
    signal = c * b * float(np.mean(W * W))
    S *= gain / signal
    return core, R, S.astype(np.float32), O
