"""Tucker / shared-factor Tucker containers (the subset of ``tucker_riemopt.Tucker`` /
``tucker_riemopt.SFTucker`` the reference touches; that package is pinned at 1.0.1 in the reference's
``poetry.lock`` but is neither vendored nor installable offline, so this is our own implementation of
the textbook operations -- SURVEY.md Appendix C; parity with the package is UNPINNED, the algebra is
validated by identities in ``tests/test_riemannian.py``).

On the scoring path they are attribute holders: ``score_fn`` reads ``T.core``, ``T.factors`` /
``T.regular_factors``, ``T.shared_factor`` (``src/model/asymmetric/R_TuckER.py:43-47``,
``src/model/symmetric/R_TuckER.py:40-44``); constructor signatures are the ones ``extract_tensor``
uses (``train.py:39,41``).  Beyond that:

* ``norm()``  -- Frobenius norm of the represented tensor, differentiable (``train.py:79``:
  ``regularization_coeff * T.norm() ** 2`` inside the differentiated ``loss_fn``), computed through the
  factor Gram matrices (cost O(sum n_i r_i^2), never the dense tensor);
* ``round(rank)`` -- retraction by truncated HOSVD: thin QR of every factor, the R factors absorbed
  into the core, truncated SVD of every core unfolding (``asymmetric/optim.py:108``,
  ``symmetric/optim.py:102``); for the shared-factor tensor the shared modes get ONE common factor;
* ``full()`` -- the dense tensor (tests, tiny sizes only).

Everything is plain torch on whatever device the operands live on (QR / SVD of small matrices: the
reference does the same through tucker_riemopt's torch backend); the scoring arithmetic is NOT here.
"""
from __future__ import annotations

from typing import Sequence

import torch

from . import smalllinalg as _sl
from .smalllinalg import dominant_left_subspace, gram_factor, gram_factor_many


def _tn(A: torch.Tensor, B: torch.Tensor) -> torch.Tensor:
    """``A.T @ B`` -- the Gram-type product (tall-skinny on the GPU: split-K HIP GEMM, ``ops.gram_tn``)."""
    from .ops import gram_tn
    return gram_tn(A, B)


def _mode_dot(core: torch.Tensor, mat: torch.Tensor, mode: int) -> torch.Tensor:
    """core x_mode mat, mat of shape (new, old): contracts axis ``mode`` of the core with mat's columns."""
    return torch.movedim(torch.tensordot(mat, core, dims=([1], [mode])), 0, mode)


def _unfold(core: torch.Tensor, mode: int) -> torch.Tensor:
    return torch.movedim(core, mode, 0).reshape(core.shape[mode], -1)


def _gram_norm(core: torch.Tensor, grams: Sequence[torch.Tensor]) -> torch.Tensor:
    """||core x_0 U_0 x_1 U_1 x_2 U_2||_F from the Gram matrices U_i^T U_i:  <core x_i gram_i, core>^(1/2)."""
    t = core
    for i, g in enumerate(grams):
        t = _mode_dot(t, g, i)
    return torch.sqrt(torch.clamp((t * core).sum(), min=0.0))


# Device-side health of the retraction, read (one sync) and cleared by the driver once per epoch: the largest
# deviation from orthonormality of a new factor BEFORE its final polish, and whether anything was non-finite.
# Nothing in a step branches on them: a step has no host synchronisation (tools/graphstep.py can capture it).
HEALTH = {}


def _note_health(err: torch.Tensor) -> None:
    """Fold ``err`` into the device's persistent health word IN PLACE (a replayed HIP graph keeps accumulating)."""
    key = err.device
    e = torch.nan_to_num(err.detach().double(), nan=float("inf")).reshape(())
    cur = HEALTH.get(key)
    if cur is None:
        if err.is_cuda and torch.cuda.is_current_stream_capturing():
            raise RuntimeError("tucker.HEALTH must exist before graph capture (run one eager step first)")
        HEALTH[key] = cur = torch.zeros((), dtype=torch.float64, device=key)
    cur.copy_(torch.maximum(cur, e))


def read_health(device=None, clear: bool = True):
    """Largest pre-polish orthonormality error since the last call (``inf``: something was not finite); one sync."""
    out = {}
    for k, v in HEALTH.items():
        if device is None or (k.type == torch.device(device).type and torch.device(device).index in (None, k.index)):
            out[str(k)] = float(v)
            if clear:
                v.zero_()
    return out


_CHECK = __import__("os").environ.get("R_TUCKER_AMD_CHECK_FINITE", "0") == "1"


def _dbg(what: str, t: torch.Tensor) -> None:
    """``R_TUCKER_AMD_CHECK_FINITE=1`` (one sync per call): name the first non-finite intermediate of the retraction."""
    if _CHECK and not bool(torch.isfinite(t).all()):
        bad = (~torch.isfinite(t)).nonzero()
        raise FloatingPointError(f"retraction: {what} of shape {tuple(t.shape)} has {bad.shape[0]} non-finite entries, "
                                 f"first at {bad[0].tolist()}; finite abs max {t[torch.isfinite(t)].abs().max().item() if bad.shape[0] < t.numel() else 'n/a'}")


def _orth_tall_many(Ds):
    """``[(Q, R), ...]`` with ``D = Q R``, Q orthonormal columns, R (k x k, float64) upper triangular, for tall-skinny
    blocks by two rounds of Cholesky QR: the Gram matrices are accumulated in float64 (a batched product over row
    chunks), the k x k factors and their inverses come from ``smalllinalg.gram_factor_many`` (float64, equilibrated,
    shifted: no failure path, no host synchronisation; blocks of one width share a launch).  rocSOLVER's Householder
    ``geqrf`` spends 90 ms on a 40 943 x 400 factor; this is four chip-filling GEMMs per block.  A column of zeros
    stays zero (its row of R is zero)."""
    f32 = Ds[0].dtype == torch.float32
    # (float64 accumulation: the block may be rank deficient to fp32 precision, and its Gram matrix must stay positive
    # semi-definite to well below the shift -- smalllinalg.tall_gram_f64)
    S1 = [_sl.tall_gram_f64(D) for D in Ds]
    for S in S1:
        _dbg("Gram matrix of the new factor block", S)
    XR1 = gram_factor_many(S1, shift=3e-6 if f32 else None)          # fp32 block: |X| <= 600 / column norm, Q = D X in fp32
    Qs = [D @ X1.to(D.dtype) for D, (X1, _) in zip(Ds, XR1)]
    for D, S, (X1, _), Q in zip(Ds, S1, XR1, Qs):
        if _CHECK and not bool(torch.isfinite(Q).all()):
            cn = torch.linalg.vector_norm(D.double(), dim=0)
            raise FloatingPointError(
                f"retraction: first Cholesky-QR round of a {tuple(D.shape)} block: Q not finite.  column norms of the block: "
                f"min {cn.min().item():.3e} max {cn.max().item():.3e} last four {cn[-4:].tolist()}; diag of its Gram matrix: min "
                f"{S.diagonal().min().item():.3e} max {S.diagonal().max().item():.3e}; |X| max {X1.abs().max().item():.3e} at "
                f"{divmod(int(X1.abs().argmax()), X1.shape[1])}; diag X last four {X1.diagonal()[-4:].tolist()}; "
                f"non-finite in D {int((~torch.isfinite(D)).sum())}, in S {int((~torch.isfinite(S)).sum())}, in X {int((~torch.isfinite(X1)).sum())}")
    XR2 = gram_factor_many([_sl.tall_gram_f64(Q) for Q in Qs], shift=1e-6 if f32 else None)
    return [(Q @ X2.to(Q.dtype), R2 @ R1) for Q, (_, R1), (X2, R2) in zip(Qs, XR1, XR2)]


def _orth_tall(D: torch.Tensor):
    return _orth_tall_many([D])[0]


def _round_tangent_step(core: torch.Tensor, pairs, mode_factor, ranks):
    """Truncated HOSVD of ``core x_m [U_f(m), D_f(m)]`` where every ``U`` has orthonormal columns and
    ``U^T D = 0`` up to rounding -- the shape ``TangentVector.construct()`` produces (``pairs[f] = (U, D)``,
    ``mode_factor[m]`` = which pair serves core axis m: (0,1,2) for Tucker, (0,1,1) for the shared-factor tensor;
    ``ranks[f]`` the target rank of that factor).  Returns ``(new core, [new factor per pair])``.

    1. ``D = Q R_d`` (thin QR of the NEW block only, Cholesky QR), so ``[U, D] = [U, Q] [[I, 0], [0, R_d]]`` (the gauge
       condition makes ``[U, Q]`` orthonormal; what rounding leaves of ``U^T D`` is removed by step 4);
    2. the 2r x 2r triangular blocks are absorbed into the (small) core, in float64;
    3. per factor: the leading left singular subspace ``W`` of the core unfolding(s) -- for a shared factor, of
       the concatenation of its modes' unfoldings: the one basis that serves both best in the least-squares sense
       -- by warm-started subspace iteration (``smalllinalg.dominant_left_subspace``); core and factor are
       truncated at once (sequentially truncated HOSVD);
    4. the new factor ``[U, Q] W`` is re-orthonormalised by one more Cholesky QR round so drift cannot add up
       over the steps, its triangular factor absorbed into the core."""
    dt = core.dtype
    blocks, bases = [], []
    for (U, D), (Q, Rd) in zip(pairs, _orth_tall_many([D for _, D in pairs])):   # U^T D = 0 (gauge): [U, Q] is orthonormal
        r = U.shape[1]
        k = r + D.shape[1]
        blk = torch.zeros((k, k), dtype=torch.float64, device=core.device)
        blk[:r, :r] = torch.eye(r, dtype=torch.float64, device=core.device)
        blk[r:, r:] = Rd
        blocks.append(blk)
        bases.append((U, Q))
    c64 = core.double()
    for m, f in enumerate(mode_factor):
        c64 = _mode_dot(c64, blocks[f], m)
    modes_of = [[m for m, g in enumerate(mode_factor) if g == f] for f in range(len(bases))]
    # Factors are truncated in order (sequentially truncated HOSVD), except that consecutive factors which serve ONE
    # mode each and have the same shapes (the subject and object factors of the asymmetric model) take their bases from
    # the same core state and go through the iteration, and the polish, as one batch (plain HOSVD among themselves:
    # the same quasi-optimality bound, half the serial factorisations).
    groups, f = [], 0
    while f < len(bases):
        g = [f]
        while (_sl.BATCH_EQUAL_SIZES and g[-1] + 1 < len(bases) and len(modes_of[f]) == 1 and len(modes_of[g[-1] + 1]) == 1
               and bases[g[-1] + 1][0].shape == bases[f][0].shape and bases[g[-1] + 1][1].shape == bases[f][1].shape
               and int(ranks[g[-1] + 1]) == int(ranks[f]) and c64.shape[modes_of[f][0]] == c64.shape[modes_of[g[-1] + 1][0]]):
            g.append(g[-1] + 1)
        groups.append(g)
        f = g[-1] + 1
    new_factors = [None] * len(bases)
    for g in groups:
        if len(g) == 1:
            modes = modes_of[g[0]]
            M = torch.cat([_unfold(c64, m) for m in modes], dim=1) if len(modes) > 1 else _unfold(c64, modes[0])
            _dbg(f"core unfolding of factor {g[0]}", M)
            Ws = [dominant_left_subspace(M, int(ranks[g[0]]))]
        else:
            M = torch.stack([_unfold(c64, modes_of[f][0]) for f in g])
            _dbg(f"core unfoldings of factors {g}", M)
            Ws = list(dominant_left_subspace(M, int(ranks[g[0]])))
        news = []
        for f, W in zip(g, Ws):
            _dbg(f"subspace basis of factor {f}", W)
            U, Q = bases[f]
            r = U.shape[1]
            Wd = W.to(dt)
            news.append(U @ Wd[:r] + Q @ Wd[r:])
        Ss = [_tn(nu, nu) for nu in news]
        for S in Ss:
            _note_health((S - torch.eye(S.shape[0], dtype=S.dtype, device=S.device)).abs().max())
        for f, W, nu, (X, Rfix) in zip(g, Ws, news, gram_factor_many(Ss, shift=0.0 if dt == torch.float64 else 1e-7)):
            _dbg(f"polish of factor {f}: X", X)
            new_factors[f] = nu @ X.to(dt)
            T = Rfix @ W.transpose(0, 1)                 # (r_new x 2r): truncate, then the polish's triangular factor
            for m in modes_of[f]:
                c64 = _mode_dot(c64, T, m)
    out = c64.to(dt)
    inf = torch.full((), float("inf"), dtype=torch.float64, device=out.device)
    _note_health(torch.where(torch.isfinite(out).all(), torch.zeros_like(inf), inf))
    return out, new_factors


def _truncated_left_basis(mat: torch.Tensor, r: int) -> torch.Tensor:
    """The r leading left singular vectors of ``mat`` (columns) by SVD: the generic ``round()`` of a tensor that
    did not come from ``TangentVector.construct()`` (rank tuning, tests); not on the training path."""
    r = min(r, mat.shape[0])
    U, _, _ = torch.linalg.svd(mat, full_matrices=False)
    return U[:, :r]


class Tucker:
    """``X = core x_0 factors[0] x_1 factors[1] x_2 factors[2]`` with factors
    ``[R (nR,a), S (N,b), O (N,c)]`` and core axes (relation, subject, object)."""

    def __init__(self, core: torch.Tensor, factors: Sequence[torch.Tensor], orth_cols: Sequence[int] = None):
        self.core = core
        self.factors = list(factors)
        # optional hint for round(): factor i starts with orth_cols[i] orthonormal columns that are orthogonal
        # to its remaining ones (set by TangentVector.construct(); never required)
        self.orth_cols = list(orth_cols) if orth_cols is not None else None

    @property
    def rank(self):
        return tuple(self.core.shape)

    @property
    def shape(self):
        return tuple(f.shape[0] for f in self.factors)

    def norm(self) -> torch.Tensor:
        return _gram_norm(self.core, [_tn(f, f) for f in self.factors])

    def full(self) -> torch.Tensor:
        t = self.core
        for i, f in enumerate(self.factors):
            t = _mode_dot(t, f, i)
        return t

    def round(self, rank: Sequence[int]) -> "Tucker":
        """Best-effort rank-``rank`` approximation with orthonormal factors (truncated HOSVD).  A tensor built by
        ``TangentVector.construct()`` (``orth_cols`` set) takes the structured, synchronisation-free path."""
        if self.orth_cols is not None:
            pairs = [(f[:, :k], f[:, k:]) for f, k in zip(self.factors, self.orth_cols)]
            core, new = _round_tangent_step(self.core, pairs, list(range(len(pairs))), [int(r) for r in rank])
            return Tucker(core, new)
        core = self.core
        qs = []
        for i, f in enumerate(self.factors):
            q, r = torch.linalg.qr(f)                     # thin: (n, k), (k, k)
            qs.append(q)
            core = _mode_dot(core, r, i)
        new_factors = []
        for i, q in enumerate(qs):
            u = _truncated_left_basis(_unfold(core, i), int(rank[i]))
            new_factors.append(q @ u)
            core = _mode_dot(core, u.transpose(0, 1), i)
        return Tucker(core, new_factors)

    def __add__(self, other: "Tucker") -> "Tucker":
        """Block-diagonal sum (ranks add)."""
        a0, b0, c0 = self.core.shape
        a1, b1, c1 = other.core.shape
        core = self.core.new_zeros((a0 + a1, b0 + b1, c0 + c1))
        core[:a0, :b0, :c0] = self.core
        core[a0:, b0:, c0:] = other.core
        return Tucker(core, [torch.cat([f, g], dim=1) for f, g in zip(self.factors, other.factors)])

    def __rmul__(self, scalar) -> "Tucker":
        return Tucker(scalar * self.core, list(self.factors))


class SFTucker:
    """Shared-factor Tucker: the last ``num_shared_factors`` modes use ``shared_factor``
    (``X = core x_0 R x_1 E x_2 E`` for the symmetric model)."""

    def __init__(self, core: torch.Tensor, regular_factors: Sequence[torch.Tensor],
                 num_shared_factors: int, shared_factor: torch.Tensor, orth_cols: Sequence[int] = None):
        self.core = core
        self.regular_factors = list(regular_factors)
        self.num_shared_factors = num_shared_factors
        self.shared_factor = shared_factor
        self.orth_cols = list(orth_cols) if orth_cols is not None else None     # [regular..., shared]; see Tucker

    @property
    def rank(self):
        return tuple(self.core.shape)

    @property
    def factors(self):
        """All mode factors in core-axis order (the shared one repeated)."""
        return self.regular_factors + [self.shared_factor] * self.num_shared_factors

    def norm(self) -> torch.Tensor:
        ge = _tn(self.shared_factor, self.shared_factor)
        grams = [_tn(f, f) for f in self.regular_factors] + [ge] * self.num_shared_factors
        return _gram_norm(self.core, grams)

    def full(self) -> torch.Tensor:
        t = self.core
        for i, f in enumerate(self.factors):
            t = _mode_dot(t, f, i)
        return t

    def round(self, rank: Sequence[int]) -> "SFTucker":
        """Truncated HOSVD with ONE basis for the shared modes: the leading left singular vectors of the
        concatenated shared-mode unfoldings (the subspace that serves both modes best in the least-squares
        sense).  ``orth_cols`` set (``TangentVector.construct()``): the structured, synchronisation-free path."""
        nreg = len(self.regular_factors)
        ns = self.num_shared_factors
        if self.orth_cols is not None:
            fs = self.regular_factors + [self.shared_factor]
            pairs = [(f[:, :k], f[:, k:]) for f, k in zip(fs, self.orth_cols)]
            core, new = _round_tangent_step(self.core, pairs, list(range(nreg)) + [nreg] * ns,
                                            [int(rank[i]) for i in range(nreg)] + [int(rank[nreg])])
            return SFTucker(core, new[:nreg], ns, new[nreg])
        core = self.core
        qs = []
        for i, f in enumerate(self.regular_factors):
            q, r = torch.linalg.qr(f)
            qs.append(q)
            core = _mode_dot(core, r, i)
        qe, re = torch.linalg.qr(self.shared_factor)
        for m in range(nreg, nreg + ns):
            core = _mode_dot(core, re, m)
        new_regular = []
        for i, q in enumerate(qs):
            u = _truncated_left_basis(_unfold(core, i), int(rank[i]))
            new_regular.append(q @ u)
            core = _mode_dot(core, u.transpose(0, 1), i)
        cat = torch.cat([_unfold(core, m) for m in range(nreg, nreg + ns)], dim=1)
        ue = _truncated_left_basis(cat, int(rank[nreg]))
        for m in range(nreg, nreg + ns):
            core = _mode_dot(core, ue.transpose(0, 1), m)
        return SFTucker(core, new_regular, ns, qe @ ue)

    def __rmul__(self, scalar) -> "SFTucker":
        return SFTucker(scalar * self.core, list(self.regular_factors), self.num_shared_factors, self.shared_factor)
