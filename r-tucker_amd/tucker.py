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

import collections

from typing import Sequence

import torch


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


# How often the fast paths below gave way to a slower one (the driver logs and clears this once per epoch).
FALLBACKS = collections.Counter()


def _cholesky_qr2(D: torch.Tensor):
    """Thin QR of a tall-skinny ``D`` by two rounds of Cholesky QR on its Gram matrix (the Gram products are the
    split-K HIP GEMM on the GPU): O(n k^2) flops in three chip-filling GEMMs instead of k Householder
    reflections applied one after the other (rocSOLVER's geqrf spends 90 ms on a 40 943 x 400 factor, 55 % of a
    training step once the Gram products are fixed).  Columns are normalised first (Cholesky QR squares the
    condition number).  Returns ``None`` when the Gram matrix is numerically singular or the result is not
    orthonormal to 1e-4 -- the caller then falls back to Householder QR."""
    scale = torch.linalg.vector_norm(D, dim=0)
    if not bool((scale > 0).all()):
        return None
    Dn = D / scale
    k = D.shape[1]
    eye = torch.eye(k, dtype=D.dtype, device=D.device)
    R_total = None
    Q = Dn
    rounds = 2
    i = 0
    while i < rounds:
        S = _tn(Q, Q)
        L, info = torch.linalg.cholesky_ex(S)
        if int(info) != 0:
            if i > 0:
                return None
            # condition number above ~3e3 (its square does not survive fp32): one round on the SHIFTED Gram matrix
            # brings it down to ~1e3, two plain rounds finish (shifted Cholesky QR 3)
            FALLBACKS["cholesky_qr_shifted"] += 1
            L, info = torch.linalg.cholesky_ex(S + 1e-6 * k * eye)            # diag(S) = 1: trace = k
            if int(info) != 0:
                return None
            rounds = 3
        Q = torch.linalg.solve_triangular(L.transpose(0, 1), Q, upper=True, left=False)      # Q <- Q L^-T
        R_total = L.transpose(0, 1) if R_total is None else L.transpose(0, 1) @ R_total
        i += 1
    err = (_tn(Q, Q) - eye).abs().max()
    if not bool(err < 1e-4):
        return None
    return Q, R_total * scale          # D = Q (R diag(scale))


def _qr_thin(f: torch.Tensor, n_orth: int = 0):
    """Thin QR of a factor.  ``n_orth`` > 0: the first ``n_orth`` columns are known to be orthonormal and (up to
    rounding) orthogonal to the rest -- the ``[U, dU]`` factors of ``TangentVector.construct()`` -- so only the
    remaining block needs work: ``[U, D] = [U, Q_D] [[I, U^T D], [0, R_D]]``."""
    n, k = f.shape
    if n_orth <= 0 or n_orth >= k or not f.is_cuda or f.dtype != torch.float32 or n < 8192:
        return torch.linalg.qr(f)
    U, D = f[:, :n_orth], f[:, n_orth:]
    C = _tn(U, D)                               # ~0 by the gauge condition; removed explicitly
    D = D - U @ C
    qr = _cholesky_qr2(D)
    if qr is None:
        FALLBACKS["householder_qr"] += 1
        qr = torch.linalg.qr(D)                 # of the new block only: a quarter of the flops of qr(f)
    Qd, Rd = qr
    R = f.new_zeros((k, k))
    R[:n_orth, :n_orth] = torch.eye(n_orth, dtype=f.dtype, device=f.device)
    R[:n_orth, n_orth:] = C
    R[n_orth:, n_orth:] = Rd
    return torch.cat([U, Qd], dim=1), R


def _polish(factor: torch.Tensor):
    """``(Q, R)`` with ``factor = Q R`` and Q orthonormal to fp32 rounding, for a factor that is orthonormal only up
    to accumulated drift: the structured QR above takes the leading block ``U`` of ``[U, dU]`` as given, so
    without this the error of the factors would add up step after step.  ``None``: leave the factor alone."""
    if not factor.is_cuda or factor.dtype != torch.float32 or factor.shape[0] < 8192:
        return None
    return _cholesky_qr2(factor)


def _host_eigh(gram: torch.Tensor):
    """``eigh`` of a small symmetric matrix on the host with at most 8 threads: torch sizes its intra-op pool to
    the machine (256 hardware threads on the MI355X hosts), which slows a 400 x 400 problem from 8 ms to > 100 ms."""
    n = torch.get_num_threads()
    if n <= 8:
        return torch.linalg.eigh(gram)
    try:
        torch.set_num_threads(8)
        return torch.linalg.eigh(gram)
    finally:
        torch.set_num_threads(n)


def _truncated_left_basis(mat: torch.Tensor, r: int) -> torch.Tensor:
    """The r leading left singular vectors of ``mat`` (columns).  Core unfoldings are short and wide (2r x 4r^2):
    on the GPU in fp32 they come from the eigenvectors of the small Gram matrix ``mat mat^T`` (one GEMM + one
    2r x 2r symmetric eigenproblem instead of a Jacobi SVD of the wide matrix).

    Squaring the spectrum puts the fp32 noise floor at 3e-4 of the largest singular value.  That is harmless
    while the r-th direction is well above it -- and wrong once the core has dead directions that the step's
    new ones (singular values ~ the step length, 1e-3 of ||T|| and less) should replace: measured on WN18RR, half
    of a step's change was lost in ``round()`` from epoch 16 on.  When the smallest kept eigenvalue is within
    1e-5 of the largest, the Gram matrix is therefore accumulated in float64 (exact products of fp32 values) and
    its eigenvectors are taken on the host (LAPACK: 15 ms for 400 x 400; rocSOLVER's float64 ``syevd`` needs
    80 ms and its fp32 divide-and-conquer does not always converge on such spectra).  Elsewhere (CPU, float64:
    the identity tests) the SVD itself."""
    if mat.shape[0] <= r:
        r = mat.shape[0]
    if mat.is_cuda and mat.dtype == torch.float32 and mat.shape[1] >= 4 * mat.shape[0]:
        try:
            w, V = torch.linalg.eigh(mat @ mat.transpose(0, 1))        # ascending eigenvalues
            if bool(torch.isfinite(V).all()) and bool(w[-r] > 1e-5 * w[-1]):
                return V[:, -r:].flip(1)
        except torch.linalg.LinAlgError:
            pass
        FALLBACKS["eigh_float64_host"] += 1
        m64 = mat.double()
        try:
            w, V = _host_eigh((m64 @ m64.transpose(0, 1)).cpu())
            if bool(torch.isfinite(V).all()):
                return V[:, -r:].flip(1).to(device=mat.device, dtype=mat.dtype)
        except torch.linalg.LinAlgError:
            pass
        FALLBACKS["svd"] += 1
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
        """Best-effort rank-``rank`` approximation with orthonormal factors (truncated HOSVD)."""
        core = self.core
        qs = []
        for i, f in enumerate(self.factors):
            q, r = _qr_thin(f, self.orth_cols[i] if self.orth_cols else 0)    # thin: (n, k), (k, k)
            qs.append(q)
            core = _mode_dot(core, r, i)
        new_factors = []
        for i, q in enumerate(qs):
            u = _truncated_left_basis(_unfold(core, i), int(rank[i]))
            f = q @ u
            core = _mode_dot(core, u.transpose(0, 1), i)
            fixed = _polish(f) if self.orth_cols else None
            if fixed is not None:
                f, rfix = fixed
                core = _mode_dot(core, rfix, i)
            new_factors.append(f)
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
        sense)."""
        nreg = len(self.regular_factors)
        core = self.core
        qs = []
        for i, f in enumerate(self.regular_factors):
            q, r = _qr_thin(f, self.orth_cols[i] if self.orth_cols else 0)
            qs.append(q)
            core = _mode_dot(core, r, i)
        qe, re = _qr_thin(self.shared_factor, self.orth_cols[-1] if self.orth_cols else 0)
        for m in range(nreg, nreg + self.num_shared_factors):
            core = _mode_dot(core, re, m)
        new_regular = []
        for i, q in enumerate(qs):
            u = _truncated_left_basis(_unfold(core, i), int(rank[i]))
            f = q @ u
            core = _mode_dot(core, u.transpose(0, 1), i)
            fixed = _polish(f) if self.orth_cols else None
            if fixed is not None:
                f, rfix = fixed
                core = _mode_dot(core, rfix, i)
            new_regular.append(f)
        cat = torch.cat([_unfold(core, m) for m in range(nreg, nreg + self.num_shared_factors)], dim=1)
        ue = _truncated_left_basis(cat, int(rank[nreg]))
        for m in range(nreg, nreg + self.num_shared_factors):
            core = _mode_dot(core, ue.transpose(0, 1), m)
        fe = qe @ ue
        fixed = _polish(fe) if self.orth_cols else None
        if fixed is not None:
            fe, rfix = fixed
            for m in range(nreg, nreg + self.num_shared_factors):
                core = _mode_dot(core, rfix, m)
        return SFTucker(core, new_regular, self.num_shared_factors, fe)

    def __rmul__(self, scalar) -> "SFTucker":
        return SFTucker(scalar * self.core, list(self.regular_factors), self.num_shared_factors, self.shared_factor)
