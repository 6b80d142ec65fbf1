"""Small dense linear algebra of the Riemannian layer, written so that a whole optimizer step has NO host
synchronisation and no eigensolver (it can be captured into one HIP graph: tools/graphstep.py).

Round 2's retraction took the left bases of the core unfoldings from ``eigh`` of their Gram matrices: rocSOLVER's
``syevd`` is ~15 000 tiny launches per call (80 ms in float64), the host LAPACK detour 15 ms plus a device->host
copy per mode, and either way the spectrum is squared.  Here the leading left singular subspace comes from a
few steps of subspace iteration on the UNSQUARED operator (``Z = orth(M^T W); W = orth(M Z)``), warm-started at the
previous basis (``[I; 0]`` in the coordinates of ``[U, Q_dU]``), with every orthonormalisation a Cholesky QR on a
small Gram matrix in float64: GEMMs, one r x r Cholesky factor and its inverse per half step, nothing data
dependent on the host.  What a Tucker point needs is the SUBSPACE (any orthonormal basis of it gives the same
tensor), so no Rayleigh-Ritz step and no eigenvalues are required.

``gram_factor`` is the only factorisation: on the GPU it is one launch of ``rtk_gram_factor_f64`` (batched, one
workgroup per matrix, equilibration + shift + Cholesky + inverse + transposed outputs fused, csrc/rtk_chol.hip);
elsewhere (CPU tests) ``torch.linalg.cholesky_ex`` + ``solve_triangular``.  A failed factorisation cannot raise (that would need a sync): pivots are kept positive by a
relative diagonal shift, and the training driver reads one device-side health word per epoch.
"""
from __future__ import annotations

import os

import torch

# relative diagonal shifts (the Gram matrices are equilibrated to unit diagonal first)
SHIFT = {torch.float64: 1e-13, torch.float32: 1e-6}
SUBSPACE_ITERS = int(os.environ.get("R_TUCKER_AMD_SUBSPACE_ITERS", "3"))
_USE_HIP_CHOL = os.environ.get("R_TUCKER_AMD_HIP_CHOL", "1") == "1"
_HIP_CHOL_MAX = 256
BATCH_EQUAL_SIZES = os.environ.get("R_TUCKER_AMD_BATCH_FACTOR", "1") == "1"    # gram_factor_many / the round's groups

_start_cache = {}


def gram_factor(S: torch.Tensor, shift: float = None, equilibrate: bool = True, shift_trace: float = 0.0):
    """For a Gram matrix ``S = W^T W`` (k x k, any precision; factored in float64): ``(X, R)``, both upper
    triangular, with ``Q = W X`` orthonormal and ``W = Q R``:  with ``D = sqrt(diag S)`` (equilibration:
    Cholesky's accuracy depends on the condition number of the SCALED matrix; ``D = I`` if not ``equilibrate``) and
    ``L`` the Cholesky factor of ``D^-1 S D^-1 + (shift + shift_trace * trace S) I``,  ``X = D^-1 L^-T`` and
    ``R = L^T D``.  Not equilibrated, ``X X^T = (S + shift)^-1``.  A zero column stays zero; a zero matrix gives
    zeros.  On the GPU: ONE launch of ``rtk_gram_factor_f64`` (csrc/rtk_chol.hip) for k <= 256."""
    k = S.shape[-1]
    sh = SHIFT[torch.float64] if shift is None else float(shift)
    S64 = S.double()
    if S.is_cuda and _USE_HIP_CHOL and k <= _HIP_CHOL_MAX:
        from . import _lib
        lib = _lib.load()
        Sc = S64.contiguous()
        nb = Sc.numel() // (k * k)
        R = torch.empty_like(Sc)
        X = torch.empty_like(Sc)
        with torch.cuda.device(S.device):
            _lib.check(lib.rtk_gram_factor_f64(Sc.data_ptr(), nb, k, 1 if equilibrate else 0, sh, float(shift_trace),
                                               R.data_ptr(), X.data_ptr(), torch.cuda.current_stream(S.device).cuda_stream),
                       "rtk_gram_factor_f64")
        return X, R
    # CPU / large-k fallback with the HIP kernel's no-failure semantics (csrc/rtk_chol.hip): a matrix whose trace is
    # zero or not finite gives zero outputs; a pivot that cancelled (an indefinite or inconsistent Gram matrix) is
    # floored at half the shift instead of failing -- here by clamping the eigenvalues of the equilibrated, shifted
    # matrix, which leaves a positive definite input untouched to rounding
    S64 = torch.where(torch.isfinite(S64), S64, torch.zeros_like(S64))
    diag = S64.diagonal(dim1=-2, dim2=-1)
    tr = diag.sum(-1)
    ok = torch.isfinite(S.double().reshape(*S.shape[:-2], -1).sum(-1))
    live = ((tr > 0) & ok).to(torch.float64)
    if equilibrate:
        d = torch.sqrt(torch.clamp(diag, min=0.0))
        ds = torch.where(d > 0, d, torch.ones_like(d))
    else:
        ds = torch.ones_like(diag)
    Sn = S64 / (ds.unsqueeze(-1) * ds.unsqueeze(-2))
    eye = torch.eye(k, dtype=torch.float64, device=S.device)
    Sn = Sn + (sh + shift_trace * tr + (1.0 - live))[..., None, None] * eye        # (trace 0: factor I, result masked)
    floor = 0.5 * (sh + shift_trace * tr)
    if bool((floor > 0).any()):
        lam, V = torch.linalg.eigh(Sn)
        if bool((lam < floor.unsqueeze(-1)).any()):
            lam = torch.maximum(lam, floor.unsqueeze(-1))
            Sn = (V * lam.unsqueeze(-2)) @ V.transpose(-1, -2)
    L = torch.linalg.cholesky_ex(Sn).L
    Linv = torch.linalg.solve_triangular(L, eye.expand_as(Sn), upper=False)
    X = (Linv.transpose(-1, -2) / ds.unsqueeze(-1)) * live[..., None, None]
    R = (L.transpose(-1, -2) * ds.unsqueeze(-2)) * live[..., None, None]
    return X, R


def spd_inverse(gram: torch.Tensor, rcond: float) -> torch.Tensor:
    """``(gram + rcond * trace(gram) I)^-1`` in float64 (zero for a zero matrix): one factorisation launch + one
    small GEMM."""
    X, _ = gram_factor(gram, shift=0.0, equilibrate=False, shift_trace=rcond)
    return X @ X.transpose(-1, -2)


def gram_factor_many(grams, **kw):
    """``gram_factor`` of several Gram matrices with as few launches as possible: matrices of equal size go through
    ONE batched call (the HIP kernel takes one workgroup per matrix -- the S and O factors of the asymmetric model,
    the two entity modes of its core: two 200 x 200 problems side by side instead of one after the other; a
    factorisation is ~0.5 ms on a single CU and there are ~37 per optimizer step).  Returns ``[(X, R), ...]``."""
    out = [None] * len(grams)
    by_k = {}
    for i, g in enumerate(grams):
        by_k.setdefault((g.shape[-1], g.device), []).append(i)
    for idx in by_k.values():
        if len(idx) == 1 or not BATCH_EQUAL_SIZES:
            for i in idx:
                out[i] = gram_factor(grams[i], **kw)
        else:
            X, R = gram_factor(torch.stack([grams[i].double() for i in idx]), **kw)
            for j, i in enumerate(idx):
                out[i] = (X[j], R[j])
    return out


def spd_inverse_many(grams, rcond: float):
    """``spd_inverse`` of several matrices, equal sizes batched into one factorisation launch."""
    return [X @ X.transpose(-1, -2) for X, _ in gram_factor_many(grams, shift=0.0, equilibrate=False, shift_trace=rcond)]


def orth(W: torch.Tensor, rounds: int = 1):
    """Orthonormal basis of the columns of a small dense ``W`` (float64) by ``rounds`` of Cholesky QR."""
    for _ in range(rounds):
        X, _ = gram_factor(W.transpose(-1, -2) @ W)
        W = W @ X.to(W.dtype)
    return W


def _start(p: int, r: int, device, dtype):
    """``[I_r; eps * Omega]`` (p x r): the old basis plus a fixed small pseudo-random component in the new
    coordinates, so that a new direction that happens to be exactly orthogonal to the old row space is still
    reachable by the iteration."""
    key = (p, r, str(device), dtype)
    w = _start_cache.get(key)
    if w is None:
        g = torch.Generator().manual_seed(1234567 + 31 * p + r)
        w = torch.zeros((p, r), dtype=torch.float64)
        w[:r] = torch.eye(r, dtype=torch.float64)
        if p > r:
            w[r:] = 1e-3 * torch.randn((p - r, r), dtype=torch.float64, generator=g)
        w = torch.linalg.qr(w)[0].to(device=device, dtype=dtype)
        _start_cache[key] = w
    return w


def wide_gram(M: torch.Tensor) -> torch.Tensor:
    """``M M^T`` for a short and very wide ``M (p x m)``: summed over column chunks as a batched product (parallel
    over the chunks).  rocBLAS runs a GEMM with a p x p result and K = m on ONE workgroup: 6-17 ms for the
    relation-mode unfoldings of the WN18RR training core (10 x 40 000, 20 x 160 000) in float64."""
    p, m = M.shape
    if m < 16 * p or m < 4096:
        return M @ M.transpose(0, 1)
    c = 1024
    pad = (-m) % c
    if pad:
        M = torch.nn.functional.pad(M, (0, pad))
    Mc = M.reshape(p, -1, c).permute(1, 0, 2)                      # (chunks, p, c)
    return torch.bmm(Mc, Mc.transpose(1, 2)).sum(0)


def tall_gram_f64(D: torch.Tensor) -> torch.Tensor:
    """``D^T D`` of a tall-skinny block (n x k, any precision) accumulated in float64, as a batched product over row
    chunks.  The products of fp32 entries are exact in float64, so the result is positive semi-definite to 1e-16 -- the
    fp32 Gram kernel's result is not: over 40 943 rows its rounding noise is ~1e-5 of the diagonal, and a block whose
    columns are dependent to that level (the new block of a tangent step late in training) then has an INDEFINITE Gram
    matrix: cancelled pivots, L entries of 100 behind them, |L^-1| = 1e107 and NaN factors at epoch 74 of the
    reference's default configuration."""
    n, k = D.shape
    c = 1024
    rows = ((n + c - 1) // c) * c
    Dd = torch.zeros((rows, k), dtype=torch.float64, device=D.device)
    Dd[:n].copy_(D)
    Dc = Dd.view(-1, c, k)
    return torch.bmm(Dc.transpose(1, 2), Dc).sum(0)


def _wide_to_square(M: torch.Tensor) -> torch.Tensor:
    """A p x p matrix with the left singular vectors and singular values of a very wide ``M (p x m)``, m >> p: the
    transposed Cholesky factor of ``M M^T`` accumulated in float64 (the relation-mode unfolding of the WN18RR
    training core is 20 x 160 000: a GEMM with a 20 x 10 result and K = 160 000 is 17 ms in rocBLAS, ``wide_gram``).
    Squaring costs nothing the caller can see in float64: directions down to ~1e-7 of the largest are resolved."""
    G = wide_gram(M)
    _, R = gram_factor(G, shift=1e-15, equilibrate=True)
    return R.transpose(0, 1)                                        # L D: (L D)(L D)^T = G


def dominant_left_subspace(M: torch.Tensor, r: int, iters: int = None) -> torch.Tensor:
    """Orthonormal ``W (p x r)`` spanning (to the accuracy of ``iters`` subspace-iteration steps) the r leading
    left singular vectors of ``M (p x m)``, warm-started at the first r coordinate axes.  ``M`` may carry a leading
    batch dimension (independent problems of one shape: every step is then one batched GEMM / factorisation).

    One step is ``Z = orth(M^T W)``, ``W = orth(M Z + eta s W)``: the operator is applied unsquared between two
    orthonormalisations (conditioning sigma_1 / sigma_r per half step, not its square), and the tiny multiple of
    the previous basis (``eta = 1e-12`` of the largest column norm) keeps a basis vector alive when ``M`` has rank
    below r (a core that lost a direction and gained none).  Convergence of the angle to the true subspace is
    ``(sigma_{r+1} / sigma_r)^2`` per step; what is lost when that ratio is near one is a direction as weak as
    the one kept in its place."""
    p = M.shape[-2]
    r = min(r, p)
    if r == p:
        return torch.eye(p, dtype=M.dtype, device=M.device).expand(M.shape[:-2] + (p, p))
    iters = SUBSPACE_ITERS if iters is None else iters
    if M.dim() == 2 and M.shape[1] >= 16 * p and p <= _HIP_CHOL_MAX:
        M = _wide_to_square(M)
    W = _start(p, r, M.device, M.dtype)
    if M.dim() == 3:
        W = W.expand(M.shape[0], p, r)
    Mt = M.transpose(-1, -2)
    for _ in range(iters):
        Z = orth(Mt @ W)
        Y = M @ Z
        s = torch.linalg.vector_norm(Y, dim=-2).amax(dim=-1, keepdim=True).unsqueeze(-1)
        W = orth(Y + (1e-12 * s) * W)
    return orth(W)
