"""Riemannian geometry of the fixed-multilinear-rank Tucker manifolds -- the subset of
``tucker_riemopt.TuckerRiemannian`` / ``SFTuckerRiemannian`` that the reference's optimizers call
(``src/model/asymmetric/optim.py:32,52,86-89,107-108``; ``src/model/symmetric/optim.py:34,54,80-83,101-103,
133-136,160-161``).  ``tucker_riemopt`` 1.0.1 is not vendored and not installable offline, and the
reference holds no tests for it, so this is our own implementation of the standard geometry
(Koch-Lubich; Kressner-Steinlechner-Vandereycken), named after the calls the reference makes:
**parity with the package is unpinned**; correctness is established by identities
(``tests/test_riemannian.py``): ``construct(TangentVector(x)) == x``, ``project`` idempotent and
orthogonal, ``grad`` equal to the projection of the Euclidean gradient, gauge conditions, first-order
retraction.

A point ``x`` is a ``Tucker`` (``SFTucker``) with orthonormal factor columns.  A tangent vector is

    xi = dG x_i U_i  +  sum_i  G x_i dU_i x_{j != i} U_j ,      U_i^T dU_i = 0        (gauge)

(for the shared-factor manifold the shared modes carry ONE ``dE``).  ``construct()`` writes ``xi`` as an
explicit Tucker tensor of rank 2r: factors ``[U_i, dU_i]``, core a 2a x 2b x 2c block tensor with ``dG`` in
block (0,0,0) and ``G`` in the three blocks with a single 1.  ``grad`` differentiates
``loss_fn(construct(dG, dU))`` at ``(dG, dU) = (G, 0)`` -- which is why the scoring closure is evaluated
at DOUBLED rank during training -- and maps the partial derivatives to the tangent
space: ``dU_i <- (I - U_i U_i^T) (d/d dU_i) (G_(i) G_(i)^T)^-1``.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence

import torch

from .smalllinalg import spd_inverse, spd_inverse_many, wide_gram
from .tucker import SFTucker, Tucker, _mode_dot, _tn, _unfold


def _core_gram(core: torch.Tensor, mode: int, wide: bool = False) -> torch.Tensor:
    """``G_(mode) G_(mode)^T``.  ``wide``: accumulate an fp32 core's Gram matrix in float64 (the products are
    exact, so eigenvalues down to 1e-14 of the largest survive instead of 1e-7) -- for the inverse below."""
    u = _unfold(core, mode)
    if wide and u.dtype == torch.float32:
        u = u.double()
    return wide_gram(u)


def _point_grams(x, modes):
    """float64 core Gram matrices of a point, computed once per point object: ``fit`` projects the previous
    direction and takes the gradient at the same ``x_k`` (``asymmetric/optim.py:86-89``) and both need all of them."""
    # keyed by the core tensor and its version counter.  (x.core aliases model.core.data, which optimizer.step()
    # overwrites through ANOTHER .data alias -- that write is invisible to this counter, so the optimizer drops the
    # cache itself after its write, optim._retract_and_write.)
    cache = getattr(x, "_core_grams64", None)
    if cache is None or cache[0] is not x.core or cache[2] != x.core._version:
        cache = (x.core, {}, x.core._version)
        try:
            x._core_grams64 = cache
        except AttributeError:
            pass
    out = []
    for m in modes:
        if m not in cache[1]:
            cache[1][m] = _core_gram(x.core.detach(), m, wide=True)
        out.append(cache[1][m])
    return out


# Regularisation of the core Gram matrix in ``_solve_right``, relative to its trace, by the precision of the data.
RCOND = {torch.float32: 1e-8, torch.float64: 1e-14}


def _solve_right(mat: torch.Tensor, gram: torch.Tensor) -> torch.Tensor:
    """``mat @ (gram + eps I)^-1`` for the symmetric positive semidefinite core Gram matrix, ``eps = RCOND *
    trace(gram)`` (Cholesky in ``gram``'s precision, one small inverse; no eigensolver, no host sync).  The
    gauge-fixed factor components of a tangent vector carry the inverse of the core's Gram matrix; a core
    unfolding that has lost a direction numerically (seen after ~1400 small steps on WN18RR: one singular value
    of the mode-1 unfolding collapsed) made the plain solve return entries of 1e37 in one column, finite until
    multiplied by the learning rate.  With the shift the amplification is bounded by 1e8 / trace, directions the
    core uses (eigenvalues >> eps) are untouched to 1e-8 relative, and a direction it does not use gets next to
    no factor component of its own -- the core component ``dG`` still moves it.  A zero core (trace 0) gives 0."""
    inv = spd_inverse(gram, RCOND.get(mat.dtype, 1e-8))
    return mat @ inv.to(mat.dtype)


# Semantics of ``tucker_riemopt`` that the reference's call sites do not pin (the package is not available offline;
# SURVEY.md Appendix C) and that decide how long a NORMALISED step is (asymmetric/optim.py:89-92).  The defaults are
# the textbook ones; tools/train_lease.py --variant-* flips them to test the alternatives against the README recipe
# (DESIGN.md section 8 holds the table).  Asymmetric (Tucker) geometry only.
#   "norm": "frobenius"  -- TangentVector.norm() = Frobenius norm of the tangent vector as a tensor
#           "coordinate" -- sqrt(|dG|^2 + sum |dU_i|^2), the norm of its coordinates
#   "reg_in_norm": True  -- grad() differentiates the whole loss_fn (BCE + coeff |T|^2), so the norm the step is
#                           normalised by includes the regulariser's gradient;  False -- the data term's gradient only
EXPERIMENT = {"norm": "frobenius", "reg_in_norm": True}


def _split_loss(loss_fn):
    """``(data_fn, coeff)`` when ``loss_fn`` declares itself as ``data_fn(T) + coeff * T.norm() ** 2`` (the training
    loss of train.py:79; ``driver.RegularisedLoss``), else ``(loss_fn, None)``.

    At a point of the manifold (orthonormal factors) the squared-norm term needs no autodiff: ``||T|| = ||G||``,
    its Euclidean gradient ``2 coeff T`` lies in the tangent space and is the tangent vector with
    ``delta_core = 2 coeff G`` and zero factor components (the derivative w.r.t. ``dU_i`` at the construct point is
    ``2 coeff U_i G_(i) G_(i)^T``, in the span of ``U_i``, which the gauge projection removes).  Differentiating it
    through the Gram matrices of the 2r-wide factors instead costs three 40 943 x 400 x 400 products forward and six
    backward per step for the same numbers (``tests/test_smalllinalg.py::test_split_regulariser_equals_autodiff``)."""
    parts = getattr(loss_fn, "riemannian_split", None)
    if parts is None:
        return loss_fn, None
    return parts


def _solve_right_many(mats, grams):
    """``_solve_right`` for several (matrix, Gram) pairs: Gram matrices of equal size share one factorisation launch."""
    invs = spd_inverse_many(grams, RCOND.get(mats[0].dtype, 1e-8))
    return [m @ inv.to(m.dtype) for m, inv in zip(mats, invs)]


def _project_out(U: torch.Tensor, M: torch.Tensor) -> torch.Tensor:
    """(I - U U^T) M"""
    return M - U @ _tn(U, M)


def _block_core(dG: torch.Tensor, G: torch.Tensor) -> torch.Tensor:
    """The 2a x 2b x 2c core of ``construct``: dG at (0,0,0); G at (1,0,0), (0,1,0), (0,0,1)."""
    a, b, c = G.shape
    z_a = G.new_zeros((a, b, c))
    top = torch.cat([torch.cat([dG, G], dim=2), torch.cat([G, z_a], dim=2)], dim=1)            # (a, 2b, 2c)
    bot = torch.cat([torch.cat([G, z_a], dim=2), torch.cat([z_a, z_a], dim=2)], dim=1)
    return torch.cat([top, bot], dim=0)


# ======================================================================================================
# Tucker (asymmetric model: three distinct factors)
# ======================================================================================================
class TuckerTangentVector:
    def __init__(self, point: Tucker, delta_core: Optional[torch.Tensor] = None,
                 delta_factors: Optional[Sequence[torch.Tensor]] = None):
        self.point = point
        self.delta_core = point.core if delta_core is None else delta_core
        self.delta_factors: List[torch.Tensor] = ([torch.zeros_like(f) for f in point.factors]
                                                  if delta_factors is None else list(delta_factors))

    def __rmul__(self, a):
        return TuckerTangentVector(self.point, a * self.delta_core, [a * d for d in self.delta_factors])

    def __neg__(self):
        return (-1.0) * self

    def __add__(self, other: "TuckerTangentVector"):
        return TuckerTangentVector(self.point, self.delta_core + other.delta_core,
                                   [p + q for p, q in zip(self.delta_factors, other.delta_factors)])

    def norm(self) -> torch.Tensor:
        override = getattr(self, "_norm_override", None)
        if override is not None:
            return override
        G = self.point.core
        s = (self.delta_core * self.delta_core).sum()
        for i, d in enumerate(self.delta_factors):
            if EXPERIMENT["norm"] == "coordinate":
                s = s + (d * d).sum()
            else:
                s = s + (_tn(d, d) * _core_gram(G, i)).sum()
        return torch.sqrt(torch.clamp(s, min=0.0))

    def construct(self) -> Tucker:
        x = self.point
        return Tucker(_block_core(self.delta_core, x.core),
                      [torch.cat([u, d], dim=1) for u, d in zip(x.factors, self.delta_factors)],
                      orth_cols=[u.shape[1] for u in x.factors])


class TuckerRiemannian:
    TangentVector = TuckerTangentVector

    @staticmethod
    def grad(loss_fn: Callable[[Tucker], torch.Tensor], x: Tucker, retain_graph: bool = False):
        """Riemannian gradient of ``loss_fn`` at ``x`` -> ``(TangentVector, loss value)``."""
        G = x.core.detach()
        Us = [u.detach() for u in x.factors]
        dG = G.clone().requires_grad_(True)
        dUs = [torch.zeros_like(u).requires_grad_(True) for u in Us]
        loss_fn, coeff = _split_loss(loss_fn)
        with torch.enable_grad():
            T = Tucker(_block_core(dG, G), [torch.cat([u, d], dim=1) for u, d in zip(Us, dUs)])
            loss = loss_fn(T)
            grads = torch.autograd.grad(loss, [dG] + dUs, retain_graph=retain_graph)
        g_core, g_fac = grads[0], grads[1:]
        g_core_data = g_core
        if coeff is not None:
            g_core = g_core + (2.0 * coeff) * G
            loss = loss + coeff * (G * G).sum()
        grams = _point_grams(x, range(len(Us)))
        deltas = _solve_right_many([_project_out(u, g) for u, g in zip(Us, g_fac)], grams)
        pt = x if x.core is G else Tucker(G, Us)
        tv = TuckerTangentVector(pt, g_core, deltas)
        if coeff is not None and not EXPERIMENT["reg_in_norm"]:
            tv._norm_override = TuckerTangentVector(pt, g_core_data, deltas).norm().detach()
        return tv, loss.detach()

    @staticmethod
    def project(x: Tucker, Z: Tucker) -> TuckerTangentVector:
        """Orthogonal projection of an explicit Tucker tensor ``Z`` onto the tangent space at ``x``."""
        G = x.core
        Us = x.factors
        Ms = [_tn(u, v) for u, v in zip(Us, Z.factors)]          # r_i x k_i
        dG = Z.core
        for i, m in enumerate(Ms):
            dG = _mode_dot(dG, m, i)
        ws = []
        for i, (u, v) in enumerate(zip(Us, Z.factors)):
            t = Z.core
            for j, m in enumerate(Ms):
                if j != i:
                    t = _mode_dot(t, m, j)
            w = v @ (_unfold(t, i) @ _unfold(G, i).transpose(0, 1))           # n_i x r_i
            ws.append(_project_out(u, w))
        deltas = _solve_right_many(ws, _point_grams(x, range(len(Us))))
        return TuckerTangentVector(x, dG, deltas)


# ======================================================================================================
# SFTucker (symmetric model: modes 1 and 2 share the entity factor)
# ======================================================================================================
class SFTuckerTangentVector:
    def __init__(self, point: SFTucker, delta_core: Optional[torch.Tensor] = None,
                 delta_regular_factors: Optional[Sequence[torch.Tensor]] = None,
                 delta_shared_factor: Optional[torch.Tensor] = None):
        self.point = point
        self.delta_core = point.core if delta_core is None else delta_core
        self.delta_regular_factors: List[torch.Tensor] = (
            [torch.zeros_like(f) for f in point.regular_factors] if delta_regular_factors is None
            else list(delta_regular_factors))
        self.delta_shared_factor = (torch.zeros_like(point.shared_factor) if delta_shared_factor is None
                                    else delta_shared_factor)

    def __rmul__(self, a):
        return SFTuckerTangentVector(self.point, a * self.delta_core, [a * d for d in self.delta_regular_factors],
                                     a * self.delta_shared_factor)

    def __neg__(self):
        return (-1.0) * self

    def __add__(self, other: "SFTuckerTangentVector"):
        return SFTuckerTangentVector(self.point, self.delta_core + other.delta_core,
                                     [p + q for p, q in zip(self.delta_regular_factors, other.delta_regular_factors)],
                                     self.delta_shared_factor + other.delta_shared_factor)

    def _shared_gram(self) -> torch.Tensor:
        x = self.point
        nreg = len(x.regular_factors)
        return sum(_core_gram(x.core, m) for m in range(nreg, nreg + x.num_shared_factors))

    def norm(self) -> torch.Tensor:
        G = self.point.core
        s = (self.delta_core * self.delta_core).sum()
        for i, d in enumerate(self.delta_regular_factors):
            s = s + (_tn(d, d) * _core_gram(G, i)).sum()
        de = self.delta_shared_factor
        s = s + (_tn(de, de) * self._shared_gram()).sum()
        return torch.sqrt(torch.clamp(s, min=0.0))

    def construct(self) -> SFTucker:
        x = self.point
        assert len(x.regular_factors) == 1 and x.num_shared_factors == 2, "R-TuckER's symmetric model: (R, E, E)"
        return SFTucker(_block_core(self.delta_core, x.core),
                        [torch.cat([u, d], dim=1) for u, d in zip(x.regular_factors, self.delta_regular_factors)],
                        x.num_shared_factors, torch.cat([x.shared_factor, self.delta_shared_factor], dim=1),
                        orth_cols=[u.shape[1] for u in x.regular_factors] + [x.shared_factor.shape[1]])


class SFTuckerRiemannian:
    TangentVector = SFTuckerTangentVector

    @staticmethod
    def _shared_gram(G: torch.Tensor, nreg: int, ns: int) -> torch.Tensor:
        return sum(_core_gram(G, m, wide=True) for m in range(nreg, nreg + ns))     # only ever inverted

    @staticmethod
    def grad(loss_fn: Callable[[SFTucker], torch.Tensor], x: SFTucker, retain_graph: bool = False):
        G = x.core.detach()
        Rs = [u.detach() for u in x.regular_factors]
        E = x.shared_factor.detach()
        nreg, ns = len(Rs), x.num_shared_factors
        dG = G.clone().requires_grad_(True)
        dRs = [torch.zeros_like(u).requires_grad_(True) for u in Rs]
        dE = torch.zeros_like(E).requires_grad_(True)
        loss_fn, coeff = _split_loss(loss_fn)
        with torch.enable_grad():
            T = SFTucker(_block_core(dG, G), [torch.cat([u, d], dim=1) for u, d in zip(Rs, dRs)], ns,
                         torch.cat([E, dE], dim=1))
            loss = loss_fn(T)
            grads = torch.autograd.grad(loss, [dG] + dRs + [dE], retain_graph=retain_graph)
        g_core, g_reg, g_e = grads[0], grads[1:1 + nreg], grads[-1]
        if coeff is not None:
            g_core = g_core + (2.0 * coeff) * G
            loss = loss + coeff * (G * G).sum()
        d_reg = [_solve_right(_project_out(u, g), _core_gram(G, i, wide=True)) for i, (u, g) in enumerate(zip(Rs, g_reg))]
        d_e = _solve_right(_project_out(E, g_e), SFTuckerRiemannian._shared_gram(G, nreg, ns))
        return SFTuckerTangentVector(SFTucker(G, Rs, ns, E), g_core, d_reg, d_e), loss.detach()

    @staticmethod
    def project(x: SFTucker, Z: SFTucker) -> SFTuckerTangentVector:
        G = x.core
        nreg, ns = len(x.regular_factors), x.num_shared_factors
        xf, zf = x.factors, Z.factors                        # per-mode lists (shared factor repeated)
        Ms = [_tn(u, v) for u, v in zip(xf, zf)]
        dG = Z.core
        for i, m in enumerate(Ms):
            dG = _mode_dot(dG, m, i)

        def mode_term(i):
            t = Z.core
            for j, m in enumerate(Ms):
                if j != i:
                    t = _mode_dot(t, m, j)
            return zf[i] @ (_unfold(t, i) @ _unfold(G, i).transpose(0, 1))

        d_reg = [_solve_right(_project_out(xf[i], mode_term(i)), _core_gram(G, i, wide=True)) for i in range(nreg)]
        w = sum(mode_term(m) for m in range(nreg, nreg + ns))
        d_e = _solve_right(_project_out(x.shared_factor, w), SFTuckerRiemannian._shared_gram(G, nreg, ns))
        return SFTuckerTangentVector(x, dG, d_reg, d_e)


def set_backend(name: str = "pytorch") -> None:
    """``tucker_riemopt.set_backend`` (train.py:10,197): torch is the only backend here."""
    if name not in ("pytorch", "torch"):
        raise ValueError(f"r_tucker_amd runs on torch only, got backend {name!r}")
