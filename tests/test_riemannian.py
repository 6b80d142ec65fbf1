"""Identities of the Riemannian layer (r-tucker_amd/{tucker,riemannian,optim}.py) on CPU, float64.
``tucker_riemopt`` is absent offline and the reference holds no tests for it, so parity with that package
is UNPINNED; what is checked here is the textbook geometry the reference's call sites rely on
(SURVEY.md Appendix C): point <-> tangent representation, norms, projection, Riemannian gradient,
retraction, and that the optimizers built on them descend and keep the factors orthonormal."""
import numpy as np
import pytest
import torch

from r_tucker_amd.riemannian import SFTuckerRiemannian, TuckerRiemannian
from r_tucker_amd.tucker import SFTucker, Tucker

DT = torch.float64


def _orth(n, r, g):
    return torch.linalg.qr(torch.randn(n, r, dtype=DT, generator=g))[0]


def point(sym, seed=0, shape=(7, 11, 11), rank=(2, 3, 3)):
    g = torch.Generator().manual_seed(seed)
    core = torch.randn(rank, dtype=DT, generator=g)
    R = _orth(shape[0], rank[0], g)
    if sym:
        return SFTucker(core, [R], 2, _orth(shape[1], rank[1], g))
    return Tucker(core, [R, _orth(shape[1], rank[1], g), _orth(shape[2], rank[2], g)])


def geo(sym):
    return SFTuckerRiemannian if sym else TuckerRiemannian


def random_tangent(x, sym, seed=1):
    """A gauge-respecting tangent vector: project a random dense tensor."""
    g = torch.Generator().manual_seed(seed)
    Z = torch.randn(x.full().shape, dtype=DT, generator=g)
    return geo(sym).project(x, dense_as_tucker(Z, sym)), Z


def dense_as_tucker(Z, sym):
    eye = [torch.eye(n, dtype=DT) for n in Z.shape]
    if sym:      # a general dense tensor as an SFTucker needs one factor for modes 1 and 2: identity works (n1 == n2)
        return SFTucker(Z, [eye[0]], 2, eye[1])
    return Tucker(Z, eye)


def deltas(tv, sym):
    return (tv.delta_regular_factors + [tv.delta_shared_factor]) if sym else tv.delta_factors


def bases(x, sym):
    return (x.regular_factors + [x.shared_factor]) if sym else x.factors


@pytest.mark.parametrize("sym", [False, True])
def test_point_as_tangent_vector_and_construct(sym):
    x = point(sym)
    tv = geo(sym).TangentVector(x)
    c = tv.construct()
    assert tuple(c.core.shape) == tuple(2 * r for r in x.core.shape)
    assert torch.allclose(c.full(), x.full(), atol=1e-12)
    zero = geo(sym).TangentVector(x, torch.zeros_like(x.core))
    assert zero.construct().full().abs().max() == 0
    assert abs(tv.norm().item() - x.full().norm().item()) < 1e-10
    assert abs(x.norm().item() - x.full().norm().item()) < 1e-10          # container norm via Gram matrices


@pytest.mark.parametrize("sym", [False, True])
def test_tangent_arithmetic_norm_and_gauge(sym):
    x = point(sym)
    xi, _ = random_tangent(x, sym, 1)
    eta, _ = random_tangent(x, sym, 2)
    for u, d in zip(bases(x, sym), deltas(xi, sym)):
        assert (u.T @ d).abs().max().item() < 1e-12                        # gauge
    comb = 0.5 * xi + (-2.0) * eta
    assert torch.allclose(comb.construct().full(), 0.5 * xi.construct().full() - 2.0 * eta.construct().full(), atol=1e-12)
    assert abs(xi.norm().item() - xi.construct().full().norm().item()) < 1e-10
    assert abs(xi.construct().norm().item() - xi.norm().item()) < 1e-10


@pytest.mark.parametrize("sym", [False, True])
def test_projection_is_idempotent_and_orthogonal(sym):
    x = point(sym)
    xi, Z = random_tangent(x, sym, 3)
    again = geo(sym).project(x, xi.construct())
    assert torch.allclose(again.construct().full(), xi.construct().full(), atol=1e-11)
    # Z - P(Z) is orthogonal to every tangent vector (for the shared-factor manifold: for tensors symmetric in
    # modes 1,2 -- the tangent space of that manifold only contains such directions when G is; test with eta)
    eta, _ = random_tangent(x, sym, 4)
    resid = Z - xi.construct().full()
    assert abs((resid * eta.construct().full()).sum().item()) < 1e-10
    # the point itself is in its tangent space
    px = geo(sym).project(x, x)
    assert torch.allclose(px.construct().full(), x.full(), atol=1e-11)


@pytest.mark.parametrize("sym", [False, True])
def test_riemannian_gradient_is_projected_euclidean_gradient(sym):
    x = point(sym)
    g = torch.Generator().manual_seed(5)
    A = torch.randn(x.full().shape, dtype=DT, generator=g)
    loss_fn = lambda T: 0.5 * ((T.full() - A) ** 2).sum() + 0.1 * T.norm() ** 2      # noqa: E731
    rgrad, loss = geo(sym).grad(loss_fn, x)
    X = x.full()
    assert abs(loss.item() - (0.5 * ((X - A) ** 2).sum() + 0.1 * X.norm() ** 2).item()) < 1e-10
    egrad = (X - A) + 0.2 * X
    want = geo(sym).project(x, dense_as_tucker(egrad, sym))
    assert torch.allclose(rgrad.construct().full(), want.construct().full(), atol=1e-10)
    for u, d in zip(bases(x, sym), deltas(rgrad, sym)):
        assert (u.T @ d).abs().max().item() < 1e-12
    # directional derivative along a tangent direction = <rgrad, eta>
    eta, _ = random_tangent(x, sym, 6)
    t = 1e-6
    xp = ((t * eta) + geo(sym).TangentVector(x)).construct()
    xm = (((-t) * eta) + geo(sym).TangentVector(x)).construct()
    fd = (loss_fn(xp) - loss_fn(xm)).item() / (2 * t)
    ip = (rgrad.construct().full() * eta.construct().full()).sum().item()
    assert abs(fd - ip) < 1e-5 * max(1.0, abs(ip))


@pytest.mark.parametrize("sym", [False, True])
def test_round_is_a_retraction(sym):
    x = point(sym)
    xi, _ = random_tangent(x, sym, 7)
    xi = (1.0 / xi.norm().item()) * xi
    errs = []
    for t in (1e-2, 1e-3):
        moved = ((t * xi) + geo(sym).TangentVector(x)).construct()
        y = moved.round(x.rank)
        assert tuple(y.core.shape) == tuple(x.core.shape)
        for u in bases(y, sym):
            assert (u.T @ u - torch.eye(u.shape[1], dtype=DT)).abs().max().item() < 1e-12
        errs.append((y.full() - moved.full()).norm().item())
    assert errs[0] < 1e-3 and errs[1] < errs[0] * 2e-2                      # O(t^2)
    # a tensor already on the manifold is reproduced
    assert torch.allclose(x.round(x.rank).full(), x.full(), atol=1e-12)


@pytest.mark.parametrize("sym", [False, True])
@pytest.mark.parametrize("opt_name", ["RGD", "RSGDwithMomentum", "RiemannianAdam"])
def test_optimizers_descend_and_keep_the_manifold(sym, opt_name):
    """The reference's loop (train.py:78-87) on a quadratic: fit -> step -> parameters written back."""
    import r_tucker_amd as rt
    mod = __import__("r_tucker_amd.model.%s.optim" % ("symmetric" if sym else "asymmetric"), fromlist=["x"])
    shape, rank = (6, 9, 9), (2, 3, 3)
    target = point(sym, seed=11, shape=shape, rank=rank).full()
    Model = rt.SymmetricR_TuckER if sym else rt.AsymmetricR_TuckER
    torch.manual_seed(3)
    model = Model((shape[1], shape[0]), rank).double()
    model.init()
    with torch.no_grad():
        model.core.mul_(3.0)
    if sym:
        params = torch.nn.ParameterList([model.core, model.E.weight, model.R.weight])
        extract = lambda: SFTucker(model.core.data, [model.R.weight], num_shared_factors=2, shared_factor=model.E.weight)  # noqa: E731
    else:
        params = torch.nn.ParameterList([model.core, model.S.weight, model.R.weight, model.O.weight])
        extract = lambda: Tucker(model.core.data, [model.R.weight, model.S.weight, model.O.weight])  # noqa: E731
    kw = {"RGD": {}, "RSGDwithMomentum": {"momentum_beta": 0.5}, "RiemannianAdam": {}}[opt_name]
    lr = 0.3 if opt_name != "RiemannianAdam" else 0.2
    opt = getattr(mod, opt_name)(params, rank, lr, **kw)
    sched = torch.optim.lr_scheduler.ExponentialLR(opt, gamma=0.995)          # train.py:213: a torch scheduler drives lr
    loss_fn = lambda T: 0.5 * ((T.full() - target) ** 2).sum()               # noqa: E731
    first = None
    for it in range(300):
        x_k = extract()
        gn = opt.fit(loss_fn, x_k, normalize_grad=False if opt_name != "RiemannianAdam" else 1.0)
        opt.step()
        opt.zero_grad(set_to_none=True)
        sched.step()
        first = opt.loss.item() if first is None else first
        assert torch.isfinite(gn)
    final = loss_fn(extract()).item()
    assert final < 0.05 * first, (first, final)
    for w in ([model.E.weight, model.R.weight] if sym else [model.S.weight, model.R.weight, model.O.weight]):
        assert (w.T @ w - torch.eye(w.shape[1], dtype=DT)).abs().max().item() < 1e-10
    assert opt.param_groups[0]["lr"] < lr                                    # the scheduler reached the optimizer


@pytest.mark.parametrize("sym", [False, True])
@pytest.mark.parametrize("dt", [torch.float64, torch.float32])
def test_gradient_at_a_core_that_lost_a_direction_stays_bounded(sym, dt):
    """A core unfolding with a (numerically) dead direction: the gauge-fixed factor components carry the inverse
    of the core's Gram matrix, which is regularised (shift eps = RCOND * trace) -- the plain solve returned 1e37
    in one column after ~1400 RSGD steps on WN18RR.  A singular value s of the unfolding is then amplified by
    s / (s^2 + eps) <= 1 / (2 sqrt(eps)) instead of 1 / s; the step from there stays finite."""
    x = point(sym)
    core = x.core.clone()
    core[:, :, -1] = 1e-12 * core[:, :, -1]          # mode-2 direction numerically dead
    if sym:
        core[:, -1, :] = 1e-12 * core[:, -1, :]       # (shared factor: dead in both of its modes)
        xs = SFTucker(core.to(dt), [f.to(dt) for f in x.regular_factors], 2, x.shared_factor.to(dt))
    else:
        xs = Tucker(core.to(dt), [f.to(dt) for f in x.factors])
    g = torch.Generator().manual_seed(3)
    W = torch.randn(xs.full().shape, dtype=dt, generator=g)
    grad, _ = geo(sym).grad(lambda T: (T.full() * W).sum(), xs)
    assert all(torch.isfinite(d).all() for d in deltas(grad, sym)) and torch.isfinite(grad.delta_core).all()
    from r_tucker_amd.riemannian import RCOND
    eps = RCOND[dt] * float((xs.core.double() ** 2).sum())        # trace of any mode's Gram matrix = ||core||^2
    if sym:
        eps *= 2                                                   # (the shared factor sums the Gram matrices of two modes)
    bound = float(W.abs().sum()) / (2 * eps ** 0.5)                # |Euclidean gradient entry| <= sum |W| (orthonormal factors)
    assert max(d.abs().max().item() for d in deltas(grad, sym)) <= bound
    assert torch.isfinite(grad.norm())
    moved = ((-0.1) * grad + geo(sym).TangentVector(xs)).construct().round(tuple(xs.core.shape))
    assert torch.isfinite(moved.full()).all()
