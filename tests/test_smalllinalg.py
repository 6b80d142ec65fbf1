"""The synchronisation-free small linear algebra of the optimizer step (r-tucker_amd/smalllinalg.py) and the
structured retraction built on it (tucker._round_tangent_step), on CPU in float64: against SVD / QR references,
on spectra chosen to break a Cholesky-based method, plus the two properties the training loop relies on --
the optimizers' trajectories do not depend on which orthonormal basis the retraction returns, and the three
optimizers recover a planted low-rank tensor."""
import pytest
import torch

from r_tucker_amd import smalllinalg as sl
from r_tucker_amd.riemannian import SFTuckerRiemannian, TuckerRiemannian
from r_tucker_amd.tucker import SFTucker, Tucker

DT = torch.float64


def _orth(n, r, g):
    return torch.linalg.qr(torch.randn(n, r, dtype=DT, generator=g))[0]


@pytest.mark.parametrize("k", [1, 5, 40])
def test_gram_factor_and_spd_inverse(k):
    g = torch.Generator().manual_seed(k)
    W = torch.randn(3 * k + 2, k, dtype=DT, generator=g) * torch.logspace(0, -5, k, dtype=DT)   # badly scaled columns
    S = W.T @ W
    Xp, Rp = sl.gram_factor(S, shift=0.0, equilibrate=False)                # plain: R = L^T, X = L^-T
    assert torch.allclose(Rp.T @ Rp, S, rtol=1e-12, atol=1e-300) and (Xp.T @ Rp.T - torch.eye(k, dtype=DT)).abs().max() < 1e-6
    assert Rp.tril(-1).abs().max() == 0 and Xp.tril(-1).abs().max() == 0
    A = S + 0.5 * S.diagonal().sum() * torch.eye(k, dtype=DT)
    assert torch.allclose(sl.spd_inverse(S, 0.5) @ A, torch.eye(k, dtype=DT), atol=1e-9)
    assert sl.spd_inverse(torch.zeros(k, k, dtype=DT), 1e-8).abs().max() == 0    # zero core: zero, not NaN
    X, R = sl.gram_factor(S)
    Q = W @ X
    assert (Q.T @ Q - torch.eye(k, dtype=DT)).abs().max() < 1e-10          # equilibration: scaling costs nothing
    assert torch.allclose(Q @ R, W, rtol=1e-9, atol=1e-14)
    Wz = W.clone()
    Wz[:, k // 2] = 0                                                     # a zero column stays zero, nothing is NaN
    Xz, Rz = sl.gram_factor(Wz.T @ Wz)
    assert torch.isfinite(Xz).all() and (Wz @ Xz)[:, k // 2].abs().max() == 0


def _kept(W, M):
    return ((W.T @ M) ** 2).sum().item()


@pytest.mark.parametrize("case", ["gap", "flat", "dead_and_new", "rank_deficient", "wide_spectrum"])
def test_dominant_left_subspace(case):
    g = torch.Generator().manual_seed(7)
    r, p, m = 12, 24, 300
    V = _orth(m, p, g)
    Ul = _orth(p, p, g)
    if case == "gap":
        sv = torch.cat([torch.logspace(2, 0, r, dtype=DT), torch.logspace(-3, -5, p - r, dtype=DT)])
    elif case == "flat":
        sv = torch.linspace(1.0, 0.5, p, dtype=DT)
    elif case == "wide_spectrum":
        sv = torch.logspace(6, -6, p, dtype=DT)
    else:
        sv = None
    if sv is not None:
        M = (Ul * sv) @ V.T
        # the old basis = the first r coordinate axes: rotate so that it is a fair warm start, not the answer
    elif case == "dead_and_new":
        # old block: 9 strong directions and 3 dead ones; new block: three directions of size 1 that are EXACTLY
        # orthogonal to the old row space (worst case for an iteration started in the old coordinates)
        svo = torch.cat([torch.logspace(4, 3, r - 3, dtype=DT), torch.full((3,), 1e-6, dtype=DT)])
        A = (_orth(r, r, g) * svo) @ V[:, :r].T
        B = torch.zeros(p - r, m, dtype=DT)
        B[:3] = V[:, r:r + 3].T
        M = torch.cat([A, B])
    else:   # rank_deficient: rank 7 < r
        M = (_orth(p, 7, g) * torch.logspace(1, 0, 7, dtype=DT)) @ V[:, :7].T
    W = sl.dominant_left_subspace(M, r)
    assert W.shape == (p, r) and (W.T @ W - torch.eye(r, dtype=DT)).abs().max() < 1e-9
    s = torch.linalg.svdvals(M)
    best = (s[:r] ** 2).sum().item()
    kept = _kept(W, M)
    if case == "flat":
        # no gap: three steps of iteration are not converged, but what is lost is bounded by the weakest kept energy
        assert kept >= best - 0.6 * (s[r - 1] ** 2).item() * r and kept <= best * (1 + 1e-12)
    else:
        assert abs(kept - best) <= 1e-6 * best
    if case == "dead_and_new":
        assert (W[r:r + 3] ** 2).sum().item() > 2.999          # the new coordinates entered the basis


@pytest.mark.parametrize("sym", [False, True])
def test_structured_round_matches_the_svd_round(sym):
    """round() of a constructed tangent step (hinted path) against the generic QR + SVD path on the same tensor."""
    import test_riemannian as T
    geo = SFTuckerRiemannian if sym else TuckerRiemannian
    x = T.point(sym, shape=(9, 30, 30), rank=(3, 6, 6))
    xi, _ = T.random_tangent(x, sym, 7)
    for t in (1e-3, 0.3, 30.0):
        moved = ((t * xi) + geo.TangentVector(x)).construct()
        plain = (SFTucker(moved.core, moved.regular_factors, 2, moved.shared_factor) if sym
                 else Tucker(moved.core, moved.factors))                      # no hint: QR + SVD
        y, y2 = moved.round(x.rank), plain.round(x.rank)
        full = moved.full()
        e, e2 = (y.full() - full).norm().item(), (y2.full() - full).norm().item()
        # (t = 30: two thirds of the tensor is truncated and the spectrum at the cut is flat -- three iteration
        # steps are within a few per cent of the SVD's truncation error there, and exact where there is a gap)
        assert e <= e2 * (1.05 if t > 1 else 1.001) + 1e-12 * full.norm().item(), (t, e, e2)
        for u in (y.regular_factors + [y.shared_factor] if sym else y.factors):
            assert (u.T @ u - torch.eye(u.shape[1], dtype=DT)).abs().max().item() < 1e-12


@pytest.mark.parametrize("sym", [False, True])
@pytest.mark.parametrize("opt_name", ["RSGDwithMomentum", "RiemannianAdam"])
def test_trajectory_does_not_depend_on_the_basis_of_the_retraction(sym, opt_name, monkeypatch):
    """The state carried over a step (previous direction / first moment) is an explicit tensor built BEFORE the
    parameters are overwritten in place (the reference constructs before its ``W.data.add_`` too,
    asymmetric/optim.py:109-114).  Round 2 built it afterwards, i.e. from the NEW factors and the old deltas: the
    momentum then depended on which basis the retraction happened to return -- two exact retractions (SVD basis,
    subspace-iteration basis) must give the same losses."""
    import r_tucker_amd as rt
    import test_riemannian as T
    mod = __import__("r_tucker_amd.model.%s.optim" % ("symmetric" if sym else "asymmetric"), fromlist=["x"])

    def run(generic):
        shape, rank = (6, 9, 9), (2, 3, 3)
        target = T.point(sym, seed=11, shape=shape, rank=rank).full()
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
        opt = getattr(mod, opt_name)(params, rank, 0.2)
        loss_fn = lambda T_: 0.5 * ((T_.full() - target) ** 2).sum()          # noqa: E731
        if generic:     # strip the hint: QR + SVD retraction (a different orthonormal basis of the same subspaces)
            for cls in (Tucker, SFTucker):
                orig = cls.round

                def plain(self, rank, _orig=orig):
                    self.orth_cols = None
                    return _orig(self, rank)
                monkeypatch.setattr(cls, "round", plain)
        hist = []
        for _ in range(60):
            opt.fit(loss_fn, extract(), normalize_grad=1.0)
            opt.step()
            hist.append(opt.loss.item())
        monkeypatch.undo()
        return hist

    a, b = run(False), run(True)
    assert a[-1] < 0.2 * a[0]
    # (not bit-equal: the subspace iteration is converged to ~1e-5, the SVD to rounding; the round-2 ordering
    # made the two runs differ by 15 % of the initial loss within ten steps)
    assert max(abs(p - q) for p, q in zip(a, b)) < 1e-3 * a[0]


@pytest.mark.parametrize("sym", [False, True])
@pytest.mark.parametrize("opt_name,lr,iters", [("RGD", 0.5, 400), ("RSGDwithMomentum", 0.3, 400), ("RiemannianAdam", 0.3, 800)])
def test_planted_tensor_is_recovered(sym, opt_name, lr, iters):
    """float64 planted rank-(2,3,3) tensor: every optimizer on both manifolds drives the squared error down by
    six orders of magnitude (Adam: three) from a random start (the geometry -- grad, project, round -- end to end)."""
    import r_tucker_amd as rt
    import test_riemannian as T
    mod = __import__("r_tucker_amd.model.%s.optim" % ("symmetric" if sym else "asymmetric"), fromlist=["x"])
    shape, rank = (5, 8, 8), (2, 3, 3)
    target = T.point(sym, seed=21, shape=shape, rank=rank).full()
    Model = rt.SymmetricR_TuckER if sym else rt.AsymmetricR_TuckER
    torch.manual_seed(9)
    model = Model((shape[1], shape[0]), rank).double()
    model.init()
    with torch.no_grad():
        model.core.copy_(torch.randn(rank, dtype=DT))
    if sym:
        params = torch.nn.ParameterList([model.core, model.E.weight, model.R.weight])
        extract = lambda: SFTucker(model.core.data, [model.R.weight], num_shared_factors=2, shared_factor=model.E.weight)  # noqa: E731
    else:
        params = torch.nn.ParameterList([model.core, model.S.weight, model.R.weight, model.O.weight])
        extract = lambda: Tucker(model.core.data, [model.R.weight, model.S.weight, model.O.weight])  # noqa: E731
    kw = {"momentum_beta": 0.5} if opt_name == "RSGDwithMomentum" else {}
    opt = getattr(mod, opt_name)(params, rank, lr, **kw)
    sched = torch.optim.lr_scheduler.ExponentialLR(opt, gamma=0.975 if opt_name == "RiemannianAdam" else 1.0)
    loss_fn = lambda T_: 0.5 * ((T_.full() - target) ** 2).sum()              # noqa: E731
    first = None
    for _ in range(iters):
        opt.fit(loss_fn, extract(), normalize_grad=False if opt_name != "RiemannianAdam" else 1.0)
        opt.step()
        sched.step()
        first = opt.loss.item() if first is None else first
    final = loss_fn(extract()).item()
    # (Adam here has a scalar second moment and a unit-length step: it anneals with the learning rate)
    assert final < (1e-3 if opt_name == "RiemannianAdam" else 1e-6) * first, (first, final)


@pytest.mark.parametrize("sym", [False, True])
def test_split_regulariser_equals_autodiff(sym):
    """``driver.RegularisedLoss``: the squared-norm term taken analytically by ``grad`` (delta_core += 2 c G, loss +=
    c ||G||^2) against the same loss differentiated as a whole through ``T.norm()``."""
    import test_riemannian as T
    from r_tucker_amd.driver import RegularisedLoss
    geo = SFTuckerRiemannian if sym else TuckerRiemannian
    x = T.point(sym)
    g = torch.Generator().manual_seed(5)
    A = torch.randn(x.full().shape, dtype=DT, generator=g)
    data = lambda t: 0.5 * ((t.full() - A) ** 2).sum()                         # noqa: E731
    whole = lambda t: data(t) + 0.37 * t.norm() ** 2                          # noqa: E731
    ga, la = geo.grad(RegularisedLoss(data, 0.37), x)
    gb, lb = geo.grad(whole, x)
    assert abs(la.item() - lb.item()) < 1e-10 * abs(lb.item())
    assert torch.allclose(ga.construct().full(), gb.construct().full(), atol=1e-10)
    assert abs(RegularisedLoss(data, 0.37)(x).item() - whole(x).item()) < 1e-12


def test_many_forms_equal_the_single_ones():
    """gram_factor_many / spd_inverse_many (equal sizes share one batched factorisation) and the batched subspace
    iteration against the one-matrix calls."""
    g = torch.Generator().manual_seed(5)
    mats = []
    for k in (5, 7, 5, 5, 7):
        w = torch.randn((3 * k, k), dtype=DT, generator=g)
        mats.append(w.T @ w)
    for (X, R), S in zip(sl.gram_factor_many(mats), mats):
        X1, R1 = sl.gram_factor(S)
        assert torch.equal(X, X1) and torch.equal(R, R1)
    for inv, S in zip(sl.spd_inverse_many(mats, 1e-10), mats):
        assert torch.equal(inv, sl.spd_inverse(S, 1e-10))
    M = torch.randn((3, 12, 40), dtype=DT, generator=g) * torch.logspace(0, -3, 12, dtype=DT)[None, :, None]
    Wb = sl.dominant_left_subspace(M, 4)
    for i in range(3):
        W1 = sl.dominant_left_subspace(M[i], 4)
        assert (Wb[i] @ Wb[i].T - W1 @ W1.T).abs().max().item() < 1e-12      # the same subspace


def test_round_is_the_same_with_and_without_grouped_factors(monkeypatch):
    """The S and O factors of the asymmetric model are truncated as one batch from the same core state (plain HOSVD
    among themselves); one after the other (sequentially truncated) must give a tensor as close to the untruncated
    one, and an equally orthonormal point."""
    import test_riemannian as T
    x = T.point(False, shape=(9, 30, 30), rank=(3, 6, 6))
    xi, _ = T.random_tangent(x, False, 11)
    for t in (1e-3, 0.3):
        moved = ((t * xi) + TuckerRiemannian.TangentVector(x)).construct()
        full = moved.full()
        errs = []
        for grouped in (True, False):
            monkeypatch.setattr(sl, "BATCH_EQUAL_SIZES", grouped)
            y = moved.round(x.rank)
            errs.append((y.full() - full).norm().item())
            for u in y.factors:
                assert (u.T @ u - torch.eye(u.shape[1], dtype=DT)).abs().max().item() < 1e-12
        assert abs(errs[0] - errs[1]) <= 1e-3 * max(errs) + 1e-12 * full.norm().item(), errs


def test_cholesky_qr_of_a_rank_deficient_fp32_block_cpu():
    """CPU counterpart of tests/test_gpu_driver.py::test_cholesky_qr_of_a_rank_deficient_fp32_block_stays_bounded: the
    Gram matrices of the Cholesky-QR rounds are accumulated in float64 (tall_gram_f64), so a numerically rank-deficient
    fp32 block keeps every pivot above the shift: finite, bounded factors and D = Q R."""
    from r_tucker_amd import tucker
    g = torch.Generator().manual_seed(7)
    n, k, rk = 5000, 60, 20
    D = (torch.randn(n, rk, generator=g) @ torch.randn(rk, k, generator=g)) * torch.logspace(-2, -6, k)
    D[:, 11] = 0.0
    S = sl.tall_gram_f64(D)
    assert S.dtype == torch.float64 and torch.allclose(S, D.double().T @ D.double(), rtol=1e-12, atol=0)
    Q, R = tucker._orth_tall(D)
    assert torch.isfinite(Q).all() and torch.isfinite(R).all() and Q.abs().max().item() < 50.0
    assert (Q[:, 11] == 0).all()
    err = torch.linalg.vector_norm(Q.double() @ R - D.double(), dim=0)
    ref = torch.linalg.vector_norm(D.double(), dim=0)
    assert (err <= 2e-3 * ref + 1e-12).all(), (err / ref.clamp_min(1e-30)).max().item()


def test_gram_factor_has_no_failure_path_on_the_cpu_either():
    """The CPU fallback mirrors the HIP kernel (csrc/rtk_chol.hip): an INDEFINITE "Gram" matrix (a pivot cancels) is
    floored instead of failing, a non-finite one gives zero outputs; a positive definite one is untouched."""
    torch.manual_seed(3)
    k = 12
    A = torch.randn(40, k, dtype=torch.float64)
    good = A.T @ A
    bad = good.clone()
    bad[3, 3] = -0.5 * good[3, 3]                       # indefinite: plain Cholesky breaks down here
    nan = good.clone()
    nan[5, 7] = float("nan")
    X, R = sl.gram_factor(torch.stack([good, bad, nan]))
    assert torch.isfinite(X).all() and torch.isfinite(R).all()
    assert torch.equal(X[2], torch.zeros_like(X[2])) and torch.equal(R[2], torch.zeros_like(R[2]))
    Xg, Rg = sl.gram_factor(good)
    assert torch.allclose(X[0], Xg) and torch.allclose(R[0], Rg)
    # the good one is a Cholesky-QR step: (A X) has orthonormal columns up to the shift
    Q = A @ X[0]
    assert (Q.T @ Q - torch.eye(k, dtype=torch.float64)).abs().max() < 1e-5
    assert X[1].abs().max() < 1e8                       # bounded, however bad the input
