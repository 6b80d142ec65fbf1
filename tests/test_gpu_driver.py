"""The train.py / configs surface end to end on the GPU: the reference's command line,
Riemannian SGD with momentum differentiating the HIP loss at doubled rank, per-epoch evaluation in the
reference's dictionary format, checkpoints that load back."""
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"mrr", "hits@1", "hits@3", "hits@10"}


def _orthonormal(w):
    w = w.detach().double()
    return (w.T @ w - torch.eye(w.shape[1], dtype=torch.float64, device=w.device)).abs().max().item()


@pytest.mark.parametrize("mode,optim", [("asymmetric", "rsgd"), ("symmetric", "rsgd"), ("asymmetric", "adam"), ("symmetric", "rgd")])
def test_train_py_short_run(tmp_path, capsys, mode, optim):
    assert torch.cuda.is_available()
    import train
    from r_tucker_amd.utils.storage import StateDict
    state = train.main(["--mode", mode, "--optim", optim, "--seed", "322", "--data", os.path.join(ROOT, "data", "WN18RR") + "/",
                        "--config", "wn18rr_readme", "--epochs", "2", "--max-batches", "12", "--rank", "6", "24", "24",
                        "--checkpoint-path", str(tmp_path)])
    out = capsys.readouterr().out
    assert "Final mrr value:" in out and "Final hits@10 value:" in out
    assert len(state.losses.train) == 2 and len(state.metrics.mrr.test) == 2 and state.last_epoch == 2
    for hist in (state.losses.train, state.losses.val, state.losses.test, state.losses.norms):
        assert all(math.isfinite(float(x)) for x in hist)
    for m in (state.metrics.mrr, state.metrics.hits_1, state.metrics.hits_3, state.metrics.hits_10):
        assert all(0.0 <= x <= 1.0 for x in m.val + m.test)
    # (no claim on the trajectory here: with the README recipe the loss is dominated by 1e-4 * ||T||^2 at
    # unit-normalised steps of length 2000 in the first epochs; see test_descends_with_a_small_step)
    # the retraction keeps the factors orthonormal (fp32 Cholesky QR + float64 subspace iteration)
    for k, w in state.model.items():
        if k.endswith(".weight"):
            assert _orthonormal(w) < 5e-5, k
    assert set(state.model.keys()) == ({"core", "E.weight", "R.weight"} if mode == "symmetric"
                                       else {"core", "S.weight", "R.weight", "O.weight"})
    back = StateDict.load(os.path.join(str(tmp_path), "snapshot"))
    assert back.last_epoch == 2 and torch.equal(back.model["core"].cpu(), state.model["core"].cpu())
    assert os.path.exists(os.path.join(str(tmp_path), f"rk_24_final.pth"))


@pytest.mark.parametrize("mode", ["asymmetric", "symmetric"])
def test_riemannian_gradient_through_the_hip_loss_is_a_descent_direction(mode):
    """loss_fn of train.py:79 built on the HIP loss (CSR targets, doubled-rank construct differentiated by the
    native backward): a short step along -grad followed by the retraction lowers the loss by about
    t * |grad|^2, and the retracted point stays on the manifold."""
    import r_tucker_amd as rt
    from r_tucker_amd import driver
    from r_tucker_amd.data import Data, KG_dataset
    from r_tucker_amd.riemannian import SFTuckerRiemannian, TuckerRiemannian
    torch.manual_seed(5)
    data = Data(os.path.join(ROOT, "data", "WN18RR") + "/", reverse=True)
    train_set = KG_dataset(data, data.train_data, label_smoothing=0.1)
    flt = rt.DeviceFilter(train_set, "cuda")
    rank = (6, 24, 24)
    Model = rt.SymmetricR_TuckER if mode == "symmetric" else rt.AsymmetricR_TuckER
    model = Model((len(data.entities), len(data.relations)), rank)
    model.init()
    with torch.no_grad():
        model.core.mul_(2000.0)            # logits of order one: the loss surface has curvature to see
    model.cuda()
    geo = SFTuckerRiemannian if mode == "symmetric" else TuckerRiemannian
    ids = torch.arange(1000, 1512, device="cuda")
    f = flt.features[ids]
    loss_fn = driver.batch_loss_fn(model, f[:, 0].contiguous(), f[:, 1].contiguous(), flt, ids, 0.1, 1e-9)
    x = driver.extract_tensor(model)
    rgrad, loss0 = geo.grad(loss_fn, x)
    gn = rgrad.norm().item()
    assert gn > 0 and torch.isfinite(loss0)
    with torch.no_grad():
        drops = []
        for t in (0.02 * x.norm().item() / gn, 0.005 * x.norm().item() / gn):
            y = (((-t) * rgrad) + geo.TangentVector(rgrad.point)).construct().round(rank)
            drops.append((loss0 - loss_fn(y)).item() / (t * gn * gn))
            for w in ([y.shared_factor] + y.regular_factors if mode == "symmetric" else y.factors):
                w = w.double()
                assert (w.T @ w - torch.eye(w.shape[1], dtype=torch.float64, device=w.device)).abs().max().item() < 5e-5
    # first order: (loss(x) - loss(R(x - t g))) / (t |g|^2) -> 1 as t -> 0
    assert all(0.5 < d < 1.5 for d in drops), drops


def test_one_full_epoch_at_the_readme_rank(tmp_path, capsys):
    """One whole epoch of WN18RR (202 batches of 512 pairs, rank (10,200,200): the Riemannian gradient scores at
    core (20,400,400)) with the README recipe, then validation + test evaluation."""
    import json
    import train
    state = train.main(["--mode", "asymmetric", "--optim", "rsgd", "--seed", "322", "--data", os.path.join(ROOT, "data", "WN18RR") + "/",
                        "--config", "wn18rr_readme", "--epochs", "1", "--checkpoint-path", str(tmp_path)])
    out = capsys.readouterr().out
    rec = [json.loads(l) for l in out.splitlines() if l.startswith("{")][-1]
    assert KEYS <= {k[len("val_"):] for k in rec if k.startswith("val_")}
    assert rec["epoch"] == 1 and math.isfinite(rec["train_loss"]) and rec["grad_norm"] > 0
    assert rec["train_loss"] > 0.0
    print(f"\none WN18RR epoch: {rec['epoch_time']:.1f} s train, {rec['eval_time']:.2f} s test eval, "
          f"loss {rec['train_loss']:.5f}, val MRR {rec['val_mrr']:.4f}")
    assert len(state.metrics.mrr.val) == 1


def test_gram_factor_kernel_against_torch():
    """rtk_gram_factor_f64 (one workgroup per matrix: equilibration, shift, blocked Cholesky, the inverse from the
    same sweep, transposed outputs) vs torch in float64."""
    from r_tucker_amd import smalllinalg as sl
    g = torch.Generator(device="cuda").manual_seed(3)
    for k, nb in ((1, 2), (7, 3), (32, 1), (33, 2), (64, 1), (200, 3), (256, 2)):
        W = torch.randn(nb, 2 * k + 3, k, device="cuda", dtype=torch.float64, generator=g)
        W = W * torch.logspace(0, -4, k, device="cuda", dtype=torch.float64)          # badly scaled columns
        S = W.transpose(1, 2) @ W
        eye = torch.eye(k, device="cuda", dtype=torch.float64)
        X, R = sl.gram_factor(S, shift=0.0, equilibrate=False)
        ref = torch.linalg.cholesky(S).transpose(1, 2)
        assert (R - ref).abs().max().item() <= 1e-9 * ref.abs().max().item(), k
        assert (R.tril(-1).abs().max().item() == 0 and X.tril(-1).abs().max().item() == 0) or k == 1
        X, R = sl.gram_factor(S)                                                       # equilibrated + 1e-13 shift
        Q = W @ X
        assert (Q.transpose(1, 2) @ Q - eye).abs().max().item() < 1e-9, k
        assert (Q @ R - W).abs().max().item() <= 1e-9 * W.abs().max().item(), k
        inv = sl.spd_inverse(S, 1e-3)
        A = S + 1e-3 * S.diagonal(dim1=1, dim2=2).sum(-1)[:, None, None] * eye
        assert (inv @ A - eye).abs().max().item() < 1e-8, k
    # numerically singular and zero matrices: finite output, no status (pivots floored / zeros)
    W = torch.randn(40, 24, device="cuda", dtype=torch.float64, generator=g)
    W[:, 5] = W[:, 4]
    X, R = sl.gram_factor(W.T @ W)
    assert torch.isfinite(X).all() and torch.isfinite(R).all()
    X, R = sl.gram_factor(torch.zeros(9, 9, device="cuda", dtype=torch.float64))
    assert X.abs().max().item() == 0 and R.abs().max().item() == 0


def test_subspace_iteration_on_device_keeps_new_directions():
    """Dead directions in the old block, new directions 1e-4 of the largest and exactly orthogonal to the old row
    space: round 2's fp32 ``eigh`` of the squared spectrum could not tell them from noise (half of every step was
    lost from epoch 16 on) and its float64 fallback went through the host; the unsquared float64 iteration keeps
    them, on the device, without a synchronisation."""
    from r_tucker_amd import smalllinalg as sl
    g = torch.Generator(device="cuda").manual_seed(5)
    r, m = 12, 1600
    Va = torch.linalg.qr(torch.randn(m, 2 * r, device="cuda", dtype=torch.float64, generator=g))[0]
    sv = torch.cat([torch.logspace(4, 3, r - 3, device="cuda", dtype=torch.float64),
                    torch.full((3,), 1e-6, device="cuda", dtype=torch.float64)])
    A = (torch.linalg.qr(torch.randn(r, r, device="cuda", dtype=torch.float64, generator=g))[0] * sv) @ Va[:, :r].T
    Bm = torch.zeros(r, m, device="cuda", dtype=torch.float64)
    Bm[:3] = Va[:, r:r + 3].T
    mat = torch.cat([A, Bm])
    W = sl.dominant_left_subspace(mat, r)
    assert (W.T @ W - torch.eye(r, device="cuda", dtype=torch.float64)).abs().max().item() < 1e-9
    kept = ((W.T @ mat) ** 2).sum().item()
    best = (torch.linalg.svdvals(mat)[:r] ** 2).sum().item()
    assert abs(kept - best) <= 1e-6 * best
    assert (W[r:r + 3] ** 2).sum().item() > 2.999


def test_structured_round_at_the_wn18rr_shape():
    """The retraction of a tangent step at rank (10,200,200), 40 943 entities, fp32 on the device: new factors
    orthonormal, truncation error no worse than the generic QR + SVD path's."""
    import r_tucker_amd as rt
    from r_tucker_amd.riemannian import TuckerRiemannian as geo
    from r_tucker_amd.tucker import Tucker, read_health
    torch.manual_seed(1)
    model = rt.AsymmetricR_TuckER((40943, 22), (10, 200, 200))
    model.init()
    with torch.no_grad():
        model.core.mul_(50.0)
    model.cuda()
    x = Tucker(model.core.data, [model.R.weight.data, model.S.weight.data, model.O.weight.data])
    g = torch.Generator(device="cuda").manual_seed(2)
    Z = Tucker(torch.randn(10, 200, 200, device="cuda", generator=g),
               [torch.linalg.qr(torch.randn(n, k, device="cuda", generator=g))[0] for n, k in ((22, 10), (40943, 200), (40943, 200))])
    xi = geo.project(x, Z)
    xi = (x.norm().item() * 0.05 / xi.norm().item()) * xi
    moved = (xi + geo.TangentVector(x)).construct()
    read_health()
    y = moved.round((10, 200, 200))
    pre = max(read_health().values())
    assert pre < 1e-2                                                     # before the polish
    for w in y.factors:
        assert _orthonormal(w) < 5e-5
    y2 = Tucker(moved.core, moved.factors).round((10, 200, 200))

    def err(t):       # || t - moved || through Gram contractions of the difference (no dense tensor)
        d = t + ((-1.0) * moved)
        return d.norm().item()

    assert err(y) <= 1.01 * err(y2) + 1e-4 * moved.norm().item(), (err(y), err(y2))


def test_one_step_turns_the_subject_subspace_by_lr_over_core_scale():
    """What limits the README recipe under this geometry: the step is the NORMALISED gradient
    times lr, and the factor component of a tangent vector carries the inverse of the core's Gram matrix, so one step
    turns a factor's column space by ~ lr / sigma(core).  At the recipe's working point (core norm ~1e3 held down by
    the regulariser, lr 100-2000) that is a large angle per batch -- the model is a short-memory average of the last
    few batches; with the core 20x larger and lr 10 the same step is a small perturbation."""
    import r_tucker_amd as rt
    from configs.base_config import wn18rr_readme_config
    from r_tucker_amd import driver
    from r_tucker_amd.data import Data, KG_dataset
    data = Data(os.path.join(ROOT, "data", "WN18RR") + "/", reverse=True)
    train_set = KG_dataset(data, data.train_data, label_smoothing=0.1)
    flt = rt.DeviceFilter(train_set, "cuda")

    def turn(lr, core_norm):
        torch.manual_seed(322)
        cfg = wn18rr_readme_config()
        cfg.train_cfg.learning_rate = lr
        model = rt.AsymmetricR_TuckER((len(data.entities), len(data.relations)), cfg.model_cfg.manifold_rank)
        model.init()
        if core_norm is not None:
            with torch.no_grad():
                model.core.mul_(core_norm / float(model.core.norm()))
        model.cuda()
        opt = driver.define_optimizer(model, cfg, "asymmetric", "rsgd")
        U0 = model.S.weight.detach().clone()
        driver.train_one_epoch(model, opt, flt, 512, 0.1, regularization_coeff=3e-9, max_batches=1)
        U1 = model.S.weight.detach()
        assert _orthonormal(U1) < 5e-5
        return 1.0 - float(((U0.T @ U1) ** 2).sum()) / U0.shape[1]          # 1 - mean cos^2 of the principal angles

    big, small = turn(100.0, None), turn(10.0, 2.0e4)
    print(f"\nsubject subspace turned by sin^2 = {big:.3e} (lr 100, init core) / {small:.3e} (lr 10, |core| 2e4)")
    assert big > 1e-2 and small < 1e-5


def test_cholesky_qr_of_a_rank_deficient_fp32_block_stays_bounded():
    """The new block of a tangent step late in training: 200 fp32 columns of numerical rank well below 200 and
    very different sizes.  Its fp32 Gram matrix is indefinite at the 1e-5 level; the factorisation must neither
    overflow (|L^-1| reached 1e107 when cancelled pivots were floored at 1e-14 instead of at the shift) nor lose
    D = Q R."""
    from r_tucker_amd import tucker
    g = torch.Generator(device="cuda").manual_seed(7)
    n, k, rk = 40943, 200, 60
    D = (torch.randn(n, rk, device="cuda", generator=g) @ torch.randn(rk, k, device="cuda", generator=g))
    D = D * torch.logspace(-2, -6, k, device="cuda")
    D[:, 37] = 0.0                                                  # and a dead column
    Q, R = tucker._orth_tall(D)
    assert torch.isfinite(Q).all() and torch.isfinite(R).all()
    assert Q.abs().max().item() < 50.0 and (Q[:, 37] == 0).all()
    rec = Q.double() @ R
    err = torch.linalg.vector_norm(rec - D.double(), dim=0)
    ref = torch.linalg.vector_norm(D.double(), dim=0)
    assert (err <= 2e-3 * ref + 1e-12).all(), (err / ref.clamp_min(1e-30)).max().item()
    # the independent part is orthonormalised: the leading rk columns of Q (bar the dead one)
    keep = [j for j in range(rk) if j != 37]
    G = (Q[:, keep].double().T @ Q[:, keep].double())
    assert (G - torch.eye(rk - 1, device="cuda", dtype=torch.float64)).abs().max().item() < 1e-3
