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
    # the retraction keeps the factors orthonormal (fp32 QR + SVD)
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


def test_core_basis_survives_an_eigensolver_failure(monkeypatch):
    """``round()`` takes the left basis of a core unfolding from ``eigh`` of its Gram matrix: in fp32 on the GPU
    while the kept directions are well above the noise floor of the squared spectrum, in float64 on the host
    otherwise or when rocSOLVER's divide-and-conquer does not converge (seen after 7 epochs of small-step RSGD on
    WN18RR), by SVD as the last resort."""
    from r_tucker_amd import tucker
    g = torch.Generator(device="cuda").manual_seed(5)
    flat = torch.randn(40, 1600, device="cuda", generator=g)
    want = tucker._truncated_left_basis(flat, 12)
    real = torch.linalg.eigh
    seen = []

    def device_fails(a, *args, **kw):
        seen.append((a.device.type, a.dtype))
        if a.is_cuda:
            raise torch.linalg.LinAlgError("forced: did not converge")
        return real(a, *args, **kw)

    def same_subspace(got, ref, tol):
        assert got.is_cuda and got.dtype == torch.float32
        assert (got.T @ got - torch.eye(got.shape[1], device="cuda")).abs().max().item() < 1e-5
        assert (got @ got.T - ref @ ref.T).abs().max().item() < tol

    monkeypatch.setattr(torch.linalg, "eigh", device_fails)
    tucker.FALLBACKS.clear()
    same_subspace(tucker._truncated_left_basis(flat, 12), want, 1e-3)
    assert seen == [("cuda", torch.float32), ("cpu", torch.float64)] and dict(tucker.FALLBACKS) == {"eigh_float64_host": 1}

    def always_fails(a, *args, **kw):
        raise torch.linalg.LinAlgError("forced")

    monkeypatch.setattr(torch.linalg, "eigh", always_fails)
    same_subspace(tucker._truncated_left_basis(flat, 12), want, 1e-3)
    assert tucker.FALLBACKS["svd"] == 1
    monkeypatch.setattr(torch.linalg, "eigh", real)

    # dead directions in the old block, new directions 1e-4 of the largest: fp32 on the squared spectrum cannot
    # tell them from noise, the float64 path keeps the new ones
    tucker.FALLBACKS.clear()
    r = 12
    Va = torch.linalg.qr(torch.randn(1600, 2 * r, device="cuda", generator=g))[0]
    sv = torch.cat([torch.logspace(4, 3, r - 3, device="cuda"), torch.full((3,), 1e-6, device="cuda")])
    A = (torch.linalg.qr(torch.randn(r, r, device="cuda", generator=g))[0] * sv) @ Va[:, :r].T
    Bm = torch.zeros(r, 1600, device="cuda")
    Bm[:3] = 1.0 * Va[:, r:r + 3].T                                   # three new directions with singular value 1
    mat = torch.cat([A, Bm])
    got = tucker._truncated_left_basis(mat, r)
    assert tucker.FALLBACKS["eigh_float64_host"] == 1
    kept = (got.T @ mat).norm() ** 2
    best = torch.linalg.svdvals(mat.double())[:r].pow(2).sum()
    assert abs(kept.item() - best.item()) <= 1e-6 * best.item()
    new_energy = (got[r:r + 3].norm() ** 2).item()                     # the three new coordinates are in the basis
    assert new_energy > 2.9
    tucker.FALLBACKS.clear()
