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


def test_descends_with_a_small_step(tmp_path, capsys):
    """Without the norm penalty and with a moderate step the BCE part goes down epoch over epoch."""
    import train
    state = train.main(["--mode", "asymmetric", "--optim", "rsgd", "--seed", "1", "--data", os.path.join(ROOT, "data", "WN18RR") + "/",
                        "--config", "wn18rr_readme", "--epochs", "3", "--max-batches", "20", "--rank", "6", "24", "24",
                        "--checkpoint-path", str(tmp_path), "--set", "train_cfg.learning_rate=300",
                        "--set", "train_cfg.base_regularization_coeff=1e-30", "--set", "train_cfg.final_regularization_coeff=1e-31",
                        "--set", "train_cfg.scheduler_step=1.0"])
    capsys.readouterr()
    tl = state.losses.train
    assert tl[0] < 0.6932 and tl[2] < tl[1] < tl[0], tl                  # starts at ln 2 (all scores 0.5) and descends


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
