"""Training / evaluation driver with the reference's surface (``train.py:20-167``): ``define_optimizer``,
``extract_tensor``, ``train_one_epoch``, ``evaluate``, ``train`` with the same arguments and return values
(loss / grad-norm averages, metric dictionaries ``{"mrr", "hits@1", "hits@3", "hits@10"}``, a ``StateDict``),
built MI355X-first on this package's pieces:

* no ``DataLoader`` and no dense ``(B, n_ent)`` target matrix: the ``(subject, relation) -> objects`` lists of a
  split live on the GPU once as a CSR (``evaluation.DeviceFilter``) and a batch is an index tensor; the loss
  term ``BCELoss(score_fn(T), targets)`` of ``train.py:79`` is ``ops.bce_loss_1vN`` (scores, label-smoothed BCE
  and d loss / d logits in HIP kernels; ``targets = (1 - eps) * multi_hot + eps / N`` as ``Dataset.py:51-52``);
* ``loss_fn(T) = BCE + regularization_coeff * T.norm() ** 2`` is differentiated by the Riemannian optimizer at
  the doubled-rank construct exactly as the reference's is (``optim.fit(loss_fn, x_k)``, then ``optim.step()``);
* evaluation is ``evaluation.evaluate``: HIP scores + on-device filtered ranking, relation tables cached.

The shuffle uses ``torch.randperm`` on the device with ``drop_last`` semantics (``train.py:227-228``).
"""
from __future__ import annotations

import torch
from torch import nn

from . import ops
from .evaluation import DeviceFilter, evaluate as _evaluate
from .tucker import SFTucker, Tucker
from . import tucker as _tucker
from .utils.storage import Losses, Metrics, StateDict
from .utils.utils import Timer


def _is_symmetric(model) -> bool:
    return hasattr(model, "E")


def define_optimizer(model, cfg, mode: str, opt: str):
    """``train.py:20-34``: the Riemannian optimizer over ``[core, S, R, O]`` (asymmetric) / ``[core, E, R]``."""
    if mode == "symmetric":
        from .model.symmetric.optim import RGD, RSGDwithMomentum, RiemannianAdam
        param_list = nn.ParameterList([model.core, model.E.weight, model.R.weight])
    else:
        from .model.asymmetric.optim import RGD, RSGDwithMomentum, RiemannianAdam
        param_list = nn.ParameterList([model.core, model.S.weight, model.R.weight, model.O.weight])
    rank, lr = cfg.model_cfg.manifold_rank, cfg.train_cfg.learning_rate
    if opt == "rsgd":
        return RSGDwithMomentum(param_list, rank, lr, cfg.train_cfg.momentum_beta)
    if opt == "rgd":
        return RGD(param_list, rank, lr)
    if opt == "adam":
        return RiemannianAdam(param_list, rank, lr, step_velocity=1)
    raise NotImplementedError("Such optimization method is not implemented")


def extract_tensor(model):
    """``train.py:37-42``: the live parameters as a Tucker / SFTucker (no copy)."""
    if _is_symmetric(model):
        return SFTucker(model.core.data, [model.R.weight], num_shared_factors=2, shared_factor=model.E.weight)
    return Tucker(model.core.data, [model.R.weight, model.S.weight, model.O.weight])


class RegularisedLoss:
    """``loss_fn`` of ``train.py:79``: ``T -> data_term(T) + coeff * T.norm() ** 2``, callable on any container like
    the reference's lambda.  It also says what it is made of (``riemannian_split``), which lets
    ``TuckerRiemannian.grad`` / ``SFTuckerRiemannian.grad`` take the squared-norm term analytically at a manifold
    point instead of differentiating it through the factor Gram matrices (``riemannian._split_loss``)."""

    def __init__(self, data_term, coeff):
        self.data_term, self.coeff = data_term, coeff
        self.riemannian_split = (data_term, coeff)

    def __call__(self, T):
        return self.data_term(T) + self.coeff * T.norm() ** 2


def batch_loss_fn(model, subject_idx, relation_idx, flt, item_ids, label_smoothing, regularization_coeff):
    """``loss_fn`` of ``train.py:79`` for one batch, as a function of the container ``T``.
    ``regularization_coeff``: a float or a 0-dim device tensor (the captured step keeps it in device memory)."""
    sym = _is_symmetric(model)

    def bce(T):
        if sym:
            core, R, S, O = T.core, T.regular_factors[0], T.shared_factor, T.shared_factor
        else:
            core, (R, S, O) = T.core, T.factors
        return ops.bce_loss_1vN(core, R, S, O, subject_idx, relation_idx, flt, item_ids, label_smoothing=label_smoothing)

    return RegularisedLoss(bce, regularization_coeff)


class EagerTrainStep:
    """One batch of ``train.py:76-85`` -- ``optimizer.fit(loss_fn, x_k); optimizer.step()`` -- on the ids handed to
    ``run``; the loss and gradient-norm sums that ``train_one_epoch`` reports are accumulated on the device (one
    synchronisation per epoch instead of the reference's two ``.item()`` per batch in its progress bar)."""

    def __init__(self, model, optimizer, flt, batch_size: int, label_smoothing: float, extract_tensor, batch_loss_fn):
        self.model, self.opt, self.flt = model, optimizer, flt
        self.B = int(batch_size)
        self.ls = float(label_smoothing)
        self._extract, self._loss_fn = extract_tensor, batch_loss_fn
        dev = flt.device
        self.dev = dev
        self.reg = torch.zeros((), dtype=torch.float32, device=dev)
        self.loss_sum = torch.zeros((), dtype=torch.float32, device=dev)
        self.gnorm_sum = torch.zeros((), dtype=torch.float32, device=dev)
        self.replays = 0

    def begin_epoch(self, regularization_coeff: float):
        self.reg.fill_(float(regularization_coeff))
        self.opt.refresh_lr()
        self.loss_sum.zero_()
        self.gnorm_sum.zero_()

    def run(self, ids: torch.Tensor):
        f = self.flt.features[ids]
        loss_fn = self._loss_fn(self.model, f[:, 0].contiguous(), f[:, 1].contiguous(), self.flt, ids, self.ls, self.reg)
        x_k = self._extract(self.model)
        gn = self.opt.fit(loss_fn, x_k)
        self.opt.step()
        self.loss_sum += self.opt.loss.detach().to(torch.float32)
        self.gnorm_sum += gn.detach().to(torch.float32)

    def totals(self):
        """(sum of losses, sum of gradient norms) over the steps since ``begin_epoch`` -- one synchronisation."""
        return float(self.loss_sum), float(self.gnorm_sum)


# what builds the per-batch step object (``begin_epoch`` / ``run(ids)`` / ``totals``); the HIP-graph replay of the step
# is an experiment outside the package (tools/graphstep.py: ``graphstep.install()`` swaps it in)
TRAIN_STEP_FACTORY = EagerTrainStep


def _captured_step(model, optimizer, train_flt, batch_size, label_smoothing):
    """The per-batch step object, kept on the optimizer so that later epochs reuse it (and its device buffers)."""
    key = (id(model), id(train_flt), int(batch_size), float(label_smoothing), TRAIN_STEP_FACTORY)
    cur = getattr(optimizer, "_rtk_captured", None)
    if cur is None or cur[0] != key:
        cur = (key, TRAIN_STEP_FACTORY(model, optimizer, train_flt, batch_size, label_smoothing, extract_tensor, batch_loss_fn))
        optimizer._rtk_captured = cur
    return cur[1]


def train_one_epoch(model, optimizer, train_flt: DeviceFilter, batch_size, label_smoothing, regularization_coeff=1e-4,
                    max_batches=None, log=None):
    """``train.py:69-91``: one pass over the (s, r) pairs of the train split; returns the mean loss and mean
    Riemannian gradient norm over the batches.  The batch step (``fit`` + ``step``) runs eagerly (``EagerTrainStep``);
    loss and gradient norm are summed on the device (one synchronisation per epoch instead of the reference's two
    ``.item()`` per batch in its progress bar)."""
    model.train()
    dev = train_flt.device
    n = train_flt.features.shape[0]
    n_batches = n // batch_size                         # drop_last=True
    if max_batches is not None:
        n_batches = min(n_batches, max_batches)
    perm = torch.randperm(n, device=dev)
    step = _captured_step(model, optimizer, train_flt, batch_size, label_smoothing)
    with ops.index_check("deferred"):                   # ids come from the dataset's own vocabulary: one check per epoch
        step.begin_epoch(regularization_coeff)
        for b in range(n_batches):
            step.run(perm[b * batch_size:(b + 1) * batch_size])
            optimizer.zero_grad(set_to_none=True)
            if log is not None and (b + 1) % 50 == 0:
                ls, gs = step.totals()
                log(f"  batch {b + 1}/{n_batches}: loss {ls / (b + 1):.6f}  grad norm {gs / (b + 1):.4e}")
        train_loss, train_grad_norm = step.totals()
    ops.check_device_errors(dev)
    denom = max(n_batches, 1)
    return train_loss / denom, train_grad_norm / denom


def evaluate(model, dataset, batch_size=512, flt: DeviceFilter = None):
    """``train.py:94-125``: ``(metrics dict averaged over the queries, mean BCE loss)``."""
    return _evaluate(model, dataset, batch_size=batch_size, flt=flt)


def train(model, optimizer, train_set, val_set, test_set, config, regulizer, scheduler=None, log=print, wandb_run=None,
          max_batches_per_epoch=None) -> StateDict:
    """``train.py:128-167``: the epoch loop with per-epoch validation / test evaluation, histories, checkpoints
    (``snapshot`` every epoch, ``rk_<rank>_<epoch>`` when the validation MRR improves by more than 5e-4)."""
    dev = next(model.parameters()).device
    timer = Timer()
    losses = Losses() if not config.state_dict else config.state_dict.losses
    metrics = Metrics() if not config.state_dict else config.state_dict.metrics
    tc = config.train_cfg
    start_epoch = 1 if not config.state_dict else config.state_dict.last_epoch
    train_flt = DeviceFilter(train_set, dev)
    val_flt, test_flt = DeviceFilter(val_set, dev), DeviceFilter(test_set, dev)
    prev_val_mrr = evaluate(model, val_set, tc.eval_batch_size, val_flt)[0]["mrr"]
    state = None
    for epoch in range(start_epoch, tc.num_epoches + start_epoch):
        regularization_coeff = regulizer.step()
        with timer:
            train_loss, train_norm = train_one_epoch(model, optimizer, train_flt, tc.train_batch_size, tc.label_smoothig,
                                                     regularization_coeff=regularization_coeff,
                                                     max_batches=max_batches_per_epoch)
        epoch_time = timer.time
        val_metrics, val_loss = evaluate(model, val_set, tc.eval_batch_size, val_flt)
        with timer:
            test_metrics, test_loss = evaluate(model, test_set, tc.eval_batch_size, test_flt)
        eval_time = timer.time
        metrics.update(val_metrics, "val")
        metrics.update(test_metrics, "test")
        losses.update(train_loss, train_norm, val_loss, test_loss)
        state = StateDict(model.state_dict(), losses, metrics, epoch, None,
                          scheduler.state_dict() if scheduler is not None else None)
        state.save(tc.checkpoint_path, "snapshot", add_epoch=False)
        if val_metrics["mrr"] - prev_val_mrr > 5e-4:
            prev_val_mrr = val_metrics["mrr"]
            state.save(tc.checkpoint_path, f"rk_{model.rank[1]}")
        lr = optimizer.param_groups[0]["lr"]
        if scheduler is not None:
            scheduler.step()
        record = {"epoch": epoch, "train_loss": train_loss, "val_loss": val_loss, "test_loss": test_loss,
                  "grad_norm": train_norm, "lr": lr, "reg_coeff": regularization_coeff, "epoch_time": epoch_time,
                  "eval_time": eval_time}
        for split, m in (("val", val_metrics), ("test", test_metrics)):
            for k, v in m.items():
                record[f"{split}_{k}"] = v
        health = _tucker.read_health(dev)     # largest pre-polish orthonormality error of a new factor this epoch
        if health:
            record["retraction_health"] = max(health.values())
        if wandb_run is not None:
            wandb_run.log(record)
        if log is not None:
            log(record)
    return state
