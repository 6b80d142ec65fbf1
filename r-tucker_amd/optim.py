"""Riemannian optimizers on the Tucker manifolds: gradient descent, SGD with momentum, Adam.

Our own implementation of what ``src/model/asymmetric/optim.py`` and ``src/model/symmetric/optim.py``
do through ``tucker_riemopt`` (absent offline; ``riemannian.py`` supplies the geometry), with the same
class names, constructor arguments, ``fit(loss_fn, x_k, normalize_grad) -> grad norm`` / ``step()``
protocol, ``.loss`` / ``.direction`` attributes and ``param_groups[0]["lr"]`` handling
(``train.py:82-85``, ``:213-215`` attaches a torch LR scheduler to them).  One implementation serves both
manifolds; the per-model modules (``model/asymmetric/optim.py``, ``model/symmetric/optim.py``) bind the
parameter order of ``train.py:22-24``:  asymmetric ``[core, S, R, O]``, symmetric ``[core, E, R]``.

Defects of the reference not reproduced: the asymmetric ``RGD.step`` unpacks 3
of 4 parameters and uses shared-factor attributes (A2); ``RiemannianAdam`` is imported by ``train.py``
but defined nowhere (A1) -- here it is the algorithm of the symmetric module's ``SFTuckerAdam`` for
either manifold; ``SFTuckerAdam`` hard-codes ``device="cuda"`` (A5) -- the second moment lives on the
parameters' device.
"""
from __future__ import annotations

import os
from typing import Callable, Union

import torch
from torch.optim import Optimizer

from .riemannian import SFTuckerRiemannian, TuckerRiemannian
from .tucker import SFTucker, Tucker


def _bump(p: torch.Tensor) -> None:
    """Make a write through ``.data`` visible to autograd's version counter (the relation-table cache of
    the model closures keys on it)."""
    inc = getattr(torch.autograd.graph, "increment_version", None)
    if inc is not None:
        inc(p)


CHECK_FINITE = os.environ.get("R_TUCKER_AMD_CHECK_FINITE", "0") == "1"


def _check_finite(what, vec_or_tensor):
    """Debugging aid (``R_TUCKER_AMD_CHECK_FINITE=1``; one device sync per call): name the first non-finite
    piece of a tangent vector / container / tensor instead of failing later inside a factorisation."""
    if not CHECK_FINITE:
        return
    pieces = {}
    for name in ("delta_core", "delta_factors", "delta_regular_factors", "delta_shared_factor", "core", "factors",
                 "regular_factors", "shared_factor"):
        v = getattr(vec_or_tensor, name, None)
        if v is not None:
            pieces[name] = v
    if torch.is_tensor(vec_or_tensor):
        pieces = {"tensor": vec_or_tensor}
    for name, v in pieces.items():
        for i, t in enumerate(v if isinstance(v, (list, tuple)) else [v]):
            if not torch.isfinite(t).all():
                where = (~torch.isfinite(t)).nonzero()
                lo, hi = where.min(dim=0).values.tolist(), where.max(dim=0).values.tolist()
                raise FloatingPointError(f"{what}: {name}[{i}] of shape {tuple(t.shape)} has {where.shape[0]} non-finite "
                                         f"entries, index range {lo} .. {hi}, first {where[:4].tolist()}, "
                                         f"values {t[tuple(where[0].tolist())].item()}")


class _ManifoldOptimizer(Optimizer):
    """Shared machinery.  ``symmetric`` selects the manifold and the parameter order.

    State that survives a step (the previous direction of RSGD, Adam's first moment) is an EXPLICIT rank-2r tensor
    held in persistent buffers: it is built from the tangent vector BEFORE the parameters are overwritten in place
    (the point of a tangent vector aliases the parameter storage: ``extract_tensor`` wraps ``.data``,
    ``train.py:37-42``; the reference constructs before its ``W.data.add_`` too, ``asymmetric/optim.py:109-114``)
    and later steps ``copy_`` into the same storage, so a step captured into a HIP graph (``tools/graphstep.py``) reads
    and writes fixed addresses.  The learning rate is read from ``param_groups[0]["lr"]`` (a torch scheduler
    drives it, ``train.py:213-215``) into a device scalar outside of any capture."""

    symmetric = False
    capturable = True          # a step has no host-side state that changes from step to step

    def __init__(self, params, rank, max_lr, **extra):
        self.rank = tuple(rank)
        self.max_lr = max_lr
        self.lr = max_lr
        defaults = dict(rank=self.rank, max_lr=self.max_lr, lr=self.lr, **extra)
        super().__init__(params, defaults)
        self.direction = None
        self.loss = None
        self._lr_dev = None

    @property
    def geometry(self):
        return SFTuckerRiemannian if self.symmetric else TuckerRiemannian

    # ---- helpers ---------------------------------------------------------------------------------
    def _normalised(self, rgrad, rgrad_norm, normalize_grad):
        normalize_grad = rgrad_norm if not normalize_grad else normalize_grad
        # a vanishing gradient (e.g. exactly at an optimum) must not produce NaNs
        return (normalize_grad / torch.clamp(rgrad_norm, min=torch.finfo(rgrad_norm.dtype).tiny)) * rgrad

    def refresh_lr(self):
        """``param_groups[0]["lr"]`` -> the device scalar the step multiplies with (GPU parameters only).  Called
        by ``step()`` unless the stream is being captured; a captured step's owner calls it before each replay."""
        p = self.param_groups[0]["params"][0]
        if not p.is_cuda:
            return
        if self._lr_dev is None:
            self._lr_dev = torch.zeros((), dtype=p.dtype, device=p.device)
        self._lr_dev.fill_(float(self.param_groups[0]["lr"]))

    def _lr(self):
        p = self.param_groups[0]["params"][0]
        if not p.is_cuda:
            return self.param_groups[0]["lr"]
        if not torch.cuda.is_current_stream_capturing():
            self.refresh_lr()
        elif self._lr_dev is None:
            raise RuntimeError("run one eager step before capturing an optimizer step")
        return self._lr_dev

    @staticmethod
    def _tensors_of(t):
        if isinstance(t, SFTucker):
            return [t.core] + list(t.regular_factors) + [t.shared_factor]
        return [t.core] + list(t.factors)

    def _keep(self, slot: str, built):
        """Store the explicit tensor ``built`` in the persistent buffers of ``slot`` (allocated on first use)."""
        cur = getattr(self, slot, None)
        if cur is None:
            if built.core.is_cuda and torch.cuda.is_current_stream_capturing():
                raise RuntimeError("run one eager step before capturing an optimizer step")
            if isinstance(built, SFTucker):
                cur = SFTucker(built.core.clone(), [f.clone() for f in built.regular_factors], built.num_shared_factors,
                               built.shared_factor.clone())
            else:
                cur = Tucker(built.core.clone(), [f.clone() for f in built.factors])
            setattr(self, slot, cur)
        else:
            for dst, src in zip(self._tensors_of(cur), self._tensors_of(built)):
                dst.copy_(src)
        return cur

    @torch.no_grad()
    def _retract_and_write(self):
        lr = self._lr()
        x_k = self.direction.point
        _check_finite("direction before the step", self.direction)
        _check_finite("point before the step", x_k)
        moved = (-lr) * self.direction + self.geometry.TangentVector(x_k)
        _check_finite("moved tangent vector", moved)
        built = moved.construct()
        _check_finite("constructed point x - lr d", built)
        x_new = built.round(self.rank)
        _check_finite("rounded point", x_new)
        params = self.param_groups[0]["params"]
        if self.symmetric:
            W, E, R = params
            targets = ((W, x_new.core), (R, x_new.regular_factors[0]), (E, x_new.shared_factor))
        else:
            W, S, R, O = params
            targets = ((W, x_new.core), (R, x_new.factors[0]), (S, x_new.factors[1]), (O, x_new.factors[2]))
        for p, new in targets:
            p.data.copy_(new)
            _bump(p)
        # x_k aliases the parameter storage that was just overwritten: what was cached on the point object for the OLD
        # core (riemannian._point_grams: float64 Gram matrices) must not survive for a caller that reuses x_k
        if getattr(x_k, "_core_grams64", None) is not None:
            x_k._core_grams64 = None
        return x_new


class RGD(_ManifoldOptimizer):
    """Riemannian gradient descent (``asymmetric/optim.py:10-57``, ``symmetric/optim.py:11-59``)."""

    def fit(self, loss_fn: Callable[[Union[Tucker, SFTucker]], torch.Tensor], x_k,
            normalize_grad: Union[float, bool] = 1.):
        """Riemannian gradient of ``loss_fn`` at ``x_k``; returns its Frobenius norm.  ``normalize_grad``:
        ``False`` keeps the gradient's length, a float rescales it to that length."""
        rgrad, self.loss = self.geometry.grad(loss_fn, x_k)
        rgrad_norm = rgrad.norm().detach()
        self.direction = self._normalised(rgrad, rgrad_norm, normalize_grad)
        return rgrad_norm

    @torch.no_grad()
    def step(self, closure=None):
        self._retract_and_write()


class RSGDwithMomentum(_ManifoldOptimizer):
    """Riemannian SGD with momentum: the previous direction (kept as an explicit rank-2r tensor) is
    transported to the new point by projection (``asymmetric/optim.py:60-114``, ``symmetric/optim.py:62-107``)."""

    def __init__(self, params, rank, max_lr, momentum_beta=0.9):
        super().__init__(params, rank, max_lr, momentum_beta=momentum_beta)
        self.momentum_beta = momentum_beta
        self.momentum = None
        self._prev = None              # the previous direction as an explicit tensor (persistent buffers)

    def fit(self, loss_fn, x_k, normalize_grad: Union[float, bool] = 1.):
        geo = self.geometry
        if self._prev is not None:
            self.momentum = geo.project(x_k, self._prev)
        else:
            self.momentum = geo.TangentVector(x_k, torch.zeros_like(x_k.core))
        _check_finite("momentum projected to the new point", self.momentum)
        rgrad, self.loss = geo.grad(loss_fn, x_k)
        _check_finite("loss", self.loss)
        _check_finite("Riemannian gradient", rgrad)
        rgrad_norm = rgrad.norm().detach()
        _check_finite("gradient norm", rgrad_norm)
        self.direction = self._normalised(rgrad, rgrad_norm, normalize_grad) + self.momentum_beta * self.momentum
        return rgrad_norm

    @torch.no_grad()
    def step(self, closure=None):
        built = self.direction.construct()      # BEFORE the write: its factors [U, dU] use the point it belongs to
        self._retract_and_write()
        self.direction = self._keep("_prev", built)   # explicit tensor: projected at the next point


class RiemannianAdam(_ManifoldOptimizer):
    """Adam with a scalar second moment (the squared gradient norm), first moment transported by projection:
    the algorithm of ``SFTuckerAdam`` (``symmetric/optim.py:110-167``) on either manifold."""

    capturable = False         # the bias correction uses a host-side step counter

    def __init__(self, params, rank, max_lr, betas=(0.9, 0.999), eps=1e-8, step_velocity=1):
        super().__init__(params, rank, max_lr, betas=betas, eps=eps, step_velocity=step_velocity)
        self.betas = betas
        self.eps = eps
        self.step_velocity = step_velocity
        self.momentum = None
        self._m1 = None                # the first moment as an explicit tensor (persistent buffers)
        self.second_momentum = None
        self.step_t = 1

    def fit(self, loss_fn, x_k, normalize_grad: Union[float, bool] = 1.):
        geo = self.geometry
        rgrad, self.loss = geo.grad(loss_fn, x_k)
        rgrad_norm = rgrad.norm().detach()
        b1, b2 = self.betas
        if self._m1 is not None:
            self.momentum = b1 * geo.project(x_k, self._m1) + (1 - b1) * rgrad
        else:
            self.momentum = (1 - b1) * rgrad
        if self.second_momentum is None:
            self.second_momentum = torch.zeros((), device=rgrad_norm.device, dtype=rgrad_norm.dtype)
        self.second_momentum = b2 * self.second_momentum + (1 - b2) * rgrad_norm ** 2
        t = self.step_t // self.step_velocity + 1
        second_corrected = self.second_momentum / (1 - b2 ** t)
        ratio = (1 - b1 ** t) * torch.sqrt(second_corrected) + self.eps
        self.direction = (1 / ratio) * self.momentum
        return rgrad_norm

    @torch.no_grad()
    def step(self, closure=None):
        built = self.momentum.construct()       # the first moment belongs to the point about to be overwritten
        self._retract_and_write()
        self._keep("_m1", built)
        self.step_t += 1
