"""Riemannian optimizers of the asymmetric model -- the names ``train.py:191`` imports from
``src.model.asymmetric.optim``; parameter order ``[core, S.weight, R.weight, O.weight]`` (``train.py:24``)."""
from ...optim import RGD as _RGD, RSGDwithMomentum as _RSGD, RiemannianAdam as _Adam


class RGD(_RGD):
    symmetric = False


class RSGDwithMomentum(_RSGD):
    symmetric = False


class RiemannianAdam(_Adam):
    symmetric = False
