"""Riemannian optimizers of the symmetric model -- the names ``train.py:187`` imports from
``src.model.symmetric.optim`` (plus ``SFTuckerAdam``, the name that module defines); parameter order
``[core, E.weight, R.weight]`` (``train.py:22``)."""
from ...optim import RGD as _RGD, RSGDwithMomentum as _RSGD, RiemannianAdam as _Adam


class RGD(_RGD):
    symmetric = True


class RSGDwithMomentum(_RSGD):
    symmetric = True


class RiemannianAdam(_Adam):
    symmetric = True


SFTuckerAdam = RiemannianAdam
