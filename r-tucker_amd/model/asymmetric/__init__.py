"""Asymmetric R-TuckER: ``R_TuckER`` (model, ``src/model/asymmetric/R_TuckER.py``) and ``optim``
(Riemannian optimizers, ``src/model/asymmetric/optim.py``)."""
from .R_TuckER import R_TuckER  # noqa: F401
