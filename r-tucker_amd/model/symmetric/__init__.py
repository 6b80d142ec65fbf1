"""Symmetric (shared-factor) R-TuckER: ``R_TuckER`` (model, ``src/model/symmetric/R_TuckER.py``) and
``optim`` (Riemannian optimizers, ``src/model/symmetric/optim.py``)."""
from .R_TuckER import R_TuckER  # noqa: F401
