"""Relation-table cache shared by both model flavours.

``M_r = G x_0 R[r]`` (the einsum of ``R_TuckER.py:45``) is a function of the parameters only.  The
reference's ``evaluate()`` (train.py:94-125) runs ``model.eval()`` under ``torch.no_grad()`` and scores
every batch of the split with the SAME ``extract_tensor(model)``; the closure of a model in that state
builds the tables of all relations once and every following batch only does the subject-mode
contraction.  The cache is dropped whenever the parameters can have changed:

* ``model.train()`` / ``model.eval()`` (the reference switches mode around every training epoch),
  ``init``, ``load_state_dict``, ``.to()`` and friends (``_apply``);
* an in-place update of ``core`` / ``R.weight`` that autograd's version counter sees.

NOT seen: writes through ``.data`` while the model stays in eval mode (``W.data.add_(...)`` does not
bump ``W._version``; the reference's optimizers write that way, but only between ``model.train()`` and
the next ``model.eval()``).  Call ``model.invalidate_tables()`` after such a write, or set
``model.cache_tables = False``.
"""
from __future__ import annotations

import torch


class TablesCacheMixin:
    cache_tables = True

    def _tables_reset(self):
        self.__dict__["_tables"] = None
        self.__dict__["_tables_key"] = None

    def invalidate_tables(self):
        self._tables_reset()

    def train(self, mode: bool = True):
        self._tables_reset()
        return super().train(mode)

    def _apply(self, fn, *args, **kwargs):
        self._tables_reset()
        return super()._apply(fn, *args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        self._tables_reset()
        return super().load_state_dict(*args, **kwargs)

    def _cached_tables(self, core, R):
        """Tables for ``(core, R)`` if they are this model's own parameters in a frozen state, else None."""
        if (not self.cache_tables or self.training or torch.is_grad_enabled() or not core.is_cuda
                or core.data_ptr() != self.core.data_ptr() or R.data_ptr() != self.R.weight.data_ptr()
                or tuple(core.shape) != tuple(self.core.shape) or core.dtype != self.core.dtype
                or torch.cuda.is_current_stream_capturing()):
            return None
        key = (self.core._version, self.R.weight._version, self.core.data_ptr(), self.R.weight.data_ptr(),
               self.core.dtype, self.core.device)
        if self.__dict__.get("_tables") is None or self.__dict__.get("_tables_key") != key:
            from ..ops import relation_tables
            self.__dict__["_tables"] = relation_tables(self.core.detach(), self.R.weight.detach())
            self.__dict__["_tables_key"] = key
        return self.__dict__["_tables"]
