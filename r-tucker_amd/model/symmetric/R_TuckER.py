"""Symmetric (shared-factor) R-TuckER on the HIP scoring path.

Drop-in for ``src/model/symmetric/R_TuckER.py``: one entity matrix ``E`` serves as
subject gather source and as the 1-vs-all score operand; ``state_dict`` keys
``core``, ``E.weight``, ``R.weight``; ``score_fn`` reads ``T.regular_factors[0]``
and ``T.shared_factor`` (reference lines 40-44).
"""
from __future__ import annotations

import torch
from torch import nn

from ...ops import score_1vN
from .._tables import TablesCacheMixin


class R_TuckER(TablesCacheMixin, nn.Module):
    def __init__(self, data_count, rank=None, **kwargs):
        super().__init__()
        n_ent, n_rel = data_count
        self.E = nn.Embedding(n_ent, rank[1])
        self.R = nn.Embedding(n_rel, rank[0])
        self.core = nn.Parameter(torch.zeros(tuple(rank), dtype=torch.float32))
        self.rank = rank
        self._tables_reset()

    def init(self, state_dict=None):
        self._tables_reset()
        if state_dict:
            self.load_state_dict(state_dict)
            return
        nn.init.xavier_uniform_(self.core)
        with torch.no_grad():
            for emb in (self.E, self.R):
                nn.init.xavier_normal_(emb.weight)
            for emb in (self.E, self.R):
                emb.weight.data = torch.linalg.qr(emb.weight)[0]

    def forward(self, subject_idx, relation_idx):
        def score_fn(T):
            E = T.shared_factor
            tables = self._cached_tables(T.core, T.regular_factors[0])
            return score_1vN(T.core, T.regular_factors[0], E, E, subject_idx, relation_idx, tables=tables)

        return score_fn
