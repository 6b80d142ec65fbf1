"""Asymmetric R-TuckER (distinct subject / object embeddings) on the HIP scoring path.

Drop-in for ``src/model/asymmetric/R_TuckER.py`` of the reference: same constructor
(``R_TuckER((n_ent, n_rel), rank, **kwargs)``, train.py:203), same parameter names
and ``state_dict`` keys (``core``, ``S.weight``, ``R.weight``, ``O.weight``), same
``init`` recipe, same ``forward(subject_idx, relation_idx) -> score_fn(T)`` closure
protocol.  Only the closure body differs: one call into the C ABI instead of five
torch ops.
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
        self.S = nn.Embedding(n_ent, rank[1])
        self.R = nn.Embedding(n_rel, rank[0])
        self.O = nn.Embedding(n_ent, rank[2])
        self.core = nn.Parameter(torch.zeros(tuple(rank), dtype=torch.float32))
        self.rank = rank
        self._tables_reset()

    def init(self, state_dict=None):
        """Load a state dict, or Xavier-initialise and orthonormalise the factor
        columns by thin QR (reference: R_TuckER.py:27-39)."""
        self._tables_reset()
        if state_dict:
            self.load_state_dict(state_dict)
            return
        nn.init.xavier_uniform_(self.core)
        with torch.no_grad():
            for emb in (self.S, self.R, self.O):   # same RNG draw order as the reference
                nn.init.xavier_normal_(emb.weight)
            for emb in (self.S, self.O, self.R):
                emb.weight.data = torch.linalg.qr(emb.weight)[0]

    def forward(self, subject_idx, relation_idx):
        def score_fn(T):
            # sizes, dtype and device come from T, never from self.rank: during training
            # T is the doubled-rank tangent-space construct
            tables = self._cached_tables(T.core, T.factors[0])
            return score_1vN(T.core, T.factors[0], T.factors[1], T.factors[2], subject_idx, relation_idx, tables=tables)

        return score_fn
