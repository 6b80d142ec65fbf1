"""Container types the scoring closure reads (``T.core``, ``T.factors`` /
``T.regular_factors``, ``T.shared_factor``).

They take the constructor signatures the reference uses for
``tucker_riemopt.Tucker`` / ``tucker_riemopt.SFTucker`` (``train.py:39,41``) so that
``extract_tensor`` can build them unchanged; on the scoring path they are plain
attribute holders, exactly what ``score_fn`` needs
(``src/model/asymmetric/R_TuckER.py:43-47``, ``src/model/symmetric/R_TuckER.py:40-44``).
The Riemannian machinery of ``tucker_riemopt`` (round / project / grad) is outside
this package's scope.
"""
from __future__ import annotations

from typing import Sequence

import torch


class Tucker:
    """``X = core x_0 factors[0] x_1 factors[1] x_2 factors[2]`` with factors
    ``[R (nR,a), S (N,b), O (N,c)]`` and core axes (relation, subject, object)."""

    def __init__(self, core: torch.Tensor, factors: Sequence[torch.Tensor]):
        self.core = core
        self.factors = list(factors)

    @property
    def rank(self):
        return tuple(self.core.shape)


class SFTucker:
    """Shared-factor Tucker: the last ``num_shared_factors`` modes use ``shared_factor``."""

    def __init__(self, core: torch.Tensor, regular_factors: Sequence[torch.Tensor],
                 num_shared_factors: int, shared_factor: torch.Tensor):
        self.core = core
        self.regular_factors = list(regular_factors)
        self.num_shared_factors = num_shared_factors
        self.shared_factor = shared_factor

    @property
    def rank(self):
        return tuple(self.core.shape)
