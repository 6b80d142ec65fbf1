"""r-tucker_amd: MI355X-native (gfx950) implementation of R-TuckER's 1-vs-all Tucker
scoring path -- hand-written HIP kernels behind a C ABI (``include/rtucker_hip.h``),
with a Python host that mirrors the reference's ``R_TuckER.forward -> score_fn(T)``
surface (``src/model/{asymmetric,symmetric}/R_TuckER.py`` in johanDDC/R-TuckER).

Nothing here falls back to CPU or to the oracle: without the HIP library, or with
non-GPU tensors, the scoring calls raise.
"""
from .tucker import Tucker, SFTucker  # noqa: F401
from .ops import (score_1vN, score_1vN_into, query_vectors, check_device_errors, bce_loss_1vN,  # noqa: F401
                  relation_tables, index_check, pack_query_vectors, score_packed_into)
from .sharded import EntityShards, ShardedEntityScorer  # noqa: F401
from . import _lib  # noqa: F401
from .evaluation import DeviceFilter, evaluate, filtered_ranks, metrics_from_ranks  # noqa: F401
from .model.asymmetric import R_TuckER as AsymmetricR_TuckER  # noqa: F401
from .model.symmetric import R_TuckER as SymmetricR_TuckER  # noqa: F401
from .riemannian import TuckerRiemannian, SFTuckerRiemannian, set_backend  # noqa: F401

__version__ = "0.1.0"
