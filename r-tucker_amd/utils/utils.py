"""``set_random_seed`` and ``Timer`` of ``src/utils/utils.py`` (``:8-12``, ``:103-109``); the timer only
synchronises a GPU when there is one (the reference's calls ``torch.cuda.synchronize()`` even with
``--device cpu``, Appendix A4)."""
from __future__ import annotations

from time import perf_counter

import numpy as np
import torch


def set_random_seed(seed):
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    np.random.seed(seed)


class Timer:
    time = 0.0

    def __enter__(self):
        self.start = perf_counter()
        return self

    def __exit__(self, exc_type, exc_val, exc_tb):
        if torch.cuda.is_available() and torch.cuda.is_initialized():
            torch.cuda.synchronize()
        self.time = perf_counter() - self.start
        return False
