#!/usr/bin/env python3
"""Time rtk_gram_factor_f64 (csrc/rtk_chol.hip) on k x k Gram matrices: us per launch for a batch of 1 and of 4, and the
error of the factor against float64 torch.   usage: tools/chol_timing.py [k]      (RTK_CHOL_TUNE: phase ablations)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import r_tucker_amd as rt  # noqa: E402,F401
from r_tucker_amd import _lib  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 200
lib = _lib.load()
g = torch.Generator().manual_seed(5)
for nb in (1, 4):
    W = torch.randn((nb, 3000, k), generator=g, dtype=torch.float64)
    S = (W.transpose(1, 2) @ W).cuda().contiguous()
    R, X = torch.empty_like(S), torch.empty_like(S)
    sp = torch.cuda.current_stream().cuda_stream

    def launch():
        _lib.check(lib.rtk_gram_factor_f64(S.data_ptr(), nb, k, 1, 1e-13, 0.0, R.data_ptr(), X.data_ptr(), sp), "gram_factor")

    for _ in range(5):
        launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        launch()
    e1.record()
    torch.cuda.synchronize()
    err_r = ((R.transpose(1, 2) @ R - S).abs().max() / S.abs().max()).item()
    err_x = ((X.transpose(1, 2) @ S @ X - torch.eye(k, dtype=torch.float64, device="cuda")).abs().max()).item()
    print(f"k {k} batch {nb}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per launch   |R^T R - S| / |S| = {err_r:.1e}   |X^T S X - I| = {err_x:.1e}   tune {os.environ.get('RTK_CHOL_TUNE', '0')}")
