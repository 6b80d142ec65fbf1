#!/usr/bin/env python3
"""Time rtk_gram_factor_f64 at the sizes the optimizer step uses."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from r_tucker_amd import smalllinalg as sl
g = torch.Generator(device="cuda").manual_seed(1)
for k, nb in ((10, 1), (100, 1), (200, 1), (200, 2), (256, 1)):
    W = torch.randn(nb, 3 * k, k, device="cuda", dtype=torch.float64, generator=g)
    S = W.transpose(1, 2) @ W
    for _ in range(5):
        sl.gram_factor(S)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        sl.gram_factor(S)
    torch.cuda.synchronize()
    print(f"k={k} batch={nb}: {(time.perf_counter() - t0) / 50 * 1e6:.1f} us per call", flush=True)
