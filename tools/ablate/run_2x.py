#!/usr/bin/env python3
"""Score kernels at the doubled rank the Riemannian gradient evaluates (WN18RR: core (20,400,400), B 512)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import gen, r_tucker_amd as rt
n_ent, n_rel, B, rank = 40943, 22, 512, (20, 400, 400)
core, R, S, O = [torch.from_numpy(x).cuda() for x in gen.make_params(n_ent, n_rel, rank, 322)]
h, r = [torch.from_numpy(x).cuda() for x in gen.make_queries(n_ent, n_rel, B, 1)]
for exact in (False, True):
    ts = []
    for _ in range(20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        p = rt.score_1vN(core, R, S, O, h, r, exact=exact)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    print(f"exact={exact}: whole call {np.median(ts):.1f} us")
