#!/usr/bin/env python3
"""Times the stand-alone slot-order kernel (fp32, relation rank > 32 path launches it separately)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import gen, r_tucker_amd as rt
n_ent, n_rel, B, rank = 5000, 1000, 8192, (40, 32, 32)
core, R, S, O = [torch.from_numpy(x).cuda() for x in gen.make_params(n_ent, n_rel, rank, 1)]
h, r = [torch.from_numpy(x).cuda() for x in gen.make_queries(n_ent, n_rel, B, 1)]
for _ in range(10):
    v = rt.query_vectors(core, R, S, h, r)
torch.cuda.synchronize()
print(v.shape)
