#!/usr/bin/env python3
"""Stage-1 time (tables + contract kernels, one rtk_query_vectors call) at a bench shape."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import gen  # noqa: E402
import r_tucker_amd as rt  # noqa: E402,F401
from r_tucker_amd import _lib  # noqa: E402

lib = _lib.load()
n_ent, n_rel, B, rank = 40943, 22, 512, (10, 200, 200)
a, b, c = rank
dev = torch.device("cuda:0")
core, R, S, O = [torch.from_numpy(x).to(dev) for x in gen.make_params(n_ent, n_rel, rank, 322)]
h, r = [torch.from_numpy(x).to(dev) for x in gen.make_queries(n_ent, n_rel, B, 1)]
ws = torch.zeros(lib.rtk_workspace_bytes(0, B, n_rel, a, b, c), dtype=torch.uint8, device=dev)
qp = torch.empty(lib.rtk_packed_query_bytes(0, B, c), dtype=torch.uint8, device=dev)
sp = torch.cuda.current_stream().cuda_stream
ts = []
for _ in range(50):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _lib.check(lib.rtk_query_vectors_f32(core.data_ptr(), a, b, c, R.data_ptr(), n_rel, S.data_ptr(), n_ent, r.data_ptr(),
                                         h.data_ptr(), B, None, qp.data_ptr(), ws.data_ptr(), ws.numel(), sp), "qv")
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
print(f"RTK_CONTRACT={os.environ.get('RTK_CONTRACT')} LB={os.environ.get('RTK_CONTRACT_LB')}: stage 1 event time median {np.median(ts):.1f} us  min {np.min(ts):.1f}")
