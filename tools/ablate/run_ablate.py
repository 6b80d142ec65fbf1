#!/usr/bin/env python3
"""Where does the split-fp16 score kernel spend its time?  Runs the DBG build of the
kernel (tools/ablate/librtk_ablate.so) with parts switched off, interleaved rounds in
one process (cdna_hip_programming.md section 5.4 rule 24), C2 shape by default."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import gen  # noqa: E402
import r_tucker_amd as rt  # noqa: E402
from r_tucker_amd import _lib  # noqa: E402

lib = _lib.load()
abl = C.CDLL(os.path.join(ROOT, "tools", "ablate", "librtk_ablate.so"))
abl.rtk_ablate_score_packed_f32.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int64, C.c_void_p,
                                            C.c_int64, C.c_int, C.c_int, C.c_uint, C.c_void_p]
n_ent, n_rel, B, rank = 40943, 22, int(os.environ.get("B", 512)), (10, 200, 200)
a, b, c = rank
dev = torch.device("cuda:0")
core, R, S, O = [torch.from_numpy(x).to(dev) for x in gen.make_params(n_ent, n_rel, rank, 322)]
h, r = [torch.from_numpy(x).to(dev) for x in gen.make_queries(n_ent, n_rel, B, 1)]
ws = torch.zeros(lib.rtk_workspace_bytes(0, B, n_rel, a, b, c), dtype=torch.uint8, device=dev)
qp = torch.empty(lib.rtk_packed_query_bytes(0, B, c), dtype=torch.uint8, device=dev)
out = torch.empty((B, n_ent), dtype=torch.float32, device=dev)
sp = torch.cuda.current_stream().cuda_stream
_lib.check(lib.rtk_query_vectors_f32(core.data_ptr(), a, b, c, R.data_ptr(), n_rel, S.data_ptr(), n_ent, r.data_ptr(),
                                     h.data_ptr(), B, None, qp.data_ptr(), ws.data_ptr(), ws.numel(), sp), "qv")

variants = []
for grid in (256, 320, 512, 640, 1024):
    variants.append((f"full fast-sigmoid grid={grid}", 2, grid, 0))
variants += [("full exact-sigmoid grid=512", 1, 512, 0), ("logits grid=512", 0, 512, 0),
             ("no-staging(1)", 2, 512, 1), ("no-mfma(2)", 2, 512, 2), ("no-stores(4)", 2, 512, 4),
             ("prologue-only(8)", 2, 512, 8), ("no-prologue(16)", 2, 512, 16), ("no-prologue,no-stores(20)", 2, 512, 20),
             ("no-barrier(32)", 2, 512, 32), ("no-barrier,no-stores(36)", 2, 512, 36),
             ("no-staging,no-stores(5)", 2, 512, 5), ("no-staging,no-stores,no-barrier(37)", 2, 512, 37),
             ("only mfma+sigmoid (21)", 2, 512, 21), ("only mfma+sigmoid, no barrier (53)", 2, 512, 53),
             ("only mfma logits (21)", 0, 512, 21), ("only mfma logits no barrier (53)", 0, 512, 53),
             ("only staging logits (22)", 0, 512, 22), ("only sigmoid+stores (19)", 2, 512, 19)]
times = {v[0]: [] for v in variants}
for rnd in range(12):
    for name, sg, grid, dbg in variants:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = abl.rtk_ablate_score_packed_f32(qp.data_ptr(), B, c, O.data_ptr(), n_ent, out.data_ptr(), n_ent, sg, grid, dbg, sp)
        e1.record()
        assert rc == 0, rc
        torch.cuda.synchronize()
        if rnd >= 2:
            times[name].append(e0.elapsed_time(e1) * 1e3)
for name, ts in times.items():
    print(f"{name:45s} median {np.median(ts):8.1f} us   min {np.min(ts):8.1f} us")
