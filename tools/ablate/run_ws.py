#!/usr/bin/env python3
"""Cycle stamps of the wave-specialised score kernel (tools/ablate build, STAMP = true)."""
import ctypes as C
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
ab = C.CDLL(os.path.join(ROOT, "tools", "ablate", "librtk_ablate.so"))
ab.rtk_ablate_ws.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]
ab.rtk_ablate_ws_stamps.argtypes = [C.c_void_p, C.c_int, C.c_int]
n_ent, n_rel, B, rank = 40943, 22, 512, (10, 200, 200)
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
for grid, xp in ((256, 0), (256, 2)):
    ts = []
    for _ in range(8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        assert ab.rtk_ablate_ws(qp.data_ptr(), B, c, O.data_ptr(), n_ent, out.data_ptr(), n_ent, grid, xp, sp) == 0
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    print(f"ws grid={grid} xp={xp} (1: M setprio 3, 2: H epilogue off): event time median {np.median(ts):.1f} us")
    ab.rtk_ablate_ws_stamps(None, 0, 1)
    torch.cuda.synchronize()
    assert ab.rtk_ablate_ws(qp.data_ptr(), B, c, O.data_ptr(), n_ent, out.data_ptr(), n_ent, grid, xp, sp) == 0
    torch.cuda.synchronize()
    st = np.zeros(256 * 64, dtype=np.uint64)
    assert ab.rtk_ablate_ws_stamps(st.ctypes.data, 256 * 64, 0) == 0
    st = st.reshape(256, 8, 8)[:grid].astype(np.int64)
    m, hh = st[:, :4, :], st[:, 4:, :]
    print("M iteration 5 (cycles): chain+handover", np.median(m[:, :, 1] - m[:, :, 0]), " barrier wait", np.median(m[:, :, 2] - m[:, :, 1]),
          " p90 barrier", np.percentile(m[:, :, 2] - m[:, :, 1], 90))
    print("M detail: loads issued", np.median(m[:, :, 3] - m[:, :, 0]), " to first staged write", np.median(m[:, :, 4] - m[:, :, 3]), " rest of chain", np.median(m[:, :, 5] - m[:, :, 4]), " tail (svp, prev)", np.median(m[:, :, 6] - m[:, :, 5]), " to barrier", np.median(m[:, :, 1] - m[:, :, 6]))
    print("H iteration 5 (cycles): epilogue", np.median(hh[:, :, 1] - hh[:, :, 0]), " stage_store", np.median(hh[:, :, 2] - hh[:, :, 1]),
          " stage_load", np.median(hh[:, :, 3] - hh[:, :, 2]), " barrier wait", np.median(hh[:, :, 4] - hh[:, :, 3]),
          " total", np.median(hh[:, :, 4] - hh[:, :, 0]))
    t0 = hh[:, :, 5].min()
    print("H realtime (us): first S2 reached", np.median(hh[:, :, 6] - t0) / 100.0, " H end", np.median(hh[:, :, 7] - t0) / 100.0,
          " max end", (hh[:, :, 7] - t0).max() / 100.0, " M end median", np.median(m[:, :, 7] - t0) / 100.0)
