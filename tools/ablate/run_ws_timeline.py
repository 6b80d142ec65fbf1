#!/usr/bin/env python3
"""Timeline of the wave-specialised score kernel (tools/ablate STAMP build): for M wave 0 and H wave 0 of every
workgroup, s_memtime at start, S1, conversion done, S2, every chain end / barrier exit, end.  Prints the median
over workgroups of each event (cycles from the workgroup's own start) and the per-iteration differences."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import r_tucker_amd as rt  # noqa: E402,F401
from r_tucker_amd import _lib, synthetic as gen  # noqa: E402

lib = _lib.load()
ab = C.CDLL(os.path.join(ROOT, "tools", "ablate", "librtk_ablate.so"))
ab.rtk_ablate_ws.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]
ab.rtk_ablate_ws_timeline.argtypes = [C.c_void_p, C.c_int, C.c_int]
n_ent, n_rel, B, rank = 40943, 22, 512, (10, 200, 200)
a, b, c = rank
dev = torch.device("cuda:0")
core, R, S, O = [torch.from_numpy(x).to(dev) for x in gen.make_params(n_ent, n_rel, rank, 322)]
h, r = [torch.from_numpy(x).to(dev) for x in gen.make_queries(n_ent, n_rel, B, 1)]
ws = torch.zeros(lib.rtk_workspace_bytes(0, B, n_rel, a, b, c), dtype=torch.uint8, device=dev)
qp = torch.empty(lib.rtk_packed_query_bytes(0, B, c), dtype=torch.uint8, device=dev)
LD = 40960
out = torch.empty((B, LD), dtype=torch.float32, device=dev)
sp = torch.cuda.current_stream().cuda_stream
_lib.check(lib.rtk_query_vectors_f32(core.data_ptr(), a, b, c, R.data_ptr(), n_rel, S.data_ptr(), n_ent, r.data_ptr(),
                                     h.data_ptr(), B, None, qp.data_ptr(), ws.data_ptr(), ws.numel(), sp), "qv")
XPS = [int(x) for x in os.environ.get('XPS', '0,2').split(',')]
BRIEF = os.environ.get('BRIEF') is not None
for xp in XPS:
    for _ in range(20):       # warm clocks and caches
        assert ab.rtk_ablate_ws(qp.data_ptr(), B, c, O.data_ptr(), n_ent, out.data_ptr(), LD, 256, xp, sp) == 0
    torch.cuda.synchronize()
    ab.rtk_ablate_ws_timeline(None, 0, 1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    assert ab.rtk_ablate_ws(qp.data_ptr(), B, c, O.data_ptr(), n_ent, out.data_ptr(), LD, 256, xp, sp) == 0
    e1.record()
    torch.cuda.synchronize()
    tl = np.zeros(256 * 2 * 64, dtype=np.uint64)
    assert ab.rtk_ablate_ws_timeline(tl.ctypes.data, tl.size, 0) == 0
    tl = tl.reshape(256, 2, 64)
    code = (tl >> np.uint64(56)).astype(np.int64)
    tm = (tl & np.uint64((1 << 56) - 1)).astype(np.int64)
    print(f"=== xp={xp} (bits: 2 helper epilogue off, 4 no gap work, 8 no A-fragment reads, 16 helpers idle): event time {e0.elapsed_time(e1) * 1e3:.1f} us")
    for role, name in ((0, "M wave 0"), (1, "H wave 0")):
        n_ev = int((code[:, role, :] != 0).sum(axis=1).min())
        t0 = tm[:, role, 0:1]
        rel = tm[:, role, :n_ev] - t0
        med = np.median(rel, axis=0)
        p90 = np.percentile(rel, 90, axis=0)
        codes = code[0, role, :n_ev]
        print(f"{name}: {n_ev} events; code: median cycles since start (delta) [p90]")
        if BRIEF:
            d = np.diff(med)
            steady = [d[k - 1] for k in range(1, n_ev) if codes[k] == 6][3:12]
            print(f"   S1 at {med[1]:.0f}, S2 at {med[3] if role == 0 else med[3]:.0f}, first chain end {med[4]:.0f}; steady chain/epilogue segments {np.round(steady)}; end {med[-1]:.0f}")
            continue
        prev = 0.0
        for k in range(n_ev):
            print(f"   {k:2d} code {codes[k]}: {med[k]:9.0f} (+{med[k] - prev:7.0f}) [{p90[k]:9.0f}]")
            prev = med[k]
    # skew between workgroups: start time spread
    st = tm[:, 0, 0]
    print("workgroup start spread (cycles): p50-p0", np.median(st) - st.min(), " p100-p0", st.max() - st.min())
