#!/usr/bin/env python3
"""Event trace (cycle stamps) of the two-tiles-per-barrier score kernel (tools/ablate build, STAMP = true)."""
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
ab.rtk_ablate_ws2.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
ab.rtk_ablate_ws2_stamps.argtypes = [C.c_void_p, C.c_int, C.c_int]
n_ent, n_rel, B, rank = 40943, 22, 512, (10, 200, 200)
a, b, c = rank
dev = torch.device("cuda:0")
core, R, S, O = [torch.from_numpy(x).to(dev) for x in gen.make_params(n_ent, n_rel, rank, 322)]
h, r = [torch.from_numpy(x).to(dev) for x in gen.make_queries(n_ent, n_rel, B, 1)]
ws = torch.zeros(lib.rtk_workspace_bytes(0, B, n_rel, a, b, c), dtype=torch.uint8, device=dev)
qp = torch.empty(lib.rtk_packed_query_bytes(0, B, c), dtype=torch.uint8, device=dev)
LD = int(os.environ.get("LD_OUT", n_ent))
out = torch.empty((B, LD), dtype=torch.float32, device=dev)
print("ld_out", LD)
sp = torch.cuda.current_stream().cuda_stream
_lib.check(lib.rtk_query_vectors_f32(core.data_ptr(), a, b, c, R.data_ptr(), n_rel, S.data_ptr(), n_ent, r.data_ptr(),
                                     h.data_ptr(), B, None, qp.data_ptr(), ws.data_ptr(), ws.numel(), sp), "qv")
NS = 96
ts = []
for _ in range(8):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    assert ab.rtk_ablate_ws2(qp.data_ptr(), B, c, O.data_ptr(), n_ent, out.data_ptr(), LD, 256, sp) == 0
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
print(f"ws2 (stamp build) event time median {np.median(ts):.1f} us")
assert ab.rtk_ablate_ws2_stamps(None, 0, 1) == 0
torch.cuda.synchronize()
assert ab.rtk_ablate_ws2(qp.data_ptr(), B, c, O.data_ptr(), n_ent, out.data_ptr(), LD, 256, sp) == 0
torch.cuda.synchronize()
st = np.zeros(256 * 8 * NS, dtype=np.uint64)
assert ab.rtk_ablate_ws2_stamps(st.ctypes.data, st.size, 0) == 0
st = st.reshape(256, 8, NS)
ids = (st & np.uint64(0xFF)).astype(np.int64)
tt = (st >> np.uint64(8)).astype(np.int64)
for wg in (100,):
    for wave in (0, 4):
        n = int((ids[wg, wave] > 0).sum())
        t0 = tt[wg, wave, 0]
        print(f"wg {wg} wave {wave} ({'M' if wave < 4 else 'H'}): " + " ".join(f"{ids[wg, wave, k]}:{tt[wg, wave, k] - t0}" for k in range(n)))
# aggregate: durations between consecutive events, keyed by (from id, to id)
for wave, name in ((0, "M"), (4, "H")):
    agg = {}
    for wg in range(256):
        n = int((ids[wg, wave] > 0).sum())
        for k in range(n - 1):
            agg.setdefault((int(ids[wg, wave, k]), int(ids[wg, wave, k + 1])), []).append(int(tt[wg, wave, k + 1] - tt[wg, wave, k]))
    print(name, "median cycles between events:", {k: (int(np.median(v)), len(v)) for k, v in sorted(agg.items())})
