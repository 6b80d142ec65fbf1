#!/usr/bin/env python3
"""Where does the split-fp16 score kernel spend its time?  Runs compile-time-ablated
builds of the kernel (tools/ablate/librtk_ablate.so), interleaved rounds in one process
(cdna_hip_programming.md section 5.4 rule 24), C2 shape by default; then prints
per-block phase stamps of the full kernel."""
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
ablib = C.CDLL(os.path.join(ROOT, "tools", "ablate", "librtk_ablate.so"))
ablib.rtk_ablate_score_packed_f32.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int64, C.c_void_p,
                                              C.c_int64, C.c_int, C.c_int, C.c_uint, C.c_void_p]
ablib.rtk_ablate_read_stamps.argtypes = [C.c_void_p, C.c_int]
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


def launch(sg, grid, abl):
    rc = ablib.rtk_ablate_score_packed_f32(qp.data_ptr(), B, c, O.data_ptr(), n_ent, out.data_ptr(), n_ent, sg, grid, abl, sp)
    assert rc == 0, (rc, sg, grid, abl)


variants = []
for grid in (256, 320, 512, 640, 1024):
    variants.append((f"full fast-sigmoid grid={grid}", 2, grid, 0))
for grid in (256, 320, 512, 1024):
    variants.append((f"prologue-only(8) grid={grid}", 2, grid, 8))
variants += [("full exact-sigmoid grid=512", 1, 512, 0), ("logits grid=512", 0, 512, 0),
             ("no-staging(1)", 2, 512, 1), ("no-mfma(2)", 2, 512, 2), ("no-stores(4)", 2, 512, 4),
             ("no-prologue(16)", 2, 512, 16), ("no-prologue,no-stores(20)", 2, 512, 20),
             ("no-barrier(32)", 2, 512, 32),
             ("only mfma+sigmoid (21)", 2, 512, 21), ("only mfma logits (21)", 0, 512, 21),
             ("only staging logits (22)", 0, 512, 22), ("only sigmoid+stores (19)", 2, 512, 19)]
times = {v[0]: [] for v in variants}
for rnd in range(12):
    for name, sg, grid, abl in variants:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        launch(sg, grid, abl)
        e1.record()
        torch.cuda.synchronize()
        if rnd >= 2:
            times[name].append(e0.elapsed_time(e1) * 1e3)
for name, ts in times.items():
    print(f"{name:45s} median {np.median(ts):8.1f} us   min {np.min(ts):8.1f} us")

# per-block phase stamps (100 MHz realtime counter): start / first prologue done / end
for grid in (320, 512):
    for _ in range(3):
        launch(2, grid, 64)
    torch.cuda.synchronize()
    st = np.zeros(4096 * 4, dtype=np.uint64)
    assert ablib.rtk_ablate_read_stamps(st.ctypes.data, 4096 * 4) == 0
    st = st.reshape(4096, 4)[:grid].astype(np.int64)
    t0 = st[:, 0].min()
    start, pro, end = (st[:, 0] - t0) / 100.0, (st[:, 1] - st[:, 0]) / 100.0, (st[:, 2] - t0) / 100.0
    print(f"grid={grid}: block start  us: min {start.min():.1f} med {np.median(start):.1f} max {start.max():.1f}")
    print(f"           first prologue us: min {pro.min():.1f} med {np.median(pro):.1f} max {pro.max():.1f}")
    print(f"           block end    us: min {end.min():.1f} med {np.median(end):.1f} p90 {np.percentile(end, 90):.1f} max {end.max():.1f}")
    dur = end - start
    print(f"           block duration us: min {dur.min():.1f} med {np.median(dur):.1f} max {dur.max():.1f}")

# s_memtime stamps inside iteration 3 of every wave: loop top / loads issued / chain issued / staged / after barrier
ablib.rtk_ablate_read_istamps.argtypes = [C.c_void_p, C.c_int]
for grid, sg in ((256, 2), (512, 2), (512, 0)):
    for _ in range(3):
        launch(sg, grid, 128)
    torch.cuda.synchronize()
    st = np.zeros(4096 * 8, dtype=np.uint64)
    assert ablib.rtk_ablate_read_istamps(st.ctypes.data, 4096 * 8) == 0
    st = st.reshape(4096, 8)[: grid * 4].astype(np.int64)
    d = np.diff(st[:, :5], axis=1)
    print(f"grid={grid} sigmoid={sg}: cycles (median over waves)  load-issue {np.median(d[:,0]):.0f}  chain {np.median(d[:,1]):.0f}  "
          f"scale+stage-store {np.median(d[:,2]):.0f}  barrier {np.median(d[:,3]):.0f}  total {np.median(st[:,4]-st[:,0]):.0f}")
    print(f"          p90: load-issue {np.percentile(d[:,0],90):.0f} chain {np.percentile(d[:,1],90):.0f} stage {np.percentile(d[:,2],90):.0f} barrier {np.percentile(d[:,3],90):.0f}")
