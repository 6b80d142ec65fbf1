#!/usr/bin/env python3
"""Timeline of the column-group score kernel (tools/ablate/build_cg_variant.sh stamps -DRTK_CG_STAMPS build): for M wave 0 and H wave 0 of
every workgroup, s_memtime at the prologue's barriers and at every iteration's barrier; prints the median over the
workgroups of each event (cycles from the workgroup's own start) and the differences.
    R_TUCKER_AMD_LIB=tools/ablate/librtk_cg_stamps.so python tools/ablate/run_cg_timeline.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("R_TUCKER_AMD_LIB", os.path.join(ROOT, "tools", "ablate", "librtk_cg_stamps.so"))
import r_tucker_amd as rt  # noqa: E402,F401
from r_tucker_amd import _lib, synthetic as gen  # noqa: E402

lib = _lib.load()
raw = C.CDLL(os.environ["R_TUCKER_AMD_LIB"])
raw.rtk_cg_timeline.argtypes = [C.c_void_p, C.c_int, C.c_int]
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
flags = _lib.RTK_SCORE_SIGMOID | _lib.RTK_SCORE_SIGMOID_FAST | _lib.RTK_SCORE_KERNEL_CG


def launch():
    _lib.check(lib.rtk_score_packed_f32(qp.data_ptr(), B, c, O.data_ptr(), n_ent, out.data_ptr(), LD, flags, sp), "score")


t_end = __import__("time").perf_counter() + 1.0
while __import__("time").perf_counter() < t_end:      # warm clocks and caches
    for _ in range(50):
        launch()
    torch.cuda.synchronize()
assert raw.rtk_cg_timeline(None, 0, 1) == 0
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
launch()
e1.record()
torch.cuda.synchronize()
tl = np.zeros(256 * 2 * 64, dtype=np.uint64)
assert raw.rtk_cg_timeline(tl.ctypes.data, tl.size, 0) == 0
tl = tl.reshape(256, 2, 64)
code = (tl >> np.uint64(56)).astype(np.int64)
tm = (tl & np.uint64((1 << 56) - 1)).astype(np.int64)
print(f"event-bracketed launch (stamped build): {e0.elapsed_time(e1) * 1e3:.1f} us")
NAMES = {0: {1: "start", 2: "loads landed, partial maxima written", 7: "P1 passed, fifth range converted", 3: "own group converted", 4: "S2 passed", 6: "chain + pieces done", 5: "barrier passed"},
         1: {1: "start", 2: "query tile 0 written", 3: "S1 passed", 4: "S2 passed", 9: "DMA of tile i+2 issued", 10: "row factors copied", 11: "LDS reads requested", 8: "fifth group summed",
             6: "stores issued", 5: "barrier passed"}}
BRIEF = os.environ.get("BRIEF") is not None
# in-kernel clock: shader cycles between the first and the last s_memtime stamp of M wave 0 over the 100 MHz
# s_memrealtime ticks between the two realtime stamps around them
rt0 = np.array([tm[w, 0, 0] for w in range(256)])
rtn = np.array([tm[w, 0, (code[w, 0, :] == 13).argmax()] for w in range(256)])
c0 = np.array([tm[w, 0, 1] for w in range(256)])
cn = np.array([tm[w, 0, (code[w, 0, :] == 13).argmax() - 1] for w in range(256)])
print(f"in-kernel: {np.median(rtn - rt0) / 100:.2f} us of 100 MHz ticks, {np.median(cn - c0):.0f} shader cycles -> clock {np.median((cn - c0) / np.maximum(rtn - rt0, 1)) * 0.1:.3f} GHz")
tm[:, 0, :-1] = tm[:, 0, 1:]          # drop the realtime stamp at the head of M wave 0's list
code[:, 0, :-1] = code[:, 0, 1:]
# the two roles' clocks against each other: s_memtime of M wave 0's first stamp minus S wave 0's, per workgroup (a wave
# reaches its first stamp when the code of its role has been fetched)
print(f"M wave 0 start - S wave 0 start (cycles): median {np.median(tm[:, 0, 0] - tm[:, 1, 0]):.0f}  "
      f"[p10 {np.percentile(tm[:, 0, 0] - tm[:, 1, 0], 10):.0f} .. p90 {np.percentile(tm[:, 0, 0] - tm[:, 1, 0], 90):.0f}]")
raw63 = tl[:, :, 63].astype(np.int64)
if raw63 is not None and raw63.min() > 0:
    for role, name in ((0, "M wave 0"), (1, "S wave 0")):
        d = tm[:, role, 0] - raw63[:, role]
        print(f"{name}: kernel entry -> first stamp of the role: median {np.median(d):.0f} cycles [p10 {np.percentile(d, 10):.0f} .. p90 {np.percentile(d, 90):.0f}]")
    print(f"entry of M wave 0 - entry of S wave 0: median {np.median(raw63[:, 0] - raw63[:, 1]):.0f} cycles")
for role, name in ((0, "M wave 0"), (1, "S wave 0")):
    n_ev = int(((code[:, role, :] != 0) & (code[:, role, :] != 13)).sum(axis=1).min())
    t0 = tm[:, role, 0:1]
    rel = tm[:, role, :n_ev] - t0
    med = np.median(rel, axis=0)
    p10 = np.percentile(rel, 10, axis=0)
    p90 = np.percentile(rel, 90, axis=0)
    codes = code[0, role, :n_ev]
    print(f"{name}: {n_ev} events; median cycles since start (delta) [p10 .. p90]")
    if BRIEF:      # one steady-state iteration: the events between two "barrier passed" in the middle of the sweep
        bp = [k for k in range(n_ev) if codes[k] == 5]
        a, b = bp[len(bp) // 2], bp[len(bp) // 2 + 2]
        print(f"   iteration period {(med[b] - med[a]) / 2:.0f}; end {med[-1]:.0f}; S2 at {med[[k for k in range(n_ev) if codes[k] == 4][0]]:.0f}")
        for k in range(a, b + 1):
            print(f"      {str(NAMES[role].get(int(codes[k]), codes[k])):28s} +{med[k] - med[k - 1]:6.0f}")
        continue
    prev = 0.0
    for k in range(n_ev):
        print(f"   {k:2d} {str(NAMES[role].get(int(codes[k]), codes[k])):34s}: {med[k]:9.0f} (+{med[k] - prev:7.0f}) [{p10[k]:8.0f} .. {p90[k]:8.0f}]")
        prev = med[k]
st = tm[:, 0, 0]
print("workgroup start spread (cycles): p50-p0", np.median(st) - st.min(), " p100-p0", st.max() - st.min())
en = tm[:, 0, :].max(axis=1)
print("workgroup end spread (cycles): p50-p0", np.median(en) - en.min(), " p100-p0", en.max() - en.min(), " first start -> last end", en.max() - st.min())
