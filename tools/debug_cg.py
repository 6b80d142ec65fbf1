#!/usr/bin/env python3
"""Localise a wrong result of the column-group score kernel: per (query tile, column group of a set) the largest
difference to the ws kernel on the same packed planes.  usage: tools/debug_cg.py N c B [flags]"""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import r_tucker_amd as rt

N, c, B = [int(x) for x in sys.argv[1:4]]
extra = int(sys.argv[4], 0) if len(sys.argv) > 4 else 0
L = rt._lib
lib = L.load()
g = torch.Generator().manual_seed(1)
v = torch.randn((B, c), generator=g)
O = torch.randn((N, c), generator=g)
if os.environ.get("SCALES"):      # per-row powers of two, like tests/test_gpu_score_cg.py
    v = v * torch.exp2(torch.randint(-6, 7, (B, 1), generator=g).float())
    O = O * torch.exp2(torch.randint(-6, 7, (N, 1), generator=g).float())
v, O = v.cuda(), O.cuda()
qp = rt.pack_query_vectors(v, torch.float32)


def score(flags):
    buf = torch.full((B, N), -7.0, dtype=torch.float32, device="cuda")
    L.check(lib.rtk_score_packed_f32(qp.data_ptr(), B, c, O.data_ptr(), N, buf.data_ptr(), N, flags | extra,
                                     torch.cuda.current_stream().cuda_stream), "score")
    torch.cuda.synchronize()
    return buf.cpu().numpy()


cg, ws = score(L.RTK_SCORE_KERNEL_CG), score(L.RTK_SCORE_KERNEL_WS)
z = (v.double() @ O.double().T).cpu().numpy()
if extra & 1:
    z = 1 / (1 + np.exp(-z))
print("ws vs f64", np.abs(ws - z).max(), " cg vs f64", np.abs(cg - z).max())
G = -(-N // 32)
sets_min = -(-G // 5)
W = min(256, sets_min)
U = W * -(-sets_min // W)
n_mt = -(-B // 32)
worst = np.zeros((n_mt, 5))
for u in range(U):
    gb, ge = G * u // U, G * (u + 1) // U
    for k in range(gb, ge):
        cols = slice(k * 32, min(N, (k + 1) * 32))
        for mt in range(n_mt):
            rows = slice(mt * 32, min(B, (mt + 1) * 32))
            worst[mt, k - gb] = max(worst[mt, k - gb], (np.abs(cg[rows, cols] - z[rows, cols]) / (1 + np.abs(z[rows, cols]))).max())
np.set_printoptions(linewidth=200, precision=2)
print("max |cg - f64| / (1 + |f64|) per (query tile, group in set):")
print(worst)
bad = np.argwhere(np.abs(cg - z) > 1e-4 * (1 + np.abs(z)))
print("bad elements:", len(bad), "of", cg.size)
for d, j in bad[:12]:
    print(f"  q {d} (tile {d // 32} row {d % 32})  col {j} (group {j // 32} col {j % 32}): cg {cg[d, j]:.6g} ws {ws[d, j]:.6g} f64 {z[d, j]:.6g}")
# element-level pattern inside the first bad (tile, group)
if len(bad):
    d0, j0 = bad[0]
    t0, g0 = d0 // 32, j0 // 32
    blk = np.abs(cg - z)[t0 * 32:(t0 + 1) * 32, g0 * 32:(g0 + 1) * 32] > 1e-4 * (1 + np.abs(z[t0 * 32:(t0 + 1) * 32, g0 * 32:(g0 + 1) * 32]))
    print(f"bad pattern in tile {t0}, group {g0} (rows down, columns across):")
    for row in blk:
        print("".join("X" if x else "." for x in row))
