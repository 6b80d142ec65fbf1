#!/usr/bin/env python3
"""Column-group against wave-specialised score kernel over entity counts (c = 200, B = 512, fast logistic): us per
launch, 64 launches back to back inside one event pair, best of three.  Decides rtk_score_cg.hip's dispatch rule.
usage: tools/ab_cg_shapes.py [N ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import r_tucker_amd as rt  # noqa: E402

L = rt._lib
lib = L.load()
Ns = [int(x) for x in sys.argv[1:]] or [8000, 14951, 20000, 26000, 32000, 36000, 40943, 46000, 61000, 81886, 122829]
c, B = 200, 512
g = torch.Generator().manual_seed(3)
v = torch.randn((B, c), generator=g).cuda()
qp = rt.pack_query_vectors(v, torch.float32)
sp = torch.cuda.current_stream().cuda_stream
base = L.RTK_SCORE_SIGMOID | L.RTK_SCORE_SIGMOID_FAST
for N in Ns:
    O = torch.randn((N, c), generator=g).cuda()
    pitch = -(-N // 32) * 32
    out = torch.empty((B, pitch), dtype=torch.float32, device="cuda")
    res = {}
    for name, hint in (("cg", L.RTK_SCORE_KERNEL_CG), ("ws", L.RTK_SCORE_KERNEL_WS), ("default", 0)):
        def launch():
            L.check(lib.rtk_score_packed_f32(qp.data_ptr(), B, c, O.data_ptr(), N, out.data_ptr(), pitch, base | hint, sp), "score")
        for _ in range(8):
            launch()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(64):
                launch()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 64 * 1e3)
        res[name] = best
    G = -(-N // 32)
    print(f"N {N:7d}  groups {G:5d} = {G / 256:5.2f} per CU   cg {res['cg']:7.2f} us   ws {res['ws']:7.2f} us   default {res['default']:7.2f}   cg/ws {res['cg'] / res['ws']:.3f}", flush=True)
