#!/usr/bin/env python3
"""Time the two B x N sized backward products dO = dZ^T v and dv = dZ O at the WN18RR shapes on the
split-fp16 GEMM (rtk_gemm_sf16_splitk) and on the exact fp32 MFMA GEMM (rtk_gemm_f32[_splitk])."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import r_tucker_amd as rt  # noqa: E402
from r_tucker_amd.ops import _splits_for, alloc_scores  # noqa: E402

lib = rt._lib.load()
dev = torch.device("cuda:0")
sp = torch.cuda.current_stream().cuda_stream


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for B, N, c in ((512, 40943, 200), (512, 40943, 400)):
    dZ = alloc_scores(B, N, dev)
    dZ.normal_()
    dZ *= 1e-8
    v = torch.randn(B, c, device=dev)
    O = torch.randn(N, c, device=dev)
    gO = torch.empty(N, c, device=dev)
    dv = torch.empty(B, c, device=dev)
    ldz = dZ.stride(0)
    bz = torch.linalg.vector_norm(dZ, ord=float("inf")).reshape(1)
    bv = torch.linalg.vector_norm(v, ord=float("inf")).reshape(1)
    bo = torch.linalg.vector_norm(O, ord=float("inf")).reshape(1)
    splits = _splits_for(B, c, N)
    ws = torch.empty(max(256, lib.rtk_gemm_f32_splitk_workspace_bytes(B, c, splits)), dtype=torch.uint8, device=dev)
    flops = 2.0 * B * N * c
    t = timeit(lambda: lib.rtk_gemm_sf16_splitk(dZ.data_ptr(), 0, ldz, bz.data_ptr(), v.data_ptr(), 0, c, bv.data_ptr(),
                                                gO.data_ptr(), c, N, c, B, 1, None, 0, sp))
    print(f"c={c} dO split-fp16: {t:7.1f} us  ({3 * flops / t / 1e6:.0f} TF f16)")
    t = timeit(lambda: lib.rtk_gemm_f32(dZ.data_ptr(), 0, ldz, v.data_ptr(), 0, c, gO.data_ptr(), c, N, c, B, 0, sp))
    print(f"c={c} dO fp32 MFMA : {t:7.1f} us")
    t = timeit(lambda: lib.rtk_gemm_sf16_splitk(dZ.data_ptr(), 1, ldz, bz.data_ptr(), O.data_ptr(), 0, c, bo.data_ptr(),
                                                dv.data_ptr(), c, B, c, N, splits, ws.data_ptr(), ws.numel(), sp))
    print(f"c={c} dv split-fp16: {t:7.1f} us  (splits {splits}, incl. slab reduction)")
    t = timeit(lambda: lib.rtk_gemm_f32_splitk(dZ.data_ptr(), 1, ldz, O.data_ptr(), 0, c, dv.data_ptr(), c, B, c, N, splits,
                                               ws.data_ptr(), ws.numel(), sp))
    print(f"c={c} dv fp32 MFMA : {t:7.1f} us")
    t = timeit(lambda: torch.linalg.vector_norm(O, ord=float("inf")))
    print(f"c={c} torch max|O| : {t:7.1f} us")
