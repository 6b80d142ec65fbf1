#!/usr/bin/env python3
"""Time rtk_query_vectors_bwd_f32 (stage-1 backward) at the WN18RR shape for different id patterns
(run under rocprofv3 --kernel-trace --stats to see scatter_rows_kernel on its own)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import r_tucker_amd as rt  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "distinct"
lib = rt._lib.load()
dev = torch.device("cuda:0")
B, a, b, c, n_rel, n_sub = 512, 10, 200, 200, 22, 40943
g = torch.Generator(device=dev).manual_seed(1)
core = torch.randn(a, b, c, device=dev, generator=g)
R = torch.randn(n_rel, a, device=dev, generator=g)
S = torch.randn(n_sub, b, device=dev, generator=g)
dv = torch.randn(B, c, device=dev, generator=g)
if mode == "distinct":
    h = torch.randperm(n_sub, device=dev, generator=g)[:B]
    r = torch.arange(B, device=dev) % n_rel
elif mode == "one_relation":
    h = torch.randperm(n_sub, device=dev, generator=g)[:B]
    r = torch.zeros(B, dtype=torch.int64, device=dev)
elif mode == "many_relations":
    h = torch.randperm(n_sub, device=dev, generator=g)[:B]
    n_rel = 600
    R = torch.randn(n_rel, a, device=dev, generator=g)
    r = torch.randperm(n_rel, device=dev, generator=g)[:B]
else:
    raise SystemExit(mode)
h, r = h.contiguous(), r.contiguous()
gc, gR, gS = torch.empty_like(core), torch.empty_like(R), torch.empty_like(S)
ws = torch.empty(lib.rtk_query_bwd_workspace_bytes(B, a, b, c), dtype=torch.uint8, device=dev)
sp = torch.cuda.current_stream().cuda_stream
for _ in range(30):
    rc = lib.rtk_query_vectors_bwd_f32(core.data_ptr(), a, b, c, R.data_ptr(), n_rel, S.data_ptr(), n_sub, r.data_ptr(), h.data_ptr(), B,
                                       dv.data_ptr(), gc.data_ptr(), gR.data_ptr(), gS.data_ptr(), ws.data_ptr(), ws.numel(), sp)
    assert rc == 0
torch.cuda.synchronize()
print(mode, "done")
