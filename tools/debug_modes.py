import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import gen
import r_tucker_amd as rt
n_ent, n_rel, B, rank = 9001, 7, 200, (5, 96, 96)
core, R, S, O = gen.make_params(n_ent, n_rel, rank, 21)
h, r = gen.make_queries(n_ent, n_rel, B, 21)
w = torch.from_numpy(np.random.default_rng(2).standard_normal((B, n_ent)).astype(np.float32)).cuda()
out = {}
for mode in ("split_fp16", "f32", "split_fp16"):
    rt.ops.BACKWARD_GEMM = mode
    leaves = [torch.from_numpy(x).cuda().requires_grad_(True) for x in (core, R, S, O)]
    (rt.score_1vN(*leaves, torch.from_numpy(h).cuda(), torch.from_numpy(r).cuda()) * w).sum().backward()
    g = [x.grad.clone() for x in leaves]
    if mode in out:
        print("repeat equal:", [torch.equal(a, b) for a, b in zip(out[mode], g)])
    out[mode] = g
for name, g0, g1 in zip("core R S O".split(), out["split_fp16"], out["f32"]):
    d = (g0 - g1).abs()
    print(name, "max diff", d.max().item(), "max ref", g1.abs().max().item(), "ratio", (d.max() / g1.abs().max()).item(), "nan", torch.isnan(g0).any().item())
