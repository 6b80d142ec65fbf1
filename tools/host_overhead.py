import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import gen, r_tucker_amd as rt
n_ent, n_rel, B, rank = 40943, 22, 512, (10, 200, 200)
core, R, S, O = [torch.from_numpy(x).cuda() for x in gen.make_params(n_ent, n_rel, rank, 322)]
pool = [tuple(torch.from_numpy(x).cuda() for x in gen.make_queries(n_ent, n_rel, B, 1000 + i)) for i in range(16)]
for i in range(50): rt.score_1vN(core, R, S, O, *pool[i % 16])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(2000): p = rt.score_1vN(core, R, S, O, *pool[i % 16])
torch.cuda.synchronize()
print(f"score_1vN via the Python wrapper: {(time.perf_counter() - t0) / 2000 * 1e6:.1f} us per call")
m = rt.AsymmetricR_TuckER((n_ent, n_rel), rank).cuda()
T = rt.Tucker(m.core.data, [m.R.weight, m.S.weight, m.O.weight])
with torch.no_grad():
    for i in range(50): m(*pool[i % 16])(T)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(2000): p = m(*pool[i % 16])(T)
    torch.cuda.synchronize()
print(f"model(h, r)(T) closure: {(time.perf_counter() - t0) / 2000 * 1e6:.1f} us per call")
