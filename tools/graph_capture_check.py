import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch, gen, r_tucker_amd as rt
n_ent, n_rel, B, rank = 40943, 22, 512, (10, 200, 200)
core, R, S, O = [torch.from_numpy(x).cuda() for x in gen.make_params(n_ent, n_rel, rank, 322)]
h, r = [torch.from_numpy(x).cuda() for x in gen.make_queries(n_ent, n_rel, B, 1)]
h2, r2 = [torch.from_numpy(x).cuda() for x in gen.make_queries(n_ent, n_rel, B, 2)]
out = rt.ops.alloc_scores(B, n_ent, "cuda")
hs, rs = h.clone(), r.clone()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3): rt.score_1vN_into(core, R, S, O, hs, rs, out)     # warm-up on the capture stream
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    rt.score_1vN_into(core, R, S, O, hs, rs, out)
ref1 = rt.score_1vN(core, R, S, O, h, r).clone()
ref2 = rt.score_1vN(core, R, S, O, h2, r2).clone()
g.replay(); torch.cuda.synchronize()
ok1 = torch.equal(out, ref1)
hs.copy_(h2); rs.copy_(r2)
g.replay(); torch.cuda.synchronize()
ok2 = torch.equal(out, ref2)
import time
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(2000): g.replay()
torch.cuda.synchronize()
print("graph replay matches eager:", ok1, ok2, f" {(time.perf_counter()-t0)/2000*1e6:.1f} us per replay")
