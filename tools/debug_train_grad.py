#!/usr/bin/env python3
"""Where do the HIP gradients at the WN18RR training shape differ from CPU autograd?  (debug aid)"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import gen
import make_golden_train_grad as mg
import r_tucker_amd as rt
from oracle import score_oracle as orc

core, R, S, O = gen.make_params(mg.N_ENT, mg.N_REL, mg.RANK, mg.SEED)
h, r, lists = mg.make_batch()
tc, tR, tS, tO = [torch.from_numpy(x) for x in (core, R, S, O)]
th, tr = torch.from_numpy(h), torch.from_numpy(r)
targets = torch.zeros((mg.B, mg.N_ENT))
for d, l in enumerate(lists):
    targets[d, l] = 1.0
targets = (1.0 - mg.EPS) * targets + (1.0 / targets.shape[1]) * mg.EPS
# CPU: logits, p, analytic dZ and autograd dZ
z = orc.logits_ref(tc, tR, tS, tO, th, tr).requires_grad_(True)
p = torch.sigmoid(z)
torch.nn.BCELoss(reduction="mean")(p, targets).backward()
dz_auto = z.grad
dz_analytic = (p.detach() - targets) / p.numel()
print("CPU autograd vs analytic dZ: max abs", (dz_auto - dz_analytic).abs().max().item(), " saturated p==1:", int((p == 1).sum()),
      " p==0:", int((p == 0).sum()), " max|z|", z.abs().max().item())
bad = (dz_auto - dz_analytic).abs() > 1e-9
print("entries where autograd != analytic by > 1e-9:", int(bad.sum()), bad.nonzero()[:10].tolist(), z.detach()[bad][:10].tolist())
# device dZ
class Flt: pass
flt = Flt()
flt.slot_of_item = torch.arange(mg.B, device="cuda")
flt.pair_ptr = torch.from_numpy(np.concatenate([[0], np.cumsum([len(x) for x in lists])]).astype(np.int64)).cuda()
flt.pair_obj = torch.tensor([x for l in lists for x in l], dtype=torch.int64, device="cuda")
P = rt.score_1vN(tc.cuda(), tR.cuda(), tS.cuda(), tO.cuda(), th.cuda(), tr.cuda())
print("device p vs CPU p: max abs", (P.cpu() - p.detach()).abs().max().item())
dz_dev = (P.cpu() - targets) / p.numel()
d = (dz_dev - dz_auto).abs()
print("device-analytic dZ vs CPU autograd dZ: max", d.max().item(), " #>1e-9:", int((d > 1e-9).sum()), (d > 1e-9).nonzero()[:10].tolist())
d2 = (dz_dev - dz_analytic).abs()
print("device-analytic dZ vs CPU analytic dZ: max", d2.max().item())
