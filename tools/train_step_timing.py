#!/usr/bin/env python3
"""Forward + backward of the training loss term at the WN18RR shape (B 512, rank (10,200,200), fp32):
(a) score_1vN (HIP, autograd) + torch BCELoss against a dense target matrix resident on the device,
(b) bce_loss_1vN (targets from the CSR inside the kernels).  The reference additionally builds the
dense targets on the host and copies 84 MB per batch to the device; that is not timed here."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import gen  # noqa: E402
import r_tucker_amd as rt  # noqa: E402
from r_tucker_amd.data import Data, KG_dataset  # noqa: E402

data = Data(os.path.join(ROOT, "data", "WN18RR") + "/", reverse=True)
train = KG_dataset(data, data.train_data, label_smoothing=0.1)
n_ent, n_rel, rank, B = len(data.entities), len(data.relations), (10, 200, 200), 512
core, R, S, O = [torch.from_numpy(x).cuda().requires_grad_(True) for x in gen.make_params(n_ent, n_rel, rank, 322)]
flt = rt.DeviceFilter(train, "cuda")
ids = torch.arange(2000, 2000 + B).cuda()
f = flt.features[ids]
h, r = f[:, 0].contiguous(), f[:, 1].contiguous()
targets = train.dense_targets(ids.cpu().numpy()).cuda()
crit = torch.nn.BCELoss()


def a():
    loss = crit(rt.score_1vN(core, R, S, O, h, r), targets)
    loss.backward()
    return loss


def b():
    loss = rt.bce_loss_1vN(core, R, S, O, h, r, flt, ids, label_smoothing=0.1)
    loss.backward()
    return loss


def c():
    from r_tucker_amd import ops
    ops.FUSED_BCE = False
    try:
        return b()
    finally:
        ops.FUSED_BCE = True


for name, fn in (("score_1vN + nn.BCELoss(dense targets)", a), ("bce_loss_1vN (CSR targets, loss fused into the score epilogue)", b),
                 ("bce_loss_1vN (CSR targets, three passes over the scores: round 2)", c)):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        loss = fn()
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / 30 * 1e3:.3f} ms per forward+backward, loss {loss.item():.6f}")
t0 = time.perf_counter()
for _ in range(5):
    tt = train.dense_targets(ids.cpu().numpy()).cuda()
torch.cuda.synchronize()
print(f"host build + H2D of the dense targets (vectorised builder, not the reference's per-item loop): {(time.perf_counter() - t0) / 5 * 1e3:.1f} ms per batch")
