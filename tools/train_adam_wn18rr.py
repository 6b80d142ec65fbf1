#!/usr/bin/env python3
"""End-to-end exercise of the HIP path in a training loop on WN18RR: R_TuckER parameters trained with
torch.optim.Adam on the 1-vs-all BCE loss (bce_loss_1vN: scores, loss and d loss / d logits in HIP
kernels, targets from the CSR) and evaluated with the on-device filtered ranking.
NOT the reference's optimizer (its Riemannian SGD/Adam lives in tucker_riemopt): this
only shows that forward, backward, loss and evaluation work together and produce a model that ranks."""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import r_tucker_amd as rt  # noqa: E402
from r_tucker_amd.data import Data, KG_dataset  # noqa: E402


def train(epochs=30, rank=(10, 200, 200), batch=512, lr=3e-3, seed=322, smoothing=0.1, log=print, lr_decay=1.0):
    torch.manual_seed(seed)
    np.random.seed(seed)
    data = Data(os.path.join(ROOT, "data", "WN18RR") + "/", reverse=True)
    train_set = KG_dataset(data, data.train_data, label_smoothing=smoothing)
    test_set = KG_dataset(data, data.test_data, test_set=True)
    n_ent, n_rel = len(data.entities), len(data.relations)
    model = rt.AsymmetricR_TuckER((n_ent, n_rel), rank).cuda()
    with torch.no_grad():                                 # N(0, small) start: the orthonormal init() scores are all 0.5
        for p in model.parameters():
            p.copy_(torch.randn_like(p) * (0.3 if p.dim() == 3 else 0.1))
    opt = torch.optim.Adam(model.parameters(), lr=lr)
    sched = torch.optim.lr_scheduler.ExponentialLR(opt, gamma=lr_decay)     # per epoch; 1.0 = constant
    flt = rt.DeviceFilter(train_set, "cuda")
    n = len(train_set)
    t0 = time.perf_counter()
    for ep in range(epochs):
        perm = torch.randperm(n, device="cuda")
        tot = 0.0
        for lo in range(0, n, batch):
            ids = perm[lo:lo + batch]
            f = flt.features[ids]
            opt.zero_grad(set_to_none=True)
            loss = rt.bce_loss_1vN(model.core, model.R.weight, model.S.weight, model.O.weight,
                                   f[:, 0].contiguous(), f[:, 1].contiguous(), flt, ids, label_smoothing=smoothing)
            loss.backward()
            opt.step()
            tot += float(loss.detach()) if (lo // batch) % 50 == 0 else 0.0
        sched.step()
        if ep % 5 == 4 or ep == epochs - 1:
            m, l = rt.evaluate(model, test_set, batch_size=batch)
            torch.cuda.synchronize()
            log(f"epoch {ep + 1}: {time.perf_counter() - t0:.1f} s  test MRR {m['mrr']:.4f} hits@1 {m['hits@1']:.4f} hits@10 {m['hits@10']:.4f} loss {l:.5f}")
    return model, data, test_set


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=30)
    ap.add_argument("--lr", type=float, default=3e-3)
    ap.add_argument("--lr-decay", type=float, default=1.0)
    a = ap.parse_args()
    train(a.epochs, lr=a.lr, lr_decay=a.lr_decay)
