#!/usr/bin/env python3
"""A one-minute Riemannian training run on WN18RR -- the reference's optimizer protocol (RSGDwithMomentum.fit / .step
through driver.train_one_epoch, src/model/asymmetric/optim.py:60-114, train.py:69-91) on the HIP loss, with a step
length that learns immediately (lr 100 decayed 0.97 / epoch, regulariser at the README's final 3e-9; the README's own
schedule spends its first ~300 epochs with the regulariser in charge): validation MRR ~0.03 after
9 epochs, 100x chance.  Used by tests/test_gpu_trained.py to produce TRAINED parameters for the MRR-parity check."""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import r_tucker_amd as rt  # noqa: E402
from configs.base_config import wn18rr_readme_config  # noqa: E402
from r_tucker_amd import driver  # noqa: E402
from r_tucker_amd.data import Data, KG_dataset  # noqa: E402


def train(epochs=9, lr=100.0, lr_decay=0.97, reg=3e-9, seed=322, log=print):
    torch.manual_seed(seed)
    np.random.seed(seed)
    data = Data(os.path.join(ROOT, "data", "WN18RR") + "/", reverse=True)
    train_set = KG_dataset(data, data.train_data, label_smoothing=0.1)
    test_set = KG_dataset(data, data.test_data, test_set=True)
    cfg = wn18rr_readme_config()
    cfg.train_cfg.learning_rate = lr
    model = rt.AsymmetricR_TuckER((len(data.entities), len(data.relations)), cfg.model_cfg.manifold_rank)
    model.init()
    model.cuda()
    opt = driver.define_optimizer(model, cfg, "asymmetric", "rsgd")
    flt = rt.DeviceFilter(train_set, "cuda")
    t0 = time.perf_counter()
    for ep in range(1, epochs + 1):
        loss, gn = driver.train_one_epoch(model, opt, flt, cfg.train_cfg.train_batch_size, cfg.train_cfg.label_smoothig,
                                          regularization_coeff=reg)
        for g in opt.param_groups:
            g["lr"] = lr * lr_decay ** ep
        if ep % 3 == 0 or ep == epochs:
            m, l = rt.evaluate(model, test_set, batch_size=512)
            log(f"epoch {ep}: {time.perf_counter() - t0:.1f} s  train loss {loss:.5f} grad norm {gn:.2e}  test MRR {m['mrr']:.4f} "
                f"hits@10 {m['hits@10']:.4f}")
    return model, data, test_set


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=9)
    ap.add_argument("--lr", type=float, default=100.0)
    a = ap.parse_args()
    train(a.epochs, lr=a.lr)
