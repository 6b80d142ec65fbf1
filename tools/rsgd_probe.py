#!/usr/bin/env python3
"""Short RSGD-with-momentum runs on WN18RR at a few (lr, regulariser) settings: which one ranks after a minute?
(Used to pick the configuration of tests/test_gpu_trained.py.)  python tools/rsgd_probe.py "lr,reg,epochs;lr,reg,epochs" """
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import r_tucker_amd as rt
from configs.base_config import wn18rr_readme_config
from r_tucker_amd import driver
from r_tucker_amd.data import Data, KG_dataset

spec = sys.argv[1] if len(sys.argv) > 1 else "300,3e-9,12;100,3e-9,12;1000,1e-8,12"
data = Data(os.path.join(ROOT, "data", "WN18RR") + "/", reverse=True)
train_set = KG_dataset(data, data.train_data, label_smoothing=0.1)
val_set = KG_dataset(data, data.valid_data, test_set=True)
flt = rt.DeviceFilter(train_set, "cuda"); vflt = rt.DeviceFilter(val_set, "cuda")
for item in spec.split(";"):
    parts = item.split(",")
    lr, reg, epochs = float(parts[0]), float(parts[1]), int(parts[2])
    core_norm = float(parts[3]) if len(parts) > 3 else None        # rescale the initial core to this Frobenius norm
    torch.manual_seed(322); np.random.seed(322)
    cfg = wn18rr_readme_config()
    model = rt.AsymmetricR_TuckER((len(data.entities), len(data.relations)), cfg.model_cfg.manifold_rank); model.init(); model.cuda()
    if core_norm is not None:
        with torch.no_grad():
            model.core.mul_(core_norm / float(model.core.norm()))
    cfg.train_cfg.learning_rate = lr
    opt = driver.define_optimizer(model, cfg, "asymmetric", "rsgd")
    t0 = time.time()
    for ep in range(1, epochs + 1):
        torch.manual_seed(1000 + ep)
        U0 = model.S.weight.detach().clone()
        loss, gn = driver.train_one_epoch(model, opt, flt, 512, 0.1, regularization_coeff=reg, max_batches=1 if ep == 1 else None)
        if ep == 1:      # how far ONE step turns the subject subspace: 1 - mean cos^2 of the principal angles
            U1 = model.S.weight.detach()
            turn = 1.0 - float(((U0.T @ U1) ** 2).sum()) / U0.shape[1]
            print(f"lr {lr} reg {reg} |G|0 {core_norm}: one step turns the subject subspace by sin^2 = {turn:.3e}", flush=True)
        for g in opt.param_groups:
            g["lr"] = lr * 0.97 ** ep
        if ep % 3 == 0 or ep == epochs:
            m, vl = driver.evaluate(model, val_set, 512, vflt)
            print(f"lr {lr} reg {reg}: epoch {ep} ({time.time() - t0:.0f} s) loss {loss:.5f} |G| {float(model.core.norm()):.1f} val MRR {m['mrr']:.4f} hits@10 {m['hits@10']:.4f}", flush=True)
