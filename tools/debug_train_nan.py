#!/usr/bin/env python3
"""Run the driver's training loop with finiteness checks after every fit/step and stop at the first
non-finite quantity, reporting where it appeared (debugging aid for the Riemannian layer)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import r_tucker_amd as rt  # noqa: E402
from r_tucker_amd import driver, ops  # noqa: E402
from r_tucker_amd.data import Data, KG_dataset  # noqa: E402
from r_tucker_amd.model.asymmetric.R_TuckER import R_TuckER  # noqa: E402
from r_tucker_amd.model.asymmetric.optim import RSGDwithMomentum  # noqa: E402

lr, decay = float(os.environ.get("LR", "60")), float(os.environ.get("DECAY", "0.97"))
max_epochs = int(os.environ.get("EPOCHS", "12"))
torch.manual_seed(322)
data = Data(os.path.join(ROOT, "data", "WN18RR") + "/", reverse=True)
train = KG_dataset(data, data.train_data, label_smoothing=0.1)
dev = torch.device("cuda:0")
model = R_TuckER((len(data.entities), len(data.relations)), (10, 200, 200), device=dev)
model.init(None)
model.to(dev)
flt = rt.DeviceFilter(train, dev)
opt = RSGDwithMomentum([model.core, model.S.weight, model.R.weight, model.O.weight], (10, 200, 200), lr, 0.8)
B = 512
n = flt.features.shape[0]


def finite(name, *ts):
    for i, t in enumerate(ts):
        if t is not None and not torch.isfinite(t).all():
            bad = (~torch.isfinite(t)).sum().item()
            print(f"NON-FINITE: {name}[{i}] shape {tuple(t.shape)}: {bad} entries, absmax of finite {t[torch.isfinite(t)].abs().max().item() if bad < t.numel() else 'n/a'}")
            return False
    return True


step = 0
for epoch in range(1, max_epochs + 1):
    model.train()
    perm = torch.randperm(n, device=dev)
    tot = 0.0
    for b in range(n // B):
        ids = perm[b * B:(b + 1) * B]
        f = flt.features[ids]
        loss_fn = driver.batch_loss_fn(model, f[:, 0].contiguous(), f[:, 1].contiguous(), flt, ids, 0.1, 1e-12)
        x_k = driver.extract_tensor(model)
        gn = opt.fit(loss_fn, x_k)
        d = opt.direction
        ok = finite("loss", opt.loss) and finite("grad norm", gn) and finite("direction core", d.delta_core if hasattr(d, "delta_core") else None)
        if ok and hasattr(d, "delta_factors"):
            ok = finite("direction factors", *d.delta_factors)
        if not ok:
            print(f"first failure in fit: epoch {epoch} batch {b} step {step}; attributes of direction: {[k for k in vars(d)]}")
            sys.exit(0)
        opt.step()
        if not finite("params after step", model.core.data, model.R.weight.data, model.S.weight.data, model.O.weight.data):
            print(f"first failure in step: epoch {epoch} batch {b} step {step}; loss {opt.loss.item()} grad norm {gn.item()}")
            sys.exit(0)
        tot += opt.loss.item()
        step += 1
    opt.param_groups[0]["lr"] *= decay
    print(f"epoch {epoch}: mean loss {tot / (n // B):.6f}, core norm {model.core.data.norm().item():.4e}, lr {opt.param_groups[0]['lr']:.2f}", flush=True)
print("no non-finite value")
