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


from r_tucker_amd import tucker as _tk  # noqa: E402


def core_spectrum(core):
    out = []
    for m in range(3):
        u = core.movedim(m, 0).reshape(core.shape[m], -1).double()
        sv = torch.linalg.svdvals(u)
        out.append((sv[0].item(), sv[-1].item()))
    return out


gs = torch.Generator(device=dev).manual_seed(11)
sh = torch.randint(0, len(data.entities), (4096,), device=dev, generator=gs)
so = torch.randint(0, len(data.entities), (4096,), device=dev, generator=gs)
sr = torch.randint(0, len(data.relations), (4096,), device=dev, generator=gs)


def entries(core, R, S, O):
    """T[r, h, o] at the sampled triples (float64)."""
    t = torch.einsum("abc,na->nbc", core.double(), R[sr].double())
    t = torch.einsum("nbc,nb->nc", t, S[sh].double())
    return (t * O[so].double()).sum(dim=1)


hist = []
prev_loss = None
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
        lr_now = opt.param_groups[0]["lr"]
        built = ((-lr_now) * d + type(d)(d.point)).construct()
        want = entries(built.core, *built.factors)
        old = entries(d.point.core, *d.point.factors)
        opt.step()
        got = entries(model.core.data, model.R.weight.data, model.S.weight.data, model.O.weight.data)
        rel = ((got - want).norm() / want.norm()).item()
        upd = ((want - old).norm() / want.norm()).item()
        if rel > 0.5 * max(upd, 1e-6) and rel > 1e-4:
            print(f"RETRACTION MISMATCH at epoch {epoch} batch {b} step {step}: |round(x - lr d) - (x - lr d)| / |.| = {rel:.3e}, "
                  f"update size {upd:.3e}, health {_tk.read_health(clear=False)}")
            print("   core spectrum after:", core_spectrum(model.core.data))
            sys.exit(0)
        if not finite("params after step", model.core.data, model.R.weight.data, model.S.weight.data, model.O.weight.data):
            print(f"first failure in step: epoch {epoch} batch {b} step {step}; loss {opt.loss.item()} grad norm {gn.item()}")
            sys.exit(0)
        cur = opt.loss.item()
        hist.append((step, cur, gn.item(), _tk.read_health(), max(x.abs().max().item() for x in d.delta_factors),
                     d.delta_core.abs().max().item()))
        if prev_loss is not None and cur > prev_loss + 0.05:
            print(f"LOSS JUMP at epoch {epoch} batch {b} step {step}: {prev_loss:.4f} -> {cur:.4f}")
            for h in hist[-6:]:
                print("   step %d loss %.5f gnorm %.3e fallbacks %s max|dU| %.3e max|dG| %.3e" % h)
            print("   core spectrum (max, min singular value per mode):", core_spectrum(model.core.data))
            print("   factor orthonormality:", [(w.data.T @ w.data - torch.eye(w.shape[1], device=dev)).abs().max().item() for w in (model.R.weight, model.S.weight, model.O.weight)])
            sys.exit(0)
        prev_loss = 0.9 * prev_loss + 0.1 * cur if prev_loss is not None else cur
        tot += cur
        step += 1
    opt.param_groups[0]["lr"] *= decay
    print(f"epoch {epoch}: mean loss {tot / (n // B):.6f}, core norm {model.core.data.norm().item():.4e}, lr {opt.param_groups[0]['lr']:.2f}", flush=True)
print("no non-finite value")
