#!/usr/bin/env python3
"""Find the first optimizer step whose HIP-graph replay produces a non-finite parameter, then redo that very step
eagerly from a snapshot taken just before it (same parameters, same previous direction, same ids) and compare."""
import json, math, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import r_tucker_amd as rt
from configs.base_config import wn18rr_readme_config
from r_tucker_amd import driver, tucker, ops
import graphstep
graphstep.install()
from r_tucker_amd.data import Data, KG_dataset
from r_tucker_amd.utils.regularization import SimpleDecreasingPolicy

k = 5.0
check_from = int(sys.argv[1]) if len(sys.argv) > 1 else 12
max_epoch = int(sys.argv[2]) if len(sys.argv) > 2 else 16
cfg = wn18rr_readme_config(); tc = cfg.train_cfg
reg_steps = int(round(tc.num_regularizer_decreasing_steps / k)); gamma = tc.scheduler_step ** k
dev = torch.device("cuda")
data = Data(os.path.join(ROOT, "data", "WN18RR") + "/", reverse=True)
torch.manual_seed(322); np.random.seed(322)
model = rt.AsymmetricR_TuckER((len(data.entities), len(data.relations)), cfg.model_cfg.manifold_rank); model.init(); model.to(dev)
opt = driver.define_optimizer(model, cfg, "asymmetric", "rsgd")
regulizer = SimpleDecreasingPolicy(tc.base_regularization_coeff, reg_steps, tc.final_regularization_coeff, tc.coeff_adjusting_policy)
train_set = KG_dataset(data, data.train_data, label_smoothing=tc.label_smoothig)
flt = rt.DeviceFilter(train_set, dev)
val_set = KG_dataset(data, data.valid_data, test_set=True); val_flt = rt.DeviceFilter(val_set, dev)
params = list(opt.param_groups[0]["params"])
B = tc.train_batch_size

def state_tensors():
    ts = list(params)
    if opt._prev is not None:
        ts += opt._tensors_of(opt._prev)
    return ts

for epoch in range(1, max_epoch + 1):
    coeff = regulizer.step()
    torch.manual_seed(322 * 100003 + epoch)
    model.train()
    n = flt.features.shape[0]; nb = n // B
    perm = torch.randperm(n, device=dev)
    step = driver._captured_step(model, opt, flt, B, tc.label_smoothig)
    with ops.index_check("off"):
        step.begin_epoch(coeff)
        for b in range(nb):
            ids = perm[b * B:(b + 1) * B]
            if epoch >= check_from and step.graph is not None:
                snap = [t.detach().clone() for t in state_tensors()]
                step.run(ids)
                torch.cuda.synchronize()
                bad = [i for i, t in enumerate(state_tensors()) if not bool(torch.isfinite(t).all())]
                if bad or not math.isfinite(step.totals()[0]):
                    print(f"epoch {epoch} step {b}: graph replay left non-finite state tensors {bad}, loss_sum {step.totals()}", flush=True)
                    after = [t.detach().clone() for t in state_tensors()]
                    for t, s in zip(state_tensors(), snap):
                        t.detach().copy_(s)
                    step.loss_sum.zero_(); step.gnorm_sum.zero_()
                    # same step again from the graph
                    step.graph.replay(); torch.cuda.synchronize()
                    bad2 = [i for i, t in enumerate(state_tensors()) if not bool(torch.isfinite(t).all())]
                    print(f"  second replay from the snapshot: non-finite {bad2}, loss {step.totals()}", flush=True)
                    for t, s in zip(state_tensors(), snap):
                        t.detach().copy_(s)
                    step.loss_sum.zero_(); step.gnorm_sum.zero_()
                    try:
                        step._body(); torch.cuda.synchronize()
                        bad3 = [i for i, t in enumerate(state_tensors()) if not bool(torch.isfinite(t).all())]
                        print(f"  eager step from the snapshot: non-finite {bad3}, loss {step.totals()}", flush=True)
                        diffs = [(a - t).abs().max().item() if torch.isfinite(a).all() else float('nan') for a, t in zip(after, state_tensors())]
                        print("  |graph - eager| per tensor:", diffs, flush=True)
                    except Exception as e:
                        print("  eager step raised:", repr(e)[:500], flush=True)
                    sys.exit(0)
            else:
                step.run(ids)
    tl, gs = step.totals()
    print(f"epoch {epoch}: loss {tl / nb:.5f} gnorm {gs / nb:.5f} core {float(model.core.norm()):.3f}", flush=True)
    for g in opt.param_groups:
        g["lr"] = tc.learning_rate * gamma ** epoch
    driver.evaluate(model, val_set, tc.eval_batch_size, val_flt)
print("no non-finite state found")
