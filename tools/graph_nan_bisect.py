#!/usr/bin/env python3
"""Replay the captured optimizer step (WN18RR recipe shape) one launch at a time, with a full synchronisation and a
finiteness check of the optimizer's state after every replay; report the first replay that leaves a non-finite value,
whether a second replay from the same snapshot does so again, and what the same step gives eagerly.
Usage: python tools/graph_nan_bisect.py [n_replays]   (variants are selected through environment variables)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import r_tucker_amd as rt                                   # noqa: E402
from r_tucker_amd import driver
import graphstep                                         # tools/graphstep.py (experiment)
graphstep.install()                  # noqa: E402
from r_tucker_amd.data import Data, KG_dataset              # noqa: E402
from r_tucker_amd.model.asymmetric.optim import RSGDwithMomentum   # noqa: E402

n_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
label = os.environ.get("VARIANT", "default")
data = Data(os.path.join(ROOT, "data", "WN18RR") + "/", reverse=True)
flt = rt.DeviceFilter(KG_dataset(data, data.train_data, label_smoothing=0.1), "cuda")
rank = (10, 200, 200)
graphstep.ENABLED = True
torch.manual_seed(5)
model = rt.AsymmetricR_TuckER((len(data.entities), len(data.relations)), rank)
model.init()
model.cuda()
params = torch.nn.ParameterList([model.core, model.S.weight, model.R.weight, model.O.weight])
opt = RSGDwithMomentum(params, rank, 2000.0, 0.8)
step = driver._captured_step(model, opt, flt, 512, 0.1)
step.begin_epoch(1e-4)
gen = torch.Generator(device="cuda").manual_seed(11)
ids = [torch.randint(0, flt.features.shape[0], (512,), device="cuda", generator=gen) for _ in range(n_steps + 3)]


def state():
    ts = list(params)
    if opt._prev is not None:
        ts += opt._tensors_of(opt._prev)
    return ts


def bad():
    return [i for i, t in enumerate(state()) if not bool(torch.isfinite(t).all())]


for i in range(3):
    step.run(ids[i])
torch.cuda.synchronize()
assert step.graph is not None and not bad(), (step.graph, bad())
for i in range(n_steps):
    snap = [t.detach().clone() for t in state()]
    step.run(ids[3 + i])
    torch.cuda.synchronize()
    b = bad()
    if b:
        after = [t.detach().clone() for t in state()]
        nan_counts = [int((~torch.isfinite(t)).sum()) for t in after]
        print(f"[{label}] replay {i}: non-finite state tensors {b} (counts {nan_counts})", flush=True)
        for t, s in zip(state(), snap):
            t.detach().copy_(s)
        step.graph.replay()
        torch.cuda.synchronize()
        print(f"[{label}]   same replay again from the snapshot: non-finite {bad()}", flush=True)
        for t, s in zip(state(), snap):
            t.detach().copy_(s)
        step._body()
        torch.cuda.synchronize()
        print(f"[{label}]   eager from the snapshot: non-finite {bad()}", flush=True)
        sys.exit(0)
print(f"[{label}] {n_steps} replays, all finite; loss sum {step.totals()[0]:.4f}", flush=True)
