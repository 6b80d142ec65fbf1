#!/usr/bin/env python3
"""Wall time of one Riemannian optimizer step (fit + step) at the WN18RR recipe shape, eager vs replayed from the
HIP graph (r_tucker_amd.graphstep).  Usage: python tools/opt_step_timing.py [n_steps]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import r_tucker_amd as rt                                   # noqa: E402
from r_tucker_amd import driver
import graphstep                                         # tools/graphstep.py (experiment)
graphstep.install()                  # noqa: E402
from r_tucker_amd.data import Data, KG_dataset              # noqa: E402
from r_tucker_amd.model.asymmetric.optim import RSGDwithMomentum   # noqa: E402

n_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
data = Data(os.path.join(ROOT, "data", "WN18RR") + "/", reverse=True)
train_set = KG_dataset(data, data.train_data, label_smoothing=0.1)
flt = rt.DeviceFilter(train_set, "cuda")
rank = (10, 200, 200)


def run(enabled):
    graphstep.ENABLED = enabled
    torch.manual_seed(5)
    model = rt.AsymmetricR_TuckER((len(data.entities), len(data.relations)), rank)
    model.init()
    model.cuda()
    params = torch.nn.ParameterList([model.core, model.S.weight, model.R.weight, model.O.weight])
    opt = RSGDwithMomentum(params, rank, 2000.0, 0.8)
    step = driver._captured_step(model, opt, flt, 512, 0.1)
    step.begin_epoch(1e-4)
    gen = torch.Generator(device="cuda").manual_seed(11)
    ids = [torch.randint(0, flt.features.shape[0], (512,), device="cuda", generator=gen) for _ in range(n_steps + 4)]
    t = []
    for i in range(4):                      # 2 eager + capture + 1 replay
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step.run(ids[i])
        torch.cuda.synchronize()
        t.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n_steps):
        step.run(ids[4 + i])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n_steps
    print(f"graph={enabled}: first four calls {[round(x * 1e3, 1) for x in t]} ms; steady {dt * 1e3:.2f} ms/step; "
          f"loss sum {step.totals()[0]:.4f}", flush=True)


run(False)
run(True)
