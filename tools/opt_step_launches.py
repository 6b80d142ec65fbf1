#!/usr/bin/env python3
"""Where one eager Riemannian optimizer step (WN18RR recipe shape) launches its kernels: torch.profiler over five steps,
kernel launches per aten op and per source line of r-tucker_amd/ (innermost frame of ours on the op's stack).
Usage: python tools/opt_step_launches.py"""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import r_tucker_amd as rt                                   # noqa: E402
from r_tucker_amd import driver                                # noqa: E402
from r_tucker_amd.data import Data, KG_dataset              # noqa: E402
from r_tucker_amd.model.asymmetric.optim import RSGDwithMomentum   # noqa: E402

data = Data(os.path.join(ROOT, "data", "WN18RR") + "/", reverse=True)
train_set = KG_dataset(data, data.train_data, label_smoothing=0.1)
flt = rt.DeviceFilter(train_set, "cuda")
rank = (10, 200, 200)
torch.manual_seed(5)
model = rt.AsymmetricR_TuckER((len(data.entities), len(data.relations)), rank)
model.init()
model.cuda()
params = torch.nn.ParameterList([model.core, model.S.weight, model.R.weight, model.O.weight])
opt = RSGDwithMomentum(params, rank, 2000.0, 0.8)
step = driver._captured_step(model, opt, flt, 512, 0.1)
step.begin_epoch(1e-4)
gen = torch.Generator(device="cuda").manual_seed(11)
ids = [torch.randint(0, flt.features.shape[0], (512,), device="cuda", generator=gen) for _ in range(12)]
for i in range(4):
    step.run(ids[i])
torch.cuda.synchronize()
N = 5
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for i in range(N):
        step.run(ids[4 + i])
    torch.cuda.synchronize()
ev = prof.events()
kern = [e for e in ev if e.device_type == torch.autograd.DeviceType.CUDA]
print(f"{len(kern) / N:.0f} device activities per step")
by_op = collections.Counter()
by_line = collections.Counter()
for e in ev:
    if e.device_type != torch.autograd.DeviceType.CPU or not e.kernels:
        continue
    n = len(e.kernels)
    by_op[e.name] += n
    where = "?"
    for fr in (e.stack or []):
        if "r-tucker_amd" in fr or "r_tucker_amd" in fr:
            where = fr.split("r-tucker_amd/")[-1] if "r-tucker_amd/" in fr else fr
            break
    by_line[where] += n
print("-- by op (kernels per step)")
for k, v in by_op.most_common(25):
    print(f"{v / N:7.1f}  {k}")
print("-- by source line (kernels per step)")
for k, v in by_line.most_common(45):
    print(f"{v / N:7.1f}  {k}")
