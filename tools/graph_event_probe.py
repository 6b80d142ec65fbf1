#!/usr/bin/env python3
"""Does an event recorded on the stream right after a HIP-graph launch complete only when the graph has?  Replays the
captured optimizer step (WN18RR recipe shape); after every launch records an event, waits for it on the host, and
then asks the stream whether it is idle (``hipStreamQuery``).  Counts the launches after which the event had
completed while the stream still had work -- the condition under which the next launch overlaps this one.
Usage: python tools/graph_event_probe.py [n_replays]"""
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

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
DRAIN = os.environ.get("DRAIN", "1") == "1"        # 0: wait for the event only, as graphstep's WAIT=event does
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
ids = [torch.randint(0, flt.features.shape[0], (512,), device="cuda", generator=gen) for _ in range(n + 3)]
for i in range(3):
    step.run(ids[i])
torch.cuda.synchronize()
st = torch.cuda.current_stream()
early, waits, drains = 0, [], []
for i in range(n):
    step.ids.copy_(ids[3 + i])
    step.graph.replay()
    ev = torch.cuda.Event()
    ev.record(st)
    t0 = time.perf_counter()
    ev.synchronize()
    t1 = time.perf_counter()
    busy = not st.query()
    if DRAIN:
        st.synchronize()
    t2 = time.perf_counter()
    early += int(busy)
    waits.append(t1 - t0)
    drains.append(t2 - t1)
torch.cuda.synchronize()
print(f"drain={DRAIN}: {n} launches: event completed while the stream was still busy after {early} of them; "
      f"mean wait for the event {1e3 * sum(waits) / n:.2f} ms, mean further wait for the stream {1e3 * sum(drains) / n:.2f} ms; "
      f"loss sum {step.totals()[0]:.4f}", flush=True)
