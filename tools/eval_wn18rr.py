#!/usr/bin/env python3
"""End-to-end WN18RR evaluation on the device (score path + on-device filtered ranking) with the
"planted" stand-in parameters; prints metrics and wall time per split.  GPU box only."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import gen  # noqa: E402
import r_tucker_amd as rt  # noqa: E402
from r_tucker_amd.data import Data, KG_dataset  # noqa: E402

data = Data(os.path.join(ROOT, "data", "WN18RR") + "/", reverse=True)
n_ent, n_rel, rank = len(data.entities), len(data.relations), (10, 200, 200)
train = KG_dataset(data, data.train_data, label_smoothing=0.1)
sets = {"valid": KG_dataset(data, data.valid_data, test_set=True), "test": KG_dataset(data, data.test_data, test_set=True)}
planted = np.concatenate([np.asarray(train.data_index, dtype=np.int64), sets["valid"].features[::2], sets["test"].features[::2]])
params = gen.make_planted_params(planted, n_ent, n_rel, rank, 322)
model = rt.AsymmetricR_TuckER((n_ent, n_rel), rank)
model.init({"core": torch.from_numpy(params[0]), "R.weight": torch.from_numpy(params[1]),
            "S.weight": torch.from_numpy(params[2]), "O.weight": torch.from_numpy(params[3])})
model.cuda().eval()
for name, ds in sets.items():
    flt = rt.DeviceFilter(ds, "cuda")
    rt.evaluate(model, ds, flt=flt)              # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        m, loss = rt.evaluate(model, ds, flt=flt)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"{name}: {len(ds)} queries in {dt * 1e3:.2f} ms ({len(ds) / dt / 1e6:.2f} M queries/s end to end)  "
          f"MRR {m['mrr']:.4f} hits@1 {m['hits@1']:.4f} hits@10 {m['hits@10']:.4f} loss {loss:.4f}")
