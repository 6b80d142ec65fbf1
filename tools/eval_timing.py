#!/usr/bin/env python3
"""Wall time of the evaluation tail (score kernel -> filtered-rank kernel over the score matrix) on WN18RR test queries
with seeded parameters: does a cache policy of the score stores cost the consumer anything?  RTK_WS_NT=0|1."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import r_tucker_amd as rt                                   # noqa: E402
from r_tucker_amd.data import Data, KG_dataset              # noqa: E402

data = Data(os.path.join(ROOT, "data", "WN18RR") + "/", reverse=True)
test_set = KG_dataset(data, data.test_data, test_set=True)
torch.manual_seed(1)
model = rt.AsymmetricR_TuckER((len(data.entities), len(data.relations)), (10, 200, 200))
model.init()
with torch.no_grad():
    model.core.mul_(3000.0)
model.cuda()
for _ in range(3):
    m, _ = rt.evaluate(model, test_set, batch_size=512)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 20
for _ in range(n):
    m, _ = rt.evaluate(model, test_set, batch_size=512)
torch.cuda.synchronize()
print(f"RTK_WS_NT={os.environ.get('RTK_WS_NT', 'default')}: evaluate() over {len(test_set.features)} test queries: "
      f"{(time.perf_counter() - t0) / n * 1e3:.2f} ms per pass, MRR {m['mrr']:.6f}")
