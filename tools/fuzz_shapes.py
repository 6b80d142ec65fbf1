#!/usr/bin/env python3
"""Random-shape sweep of the scoring path against the float64 restatement (GPU box; not a pytest file:
tests/ hold the curated cases, this looks for shapes nobody thought of).  Exit code 1 on a failure."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import gen  # noqa: E402
import r_tucker_amd as rt  # noqa: E402
from oracle import score_oracle as orc  # noqa: E402  (checker only)

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad = 0
for case in range(n_cases):
    bf16 = case % 3 == 2
    n_ent = int(rng.choice([1, 2, 31, 33, 127, 129, 500, 1000, 2049, 4097, 7001]))
    n_rel = int(rng.choice([1, 2, 7, 40, 300, 2500]))
    B = int(rng.choice([1, 2, 31, 32, 33, 100, 511, 1024, 2047, 2048, 2500]))
    a = int(rng.choice([1, 2, 7, 10, 31, 32, 33, 40, 64]))
    cmax = 512 if bf16 else 416
    c = int(rng.choice([1, 3, 4, 7, 8, 16, 17, 32, 100, 200, 208, 209, 256, 257, 300, 400, cmax]))
    if n_ent * c > 3_000_000 or B * n_ent > 8_000_000 or a * c * c > 6_000_000:
        n_ent, B = min(n_ent, 1000), min(B, 1024)
    sym = bool(rng.integers(0, 2))
    core, R, S, O = gen.make_params(n_ent, n_rel, (a, c, c), 100 + case, shared=sym)
    h, r = gen.make_queries(n_ent, n_rel, B, 100 + case)
    try:
        if bf16:
            tb = [torch.from_numpy(x).to(torch.bfloat16) for x in (core, R, S, O)]
            if sym:
                tb[3] = tb[2]
            d = [t.cuda() for t in tb]
            f = [t.float().numpy() for t in tb]
            z = rt.score_1vN(*d, torch.from_numpy(h).cuda(), torch.from_numpy(r).cuda(), sigmoid=False).cpu().numpy().astype(np.float64)
            ze = orc.logits_exact(f[0], f[1], f[2], f[3], h, r)
            ve = np.abs(orc.query_vectors_exact(f[0], f[1], f[2], h, r))
            err = np.max(np.abs(z - ze) / (2.0 ** -8 * (ve @ np.abs(f[3].astype(np.float64)).T) + 1e-30))
            ok = err <= 1.0
            pb = rt.score_1vN(*d, torch.from_numpy(h).cuda(), torch.from_numpy(r).cuda(), out_dtype=torch.bfloat16)
            p32 = rt.score_1vN(*d, torch.from_numpy(h).cuda(), torch.from_numpy(r).cuda())
            ok = ok and torch.equal(pb, p32.to(torch.bfloat16))
        else:
            d = [torch.from_numpy(x).cuda() for x in (core, R, S, O)]
            if sym:
                d[3] = d[2]
            z = rt.score_1vN(*d, torch.from_numpy(h).cuda(), torch.from_numpy(r).cuda(), sigmoid=False).cpu().numpy().astype(np.float64)
            ze = orc.logits_exact(core, R, S, O, h, r)
            err = np.max(np.abs(z - ze) / (1 + np.abs(ze)))
            ok = err <= 2e-5
            p = rt.score_1vN(*d, torch.from_numpy(h).cuda(), torch.from_numpy(r).cuda()).cpu().numpy().astype(np.float64)
            ok = ok and np.max(np.abs(p - 1 / (1 + np.exp(-ze)))) <= 3e-6 + 0.25 * 2e-5 * (1 + np.abs(ze).max()) * 0 + 6e-6
        rt.check_device_errors()
    except Exception as e:  # noqa: BLE001
        ok, err = False, repr(e)
    print(f"{'ok  ' if ok else 'FAIL'} {'bf16' if bf16 else 'f32 '} N={n_ent} nR={n_rel} B={B} a={a} c={c} sym={sym} err={err}", flush=True)
    bad += not ok
print(f"{n_cases - bad} / {n_cases} ok")
sys.exit(1 if bad else 0)
