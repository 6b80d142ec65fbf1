"""Random-shape sweep of the scoring path against the oracle's float64 restatement: the curated
cases live in test_gpu_parity.py / test_gpu_bf16.py, this looks for shapes nobody thought of
(sizes around every tile, k-step and batch threshold; fp32 up to c = 416, bf16 up to 512, bf16
scores; grouped contract, planned relation slots, symmetric and asymmetric).
`python tests/test_gpu_fuzz.py SEED CASES` runs a longer sweep by hand."""
import sys

import numpy as np
import pytest
import torch

import gen
from oracle import score_oracle as orc

pytestmark = pytest.mark.gpu


def sweep(seed, n_cases, rt, verbose=False):
  rng = np.random.default_rng(seed)
  bad = []
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
              ok = ok and np.max(np.abs(p - 1 / (1 + np.exp(-ze)))) <= 1e-5   # |dp| <= |dz| / 4 plus the logistic's own error
          rt.check_device_errors()
      except Exception as e:  # noqa: BLE001
          ok, err = False, repr(e)
      line = f"{'ok  ' if ok else 'FAIL'} {'bf16' if bf16 else 'f32 '} N={n_ent} nR={n_rel} B={B} a={a} c={c} sym={sym} err={err}"
      if verbose:
          print(line, flush=True)
      if not ok:
          bad.append(line)
  return bad


@pytest.mark.parametrize("seed", [0, 1])
def test_random_shapes(seed):
    assert torch.cuda.is_available()
    import r_tucker_amd as rt
    rt._lib.load()
    bad = sweep(seed, 45, rt)
    assert not bad, "\n".join(bad)


if __name__ == "__main__":
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import r_tucker_amd
    failures = sweep(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 100,
                     r_tucker_amd, verbose=True)
    print(f"{len(failures)} failures")
    sys.exit(1 if failures else 0)
