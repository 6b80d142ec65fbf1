"""Pack a tools/train_lease.py checkpoint into the <= 20 MB fixture tests/golden/wn18rr_trained_q8.npz that
tests/test_gpu_trained_checkpoint.py falls back to on a clean clone: core and R in fp32, the two 40 943 x 200 factor
matrices as int8 with one fp32 scale per row (row maximum -> 127).  The test re-orthonormalises the factors after
loading, so the fixture is a trained MODEL (same distribution of scores: a core of norm 8e5, most probabilities
saturated), not the bit pattern of the checkpoint; device and oracle are compared on the dequantised parameters.

    python tools/pack_checkpoint_q8.py ckpt_tmp/ckpt_onecyc_e500.npz

The checkpoint itself came from (profiles/r03_train_onecycle_lease{1,2}.log):
    python tools/train_lease.py --tag onecyc --onecycle --default-config --minutes 17        (x2 leases, 500 epochs)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def quantise(w: np.ndarray):
    scale = np.abs(w).max(axis=1, keepdims=True).astype(np.float32) / 127.0
    scale[scale == 0] = 1.0
    return np.clip(np.rint(w / scale), -127, 127).astype(np.int8), scale[:, 0]


def dequantise(q: np.ndarray, scale: np.ndarray) -> np.ndarray:
    return q.astype(np.float32) * scale[:, None].astype(np.float32)


def main():
    from train_lease import unpack24
    z = np.load(sys.argv[1], allow_pickle=False)
    out = {"core": z["core"].astype(np.float32), "R": z["R"].astype(np.float32), "epoch": z["epoch"]}
    for name in ("S", "O"):
        q, s = quantise(unpack24(z[name]).numpy())
        out[name + "_q8"], out[name + "_scale"] = q, s
    dst = os.path.join(ROOT, "tests", "golden", "wn18rr_trained_q8.npz")
    np.savez_compressed(dst, **out)
    print(dst, os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    main()
