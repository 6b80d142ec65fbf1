"""Host helpers of the lease trainer (tools/train_lease.py): the 24-bit checkpoint packing a GPU lease hands back
(<= 64 MiB per call) and the learning-rate schedules, resumed mid-run."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import train_lease as tl  # noqa: E402


def test_pack24_round_trip():
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1000, 7, generator=g) * torch.logspace(-20, 20, 7)
    x[0, 0], x[1, 1], x[2, 2] = 0.0, -0.0, 1.0
    b = tl.pack24(x)
    assert b.dtype == np.uint8 and b.shape == (1000, 7, 3)
    y = tl.unpack24(b)
    assert y.shape == x.shape and y.dtype == torch.float32
    assert torch.equal(y[0, 0], x[0, 0]) and y[2, 2] == 1.0
    rel = ((y - x).abs() / x.abs().clamp_min(1e-38))
    assert rel.max().item() <= 2.0 ** -16                  # round to nearest at 24 of 32 bits: 15 mantissa bits kept
    # orthonormal columns survive to 1e-4 and are repaired by one QR (what --resume does)
    q = torch.linalg.qr(torch.randn(500, 20, generator=g))[0]
    q2 = tl.unpack24(tl.pack24(q)).double()
    assert (q2.T @ q2 - torch.eye(20, dtype=torch.float64)).abs().max().item() < 1e-4


def test_onecycle_resume_is_the_uninterrupted_schedule():
    """--onecycle on a resumed lease fast-forwards the reference's scheduler (train.py:213-215): the learning rate of
    epoch e must not depend on where the leases were cut."""
    def lrs(n_epochs, start):
        p = [torch.nn.Parameter(torch.zeros(1))]
        opt = torch.optim.SGD(p, lr=1.0)
        s = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=600, total_steps=n_epochs, pct_start=100 / n_epochs, div_factor=5.5,
                                                cycle_momentum=False, anneal_strategy="linear")
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for _ in range(start):
                s.step()
            out = []
            for e in range(start, n_epochs):
                out.append(opt.param_groups[0]["lr"])
                opt.step()
                if e + 1 < n_epochs:
                    s.step()
        return out
    whole = lrs(500, 0)
    assert abs(whole[0] - 600 / 5.5) < 1e-9 and abs(whole[99] - 600.0) < 1e-9 and whole[-1] < 0.02
    assert lrs(500, 333) == whole[333:]
