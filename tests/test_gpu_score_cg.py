"""GPU tests of the column-group score kernel (csrc/rtk_score_cg_kernel.h) through the C ABI's kernel hints.

The same packed query planes are scored by all three split-fp16 kernels (RTK_SCORE_KERNEL_CG / _WS / _V3) and
checked against a float64 product of the fp32 operands.  The rows carry power-of-two scales over 2^-6 .. 2^6,
so a logit is not O(1) and the test states the split product's guarantee as the header does, normwise per row
(include/rtucker_hip.h):
    logits         |dz| <= 2e-5 * (1 + |z|)  +  2^-20 * K * max|v_d| * max|O_j|
    probabilities  |dp| <= 3e-6  +  the same bound (the logistic's slope is <= 1/4)
and against each other: the columns of a set's four register-resident groups are computed by the ws kernel's
instruction sequence (bit-equal), the columns of a fifth group are four K-range chains added in a fixed order
(equal to the stated tolerance, identical from run to run).
Reference path replaced: src/model/asymmetric/R_TuckER.py:47-48.
"""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

Z_TOL, P_TOL = 2e-5, 3e-6


@pytest.fixture(scope="module")
def rt():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    import r_tucker_amd
    r_tucker_amd._lib.load()
    return r_tucker_amd


def _score(rt, qp, B, O, flags, pitch=None, fill=None):
    """rtk_score_packed_f32 with explicit flags -> (B, N) view of a (B, pitch) buffer"""
    lib = rt._lib.load()
    N, c = O.shape
    pitch = pitch or N
    buf = torch.empty((B, pitch), dtype=torch.float32, device=O.device)
    if fill is not None:
        buf.fill_(fill)
    rt._lib.check(lib.rtk_score_packed_f32(qp.data_ptr(), B, c, O.data_ptr(), N, buf.data_ptr(), pitch, flags,
                                           torch.cuda.current_stream().cuda_stream), "rtk_score_packed_f32")
    return buf


def _fifth_group_columns(N, W_max=256):
    """columns that the cg schedule puts into a set's fifth (K-split) group"""
    G = -(-N // 32)
    sets = -(-G // 5)
    if sets < W_max:
        sets = min(G, W_max)
    W = min(W_max, sets)
    U = W * -(-sets // W)
    fifth = np.zeros(N, dtype=bool)
    for u in range(U):
        gb, ge = G * u // U, G * (u + 1) // U
        if ge - gb == 5:
            fifth[(gb + 4) * 32:min(N, (gb + 5) * 32)] = True
    return fifth


SHAPES = [
    # (N, c, B, pitch)                      what it exercises (sets: ceil(G / 5) of them when that fills 256 workgroups,
    #                                       else min(G, 256) sets of 1-5 groups)
    (40943, 200, 512, 40960),             # the bench workload: 1280 groups = 5 per workgroup, nontemporal stores
    (40943, 200, 500, None),              # dense rows (plain stores), ragged last query tile
    (20000, 200, 64, None),               # 625 groups in 256 sets of 2 and 3 (one or two M waves without a group)
    (36000, 64, 40, None),                # 1125 groups in 256 sets of 4 and 5: sets with and without a fifth group side by side
    (2000, 200, 64, None),                # 63 groups, one per workgroup
    (333, 36, 70, None),                  # KS = 3: two M waves have an empty k-range; N % 32 != 0
    (100, 4, 5, None),                    # KS = 1, last group 4 columns wide
    (20, 8, 33, None),                    # fewer entities than one group
    (167, 208, 96, 192),                  # KS = 13 at the widest c
    (100000, 64, 40, None),               # 3125 groups: 768 sets of 4 and 5, three per workgroup, one after the other
    (5 * 32 * 256 + 1, 16, 32, None),     # one group more than a full single pass holds: two passes
]


@pytest.mark.parametrize("N,c,B,pitch", SHAPES)
def test_all_kernels_against_float64(rt, N, c, B, pitch):
    g = torch.Generator().manual_seed(7 * N + c)
    v = torch.randn((B, c), generator=g) * torch.exp2(torch.randint(-6, 7, (B, 1), generator=g).float())
    O = torch.randn((N, c), generator=g) * torch.exp2(torch.randint(-6, 7, (N, 1), generator=g).float())
    z64 = v.double().numpy() @ O.double().numpy().T
    # normwise bound of the split product (include/rtucker_hip.h), with a factor two of slack
    nb = 2.0 ** -20 * c * v.abs().max(dim=1).values.double().numpy()[:, None] * O.abs().max(dim=1).values.double().numpy()[None, :]
    vd, Od = v.cuda(), O.cuda()
    qp = rt.pack_query_vectors(vd, torch.float32)
    L = rt._lib
    out = {}
    for name, hint in (("cg", L.RTK_SCORE_KERNEL_CG), ("ws", L.RTK_SCORE_KERNEL_WS), ("v3", L.RTK_SCORE_KERNEL_V3)):
        buf = _score(rt, qp, B, Od, hint, pitch, fill=-7.0)
        torch.cuda.synchronize()
        if pitch and pitch > N:
            assert torch.all(buf[:, N:] == -7.0), f"{name}: wrote past column N"
        out[name] = buf[:, :N].cpu().numpy()
        err = np.max(np.abs(out[name] - z64) / (Z_TOL * (1 + np.abs(z64)) + nb))
        print(f"{name}: max |dz| / (2e-5 (1+|z|) + normwise bound) = {err:.2e}")
        assert err <= 1.0, name
    fifth = _fifth_group_columns(N)
    assert np.array_equal(out["cg"][:, ~fifth], out["ws"][:, ~fifth]), "register groups must equal the ws kernel bit for bit"
    if fifth.any():
        d = np.abs(out["cg"][:, fifth] - out["ws"][:, fifth]) / (1 + np.abs(z64[:, fifth]))
        print(f"fifth-group columns: {int(fifth.sum())}, max relative difference to ws {d.max():.2e}")
    # the three logistic variants of the cg kernel
    p64 = 1.0 / (1.0 + np.exp(-z64))
    for flags in (L.RTK_SCORE_SIGMOID, L.RTK_SCORE_SIGMOID | L.RTK_SCORE_SIGMOID_FAST):
        p = _score(rt, qp, B, Od, flags | L.RTK_SCORE_KERNEL_CG, pitch)[:, :N].cpu().numpy()
        pw = _score(rt, qp, B, Od, flags | L.RTK_SCORE_KERNEL_WS, pitch)[:, :N].cpu().numpy()
        assert np.max(np.abs(p - p64) / (P_TOL + nb)) <= 1.0
        assert np.array_equal(p[:, ~fifth], pw[:, ~fifth])
    # deterministic
    again = _score(rt, qp, B, Od, L.RTK_SCORE_KERNEL_CG, pitch)[:, :N].cpu().numpy()
    assert np.array_equal(again, out["cg"])


def test_default_dispatch_takes_cg_at_the_wn18rr_shape(rt):
    """no hint: 18 432 <= N <= 40 960 runs the cg kernel (one set per workgroup), other shapes the ws kernel.  Visible
    where a set has a fifth group (more than 1024 groups: N = 40 943) -- those columns are the only ones that differ
    between the two; up to 1024 groups (N = 20 000, 14 951) the cg kernel has the ws kernel's bits everywhere."""
    g = torch.Generator().manual_seed(11)
    L = rt._lib
    for N in (40943, 20000, 14951, 46000, 100000):
        fifth = bool(_fifth_group_columns(N).any())
        assert fifth == (N in (40943, 100000))
        v = torch.randn((64, 200), generator=g).cuda()
        O = torch.randn((N, 200), generator=g).cuda()
        qp = rt.pack_query_vectors(v, torch.float32)
        auto = _score(rt, qp, 64, O, 0).cpu().numpy()
        cg = _score(rt, qp, 64, O, L.RTK_SCORE_KERNEL_CG).cpu().numpy()
        ws = _score(rt, qp, 64, O, L.RTK_SCORE_KERNEL_WS).cpu().numpy()
        assert np.array_equal(cg, ws) == (not fifth)
        assert np.array_equal(cg[:, ~_fifth_group_columns(N)], ws[:, ~_fifth_group_columns(N)])
        assert np.array_equal(auto, ws if N > 40960 else cg)
        assert bool(rt.ops.cg_fifth_group_columns(N, 200).any()) == (fifth and N <= 40960)
