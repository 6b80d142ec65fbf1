"""bf16 operand path (BASELINE.json configs[2]/[4]): bf16 core / factors, fp32 accumulation,
query vectors rounded to bf16, fp32 scores.  Checked against a float64 evaluation of the SAME
bf16-rounded parameters (oracle.logits_exact).

Stated tolerance.  The only rounding besides fp32 accumulation is v -> bf16 (half-ulp 2^-9 per
element), so the rigorous bound is
    |dz[d,j]| <= 2^-8 * sum_k |v[d,k]| * |O[j,k]|
and the tests check it element-wise; relative to (1 + |z|) that is 0.5-3e-2 at these ranks
(SURVEY.md section 8c quotes 2e-2), capped here at 5e-2, with MRR as the binding check.
"""
import numpy as np
import pytest
import torch

import gen
from oracle import score_oracle as orc

pytestmark = pytest.mark.gpu
Z_TOL_BF16 = 5e-2


@pytest.fixture(scope="module")
def rt():
    assert torch.cuda.is_available()
    import r_tucker_amd
    r_tucker_amd._lib.load()
    return r_tucker_amd


def bf16_round(x):
    return torch.from_numpy(x).to(torch.bfloat16)


@pytest.mark.parametrize("shape", [
    (3000, 22, 96, (10, 200, 200)),      # WN18RR-like rank
    (2000, 474, 256, (200, 200, 200)),   # FB15k-237 symmetric rank: a > 32 -> MFMA tables from bf16 operands
    (700, 900, 64, (40, 64, 64)),        # n_rel > batch: device-side relation plan
    (513, 9, 97, (4, 20, 20)),           # c % 8 != 0: scalar fragment loads
    (300, 5, 33, (3, 7, 7)), (129, 3, 1, (2, 512, 512)),
])
@pytest.mark.parametrize("sym", [False, True])
def test_bf16_scores_against_float64_of_bf16_params(rt, shape, sym):
    n_ent, n_rel, B, rank = shape
    core, R, S, O = gen.make_params(n_ent, n_rel, rank, 21, shared=sym)
    h, r = gen.make_queries(n_ent, n_rel, B, 21)
    tb = [bf16_round(x) for x in (core, R, S, O)]
    if sym:
        tb[3] = tb[2]
    d = [t.cuda() for t in tb]
    hh, rr = torch.from_numpy(h).cuda(), torch.from_numpy(r).cuda()
    z = rt.score_1vN(d[0], d[1], d[2], d[3], hh, rr, sigmoid=False).cpu().numpy().astype(np.float64)
    p = rt.score_1vN(d[0], d[1], d[2], d[3], hh, rr).cpu().numpy().astype(np.float64)
    assert z.dtype == np.float64 and z.shape == (B, n_ent)
    f = [t.float().numpy() for t in tb]
    ze = orc.logits_exact(f[0], f[1], f[2], f[3], h, r)
    err = np.max(np.abs(z - ze) / (1 + np.abs(ze)))
    ve = np.abs(orc.query_vectors_exact(f[0], f[1], f[2], h, r))
    bound = 2.0 ** -8 * (ve @ np.abs(f[3].astype(np.float64)).T) + 1e-30
    nerr = np.max(np.abs(z - ze) / bound)
    print(f"\nbf16 {shape} sym={sym}: max |dz|/(1+|z|) = {err:.2e}   max |dz| / (2^-8 sum|v||o|) = {nerr:.2f}")
    assert nerr <= 1.0
    assert err <= Z_TOL_BF16
    assert np.max(np.abs(p - 1 / (1 + np.exp(-ze)))) <= 0.25 * Z_TOL_BF16 * 2   # |dp| <= |dz| / 4


@pytest.mark.parametrize("shape", [
    (3000, 22, 96, (10, 200, 200)), (2001, 5, 70, (3, 24, 24)),       # even / odd entity count (last column alone)
    (129, 3, 33, (2, 512, 512)), (1, 1, 1, (1, 8, 8)), (4097, 7, 300, (4, 272, 272)),
])
@pytest.mark.parametrize("dense", [False, True])
def test_bf16_scores_written_as_bf16(rt, shape, dense, monkeypatch):
    """out_dtype=bfloat16: the kernel rounds the probabilities itself (pairs of columns traded
    between neighbouring lanes, v_cvt_pk_bf16_f32) -- bit-identical to rounding the fp32 scores,
    with 128-byte aligned rows (default) and with the dense row pitch (odd N: unaligned pairs)."""
    n_ent, n_rel, B, rank = shape
    from r_tucker_amd import ops
    monkeypatch.setattr(ops, "ROW_ALIGN", 1 if dense else 32)
    core, R, S, O = gen.make_params(n_ent, n_rel, rank, 33)
    h, r = gen.make_queries(n_ent, n_rel, B, 33)
    d = [bf16_round(x).cuda() for x in (core, R, S, O)]
    hh, rr = torch.from_numpy(h).cuda(), torch.from_numpy(r).cuda()
    p32 = rt.score_1vN(*d, hh, rr)
    pb = rt.score_1vN(*d, hh, rr, out_dtype=torch.bfloat16)
    assert pb.dtype == torch.bfloat16 and pb.shape == (B, n_ent)
    assert torch.equal(pb, p32.to(torch.bfloat16))
    with pytest.raises(RuntimeError):
        rt.score_1vN(*d, hh, rr, sigmoid=False, out_dtype=torch.bfloat16)


def test_bf16_closure_and_grad(rt):
    """model surface with bf16 parameters; gradients flow (computed in fp32, returned in bf16)."""
    n_ent, n_rel, B, rank = 400, 6, 20, (4, 16, 16)
    core, R, S, O = [bf16_round(x).cuda().requires_grad_(True) for x in gen.make_params(n_ent, n_rel, rank, 3)]
    h, r = [torch.from_numpy(x).cuda() for x in gen.make_queries(n_ent, n_rel, B, 3)]
    model = rt.AsymmetricR_TuckER((n_ent, n_rel), rank)
    P = model(h, r)(rt.Tucker(core, [R, S, O]))
    assert P.dtype == torch.float32 and P.shape == (B, n_ent)
    P.sum().backward()
    ref = [t.detach().float().cpu().requires_grad_(True) for t in (core, R, S, O)]
    orc.score_ref(ref[0], ref[1], ref[2], ref[3], h.cpu(), r.cpu()).sum().backward()
    for g, rg in zip((core, R, S, O), ref):
        assert g.grad.dtype == torch.bfloat16
        denom = rg.grad.abs().max().item() + 1e-6
        assert (g.grad.float().cpu() - rg.grad).abs().max().item() / denom < 5e-2


def test_mixed_dtypes_rejected(rt):
    core, R, S, O = [torch.from_numpy(x).cuda() for x in gen.make_params(50, 4, (3, 5, 5), 0)]
    with pytest.raises(RuntimeError, match="share one dtype"):
        rt.score_1vN(core.bfloat16(), R, S, O, torch.tensor([0]).cuda(), torch.tensor([0]).cuda())

