"""Round-2 GPU tests: cached relation tables, error semantics, deterministic native backward (incl. the
WN18RR training shape against a reference-generated fixture), BASELINE configs[2] at full size and
configs[3] (FB15k, 8-way entity shard) consistency.  All calls go through the C ABI."""
import os

import numpy as np
import pytest
import torch

import gen
from oracle import score_oracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def rt():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    import r_tucker_amd
    r_tucker_amd._lib.load()
    return r_tucker_amd


def dev(*xs):
    return [torch.as_tensor(x).cuda() for x in xs]


# ------------------------------------------------------------------ cached relation tables ----------
@pytest.mark.parametrize("case", [
    # (n_ent, n_rel, B, rank, dtype)   small relation rank (VALU tables), planned batch (n_rel > B),
    # large relation rank fp32 (GEMM tables), bf16 large rank (MFMA tables), grouped contract (B >= 2048)
    (3000, 22, 96, (10, 200, 200), torch.float32),
    (1500, 300, 64, (6, 48, 48), torch.float32),
    (1200, 40, 80, (64, 64, 64), torch.float32),
    (2000, 37, 256, (96, 128, 128), torch.bfloat16),
    (900, 11, 2100, (8, 64, 64), torch.float32),
    (900, 50, 2048, (40, 64, 64), torch.bfloat16),
])
def test_cached_tables_bit_identical_to_per_batch(rt, case):
    """rtk_relation_tables_* + rtk_query_vectors_from_tables_* give the bits of rtk_query_vectors_*: the
    tables of all relations are the tables of the batch's relations, the contraction is the same kernel."""
    n_ent, n_rel, B, rank, dt = case
    core, R, S, O = [t.to(dt) for t in dev(*gen.make_params(n_ent, n_rel, rank, 3))]
    h, r = dev(*gen.make_queries(n_ent, n_rel, B, 3))
    tables = rt.relation_tables(core, R)
    assert tables.shape == (n_rel, rank[1], rank[2]) and tables.dtype == torch.float32
    v0 = rt.query_vectors(core, R, S, h, r)
    v1 = rt.query_vectors(core, R, S, h, r, tables=tables)
    assert torch.equal(v0, v1)
    p0 = rt.score_1vN(core, R, S, O, h, r)
    p1 = rt.score_1vN(core, R, S, O, h, r, tables=tables)
    assert torch.equal(p0, p1)
    if dt == torch.float32:    # and both agree with the oracle
        ref = orc.score_ref(*[x.cpu() for x in (core, R, S, O, h, r)])
        assert (p1.cpu() - ref).abs().max().item() <= 3e-6


@pytest.mark.parametrize("mode", ["asym", "sym"])
def test_model_closure_caches_tables_and_invalidates(rt, mode):
    n_ent, n_rel, B, rank = 1000, 9, 50, (4, 32, 32)
    core, R, S, O = gen.make_params(n_ent, n_rel, rank, 8, shared=(mode == "sym"))
    h, r = dev(*gen.make_queries(n_ent, n_rel, B, 8))
    if mode == "sym":
        model = rt.SymmetricR_TuckER((n_ent, n_rel), rank)
        with torch.no_grad():
            model.core.copy_(torch.from_numpy(core)); model.R.weight.copy_(torch.from_numpy(R)); model.E.weight.copy_(torch.from_numpy(S))
        model.cuda()
        mk = lambda: rt.SFTucker(model.core.data, [model.R.weight], num_shared_factors=2, shared_factor=model.E.weight)  # noqa: E731
        ref_of = lambda: orc.score_ref(model.core.detach().cpu(), model.R.weight.detach().cpu(), model.E.weight.detach().cpu(),  # noqa: E731
                                       model.E.weight.detach().cpu(), h.cpu(), r.cpu())
    else:
        model = rt.AsymmetricR_TuckER((n_ent, n_rel), rank)
        with torch.no_grad():
            model.core.copy_(torch.from_numpy(core)); model.R.weight.copy_(torch.from_numpy(R))
            model.S.weight.copy_(torch.from_numpy(S)); model.O.weight.copy_(torch.from_numpy(O))
        model.cuda()
        mk = lambda: rt.Tucker(model.core.data, [model.R.weight, model.S.weight, model.O.weight])  # noqa: E731
        ref_of = lambda: orc.score_ref(model.core.detach().cpu(), model.R.weight.detach().cpu(), model.S.weight.detach().cpu(),  # noqa: E731
                                       model.O.weight.detach().cpu(), h.cpu(), r.cpu())
    # training mode / grad enabled: no cache
    with torch.no_grad():
        model(h, r)(mk())
    assert model._tables is None
    model.eval()
    with torch.no_grad():
        p = model(h, r)(mk())
        assert model._tables is not None
        t0 = model._tables
        p2 = model(h, r)(mk())
        assert model._tables is t0 and torch.equal(p, p2)          # reused
    assert (p.cpu() - ref_of()).abs().max().item() <= 3e-6
    # an in-place update autograd sees: rebuilt
    with torch.no_grad():
        model.core.mul_(1.5)
        p3 = model(h, r)(mk())
    assert model._tables is not t0
    assert (p3.cpu() - ref_of()).abs().max().item() <= 3e-6
    # the reference's optimizer writes through .data (invisible to the version counter) between
    # model.train() and the next model.eval(): the mode switch drops the cache
    model.train()
    model.R.weight.data.add_(0.25)
    model.eval()
    assert model._tables is None
    with torch.no_grad():
        p4 = model(h, r)(mk())
    assert (p4.cpu() - ref_of()).abs().max().item() <= 3e-6
    # a foreign T (not this model's parameters) is never served from the cache
    with torch.no_grad():
        Tf = mk()
        Tf.core = model.core.data.clone() * 2.0
        pf = model(h, r)(Tf)
    assert not torch.equal(pf, p4)


# ------------------------------------------------------------------ error behaviour -----------------
def test_out_of_range_ids_raise_index_error_through_the_closure(rt):
    """Reference behaviour: an out-of-range subject / relation id raises IndexError."""
    n_ent, n_rel, B, rank = 500, 6, 20, (3, 16, 16)
    core, R, S, O = gen.make_params(n_ent, n_rel, rank, 2)
    model = rt.AsymmetricR_TuckER((n_ent, n_rel), rank)
    with torch.no_grad():
        model.core.copy_(torch.from_numpy(core)); model.R.weight.copy_(torch.from_numpy(R))
        model.S.weight.copy_(torch.from_numpy(S)); model.O.weight.copy_(torch.from_numpy(O))
    model.cuda().eval()
    T = rt.Tucker(model.core.data, [model.R.weight, model.S.weight, model.O.weight])
    h, r = dev(*gen.make_queries(n_ent, n_rel, B, 2))
    rt.check_device_errors()
    with torch.no_grad():
        model(h, r)(T)                                     # clean ids: nothing raised
        for bad_h, bad_r in ((n_ent, 0), (-1, 0), (0, n_rel), (0, -3)):
            hh, rr = h.clone(), r.clone()
            if bad_h:
                hh[7] = bad_h
            if bad_r:
                rr[11] = bad_r
            with pytest.raises(IndexError):
                model(hh, rr)(T)
            model(h, r)(T)                                 # the word was cleared: the next clean call passes
        # deferred: nothing per call, IndexError at the caller's sync point; survives a workspace regrow
        with rt.index_check("deferred"):
            hh = h.clone(); hh[0] = n_ent + 5
            model(hh, r)(T)
            big_h, big_r = dev(*gen.make_queries(n_ent, n_rel, 40000, 5))     # forces a larger workspace
            rt.score_1vN(model.core.data, model.R.weight, model.S.weight, model.O.weight, big_h, big_r)
            with pytest.raises(IndexError):
                rt.check_device_errors()
            rt.check_device_errors()                       # cleared
        with rt.index_check("off"):
            model(hh, r)(T)
        with pytest.raises(IndexError):
            rt.check_device_errors()


def test_evaluate_surfaces_bad_ids(rt):
    from r_tucker_amd.data import Data, KG_dataset
    data = Data(os.path.join(ROOT, "data", "WN18RR") + "/", reverse=True)
    ds = KG_dataset(data, data.valid_data[:300], test_set=True)
    n_ent, n_rel, rank = len(data.entities), len(data.relations), (4, 32, 32)
    model = rt.AsymmetricR_TuckER((n_ent, n_rel), rank)
    model.init()
    model.S = torch.nn.Embedding(n_ent // 2, rank[1])      # too few subject rows: about half the subject ids are out of range
    model.cuda()
    assert int(ds.features[:, 0].max()) >= n_ent // 2
    with pytest.raises(IndexError):
        rt.evaluate(model, ds, batch_size=128)
    rt.check_device_errors()


# ------------------------------------------------------------------ backward ------------------------
def test_splitk_and_backward_are_deterministic(rt):
    """Two runs give bit-identical gradients (fixed-order split-K, ordered row scatter; no atomics)."""
    n_ent, n_rel, B, rank = 20011, 5, 333, (6, 72, 72)
    core, R, S, O = gen.make_params(n_ent, n_rel, rank, 12)
    h, r = gen.make_queries(n_ent, n_rel, B, 12)
    h[::5] = h[0]                                          # repeated subjects: multi-row scatter
    w = torch.from_numpy(np.random.default_rng(1).standard_normal((B, n_ent)).astype(np.float32)).cuda()
    grads = []
    for _ in range(2):
        leaves = [torch.from_numpy(x).cuda().requires_grad_(True) for x in (core, R, S, O)]
        (rt.score_1vN(*leaves, torch.from_numpy(h).cuda(), torch.from_numpy(r).cuda()) * w).sum().backward()
        grads.append([x.grad.clone() for x in leaves])
    for g0, g1 in zip(*grads):
        assert torch.equal(g0, g1)
    ref = orc.score_grads_ref(*[torch.from_numpy(x) for x in (core, R, S, O)], torch.from_numpy(h), torch.from_numpy(r), w.cpu())
    for g, e in zip(grads[0], ref):
        assert (g.cpu() - e).abs().max().item() <= 3e-4 * e.abs().max().item() + 1e-9


def test_query_vectors_bwd_abi_against_autograd(rt):
    """rtk_query_vectors_bwd_f32 alone: (g_core, g_R, g_S) from dv vs torch autograd through the oracle's stage 1."""
    import ctypes as C  # noqa: F401
    from r_tucker_amd import _lib
    lib = _lib.load()
    n_ent, n_rel, B, rank = 700, 9, 130, (7, 40, 40)
    core, R, S, _ = gen.make_params(n_ent, n_rel, rank, 4)
    h, r = gen.make_queries(n_ent, n_rel, B, 4)
    r[:40] = 3
    h[:10] = 17
    dv = np.random.default_rng(4).standard_normal((B, rank[2])).astype(np.float32)
    tc, tR, tS = [torch.from_numpy(x).clone().requires_grad_(True) for x in (core, R, S)]
    v = orc.query_vectors_ref(tc, tR, tS, torch.from_numpy(h), torch.from_numpy(r))
    (v * torch.from_numpy(dv)).sum().backward()
    dcore, dR, dS, dh, dr, ddv = dev(core, R, S, h, r, dv)
    a, b, c = rank
    gc, gR, gS = torch.empty_like(dcore), torch.empty_like(dR), torch.empty_like(dS)
    ws = torch.empty(lib.rtk_query_bwd_workspace_bytes(B, a, b, c), dtype=torch.uint8, device="cuda")
    sp = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.rtk_query_vectors_bwd_f32(dcore.data_ptr(), a, b, c, dR.data_ptr(), n_rel, dS.data_ptr(), n_ent,
                                             dr.data_ptr(), dh.data_ptr(), B, ddv.data_ptr(), gc.data_ptr(), gR.data_ptr(),
                                             gS.data_ptr(), ws.data_ptr(), ws.numel(), sp), "bwd")
    for g, e in ((gc, tc.grad), (gR, tR.grad), (gS, tS.grad)):
        assert (g.cpu() - e).abs().max().item() <= 2e-5 * e.abs().max().item() + 1e-9
    # skipping outputs is allowed
    _lib.check(lib.rtk_query_vectors_bwd_f32(dcore.data_ptr(), a, b, c, dR.data_ptr(), n_rel, dS.data_ptr(), n_ent,
                                             dr.data_ptr(), dh.data_ptr(), B, ddv.data_ptr(), None, gR.data_ptr(), None,
                                             ws.data_ptr(), ws.numel(), sp), "bwd")
    assert (gR.cpu() - tR.grad).abs().max().item() <= 2e-5 * tR.grad.abs().max().item()


def test_training_shape_gradients_against_reference_fixture(rt, golden, golden_meta):
    """The WN18RR TRAINING shape: core (20,400,400), 40 943 entities, a ragged batch of 500
    with repeated subjects, BCE on label-smoothed CSR targets, padded score pitch -- loss and gradients
    against vectors produced by the reference closure under torch autograd (make_golden_train_grad.py)."""
    import make_golden_train_grad as mg
    c = golden_meta["cases"]["wn18rr_train_grad"]
    g = golden("wn18rr_train_grad")
    core, R, S, O = gen.make_params(mg.N_ENT, mg.N_REL, mg.RANK, mg.SEED)
    h, r, lists = mg.make_batch()
    assert gen.digest(core, R, S, O, h, r) == c["inputs_sha256"]
    assert gen.digest(np.asarray([x for l in lists for x in l], dtype=np.int64)) == c["lists_sha256"]

    class Flt:   # the three arrays bce_loss_1vN reads from a DeviceFilter
        pass
    flt = Flt()
    flt.slot_of_item = torch.arange(mg.B, device="cuda")
    flt.pair_ptr = torch.from_numpy(np.concatenate([[0], np.cumsum([len(x) for x in lists])]).astype(np.int64)).cuda()
    flt.pair_obj = torch.tensor([x for l in lists for x in l], dtype=torch.int64, device="cuda")
    leaves = [torch.from_numpy(x).cuda().requires_grad_(True) for x in (core, R, S, O)]
    loss = rt.bce_loss_1vN(*leaves, torch.from_numpy(h).cuda(), torch.from_numpy(r).cuda(), flt,
                           torch.arange(mg.B, device="cuda"), label_smoothing=mg.EPS)
    assert abs(loss.item() - float(g["loss"])) <= 2e-6 * float(g["loss"])
    loss.backward()
    g_core, g_R, g_S, g_O = [x.grad for x in leaves]

    def close(got, want, what, rel=5e-4):
        got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
        scale = np.abs(want).max()
        assert np.abs(got - want).max() <= rel * scale, (what, np.abs(got - want).max(), scale)

    close(g_R.cpu().numpy(), g["g_R"], "g_R")
    close(g_core.cpu().numpy().reshape(-1)[g["core_idx"]], g["g_core_sample"], "g_core sample")
    close(g_core.double().sum(dim=(0, 1)).cpu().numpy(), g["g_core_colsum"], "g_core column sums")
    close(g_O.cpu().numpy().reshape(-1)[g["O_idx"]], g["g_O_sample"], "g_O sample")
    close(g_O.double().sum(dim=0).cpu().numpy(), g["g_O_colsum"], "g_O column sums")
    close(g_S.cpu().numpy()[g["S_rows"]], g["g_S_rows"], "g_S rows")
    close(g_S.double().sum(dim=0).cpu().numpy(), g["g_S_colsum"], "g_S column sums")
    assert int((g_S.abs().sum(dim=1) > 0).sum()) == int(g["g_S_nonzero_rows"])


@pytest.mark.parametrize("M,N,K,ak,bk,splits,mag", [
    (300, 400, 777, 0, 0, 1, 1.0),        # dO-like: both operands M-major, ragged everything
    (512, 400, 20011, 1, 0, 7, 1.0),      # dv-like: A K-major, split over K
    (130, 70, 515, 1, 1, 3, 1.0),
    (96, 40, 65, 1, 0, 4, 1.0),           # chunks of 32: the fourth starts past K (writes its zero slab only)
    (64, 33, 100, 0, 1, 1, 1.0),
    (257, 129, 1000, 0, 0, 4, 3e-9),      # magnitudes of a BCE gradient: nothing survives in fp16 without the scale
    (100, 100, 64, 1, 0, 1, 1e20),
])
def test_gemm_split_fp16_against_float64(rt, M, N, K, ak, bk, splits, mag):
    """rtk_gemm_sf16_splitk (the backward products' GEMM): all operand layouts, ragged tiles, split-K, operand
    magnitudes far outside fp16's range.  Error bound: each element keeps max(2^-22 |x|, 2^-40 bound) of
    absolute accuracy, so |dC| <= ~2^-21 sum_k |a||b| + K 2^-39 amax_a amax_b; checked at 2^-19 of that form."""
    lib = rt._lib.load()
    rng = np.random.default_rng(M * 7 + N)
    A = (rng.standard_normal((M, K)) * np.exp(rng.uniform(-6, 0, (M, K)))).astype(np.float32) * np.float32(mag)
    Bm = (rng.standard_normal((N, K)) * np.exp(rng.uniform(-6, 0, (N, K)))).astype(np.float32)
    dA = torch.from_numpy(A if ak else np.ascontiguousarray(A.T)).cuda()
    dB = torch.from_numpy(Bm if bk else np.ascontiguousarray(Bm.T)).cuda()
    lda, ldb = (K if ak else M), (K if bk else N)
    # a loose bound (3x the true maximum) must do: the caller of the BCE gradient only knows |dZ| <= |g| / (B N)
    ba = torch.tensor([3.0 * np.abs(A).max()], dtype=torch.float32, device="cuda")
    bb = torch.tensor([np.abs(Bm).max()], dtype=torch.float32, device="cuda")
    ldc = N if splits > 1 else N + 3
    C = torch.full((M, ldc), 7.0, dtype=torch.float32, device="cuda")
    ws = torch.empty(max(256, lib.rtk_gemm_f32_splitk_workspace_bytes(M, N, splits)), dtype=torch.uint8, device="cuda")
    sp = torch.cuda.current_stream().cuda_stream

    def run(out):
        return lib.rtk_gemm_sf16_splitk(dA.data_ptr(), ak, lda, ba.data_ptr(), dB.data_ptr(), bk, ldb, bb.data_ptr(),
                                        out.data_ptr(), ldc, M, N, K, splits, ws.data_ptr(), ws.numel(), sp)
    assert run(C) == 0, lib.rtk_last_error_string()
    ref = A.astype(np.float64) @ Bm.astype(np.float64).T
    got = C[:, :N].cpu().double().numpy()
    bound = 2.0 ** -19 * (np.abs(A).astype(np.float64) @ np.abs(Bm).astype(np.float64).T) \
        + K * 2.0 ** -37 * float(ba.item()) * float(bb.item())
    assert np.isfinite(got).all()
    assert np.max(np.abs(got - ref) / bound) <= 1.0
    assert np.abs(got - ref).max() <= 3e-6 * np.abs(ref).max()
    if ldc > N:
        assert (C[:, N:] == 7.0).all()          # nothing written past the row
    C2 = torch.full_like(C, 7.0)
    assert run(C2) == 0 and torch.equal(C, C2)  # fixed summation order
    if splits > 1:
        assert lib.rtk_gemm_sf16_splitk(dA.data_ptr(), ak, lda, ba.data_ptr(), dB.data_ptr(), bk, ldb, bb.data_ptr(),
                                        C2.data_ptr(), ldc, M, N, K, splits, ws.data_ptr(), 16, sp) == -2


def test_absmax(rt):
    """rtk_absmax_f32: strided rows (the padding of a score buffer is not read), odd sizes, NaNs skipped, zeros."""
    lib = rt._lib.load()
    sp = torch.cuda.current_stream().cuda_stream
    out = torch.full((1,), 7.0, device="cuda")
    for rows, cols, ld in ((1, 1, 1), (3, 5, 8), (512, 4099, 4128), (40943, 400, 400), (7, 1000, 1000)):
        buf = torch.randn(rows, ld, device="cuda")
        buf[:, cols:] = 1e30                                  # padding: must be ignored
        buf[rows // 2, cols // 2] = -123.5
        assert lib.rtk_absmax_f32(buf.data_ptr(), rows, cols, ld, out.data_ptr(), sp) == 0
        assert out.item() == buf[:, :cols].abs().max().item() == 123.5
    z = torch.zeros(4, 4, device="cuda")
    z[1, 1] = float("nan")
    assert lib.rtk_absmax_f32(z.data_ptr(), 4, 4, 4, out.data_ptr(), sp) == 0 and out.item() == 0.0


def test_backward_gemm_modes_agree(rt):
    """The split-fp16 backward products against the exact fp32 MFMA GEMM on the same dZ: normwise 1e-5."""
    n_ent, n_rel, B, rank = 9001, 7, 200, (5, 96, 96)
    core, R, S, O = gen.make_params(n_ent, n_rel, rank, 21)
    h, r = gen.make_queries(n_ent, n_rel, B, 21)
    w = torch.from_numpy(np.random.default_rng(2).standard_normal((B, n_ent)).astype(np.float32)).cuda()
    out = {}
    for mode in ("split_fp16", "f32"):
        rt.ops.BACKWARD_GEMM = mode
        try:
            leaves = [torch.from_numpy(x).cuda().requires_grad_(True) for x in (core, R, S, O)]
            (rt.score_1vN(*leaves, torch.from_numpy(h).cuda(), torch.from_numpy(r).cuda()) * w).sum().backward()
            out[mode] = [x.grad.clone() for x in leaves]
        finally:
            rt.ops.BACKWARD_GEMM = "split_fp16"
    for g0, g1 in zip(out["split_fp16"], out["f32"]):
        assert (g0 - g1).abs().max().item() <= 1e-5 * g1.abs().max().item()


# ------------------------------------------------------------------ BASELINE configs[2], [3] --------
def test_c3_full_size_sym_bf16(rt):
    """BASELINE configs[2] at FULL size: FB15k-237 shape, symmetric, rank (200,200,200), B = 2048, bf16 --
    the grouped contract kernel on bf16 MFMA tables (relation rank > 32) and the bf16 score kernel.
    Checked against the float64 oracle on bf16-rounded parameters, sampled rows in full."""
    n_ent, n_rel, B, rank = 14541, 474, 2048, (200, 200, 200)
    core, R, E, _ = gen.make_params(n_ent, n_rel, rank, 322, shared=True)
    h, r = gen.make_queries(n_ent, n_rel, B, 322)
    bc, bR, bE = [torch.from_numpy(x).cuda().to(torch.bfloat16) for x in (core, R, E)]
    dh, dr = dev(h, r)
    P = rt.score_1vN(bc, bR, bE, bE, dh, dr)
    Z = rt.score_1vN(bc, bR, bE, bE, dh, dr, sigmoid=False)
    assert P.shape == (B, n_ent) and torch.isfinite(P).all()
    rows = np.arange(0, B, 97)
    fc, fR, fE = [x.float().cpu() for x in (bc, bR, bE)]
    zref = orc.logits_exact(fc, fR, fE, fE, torch.from_numpy(h[rows]), torch.from_numpy(r[rows]))
    z = Z[torch.from_numpy(rows).cuda()].cpu().double().numpy()
    zref = np.asarray(zref, dtype=np.float64)
    # the bound of tests/test_gpu_bf16.py: the only rounding besides fp32 accumulation is v -> bf16,
    # |dz| <= 2^-8 sum_k |v_k| |o_k| element-wise; relative to (1 + |z|) capped at 5e-2
    ve = np.abs(orc.query_vectors_exact(fc, fR, fE, h[rows], r[rows]))
    bound = 2.0 ** -8 * (ve @ np.abs(fE.double().numpy()).T) + 1e-30
    assert np.max(np.abs(z - zref) / bound) <= 1.0
    assert np.max(np.abs(z - zref) / (1 + np.abs(zref))) <= 5e-2
    pref = 1.0 / (1.0 + np.exp(-zref))
    assert np.abs(P[torch.from_numpy(rows).cuda()].cpu().double().numpy() - pref).max() <= 2.5e-2
    # the cached-table path and the bf16-score output agree bit for bit with the default path
    tables = rt.relation_tables(bc, bR)
    assert torch.equal(rt.score_1vN(bc, bR, bE, bE, dh, dr, tables=tables), P)
    Pb = rt.score_1vN(bc, bR, bE, bE, dh, dr, out_dtype=torch.bfloat16)
    assert torch.equal(Pb, P.to(torch.bfloat16))


@pytest.mark.parametrize("B", [512, 2048])
def test_c4_fb15k_eight_way_shard_consistency(rt, B):
    """BASELINE configs[3]: FB15k shape (14 951 entities, 2 690 relations > B: planned batches), rank
    (200,200,200), the entity matrix cut into 8 row shards as 8 GPUs would hold them (last shard zero-
    padded): the concatenated shard blocks are BIT-identical to the unsharded score matrix, with stage 1
    replicated and with stage 1 split over the 'ranks' (query vectors of B/8 queries each, gathered,
    packed, scored).  fp32 parity with the oracle on sampled rows."""
    from r_tucker_amd.sharded import EntityShards
    n_ent, n_rel, rank = 14951, 2690, (200, 200, 200)
    core, R, S, O = dev(*gen.make_params(n_ent, n_rel, rank, 322))
    h, r = dev(*gen.make_queries(n_ent, n_rel, B, 322))
    full = rt.score_1vN(core, R, S, O, h, r)
    sh = EntityShards(n_ent, 8)
    # stage 1 split: 8 slices of the batch, concatenated = what the all-gather assembles
    B_loc = -(-B // 8)
    v_parts = [rt.query_vectors(core, R, S, h[p * B_loc:(p + 1) * B_loc], r[p * B_loc:(p + 1) * B_loc]) for p in range(8)]
    v_all = torch.cat(v_parts)[:B]
    assert torch.equal(v_all, rt.query_vectors(core, R, S, h, r))
    qp = rt.pack_query_vectors(v_all, torch.float32)
    blocks_rep, blocks_split = [], []
    for p in range(8):
        O_loc = sh.take(O, p)
        lo, hi = sh.bounds(p)
        out = rt.ops.alloc_scores(B, sh.n_loc, O.device)
        rt.score_1vN_into(core, R, S, O_loc, h, r, out)
        blocks_rep.append(out[:, : hi - lo].clone())
        out2 = rt.ops.alloc_scores(B, sh.n_loc, O.device)
        rt.score_packed_into(qp, B, O_loc, out2)
        blocks_split.append(out2[:, : hi - lo].clone())
        if hi - lo < sh.n_loc:                             # padding rows of the last shard score sigmoid(0)
            assert torch.all(out[:, hi - lo:] == 0.5)
    assert torch.equal(torch.cat(blocks_rep, dim=1), full)
    assert torch.equal(torch.cat(blocks_split, dim=1), full)
    rows = np.arange(0, B, 61)
    ref = orc.score_ref(*[x.cpu() for x in (core, R, S, O)], h.cpu()[rows], r.cpu()[rows])
    assert (full.cpu()[rows] - ref).abs().max().item() <= 3e-6


@pytest.mark.parametrize("B,bf16", [(300, False), (2500, True)])      # the per-query and the grouped contract kernel
def test_stage1_by_relation_parts_reassemble_the_full_vectors(rt, B, bf16):
    """rtk_query_vectors_from_tables_part_*: the rows of the queries with relation id = part (mod n_parts), nothing
    else; the n_parts launches together give the unsplit stage 1 bit for bit (what the all-reduce(SUM) of the entity-
    sharded scorer's stage1="relation" adds up)."""
    from r_tucker_amd import ops
    n_ent, n_rel, rank = 900, 37, (40, 64, 64)
    core, R, S, O = dev(*gen.make_params(n_ent, n_rel, rank, 9))
    if bf16:
        core, R, S = core.bfloat16(), R.bfloat16(), S.bfloat16()
    h, r = dev(*gen.make_queries(n_ent, n_rel, B, 9))
    tables = rt.relation_tables(core, R)
    full = ops.query_vectors(core, R, S, h, r, tables=tables)
    P = 8
    acc = torch.zeros_like(full)
    for p in range(P):
        part = torch.full_like(full, 7.0)
        ops.query_vectors_part(core, R, S, h, r, tables, p, P, part)
        sel = (r % P) == p
        assert torch.equal(part[sel], full[sel])
        assert torch.all(part[~sel] == 7.0)                    # untouched
        part[~sel] = 0
        acc += part
    assert torch.equal(acc, full)


def test_pack_query_vectors_matches_stage1_planes(rt):
    """rtk_pack_query_vectors writes the planes rtk_query_vectors_* writes (fp32 hi/lo and bf16)."""
    for dt, rank in ((torch.float32, (5, 72, 72)), (torch.bfloat16, (5, 72, 72)), (torch.float32, (4, 200, 200))):
        n_ent, n_rel, B = 800, 7, 77
        core, R, S, _ = [t.to(dt) for t in dev(*gen.make_params(n_ent, n_rel, rank, 6))]
        h, r = dev(*gen.make_queries(n_ent, n_rel, B, 6))
        v, qp = rt.query_vectors(core, R, S, h, r, packed=True)
        qp2 = rt.pack_query_vectors(v, dt)
        # rows of the last 32-query tile beyond B are never written by either producer: compare scores instead of raw bytes
        O = dev(gen.make_params(n_ent, n_rel, rank, 6)[3])[0].to(dt)
        o1 = rt.ops.alloc_scores(B, n_ent, O.device)
        o2 = rt.ops.alloc_scores(B, n_ent, O.device)
        rt.score_packed_into(qp, B, O, o1)
        rt.score_packed_into(qp2, B, O, o2)
        assert torch.equal(o1, o2)


def test_container_norm_in_the_reference_loss_fn(rt):
    """train.py:79: loss_fn = criterion(score_fn(T), targets) + coeff * T.norm() ** 2 built from this package's
    own containers, differentiated w.r.t. core and factors; norm() against the dense tensor."""
    n_ent, n_rel, B, rank = 60, 4, 9, (3, 5, 5)
    core, R, S, O = gen.make_params(n_ent, n_rel, rank, 1)
    dense = np.einsum("abc,ia,jb,kc->ijk", core.astype(np.float64), R.astype(np.float64), S.astype(np.float64), O.astype(np.float64))
    leaves = [torch.from_numpy(x).cuda().requires_grad_(True) for x in (core, R, S, O)]
    T = rt.Tucker(leaves[0], leaves[1:])
    assert abs(T.norm().item() - np.linalg.norm(dense)) <= 1e-4 * np.linalg.norm(dense)
    model = rt.AsymmetricR_TuckER((n_ent, n_rel), rank).cuda()
    h, r = dev(*gen.make_queries(n_ent, n_rel, B, 1))
    targets = (torch.rand(B, n_ent, device="cuda") < 0.1).float()
    criterion = torch.nn.BCELoss(reduction="mean")
    score_fn = model(h, r)
    loss_fn = lambda T: criterion(score_fn(T), targets) + 1e-3 * T.norm() ** 2      # noqa: E731
    loss_fn(T).backward()
    cl = [torch.from_numpy(x).clone().requires_grad_(True) for x in (core, R, S, O)]
    Pc = orc.score_ref(*cl, h.cpu(), r.cpu())
    dc = torch.einsum("abc,ia,jb,kc->ijk", *cl)
    (criterion(Pc, targets.cpu()) + 1e-3 * dc.norm() ** 2).backward()
    for got, want in zip(leaves, cl):
        assert (got.grad.cpu() - want.grad).abs().max().item() <= 1e-4 * want.grad.abs().max().item() + 1e-8
    # shared-factor container
    Ts = rt.SFTucker(leaves[0].detach(), [leaves[1].detach()], num_shared_factors=2, shared_factor=leaves[2].detach())
    dense_s = np.einsum("abc,ia,jb,kc->ijk", core.astype(np.float64), R.astype(np.float64), S.astype(np.float64), S.astype(np.float64))
    assert abs(Ts.norm().item() - np.linalg.norm(dense_s)) <= 1e-4 * np.linalg.norm(dense_s)


def test_abi_collective_one_rank_roundtrip():
    """rtk_comm_unique_id / rtk_comm_init / rtk_allgather_scores / rtk_comm_destroy through ctypes on one GPU (a
    one-rank RCCL communicator: the in-place all-gather leaves the buffer as it is), and the sharded scorer driven
    through it gives the single-device scores."""
    import ctypes as C
    import r_tucker_amd as rt
    from r_tucker_amd import _lib, synthetic as gen
    lib = _lib.load()
    ident = (C.c_ubyte * 128)()
    _lib.check(lib.rtk_comm_unique_id(ident), "rtk_comm_unique_id")
    assert any(ident)
    comm = C.c_void_p()
    _lib.check(lib.rtk_comm_init(0, 1, ident, C.byref(comm)), "rtk_comm_init")
    buf = torch.arange(4096, dtype=torch.float32, device="cuda")
    want = buf.clone()
    _lib.check(lib.rtk_allgather_scores(comm, buf.data_ptr(), buf.numel() * 4, torch.cuda.current_stream().cuda_stream),
               "rtk_allgather_scores")
    torch.cuda.synchronize()
    assert torch.equal(buf, want)
    _lib.check(lib.rtk_comm_destroy(comm), "rtk_comm_destroy")

    n_ent, n_rel, B, rank = 3000, 11, 40, (4, 32, 32)
    core, R, S, O = [torch.from_numpy(x).cuda() for x in gen.make_params(n_ent, n_rel, rank, 5)]
    h, r = [torch.from_numpy(x).cuda() for x in gen.make_queries(n_ent, n_rel, B, 5)]
    sc = rt.ShardedEntityScorer(n_ent, collective="abi")
    P = sc.score(core, R, S, sc.local_block(O), h, r)
    sc.close()
    assert torch.equal(P, rt.score_1vN(core, R, S, O, h, r))
