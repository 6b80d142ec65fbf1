"""Host side of the scoring path: tensor checks, workspace, stream, autograd.

``score_1vN`` is what ``score_fn(T)`` of both model flavours calls; it replaces the
five torch ops of ``src/model/asymmetric/R_TuckER.py:43-48`` with one call into the
C ABI (``rtk_score_1vN_f32`` / ``_bf16``).  Forward runs entirely in the hand-written
HIP kernels.  Backward (needed because the optimizer differentiates ``loss_fn(T)``,
``train.py:79-82``): the three B x N-sized steps -- logistic derivative, dO = dZ^T v,
dv = dZ O -- are HIP kernels (``rtk_sigmoid_grad_f32``, ``rtk_gemm_f32``,
``rtk_gemm_f32_splitk``); the small trilinear remainder (gradients of core, R, S from
dv) is device-side torch ops.  Nothing leaves the GPU.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _lib

_workspaces = {}

# Logistic used by the fused score epilogue: "exact" = expf + IEEE divide (the formula
# torch's CPU kernel evaluates), "fast" = 1 / (1 + 2^(-z log2 e)) on v_exp_f32 + v_rcp_f32
# (1 ulp each; ~6x fewer VALU instructions).  Override with R_TUCKER_AMD_SIGMOID.
DEFAULT_SIGMOID = os.environ.get("R_TUCKER_AMD_SIGMOID", "fast")

# Row pitch of a freshly allocated score matrix, in elements.  Rows that start on a 128-byte
# boundary are written in full lines (and, in the bf16 kernel, with nontemporal stores): measured
# 5-20 % of the score kernel (tools/ubench/vmem_rate.hip).  With N not a
# multiple of the pitch unit the result is the (B, N) view of a (B, pitch) buffer, i.e. NOT
# contiguous; R_TUCKER_AMD_ROW_ALIGN=1 restores the dense layout.  The autograd path (training)
# always uses the dense layout.
ROW_ALIGN = max(1, int(os.environ.get("R_TUCKER_AMD_ROW_ALIGN", "32")))
# The two B x N sized backward products (dO = dZ^T v, dv = dZ O): "split_fp16" = three f16 MFMAs per k-step on
# hi/lo halves like the forward (normwise fp32-class accuracy), "f32" = the exact fp32 MFMA GEMM (5x slower).
BACKWARD_GEMM = os.environ.get("R_TUCKER_AMD_BWD_GEMM", "split_fp16")
# Training loss: "1" = BCE terms and the logit gradient's base written by the score kernel's epilogue (the B x N matrix
# is written once in the forward and read only by the backward GEMMs); "0" = scores, then two more passes over them.
FUSED_BCE = os.environ.get("R_TUCKER_AMD_FUSED_BCE", "1") == "1"


def alloc_scores(B, N, device, lead=(), dtype=torch.float32):
    """(lead..., B, N) score buffer whose rows start on 128-byte boundaries (ROW_ALIGN float32
    elements; twice as many bf16 ones)."""
    unit = ROW_ALIGN * (4 // torch.empty((), dtype=dtype).element_size()) if ROW_ALIGN > 1 else 1
    pitch = -(-N // unit) * unit
    buf = torch.empty(tuple(lead) + (B, pitch), dtype=dtype, device=device)
    return buf[..., :N] if pitch != N else buf


def _require_gpu(name, t):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor, got {type(t)}")
    if not t.is_cuda:
        raise RuntimeError(
            f"{name} is on {t.device}: r_tucker_amd scores on MI355X only (no CPU path; "
            "use the reference implementation or oracle/ for CPU runs)")


def _operand(name, t, dtype):
    _require_gpu(name, t)
    if t.dtype != dtype:
        raise RuntimeError(f"{name} is {t.dtype} but the core is {dtype}: all operands must share one dtype "
                           "(float32, or bfloat16 for the bf16 path)")
    return t.contiguous()


def _f32c(name, t):
    return _operand(name, t, torch.float32)


def _idx(name, t, device):
    if isinstance(t, torch.Tensor) and t.dtype == torch.int64 and t.device == device and t.dim() == 1 and t.is_contiguous():
        return t
    if not isinstance(t, torch.Tensor):
        t = torch.as_tensor(t)
    if t.dtype not in (torch.int64, torch.int32, torch.int16, torch.uint8, torch.int8):
        raise IndexError(f"{name} must be an integer tensor, got {t.dtype}")  # torch: "tensors used as indices must be long..."
    return t.to(device=device, dtype=torch.int64).contiguous().view(-1)


_sizes = {}      # memoised size queries of the C ABI (one ctypes call each otherwise, per scoring call)
_packed = {}     # (device index, stream) -> packed-query-plane buffer of the no-autograd path


def _size(fn_name, *args):
    key = (fn_name,) + args
    v = _sizes.get(key)
    if v is None:
        v = _sizes[key] = getattr(_lib.load(), fn_name)(*args)
    return v


def _packed_buffer(device, stream_ptr, nbytes):
    """Packed query planes of a call that does not hand them out: written by stage 1 and read by the score kernel
    of the same call on the same stream, so one buffer per (device, stream) serves every call (no allocation)."""
    key = (device.index, stream_ptr)
    b = _packed.get(key)
    if b is None or b.numel() < nbytes:
        b = _packed[key] = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
    return b


def _workspace(device, stream_ptr, nbytes):
    """The caller-owned scratch of the C ABI, one per (device, stream).  Its 256-byte header holds the
    sticky device error word (include/rtucker_hip.h); when the buffer has to grow the header is carried
    over, so an out-of-range id seen before the regrow is still reported.  Called with the owning
    stream current, so the header copy is ordered behind the kernels that may have set the word."""
    key = (device.index, stream_ptr)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        new = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        if ws is None:
            new[:256].zero_()
        else:
            new[:256].copy_(ws[:256])
        _workspaces[key] = ws = new
    return ws


# What happens when a kernel meets an out-of-range subject / relation id (the kernels clamp it and raise
# a sticky flag; the reference raises IndexError, SURVEY.md section 8b "Errors"):
#   "strict"   (default) every scoring call reads the flag back before returning -> IndexError like the
#              reference, at the price of one stream synchronisation per call;
#   "deferred" nothing per call; ``check_device_errors()`` raises at the caller's own sync point
#              (``evaluate()`` and the training driver do that once per loop);
#   "off"      never checked (graph capture, benchmarks).
# Calls made while the stream is being captured into a HIP graph are never synchronised.
INDEX_CHECK = os.environ.get("R_TUCKER_AMD_INDEX_CHECK", "strict")


class index_check:
    """Context manager / setter for the out-of-range-id policy: ``with index_check("deferred"): ...``."""

    def __init__(self, mode):
        if mode not in ("strict", "deferred", "off"):
            raise ValueError(f"index check mode must be strict | deferred | off, got {mode!r}")
        self.mode = mode

    def __enter__(self):
        global INDEX_CHECK
        self.prev, INDEX_CHECK = INDEX_CHECK, self.mode
        return self

    def __exit__(self, *exc):
        global INDEX_CHECK
        INDEX_CHECK = self.prev
        return False


def _check_now(ws, sp):
    """Read (and clear) the error word of one workspace on its own stream; raise like torch's indexing."""
    flag = C.c_uint32(0)
    _lib.check(_lib.load().rtk_read_error_flag(ws.data_ptr(), sp, C.byref(flag)), "rtk_read_error_flag")
    if flag.value & 1:
        raise IndexError("index out of range in self (subject_idx / relation_idx)")


def _strict_check(dev, ws, sp):
    if INDEX_CHECK == "strict" and not torch.cuda.is_current_stream_capturing():
        _check_now(ws, sp)


def _stream_ptr(device):
    return torch.cuda.current_stream(device).cuda_stream


def relation_tables(core, R):
    """``tables[u] = G x_0 R[u]`` for ALL relations -> ``(n_rel, b, c)`` fp32: the part of stage 1 that only
    depends on the parameters (einsum of asymmetric/R_TuckER.py:45 applied to every relation row).  Pass the
    result as ``tables=`` to ``score_1vN`` / ``query_vectors`` while the parameters stay unchanged."""
    lib = _lib.load()
    _require_gpu("core", core)
    if core.dtype not in (torch.float32, torch.bfloat16):
        raise RuntimeError(f"core must be float32 or bfloat16, got {core.dtype}")
    bf16 = core.dtype == torch.bfloat16
    core, R = _operand("core", core.detach(), core.dtype), _operand("R", R.detach(), core.dtype)
    if core.dim() != 3 or R.dim() != 2 or R.shape[1] != core.shape[0]:
        raise RuntimeError("expected core (a,b,c), R (nR,a)")
    a, b, c = core.shape
    if b != c:
        raise RuntimeError(f"subject rank {b} must equal object rank {c} (asymmetric/R_TuckER.py:46)")
    dev = core.device
    n_rel = R.shape[0]
    tables = torch.empty((n_rel, b, c), dtype=torch.float32, device=dev)
    dcode = _lib.RTK_BF16 if bf16 else _lib.RTK_F32
    with torch.cuda.device(dev):
        sp = _stream_ptr(dev)
        scratch = torch.empty(lib.rtk_relation_tables_workspace_bytes(dcode, n_rel, a, b, c), dtype=torch.uint8, device=dev)
        fn = lib.rtk_relation_tables_bf16 if bf16 else lib.rtk_relation_tables_f32
        _lib.check(fn(core.data_ptr(), a, b, c, R.data_ptr(), n_rel, tables.data_ptr(), scratch.data_ptr(), scratch.numel(), sp),
                   "rtk_relation_tables")
    return tables


def _forward(core, R, S, O, subject_idx, relation_idx, sigmoid, exact, want_v, sigmoid_mode=None, out=None,
             out_dtype=torch.float32, padded=None, tables=None):
    lib = _lib.load()
    _require_gpu("core", core)
    if core.dtype not in (torch.float32, torch.bfloat16):
        raise RuntimeError(f"core must be float32 or bfloat16, got {core.dtype}")
    bf16 = core.dtype == torch.bfloat16
    dt = core.dtype
    core, R, S, O = _operand("core", core, dt), _operand("R", R, dt), _operand("S", S, dt), _operand("O", O, dt)
    dev = core.device
    for n, t in (("R", R), ("S", S), ("O", O)):
        if t.device != dev:
            raise RuntimeError(f"{n} is on {t.device}, core on {dev}")
    if core.dim() != 3 or R.dim() != 2 or S.dim() != 2 or O.dim() != 2:
        raise RuntimeError("expected core (a,b,c), R (nR,a), S (N,b), O (N,c)")
    a, b, c = core.shape
    if R.shape[1] != a or S.shape[1] != b or O.shape[1] != c:
        raise RuntimeError(f"factor widths {R.shape[1]},{S.shape[1]},{O.shape[1]} do not match core {tuple(core.shape)}")
    h = _idx("subject_idx", subject_idx, dev)
    r = _idx("relation_idx", relation_idx, dev)
    if h.numel() != r.numel():
        raise RuntimeError(f"subject_idx has {h.numel()} entries, relation_idx {r.numel()}")
    B, N = h.numel(), O.shape[0]
    if b != c:
        # asymmetric/R_TuckER.py:46: .view(-1, b) of a (B,1,c) tensor
        raise RuntimeError(f"shape '[-1, {b}]' is invalid for input of size {B * c}")
    if out_dtype not in (torch.float32, torch.bfloat16):
        raise RuntimeError(f"out_dtype must be float32 or bfloat16, got {out_dtype}")
    if out_dtype == torch.bfloat16 and (not bf16 or want_v or not sigmoid or exact):
        raise RuntimeError("bfloat16 scores: bf16 operands, sigmoid=True, no autograd (the reference's bf16 eval path)")
    if out is None:
        # dense when the scores are handed to autograd's caller, 128-byte aligned rows otherwise
        dense = want_v if padded is None else not padded
        out = torch.empty((B, N), dtype=torch.float32, device=dev) if dense else alloc_scores(B, N, dev, dtype=out_dtype)
    elif (tuple(out.shape) != (B, N) or out.dtype != out_dtype or out.device != dev or out.stride(1) != 1
          or out.stride(0) < N):
        raise RuntimeError(f"out must be a {out_dtype} ({B}, {N}) tensor on {dev} with unit column stride")
    ld = out.stride(0) if B > 1 else N
    if B == 0:
        return out, None
    with torch.cuda.device(dev):
        sp = _stream_ptr(dev)
        dcode = _lib.RTK_BF16 if bf16 else _lib.RTK_F32
        if tables is not None:
            if (tables.dtype != torch.float32 or tables.device != dev or tuple(tables.shape) != (R.shape[0], b, c)
                    or not tables.is_contiguous()):
                raise RuntimeError(f"tables must be a contiguous float32 ({R.shape[0]}, {b}, {c}) tensor on {dev} "
                                   "(ops.relation_tables)")
            need = _size("rtk_from_tables_workspace_bytes", B, R.shape[0])
        else:
            need = _size("rtk_workspace_bytes", dcode, B, R.shape[0], a, b, c)
        ws = _workspace(dev, sp, need)
        if bf16 and (exact or c > 512):
            raise RuntimeError("bf16 operands: only the bf16 MFMA score kernel exists (c <= 512, exact=False)")
        qv = lib.rtk_query_vectors_bf16 if bf16 else lib.rtk_query_vectors_f32
        sp_fn = lib.rtk_score_packed_bf16 if bf16 else lib.rtk_score_packed_f32
        s1vn = lib.rtk_score_1vN_bf16 if bf16 else lib.rtk_score_1vN_f32
        mode = sigmoid_mode or DEFAULT_SIGMOID
        if mode not in ("fast", "exact"):
            raise ValueError(f"sigmoid mode must be 'fast' or 'exact', got {mode!r}")
        flags = (_lib.RTK_SCORE_SIGMOID if sigmoid else 0) | (_lib.RTK_SCORE_EXACT_F32 if exact else 0)
        flags |= _lib.RTK_SCORE_SIGMOID_FAST if (sigmoid and mode == "fast") else 0
        if out_dtype == torch.bfloat16:
            if mode != "fast":
                raise RuntimeError("bfloat16 scores use the fast logistic (sigmoid_mode='fast')")
            flags |= _lib.RTK_SCORE_OUT_BF16
        sflags = flags & (_lib.RTK_SCORE_SIGMOID | _lib.RTK_SCORE_SIGMOID_FAST | _lib.RTK_SCORE_OUT_BF16)
        v = None
        if tables is not None:
            # stage 1 against the prebuilt relation tables, then the score kernel on the packed planes
            if exact or (not bf16 and c > 512):
                v = torch.empty((B, c), dtype=torch.float32, device=dev)
                qp = None
            else:
                v = torch.empty((B, c), dtype=torch.float32, device=dev) if want_v else None
                qp = _packed_buffer(dev, sp, _size("rtk_packed_query_bytes", dcode, B, c))
            ft = lib.rtk_query_vectors_from_tables_bf16 if bf16 else lib.rtk_query_vectors_from_tables_f32
            _lib.check(ft(tables.data_ptr(), R.shape[0], b, c, S.data_ptr(), S.shape[0], r.data_ptr(), h.data_ptr(), B,
                          v.data_ptr() if v is not None else None, qp.data_ptr() if qp is not None else None,
                          ws.data_ptr(), ws.numel(), sp), "rtk_query_vectors_from_tables")
            if qp is not None:
                _lib.check(sp_fn(qp.data_ptr(), B, c, O.data_ptr(), N, out.data_ptr(), ld, sflags, sp), "rtk_score_packed")
            else:
                _lib.check(lib.rtk_score_f32(v.data_ptr(), B, c, O.data_ptr(), N, out.data_ptr(), ld,
                                             flags & _lib.RTK_SCORE_SIGMOID, sp), "rtk_score_f32")
        elif want_v:
            # two-call form so the fp32 query vectors are kept for backward
            v = torch.empty((B, c), dtype=torch.float32, device=dev)
            use_packed = bf16 or (not exact and c <= 512)
            qp = None
            if use_packed:
                qp = torch.empty(lib.rtk_packed_query_bytes(dcode, B, c), dtype=torch.uint8, device=dev)
            _lib.check(qv(core.data_ptr(), a, b, c, R.data_ptr(), R.shape[0],
                                                 S.data_ptr(), S.shape[0], r.data_ptr(), h.data_ptr(), B,
                                                 v.data_ptr(), qp.data_ptr() if use_packed else None,
                                                 ws.data_ptr(), ws.numel(), sp), "rtk_query_vectors")
            if use_packed:
                _lib.check(sp_fn(qp.data_ptr(), B, c, O.data_ptr(), N, out.data_ptr(), ld,
                                 sflags & ~_lib.RTK_SCORE_OUT_BF16, sp), "rtk_score_packed_f32")
            else:
                _lib.check(lib.rtk_score_f32(v.data_ptr(), B, c, O.data_ptr(), N, out.data_ptr(), ld,
                                             flags & _lib.RTK_SCORE_SIGMOID, sp), "rtk_score_f32")
        else:
            _lib.check(s1vn(core.data_ptr(), a, b, c, R.data_ptr(), R.shape[0],
                                             S.data_ptr(), S.shape[0], O.data_ptr(), N,
                                             r.data_ptr(), h.data_ptr(), B, out.data_ptr(), ld, flags,
                                             ws.data_ptr(), ws.numel(), sp), "rtk_score_1vN_f32")
        _strict_check(dev, ws, sp)
    return out, v


class _Score1vN(torch.autograd.Function):
    @staticmethod
    def forward(ctx, core, R, S, O, subject_idx, relation_idx, sigmoid, exact, sigmoid_mode):
        out, v = _forward(core, R, S, O, subject_idx, relation_idx, sigmoid, exact, want_v=True, sigmoid_mode=sigmoid_mode)
        dev = core.device
        ctx.save_for_backward(core, R, S, O, _idx("s", subject_idx, dev), _idx("r", relation_idx, dev), v, out)
        ctx.sigmoid = sigmoid
        return out

    @staticmethod
    def backward(ctx, grad_out):
        core, R, S, O, h, r, v, out = ctx.saved_tensors
        pdt = core.dtype                      # bf16 operands: gradients computed in fp32, returned in bf16
        if pdt != torch.float32:
            core, R, S, O = core.float(), R.float(), S.float(), O.float()
        lib = _lib.load()
        dev = out.device
        B, N = out.shape
        c = O.shape[1]
        grad_out = grad_out.contiguous().float()
        with torch.cuda.device(dev):
            sp = _stream_ptr(dev)
            if ctx.sigmoid and B > 65535:   # (the row-pitch kernel's grid limit) dense dZ
                dZ = torch.empty_like(out)
                _lib.check(lib.rtk_sigmoid_grad_f32(grad_out.data_ptr(), out.data_ptr(), dZ.data_ptr(), out.numel(), sp),
                           "rtk_sigmoid_grad_f32")
            elif ctx.sigmoid:    # dZ = dP * P * (1 - P)   (HIP), written on 128-byte aligned rows for the GEMMs
                dZ = alloc_scores(B, N, dev)
                _lib.check(lib.rtk_sigmoid_grad_rows_f32(grad_out.data_ptr(), N, out.data_ptr(), out.stride(0) if B > 1 else N,
                                                         dZ.data_ptr(), dZ.stride(0) if B > 1 else N, B, N, sp),
                           "rtk_sigmoid_grad_rows_f32")
            else:
                dZ = grad_out
        return _grads_from_dZ(core, R, S, O, h, r, v, dZ, ctx.needs_input_grad, pdt) + (None,) * 5


def _splits_for(M, N, K):
    """Split-K factor of the dv product: enough 128x128 tiles x chunks to fill 256 CUs twice, chunks >= 512 deep."""
    tiles = -(-M // 128) * -(-N // 128)
    return int(max(1, min(64, 512 // max(tiles, 1), K // 512)))


def _grads_from_dZ(core, R, S, O, h, r, v, dZ, needs, pdt, dz_bound=None, scale=None):
    """(g_core, g_R, g_S, g_O) from dZ = d loss / d logits (B, N) fp32 and the saved fp32 query vectors,
    all in HIP kernels with a fixed summation order (bit-identical from run to run): dO = dZ^T v and
    dv = dZ O (split-K slabs added in chunk order) on the split-fp16 MFMA path of the forward
    (``BACKWARD_GEMM = "f32"`` / ``R_TUCKER_AMD_BWD_GEMM=f32``: the exact fp32 MFMA GEMM), then the stage-1
    backward ``rtk_query_vectors_bwd_f32`` (two more GEMMs and a deterministic row scatter).  ``dz_bound``: a
    one-element device tensor >= max|dZ| when the caller knows one (the BCE gradient does); otherwise one
    max-reduction over dZ finds it.  ``scale``: a one-element device tensor s -- the gradients of ``s * dZ`` are
    returned without a pass over dZ: everything is linear in dZ, so s multiplies the small operands
    (``v`` for dO, ``dv`` before the stage-1 backward)."""
    lib = _lib.load()
    dev = dZ.device
    B, N = dZ.shape
    ldz = dZ.stride(0) if B > 1 else N      # dZ may be the (B, N) view of a padded buffer (bce_loss_1vN)
    a, b, c = core.shape
    core, R, S, Of = core.contiguous(), R.contiguous(), S.contiguous(), O.contiguous()
    gO = gcore = gR = gS = None
    sf16 = BACKWARD_GEMM != "f32"
    with torch.cuda.device(dev):
        sp = _stream_ptr(dev)
        if sf16:
            bounds = torch.empty(3, dtype=torch.float32, device=dev)      # max|dZ|, max|v|, max|O|

            def absmax(x, ld, slot):
                _lib.check(lib.rtk_absmax_f32(x.data_ptr(), x.shape[0], x.shape[1], ld, bounds[slot:].data_ptr(), sp),
                           "rtk_absmax_f32")
                return bounds[slot:slot + 1]
            if dz_bound is None:
                dz_bound = absmax(dZ, ldz, 0)
            dz_bound = dz_bound.to(device=dev, dtype=torch.float32).reshape(1).contiguous()
        v_o = v if scale is None else v * scale.to(device=dev, dtype=torch.float32).reshape(1)      # (B, c): dO's small operand
        if sf16:
            v_bound = absmax(v_o, c, 1)
        if needs[3]:
            # gO[j, k] = sum_d dZ[d, j] * v[d, k]   -- fp32 MFMA GEMM, both operands M-major
            gO = torch.empty((N, c), dtype=torch.float32, device=dev)
            if sf16:
                _lib.check(lib.rtk_gemm_sf16_splitk(dZ.data_ptr(), 0, ldz, dz_bound.data_ptr(), v_o.data_ptr(), 0, c,
                                                    v_bound.data_ptr(), gO.data_ptr(), c, N, c, B, 1, None, 0, sp),
                           "rtk_gemm_sf16_splitk (dO)")
            else:
                _lib.check(lib.rtk_gemm_f32(dZ.data_ptr(), 0, ldz, v_o.data_ptr(), 0, c, gO.data_ptr(), c, N, c, B, 0, sp),
                           "rtk_gemm_f32 (dO)")
        if needs[0] or needs[1] or needs[2]:
            # dv[d, k] = sum_j dZ[d, j] * O[j, k]   -- K = N entities: split-K, slabs reduced in chunk order
            dv = torch.empty((B, c), dtype=torch.float32, device=dev)
            splits = _splits_for(B, c, N)
            skw = torch.empty(max(256, lib.rtk_gemm_f32_splitk_workspace_bytes(B, c, splits)), dtype=torch.uint8, device=dev)
            if sf16:
                o_bound = absmax(Of, c, 2)
                _lib.check(lib.rtk_gemm_sf16_splitk(dZ.data_ptr(), 1, ldz, dz_bound.data_ptr(), Of.data_ptr(), 0, c,
                                                    o_bound.data_ptr(), dv.data_ptr(), c, B, c, N, splits, skw.data_ptr(),
                                                    skw.numel(), sp), "rtk_gemm_sf16_splitk (dv)")
            else:
                _lib.check(lib.rtk_gemm_f32_splitk(dZ.data_ptr(), 1, ldz, Of.data_ptr(), 0, c, dv.data_ptr(), c, B, c, N,
                                                   splits, skw.data_ptr(), skw.numel(), sp), "rtk_gemm_f32_splitk (dv)")
            if scale is not None:
                dv = dv * scale.to(device=dev, dtype=torch.float32).reshape(1)
            gcore = torch.empty_like(core) if needs[0] else None
            gR = torch.empty_like(R) if needs[1] else None
            gS = torch.empty_like(S) if needs[2] else None
            bws = torch.empty(lib.rtk_query_bwd_workspace_bytes(B, a, b, c), dtype=torch.uint8, device=dev)
            _lib.check(lib.rtk_query_vectors_bwd_f32(core.data_ptr(), a, b, c, R.data_ptr(), R.shape[0], S.data_ptr(),
                                                     S.shape[0], r.data_ptr(), h.data_ptr(), B, dv.data_ptr(),
                                                     gcore.data_ptr() if needs[0] else None,
                                                     gR.data_ptr() if needs[1] else None,
                                                     gS.data_ptr() if needs[2] else None,
                                                     bws.data_ptr(), bws.numel(), sp), "rtk_query_vectors_bwd_f32")
    if pdt != torch.float32:
        gcore, gR, gS, gO = [g.to(pdt) if g is not None else None for g in (gcore, gR, gS, gO)]
    # symmetric model: S and O are the same tensor passed twice; autograd sums gS + gO
    return gcore, gR, gS, gO


class _GramTN(torch.autograd.Function):
    """``A^T B`` for tall-skinny fp32 GPU operands ``A (n, p)``, ``B (n, q)``, ``n >> p, q`` -- the Gram-type
    products of the Riemannian layer (factor Gram matrices in ``T.norm()``, ``U^T M`` in the tangent-space
    projection; n = 40 943, p = q = 400 at the WN18RR training shape).  rocBLAS runs this shape on
    ceil(p/256) * ceil(q/256) = 4 workgroups without splitting K (9 ms per product, 70 % of a training step,
    profiles/r02_train_kernel_stats.csv); here it is the split-K fp32-MFMA GEMM of the C ABI, K cut over
    the chip and the slabs added in a fixed order (deterministic)."""

    @staticmethod
    def forward(ctx, A, B):
        lib = _lib.load()
        A, B = A.contiguous(), B.contiguous()
        n, p = A.shape
        q = B.shape[1]
        dev = A.device
        C_ = torch.empty((p, q), dtype=torch.float32, device=dev)
        splits = _splits_for(p, q, n)
        with torch.cuda.device(dev):
            skw = torch.empty(max(256, lib.rtk_gemm_f32_splitk_workspace_bytes(p, q, splits)), dtype=torch.uint8, device=dev)
            _lib.check(lib.rtk_gemm_f32_splitk(A.data_ptr(), 0, p, B.data_ptr(), 0, q, C_.data_ptr(), q, p, q, n, splits,
                                               skw.data_ptr(), skw.numel(), _stream_ptr(dev)), "rtk_gemm_f32_splitk (A^T B)")
        ctx.save_for_backward(A, B)
        return C_

    @staticmethod
    def backward(ctx, dC):
        A, B = ctx.saved_tensors
        dA = B @ dC.transpose(0, 1) if ctx.needs_input_grad[0] else None      # (n, q) @ (q, p): n row blocks, fills the chip
        dB = A @ dC if ctx.needs_input_grad[1] else None
        return dA, dB


def gram_tn(A, B):
    """``A.T @ B``; tall-skinny fp32 GPU operands go through the split-K HIP GEMM, anything else through torch
    (the Riemannian layer is generic torch code: float64 CPU tensors in its tests)."""
    if (A.is_cuda and A.dtype == torch.float32 and B.dtype == torch.float32 and A.dim() == 2 and B.dim() == 2
            and A.shape[0] == B.shape[0] and A.shape[0] >= 8192 and A.shape[0] >= 8 * max(A.shape[1], B.shape[1])):
        return _GramTN.apply(A, B)
    return A.transpose(0, 1) @ B


class _BceLoss1vN(torch.autograd.Function):
    """mean BCE(sigmoid(logits), smoothed multi-hot targets) with the targets given as a CSR."""

    @staticmethod
    def forward(ctx, core, R, S, O, subject_idx, relation_idx, pair_slot, pair_ptr, pair_obj, label_smoothing):
        ctx.fused = False
        if (FUSED_BCE and core.dtype == torch.float32 and core.is_cuda and core.shape[2] <= 512 and core.shape[1] == core.shape[2]
                and subject_idx.numel() > 0):
            return _BceLoss1vN._forward_fused(ctx, core, R, S, O, subject_idx, relation_idx, pair_slot, pair_ptr, pair_obj,
                                              float(label_smoothing))
        # the scores never leave this function pair: aligned rows for them too
        P, v = _forward(core, R, S, O, subject_idx, relation_idx, True, False, want_v=True, padded=True)
        lib = _lib.load()
        dev = P.device
        B, N = P.shape
        rows = torch.empty(B, dtype=torch.float64, device=dev)
        with torch.cuda.device(dev):
            _lib.check(lib.rtk_bce_rows_f32(P.data_ptr(), B, N, P.stride(0), pair_slot.data_ptr(), pair_ptr.data_ptr(),
                                            pair_obj.data_ptr(), float(label_smoothing), rows.data_ptr(), _stream_ptr(dev)),
                       "rtk_bce_rows_f32")
        ctx.save_for_backward(core, R, S, O, _idx("s", subject_idx, dev), _idx("r", relation_idx, dev), v, P,
                              pair_slot, pair_ptr, pair_obj)
        ctx.eps = float(label_smoothing)
        return (rows.sum() / (B * N)).to(torch.float32)

    @staticmethod
    def _forward_fused(ctx, core, R, S, O, subject_idx, relation_idx, pair_slot, pair_ptr, pair_obj, eps):
        """Loss fused into the score kernel's epilogue (``rtk_score_packed_bce_f32``): the B x N matrix is written once,
        as ``x = p - eps / N`` (the logit gradient of a negative, up to g / (B N)), never re-read in the forward; the few
        positives are patched by ``rtk_bce_patch_pos_f32``."""
        lib = _lib.load()
        dev = core.device
        core, R, S, O = _f32c("core", core), _f32c("R", R), _f32c("S", S), _f32c("O", O)
        h, r = _idx("subject_idx", subject_idx, dev), _idx("relation_idx", relation_idx, dev)
        B, N, c = h.numel(), O.shape[0], core.shape[2]
        v, qp = query_vectors(core.detach(), R.detach(), S.detach(), h, r, packed=True)
        X = alloc_scores(B, N, dev)
        partials = torch.empty(lib.rtk_score_bce_partials(), dtype=torch.float64, device=dev)
        rows_pos = torch.empty(4 * B, dtype=torch.float64, device=dev)      # four partial sums per row
        ld = X.stride(0) if B > 1 else N
        with torch.cuda.device(dev):
            sp = _stream_ptr(dev)
            _lib.check(lib.rtk_score_packed_bce_f32(qp.data_ptr(), B, c, O.data_ptr(), N, X.data_ptr(), ld, eps,
                                                    partials.data_ptr(), sp), "rtk_score_packed_bce_f32")
            _lib.check(lib.rtk_bce_patch_pos_f32(X.data_ptr(), B, N, ld, pair_slot.data_ptr(), pair_ptr.data_ptr(),
                                                 pair_obj.data_ptr(), eps, v.data_ptr(), O.data_ptr(), c, rows_pos.data_ptr(), sp),
                       "rtk_bce_patch_pos_f32")
        ctx.save_for_backward(core, R, S, O, h, r, v, X, pair_slot, pair_ptr, pair_obj)
        ctx.eps = eps
        ctx.fused = True
        return ((partials.sum() + rows_pos.sum()) / (B * N)).to(torch.float32)

    @staticmethod
    def backward(ctx, grad_loss):
        core, R, S, O, h, r, v, P, pair_slot, pair_ptr, pair_obj = ctx.saved_tensors
        if ctx.fused:
            B, N = P.shape
            g = grad_loss.to(device=P.device, dtype=torch.float32).reshape(1) * (1.0 / (B * N))
            # X = (p - y) unscaled, |X| <= 1: the GEMMs' operand bound; g / (B N) rides on the small operands
            return _grads_from_dZ(core, R, S, O, h, r, v, P, ctx.needs_input_grad, core.dtype,
                                  dz_bound=torch.ones(1, dtype=torch.float32, device=P.device), scale=g) + (None,) * 6
        if getattr(ctx, "spent", False):
            raise RuntimeError("bce_loss_1vN: backward called twice (the saved scores are overwritten by the first pass)")
        ctx.spent = True
        pdt = core.dtype
        if pdt != torch.float32:
            core, R, S, O = core.float(), R.float(), S.float(), O.float()
        lib = _lib.load()
        dev = P.device
        B, N = P.shape
        g = grad_loss.to(device=dev, dtype=torch.float32).reshape(1).contiguous()
        with torch.cuda.device(dev):
            # in place: P <- (P - y) * g / (B * N) = d loss / d logits  (the saved scores are spent)
            _lib.check(lib.rtk_bce_grad_f32(P.data_ptr(), B, N, P.stride(0), pair_slot.data_ptr(), pair_ptr.data_ptr(),
                                            pair_obj.data_ptr(), ctx.eps, g.data_ptr(), 1.0 / (B * N), _stream_ptr(dev)),
                       "rtk_bce_grad_f32")
        # |dZ| = |P - y| |g| / (B N) <= |g| / (B N): the operand bound of the split-fp16 GEMMs, no pass over dZ
        return _grads_from_dZ(core, R, S, O, h, r, v, P, ctx.needs_input_grad, pdt,
                              dz_bound=g.abs() * (1.0 / (B * N))) + (None,) * 6


def bce_loss_1vN(core, R, S, O, subject_idx, relation_idx, flt, item_ids, label_smoothing=0.0):
    """The reference's training loss term ``nn.BCELoss()(score_fn(T), targets)`` (train.py:79,136) for a
    batch of (subject, relation) items WITHOUT the dense target matrix: ``flt`` is the
    ``evaluation.DeviceFilter`` of the train-mode ``KG_dataset`` (CSR of known objects per pair, on
    the device), ``item_ids`` the dataset indices of the batch; the targets
    ``(1 - eps) * multi_hot + eps / N`` (Dataset.py:51-52) are applied inside the kernels.
    Differentiable w.r.t. core and factors like ``score_1vN``."""
    dev = core.device
    slot = flt.slot_of_item[item_ids.to(dev)].contiguous()
    return _BceLoss1vN.apply(core, R, S, O, subject_idx, relation_idx, slot, flt.pair_ptr, flt.pair_obj,
                             float(label_smoothing))


def score_1vN(core, R, S, O, subject_idx, relation_idx, sigmoid=True, exact=False, sigmoid_mode=None,
              out_dtype=torch.float32, tables=None):
    """``sigmoid((G x_0 R[r] x_1 S[h]) . O^T)`` for a batch of (h, r) queries -> ``(B, N)``.

    Same operands and result as the body of the reference's ``score_fn``
    (asymmetric/R_TuckER.py:43-48; symmetric: pass ``S is O``).  ``exact=True``
    selects the exact-fp32 MFMA score kernel instead of the split-fp16 one;
    ``sigmoid_mode`` ("fast" | "exact") picks the logistic of the fused epilogue.
    ``out_dtype=torch.bfloat16`` (bf16 operands, no autograd): the scores are rounded to bf16 in the
    kernel -- the dtype the reference's bf16 model returns -- which halves the dominant HBM traffic;
    bit-identical to ``score_1vN(...).to(torch.bfloat16)``.
    ``tables`` (no autograd): the prebuilt relation tables of these parameters (``relation_tables(core, R)``);
    stage 1 then only does the subject-mode contraction -- same bits as without them.
    """
    needs_grad = torch.is_grad_enabled() and any(
        isinstance(t, torch.Tensor) and t.requires_grad for t in (core, R, S, O))
    if needs_grad:
        if out_dtype != torch.float32:
            raise RuntimeError("bfloat16 scores are an inference option (no autograd)")
        return _Score1vN.apply(core, R, S, O, subject_idx, relation_idx, sigmoid, exact, sigmoid_mode)
    out, _ = _forward(core, R, S, O, subject_idx, relation_idx, sigmoid, exact, want_v=False, sigmoid_mode=sigmoid_mode,
                      out_dtype=out_dtype, tables=tables)
    return out


def score_1vN_into(core, R, S, O, subject_idx, relation_idx, out, sigmoid=True, exact=False, sigmoid_mode=None,
                   tables=None):
    """``score_1vN`` writing into a caller-provided (B, N) buffer (row stride >= N; float32, or bfloat16
    for bf16 operands with the fast logistic); no autograd.  Used by the entity-sharded scorer so the
    local block lands in its all-gather slot."""
    with torch.no_grad():
        _forward(core, R, S, O, subject_idx, relation_idx, sigmoid, exact, want_v=False,
                 sigmoid_mode=sigmoid_mode, out=out, out_dtype=out.dtype, tables=tables)
    return out


def query_vectors_part(core, R, S, subject_idx, relation_idx, tables, part, n_parts, out):
    """Stage 1 for the queries whose relation id is congruent to ``part`` modulo ``n_parts`` only, against the prebuilt
    relation ``tables``: their rows of ``out`` (B, c) fp32 are written, the others left untouched
    (``rtk_query_vectors_from_tables_part_*``; the entity-sharded scorer's ``stage1="relation"``)."""
    lib = _lib.load()
    _require_gpu("core", core)
    bf16 = core.dtype == torch.bfloat16
    dt = core.dtype
    core, R, S = _operand("core", core, dt), _operand("R", R, dt), _operand("S", S, dt)
    dev = core.device
    a, b, c = core.shape
    h, r = _idx("subject_idx", subject_idx, dev), _idx("relation_idx", relation_idx, dev)
    B = h.numel()
    if tuple(out.shape) != (B, c) or out.dtype != torch.float32 or not out.is_contiguous() or out.device != dev:
        raise RuntimeError(f"out must be a contiguous float32 ({B}, {c}) tensor on {dev}")
    if tables is None or tuple(tables.shape) != (R.shape[0], b, c) or tables.dtype != torch.float32 or not tables.is_contiguous():
        raise RuntimeError("query_vectors_part needs the prebuilt relation tables (ops.relation_tables)")
    if B == 0:
        return out
    with torch.cuda.device(dev):
        sp = _stream_ptr(dev)
        ws = _workspace(dev, sp, _size("rtk_from_tables_workspace_bytes", B, R.shape[0]))
        fn = lib.rtk_query_vectors_from_tables_part_bf16 if bf16 else lib.rtk_query_vectors_from_tables_part_f32
        _lib.check(fn(tables.data_ptr(), R.shape[0], b, c, S.data_ptr(), S.shape[0], r.data_ptr(), h.data_ptr(), B,
                      int(part), int(n_parts), out.data_ptr(), ws.data_ptr(), ws.numel(), sp),
                   "rtk_query_vectors_from_tables_part")
        _strict_check(dev, ws, sp)
    return out


def query_vectors(core, R, S, subject_idx, relation_idx, tables=None, packed=False):
    """Stage 1 only: ``v[d] = S[h_d] . (G x_0 R[r_d])`` -> ``(B, c)`` fp32 (R_TuckER.py:43-46).
    ``packed=True`` returns ``(v, q_packed)`` with the packed query planes the score kernels consume."""
    lib = _lib.load()
    _require_gpu("core", core)
    bf16 = core.dtype == torch.bfloat16
    dt = core.dtype
    core, R, S = _operand("core", core, dt), _operand("R", R, dt), _operand("S", S, dt)
    dev = core.device
    a, b, c = core.shape
    h, r = _idx("subject_idx", subject_idx, dev), _idx("relation_idx", relation_idx, dev)
    B = h.numel()
    dcode = _lib.RTK_BF16 if bf16 else _lib.RTK_F32
    v = torch.empty((B, c), dtype=torch.float32, device=dev)
    qp = torch.empty(lib.rtk_packed_query_bytes(dcode, B, c), dtype=torch.uint8, device=dev) if packed else None
    with torch.cuda.device(dev):
        sp = _stream_ptr(dev)
        if tables is not None:
            ws = _workspace(dev, sp, lib.rtk_from_tables_workspace_bytes(B, R.shape[0]))
            ft = lib.rtk_query_vectors_from_tables_bf16 if bf16 else lib.rtk_query_vectors_from_tables_f32
            _lib.check(ft(tables.data_ptr(), R.shape[0], b, c, S.data_ptr(), S.shape[0], r.data_ptr(), h.data_ptr(), B,
                          v.data_ptr(), qp.data_ptr() if packed else None, ws.data_ptr(), ws.numel(), sp),
                       "rtk_query_vectors_from_tables")
        else:
            ws = _workspace(dev, sp, lib.rtk_workspace_bytes(dcode, B, R.shape[0], a, b, c))
            qv = lib.rtk_query_vectors_bf16 if bf16 else lib.rtk_query_vectors_f32
            _lib.check(qv(core.data_ptr(), a, b, c, R.data_ptr(), R.shape[0], S.data_ptr(),
                          S.shape[0], r.data_ptr(), h.data_ptr(), B, v.data_ptr(), qp.data_ptr() if packed else None,
                          ws.data_ptr(), ws.numel(), sp), "rtk_query_vectors")
        _strict_check(dev, ws, sp)
    return (v, qp) if packed else v


def pack_query_vectors(v, dtype):
    """Packed query planes (csrc/rtk_pack.h) of fp32 query vectors ``v (B, c)`` for the score kernels of
    operand type ``dtype`` -- the hand-over when stage 1 ran elsewhere (entity-sharded scoring with
    stage 1 split over the ranks: every rank contracts its slice of the batch, the B x c vectors are
    all-gathered, each rank packs them and scores its entity shard)."""
    lib = _lib.load()
    _require_gpu("v", v)
    v = v.contiguous().float()
    B, c = v.shape
    bf16 = dtype == torch.bfloat16
    dcode = _lib.RTK_BF16 if bf16 else _lib.RTK_F32
    qp = torch.empty(lib.rtk_packed_query_bytes(dcode, B, c), dtype=torch.uint8, device=v.device)
    with torch.cuda.device(v.device):
        _lib.check(lib.rtk_pack_query_vectors(v.data_ptr(), B, c, dcode, qp.data_ptr(), _stream_ptr(v.device)),
                   "rtk_pack_query_vectors")
    return qp


def score_packed_into(qp, B, O, out, sigmoid=True, sigmoid_mode=None):
    """Stage 2 alone: ``out[d, j] = logistic(v_d . O[j])`` from packed query planes into a (B, n_local)
    buffer (float32, or bfloat16 for bf16 operands)."""
    lib = _lib.load()
    _require_gpu("O", O)
    dev = O.device
    bf16 = O.dtype == torch.bfloat16
    O = O.contiguous()
    N, c = O.shape
    mode = sigmoid_mode or DEFAULT_SIGMOID
    flags = (_lib.RTK_SCORE_SIGMOID if sigmoid else 0) | (_lib.RTK_SCORE_SIGMOID_FAST if (sigmoid and mode == "fast") else 0)
    if out.dtype == torch.bfloat16:
        if not bf16 or not sigmoid or mode != "fast":
            raise RuntimeError("bfloat16 scores: bf16 operands and the fast logistic")
        flags |= _lib.RTK_SCORE_OUT_BF16
    if tuple(out.shape) != (B, N) or out.stride(1) != 1 or out.device != dev:
        raise RuntimeError(f"out must be ({B}, {N}) on {dev} with unit column stride")
    fn = lib.rtk_score_packed_bf16 if bf16 else lib.rtk_score_packed_f32
    with torch.cuda.device(dev):
        _lib.check(fn(qp.data_ptr(), B, c, O.data_ptr(), N, out.data_ptr(), out.stride(0) if B > 1 else N, flags,
                      _stream_ptr(dev)), "rtk_score_packed")
    return out


def cg_fifth_group_columns(N, c, n_cu=256):
    """Boolean mask (N,) of the entity columns that the default fp32 score kernel computes as FOUR K-range chains added
    in a fixed order instead of one chain -- the fifth column group of a workgroup's set in the column-group kernel
    (csrc/rtk_score_cg.hip: chosen for one set per workgroup, 18 432 <= N <= 40 960 on 256 CUs, c <= 208, c % 4 == 0;
    a set has a fifth group only above 1024 groups, N > 32 768).  Everywhere else (all False) a score does not depend
    on how many other entities are scored with it; on these columns an entity-sharded run and a single-device run
    differ in the last bits."""
    import numpy as np
    mask = np.zeros(N, dtype=bool)
    G = -(-N // 32)
    sets = -(-G // 5)
    if sets < n_cu:
        sets = min(G, n_cu)
    W = min(n_cu, sets)
    P = -(-sets // W)
    if c > 208 or c % 4 or not (P == 1 and G >= 576):
        return mask
    U = W * P
    for u in range(U):
        gb, ge = G * u // U, G * (u + 1) // U
        if ge - gb == 5:
            mask[(gb + 4) * 32:min(N, (gb + 5) * 32)] = True
    return mask


def check_device_errors(device=None):
    """Synchronise and raise ``IndexError`` if a kernel saw an out-of-range subject /
    relation id since the last check (the kernels clamp such ids instead of faulting;
    the reference raises IndexError on CPU / asserts on device).  Each workspace's word is read and
    cleared on the stream that owns it."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    bad = None
    for (di, sp), ws in list(_workspaces.items()):
        if di != dev.index:
            continue
        try:
            _check_now(ws, sp)
        except IndexError as e:
            bad = e
    if bad is not None:
        raise bad
