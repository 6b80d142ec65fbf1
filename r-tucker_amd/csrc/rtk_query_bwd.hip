// Backward of stage 1 (the query vectors), i.e. autograd of
//     v[d,:] = S[h_d,:] . ( G x_0 R[r_d,:] )            (reference: asymmetric/R_TuckER.py:43-46)
// given dv = d loss / d v  (B x c):
//     W[d,a,b]   = sum_c G[a,b,c] dv[d,c]                       GEMM  (B x c) . (ab x c)^T
//     gRb[d,a]   = sum_b W[d,a,b] S[h_d,b]
//     gSb[d,b]   = sum_a W[d,a,b] R[r_d,a]
//     gG[a,b,c]  = sum_d R[r_d,a] S[h_d,b] dv[d,c]              GEMM  X^T (ab x B) . dv (B x c),  X[d,(a,b)] = R[r_d,a] S[h_d,b]
//     gR[u,:]    = sum_{d: r_d = u} gRb[d,:]     gS[j,:] = sum_{d: h_d = j} gSb[d,:]
// The two GEMMs run on the exact-fp32 MFMA kernel (rtk_gemm_f32.hip).  The row scatter is
// DETERMINISTIC: the first query of every distinct id adds the rows of all queries with that id in
// increasing query order (no atomics, no index_add_), so two runs give bit-identical gradients.
#include "rtk_common.h"

int rtk_gemm_f32_ex(const void *A, int a_kmajor, int64_t lda, const int32_t *a_rows, const void *B,
                    int b_kmajor, int64_t ldb, float *C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                    unsigned flags, const uint32_t *m_dev, int in_bf16, hipStream_t st);

namespace {

__device__ __forceinline__ int64_t clamp_id(int64_t x, int64_t n) { return x < 0 ? 0 : (x >= n ? n - 1 : x); }

// One workgroup per query: the two small contractions of W[d] and the Khatri-Rao row X[d].
__global__ __launch_bounds__(256) void bwd_rows_kernel(const float *__restrict__ W, int a, int b,
                                                       const float *__restrict__ R, int64_t n_rel,
                                                       const float *__restrict__ S, int64_t n_sub,
                                                       const int64_t *__restrict__ rel_idx,
                                                       const int64_t *__restrict__ sub_idx,
                                                       float *__restrict__ rows_R, float *__restrict__ rows_S,
                                                       float *__restrict__ X) {
    extern __shared__ float sm[];
    float *Rs = sm, *Ss = sm + a;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int64_t d = blockIdx.x;
    const int64_t r = clamp_id(rel_idx[d], n_rel), h = clamp_id(sub_idx[d], n_sub);   // bad ids were flagged by the forward
    for (int i = t; i < a; i += 256) Rs[i] = R[r * a + i];
    for (int i = t; i < b; i += 256) Ss[i] = S[h * b + i];
    __syncthreads();
    const int64_t ab = (int64_t)a * b;
    const float *Wd = W + d * ab;
    if (rows_S) {
        for (int col = t; col < b; col += 256) {
            float acc = 0.f;
            for (int ai = 0; ai < a; ++ai) acc = fmaf(Wd[(int64_t)ai * b + col], Rs[ai], acc);
            rows_S[d * b + col] = acc;
        }
    }
    if (rows_R) {
        for (int ai = wave; ai < a; ai += 4) {
            float acc = 0.f;
            for (int col = lane; col < b; col += 64) acc = fmaf(Wd[(int64_t)ai * b + col], Ss[col], acc);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
            if (lane == 0) rows_R[d * a + ai] = acc;
        }
    }
    if (X) {
        float *Xd = X + d * ab;
        for (int64_t i = t; i < ab; i += 256) Xd[i] = Rs[i / b] * Ss[i % b];
    }
}

// Workgroups [0, B): subject rows -> gS;  [B, 2B): relation rows -> gR.  A workgroup whose query is
// not the first with its id exits; the first one sums the rows of all queries with that id, in
// increasing query order, and writes the destination row (the rest of the matrix was zeroed).
// The ids of the batch are copied to LDS once (B <= SC_IDS): the duplicate check and the match lists then
// cost no memory round trip -- read from global memory at every step the kernel was a chain of ~8 dependent
// latencies (34 us for 512 rows).
constexpr int SC_CB = 4;      // column blocks of 256 held in registers: row width <= 1024
constexpr int SC_IDS = 8192;  // ids kept in LDS (32 KB); larger batches read them from global memory
__global__ __launch_bounds__(256) void scatter_rows_kernel(const int64_t *__restrict__ sub_idx,
                                                           const float *__restrict__ rows_S, int b,
                                                           float *__restrict__ gS, int64_t n_sub,
                                                           const int64_t *__restrict__ rel_idx,
                                                           const float *__restrict__ rows_R, int a,
                                                           float *__restrict__ gR, int64_t n_rel, int B) {
    __shared__ unsigned long long masks[4];
    __shared__ int lst[256];
    __shared__ int sid[SC_IDS];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const bool rel = (int)blockIdx.x >= B;
    const int d = rel ? blockIdx.x - B : blockIdx.x;
    const int64_t *ids = rel ? rel_idx : sub_idx;
    const float *rows = rel ? rows_R : rows_S;
    float *dst = rel ? gR : gS;
    const int w = rel ? a : b;
    const int64_t n = rel ? n_rel : n_sub;        // < 2^31 (host)
    if (!dst) return;
    const bool in_lds = B <= SC_IDS;
    if (in_lds) {
        for (int i = t; i < B; i += 256) sid[i] = (int)clamp_id(ids[i], n);
        __syncthreads();
    }
    auto id_at = [&](int i) -> int { return in_lds ? sid[i] : (int)clamp_id(ids[i], n); };
    const int my = id_at(d);
    int dup = 0;
    for (int i = t; i < d; i += 256) dup |= (id_at(i) == my);
    if (__syncthreads_or(dup)) return;
    // Narrow rows (the relation rows: w = a = 10) leave 246 of the 256 threads idle and make the one workgroup of
    // a frequent relation the long pole (all 512 queries on one relation: 75 us; a WN18RR training batch: 36 us).
    // They are summed by `slots` = 256 / w groups of w threads: slot s adds list entries s, s + slots, ... in
    // that order, and the slot sums are added in slot order at the end -- a fixed shape, so still deterministic.
    const int slots = w <= 128 ? 256 / w : 1;
    const int slot = slots > 1 ? t / w : 0, scol = slots > 1 ? t - slot * w : 0;
    const bool sact = slots > 1 && slot < slots;
    float sacc = 0.f;
    float acc[SC_CB];
#pragma unroll
    for (int k = 0; k < SC_CB; ++k) acc[k] = 0.f;
    for (int base = d; base < B; base += 256) {
        // the queries of this block of 256 that carry the id, compacted in query order into lst[]
        const int i = base + t;
        const bool m = i < B && id_at(i) == my;
        const unsigned long long bal = __ballot(m);
        if (lane == 0) masks[wave] = bal;
        __syncthreads();
        int before = 0, count = 0;
#pragma unroll
        for (int wv = 0; wv < 4; ++wv) {
            const int pc = __popcll(masks[wv]);
            before += wv < wave ? pc : 0;
            count += pc;
        }
        if (m) lst[before + __popcll(bal & ((1ull << lane) - 1ull))] = i;
        __syncthreads();
        if (slots > 1) {
            for (int j0 = slot; j0 < count; j0 += 8 * slots) {       // (uniform trip count per wave up to the guard)
                float x[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int j = j0 + q * slots;
                    x[q] = (sact && j < count) ? rows[(int64_t)lst[j] * w + scol] : 0.f;
                }
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (j0 + q * slots < count) sacc += x[q];
            }
        } else {
            // rows added in list (= query) order; the loads of eight rows are in flight together
            for (int j0 = 0; j0 < count; j0 += 8) {
                float x[8][SC_CB];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float *src = rows + (int64_t)lst[min(j0 + q, count - 1)] * w;
#pragma unroll
                    for (int k = 0; k < SC_CB; ++k) x[q][k] = (k * 256 + t < w) ? src[k * 256 + t] : 0.f;
                }
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (j0 + q < count) {
#pragma unroll
                        for (int k = 0; k < SC_CB; ++k) acc[k] += x[q][k];
                    }
            }
        }
        __syncthreads();
    }
    if (slots > 1) {
        float *part = reinterpret_cast<float *>(lst);        // 256 floats: slot sums, then added in slot order
        if (sact) part[t] = sacc;
        __syncthreads();
        if (t < w) {
            float sum = part[t];
            for (int sl = 1; sl < slots; ++sl) sum += part[sl * w + t];
            dst[(int64_t)my * w + t] = sum;
        }
        return;
    }
#pragma unroll
    for (int k = 0; k < SC_CB; ++k)
        if (k * 256 + t < w) dst[(int64_t)my * w + k * 256 + t] = acc[k];
}

struct BwdWs {
    float *W, *X, *rows_R, *rows_S;
    void *slabs;          // split-K slabs of the core-gradient GEMM
    size_t slab_bytes;
    int splits;
    size_t total;
};
// g_core = X^T dv is (a b) x c with K = batch: at the WN18RR rank that is 32 tiles of 128 x 128 for 256 CUs
// (63 us); K is cut until the tiles fill the chip (slabs added in chunk order: deterministic).
int gcore_splits(int64_t batch, int a, int b, int c) {
    const int64_t tiles = rtk_cdiv((int64_t)a * b, 128) * rtk_cdiv(c, 128);
    int64_t s = 256 / (tiles > 0 ? tiles : 1);
    if (s > batch / 64) s = batch / 64;
    return (int)(s < 1 ? 1 : (s > 16 ? 16 : s));
}
BwdWs carve_bwd(void *base, int64_t batch, int a, int b, int c) {
    BwdWs w;
    unsigned char *p = (unsigned char *)base;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        unsigned char *q = p ? p + off : nullptr;
        off += rtk_align_up(bytes, 256);
        return (float *)q;
    };
    const size_t ab = (size_t)a * b;
    w.W = take((size_t)batch * ab * 4);
    w.X = take((size_t)batch * ab * 4);
    w.rows_R = take((size_t)batch * a * 4);
    w.rows_S = take((size_t)batch * b * 4);
    w.splits = gcore_splits(batch, a, b, c);
    w.slab_bytes = w.splits > 1 ? rtk_gemm_f32_splitk_workspace_bytes((int64_t)ab, c, w.splits) : 0;
    w.slabs = take(w.slab_bytes);
    w.total = off;
    return w;
}

}  // namespace

extern "C" size_t rtk_query_bwd_workspace_bytes(int64_t batch, int a, int b, int c) {
    if (batch <= 0 || a <= 0 || b <= 0 || c <= 0) return 0;
    return carve_bwd(nullptr, batch, a, b, c).total;
}

extern "C" int rtk_query_vectors_bwd_f32(const float *core, int a, int b, int c, const float *R, int64_t n_rel,
                                         const float *S, int64_t n_sub, const int64_t *rel_idx,
                                         const int64_t *sub_idx, int64_t batch, const float *dv, float *g_core,
                                         float *g_R, float *g_S, void *workspace, size_t workspace_bytes,
                                         void *stream) {
    const char *fn = "rtk_query_vectors_bwd_f32";
    RTK_REQUIRE(core && R && S && rel_idx && sub_idx && dv, RTK_ERR_BAD_ARG, "%s: null operand", fn);
    RTK_REQUIRE(a > 0 && b > 0 && c > 0 && n_rel > 0 && n_sub > 0 && batch > 0, RTK_ERR_BAD_ARG, "%s: sizes must be positive", fn);
    RTK_REQUIRE(batch < (1ll << 30), RTK_ERR_UNSUPPORTED, "%s: batch too large", fn);
    RTK_REQUIRE(n_rel < (1ll << 31) && n_sub < (1ll << 31), RTK_ERR_UNSUPPORTED, "%s: more than 2^31-1 rows", fn);
    RTK_REQUIRE(a <= 256 * SC_CB && b <= 256 * SC_CB, RTK_ERR_UNSUPPORTED, "%s: rank above %d not supported (a=%d b=%d)", fn, 256 * SC_CB, a, b);
    RTK_REQUIRE((size_t)(a + b) * 4 <= 64 * 1024, RTK_ERR_UNSUPPORTED, "%s: a + b too large for LDS", fn);
    const size_t need = rtk_query_bwd_workspace_bytes(batch, a, b, c);
    RTK_REQUIRE(workspace && workspace_bytes >= need, RTK_ERR_WORKSPACE, "%s: workspace of %zu bytes given, %zu needed", fn, workspace_bytes, need);
    RTK_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, RTK_ERR_WORKSPACE, "%s: workspace must be 256-byte aligned", fn);
    if (!g_core && !g_R && !g_S) return RTK_OK;
    hipStream_t st = (hipStream_t)stream;
    const BwdWs ws = carve_bwd(workspace, batch, a, b, c);
    const int64_t ab = (int64_t)a * b;
    int rc;
    if (g_R || g_S) {
        // W (B x ab) = dv (B x c, K-major) . core viewed as (ab x c, K-major)^T
        rc = rtk_gemm_f32_ex(dv, 1, c, nullptr, core, 1, c, ws.W, ab, batch, ab, c, 0, nullptr, 0, st);
        if (rc != RTK_OK) return rc;
    }
    hipLaunchKernelGGL(bwd_rows_kernel, dim3((unsigned)batch), dim3(256), (size_t)(a + b) * 4, st, ws.W, a, b, R, n_rel, S,
                       n_sub, rel_idx, sub_idx, g_R ? ws.rows_R : nullptr, g_S ? ws.rows_S : nullptr,
                       g_core ? ws.X : nullptr);
    if (g_core) {
        // gG (ab x c) = X^T . dv : A(m,k) = X[k*ab + m] (M-major), B(n,k) = dv[k*c + n] (M-major), K = B in query order
        if (ws.splits > 1)
            rc = rtk_gemm_f32_splitk(ws.X, 0, ab, dv, 0, c, g_core, c, ab, c, batch, ws.splits, ws.slabs, ws.slab_bytes, st);
        else
            rc = rtk_gemm_f32_ex(ws.X, 0, ab, nullptr, dv, 0, c, g_core, c, ab, c, batch, 0, nullptr, 0, st);
        if (rc != RTK_OK) return rc;
    }
    hipError_t e = hipSuccess;
    if (g_R) e = hipMemsetAsync(g_R, 0, (size_t)n_rel * a * 4, st);
    if (e == hipSuccess && g_S) e = hipMemsetAsync(g_S, 0, (size_t)n_sub * b * 4, st);
    if (e != hipSuccess) {
        rtk_set_error("%s: memset: %s", fn, hipGetErrorString(e));
        return RTK_ERR_LAUNCH;
    }
    if (g_R || g_S)
        hipLaunchKernelGGL(scatter_rows_kernel, dim3((unsigned)(2 * batch)), dim3(256), 0, st, sub_idx, ws.rows_S, b, g_S, n_sub,
                           rel_idx, ws.rows_R, a, g_R, n_rel, (int)batch);
    return rtk_check_launch(fn);
}
