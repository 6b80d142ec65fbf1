// Exact-fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32).
//
//   C[m,n] = epi( sum_k A(m,k) * B(n,k) )
//
// Used for (a) the exact-fp32 score stage  Z = v . O^T  (+ fused sigmoid)
// [reference: src/model/asymmetric/R_TuckER.py:47-48], (b) the relation tables
// M_u = R[u,:] . G_(0) when the relation rank is large [R_TuckER.py:45 regrouped by
// distinct relation], and (c) the backward GEMMs.  fp32-input MFMA is bit-for-bit
// a k-ordered fmaf chain (MI355X_MICROARCH.md, Matrix cores), i.e. the same
// numerics class as the reference's CPU sgemm.
//
// Tiling: 128x128 block tile, BK = 16, 4 waves as 2x2, each wave 64x64 = 2x2
// MFMA tiles of 32x32 (64 accumulator VGPRs).  Both operands are staged in LDS
// k-major-transposed ([k][m], m contiguous) so a fragment read is one
// conflict-free ds_read_b32 per lane; the next k-tile's global loads are issued
// before the MFMAs of the current one (register staging, write after barrier).
#include "rtk_common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 16, LDT = 132;  // LDT: padded LDS row (floats)

// 8 consecutive elements of one operand's tile for this thread.
struct Stage8 {
    float v[8];
};

// K-major operand: element (row, k) at base[row*ld + k].  Thread t owns row
// (t & 127) of the tile and k in [kh, kh+8), kh = 8*(t >> 7).
template <typename T>
__device__ __forceinline__ void load_kmajor(Stage8 &s, const T *__restrict__ base, int64_t ld,
                                            const int32_t *__restrict__ rows, int row0, int nrows,
                                            int k0, int K, int t, bool vec_ok) {
    const int r = row0 + (t & 127);
    const int k = k0 + 8 * (t >> 7);
#pragma unroll
    for (int j = 0; j < 8; ++j) s.v[j] = 0.f;
    if (r >= nrows) return;
    const int64_t rr = rows ? (int64_t)rows[r] : (int64_t)r;
    const T *p = base + rr * ld + k;
    if (vec_ok && k + 8 <= K) {
        const f32x4 x0 = rtk_load4(p);
        const f32x4 x1 = rtk_load4(p + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s.v[j] = x0[j];
            s.v[4 + j] = x1[j];
        }
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (k + j < K) s.v[j] = rtk_to_f32(p[j]);
    }
}
__device__ __forceinline__ void store_kmajor(const Stage8 &s, float *__restrict__ tile, int t) {
    const int m = t & 127, kh = 8 * (t >> 7);
#pragma unroll
    for (int j = 0; j < 8; ++j) tile[(kh + j) * LDT + m] = s.v[j];
}

// M-major operand: element (row, k) at base[k*ld + row].  Thread t owns k = t >> 4
// and rows [m8, m8+8), m8 = 8*(t & 15).
template <typename T>
__device__ __forceinline__ void load_mmajor(Stage8 &s, const T *__restrict__ base, int64_t ld,
                                            int row0, int nrows, int k0, int K, int t, bool vec_ok) {
    const int k = k0 + (t >> 4);
    const int r = row0 + 8 * (t & 15);
#pragma unroll
    for (int j = 0; j < 8; ++j) s.v[j] = 0.f;
    if (k >= K) return;
    const T *p = base + (int64_t)k * ld + r;
    if (vec_ok && r + 8 <= nrows) {
        const f32x4 x0 = rtk_load4(p);
        const f32x4 x1 = rtk_load4(p + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s.v[j] = x0[j];
            s.v[4 + j] = x1[j];
        }
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (r + j < nrows) s.v[j] = rtk_to_f32(p[j]);
    }
}
__device__ __forceinline__ void store_mmajor(const Stage8 &s, float *__restrict__ tile, int t) {
    const int k = t >> 4, m8 = 8 * (t & 15);
    f32x4 x0, x1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        x0[j] = s.v[j];
        x1[j] = s.v[4 + j];
    }
    *reinterpret_cast<f32x4 *>(&tile[k * LDT + m8]) = x0;
    *reinterpret_cast<f32x4 *>(&tile[k * LDT + m8 + 4]) = x1;
}

template <typename T, bool A_KMAJOR, bool B_KMAJOR, bool SIGMOID>
__global__ __launch_bounds__(256) void gemm_f32_kernel(
    const T *__restrict__ A, int64_t lda, const int32_t *__restrict__ a_rows, bool a_vec,
    const T *__restrict__ B, int64_t ldb, bool b_vec,
    float *__restrict__ C, int64_t ldc, int M, int N, int K,
    const uint32_t *__restrict__ m_dev, int k_chunk, int64_t slab) {
    __shared__ __attribute__((aligned(16))) float As[BK * LDT];
    __shared__ __attribute__((aligned(16))) float Bs[BK * LDT];

    if (m_dev) M = min(M, (int)*m_dev);
    const int tile_m = blockIdx.y * BM, tile_n = blockIdx.x * BN;
    if (tile_m >= M) return;

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    Stage8 sa, sb;
    auto load = [&](int k0) {
        if (A_KMAJOR) load_kmajor(sa, A, lda, a_rows, tile_m, M, k0, K, t, a_vec);
        else load_mmajor(sa, A, lda, tile_m, M, k0, K, t, a_vec);
        if (B_KMAJOR) load_kmajor(sb, B, ldb, nullptr, tile_n, N, k0, K, t, b_vec);
        else load_mmajor(sb, B, ldb, tile_n, N, k0, K, t, b_vec);
    };
    auto stash = [&]() {
        if (A_KMAJOR) store_kmajor(sa, As, t); else store_mmajor(sa, As, t);
        if (B_KMAJOR) store_kmajor(sb, Bs, t); else store_mmajor(sb, Bs, t);
    };

    // split-K: blockIdx.z owns k in [kb, ke) and writes its partial tile to slab z of C
    // (C + z * slab, slab = M * ldc elements); a second kernel adds the slabs in z order, so the
    // summation order is fixed (no float atomics: run-to-run bit-identical).  k_chunk == 0: the whole K.
    const bool split = k_chunk > 0;
    const int kb = split ? blockIdx.z * k_chunk : 0;
    const int ke = split ? min(K, kb + k_chunk) : K;
    if (split) C += (int64_t)blockIdx.z * slab;
    const bool empty = kb >= ke;     // a trailing chunk past K still writes its (zero) slab
    K = empty ? kb : ke;
    if (!empty) {
        load(kb);
        stash();
    }
    __syncthreads();
    for (int k0 = kb; k0 < K; k0 += BK) {
        const bool more = k0 + BK < K;
        if (more) load(k0 + BK);
#pragma unroll
        for (int ks = 0; ks < BK / 2; ++ks) {
            const int kk = 2 * ks + h;
            const float a0 = As[kk * LDT + wm * 64 + r];
            const float a1 = As[kk * LDT + wm * 64 + 32 + r];
            const float b0 = Bs[kk * LDT + wn * 64 + r];
            const float b1 = Bs[kk * LDT + wn * 64 + 32 + r];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();
        if (more) {
            stash();
            __syncthreads();
        }
    }

    // C/D map of the 32x32 MFMA: column = lane & 31, row = (e & 3) + 8*(e >> 2) + 4*(lane >> 5)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = tile_n + wn * 64 + j * 32 + r;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = tile_m + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (m < M && n < N) {
                    float x = acc[i][j][e];
                    if (SIGMOID) x = rtk_sigmoid(x);
                    C[(int64_t)m * ldc + n] = x;
                }
            }
        }
}

template <typename T, bool AK, bool BK_, bool SG>
void launch(const T *A, int64_t lda, const int32_t *a_rows, bool a_vec, const T *B, int64_t ldb,
            bool b_vec, float *C, int64_t ldc, int M, int N, int K, const uint32_t *m_dev,
            hipStream_t st, int k_chunk = 0, int splits = 1, int64_t slab = 0) {
    dim3 grid((unsigned)rtk_cdiv(N, BN), (unsigned)rtk_cdiv(M, BM), (unsigned)(k_chunk > 0 ? splits : 1));
    hipLaunchKernelGGL((gemm_f32_kernel<T, AK, BK_, SG>), grid, dim3(256), 0, st, A, lda, a_rows, a_vec, B,
                       ldb, b_vec, C, ldc, M, N, K, m_dev, k_chunk, slab);
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

// Internal entry (also used by the query-vector stage for large relation rank):
// optional row gather on A (K-major only) and a device-side row count.
template <typename T>
static int gemm_dispatch(const T *A, int a_kmajor, int64_t lda, const int32_t *a_rows, const T *B, int b_kmajor,
                         int64_t ldb, float *C, int64_t ldc, int m, int n, int k, bool sg, const uint32_t *m_dev,
                         hipStream_t st, int k_chunk = 0, int splits = 1, int64_t slab = 0) {
    constexpr uintptr_t VA = rtk_vec4_align<T>();
    const bool a_vec = ((reinterpret_cast<uintptr_t>(A) & (VA - 1)) == 0) && (lda % 4 == 0);
    const bool b_vec = ((reinterpret_cast<uintptr_t>(B) & (VA - 1)) == 0) && (ldb % 4 == 0);
#define RTK_GO(AK, BK_)                                                                                        \
    do {                                                                                                       \
        if (sg) launch<T, AK, BK_, true>(A, lda, a_rows, a_vec, B, ldb, b_vec, C, ldc, m, n, k, m_dev, st, k_chunk, splits, slab);    \
        else launch<T, AK, BK_, false>(A, lda, a_rows, a_vec, B, ldb, b_vec, C, ldc, m, n, k, m_dev, st, k_chunk, splits, slab);      \
    } while (0)
    if (a_kmajor && b_kmajor) RTK_GO(true, true);
    else if (a_kmajor && !b_kmajor) RTK_GO(true, false);
    else if (!a_kmajor && b_kmajor) RTK_GO(false, true);
    else RTK_GO(false, false);
#undef RTK_GO
    return rtk_check_launch("rtk_gemm_f32");
}

// Internal entry (also used by the query-vector stage for large relation rank):
// optional row gather on A (K-major only), a device-side row count, bf16 operands.
int rtk_gemm_f32_ex(const void *A, int a_kmajor, int64_t lda, const int32_t *a_rows,
                    const void *B, int b_kmajor, int64_t ldb, float *C, int64_t ldc,
                    int64_t M, int64_t N, int64_t K, unsigned flags, const uint32_t *m_dev, int in_bf16,
                    hipStream_t st) {
    RTK_REQUIRE(A && B && C, RTK_ERR_BAD_ARG, "rtk_gemm_f32: null operand");
    RTK_REQUIRE(M > 0 && N > 0 && K > 0, RTK_ERR_BAD_ARG, "rtk_gemm_f32: sizes must be positive (M=%lld N=%lld K=%lld)",
                (long long)M, (long long)N, (long long)K);
    RTK_REQUIRE(M < (1ll << 31) && N < (1ll << 31) && K < (1ll << 31), RTK_ERR_UNSUPPORTED,
                "rtk_gemm_f32: dimension exceeds 2^31-1");
    RTK_REQUIRE(rtk_cdiv(M, BM) <= 65535, RTK_ERR_UNSUPPORTED, "rtk_gemm_f32: M too large for grid.y");
    RTK_REQUIRE(!(a_rows && !a_kmajor), RTK_ERR_BAD_ARG, "rtk_gemm_f32: row gather needs a K-major A");
    RTK_REQUIRE(ldc >= N, RTK_ERR_BAD_ARG, "rtk_gemm_f32: ldc < N");
    const bool sg = (flags & RTK_SCORE_SIGMOID) != 0;
    if (in_bf16)
        return gemm_dispatch<rtk_bf16>((const rtk_bf16 *)A, a_kmajor, lda, a_rows, (const rtk_bf16 *)B, b_kmajor, ldb, C,
                                       ldc, (int)M, (int)N, (int)K, sg, m_dev, st);
    return gemm_dispatch<float>((const float *)A, a_kmajor, lda, a_rows, (const float *)B, b_kmajor, ldb, C, ldc,
                                (int)M, (int)N, (int)K, sg, m_dev, st);
}

extern "C" int rtk_gemm_f32(const float *A, int a_kmajor, int64_t lda, const float *B, int b_kmajor,
                            int64_t ldb, float *C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                            unsigned flags, void *stream) {
    return rtk_gemm_f32_ex(A, a_kmajor, lda, nullptr, B, b_kmajor, ldb, C, ldc, M, N, K, flags, nullptr, 0,
                           (hipStream_t)stream);
}

extern "C" int rtk_score_f32(const float *v, int64_t batch, int c, const float *O, int64_t n_local,
                             float *out, int64_t ld_out, unsigned flags, void *stream) {
    RTK_REQUIRE(c > 0, RTK_ERR_BAD_ARG, "rtk_score_f32: c must be positive");
    return rtk_gemm_f32_ex(v, 1, c, nullptr, O, 1, c, out, ld_out, batch, n_local, c,
                           flags & RTK_SCORE_SIGMOID, nullptr, 0, (hipStream_t)stream);
}

namespace {
// C[i] = sum_z slabs[z * slab + i], z ascending (fixed order); i over M*N contiguous floats
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float *__restrict__ slabs, int64_t slab, int splits,
                                                            float *__restrict__ C, int64_t n4, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        f32x4 acc = reinterpret_cast<const f32x4 *>(slabs)[i];
        for (int z = 1; z < splits; ++z) {
            const f32x4 x = reinterpret_cast<const f32x4 *>(slabs + (int64_t)z * slab)[i];
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] += x[q];
        }
        reinterpret_cast<f32x4 *>(C)[i] = acc;
    }
    if (blockIdx.x == 0)
        for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) {
            float acc = slabs[i];
            for (int z = 1; z < splits; ++z) acc += slabs[(int64_t)z * slab + i];
            C[i] = acc;
        }
}
}  // namespace

// C[i] = slabs[0][i] + slabs[1][i] + ... in slab order (shared with rtk_gemm_sf16.hip)
int rtk_splitk_reduce_launch(const float *slabs, int64_t slab, int splits, float *C, int64_t n, hipStream_t st) {
    const int64_t n4 = ((reinterpret_cast<uintptr_t>(C) & 15) == 0) ? n / 4 : 0;
    const int64_t blocks = n4 > 0 ? rtk_cdiv(n4, 256) : 1;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)(blocks < 1024 ? blocks : 1024)), dim3(256), 0, st, slabs,
                       slab, splits, C, n4, n);
    return rtk_check_launch("split-K slab reduction");
}

// Split-K form for short-and-wide products (e.g. dv = dZ . O with K = n_entities): K is cut into
// `splits` chunks handled by separate workgroups, each writes its partial M x N tile to its own slab
// of the caller's workspace, and a second kernel adds the slabs in chunk order -- the summation order
// is a function of the shapes only, so two runs give bit-identical results (no float atomics).
extern "C" size_t rtk_gemm_f32_splitk_workspace_bytes(int64_t M, int64_t N, int splits) {
    if (M <= 0 || N <= 0 || splits <= 0) return 0;
    return rtk_align_up((size_t)M * (size_t)N * sizeof(float), 256) * (size_t)splits;
}

extern "C" int rtk_gemm_f32_splitk(const float *A, int a_kmajor, int64_t lda, const float *B, int b_kmajor,
                                   int64_t ldb, float *C, int64_t ldc, int64_t M, int64_t N, int64_t K, int splits,
                                   void *workspace, size_t workspace_bytes, void *stream) {
    RTK_REQUIRE(A && B && C, RTK_ERR_BAD_ARG, "rtk_gemm_f32_splitk: null operand");
    RTK_REQUIRE(M > 0 && N > 0 && K > 0 && splits > 0, RTK_ERR_BAD_ARG, "rtk_gemm_f32_splitk: sizes must be positive");
    RTK_REQUIRE(M < (1ll << 31) && N < (1ll << 31) && K < (1ll << 31), RTK_ERR_UNSUPPORTED, "rtk_gemm_f32_splitk: dimension exceeds 2^31-1");
    RTK_REQUIRE(ldc == N, RTK_ERR_BAD_ARG, "rtk_gemm_f32_splitk: C must be contiguous (ldc == N)");
    RTK_REQUIRE(rtk_cdiv(M, BM) <= 65535 && splits <= 65535, RTK_ERR_UNSUPPORTED, "rtk_gemm_f32_splitk: grid too large");
    hipStream_t st = (hipStream_t)stream;
    if (splits == 1)
        return gemm_dispatch<float>(A, a_kmajor, lda, nullptr, B, b_kmajor, ldb, C, ldc, (int)M, (int)N, (int)K, false,
                                    nullptr, st);
    const size_t need = rtk_gemm_f32_splitk_workspace_bytes(M, N, splits);
    RTK_REQUIRE(workspace && workspace_bytes >= need, RTK_ERR_WORKSPACE,
                "rtk_gemm_f32_splitk: workspace of %zu bytes given, %zu needed", workspace_bytes, need);
    RTK_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 15) == 0, RTK_ERR_WORKSPACE, "rtk_gemm_f32_splitk: workspace must be 16-byte aligned");
    const int64_t slab = (int64_t)(rtk_align_up((size_t)M * (size_t)N * sizeof(float), 256) / sizeof(float));
    const int k_chunk = (int)rtk_cdiv(rtk_cdiv(K, splits), BK) * BK;
    int rc = gemm_dispatch<float>(A, a_kmajor, lda, nullptr, B, b_kmajor, ldb, (float *)workspace, ldc, (int)M, (int)N,
                                  (int)K, false, nullptr, st, k_chunk, splits, slab);
    if (rc != RTK_OK) return rc;
    return rtk_splitk_reduce_launch((const float *)workspace, slab, splits, C, M * N, st);
}

namespace {
__global__ __launch_bounds__(256) void sigmoid_grad_kernel(const float *__restrict__ dP, const float *__restrict__ P,
                                                           float *__restrict__ dZ, int64_t n4, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const f32x4 g = reinterpret_cast<const f32x4 *>(dP)[i], p = reinterpret_cast<const f32x4 *>(P)[i];
        f32x4 z;
#pragma unroll
        for (int q = 0; q < 4; ++q) z[q] = g[q] * p[q] * (1.0f - p[q]);
        reinterpret_cast<f32x4 *>(dZ)[i] = z;
    }
    if (blockIdx.x == 0)
        for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) dZ[i] = dP[i] * P[i] * (1.0f - P[i]);
}
// the same, row by row, for arrays with different row pitches (dense dP / P, dZ on aligned rows)
__global__ __launch_bounds__(256) void sigmoid_grad_rows_kernel(const float *__restrict__ dP, int64_t ld_dp,
                                                                const float *__restrict__ P, int64_t ld_p,
                                                                float *__restrict__ dZ, int64_t ld_dz, int n) {
    const float *g = dP + (int64_t)blockIdx.y * ld_dp, *p = P + (int64_t)blockIdx.y * ld_p;
    float *z = dZ + (int64_t)blockIdx.y * ld_dz;
    for (int j = blockIdx.x * 256 + threadIdx.x; j < n; j += gridDim.x * 256) {
        const float pj = p[j];
        z[j] = g[j] * pj * (1.0f - pj);
    }
}
}  // namespace

// dZ = dP * P * (1 - P) with a row pitch per array (elements): lets dZ live on 128-byte aligned rows
// (the backward GEMMs read it with vector loads) while dP / P are the caller's dense tensors.
extern "C" int rtk_sigmoid_grad_rows_f32(const float *dP, int64_t ld_dp, const float *P, int64_t ld_p, float *dZ,
                                         int64_t ld_dz, int64_t batch, int64_t n, void *stream) {
    RTK_REQUIRE(dP && P && dZ && batch > 0 && n > 0, RTK_ERR_BAD_ARG, "rtk_sigmoid_grad_rows_f32: bad argument");
    RTK_REQUIRE(ld_dp >= n && ld_p >= n && ld_dz >= n, RTK_ERR_BAD_ARG, "rtk_sigmoid_grad_rows_f32: row pitch < n");
    RTK_REQUIRE(batch <= 65535 && n < (1ll << 31), RTK_ERR_UNSUPPORTED, "rtk_sigmoid_grad_rows_f32: dimension too large");
    const unsigned gx = (unsigned)(rtk_cdiv(n, 256 * 8) < 1 ? 1 : (rtk_cdiv(n, 256 * 8) > 32 ? 32 : rtk_cdiv(n, 256 * 8)));
    hipLaunchKernelGGL(sigmoid_grad_rows_kernel, dim3(gx, (unsigned)batch), dim3(256), 0, (hipStream_t)stream, dP, ld_dp, P,
                       ld_p, dZ, ld_dz, (int)n);
    return rtk_check_launch("rtk_sigmoid_grad_rows_f32");
}

// dZ = dP * P * (1 - P): the logistic's derivative applied to the incoming gradient (backward of
// R_TuckER.py:48).  All three arrays are contiguous with n elements; dZ may alias dP.
extern "C" int rtk_sigmoid_grad_f32(const float *dP, const float *P, float *dZ, int64_t n, void *stream) {
    RTK_REQUIRE(dP && P && dZ && n > 0, RTK_ERR_BAD_ARG, "rtk_sigmoid_grad_f32: bad argument");
    const bool al = ((reinterpret_cast<uintptr_t>(dP) | reinterpret_cast<uintptr_t>(P) | reinterpret_cast<uintptr_t>(dZ)) & 15) == 0;
    const int64_t n4 = al ? n / 4 : 0;
    const int64_t blocks = n4 > 0 ? rtk_cdiv(n4, 256) : 1;
    hipLaunchKernelGGL(sigmoid_grad_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0,
                       (hipStream_t)stream, dP, P, dZ, n4, n);
    return rtk_check_launch("rtk_sigmoid_grad_f32");
}
