// Split-fp16 GEMM for the B x N sized backward products of the scoring closure
// (autograd of /root/reference/src/model/asymmetric/R_TuckER.py:47 under train.py:79-82):
//   dO[j,k] = sum_d dZ[d,j] v[d,k]     (M = entities, N = c, K = batch)
//   dv[d,k] = sum_j dZ[d,j] O[j,k]     (M = batch, N = c, K = entities; split over K)
// fp32 operands, fp32 result.  Each operand is multiplied by ONE power of two that puts its largest
// magnitude (an upper bound the caller supplies on the device) just below 2^15, and every element is split
// x' = hi + lo into two fp16 values while the tile is staged into LDS; a k-step is three f16 MFMAs
// (hi.hi + hi.lo + lo.hi; v_mfma_f32_32x32x16_f16, fp32 accumulation), 16x the fp32-MFMA rate for a third
// of the products.  Per element the split keeps max(2^-22 |x'|, 2^-25) of absolute accuracy, i.e. the
// product is accurate NORMWISE (relative to max|A| max|B| K), not element-wise: exactly what a gradient that
// is summed over a batch needs, not what an orthogonalisation needs (those stay on rtk_gemm_f32).
#include "rtk_common.h"

int rtk_splitk_reduce_launch(const float *slabs, int64_t slab, int splits, float *C, int64_t n, hipStream_t st);   // rtk_gemm_f32.hip

namespace {

constexpr int BM = 128, BN = 128, BK = 32;            // BK = two MFMA k-steps of 16
constexpr int FRAGS_PER_PLANE = 2 * 2 * 128;          // [k-step][half][row] fragments of 8 fp16
constexpr int STAGE_FRAGS = 4 * FRAGS_PER_PLANE;      // A hi, A lo, B hi, B lo: 32 KB

__device__ __forceinline__ int frag_index(int plane, int ks, int h, int row) {
    return ((plane * 2 + ks) * 2 + h) * 128 + row;
}

// 16 staged fp32 values of one thread.  M-major operand (element (row, k) at P[k*ld + row]): thread = (row,
// k-step), its 16 values are the 16 k of that k-step -- every load instruction reads 64 consecutive rows of
// one k (256 contiguous bytes).  K-major operand (P[row*ld + k]): thread = (row, half), values [0,8) are its
// 8 k of k-step 0 and [8,16) of k-step 1 -- two 16-byte loads each.
struct Stage16 { float x[16]; };

// Buffer loads: the workgroup-uniform part of the address (tile origin, k) lives in the descriptor and the
// scalar offset, advanced by the scalar unit; the per-thread part is one 32-bit offset that never changes -- no
// vector address arithmetic in the loop, and NO branch: a conditional load makes the compiler drain every
// outstanding load (s_waitcnt vmcnt(0)) where the paths meet, which serialised the k-tiles behind the memory
// latency.  Instead
//   * rows past the operand are CLAMPED to its last row: row m of A (n of B) only ever reaches row m (column n)
//     of the product, which is not stored;
//   * k past the end of the chunk is clamped (M-major) or simply read (K-major: the next row's elements; the
//     descriptor ends with the operand, so nothing is read past the allocation) and ZEROED in stash_stage.
// `extent`: addressable elements from P ((rows-1)*ld + K for K-major, (K-1)*ld + rows for M-major).
__device__ __forceinline__ float bload(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0));
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t operand_rsrc(const float *P, int64_t first, int64_t extent) {
    const int64_t left = (extent - first) * 4;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(P + first), 0,
                                             (unsigned)(left < 0xffffffffll ? left : 0xffffffffll), 0x00020000);
}

template <bool KMAJOR>
__device__ __forceinline__ void load_stage(Stage16 &s, const float *__restrict__ P, int64_t ld, int64_t extent, int row0,
                                           int rows, int k0, int k_end, int t, bool vec) {
    const int ld4 = (int)ld * 4;                                         // ld < 2^22 (host)
    if (!KMAJOR) {
        const int voff = min(t & 127, rows - 1 - row0) * 4;
        const int ksw = __builtin_amdgcn_readfirstlane(t >> 7);          // a wave is on one k-step
        const int kmax = k_end - 1 - k0;                                 // >= 0: tiles start inside the chunk
        const __amdgpu_buffer_rsrc_t rs = operand_rsrc(P, (int64_t)k0 * ld + row0, extent);
#pragma unroll
        for (int j = 0; j < 16; ++j) s.x[j] = bload(rs, voff, min(ksw * 16 + j, kmax) * ld4);
    } else {
        const int voff = min(t >> 1, rows - 1 - row0) * ld4 + (t & 1) * 32;    // 128 rows x ld x 4 bytes < 2^31
        const __amdgpu_buffer_rsrc_t rs = operand_rsrc(P, (int64_t)row0 * ld + k0, extent);
        if (vec) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                // (bit_cast of the whole vector: element-wise bit_casts of the builtin's result compile to a
                // one-dword load broadcast to all four elements with this toolchain)
                const f32x4 a = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, ks * 64, 0));
                const f32x4 b = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, ks * 64 + 16, 0));
#pragma unroll
                for (int j = 0; j < 4; ++j) { s.x[ks * 8 + j] = a[j]; s.x[ks * 8 + 4 + j] = b[j]; }
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = 0; j < 8; ++j) s.x[ks * 8 + j] = bload(rs, voff, (ks * 16 + j) * 4);
        }
    }
}

// One fragment (8 staged values of one thread) -> scale, split into hi/lo fp16, two 16-byte LDS writes.
// `half` selects values [0,8) or [8,16) of the stage.  k past the end of the chunk (kvalid < BK, last tile only)
// must not contribute: an M-major operand multiplies those elements by a zero SCALE, which costs nothing (its
// k is wave-uniform, the scale is picked by the scalar unit; the elements themselves are valid, clamped reads);
// a K-major operand read whatever follows its row there (the next row, or the padding of a score buffer:
// possibly inf / nan bit patterns), so its VALUES are selected lane by lane (ZERO_TAIL), in the last tile only.
template <bool KMAJOR, bool ZERO_TAIL>
__device__ __forceinline__ void stash_fragment(const Stage16 &s, int half, f16x8 *__restrict__ lds, int plane_hi,
                                               float scale, int t, int kvalid) {
    int row, ks, hh;
    if (!KMAJOR) { row = t & 127; ks = t >> 7; hh = half; }
    else { row = t >> 1; ks = half; hh = t & 1; }
    const int ksw = __builtin_amdgcn_readfirstlane(t >> 7);
    f16x8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float sc = scale;
        if (!KMAJOR) sc = (ksw * 16 + half * 8 + j < kvalid) ? scale : 0.f;             // uniform
        float y = s.x[half * 8 + j] * sc;
        if (KMAJOR && ZERO_TAIL) y = (half * 16 + (t & 1) * 8 + j < kvalid) ? y : 0.f;  // per lane
        const _Float16 hv = (_Float16)y;
        hi[j] = hv;
        lo[j] = (_Float16)(y - (float)hv);
    }
    lds[frag_index(plane_hi, ks, hh, row)] = hi;
    lds[frag_index(plane_hi + 1, ks, hh, row)] = lo;
}

// 2^e with max|x| * 2^e in [2^14, 2^15) for the bound `amax` (0, inf or nan -> 1: nothing to scale / nothing to save)
__device__ __forceinline__ float operand_scale(float amax, int &e) {
    const unsigned bits = __builtin_bit_cast(unsigned, amax) & 0x7fffffffu;
    const int ex = (int)(bits >> 23);             // biased exponent; 0 = zero / subnormal
    e = (ex == 0 || ex == 255) ? 0 : 14 - (ex - 127);
    e = max(-100, min(100, e));
    return __builtin_bit_cast(float, (unsigned)(127 + e) << 23);
}

template <bool A_KMAJOR, bool B_KMAJOR>
__global__ __launch_bounds__(256, 2) void gemm_sf16_kernel(
    const float *__restrict__ A, int64_t lda, bool a_vec, const float *__restrict__ amax_a,
    const float *__restrict__ B, int64_t ldb, bool b_vec, const float *__restrict__ amax_b,
    float *__restrict__ C, int64_t ldc, int M, int N, int K, int k_chunk, int64_t slab, unsigned nx, unsigned ny,
    unsigned nz) {
    __shared__ __attribute__((aligned(16))) f16x8 lds[2][STAGE_FRAGS];      // 64 KB: two workgroups per CU

    // Workgroups are dealt round-robin to the 8 XCDs (own L2 each) in launch order.  The tiles that share an
    // operand tile -- the column tiles of one row tile, the row/column tiles of one K chunk -- are consecutive in
    // (x, y, z) order, i.e. they would land on 8 DIFFERENT L2s and each fetch the operand from HBM (dO at c = 400:
    // dZ read 4 times).  XCD x takes the x-th eighth of the tile list instead (1-D grid of 8 * per workgroups).
    const unsigned total = nx * ny * nz, per = (total + 7) / 8;
    const unsigned logical = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (logical >= total) return;
    const unsigned bx = logical % nx, by = (logical / nx) % ny, bz = logical / (nx * ny);
    const int tile_m = by * BM, tile_n = bx * BN;
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;

    int ea, eb;
    const float sa = operand_scale(*amax_a, ea), sb = operand_scale(*amax_b, eb);
    // 2^-(ea+eb) in two factors: each is a normal fp32 for |e| <= 100
    const float ua = __builtin_bit_cast(float, (unsigned)(127 - ea) << 23);
    const float ub = __builtin_bit_cast(float, (unsigned)(127 - eb) << 23);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const bool split = k_chunk > 0;
    const int kb = split ? (int)bz * k_chunk : 0;
    const int ke = split ? min(K, kb + k_chunk) : K;
    if (split) C += (int64_t)bz * slab;

    const int64_t ext_a = A_KMAJOR ? (int64_t)(M - 1) * lda + K : (int64_t)(K - 1) * lda + M;
    const int64_t ext_b = B_KMAJOR ? (int64_t)(N - 1) * ldb + K : (int64_t)(K - 1) * ldb + N;
    const int k_last = kb + max(0, (ke - kb - 1) / BK) * BK;      // start of the chunk's last k-tile
    // (tiles past the chunk are "loaded" too -- the last tile again, never used: no branch around a load)
    auto load = [&](int k0, Stage16 &xa, Stage16 &xb) {
        k0 = min(k0, k_last);
        load_stage<A_KMAJOR>(xa, A, lda, ext_a, tile_m, M, k0, ke, t, a_vec);
        load_stage<B_KMAJOR>(xb, B, ldb, ext_b, tile_n, N, k0, ke, t, b_vec);
    };
    auto stash = [&](const Stage16 &xa, const Stage16 &xb, int buf, int k0) {
        stash_fragment<A_KMAJOR, true>(xa, 0, lds[buf], 0, sa, t, ke - k0);
        stash_fragment<B_KMAJOR, true>(xb, 0, lds[buf], 2, sb, t, ke - k0);
        stash_fragment<A_KMAJOR, true>(xa, 1, lds[buf], 0, sa, t, ke - k0);
        stash_fragment<B_KMAJOR, true>(xb, 1, lds[buf], 2, sb, t, ke - k0);
    };
    // The same conversion cut into 24 pieces, one per MFMA of a step (g constant after unrolling): fragment
    // g / 6 (A half 0, B half 0, A half 1, B half 1); pieces 0..3 of a fragment convert two values each,
    // piece 4 writes its hi and lo vectors to LDS, piece 5 is empty.
    f16x8 chi, clo;
    auto piece = [&](int g, const Stage16 &xa, const Stage16 &xb, f16x8 *__restrict__ dst, int kvalid) {
        const int f = g / 6, q = g % 6, half = f >> 1;
        const bool is_b = f & 1;
        const bool kmaj = is_b ? B_KMAJOR : A_KMAJOR;
        const Stage16 &st = is_b ? xb : xa;
        const float scale = is_b ? sb : sa;
        const int ksw = __builtin_amdgcn_readfirstlane(t >> 7);
        if (q < 4) {
#pragma unroll
            for (int e = 2 * q; e < 2 * q + 2; ++e) {
                float sc = scale;
                if (!kmaj) sc = (ksw * 16 + half * 8 + e < kvalid) ? scale : 0.f;                        // uniform
                float y = st.x[half * 8 + e] * sc;
                if (kmaj && kvalid < BK) y = (half * 16 + (t & 1) * 8 + e < kvalid) ? y : 0.f;           // last tile only
                const _Float16 hv = (_Float16)y;
                chi[e] = hv;
                clo[e] = (_Float16)(y - (float)hv);
            }
        } else if (q == 4) {
            const int row = kmaj ? t >> 1 : t & 127, ks = kmaj ? half : t >> 7, hh = kmaj ? t & 1 : half;
            dst[frag_index(is_b ? 2 : 0, ks, hh, row)] = chi;
            dst[frag_index(is_b ? 3 : 1, ks, hh, row)] = clo;
        }
    };

    // Two k-tiles in flight in registers.  One step: tile i is multiplied out of lds[buf] -- all 16 fragments of
    // both k-steps read up front -- while tile i+2 is requested from memory and tile i+1, requested during the
    // previous step, is converted and written to lds[buf ^ 1] piece by piece BETWEEN the 24 MFMAs (about eight
    // VALU issue slots fit in the shadow of one 32x32x16 MFMA).  The scheduler bunches the MFMAs together
    // unless every piece is fenced (sched_barrier), and a branch anywhere in the step splits it, so edge tiles
    // run all their MFMAs (rows / columns past the matrix are never stored) and the last step converts a tile
    // nobody reads.  Measured (dO at the WN18RR shape, 67 us = 0.37 PF of f16 MFMA, 2.2x the fp32 MFMA GEMM):
    // without the MFMAs 54 us, MFMAs + fragment reads alone 32 us, + loads 41 us, + conversion 47 us -- the
    // staging side (4-byte-per-lane loads of the M-major operands: 16-byte loads were worth 10 us in an
    // ablation, then the conversion and LDS round trip), not the matrix pipe, bounds this kernel.
    // (a trailing chunk that starts past K -- possible when splits * k_chunk overshoots K by more than a chunk --
    // has nothing to read: it only writes its zero slab)
    if (kb < ke) {
    Stage16 ra[2], rb[2];
    load(kb, ra[0], rb[0]);
    load(kb + BK, ra[1], rb[1]);
    stash(ra[0], rb[0], 0, kb);
    __syncthreads();
    int buf = 0;
    int k0 = kb;
#define RTK_SF16_STEP(S0, S1)                                                                                   \
    {                                                                                                           \
        load(k0 + 2 * BK, ra[S0], rb[S0]);                                                                      \
        const f16x8 *L = lds[buf];                                                                              \
        f16x8 fa[2][2][2], fb[2][2][2];      /* [k-step][row / column block][hi, lo] */                         \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                        \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                     \
                fa[ks][i][0] = L[frag_index(0, ks, h, wm * 64 + i * 32 + r)];                                   \
                fa[ks][i][1] = L[frag_index(1, ks, h, wm * 64 + i * 32 + r)];                                   \
                fb[ks][i][0] = L[frag_index(2, ks, h, wn * 64 + i * 32 + r)];                                   \
                fb[ks][i][1] = L[frag_index(3, ks, h, wn * 64 + i * 32 + r)];                                   \
            }                                                                                                   \
        _Pragma("unroll") for (int g = 0; g < 24; ++g) {                                                        \
            const int ks = g / 12, i = (g / 6) % 2, j = (g / 3) % 2, m = g % 3;   /* m: hi.hi, hi.lo, lo.hi */  \
            __builtin_amdgcn_sched_barrier(0);                                                                  \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[ks][i][m == 2], fb[ks][j][m == 1], acc[i][j], 0, 0, 0); \
            __builtin_amdgcn_sched_barrier(0);                                                                  \
            piece(g, ra[S1], rb[S1], lds[buf ^ 1], ke - (k0 + BK));                                             \
        }                                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
        __syncthreads();                                                                                        \
        buf ^= 1;                                                                                               \
        k0 += BK;                                                                                               \
    }
    while (k0 < ke) {
        RTK_SF16_STEP(0, 1)
        if (k0 >= ke) break;
        RTK_SF16_STEP(1, 0)
    }
    }
#undef RTK_SF16_STEP

    // C/D map of the 32x32 MFMA: column = lane & 31, row = (e & 3) + 8*(e >> 2) + 4*(lane >> 5)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = tile_n + wn * 64 + j * 32 + r;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = tile_m + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (m < M && n < N) C[(int64_t)m * ldc + n] = acc[i][j][e] * ua * ub;
            }
        }
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int rtk_gemm_sf16_splitk(const float *A, int a_kmajor, int64_t lda, const float *amax_a, const float *B,
                                    int b_kmajor, int64_t ldb, const float *amax_b, float *C, int64_t ldc, int64_t M,
                                    int64_t N, int64_t K, int splits, void *workspace, size_t workspace_bytes,
                                    void *stream) {
    RTK_REQUIRE(A && B && C && amax_a && amax_b, RTK_ERR_BAD_ARG, "rtk_gemm_sf16_splitk: null operand");
    RTK_REQUIRE(M > 0 && N > 0 && K > 0 && splits > 0, RTK_ERR_BAD_ARG, "rtk_gemm_sf16_splitk: sizes must be positive");
    RTK_REQUIRE(M < (1ll << 31) && N < (1ll << 31) && K < (1ll << 31), RTK_ERR_UNSUPPORTED,
                "rtk_gemm_sf16_splitk: dimension exceeds 2^31-1");
    RTK_REQUIRE(ldc >= N, RTK_ERR_BAD_ARG, "rtk_gemm_sf16_splitk: ldc < N");
    RTK_REQUIRE(splits == 1 || ldc == N, RTK_ERR_BAD_ARG, "rtk_gemm_sf16_splitk: split-K needs a contiguous C (ldc == N)");
    RTK_REQUIRE(lda >= (a_kmajor ? K : M) && ldb >= (b_kmajor ? K : N), RTK_ERR_BAD_ARG,
                "rtk_gemm_sf16_splitk: leading dimension shorter than the row");
    RTK_REQUIRE(rtk_cdiv(M, BM) <= 65535 && splits <= 65535, RTK_ERR_UNSUPPORTED, "rtk_gemm_sf16_splitk: grid too large");
    RTK_REQUIRE(lda < (1ll << 22) && ldb < (1ll << 22), RTK_ERR_UNSUPPORTED, "rtk_gemm_sf16_splitk: leading dimension >= 2^22");
    hipStream_t st = (hipStream_t)stream;
    int k_chunk = 0;
    int64_t slab = 0;
    float *dst = C;
    if (splits > 1) {
        const size_t need = rtk_gemm_f32_splitk_workspace_bytes(M, N, splits);
        RTK_REQUIRE(workspace && workspace_bytes >= need, RTK_ERR_WORKSPACE,
                    "rtk_gemm_sf16_splitk: workspace of %zu bytes given, %zu needed", workspace_bytes, need);
        RTK_REQUIRE(aligned16(workspace), RTK_ERR_WORKSPACE, "rtk_gemm_sf16_splitk: workspace must be 16-byte aligned");
        slab = (int64_t)(rtk_align_up((size_t)M * (size_t)N * sizeof(float), 256) / sizeof(float));
        k_chunk = (int)(rtk_cdiv(rtk_cdiv(K, splits), BK) * BK);
        dst = (float *)workspace;
    }
    const bool a_vec = a_kmajor && aligned16(A) && lda % 4 == 0, b_vec = b_kmajor && aligned16(B) && ldb % 4 == 0;
    const unsigned nx = (unsigned)rtk_cdiv(N, BN), ny = (unsigned)rtk_cdiv(M, BM), nz = (unsigned)splits;
    RTK_REQUIRE((uint64_t)nx * ny * nz < (1ull << 31) - 8, RTK_ERR_UNSUPPORTED, "rtk_gemm_sf16_splitk: grid too large");
    dim3 grid(((nx * ny * nz + 7) / 8) * 8);
#define RTK_SF16_LAUNCH(AK, BK_)                                                                                     \
    hipLaunchKernelGGL((gemm_sf16_kernel<AK, BK_>), grid, dim3(256), 0, st, A, lda, a_vec, amax_a, B, ldb, b_vec,   \
                       amax_b, dst, ldc, (int)M, (int)N, (int)K, k_chunk, slab, nx, ny, nz)
    if (a_kmajor && b_kmajor) RTK_SF16_LAUNCH(true, true);
    else if (a_kmajor) RTK_SF16_LAUNCH(true, false);
    else if (b_kmajor) RTK_SF16_LAUNCH(false, true);
    else RTK_SF16_LAUNCH(false, false);
#undef RTK_SF16_LAUNCH
    int rc = rtk_check_launch("rtk_gemm_sf16_splitk");
    if (rc != RTK_OK || splits == 1) return rc;
    return rtk_splitk_reduce_launch((const float *)workspace, slab, splits, C, M * N, st);
}

// max |x| over a rows x cols fp32 matrix with row pitch ld -> *out (the operand bounds of rtk_gemm_sf16_splitk).
// Non-negative floats order like their bit patterns, so the workgroups combine with an integer atomicMax: the
// result does not depend on the order (deterministic).  NaNs are skipped (fmaxf).
namespace {
__global__ __launch_bounds__(256) void absmax_kernel(const float *__restrict__ x, int64_t rows, int64_t cols, int64_t ld,
                                                     bool vec, unsigned *__restrict__ out) {
    __shared__ float part[4];
    const int t = threadIdx.x;
    float m = 0.f;
    const int64_t nvec = vec ? cols / 4 : 0;
    for (int64_t row = blockIdx.y; row < rows; row += gridDim.y) {
        const float *p = x + row * ld;
        // four 16-byte loads in flight per thread (one at a time the kernel ran at 2 TB/s)
        const int64_t stride = (int64_t)gridDim.x * 256;
        int64_t i = (int64_t)blockIdx.x * 256 + t;
        for (; i + 3 * stride < nvec; i += 4 * stride) {
            f32x4 v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = reinterpret_cast<const f32x4 *>(p)[i + q * stride];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                m = fmaxf(fmaxf(m, fmaxf(fabsf(v[q][0]), fabsf(v[q][1]))), fmaxf(fabsf(v[q][2]), fabsf(v[q][3])));
        }
        for (; i < nvec; i += stride) {
            const f32x4 v = reinterpret_cast<const f32x4 *>(p)[i];
            m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
        }
        for (int64_t i = nvec * 4 + (int64_t)blockIdx.x * 256 + t; i < cols; i += (int64_t)gridDim.x * 256)
            m = fmaxf(m, fabsf(p[i]));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((t & 63) == 0) part[t >> 6] = m;
    __syncthreads();
    if (t == 0) {
        m = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
        if (m > 0.f) atomicMax(out, __builtin_bit_cast(unsigned, m));
    }
}
}  // namespace

extern "C" int rtk_absmax_f32(const float *x, int64_t rows, int64_t cols, int64_t ld, float *out, void *stream) {
    RTK_REQUIRE(x && out, RTK_ERR_BAD_ARG, "rtk_absmax_f32: null pointer");
    RTK_REQUIRE(rows >= 0 && cols >= 0 && ld >= cols, RTK_ERR_BAD_ARG, "rtk_absmax_f32: bad shape");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(out, 0, sizeof(float), st) != hipSuccess) {
        rtk_set_error("rtk_absmax_f32: memset failed");
        return RTK_ERR_LAUNCH;
    }
    if (rows == 0 || cols == 0) return RTK_OK;
    if (ld == cols) { cols *= rows; ld = cols; rows = 1; }            // contiguous: one long row
    const bool vec = aligned16(x) && ld % 4 == 0;
    const int64_t gx = rtk_cdiv(cols, 16384), gy = rows < 4096 ? rows : 4096;      // >= 16 values per thread and row
    dim3 grid((unsigned)(gx < 1 ? 1 : (gx > 2048 ? 2048 : gx)), (unsigned)gy);
    hipLaunchKernelGGL(absmax_kernel, grid, dim3(256), 0, st, x, rows, cols, ld, vec, reinterpret_cast<unsigned *>(out));
    return rtk_check_launch("rtk_absmax_f32");
}
