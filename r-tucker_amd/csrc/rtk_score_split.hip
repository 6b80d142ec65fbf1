// Stage 2 of the scoring path on the bf16/fp16-rate matrix cores:
//     out[d, j] = sigmoid( v[d,:] . O[j,:] )         (reference: asymmetric/R_TuckER.py:47-48)
//
// fp32 operands are evaluated as a split-precision product on v_mfma_f32_32x32x16_f16:
// x = hi + lo (two fp16 halves of a power-of-two-scaled row, ~22 significand bits) and
//     v.o  ~=  vh.oh + vh.ol + vl.oh           (fp32 accumulation inside the MFMA)
// -- 3 MFMAs at 16x the fp32-MFMA rate, i.e. 5.3x the throughput of an exact fp32 MFMA
// GEMM at about fp32 accuracy (the dropped vl.ol term is 2^-22 relative).  That moves
// this kernel from matrix-core-bound (fp32) to the HBM write of the B x N scores.
//
// Work decomposition ("entity-stationary"): a workgroup of 4 waves owns 128 entity
// rows of O; each wave converts ITS 32 rows once into register-resident B fragments
// (row max -> power-of-two scale -> hi/lo split; O is read from HBM exactly once and
// never written back), then the workgroup sweeps the query tiles: each 32-query tile
// of packed planes (rtk_pack.h) is copied linearly into LDS, every wave reads the same
// A fragments (conflict-free ds_read_b128) and issues 3*KS MFMAs into one 32x32
// accumulator, and the epilogue unscales, applies the logistic and stores
// 128-B-contiguous row segments (lane = entity) straight from the accumulator.
#include "rtk_common.h"
#include "rtk_pack.h"

namespace {

template <int KS, bool SIGMOID, int MINW>
__global__ __launch_bounds__(256, MINW) void score_split_kernel(
    const unsigned char *__restrict__ q_packed, int B, const float *__restrict__ O, int N, int c,
    float *__restrict__ out, int64_t ld_out, int m_split, bool o_vec) {
    constexpr int TILE_BYTES = RTK_PACK_HDR + 2 * KS * 1024;
    __shared__ __attribute__((aligned(16))) unsigned char lds[TILE_BYTES];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int ntile = blockIdx.x / m_split, ms = blockIdx.x % m_split;
    const int j = ntile * 128 + wave * 32 + r;  // entity (row of O, column of out)

    // ---- prologue: this lane's slice of O row j -> scaled hi/lo fp16 fragments ----
    // lane (r, h) holds k = 16*ks + 8*h + jj, jj < 8  (B-operand map of 32x32x16)
    const float *orow = O + (int64_t)min(j, N - 1) * c;
    auto load8 = [&](int ks, float (&x)[8]) {
        const int k = 16 * ks + 8 * h;
        if (o_vec && k + 8 <= c) {
            const f32x4 a = *reinterpret_cast<const f32x4 *>(orow + k);
            const f32x4 b = *reinterpret_cast<const f32x4 *>(orow + k + 4);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                x[q] = a[q];
                x[4 + q] = b[q];
            }
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) x[q] = (k + q < c) ? orow[k + q] : 0.f;
        }
    };
    float mx = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        float x[8];
        load8(ks, x);
#pragma unroll
        for (int q = 0; q < 8; ++q) mx = fmaxf(mx, fabsf(x[q]));
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const int sh = rtk_pack_shift(mx);
    const float up = ldexpf(1.0f, sh);
    const float us_o = ldexpf(1.0f, -sh);
    f16x8 Bh[KS], Bl[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        float x[8];
        load8(ks, x);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float y = x[q] * up;
            const _Float16 hi = (_Float16)y;
            Bh[ks][q] = hi;
            Bl[ks][q] = (_Float16)(y - (float)hi);
        }
    }

    // ---- sweep the query tiles assigned to this block ----
    const int n_mt = (B + 31) / 32;
    const int mt0 = (int)((int64_t)n_mt * ms / m_split), mt1 = (int)((int64_t)n_mt * (ms + 1) / m_split);
    const f16x8 *lh = reinterpret_cast<const f16x8 *>(lds + RTK_PACK_HDR);
    const f16x8 *ll = lh + KS * 64;
    const float *lscale = reinterpret_cast<const float *>(lds);
    for (int mt = mt0; mt < mt1; ++mt) {
        const u32x4 *src = reinterpret_cast<const u32x4 *>(q_packed + (int64_t)mt * TILE_BYTES);
        for (int i = t; i < TILE_BYTES / 16; i += 256) reinterpret_cast<u32x4 *>(lds)[i] = src[i];
        __syncthreads();
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const f16x8 ah = lh[ks * 64 + lane];
            const f16x8 al = ll[ks * 64 + lane];
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, Bh[ks], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, Bl[ks], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, Bh[ks], acc, 0, 0, 0);
        }
        // epilogue: C/D map  column = lane & 31 (entity), row = (e & 3) + 8*(e >> 2) + 4*h
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 sv = *reinterpret_cast<const f32x4 *>(lscale + 8 * g + 4 * h);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int d = mt * 32 + 8 * g + 4 * h + q;
                float z = acc[4 * g + q] * sv[q] * us_o;
                if (SIGMOID) z = rtk_sigmoid(z);
                if (d < B && j < N) out[(int64_t)d * ld_out + j] = z;
            }
        }
        __syncthreads();
    }
}

template <int KS, int MINW>
void launch_ks(const unsigned char *qp, int B, const float *O, int N, int c, float *out, int64_t ld,
               bool sigmoid, int m_split, bool o_vec, hipStream_t st) {
    dim3 grid((unsigned)(rtk_cdiv(N, 128) * m_split));
    if (sigmoid)
        hipLaunchKernelGGL((score_split_kernel<KS, true, MINW>), grid, dim3(256), 0, st, qp, B, O, N, c, out, ld, m_split, o_vec);
    else
        hipLaunchKernelGGL((score_split_kernel<KS, false, MINW>), grid, dim3(256), 0, st, qp, B, O, N, c, out, ld, m_split, o_vec);
}

// choose how many blocks share one 128-entity tile (each takes a slice of the query
// tiles) so that the grid covers the 256 CUs x 2 resident workgroups evenly
int pick_m_split(int64_t n_tiles, int64_t n_mt) {
    const int64_t slots = 512;
    int best = 1;
    double best_eff = 0.0;
    for (int s = 1; s <= 8 && s <= n_mt; ++s) {
        const int64_t blocks = n_tiles * s;
        const int64_t rounds = rtk_cdiv(blocks, slots);
        // efficiency of the last round + cost of converting the O tile s times
        const double eff = (double)blocks / (double)(rounds * slots) / (1.0 + 0.04 * (s - 1));
        if (eff > best_eff + 1e-9) {
            best_eff = eff;
            best = s;
        }
    }
    return best;
}

}  // namespace

int rtk_split_ksteps_supported(int c) {
    const int ks = (c + 15) / 16;
    return ks >= 1 && ks <= 32;
}

extern "C" int rtk_score_packed_f32(const void *q_packed, int64_t batch, int c, const float *O,
                                    int64_t n_local, float *out, int64_t ld_out, unsigned flags,
                                    void *stream) {
    RTK_REQUIRE(q_packed && O && out, RTK_ERR_BAD_ARG, "rtk_score_packed_f32: null operand");
    RTK_REQUIRE(batch > 0 && n_local > 0 && c > 0, RTK_ERR_BAD_ARG, "rtk_score_packed_f32: sizes must be positive");
    RTK_REQUIRE(ld_out >= n_local, RTK_ERR_BAD_ARG, "rtk_score_packed_f32: ld_out < n_local");
    RTK_REQUIRE(batch < (1ll << 31) && n_local < (1ll << 31) - 256, RTK_ERR_UNSUPPORTED, "rtk_score_packed_f32: dimension too large");
    RTK_REQUIRE(rtk_split_ksteps_supported(c), RTK_ERR_UNSUPPORTED, "rtk_score_packed_f32: c=%d > 512 not supported by the split kernel", c);
    hipStream_t st = (hipStream_t)stream;
    const int ks = (c + 15) / 16;
    const bool sg = (flags & RTK_SCORE_SIGMOID) != 0;
    const bool o_vec = (c % 4 == 0) && ((reinterpret_cast<uintptr_t>(O) & 15) == 0);
    const int ms = pick_m_split(rtk_cdiv(n_local, 128), rtk_cdiv(batch, 32));
    const int B = (int)batch, N = (int)n_local;
    const unsigned char *qp = (const unsigned char *)q_packed;
    // the packed planes were written for exactly `ks` k-steps (tile stride), so the
    // instantiation must match exactly.
#define RTK_KS(K_, W_) \
    case K_: launch_ks<K_, W_>(qp, B, O, N, c, out, ld_out, sg, ms, o_vec, st); break;
    switch (ks) {
        RTK_KS(1, 2) RTK_KS(2, 2) RTK_KS(3, 2) RTK_KS(4, 2) RTK_KS(5, 2) RTK_KS(6, 2) RTK_KS(7, 2) RTK_KS(8, 2)
        RTK_KS(9, 2) RTK_KS(10, 2) RTK_KS(11, 2) RTK_KS(12, 2) RTK_KS(13, 2) RTK_KS(14, 2) RTK_KS(15, 2) RTK_KS(16, 2)
        RTK_KS(17, 1) RTK_KS(18, 1) RTK_KS(19, 1) RTK_KS(20, 1) RTK_KS(21, 1) RTK_KS(22, 1) RTK_KS(23, 1) RTK_KS(24, 1)
        RTK_KS(25, 1) RTK_KS(26, 1) RTK_KS(27, 1) RTK_KS(28, 1) RTK_KS(29, 1) RTK_KS(30, 1) RTK_KS(31, 1) RTK_KS(32, 1)
        default:
            rtk_set_error("rtk_score_packed_f32: unsupported k-step count %d", ks);
            return RTK_ERR_UNSUPPORTED;
    }
#undef RTK_KS
    return rtk_check_launch("rtk_score_packed_f32");
}
