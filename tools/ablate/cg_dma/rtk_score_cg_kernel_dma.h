// Split-fp16 score kernel, column-group form (c <= 208, i.e. KS <= 13; fp32 operands).
//
//   out[d, j] = logistic( v[d,:] . O[j,:] )          reference: asymmetric/R_TuckER.py:47-48
//
// Successor of the wave-specialised kernel (rtk_score_ws_kernel.h) for shapes whose entity columns fill the
// chip at least four 32-column groups deep.  The entity columns are cut into G = ceil(N/32) GROUPS and the
// groups are dealt out evenly: a workgroup (512 threads, one per CU, resident for the launch) owns a SET of
// up to five consecutive groups and scores EVERY query tile against it -- at WN18RR (N = 40 943: 1280 groups,
// 256 CUs) exactly five groups per CU.  Consequences against the 128-column tiles of the ws kernel:
//   * O is read from memory exactly once, by exactly one CU, converted once; no tile switch in the middle
//     of the launch and no remainder sweep (that kernel: 1.25 tiles per CU, 22k of 74k cycles in the tail);
//   * a staged 32-query tile is used for 160 entity columns instead of 128: the packed query tiles, the
//     largest stream through the CU's vector-memory path, are re-read 256 x 16 times instead of 320 x 16.
// Roles (two waves per SIMD, as in the ws kernel):
//   waves 0-3  "M"  wave w keeps the hi/lo fp16 B fragments of group w of the set in registers and runs that
//                   group's 3*KS-MFMA chain per query tile.  The FIFTH group is split along K over the four M
//                   waves: wave w also keeps the fragments of k-steps [S0(w), S1(w)) of group 4 (3-4 of 13) and
//                   interleaves those 9-12 MFMAs, on a second accumulator, with its own chain (same A fragments);
//                   the four partial 32x32 accumulators go to LDS raw and are summed -- in wave order, by the
//                   helper waves -- so every SIMD's matrix pipe carries 48-51 MFMAs per tile-step.  The own
//                   group's logistic rides in the gaps of the NEXT tile's chain (tile-alternating accumulators).
//   waves 4-7  "H"  stream the packed query tiles global -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging
//                   registers, no ds_write pass) TWO tiles ahead into a ring of three buffers, store the own
//                   groups' probabilities (tile i-2) from the exchange slots, and sum + scale + squash + store
//                   the fifth group (tile i-1).  Their vector-memory stream is in order (loads, LDS-DMA and
//                   stores share vmcnt), so the wait for a tile in front of the barrier is a COUNTED vmcnt that
//                   leaves the younger stores and the next tile's DMA in flight.
// One barrier per tile-step.  Columns of the four own groups are computed by the same instruction sequence as
// in the ws kernel (bit-identical scores); columns of a fifth group are the sum of four K-range chains.
//
// LDS (KS = 13): [misc 640 B: row factors of the tiles i & 1 (2 x 128 B), fifth group's column factors 2 x 128 B]
//                [query tile ring slot 0][ X: raw O set (<= 160 rows x c fp32) during the prologue;
//                                  then ring slots 1, 2 | own exchange 2 x 16 KiB | partial sums 2 x 16 KiB ]
#pragma once
#include "rtk_common.h"
#include "rtk_pack.h"
#include <type_traits>

namespace rtk_cg {

// tools/ablate only (-DRTK_CG_STAMPS): timeline of M wave 0 and H wave 0 of every workgroup,
// [workgroup][role][event] = code << 56 | s_memtime.  In the product library RTK_CG_TL is empty.
#ifdef RTK_CG_STAMPS
static __device__ unsigned long long g_cg_tl[256 * 2 * 64];
#define RTK_CG_TL(role, code)                                                                                   \
    do {                                                                                                        \
        if (tl_on && tl_n < 64) {                                                                               \
            g_cg_tl[(blockIdx.x * 2 + (role)) * 64 + tl_n] =                                                    \
                ((unsigned long long)(code) << 56) | (__builtin_amdgcn_s_memtime() & 0x00ffffffffffffffull);    \
            ++tl_n;                                                                                             \
        }                                                                                                       \
    } while (0)
#else
#define RTK_CG_TL(role, code) do { (void)tl_on; (void)tl_n; } while (0)
#endif

constexpr int NG = 5;                       // groups per set (4 in registers + 1 split along K)
constexpr int EX_BYTES = 4 * 4 * 64 * 16;   // one exchange buffer: 4 waves x 16 accumulator regs x 64 lanes x f32
constexpr int MISC_BYTES = 640;

template <int KS>
__host__ __device__ constexpr int tile_bytes() { return RTK_PACK_HDR + 2 * KS * 1024; }

template <int KS>
inline size_t lds_bytes(int c) {
    const size_t sweep = 3 * (size_t)tile_bytes<KS>() + 4 * (size_t)EX_BYTES;
    const size_t prologue = (size_t)tile_bytes<KS>() + (size_t)NG * 32 * c * 4;
    return MISC_BYTES + (sweep > prologue ? sweep : prologue);
}

// shared k-step range of M wave w (the fifth group's chain cut in four): ceil(KS*w/4) .. ceil(KS*(w+1)/4)
__host__ __device__ constexpr int s_begin(int KS, int w) { return (KS * w + 3) / 4; }
__host__ __device__ constexpr int s_end(int KS, int w) { return (KS * (w + 1) + 3) / 4; }

struct Geo {
    int B, N, c, U, n_mt;
    int64_t ld_out;
    // set u of U: groups [gb, gb + n_g), n_g <= NG
    __device__ __forceinline__ void set(int u, int &gb, int &n_g) const {
        const int64_t G = ((int64_t)N + 31) / 32;
        gb = (int)(G * u / U);
        n_g = (int)(G * (u + 1) / U) - gb;
    }
};

// All 512 threads: the set's rows of O (contiguous in memory, c % 4 == 0) -> LDS, 16-B pieces, all loads of
// a thread in flight together; rows past N are zero-filled (their columns are never stored).
template <int KS>
__device__ __forceinline__ void load_raw(const float *__restrict__ O, int N, int c, int gb, int n_g,
                                         unsigned char *raw, int t) {
    // (through an empty asm: the per-thread piece addresses are otherwise computed once, ahead of the loop over
    // the sets, and kept -- or spilled -- across the sweep)
    asm volatile("" : "+v"(t));
    constexpr int NCH = (NG * 32 * 16 * KS / 4 + 511) / 512;       // 16-B pieces per thread at c = 16*KS
    const int64_t row0 = (int64_t)gb * 32;
    const int pieces = n_g * 8 * c;                                  // 32 rows x c floats / 4 per group
    const int valid = (int)max((int64_t)0, min((int64_t)n_g * 32, (int64_t)N - row0)) * (c / 4);
    // through a buffer descriptor over the valid rows: a piece past them reads as zero, no branch around
    // any load (a conditional load makes hipcc wait for the previous one: 17 dependent round trips)
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(O + row0 * c), 0, (unsigned)valid * 16u, 0x00020000);
    u32x4 x[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int pc = i * 512 + t;
        x[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)pc * 16u, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int pc = i * 512 + t;
        if (pc < pieces) reinterpret_cast<u32x4 *>(raw)[pc] = x[i];
    }
}

// One row of the raw set -> hi/lo fp16 B fragments of k-steps [K0, K1) and the row's unscale factor.
// The scale comes from the maximum over the WHOLE row (all k-steps), whatever range is converted.
template <int KS, int K0, int K1>
__device__ __forceinline__ float convert_row(const unsigned char *raw, int row, int c, int h, f16x8 *Bh, f16x8 *Bl) {
    // (the offsets go through an empty asm: otherwise the loop-invariant fragment addresses are hoisted out
    // of the sweep loop and stay live across the MFMA chains)
    int row_off = row * c, h8 = 8 * h;
    asm volatile("" : "+v"(row_off), "+v"(h8));
    const float *lrow = reinterpret_cast<const float *>(raw) + row_off;
    float mx = 0.f;
    if constexpr (K1 - K0 == KS) {
        // whole row: read once, keep in registers for the maximum and the conversion
        f32x4 rw[2 * KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int k = 16 * ks + h8;                  // k = 16*ks + 8*h + q  (B-operand map of 32x32x16)
            rw[2 * ks] = *reinterpret_cast<const f32x4 *>(lrow + ((k + 4 <= c) ? k : 0));
            rw[2 * ks + 1] = *reinterpret_cast<const f32x4 *>(lrow + ((k + 8 <= c) ? k + 4 : 0));
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int k = 16 * ks + h8;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (!(k + 4 <= c)) rw[2 * ks][q] = 0.f;
                if (!(k + 8 <= c)) rw[2 * ks + 1][q] = 0.f;
                mx = fmaxf(mx, fmaxf(fabsf(rw[2 * ks][q]), fabsf(rw[2 * ks + 1][q])));
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const int sh = rtk_pack_shift(mx);
        const float up = ldexpf(1.0f, sh);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float y0 = rw[2 * ks][q] * up, y1 = rw[2 * ks + 1][q] * up;
                const _Float16 h0 = (_Float16)y0, h1 = (_Float16)y1;
                Bh[ks][q] = h0;
                Bh[ks][4 + q] = h1;
                Bl[ks][q] = (_Float16)(y0 - (float)h0);
                Bl[ks][4 + q] = (_Float16)(y1 - (float)h1);
            }
        }
        return ldexpf(1.0f, -sh);
    } else {
        static_assert(K1 - K0 == KS, "whole rows only: a k-range goes through convert_range");
        return 0.f;
    }
}

// k-steps [K0, K1) of one row -> hi/lo fp16 B fragments, with the row's scale `up` given (the helper waves
// find the maximum of the fifth group's rows while the M waves convert their own groups)
template <int KS, int K0, int K1>
__device__ __forceinline__ void convert_range(const unsigned char *raw, int row, int c, int h, float up, f16x8 *Bh, f16x8 *Bl) {
    int row_off = row * c, h8 = 8 * h;
    asm volatile("" : "+v"(row_off), "+v"(h8));
    const float *lrow = reinterpret_cast<const float *>(raw) + row_off;
    f32x4 rw[2 * (K1 - K0) + 1];
#pragma unroll
    for (int ks = K0; ks < K1; ++ks) {
        const int k = 16 * ks + h8;
        rw[2 * (ks - K0)] = *reinterpret_cast<const f32x4 *>(lrow + ((k + 4 <= c) ? k : 0));
        rw[2 * (ks - K0) + 1] = *reinterpret_cast<const f32x4 *>(lrow + ((k + 8 <= c) ? k + 4 : 0));
    }
#pragma unroll
    for (int ks = K0; ks < K1; ++ks) {
        const int k = 16 * ks + h8;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float y0 = (k + 4 <= c) ? rw[2 * (ks - K0)][q] * up : 0.f, y1 = (k + 8 <= c) ? rw[2 * (ks - K0) + 1][q] * up : 0.f;
            const _Float16 h0 = (_Float16)y0, h1 = (_Float16)y1;
            Bh[ks - K0][q] = h0;
            Bh[ks - K0][4 + q] = h1;
            Bl[ks - K0][q] = (_Float16)(y0 - (float)h0);
            Bl[ks - K0][4 + q] = (_Float16)(y1 - (float)h1);
        }
    }
}

struct LdsMap {
    unsigned char *hdr;      // 2 x 128 B: row factors of the query tiles t & 1 (copied from the tile's header)
    float *uso5, *up5;       // 2 x 32 floats: unscale / scale factors (powers of two) of the fifth group's columns
    unsigned char *stg0;     // ring of three query tiles: tile t in slot t % 3
    unsigned char *raw;      // prologue: the set's rows of O (over ring slots 1, 2 and the exchange)
    unsigned char *exo;      // 2 x EX_BYTES: own accumulators (probabilities) of tile t & 1
    unsigned char *exp5;     // 2 x EX_BYTES: the four partial accumulators of the fifth group, tile t & 1
    template <int KS>
    __device__ __forceinline__ void init(unsigned char *lds) {
        hdr = lds;
        uso5 = reinterpret_cast<float *>(lds + 384);
        up5 = reinterpret_cast<float *>(lds + 512);
        stg0 = lds + MISC_BYTES;
        raw = stg0 + tile_bytes<KS>();
        exo = stg0 + 3 * tile_bytes<KS>();
        exp5 = exo + 2 * EX_BYTES;
    }
};

// ---- M role: the sweep over the query tiles with the set's fragments in registers --------------------------
template <int KS, int W4, int SIGMOID, bool EXTRA>
__device__ __forceinline__ void m_sweep(const LdsMap &L, int cnt, int lane, const f16x8 (&Bh)[KS], const f16x8 (&Bl)[KS],
                                        const f16x8 *Sh, const f16x8 *Sl, float us_o, bool tl_on, int &tl_n) {
    constexpr int TILE_BYTES = tile_bytes<KS>();
    constexpr int PF = KS < 3 ? KS : 3;         // A-fragment prefetch distance (k-steps)
    constexpr int S0 = s_begin(KS, W4), S1 = s_end(KS, W4), NS = EXTRA ? S1 - S0 : 0;
    constexpr int NGAP = 3 * KS + 3 * NS;       // MFMAs (= gaps) of one chain
    // the partial accumulator is complete after the last shared MFMA (gap 3*S1 + 3*NS - 1); its four 16-B
    // writes take the gaps from three later on (past the chain for the wave whose range ends the chain)
    constexpr int PW0 = !EXTRA ? (1 << 20) : (NS > 0 ? 3 * S1 + 3 * NS + 2 : 32);
    constexpr int GEND = EXTRA ? (PW0 + 4 > 36 ? PW0 + 4 : 36) : 32;
    constexpr bool MSIG = SIGMOID == 2;
    const int h = lane >> 5;
    const float kfac = MSIG ? us_o * -1.4426950408889634f : us_o;
    float ee = 0.f;
    f32x16 accA, accB, accS;
#pragma unroll
    for (int e = 0; e < 16; ++e) accB[e] = 0.f, accS[e] = 0.f;
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    // PAR = i & 1 as a compile-time constant (the loop below is unrolled by two): every LDS address of an
    // iteration but the row-factor slot is then `lane * 16 + constant`
    // CHAIN (tile i exists) and PIECES (tile i-1 exists) are compile-time too: the steady-state loop below has
    // no branch in it.  (With `if (i < cnt)` / `if (i > 0)` inside one body the accumulators met at control-flow
    // joins and hipcc moved them between register sets: ~50 v_mov per tile-step in one of the two bodies.)
    auto iteration = [&](auto par_c, auto chain_c, auto pieces_c, int i, f32x16 &accC, f32x16 &accP) {
        constexpr int PAR = decltype(par_c)::value;
        constexpr bool chain = decltype(chain_c)::value, pieces = decltype(pieces_c)::value;
        f32x4 *exw = reinterpret_cast<f32x4 *>(L.exo + (PAR ^ 1) * EX_BYTES + W4 * 4096);   // own slot of tile i-1
        f32x4 *exs = reinterpret_cast<f32x4 *>(L.exp5 + PAR * EX_BYTES + W4 * 4096);        // partial slot of tile i
        const float *rfp = reinterpret_cast<const float *>(L.hdr + (PAR ^ 1) * 128) + 4 * h;     // row factors of tile i-1
        f32x4 rfq[2];
        // gap g of the chain.  Own hand-over of tile i-1: piece p = 2e / 2e+1 turns accumulator element e into a
        // probability in place (one multiply pair + v_exp_f32, then one add + v_rcp_f32), every fourth element
        // completes a 16-B write to the exchange slot; the row factors of element group e/4 are fetched eight
        // pieces ahead.  Gaps PW0..PW0+3 carry the partial accumulator of tile i instead.
        auto gap = [&](int g) {
            if (EXTRA && g >= PW0 && g < PW0 + 4) {
                if (chain) {
                    const int q = g - PW0;
                    exs[q * 64 + lane] = f32x4{accS[4 * q], accS[4 * q + 1], accS[4 * q + 2], accS[4 * q + 3]};
                }
                return;
            }
            const int p = g - ((EXTRA && g >= PW0 + 4) ? 4 : 0);
            if (p >= 32 || !pieces) return;
            const int e = p >> 1, eg = e >> 2;
            if ((p & 7) == 1 && eg < 3) rfq[(eg + 1) & 1] = *reinterpret_cast<const f32x4 *>(rfp + 8 * (eg + 1));
            if (MSIG) {
                if (!(p & 1)) {
                    ee = __builtin_amdgcn_exp2f(accP[e] * rfq[eg & 1][e & 3] * kfac);
                } else {
                    accP[e] = __builtin_amdgcn_rcpf(1.0f + ee);
                    if ((e & 3) == 3) exw[eg * 64 + lane] = f32x4{accP[e - 3], accP[e - 2], accP[e - 1], accP[e]};
                }
            } else if ((p & 7) == 0) {               // logits / exact logistic: unscale only, 4 values per piece
                f32x4 z;
#pragma unroll
                for (int q = 0; q < 4; ++q) z[q] = accP[4 * eg + q] * rfq[eg & 1][q] * kfac;
                exw[eg * 64 + lane] = z;
            }
        };
        if (pieces) rfq[0] = *reinterpret_cast<const f32x4 *>(rfp);
        if constexpr (chain) {
            const unsigned char *tile = L.stg0 + (i % 3) * TILE_BYTES;
            const f16x8 *lh = reinterpret_cast<const f16x8 *>(tile + RTK_PACK_HDR);
            const f16x8 *ll = lh + KS * 64;
            f16x8 fa[PF], fl[PF];                // A fragments PF k-steps ahead
#pragma unroll
            for (int p = 0; p < PF; ++p) {
                fa[p] = lh[p * 64 + lane];
                fl[p] = ll[p * 64 + lane];
            }
            int g = 0;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const f16x8 ah = fa[ks % PF], al = fl[ks % PF];
                if (ks + PF < KS) {
                    fa[ks % PF] = lh[(ks + PF) * 64 + lane];
                    fl[ks % PF] = ll[(ks + PF) * 64 + lane];
                }
                const bool sh = EXTRA && ks >= S0 && ks < S1;
                // sched_barrier(0) pins the written order MFMA / piece / MFMA / piece (see the ws kernel)
                __builtin_amdgcn_sched_barrier(0);
                accC = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, Bh[ks], ks == 0 ? zero : accC, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                gap(g++);
                __builtin_amdgcn_sched_barrier(0);
                if (sh) {
                    accS = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, Sh[ks - S0], ks == S0 ? zero : accS, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    gap(g++);
                    __builtin_amdgcn_sched_barrier(0);
                }
                accC = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, Bl[ks], accC, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                gap(g++);
                __builtin_amdgcn_sched_barrier(0);
                if (sh) {
                    accS = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, Sl[ks - S0], accS, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    gap(g++);
                    __builtin_amdgcn_sched_barrier(0);
                }
                accC = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, Bh[ks], accC, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                gap(g++);
                __builtin_amdgcn_sched_barrier(0);
                if (sh) {
                    accS = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, Sh[ks - S0], accS, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    gap(g++);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#pragma unroll
            for (int gg = NGAP; gg < GEND; ++gg) gap(gg);   // what did not fit the chain
        } else if constexpr (pieces) {           // drain: hand over the last tile
#pragma unroll
            for (int gg = 0; gg < 32; ++gg) gap(gg + ((EXTRA && gg >= PW0) ? 4 : 0));
        }
        RTK_CG_TL(0, 6);
        __syncthreads();
        RTK_CG_TL(0, 5);
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    using T = std::true_type;
    using F = std::false_type;
    iteration(P0{}, T{}, F{}, 0, accA, accB);                     // tile 0: chain only
    int i = 1;
    for (; i + 1 < cnt; i += 2) {                                 // steady state, two tiles per trip (i odd)
        iteration(P1{}, T{}, T{}, i, accB, accA);
        iteration(P0{}, T{}, T{}, i + 1, accA, accB);
    }
    if (i < cnt) {                                                // cnt even: one more chain, its result in accB
        iteration(P1{}, T{}, T{}, i, accB, accA);
        iteration(P0{}, F{}, T{}, i + 1, accA, accB);
        iteration(P1{}, F{}, F{}, i + 2, accB, accA);
    } else {                                                      // cnt odd: the last tile is in accA
        iteration(P1{}, F{}, T{}, i, accB, accA);
        iteration(P0{}, F{}, F{}, i + 1, accA, accB);
    }
}

template <int KS, int W4, int SIGMOID>
__device__ __forceinline__ void m_role(const Geo &geo, const float *__restrict__ O, const LdsMap &L, int lane, int t) {
    constexpr int S0 = s_begin(KS, W4), S1 = s_end(KS, W4), NS = S1 - S0;
    const int r = lane & 31, h = lane >> 5, c = geo.c;
    const bool tl_on = W4 == 0 && lane == 0;
    int tl_n = 0;
    for (int u = blockIdx.x; u < geo.U; u += gridDim.x) {
        int gb, n_g;
        geo.set(u, gb, n_g);
        RTK_CG_TL(0, 1);
        load_raw<KS>(O, geo.N, c, gb, n_g, L.raw, t);
        __syncthreads();                             // S1: the raw set is in LDS
        RTK_CG_TL(0, 2);
        const bool own = W4 < n_g;
        f16x8 Bh[KS], Bl[KS], Sh[NS > 0 ? NS : 1], Sl[NS > 0 ? NS : 1];
        float us_o = 0.f;
        if (own) us_o = convert_row<KS, 0, KS>(L.raw, W4 * 32 + r, c, h, Bh, Bl);
        RTK_CG_TL(0, 7);
        __syncthreads();                             // S1b: the helper waves have the fifth group's row scales
        if (n_g == NG) convert_range<KS, S0, S1>(L.raw, 4 * 32 + r, c, h, L.up5[r], Sh, Sl);
        RTK_CG_TL(0, 3);
        __syncthreads();                             // S2: query tile 0 staged, the raw region is free
        RTK_CG_TL(0, 4);
        if (!own) {
            for (int i = 0; i < geo.n_mt + 2; ++i) __syncthreads();
        } else if (n_g == NG) {
            m_sweep<KS, W4, SIGMOID, true>(L, geo.n_mt, lane, Bh, Bl, Sh, Sl, us_o, tl_on, tl_n);
        } else {
            m_sweep<KS, W4, SIGMOID, false>(L, geo.n_mt, lane, Bh, Bl, Sh, Sl, us_o, tl_on, tl_n);
        }
    }
}

// ---- H role ------------------------------------------------------------------------------------------------
template <int AUX>
__device__ __forceinline__ void store_own(const float (&pp)[16], __amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned ld4) {
    unsigned off = voff;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, pp[e]), rs, off, 0, AUX);
        off += ((e & 3) == 3) ? 5u * ld4 : ld4;
    }
}
template <int AUX>
__device__ __forceinline__ void store_five(const float (&p5)[4], __amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned ld4) {
    unsigned off = voff;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, p5[q]), rs, off, 0, AUX);
        off += ld4;
    }
}

// One LDS-DMA piece: 16 bytes per active lane, global (per-lane address) -> LDS at lds_dst + 16 * lane
// (lds_dst wave-uniform, in M0 for the instruction; M0 is the compiler's: saved and restored in the statement).
// hipcc does not count this load: the helper role waits for it with its own vmcnt (h_barrier).
__device__ __forceinline__ void dma16(const void *gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}
// the helper waves' barrier: all but the wave's N youngest vector-memory operations done (its LDS-DMA of the
// tile the M waves read next; younger stores and the DMA of the tile after stay in flight), its own LDS
// traffic done, then the workgroup barrier.  One statement with a memory clobber: hipcc neither moves memory
// operations across it nor adds a vmcnt(0) of its own (it does for __syncthreads() behind an LDS-DMA builtin).
template <int N>
__device__ __forceinline__ void h_barrier() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}

template <int KS, int SIGMOID>
__device__ __forceinline__ void h_role(const Geo &geo, const unsigned char *__restrict__ q_packed,
                                       const float *__restrict__ O, float *__restrict__ out, const LdsMap &L,
                                       int lane, int w4, int t, int nts, int tune) {
    constexpr int TILE_BYTES = tile_bytes<KS>();
    constexpr int NPIECE = (TILE_BYTES + 1023) / 1024;     // 1-KiB DMA pieces of a tile (the last one: the 128-B rest)
    constexpr int NDMA = (NPIECE + 3) / 4;                 // pieces per helper wave (wave w: pieces w, w + 4, ...)
    constexpr int DMIN = NPIECE / 4;                       // ... at least
    const int r = lane & 31, h = lane >> 5, c = geo.c, N = geo.N, B = geo.B, ht = t & 255;
    const int64_t ld_out = geo.ld_out;
    const unsigned ld4 = (unsigned)(ld_out * 4);
    const int cnt = geo.n_mt;
    const unsigned stg_lds = (unsigned)(uintptr_t)L.stg0;  // LDS byte address of ring slot 0
    // tile mt -> ring slot mt % 3, this wave's pieces
    auto stage_dma = [&](int mt) {
        const unsigned char *src = q_packed + (int64_t)mt * TILE_BYTES + lane * 16;
        const unsigned dst = stg_lds + (unsigned)(mt % 3) * TILE_BYTES;
#pragma unroll
        for (int j = 0; j < NDMA; ++j) {
            const int k = __builtin_amdgcn_readfirstlane(w4 + 4 * j);
            if (k < NPIECE - 1) {                    // (a tile is 128 B of row factors + 2*KS KiB of planes)
                dma16(src + k * 1024, __builtin_amdgcn_readfirstlane(dst + k * 1024));
            } else if (k == NPIECE - 1) {
                if (lane < (TILE_BYTES - (NPIECE - 1) * 1024) / 16)
                    dma16(src + k * 1024, __builtin_amdgcn_readfirstlane(dst + k * 1024));
            }
        }
    };
    const bool tl_on = w4 == 0 && lane == 0;
    int tl_n = 0;
    for (int u = blockIdx.x; u < geo.U; u += gridDim.x) {
        int gb, n_g;
        geo.set(u, gb, n_g);
        RTK_CG_TL(1, 1);
        // first query tile of the sweep: requested before the raw set, its L2 round trip runs beside those loads
        // (ring slot 0 lies outside the raw region; slots 1 and 2 are free once the set is converted: S2)
        stage_dma(0);
        load_raw<KS>(O, N, c, gb, n_g, L.raw, t);
        RTK_CG_TL(1, 2);
        h_barrier<0>();                              // S1
        RTK_CG_TL(1, 3);
        const bool own = w4 < n_g, five = n_g == NG;
        if (five) {
            // the fifth group's row maxima (the M waves are converting their own groups meanwhile): eight lanes
            // per row, 16-B pieces, then the power-of-two scale and its inverse for all of the row's k-ranges
            const int row = ht >> 3, sub = ht & 7;
            const float *lrow = reinterpret_cast<const float *>(L.raw) + (4 * 32 + row) * c;
            float mx = 0.f;
            for (int p4 = sub; p4 * 4 < c; p4 += 8) {
                const f32x4 x = *reinterpret_cast<const f32x4 *>(lrow + 4 * p4);
                mx = fmaxf(fmaxf(mx, fmaxf(fabsf(x[0]), fabsf(x[1]))), fmaxf(fabsf(x[2]), fabsf(x[3])));
            }
            mx = fmaxf(mx, __shfl_xor(mx, 1));
            mx = fmaxf(mx, __shfl_xor(mx, 2));
            mx = fmaxf(mx, __shfl_xor(mx, 4));
            if (sub == 0) {
                const int sh = rtk_pack_shift(mx);
                L.up5[row] = ldexpf(1.0f, sh);
                L.uso5[row] = ldexpf(1.0f, -sh);
            }
        }
        h_barrier<0>();                              // S1b
        h_barrier<0>();                              // S2
        RTK_CG_TL(1, 4);
        const int j = (gb + w4) * 32 + r;            // own group: row of O, column of out
        const unsigned voff = (own && j < N) ? (unsigned)((4 * h * ld_out + j) * 4) : 0x80000000u;
        const int j5 = (gb + 4) * 32 + r;
        const unsigned voff5 = (five && j5 < N) ? (unsigned)(((8 * w4 + 4 * h) * ld_out + j5) * 4) : 0x80000000u;
        float k5 = five ? L.uso5[r] : 0.f;
        if (SIGMOID == 2) k5 *= -1.4426950408889634f;
        if (cnt > 1) stage_dma(1);
        for (int i = 0; i < cnt + 2; ++i) {
            // The order of an iteration follows the one resource that is scarce for the helpers, the store path:
            // 84 MB of scores at the chip's HBM write rate are ~1.8k cycles of EVERY tile-step, so the stores of
            // the tile that is ready (own groups: tile i-2, in the exchange slot since the last barrier) go out
            // first thing and drain under the rest of the iteration.  Then the fifth group (tile i-1), the row
            // factors, and last the LDS-DMA of tile i+2 -- two iterations before the M waves read it (requested and
            // awaited within ONE iteration its L2 round trip was the helpers' critical path: 2.1k of 3.3k cycles,
            // the M waves waiting at the barrier); its ring slot held tile i-1, done since the last barrier.
            const bool st_own = i >= 2 && own;
            if (st_own) {
                // own groups, tile i-2: probabilities (or logits) from the exchange slot
                float pp[16];
                const f32x4 *exr = reinterpret_cast<const f32x4 *>(L.exo + (i & 1) * EX_BYTES + w4 * 4096);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 z = exr[g * 64 + lane];
#pragma unroll
                    for (int q = 0; q < 4; ++q) pp[4 * g + q] = (SIGMOID == 1) ? rtk_sigmoid(z[q]) : z[q];
                }
                const int mt = i - 2;
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                    out + (int64_t)mt * 32 * ld_out, 0, (unsigned)(min(32, B - mt * 32) * ld_out * 4), 0x00020000);
                if (tune & 16) asm volatile("" ::"v"(pp[0]), "v"(pp[5]), "v"(pp[10]), "v"(pp[15]));   // (ablation: wrong results)
                else if (nts) store_own<2>(pp, rs, voff, ld4);
                else store_own<0>(pp, rs, voff, ld4);
            }
            RTK_CG_TL(1, 11);
            // fifth group, tile i-1: rows 8*w4 + 4*h + q of the four partial accumulators, added in wave order
            const bool st5 = five && i >= 1 && i <= cnt;
            if (st5) {
                float p5[4];
                const f32x4 *pr = reinterpret_cast<const f32x4 *>(L.exp5 + ((i - 1) & 1) * EX_BYTES) + w4 * 64 + lane;
                const f32x4 rf = *reinterpret_cast<const f32x4 *>(L.hdr + ((i - 1) & 1) * 128 + (8 * w4 + 4 * h) * 4);
                const f32x4 s0 = pr[0], s1 = pr[256], s2 = pr[512], s3 = pr[768];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float z = (((s0[q] + s1[q]) + s2[q]) + s3[q]) * rf[q] * k5;
                    p5[q] = SIGMOID == 2 ? __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z))
                                         : (SIGMOID == 1 ? rtk_sigmoid(z) : z);
                }
                const int mt = i - 1;
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                    out + (int64_t)mt * 32 * ld_out, 0, (unsigned)(min(32, B - mt * 32) * ld_out * 4), 0x00020000);
                if (tune & 16) asm volatile("" ::"v"(p5[0]), "v"(p5[1]), "v"(p5[2]), "v"(p5[3]));
                else if (nts) store_five<2>(p5, rs, voff5, ld4);
                else store_five<0>(p5, rs, voff5, ld4);
            }
            RTK_CG_TL(1, 8);
            // the row factors of tile i (in LDS since the last barrier) into the slot that outlives its ring slot:
            // the M waves' logistic pieces and the fifth group above read them an iteration from now
            if (w4 == 0 && lane < RTK_PACK_HDR / 16 && i < cnt)
                reinterpret_cast<u32x4 *>(L.hdr + (i & 1) * 128)[lane] =
                    reinterpret_cast<const u32x4 *>(L.stg0 + (i % 3) * TILE_BYTES)[lane];
            RTK_CG_TL(1, 10);
            if (i + 2 < cnt && !((tune & 32) && i > 0)) stage_dma(i + 2);
            RTK_CG_TL(1, 9);
            // Tile i+1 must have landed before the M waves pass this barrier.  Its DMA was issued at the end of
            // iteration i-1; younger than it in this wave's in-order stream: the stores and the DMA of this
            // iteration.  In the steady state that count is known and those operations stay in flight; at the ends
            // of the sweep the wave drains.
            if (i >= 2 && i + 2 < cnt && !(tune & 48)) {
                if (own && five) h_barrier<20 + DMIN>();
                else if (own) h_barrier<16 + DMIN>();
                else h_barrier<DMIN>();
            } else {
                h_barrier<0>();
            }
            RTK_CG_TL(1, 5);
        }
    }
}

template <int KS, int SIGMOID>
__global__ __launch_bounds__(512, 2) void score_cg_kernel(
    const unsigned char *__restrict__ q_packed, int B, const float *__restrict__ O, int N, int c,
    float *__restrict__ out, int64_t ld_out, int U, int nts, int tune) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    LdsMap L;
    L.init<KS>(lds);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    Geo geo;
    geo.B = B; geo.N = N; geo.c = c; geo.U = U; geo.n_mt = (B + 31) / 32; geo.ld_out = ld_out;
    // wave-uniform role split (readfirstlane makes the uniformity visible to the compiler); one instantiation
    // of the M role per wave: the fifth group's k-range, and with it the gap schedule, is static
    const int uwave = __builtin_amdgcn_readfirstlane(wave);
    // (A/B knob, RTK_CG_TUNE: bits 0-1 = static priority of the helper waves, bits 2-3 = of the M waves)
    {
        const int pr = __builtin_amdgcn_readfirstlane(uwave >= 4 ? (tune & 3) : ((tune >> 2) & 3));
        if (pr == 1) __builtin_amdgcn_s_setprio(1);
        else if (pr == 2) __builtin_amdgcn_s_setprio(2);
        else if (pr == 3) __builtin_amdgcn_s_setprio(3);
    }
    if (uwave == 0) m_role<KS, 0, SIGMOID>(geo, O, L, lane, t);
    else if (uwave == 1) m_role<KS, 1, SIGMOID>(geo, O, L, lane, t);
    else if (uwave == 2) m_role<KS, 2, SIGMOID>(geo, O, L, lane, t);
    else if (uwave == 3) m_role<KS, 3, SIGMOID>(geo, O, L, lane, t);
    else h_role<KS, SIGMOID>(geo, q_packed, O, out, L, lane, uwave & 3, t, __builtin_amdgcn_readfirstlane(nts),
                             __builtin_amdgcn_readfirstlane(tune));
}

}  // namespace rtk_cg
