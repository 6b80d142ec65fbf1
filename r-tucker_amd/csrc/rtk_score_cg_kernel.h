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
//   waves 4-6  "S"  store: the eighty row-segment stores of a tile-step (own groups: tile i-2 from the exchange
//                   slots; fifth group: tile i-1, four partial accumulators added in wave order, scaled, squashed)
//                   in twenty units of four, dealt round-robin to the three waves -- all LDS reads of an iteration
//                   in one batch, then the stores, first thing after the barrier; these waves never wait for memory.
//   wave 7     "L"  load: streams the packed query tiles global -> registers -> LDS TWO tiles ahead (tile i+2 is
//                   requested in iteration i and written in iteration i+1 to the buffer the M waves have just
//                   left); its only vector-memory operations are loads, so hipcc's counted vmcnt waits are
//                   exact.  (A wave's loads, stores and LDS-DMA share one in-order counter: with loads and stores
//                   in one wave every wait for a tile also waited for the score stores issued after it.  Measured
//                   alternatives, tools/ablate/cg_dma: all four helpers loading by LDS-DMA + storing, counted
//                   waits by hand -- an LDS-DMA piece costs the issuing wave ~150 cycles here, 1.1k per tile-step.)
// One barrier per tile-step.  Columns of the four own groups are computed by the same instruction sequence as
// in the ws kernel (bit-identical scores); columns of a fifth group are the sum of four K-range chains.
// Prologue (two barriers): every M wave requests its rows of O straight into registers in B-fragment order (26 + 6..8
// loads per lane, first thing in the kernel; no pass through LDS), the four partial row maxima of the fifth group meet
// in LDS (P1), both conversions run in place (fp32 pieces -> hi / lo fp16 fragments: v_pk_mul_f32, v_cvt_pk_f16_f32,
// v_fma_mix), the helper waves stage query tile 0 meanwhile (P2): first MFMA at 5.3k cycles (10.1k with the set staged
// through LDS by all waves, the round's first form).  Drain: the M waves store the own groups of the LAST tile themselves
// while the S waves store tile cnt-2 and the fifth group, so the launch ends one tile-step earlier.
//
// LDS (KS = 13): [misc 1 KiB: row-factor ring 3 x 128 B, fifth group's column factors 128 B, partial maxima 512 B]
//                [query tile 0][query tile 1][own exchange 2 x 16 KiB][partial sums 2 x 16 KiB]      = 119 KiB
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
// (the constant 100 MHz counter beside the shader clock: the in-kernel clock is their ratio)
#define RTK_CG_TLR(role, code)                                                                                  \
    do {                                                                                                        \
        if (tl_on && tl_n < 64) {                                                                               \
            g_cg_tl[(blockIdx.x * 2 + (role)) * 64 + tl_n] =                                                    \
                ((unsigned long long)(code) << 56) | (__builtin_amdgcn_s_memrealtime() & 0x00ffffffffffffffull); \
            ++tl_n;                                                                                             \
        }                                                                                                       \
    } while (0)
#else
#define RTK_CG_TL(role, code) do { (void)tl_on; (void)tl_n; } while (0)
#define RTK_CG_TLR(role, code) do { (void)tl_on; (void)tl_n; } while (0)
#endif

constexpr int NG = 5;                       // groups per set (4 in registers + 1 split along K)
constexpr int EX_BYTES = 4 * 4 * 64 * 16;   // one exchange buffer: 4 waves x 16 accumulator regs x 64 lanes x f32
constexpr int MISC_BYTES = 1024;

template <int KS>
__host__ __device__ constexpr int tile_bytes() { return RTK_PACK_HDR + 2 * KS * 1024; }

template <int KS>
inline size_t lds_bytes(int) { return MISC_BYTES + 2 * (size_t)tile_bytes<KS>() + 4 * (size_t)EX_BYTES; }

// shared k-step range of M wave w (the fifth group's chain cut in four): ceil(KS*w/4) .. ceil(KS*(w+1)/4)
__host__ __device__ constexpr int s_begin(int KS, int w) { return (KS * w + 3) / 4; }
__host__ __device__ constexpr int s_end(int KS, int w) { return (KS * (w + 1) + 3) / 4; }

struct Geo {
    int B, N, c, U, n_mt;
    int gq, gr;              // G / U and G % U (G = ceil(N / 32) groups), from the host
    int64_t ld_out;
    // set u of U: groups [gb, gb + n_g), n_g <= NG;  gb = floor(G u / U) = gq u + floor(gr u / U).
    // (G u / U in 64 bits cost every wave two ~500-cycle divisions in front of its first load; with gr < U <= 32 768 --
    // any launch of up to 128 sets per workgroup -- the remainders fit 32 bits.)
    __device__ __forceinline__ int first_group(int u) const {
        if (U <= 32768) return gq * u + (int)((unsigned)(gr * u) / (unsigned)U);
        return (int)((((int64_t)gq * U + gr) * u) / U);
    }
    __device__ __forceinline__ void set(int u, int &gb, int &n_g) const {
        gb = first_group(u);
        n_g = first_group(u + 1) - gb;
    }
};

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// eight scaled values -> the hi and lo fp16 fragments (hi = RNE(y), lo = RNE(y - hi): y - hi is exact in fp32, so the
// fused subtract-and-round of v_fma_mix gives the bits of the two-step form).  16 instructions per k-step: the multiply
// and the first rounding on pairs (v_pk_mul_f32, gfx950's v_cvt_pk_f16_f32), one v_fma_mixlo/hi_f16 per lo element
// (asm: left to itself hipcc vectorises the subtraction into v_cvt_f32_f16 x2 + v_pk_add + v_cvt_pk, 28 per k-step).
__device__ __forceinline__ void split8(const f32x4 a, const f32x4 b, float up, f16x8 &hi, f16x8 &lo) {
    const f32x2 u2 = {up, up};
    f32x2 y[4] = {{a[0], a[1]}, {a[2], a[3]}, {b[0], b[1]}, {b[2], b[3]}};
    union { f16x2 h2[4]; f16x8 h8; unsigned u[4]; } H, Lo;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        y[j] = y[j] * u2;
        H.h2[j] = __builtin_convertvector(y[j], f16x2);
        unsigned l;
        asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(y[j][0]), "v"(H.u[j]));
        asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(y[j][1]), "v"(H.u[j]));
        Lo.u[j] = l;
    }
    hi = H.h8;
    lo = Lo.h8;
}

// The set's rows of O go from memory STRAIGHT into the registers of the wave that converts them, in B-fragment order
// (lane (r, h) of M wave w: row 32 w + r, the sixteen bytes at k = 16 ks + 8 h and the next sixteen, every k-step) --
// 26 loads per lane issued first thing in the kernel, no pass through LDS, no barrier in front of the conversion.
// (Round 4's first form staged the whole set in LDS with all 512 threads and converted from there: the 27 ds_reads of a
// lane, the LDS write pass and the barrier between them put the first MFMA at 10.1k cycles; this form: see DESIGN.md.)
// Rows past N read as zero through the buffer descriptor's range (their columns are never stored); only the LAST k-step
// can reach past column c, its pieces are masked by hand (the next row's data lies there, not the end of the buffer).
template <int KS, int K0, int K1>
__device__ __forceinline__ void load_fragments_raw(__amdgpu_buffer_rsrc_t rs, unsigned n_bytes, int row, int c, int h, f32x4 *rw) {
    // ONE address register: the k-step's 64 bytes and the second piece's 16 ride in the scalar / immediate offset.  A
    // masked piece is requested at the end of the buffer: it reads as zero, and no branch surrounds a load.
    const unsigned base = (unsigned)(row * c + 8 * h) * 4u;
    const int k_last = 16 * (KS - 1) + 8 * h;
    const unsigned base_a = (k_last + 4 <= c) ? base : n_bytes, base_b = (k_last + 8 <= c) ? base : n_bytes;
#pragma unroll
    for (int ks = K0; ks < K1; ++ks) {
        const bool last = ks + 1 == KS;
        rw[2 * (ks - K0)] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, last ? base_a : base, 64 * ks, 0));
        rw[2 * (ks - K0) + 1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, last ? base_b : base, 64 * ks + 16, 0));
    }
}

// largest magnitude of NP pieces of a lane, both halves (h = 0, 1) of the row combined
template <int NP>
__device__ __forceinline__ float row_absmax(const f32x4 *rw) {
    float mx = 0.f;
#pragma unroll
    for (int i = 0; i + 1 < NP; i += 2) {
#pragma unroll
        for (int q = 0; q < 4; ++q)          // one v_max3_f32 with |.| modifiers per pair
            asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(mx) : "v"(rw[i][q]), "v"(rw[i + 1][q]));
    }
    return fmaxf(mx, __shfl_xor(mx, 32));
}

struct LdsMap {
    unsigned char *hdr;      // 3 x 128 B: row factors of the query tiles t % 3 (they outlive the tile's buffer)
    float *uso5;             // 32 floats: unscale factors (powers of two) of the fifth group's columns
    float *pmax;             // 4 x 32 floats: the fifth group's row maxima over each M wave's k-range
    unsigned char *stg0, *stg1;   // query tile t in buffer t & 1
    unsigned char *exo;      // 2 x EX_BYTES: own accumulators (probabilities) of tile t & 1
    unsigned char *exp5;     // 2 x EX_BYTES: the four partial accumulators of the fifth group, tile t & 1
    template <int KS>
    __device__ __forceinline__ void init(unsigned char *lds) {
        hdr = lds;
        uso5 = reinterpret_cast<float *>(lds + 384);
        pmax = reinterpret_cast<float *>(lds + 512);
        stg0 = lds + MISC_BYTES;
        stg1 = stg0 + tile_bytes<KS>();
        exo = stg1 + tile_bytes<KS>();
        exp5 = exo + 2 * EX_BYTES;
    }
};

// where an M wave stores its own group's scores of the LAST query tile itself (the drain)
struct MOut {
    float *out;
    int64_t ld_out;
    int B, nts;
    unsigned voff;           // byte offset of (row 4h, this lane's column) in a tile's 32 output rows, or out of range
};

// ---- M role: the sweep over the query tiles with the set's fragments in registers --------------------------
template <int KS, int W4, int SIGMOID, bool EXTRA>
__device__ __forceinline__ void m_sweep(const LdsMap &L, int cnt, int lane, const f16x8 (&BF)[2 * KS],
                                        const f16x8 *SF, float us_o, const MOut &mo, bool tl_on, int &tl_n) {
#ifndef RTK_CG_PF
#define RTK_CG_PF 3
#endif
    constexpr int PF = KS < RTK_CG_PF ? KS : RTK_CG_PF;   // A-fragment prefetch distance (k-steps)
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
    // DIRECT (the last tile, no chain left to hide under): the probabilities stay in the accumulator and the wave
    // stores them itself -- the helper waves are storing tile cnt-2 and the fifth group meanwhile, and the launch ends
    // one tile-step earlier than with a hand-over through LDS.
    auto iteration = [&](auto par_c, auto chain_c, auto pieces_c, int i, f32x16 &accC, f32x16 &accP) {
        constexpr int PAR = decltype(par_c)::value;
        constexpr bool chain = decltype(chain_c)::value, pieces = decltype(pieces_c)::value;
        constexpr bool direct = !chain && pieces;
        f32x4 *exw = reinterpret_cast<f32x4 *>(L.exo + (PAR ^ 1) * EX_BYTES + W4 * 4096);   // own slot of tile i-1
        f32x4 *exs = reinterpret_cast<f32x4 *>(L.exp5 + PAR * EX_BYTES + W4 * 4096);        // partial slot of tile i
        const float *rfp = reinterpret_cast<const float *>(L.hdr + ((i + 2) % 3) * 128) + 4 * h;  // row factors of tile i-1
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
                    if ((e & 3) == 3 && !direct) exw[eg * 64 + lane] = f32x4{accP[e - 3], accP[e - 2], accP[e - 1], accP[e]};
                }
            } else if ((p & 7) == 0) {               // logits / exact logistic: unscale only, 4 values per piece
                f32x4 z;
#pragma unroll
                for (int q = 0; q < 4; ++q) z[q] = accP[4 * eg + q] * rfq[eg & 1][q] * kfac;
                if (direct) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) accP[4 * eg + q] = (SIGMOID == 1) ? rtk_sigmoid(z[q]) : z[q];
                } else {
                    exw[eg * 64 + lane] = z;
                }
            }
        };
        if (pieces) rfq[0] = *reinterpret_cast<const f32x4 *>(rfp);
        if constexpr (chain) {
            const unsigned char *tile = PAR ? L.stg1 : L.stg0;
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
                accC = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, BF[2 * ks], ks == 0 ? zero : accC, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                gap(g++);
                __builtin_amdgcn_sched_barrier(0);
                if (sh) {
                    accS = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, SF[2 * (ks - S0)], ks == S0 ? zero : accS, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    gap(g++);
                    __builtin_amdgcn_sched_barrier(0);
                }
                accC = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, BF[2 * ks + 1], accC, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                gap(g++);
                __builtin_amdgcn_sched_barrier(0);
                if (sh) {
                    accS = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, SF[2 * (ks - S0) + 1], accS, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    gap(g++);
                    __builtin_amdgcn_sched_barrier(0);
                }
                accC = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, BF[2 * ks], accC, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                gap(g++);
                __builtin_amdgcn_sched_barrier(0);
                if (sh) {
                    accS = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, SF[2 * (ks - S0)], accS, 0, 0, 0);
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
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                mo.out + (int64_t)(i - 1) * 32 * mo.ld_out, 0, (unsigned)(min(32, mo.B - (i - 1) * 32) * mo.ld_out * 4), 0x00020000);
            const unsigned ld4 = (unsigned)(mo.ld_out * 4);
            // element e: row (e & 3) + 8 (e >> 2) + 4h.  (Through a scalar copy: __builtin_bit_cast applied to an
            // ext-vector ELEMENT reads element 0 with this hipcc -- every store then carried accP[0].)
            auto put = [&](auto aux_c) {
                unsigned off = mo.voff;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float pe = accP[e];
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, pe), rs, off, 0, decltype(aux_c)::value);
                    off += ((e & 3) == 3) ? 5u * ld4 : ld4;
                }
            };
            if (mo.nts) put(std::integral_constant<int, 2>{});
            else put(std::integral_constant<int, 0>{});
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
    } else {                                                      // cnt odd: the last tile is in accA
        iteration(P1{}, F{}, T{}, i, accB, accA);
    }
}

template <int KS, int W4, int SIGMOID>
__device__ __forceinline__ void m_role(const Geo &geo, const float *__restrict__ O, float *__restrict__ out,
                                       const LdsMap &L, int lane, int nts) {
    constexpr int S0 = s_begin(KS, W4), S1 = s_end(KS, W4), NS = S1 - S0;
    const int r = lane & 31, h = lane >> 5, c = geo.c;
    const bool tl_on = W4 == 0 && lane == 0;
    int tl_n = 0;
    RTK_CG_TLR(0, 12);
    for (int u = blockIdx.x; u < geo.U; u += gridDim.x) {
        int gb, n_g;
        geo.set(u, gb, n_g);
        RTK_CG_TL(0, 1);
        const bool own = W4 < n_g;
        // BF[2 ks] / BF[2 ks + 1]: the raw fp32 pieces of k-step ks until the conversion, its hi / lo fp16 fragments after it
        // -- the same sixteen bytes per lane, converted IN PLACE (with the fragments in arrays of their own the allocator
        // held both generations through the conversion: ~100 spills).  SF: the same for this wave's k-range of the fifth
        // group (SF[2 j] / SF[2 j + 1] = Sh / Sl of k-step S0 + j).
        constexpr int NSX = NS > 0 ? NS : 1;
        f16x8 BF[2 * KS], SF[2 * NSX];
        float us_o = 0.f;
        {
            const int64_t row0 = (int64_t)gb * 32;
            const int valid_rows = (int)max((int64_t)0, min((int64_t)n_g * 32, (int64_t)geo.N - row0));
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float *>(O + row0 * c), 0, (unsigned)valid_rows * (unsigned)c * 4u, 0x00020000);
            const unsigned n_bytes = (unsigned)valid_rows * (unsigned)c * 4u;
            if (NS > 0) load_fragments_raw<KS, S0, S1>(rs, n_bytes, 4 * 32 + r, c, h, reinterpret_cast<f32x4 *>(SF));
            load_fragments_raw<KS, 0, KS>(rs, n_bytes, W4 * 32 + r, c, h, reinterpret_cast<f32x4 *>(BF));
            // the fifth group's rows: this wave's k-range only -- the row maximum is the largest of the four waves' (a
            // group the set does not have reads as zero).  Requested first, exchanged first: the barrier then sits under
            // the own group's loads, and both conversions run behind it without another wait.
            const float pm = NS > 0 ? row_absmax<2 * NSX>(reinterpret_cast<const f32x4 *>(SF)) : 0.f;
            if (h == 0) L.pmax[W4 * 32 + r] = pm;
            RTK_CG_TL(0, 2);
            __syncthreads();                         // P1: the four partial maxima are in LDS
            if (n_g == NG) {
                const float m5 = fmaxf(fmaxf(L.pmax[r], L.pmax[32 + r]), fmaxf(L.pmax[64 + r], L.pmax[96 + r]));
                const int sh5 = rtk_pack_shift(m5);
                const float up5 = ldexpf(1.0f, sh5);
#pragma unroll
                for (int ks = 0; ks < NS; ++ks)
                    split8(__builtin_bit_cast(f32x4, SF[2 * ks]), __builtin_bit_cast(f32x4, SF[2 * ks + 1]), up5, SF[2 * ks], SF[2 * ks + 1]);
                if (W4 == 0 && h == 0) L.uso5[r] = ldexpf(1.0f, -sh5);
            }
            RTK_CG_TL(0, 7);
            if (own) {
                const int sh = rtk_pack_shift(row_absmax<2 * KS>(reinterpret_cast<const f32x4 *>(BF)));
                const float up = ldexpf(1.0f, sh);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
                    split8(__builtin_bit_cast(f32x4, BF[2 * ks]), __builtin_bit_cast(f32x4, BF[2 * ks + 1]), up, BF[2 * ks], BF[2 * ks + 1]);
                us_o = ldexpf(1.0f, -sh);
            }
        }
        RTK_CG_TL(0, 3);
        __syncthreads();                             // P2: query tile 0 staged, the fifth group's column factors written
        RTK_CG_TL(0, 4);
        MOut mo;
        mo.out = out; mo.ld_out = geo.ld_out; mo.B = geo.B; mo.nts = nts;
        {
            const int j = (gb + W4) * 32 + r;
            mo.voff = (own && j < geo.N) ? (unsigned)(4 * h * geo.ld_out * 4) + (unsigned)j * 4u : 0x80000000u;
        }
        if (!own) {
            for (int i = 0; i < geo.n_mt + 1; ++i) __syncthreads();
        } else if (n_g == NG) {
            m_sweep<KS, W4, SIGMOID, true>(L, geo.n_mt, lane, BF, SF, us_o, mo, tl_on, tl_n);
        } else {
            m_sweep<KS, W4, SIGMOID, false>(L, geo.n_mt, lane, BF, SF, us_o, mo, tl_on, tl_n);
        }
    }
    RTK_CG_TLR(0, 13);
}

// ---- H role ------------------------------------------------------------------------------------------------
template <int AUX>
__device__ __forceinline__ void store_five(const float (&p5)[4], __amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned ld4) {
    unsigned off = voff;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, p5[q]), rs, off, 0, AUX);
        off += ld4;
    }
}

// Prologue of all four helper waves: query tile 0 (256 threads) -> buffer 0 and row-factor slot 0.
template <int KS>
__device__ __forceinline__ void h_prologue(const unsigned char *__restrict__ q_packed, const LdsMap &L, int t) {
    constexpr int TILE_BYTES = tile_bytes<KS>();
    constexpr int CHUNKS = TILE_BYTES / 16;
    constexpr int NLD = (CHUNKS + 255) / 256;
    const int ht = t & 255;
    u32x4 sreg[NLD];
    const u32x4 *src = reinterpret_cast<const u32x4 *>(q_packed);
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int ch = i * 256 + ht;
        if (i + 1 < NLD || ch < CHUNKS) sreg[i] = src[ch];
    }
    u32x4 *dst = reinterpret_cast<u32x4 *>(L.stg0);
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int ch = i * 256 + ht;
        if (i + 1 < NLD || ch < CHUNKS) dst[ch] = sreg[i];
    }
    if (ht < RTK_PACK_HDR / 16) reinterpret_cast<u32x4 *>(L.hdr)[ht] = sreg[0];
}

// "S" role: store wave SW (0, 1, 2).  The stores of a tile-step come in twenty UNITS of four row-segment stores:
// unit u < 16 = rows 8q + 4h + {0..3} (q = u & 3) of own group u >> 2 (one 16-B read of M wave u >> 2's exchange
// slot, tile i-2); unit 16 + b = the same rows (q = b) of the fifth group (four partial accumulators + the row
// factors, tile i-1).  Wave SW takes the units u = SW, SW + 3, ...: 28 / 28 / 24 stores.
template <int KS, int SIGMOID, int SW>
__device__ __forceinline__ void s_role(const Geo &geo, const unsigned char *__restrict__ q_packed,
                                       float *__restrict__ out, const LdsMap &L, int lane, int t, int nts, int tune) {
    constexpr int NU = (20 - SW + 2) / 3;        // units of this wave
    const int r = lane & 31, h = lane >> 5, N = geo.N, B = geo.B;
    const int64_t ld_out = geo.ld_out;
    const unsigned ld4 = (unsigned)(ld_out * 4);
    const int cnt = geo.n_mt;
    const bool tl_on = SW == 0 && lane == 0;
    int tl_n = 0;
    for (int u = blockIdx.x; u < geo.U; u += gridDim.x) {
        int gb, n_g;
        geo.set(u, gb, n_g);
        RTK_CG_TL(1, 1);
        h_prologue<KS>(q_packed, L, t);
        RTK_CG_TL(1, 2);
        const bool five = n_g == NG;
        __syncthreads();                             // P1
        __syncthreads();                             // P2
        RTK_CG_TL(1, 4);
        // byte offset of (row 4h, this lane's column of group g) in a tile's 32 output rows; past-the-end columns and
        // groups the set does not have get an offset the buffer range check drops
        const unsigned row4h = (unsigned)(4 * h * ld_out * 4);
        unsigned colb[5];
#pragma unroll
        for (int g = 0; g < 5; ++g) {
            const int j = (gb + g) * 32 + r;
            colb[g] = (g < n_g && j < N) ? row4h + (unsigned)j * 4u : 0x80000000u;
        }
        float k5 = five ? L.uso5[r] : 0.f;
        if (SIGMOID == 2) k5 *= -1.4426950408889634f;
        for (int i = 0; i < cnt + 1; ++i) {              // (the own groups of the last tile: stored by the M waves)
            // Everything this wave reads from LDS in an iteration is requested in one batch (one LDS round trip: a
            // few hundred cycles under the M waves' fragment reads), then the stores go out back to back -- the
            // score stores are the resource that paces a tile-step (84 MB at the chip's write rate are ~1.8k cycles
            // of each), so they start right behind the barrier and drain under the rest of the iteration.
            const bool st_own = i >= 2, st5 = five && i >= 1 && i <= cnt;
            f32x4 z[NU], p1[2], p2[2], p3[2], rf[2];
            const f32x4 *ex = reinterpret_cast<const f32x4 *>(L.exo + (i & 1) * EX_BYTES) + lane;
            const f32x4 *pr = reinterpret_cast<const f32x4 *>(L.exp5 + ((i - 1) & 1) * EX_BYTES) + lane;
            const unsigned char *rp = L.hdr + ((i + 2) % 3) * 128 + 4 * h * 4;
#pragma unroll
            for (int k = 0; k < NU; ++k) {
                constexpr int dummy = 0;
                (void)dummy;
                const int uu = SW + 3 * k;
                if (uu < 16) {
                    if (st_own) z[k] = ex[(uu >> 2) * 256 + (uu & 3) * 64];
                } else if (st5) {
                    const int b = uu - 16, kk = k - (NU - (SW == 2 ? 1 : (SW == 1 ? 2 : 1)));   // index among this wave's fifth units
                    z[k] = pr[b * 64];
                    p1[kk] = pr[256 + b * 64];
                    p2[kk] = pr[512 + b * 64];
                    p3[kk] = pr[768 + b * 64];
                    rf[kk] = *reinterpret_cast<const f32x4 *>(rp + b * 32);
                }
            }
            RTK_CG_TL(1, 11);
            const __amdgpu_buffer_rsrc_t rs_own = __builtin_amdgcn_make_buffer_rsrc(
                out + (int64_t)(i - 2) * 32 * ld_out, 0, (unsigned)(min(32, B - (i - 2) * 32) * ld_out * 4), 0x00020000);
            const __amdgpu_buffer_rsrc_t rs5 = __builtin_amdgcn_make_buffer_rsrc(
                out + (int64_t)(i - 1) * 32 * ld_out, 0, (unsigned)(min(32, B - (i - 1) * 32) * ld_out * 4), 0x00020000);
#pragma unroll
            for (int k = 0; k < NU; ++k) {
                const int uu = SW + 3 * k;
                float p[4];
                if (uu < 16) {
                    if (!st_own) continue;
#pragma unroll
                    for (int q = 0; q < 4; ++q) p[q] = (SIGMOID == 1) ? rtk_sigmoid(z[k][q]) : z[k][q];
                    const unsigned voff = colb[uu >> 2] + (unsigned)(uu & 3) * 8u * ld4;
                    if (tune & 16) asm volatile("" ::"v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]));   // (ablation: no score stores)
                    else if (nts) store_five<2>(p, rs_own, voff, ld4);
                    else store_five<0>(p, rs_own, voff, ld4);
                } else {
                    if (!st5) continue;
                    const int b = uu - 16, kk = k - (NU - (SW == 2 ? 1 : (SW == 1 ? 2 : 1)));
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float zz = (((z[k][q] + p1[kk][q]) + p2[kk][q]) + p3[kk][q]) * rf[kk][q] * k5;
                        p[q] = SIGMOID == 2 ? __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(zz))
                                            : (SIGMOID == 1 ? rtk_sigmoid(zz) : zz);
                    }
                    const unsigned voff = colb[4] + (unsigned)b * 8u * ld4;
                    if (tune & 16) asm volatile("" ::"v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]));
                    else if (nts) store_five<2>(p, rs5, voff, ld4);
                    else store_five<0>(p, rs5, voff, ld4);
                }
            }
            RTK_CG_TL(1, 6);
            __syncthreads();
            RTK_CG_TL(1, 5);
        }
    }
}

// "L" role: the load wave
template <int KS>
__device__ __forceinline__ void l_role(const Geo &geo, const unsigned char *__restrict__ q_packed,
                                       const LdsMap &L, int lane, int t, int tune) {
    constexpr int TILE_BYTES = tile_bytes<KS>();
    constexpr int CHUNKS = TILE_BYTES / 16;
    constexpr int NLD = (CHUNKS + 63) / 64;     // staging 16-B chunks per lane
    const int cnt = geo.n_mt;
    u32x4 sreg[NLD];
    // (through a buffer descriptor: ONE address register, the piece and tile offsets are scalar; a piece past the
    // last tile reads as zero and is never written to LDS)
    const __amdgpu_buffer_rsrc_t rsq = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char *>(q_packed), 0, (unsigned)(cnt * TILE_BYTES), 0x00020000);
    auto stage_load = [&](int mt) {
#pragma unroll
        for (int i = 0; i < NLD; ++i)
            sreg[i] = __builtin_amdgcn_raw_buffer_load_b128(rsq, (unsigned)lane * 16u, mt * TILE_BYTES + i * 1024, 0);
    };
    // tile mt -> buffer mt & 1; its header (32 row factors) also into slot mt % 3 of the ring that outlives the
    // buffer (the M waves' logistic pieces and the fifth group read it two iterations later)
    auto stage_store = [&](int mt) {
        u32x4 *dst = reinterpret_cast<u32x4 *>((mt & 1) ? L.stg1 : L.stg0);
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int ch = i * 64 + lane;
            if (i + 1 < NLD || ch < CHUNKS) dst[ch] = sreg[i];
        }
        if (lane < RTK_PACK_HDR / 16) reinterpret_cast<u32x4 *>(L.hdr + (mt % 3) * 128)[lane] = sreg[0];
    };
    for (int u = blockIdx.x; u < geo.U; u += gridDim.x) {
        int gb, n_g;
        geo.set(u, gb, n_g);
        h_prologue<KS>(q_packed, L, t);
        if (cnt > 1 && !(tune & 32)) stage_load(1);  // in flight across the prologue's barriers
        __syncthreads();                             // P1
        __syncthreads();                             // P2
        for (int i = 0; i < cnt + 1; ++i) {
            // tile i+1 was requested an iteration ago (tile 1: in the prologue); buffer (i+1) & 1 held tile i-1, which
            // the M waves left at the last barrier.  Tile i+2 is requested right behind the write, into the same
            // registers: its L2 round trip has a whole tile-step.
            if (i + 1 < cnt && !(tune & 32)) stage_store(i + 1);
            if (i + 2 < cnt && !(tune & 32)) stage_load(i + 2);
            __syncthreads();
        }
    }
}

template <int KS, int SIGMOID>
__global__ __launch_bounds__(512, 2) void score_cg_kernel(
    const unsigned char *__restrict__ q_packed, int B, const float *__restrict__ O, int N, int c,
    float *__restrict__ out, int64_t ld_out, int U, int gq, int gr, int nts, int tune) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    LdsMap L;
    L.init<KS>(lds);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    Geo geo;
    geo.B = B; geo.N = N; geo.c = c; geo.U = U; geo.gq = gq; geo.gr = gr; geo.n_mt = (B + 31) / 32; geo.ld_out = ld_out;
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
#ifdef RTK_CG_STAMPS
    // (slot 63 of a role's list: s_memtime in the kernel's common entry code, in front of the branch to the role)
    if ((uwave == 0 || uwave == 4) && lane == 0)
        g_cg_tl[(blockIdx.x * 2 + (uwave >> 2)) * 64 + 63] = __builtin_amdgcn_s_memtime();
#endif
    if (uwave == 0) m_role<KS, 0, SIGMOID>(geo, O, out, L, lane, __builtin_amdgcn_readfirstlane(nts));
    else if (uwave == 1) m_role<KS, 1, SIGMOID>(geo, O, out, L, lane, __builtin_amdgcn_readfirstlane(nts));
    else if (uwave == 2) m_role<KS, 2, SIGMOID>(geo, O, out, L, lane, __builtin_amdgcn_readfirstlane(nts));
    else if (uwave == 3) m_role<KS, 3, SIGMOID>(geo, O, out, L, lane, __builtin_amdgcn_readfirstlane(nts));
    else if (uwave == 4) s_role<KS, SIGMOID, 0>(geo, q_packed, out, L, lane, t, __builtin_amdgcn_readfirstlane(nts), __builtin_amdgcn_readfirstlane(tune));
    else if (uwave == 5) s_role<KS, SIGMOID, 1>(geo, q_packed, out, L, lane, t, __builtin_amdgcn_readfirstlane(nts), __builtin_amdgcn_readfirstlane(tune));
    else if (uwave == 6) s_role<KS, SIGMOID, 2>(geo, q_packed, out, L, lane, t, __builtin_amdgcn_readfirstlane(nts), __builtin_amdgcn_readfirstlane(tune));
    else l_role<KS>(geo, q_packed, L, lane, t, __builtin_amdgcn_readfirstlane(tune));
}

}  // namespace rtk_cg
