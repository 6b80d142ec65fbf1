// Split-fp16 score kernel, wave-specialised persistent form (c <= 208, i.e. KS <= 13).
//
//   out[d, j] = logistic( v[d,:] . O[j,:] )          reference: asymmetric/R_TuckER.py:47-48
//
// One 512-thread workgroup per CU, resident for the whole launch, two waves per SIMD with
// different jobs (the matrix pipe and the VALU/VMEM pipes are separate, so a wave that only
// issues MFMAs and a wave that only issues vector/memory work overlap on one SIMD for free):
//
//   waves 0-3  "M"  own 32 entity columns each: B fragments (hi/lo fp16 of their O rows) stay
//                   in registers; per query tile they read the A fragments from LDS, issue the
//                   3*KS dependent MFMAs, and hand the unscaled 32x32 fp32 result to a helper
//                   through an LDS exchange slot (one tile behind, inside the MFMA gaps).
//   waves 4-7  "H"  do everything else: stream the packed query tiles global -> registers ->
//                   LDS two tiles ahead, apply the logistic and store the scores of the tile
//                   the M waves finished two iterations ago (branch-free buffer stores, 128-B
//                   row segments), and prefetch the NEXT 128-row tile of O (fully coalesced
//                   16-B loads, all in flight) so that switching entity tiles costs one LDS
//                   round trip instead of an exposed HBM latency.
//
// LDS: [2 x query tile (staging, double buffered)] [raw O tile 128 x c fp32; during a sweep the
// same region holds the 2 x 16 KiB accumulator exchange slots].  One barrier per iteration.
// Work split: the linearised (entity tile x query tile) space is cut evenly over the grid.
#pragma once
#include "rtk_common.h"
#include "rtk_pack.h"

namespace rtk_ws {

// tools/ablate only: cycle stamps (STAMP template flag).  The 131 KB stamp array exists only in the
// ablation build (-DRTK_ABLATE_STAMPS); in the product library every use below is dead code (STAMP = false).
#ifdef RTK_ABLATE_STAMPS
__device__ unsigned long long g_ws_stamps[256 * 8 * 8];
__device__ unsigned long long g_ws_tl[256 * 2 * 64];   // timeline: [workgroup][M wave 0 / H wave 0][event] = code << 56 | s_memtime
#else
static constexpr unsigned long long *g_ws_stamps = nullptr;
static constexpr unsigned long long *g_ws_tl = nullptr;
#endif
// timeline event of wave 0 of a role (STAMP builds only): `n` is the role's running event index
#define RTK_TL(role, code)                                                                                        \
    do {                                                                                                          \
        if (STAMP && w4 == 0 && lane == 0 && tl_n < 64) {                                                         \
            g_ws_tl[(blockIdx.x * 2 + (role)) * 64 + tl_n] =                                                      \
                ((unsigned long long)(code) << 56) | (__builtin_amdgcn_s_memtime() & 0x00ffffffffffffffull);      \
            ++tl_n;                                                                                               \
        }                                                                                                         \
    } while (0)

constexpr int EX_BYTES = 4 * 4 * 64 * 16;  // one exchange buffer: 4 M waves x 16 accumulator regs x 64 lanes x f32

template <int KS>
__host__ __device__ constexpr int tile_bytes() { return RTK_PACK_HDR + 2 * KS * 1024; }

// dynamic LDS the kernel needs for entity rank c
template <int KS>
inline size_t lds_bytes(int c) {
    const size_t oreg = (size_t)128 * c * 4;
    return 2 * (size_t)tile_bytes<KS>() + (oreg > 2 * (size_t)EX_BYTES ? oreg : 2 * (size_t)EX_BYTES);
}

// Both roles walk the same (entity tile, query tile) schedule and meet at the same barriers;
// they are separate functions so each gets its own register allocation (the M waves keep
// 2*KS B fragments alive, the H waves a whole raw O tile).
struct Sched {
    // Work of one workgroup: first `base` WHOLE entity tiles (all query tiles each), then an even
    // share of the remainder tiles' (entity tile, query tile) units.  With T tiles on W workgroups
    // that is floor(T/W) full sweeps plus one partial sweep: at C2 (320 tiles, 256 CUs) every
    // workgroup converts exactly 2 O tiles (an even cut of the whole linearised space gives 2.25 on
    // average and up to 3, i.e. more O re-fetch and a longer critical path).
    int B, N, c, n_mt;
    int ta, ta_end;            // whole tiles [ta, ta_end)
    int rem_tile0;             // first remainder tile
    int lin, lin_end;          // remainder units, linearised (tile - rem_tile0) * n_mt + query tile
    __device__ __forceinline__ void init(int B_, int N_, int c_, int w, int W, int xcd_remap) {
        B = B_; N = N_; c = c_;
        // Workgroups are dealt round-robin to the 8 XCDs (w % 8 says which share one).  The remainder
        // tiles are each swept by several NEIGHBOURING schedule slots: make neighbours share an XCD, so the
        // tile is fetched from HBM once and served to the others by that XCD's L2 (speed only: any
        // placement gives the same scores).  xcd_remap: 0 = off, 1 = both phases, 2 = remainder phase only.
        const int wx = (xcd_remap && (W & 7) == 0) ? (w & 7) * (W >> 3) + (w >> 3) : w;
        const int ww = xcd_remap == 1 ? wx : w;
        n_mt = (B + 31) / 32;
        const int T = (N + 127) / 128;
        const int base = T / W;
        ta = ww * base;
        ta_end = ta + base;
        rem_tile0 = base * W;
        const int64_t Ur = (int64_t)(T - rem_tile0) * n_mt;
        lin = (int)(Ur * wx / W);
        lin_end = (int)(Ur * (wx + 1) / W);
    }
    __device__ __forceinline__ bool peek(int &tile) const {
        if (ta < ta_end) { tile = ta; return true; }
        if (lin < lin_end) { tile = rem_tile0 + lin / n_mt; return true; }
        return false;
    }
    __device__ __forceinline__ bool next(int &ntile, int &mt0, int &cnt, bool &more, int &next_tile) {
        if (ta < ta_end) {
            ntile = ta++;
            mt0 = 0;
            cnt = n_mt;
        } else if (lin < lin_end) {
            ntile = rem_tile0 + lin / n_mt;
            mt0 = lin % n_mt;
            cnt = min(n_mt - mt0, lin_end - lin);
            lin += cnt;
        } else {
            return false;
        }
        more = peek(next_tile);
        return true;
    }
};

template <int KS, bool STAMP, unsigned XP, bool o_vec, int SIGMOID>
__device__ __forceinline__ void m_role(Sched sc, const unsigned char *__restrict__ q_packed, unsigned char *stg,
                                       unsigned char *oreg, int lane, int w4, int ht) {
    constexpr int TILE_BYTES = tile_bytes<KS>();
    constexpr int PF = KS < 3 ? KS : 3;         // A-fragment prefetch distance (k-steps): LDS latency > one k-step of MFMAs
    if (XP & 1) __builtin_amdgcn_s_setprio(3);
    const int r = lane & 31, h = lane >> 5, c = sc.c;
    int ntile, mt0, cnt, next_tile;
    bool more;
    int tl_n = 0;
    RTK_TL(0, 1);
    while (sc.next(ntile, mt0, cnt, more, next_tile)) {
        __syncthreads();                             // S1: raw O tile visible in LDS
        RTK_TL(0, 2);
        f16x8 Bh[KS], Bl[KS];
        float us_o;
        static_assert(o_vec, "the launcher only builds the float4 form (c % 4 == 0, aligned O); other shapes run the v3 kernel");
        {
            // One pass over LDS: this wave's 32 rows (c floats each, 2*KS float4 per lane) are read once,
            // all reads in flight together and branch-free (an out-of-row float4 is read from column 0
            // and zeroed by a select), kept in registers for the row maximum and converted from there.
            // (Two passes -- maximum, then conversion -- read the tile twice: the four waves then move
            // 2 x 100 KB through LDS, ~1600 of the tile switch's ~1900 cycles.)
            // (the row offset goes through an empty asm: otherwise the 2*KS loop-invariant fragment addresses
            // are hoisted out of the sweep loop and stay live across the MFMA chains -- 26 VGPRs, spills)
            int row_off = (w4 * 32 + r) * c, h8 = 8 * h;
            asm volatile("" : "+v"(row_off), "+v"(h8));
            const float *lrow = reinterpret_cast<const float *>(oreg) + row_off;
            f32x4 raw[2 * KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int k = 16 * ks + h8;              // k = 16*ks + 8*h + q  (B-operand map of 32x32x16)
                raw[2 * ks] = *reinterpret_cast<const f32x4 *>(lrow + ((k + 4 <= c) ? k : 0));
                raw[2 * ks + 1] = *reinterpret_cast<const f32x4 *>(lrow + ((k + 8 <= c) ? k + 4 : 0));
            }
            float mx = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int k = 16 * ks + h8;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (!(k + 4 <= c)) raw[2 * ks][q] = 0.f;
                    if (!(k + 8 <= c)) raw[2 * ks + 1][q] = 0.f;
                    mx = fmaxf(mx, fmaxf(fabsf(raw[2 * ks][q]), fabsf(raw[2 * ks + 1][q])));
                }
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const int sh = rtk_pack_shift(mx);
            const float up = ldexpf(1.0f, sh);
            us_o = ldexpf(1.0f, -sh);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float y0 = raw[2 * ks][q] * up, y1 = raw[2 * ks + 1][q] * up;
                    const _Float16 h0 = (_Float16)y0, h1 = (_Float16)y1;
                    Bh[ks][q] = h0;
                    Bh[ks][4 + q] = h1;
                    Bl[ks][q] = (_Float16)(y0 - (float)h0);
                    Bl[ks][4 + q] = (_Float16)(y1 - (float)h1);
                }
            }
        }
        RTK_TL(0, 3);
        __syncthreads();                             // S2: tile mt0 staged, O region free for the exchange
        RTK_TL(0, 4);

        // Fast logistic (SIGMOID == 2) is evaluated HERE, in the shadow of the MFMAs: per value one
        // multiply + v_exp_f32, then one add + v_rcp_f32 (both 1 ulp), a 20-cycle piece in each of the
        // first 32 MFMA gaps of the next tile; the helper waves then only store.  (With the logistic in
        // the helpers they, not the MFMA chain, set the pace: 44.2 us for logits vs 47.8 us.)
        constexpr bool MSIG = SIGMOID == 2;
        const float kfac = MSIG ? us_o * -1.4426950408889634f : us_o;
        float ee = 0.f;                              // 2^(-z log2 e) of the value in flight
        // Tile-pipelined form.  Two accumulators that ALTERNATE BETWEEN TILES (not between MFMAs): while the
        // chain of tile i runs on one of them (a single dependent chain issues a 32x32x16 MFMA every ~35
        // cycles, tools/ubench/mfma_gap.hip), the gaps turn the OTHER one -- tile i-1, complete since the last
        // barrier -- into probabilities in place.  Per tile-step this removes the 32 v_mov that zeroed two
        // accumulators (the first MFMA takes C = 0), the 16 v_add that merged them and the copy into `prev`,
        // and the row factors are fetched at the start of their own tile's chain instead of behind it; what
        // is left between two barriers besides the chain is the latency of the first fragment reads.
        f32x16 accA, accB;
        f32x4 svA[4], svB[4];
#pragma unroll
        for (int e = 0; e < 16; ++e) accB[e] = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) svB[g] = f32x4{0.f, 0.f, 0.f, 0.f};
        auto iteration = [&](int i, f32x16 &accC, f32x4 (&svC)[4], f32x16 &accP, f32x4 (&svP)[4]) {
            f32x4 *exw = reinterpret_cast<f32x4 *>(oreg + ((i + 1) & 1) * EX_BYTES + w4 * 4096);   // slot of tile i-1
            const bool chain = i < cnt;
            // gap j: pieces 0..31 hand tile i-1 over (accP -> probabilities -> exchange slot); 32..35 scale
            // the row factors of tile i (read from its header at the top of this iteration) by kfac
            auto gap = [&](int j) {
                if (j >= 32) {
                    if (j < 36 && chain) svC[j - 32] = svC[j - 32] * kfac;
                    return;
                }
                if (XP & 4) {                             // (ablation) no logistic: raw accumulators out, once
                    if (j == 31) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) exw[g * 64 + lane] = f32x4{accP[4 * g], accP[4 * g + 1], accP[4 * g + 2], accP[4 * g + 3]};
                    }
                    return;
                }
                const int e = j >> 1;
                if (MSIG) {
                    if (!(j & 1)) {
                        ee = __builtin_amdgcn_exp2f(accP[e] * svP[e >> 2][e & 3]);
                    } else {
                        accP[e] = __builtin_amdgcn_rcpf(1.0f + ee);
                        if ((e & 3) == 3) exw[(e >> 2) * 64 + lane] = f32x4{accP[e - 3], accP[e - 2], accP[e - 1], accP[e]};
                    }
                } else if ((j & 7) == 0) {           // logits / exact logistic: unscale only, 4 values per piece
                    const int g = j >> 3;
                    f32x4 z;
#pragma unroll
                    for (int q = 0; q < 4; ++q) z[q] = accP[4 * g + q] * svP[g][q];
                    exw[g * 64 + lane] = z;
                }
            };
            if (chain) {
                const unsigned char *tile = stg + (i & 1) * TILE_BYTES;
                const f16x8 *lh = reinterpret_cast<const f16x8 *>(tile + RTK_PACK_HDR);
                const f16x8 *ll = lh + KS * 64;
                f16x8 fa[PF], fl[PF];                // A fragments PF k-steps ahead
#pragma unroll
                for (int p = 0; p < PF; ++p) {
                    fa[p] = lh[p * 64 + lane];
                    fl[p] = ll[p * 64 + lane];
                }
                const float *lscale = reinterpret_cast<const float *>(tile);
#pragma unroll
                for (int g = 0; g < 4; ++g) svC[g] = *reinterpret_cast<const f32x4 *>(lscale + 8 * g + 4 * h);
                const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const f16x8 ah = fa[ks % PF], al = fl[ks % PF];
                    if (ks + PF < KS && !(XP & 8)) {          // (XP & 8, ablation: the chain re-uses its first fragments)
                        fa[ks % PF] = lh[(ks + PF) * 64 + lane];
                        fl[ks % PF] = ll[(ks + PF) * 64 + lane];
                    }
                    // sched_barrier(0) pins the written order: the scheduler otherwise sinks every fragment read
                    // next to its MFMA (one LDS latency per k-step) and hoists a gap's piece above its MFMA (two
                    // dependent transcendental pieces between one pair of MFMAs, none between the next pair)
                    __builtin_amdgcn_sched_barrier(0);
                    accC = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, Bh[ks], ks == 0 ? zero : accC, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    gap(3 * ks);
                    __builtin_amdgcn_sched_barrier(0);
                    accC = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, Bl[ks], accC, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    gap(3 * ks + 1);
                    __builtin_amdgcn_sched_barrier(0);
                    accC = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, Bh[ks], accC, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    gap(3 * ks + 2);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int j = 3 * KS; j < 36; ++j) gap(j);   // short chains: the rest of the hand-over, the row factors
            } else if (i == cnt) {                   // drain: hand over the last tile
#pragma unroll
                for (int j = 0; j < 32; ++j) gap(j);
            }
            RTK_TL(0, 6);
            __syncthreads();
            RTK_TL(0, 5);
        };
        for (int i = 0; i < cnt + 2; i += 2) {
            iteration(i, accA, svA, accB, svB);
            if (i + 1 < cnt + 2) iteration(i + 1, accB, svB, accA, svA);
        }
    }
    RTK_TL(0, 9);
    if (STAMP && lane == 0) g_ws_stamps[(blockIdx.x * 8 + w4) * 8 + 7] = __builtin_amdgcn_s_memrealtime();
}

#ifndef RTK_WS_LATE_STORES
#define RTK_WS_LATE_STORES 1        // 0: round 3's earlier order (stores between the staging loads and the staging write)
#endif
// one iteration's score stores / O prefetch of a helper wave (h_role; see the comment at their use)
#define RTK_WS_SCORES_OUT \
            if (i >= 2 && !(XP & 2) && !(XP & 16)) { \
                const int mt = mt0 + i - 2; \
                const unsigned char *slot = oreg + (i & 1) * EX_BYTES + w4 * 4096; \
                const int rows = min(32, B - mt * 32); \
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc( \
                    out + (int64_t)mt * 32 * ld_out, 0, (unsigned)(rows * ld_out * 4), 0x00020000); \
                { \
                    const f32x4 *exr = reinterpret_cast<const f32x4 *>(slot); \
                    float zz[16], pp[16]; \
_Pragma("unroll") \
                    for (int g = 0; g < 4; ++g) { \
                        const f32x4 z = exr[g * 64 + lane]; \
_Pragma("unroll") \
                        for (int q = 0; q < 4; ++q) zz[4 * g + q] = z[q]; \
                    } \
_Pragma("unroll") \
                    for (int e = 0; e < 16; ++e) pp[e] = (SIGMOID == 1) ? rtk_sigmoid(zz[e]) : zz[e]; \
                    unsigned off = voff; \
                    if (nts) { \
_Pragma("unroll") \
                        for (int e = 0; e < 16; ++e) { \
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, pp[e]), rs, off, 0, 2); \
                            off += ((e & 3) == 3) ? 5u * ld4 : ld4; \
                        } \
                    } else { \
_Pragma("unroll") \
                        for (int e = 0; e < 16; ++e) { \
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, pp[e]), rs, off, 0, 0); \
                            off += ((e & 3) == 3) ? 5u * ld4 : ld4; \
                        } \
                    } \
                } \
            }
#define RTK_WS_PREFETCH_O \
            if (i < pf_iters) { \
_Pragma("unroll") \
                for (int p = 0; p < (NOR + PFI - 1) / PFI; ++p) \
                    if (p == i) load_oraw_part(next_tile, p * PFI, min(NOR, (p + 1) * PFI)); \
            }

// (A/B'd and rejected, round 2: two staging + two storing helper waves, so that the storing waves never wait on
// vmcnt -- 43.3 us against 42.2 us; two query tiles in flight in the helpers' registers -- 40.0 against 39.4 us;
// 16-byte score stores from a transposed read of the exchange slot -- 42.1 against 40.4 us.)
template <int KS, int SIGMOID, bool STAMP, unsigned XP>
__device__ __forceinline__ void h_role(Sched sc, const unsigned char *__restrict__ q_packed,
                                       const float *__restrict__ O, float *__restrict__ out, int64_t ld_out,
                                       unsigned char *stg, unsigned char *oreg, int lane, int w4, int ht, int nts) {
    constexpr int TILE_BYTES = tile_bytes<KS>();
    constexpr int CHUNKS = TILE_BYTES / 16;
    constexpr int NLD = (CHUNKS + 255) / 256;   // staging 16-B chunks per helper thread
    constexpr int NOR = 2 * KS;                 // raw O 16-B pieces per helper thread (32*c/256 <= 2*KS)
    const int r = lane & 31, h = lane >> 5, c = sc.c, N = sc.N, B = sc.B;
    u32x4 oraw[NOR];
    // pieces [i0, i1) of this lane's share of an O tile (128 rows = 32*c pieces of 16 B, contiguous in memory,
    // c % 4 == 0); i0 / i1 are compile-time after unrolling
    auto load_oraw_part = [&](int ntile, int i0, int i1) {
        const int64_t row0 = (int64_t)ntile * 128;
        const int valid = (int)max((int64_t)0, min((int64_t)128, (int64_t)N - row0)) * c;  // floats of real rows
        const float *src = O + row0 * c;
#pragma unroll
        for (int i = 0; i < NOR; ++i) {
            if (i < i0 || i >= i1) continue;
            const int pc = i * 256 + ht;
            u32x4 x = {0u, 0u, 0u, 0u};
            if (4 * pc + 4 <= valid) x = *reinterpret_cast<const u32x4 *>(src + 4 * pc);   // valid % 4 == 0
            oraw[i] = x;
        }
    };
    // (through a buffer descriptor -- one voffset register, no per-piece compare -- the same loads made the
    // kernel 3 us SLOWER on the same box: 43.2 vs 40.2 us)
    auto load_oraw = [&](int ntile) { load_oraw_part(ntile, 0, NOR); };
    u32x4 sreg[NLD];
    auto stage_load = [&](int mt) {
        const u32x4 *src = reinterpret_cast<const u32x4 *>(q_packed + (int64_t)mt * TILE_BYTES);
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int ch = i * 256 + ht;
            if (i + 1 < NLD || ch < CHUNKS) sreg[i] = src[ch];
        }
    };
    auto stage_store = [&](int buf) {
        u32x4 *dst = reinterpret_cast<u32x4 *>(stg + buf * TILE_BYTES);
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int ch = i * 256 + ht;
            if (i + 1 < NLD || ch < CHUNKS) dst[ch] = sreg[i];
        }
    };

    if (STAMP && lane == 0) g_ws_stamps[(blockIdx.x * 8 + 4 + w4) * 8 + 5] = __builtin_amdgcn_s_memrealtime();
    int tl_n = 0;
    RTK_TL(1, 1);
    {
        int first_tile = 0;
        if (sc.peek(first_tile)) load_oraw(first_tile);
    }
    const unsigned ld4 = (unsigned)(ld_out * 4);
    int ntile, mt0, cnt, next_tile;
    bool more;
    while (sc.next(ntile, mt0, cnt, more, next_tile)) {
        const int j = ntile * 128 + w4 * 32 + r;     // entity: row of O, column of out
        // first query tile of the sweep: requested BEFORE the O tile goes to LDS, so its L2 round trip
        // runs beside that write and the S1 wait instead of after them (it took longer than the M
        // waves' conversion and held S2 back)
        stage_load(mt0);
#pragma unroll
        for (int i = 0; i < NOR; ++i) {
            const int pc = i * 256 + ht;
            if (pc < 32 * c) reinterpret_cast<u32x4 *>(oreg)[pc] = oraw[i];
        }
        RTK_TL(1, 2);
        __syncthreads();                             // S1
        RTK_TL(1, 3);
        stage_store(0);
        __syncthreads();                             // S2
        RTK_TL(1, 4);
        // The NEXT O tile stays in registers for the whole sweep.  Its 2*KS loads per lane are NOT issued in one
        // go: a wave's vector-memory instructions queue in order, so the staging loads and score stores of the
        // first iterations sat behind 100 KB of prefetch per CU (~8k cycles, with the MFMA waves waiting at S2
        // or at the first barrier all that time).  They trickle out instead, PFI per iteration, behind that
        // iteration's own loads and stores; a short sweep issues the rest before its first iteration.
        constexpr int PFI = 4;
        const int pf_iters = more ? min(cnt, (NOR + PFI - 1) / PFI) : 0;
        if (more && pf_iters * PFI < NOR) load_oraw_part(next_tile, pf_iters * PFI, NOR);
        if (STAMP && lane == 0 && g_ws_stamps[(blockIdx.x * 8 + 4 + w4) * 8 + 6] == 0) g_ws_stamps[(blockIdx.x * 8 + 4 + w4) * 8 + 6] = __builtin_amdgcn_s_memrealtime();
        const unsigned voff = (j < N) ? (unsigned)((4 * h * ld_out + j) * 4) : 0x80000000u;
        for (int i = 0; i < cnt + 2; ++i) {
            const bool st = STAMP && i == 5 && lane == 0;
            unsigned long long *sp = g_ws_stamps + (blockIdx.x * 8 + 4 + w4) * 8;
            if (st) sp[0] = __builtin_amdgcn_s_memtime();
            // Query tile i+1: its loads go out first -- in the in-order vmcnt stream they are OLDER than
            // this iteration's score stores, so the wait at the bottom only covers stores issued a whole
            // iteration ago -- and its LDS writes come last.
            const bool stage = (i + 1 < cnt) && !(XP & 16);       // (XP & 16, ablation: helpers idle)
            if (stage) stage_load(mt0 + i + 1);
            // Where the score stores sit in the wave's in-order vector-memory stream decides who waits for their
            // acknowledgements: hipcc's wait in front of stage_store's LDS writes is vmcnt(<= 5) -- every younger operation
            // but a few, i.e. THIS iteration's sixteen stores when they are issued between the staging loads and that
            // wait (the counter is in order; the compiler's bound is not tight: tools/isa_waits.py prints them).
            // RTK_WS_LATE_STORES (default) issues them after stage_store, right before the barrier: the next wait that
            // covers them is a whole iteration away (same box: 46.9 against 48.2 us per step, kernel 40.1 against 41.6).  The block is a macro so that both orders compile to
            // the same code apart from its position (as lambdas the same statements cost 19 VGPRs and 268 B of scratch).
            // scores of tile i-2 (logistic + stores; nts: nontemporal, 128-B aligned rows, the cache-policy bits are an
            // immediate, hence two copies of the loop), then this iteration's share of the next O tile
            if constexpr (!RTK_WS_LATE_STORES) {
                RTK_WS_SCORES_OUT
                RTK_WS_PREFETCH_O
            }
            if (st) sp[1] = __builtin_amdgcn_s_memtime();
            RTK_TL(1, 6);
            if (stage) stage_store((i + 1) & 1);
            if constexpr (RTK_WS_LATE_STORES != 0) {     // (the prefetch first: hipcc drains vmcnt in front of its loads)
                RTK_WS_PREFETCH_O
                RTK_WS_SCORES_OUT
            }
            if (st) sp[2] = sp[3] = __builtin_amdgcn_s_memtime();
            RTK_TL(1, 7);
            __syncthreads();
            RTK_TL(1, 5);
            if (st) sp[4] = __builtin_amdgcn_s_memtime();
        }
    }
    RTK_TL(1, 9);
    if (STAMP && lane == 0) g_ws_stamps[(blockIdx.x * 8 + 4 + w4) * 8 + 7] = __builtin_amdgcn_s_memrealtime();
}

// O_VEC: c % 4 == 0 and O 16-B aligned (compile-time so the scalar fallback's address arithmetic
// is not hoisted into -- and spilled by -- the vector build)
template <int KS, int SIGMOID, bool O_VEC, bool STAMP = false, unsigned XP = 0>
__global__ __launch_bounds__(512, 2) void score_ws_kernel(
    const unsigned char *__restrict__ q_packed, int B, const float *__restrict__ O, int N, int c,
    float *__restrict__ out, int64_t ld_out, int xcd_remap, int nts) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char *const stg = lds;
    unsigned char *const oreg = lds + 2 * tile_bytes<KS>();
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    Sched sc;
    sc.init(B, N, c, blockIdx.x, gridDim.x, xcd_remap);
    // wave-uniform role split (readfirstlane makes the uniformity visible to the compiler)
    const int uwave = __builtin_amdgcn_readfirstlane(wave);     // in an SGPR: role tests are scalar branches
    if (uwave < 4) m_role<KS, STAMP, XP, O_VEC, SIGMOID>(sc, q_packed, stg, oreg, lane, uwave & 3, t & 255);
    else h_role<KS, SIGMOID, STAMP, XP>(sc, q_packed, O, out, ld_out, stg, oreg, lane, uwave & 3, t & 255,
                                        __builtin_amdgcn_readfirstlane(nts));
}

}  // namespace rtk_ws
