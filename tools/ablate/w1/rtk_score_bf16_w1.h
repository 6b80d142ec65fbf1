// Stage 2 for bf16 operands, deep-K form (256 < c <= 512: the 1 M-entity shard, BASELINE.json configs[4]):
//     out[d, j] = logistic( v[d,:] . O[j,:] )        (reference: src/model/asymmetric/R_TuckER.py:47-48)
//
// ONE wave per SIMD (256-thread workgroup, one per CU, 512 registers per lane), each wave stationary on 64 entity
// rows (two 32-row B blocks = 256 registers of bf16 fragments), query tiles of 32 streamed through a three-slot
// LDS ring.  Why this shape (round 2's kernel: 8 waves x 32 rows, two waves per SIMD, 1.15 ms = 0.46 of the HBM
// roofline; SQ counters: matrix pipe 57 % busy, VALU 37 %, both at once 21 %, waves parked 31 %):
//   * two waves of a SIMD share its matrix pipe AND its vector issue (MI355X_MICROARCH.md, "Two waves per SIMD"):
//     moving work between them is zero-sum, and every barrier re-synchronises them into the same phase.  One wave
//     that owns the SIMD issues MFMA / epilogue piece / MFMA / piece with no arbitration: 64 MFMAs per tile
//     (2048 cycles of matrix pipe) and 64 gaps of 24 free issue cycles for 32 values x (exp piece, rcp + store
//     piece) -- 12-20 cycles per gap.
//   * each A fragment read from LDS feeds TWO MFMAs: half the LDS read traffic per score.
//   * the chain never drains: the tile staged in registers during chain i-1 is written to slot (i+1) % 3 in the
//     first gaps of chain i, the workgroup's only barrier sits right behind those writes (early in the chain: the
//     waves are a few hundred cycles into 2048 and still in step), the loads for tile i+2 follow, and the first
//     fragments of tile i+1 are prefetched in the last gaps of chain i.  (Round 2: store, barrier, fragment reads
//     between two chains = a bubble of one LDS round trip + barrier skew per tile.)
//   * no accumulator merge (one dependent chain per B block: same MFMA rate, MI355X_MICROARCH.md), accumulators
//     swap roles between tiles (no copies).
//   * TRANSPOSED product: the entity fragments are the MFMA's A operand and the query tile its B operand, so a
//     lane of the 32 x 32 result holds ONE query (column) and 16 entities in four runs of 4 consecutive ones =
//     four 16-byte stores into that query's score row (rows start on 128-byte boundaries: the four runs of the two
//     lane halves make one full line) instead of sixteen 4-byte stores, and ONE store offset per lane (the runs are
//     immediate offsets).  Only the matrix's last, ragged entity tile takes 4-byte stores.
#pragma once
#include "rtk_common.h"
#include "rtk_pack.h"
#include <type_traits>

namespace rtk_w1 {

// TR: the transposed product with 16-byte stores described above.  TR = false: queries are the A operand, a lane
// holds one entity column and 16 query rows, sixteen 4-byte stores per tile and block, each instruction two full
// 128-byte row segments (offsets precomputed once per unit).
template <int KS, int SIGMOID, bool NTS, bool TR, unsigned ABL = 0>     // ABL: tools/ubench/w1_bench.hip ablations only
__global__ __launch_bounds__(256, 1) void score_bf16_w1_kernel(
    const unsigned char *__restrict__ q_packed, int B, const rtk_bf16 *__restrict__ O, int N, int c,
    float *__restrict__ out, int64_t ld_out, bool o_vec, int QB) {
    constexpr int TILE_BYTES = RTK_PACK_HDR + KS * 1024;   // in global memory
    constexpr int SLOT = KS * 1024;                        // in LDS: the plane only
    constexpr int NLD = KS * 64 / 256;                     // 16-byte staging loads per thread (KS % 4 == 0)
    static_assert(KS % 4 == 0 && KS >= 8, "deep-K form: whole staging loads per thread");
    constexpr int PF = 4;                                  // A fragments in flight
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // 3 * SLOT

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int n_mt = (B + 31) / 32, n_nt = (N + 255) / 256;
    const __amdgpu_buffer_rsrc_t qrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char *>(q_packed), 0, (unsigned)(n_mt * TILE_BYTES), 0x00020000);
    const unsigned ld4 = (unsigned)(ld_out * 4);

    for (int qb0 = 0; qb0 < n_mt; qb0 += QB) {
    const int tq = min(QB, n_mt - qb0);
    const int64_t U = (int64_t)n_nt * tq;
    int64_t lin = U * blockIdx.x / gridDim.x;
    const int64_t lin_end = U * (blockIdx.x + 1) / gridDim.x;
    while (lin < lin_end) {
        const int ntile = (int)(lin / tq), mt0 = qb0 + (int)(lin % tq);
        const int cnt = (int)min((int64_t)(qb0 + tq - mt0), lin_end - lin);
        lin += cnt;
        const int jb = ntile * 256 + wave * 64 + r;        // entity row of this lane in block 0 (block 1: + 32)

        u32x4 stg[NLD];
        auto stage_load_one = [&](int mt, int i) {
            stg[i] = __builtin_amdgcn_raw_buffer_load_b128(qrs, (unsigned)(t * 16), mt * TILE_BYTES + RTK_PACK_HDR + i * 4096, 0);
        };
        auto stage_load = [&](int mt) {
#pragma unroll
            for (int i = 0; i < NLD; ++i) stage_load_one(mt, i);
        };
        auto stage_store_one = [&](int slot, int i) {
            reinterpret_cast<u32x4 *>(lds + slot * SLOT)[i * 256 + t] = stg[i];
        };

        __syncthreads();            // the previous unit's slots are no longer read
        stage_load(mt0);
        // B fragments: lane (r, h) holds k = 16 ks + 8 h + q, q < 8, of its row = 16 contiguous bytes
        bf16x8 Bf[2][KS];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            const rtk_bf16 *orow = O + (int64_t)min(jb + 32 * rb, N - 1) * c;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int k = 16 * ks + 8 * h;
                bf16x8 x = {0, 0, 0, 0, 0, 0, 0, 0};
                if (o_vec) {
                    if (k + 8 <= c) x = *reinterpret_cast<const bf16x8 *>(orow + k);
                } else {
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        if (k + q < c) x[q] = (short)orow[k + q];
                }
                Bf[rb][ks] = x;
            }
        }
        // Store offset of this lane relative to the first row of a query tile: the lane's query is row r, its
        // entities are ntile * 256 + wave * 64 + 32 rb + 8 g + 4 h + {0..3} for block rb, run g.  Rows past B are
        // cut by the descriptor's size; columns past N only exist in the last entity tile (slow path below).
        const int ecol = ntile * 256 + wave * 64 + 4 * h;              // first entity of run (rb = 0, g = 0)
        const unsigned voff = (unsigned)(r * ld4 + ecol * 4);
        const bool full = ntile * 256 + 256 <= N;                       // uniform: every column of this tile exists
        // TR = false: value e of block rb is row 8 (e / 4) + 4 h + e % 4 of the tile, column = the lane's entity
        unsigned voffs[TR ? 1 : 2][TR ? 1 : 16];
        if (!TR) {
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = 8 * (e >> 2) + 4 * h + (e & 3);
                    voffs[TR ? 0 : rb][TR ? 0 : e] = (jb + 32 * rb < N) ? (unsigned)(row * ld4 + (jb + 32 * rb) * 4) : 0x80000000u;
                }
        }

#pragma unroll
        for (int i = 0; i < NLD; ++i) stage_store_one(0, i);
        if (cnt > 1) stage_load(mt0 + 1);
        __syncthreads();

        __amdgpu_buffer_rsrc_t ers;
        auto epilogue_begin = [&](int mt) {
            const int rows = min(32, B - mt * 32);
            ers = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<unsigned char *>(out) + (int64_t)mt * 32 * ld_out * 4, 0,
                                                    (unsigned)(rows * ld_out * 4), 0x00020000);
        };
        float ep_d = 1.f;
        // piece pc of the 64 of a finished tile: value (rb = pc / 32, e = (pc % 32) / 2): exp half, then reciprocal
        // half (the probability replaces the logit in its accumulator register) and, behind every fourth value, the
        // 16-byte store of the run
        auto piece = [&](f32x16 &z0, f32x16 &z1, int pc) {
            const int rb = pc >> 5, e = (pc & 31) >> 1;
            f32x16 &zz = rb ? z1 : z0;
            if ((pc & 1) == 0) {
                if (SIGMOID == 2) ep_d = __builtin_amdgcn_exp2f(zz[e] * -1.4426950408889634f);
                else if (SIGMOID == 1) ep_d = 1.0f + expf(-zz[e]);
                else ep_d = zz[e];
            } else {
                float pv = ep_d;
                if (SIGMOID == 2) pv = __builtin_amdgcn_rcpf(1.0f + ep_d);
                if (SIGMOID == 1) pv = 1.0f / ep_d;
                zz[e] = pv;
                if (ABL & 1) {                                    // (ablation) no stores: keep the value alive
                    if (pv == 12345.678f) out[0] = pv;
                } else if (!TR) {
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, pv), ers, voffs[TR ? 0 : rb][TR ? 0 : e], 0, NTS ? 2 : 0);
                } else if ((e & 3) == 3) {
                    const int g = e >> 2;
                    const unsigned o = voff + (unsigned)(rb * 128 + g * 32);
                    // (element copies first: __builtin_bit_cast applied to an ext-vector ELEMENT lvalue reads element 0,
                    // clang 19 / ROCm 7.2)
                    const float p0 = zz[4 * g], p1 = zz[4 * g + 1], p2 = zz[4 * g + 2], p3 = zz[4 * g + 3];
                    if (full) {
                        const u32x4 q = {__builtin_bit_cast(unsigned, p0), __builtin_bit_cast(unsigned, p1),
                                         __builtin_bit_cast(unsigned, p2), __builtin_bit_cast(unsigned, p3)};
                        __builtin_amdgcn_raw_buffer_store_b128(q, ers, o, 0, NTS ? 2 : 0);
                    } else {
                        const float pq[4] = {p0, p1, p2, p3};
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const bool in = ecol + rb * 32 + g * 8 + q < N;
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, pq[q]), ers,
                                                                  in ? o + 4u * q : 0x80000000u, 0, 0);
                        }
                    }
                }
            }
        };

        f32x16 accA0, accA1, accB0, accB1;
#pragma unroll
        for (int e = 0; e < 16; ++e) accB0[e] = accB1[e] = 0.f;
        bf16x8 fa[PF];
        {
            const bf16x8 *la = reinterpret_cast<const bf16x8 *>(lds);
#pragma unroll
            for (int p = 0; p < PF; ++p) fa[p] = la[p * 64 + lane];
        }
        // one chain: tile i accumulates in (x0, x1) while the finished tile i - 1 in (y0, y1) is turned into
        // probabilities and stored
        auto chain = [&](auto first_tag, int i, f32x16 &x0, f32x16 &x1, f32x16 &y0, f32x16 &y1) {
            constexpr bool FIRST = decltype(first_tag)::value;     // the unit's first tile: nothing to store yet
            const int cur = i % 3, nxt = (i + 1) % 3;
            const bool more = i + 1 < cnt, more2 = i + 2 < cnt;
            const bf16x8 *la = reinterpret_cast<const bf16x8 *>(lds + cur * SLOT);
            const bf16x8 *ln = reinterpret_cast<const bf16x8 *>(lds + nxt * SLOT);
            if (!FIRST) epilogue_begin(mt0 + i - 1);
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            // The MFMAs are written in assembly with the entity fragments constrained to the ACCUMULATION half of the
            // register file ("a"): hipcc parks half of the 256 fragment registers there anyway, but then copies every
            // fragment back (4 v_accvgpr_read per MFMA) although the matrix pipe reads A/B operands from either half
            // -- with the copies and the LDS addresses it also spilled there, the wave was VALU-issue-bound at twice
            // the MFMA time (1.29 ms for raw logits).  Accumulators, query fragments and everything else stay in VGPRs.
            auto mma = [&](f32x16 &acc, const bf16x8 &q, const bf16x8 &ent, bool first) {
                if (TR) {
                    if (first) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc) : "a"(ent), "v"(q));
                    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(ent), "v"(q));
                } else {
                    if (first) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(q), "a"(ent));
                    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(q), "a"(ent));
                }
            };
            // 64 pieces over the gaps behind the MFMAs -- except the first gap: the pieces read accumulators whose last
            // MFMA (written in assembly: the compiler's hazard recogniser does not see it) issued two MFMAs earlier
            constexpr int GAPS = 2 * KS - 1;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 a = fa[ks % PF];
                // next fragment: of this tile, or (last PF gaps, behind the barrier) the first ones of the next
                if (!(ABL & 2)) {                                 // (ablation bit 1: the chain re-uses its first fragments)
                    if (ks + PF < KS) fa[ks % PF] = la[(ks + PF) * 64 + lane];
                    else if (more) fa[ks % PF] = ln[(ks + PF - KS) * 64 + lane];
                }
                mma(x0, a, Bf[0][ks], ks == 0);
                if (!FIRST) {
#pragma unroll
                    for (int pc = (ks == 0 ? 0 : (2 * ks - 1) * 64 / GAPS); pc < (ks == 0 ? 0 : (2 * ks) * 64 / GAPS); ++pc) piece(y0, y1, pc);
                }
                // the staged tile i + 1 goes to its slot in the first NLD gaps; then the barrier; then the loads of i + 2
                if (ks < NLD && more && !(ABL & 4)) stage_store_one(nxt, ks);
                __builtin_amdgcn_sched_barrier(0);
                mma(x1, a, Bf[1][ks], ks == 0);
                if (!FIRST) {
#pragma unroll
                    for (int pc = (2 * ks) * 64 / GAPS; pc < (2 * ks + 1) * 64 / GAPS; ++pc) piece(y0, y1, pc);
                }
                if (ks == NLD && !(ABL & 8)) __syncthreads();
                if (ks > NLD && ks <= 2 * NLD && more2 && !(ABL & 4)) stage_load_one(mt0 + i + 2, ks - NLD - 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        typedef std::integral_constant<bool, true> first_t;
        typedef std::integral_constant<bool, false> later_t;
        chain(first_t{}, 0, accA0, accA1, accB0, accB1);
        int i = 1;
        for (; i + 1 < cnt; i += 2) {
            chain(later_t{}, i, accB0, accB1, accA0, accA1);
            chain(later_t{}, i + 1, accA0, accA1, accB0, accB1);
        }
        auto drain_mfma = [&]() {       // the last MFMAs of the unit (assembly) must have written their results
#pragma unroll
            for (int n = 0; n < 5; ++n) asm volatile("s_nop 7");
        };
        if (i < cnt) {
            chain(later_t{}, i, accB0, accB1, accA0, accA1);
            drain_mfma();
            epilogue_begin(mt0 + cnt - 1);
#pragma unroll
            for (int pc = 0; pc < 64; ++pc) piece(accB0, accB1, pc);
        } else {
            drain_mfma();
            epilogue_begin(mt0 + cnt - 1);
#pragma unroll
            for (int pc = 0; pc < 64; ++pc) piece(accA0, accA1, pc);
        }
    }
    }
}

}  // namespace rtk_w1
