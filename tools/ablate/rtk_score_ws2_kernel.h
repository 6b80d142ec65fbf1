// Split-fp16 score kernel, wave-specialised persistent form, TWO query tiles per barrier.
//
//   out[d, j] = logistic( v[d,:] . O[j,:] )          reference: asymmetric/R_TuckER.py:47-48
//
// Same roles as rtk_score_ws_kernel.h (waves 0-3 "M": register-resident hi/lo B fragments of 32
// entity columns + MFMAs; waves 4-7 "H": query-tile staging, logistic + stores, O-tile prefetch),
// but an iteration covers a PAIR of 32-query tiles: the fixed per-barrier costs (LDS latencies,
// barrier skew, exchange) are paid once per 64 queries, and the two tiles' MFMA chains are
// independent, so they interleave at the two-chain issue rate (~35 instead of ~40 cycles per
// 32x32x16 MFMA, tools/ubench/mfma_rate.hip).
//
// LDS (c = 200: 152 KiB):  [pair buffer 0] [ O region: raw O tile 128 x c fp32 at a tile switch;
// during a sweep: pair buffer 1 | exchange slots (4 waves x 2 tiles x 4 KiB) | 8 flag words ].
// The exchange is single-buffered: an H wave copies its slot to registers right after the
// barrier and raises flag[w]; the M wave checks the flag (already up: it has a whole MFMA chain
// behind it) before overwriting the slot at the end of the iteration.
#pragma once
#include "rtk_common.h"
#include "rtk_pack.h"

namespace rtk_ws2 {

constexpr int EXB = 4 * 2 * 4096;   // exchange: 4 M waves x 2 tiles x (16 regs x 64 lanes x f32)

// tools/ablate only (STAMP template flag): per wave a trace of (cycle counter << 8 | event id)
constexpr int STAMPS_PER_WAVE = 96;
__device__ unsigned long long g_ws2_stamps[256 * 8 * STAMPS_PER_WAVE];
template <bool STAMP>
struct Trace {
    unsigned long long *p;
    int n;
    __device__ __forceinline__ void init(int wave, int lane) {
        if (STAMP) { p = g_ws2_stamps + (blockIdx.x * 8 + wave) * STAMPS_PER_WAVE; n = (lane == 0) ? 0 : STAMPS_PER_WAVE; }
    }
    __device__ __forceinline__ void operator()(unsigned id) {
        if (STAMP) {
            __builtin_amdgcn_sched_barrier(0);
            if (n < STAMPS_PER_WAVE) p[n++] = (__builtin_amdgcn_s_memtime() << 8) | id;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
};

template <int KS>
__host__ __device__ constexpr int tile_bytes() { return RTK_PACK_HDR + 2 * KS * 1024; }

template <int KS>
inline size_t lds_bytes(int c) {
    const size_t tb2 = 2 * (size_t)tile_bytes<KS>();
    const size_t oreg = (size_t)128 * c * 4, sweep = tb2 + EXB + 32;
    return tb2 + (oreg > sweep ? oreg : sweep);
}

struct Sched {   // whole entity tiles first, then an even share of the remainder (see rtk_score_ws_kernel.h)
    int B, N, c, n_mt, n_pt;   // n_pt: pairs of query tiles
    int ta, ta_end, rem_tile0, lin, lin_end;
    __device__ __forceinline__ void init(int B_, int N_, int c_, int w, int W) {
        B = B_; N = N_; c = c_;
        n_mt = (B + 31) / 32;
        n_pt = (n_mt + 1) / 2;
        const int T = (N + 127) / 128;
        const int base = T / W;
        ta = w * base;
        ta_end = ta + base;
        rem_tile0 = base * W;
        const int64_t Ur = (int64_t)(T - rem_tile0) * n_pt;
        lin = (int)(Ur * w / W);
        lin_end = (int)(Ur * (w + 1) / W);
    }
    __device__ __forceinline__ bool peek(int &tile) const {
        if (ta < ta_end) { tile = ta; return true; }
        if (lin < lin_end) { tile = rem_tile0 + lin / n_pt; return true; }
        return false;
    }
    __device__ __forceinline__ bool next(int &ntile, int &pt0, int &cnt, bool &more, int &next_tile) {
        if (ta < ta_end) {
            ntile = ta++;
            pt0 = 0;
            cnt = n_pt;
        } else if (lin < lin_end) {
            ntile = rem_tile0 + lin / n_pt;
            pt0 = lin % n_pt;
            cnt = min(n_pt - pt0, lin_end - lin);
            lin += cnt;
        } else {
            return false;
        }
        more = peek(next_tile);
        return true;
    }
};

// ------------------------------------------------------------------ M role
template <int KS, bool STAMP>
__device__ __forceinline__ void m_role(Sched sc, unsigned char *lds, int lane, int w4) {
    Trace<STAMP> tr;
    tr.init(w4, lane);
    constexpr int TB = tile_bytes<KS>();
    unsigned char *const stg0 = lds, *const oreg = lds + 2 * TB;
    unsigned char *const stg1 = oreg, *const exb = oreg + 2 * TB;
    volatile int *const flag = reinterpret_cast<volatile int *>(oreg + 2 * TB + EXB);
    const int r = lane & 31, h = lane >> 5, c = sc.c;
    int ntile, pt0, cnt, next_tile;
    bool more;
    while (sc.next(ntile, pt0, cnt, more, next_tile)) {
        tr(1);
        __syncthreads();                             // S1: raw O tile visible in LDS
        tr(2);
        f16x8 Bh[KS], Bl[KS];
        float us_o = 1.f;
        {   // fragments of this wave's 32 rows: batches of 4 k-steps read before use, two passes
            const float *lrow = reinterpret_cast<const float *>(oreg) + (w4 * 32 + r) * c;
            constexpr int CB = 4;
            float mx = 0.f;
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                float up = 1.f;
                if (pass == 1) {
                    mx = fmaxf(mx, __shfl_xor(mx, 32));
                    const int sh = rtk_pack_shift(mx);
                    up = ldexpf(1.0f, sh);
                    us_o = ldexpf(1.0f, -sh);
                }
#pragma unroll
                for (int ks0 = 0; ks0 < KS; ks0 += CB) {
                    f32x4 ta_[CB], tb_[CB];
#pragma unroll
                    for (int u = 0; u < CB; ++u) {
                        const int k = 16 * (ks0 + u) + 8 * h;   // k = 16*ks + 8*h + q  (B-operand map of 32x32x16)
                        if (ks0 + u < KS) {
                            ta_[u] = *reinterpret_cast<const f32x4 *>(lrow + ((k + 4 <= c) ? k : 0));
                            tb_[u] = *reinterpret_cast<const f32x4 *>(lrow + ((k + 8 <= c) ? k + 4 : 0));
                        }
                    }
#pragma unroll
                    for (int u = 0; u < CB; ++u) {
                        const int ks = ks0 + u, k = 16 * ks + 8 * h;
                        if (ks < KS) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const float x0 = (k + 4 <= c) ? ta_[u][q] : 0.f, x1 = (k + 8 <= c) ? tb_[u][q] : 0.f;
                                if (pass == 0) {
                                    mx = fmaxf(mx, fmaxf(fabsf(x0), fabsf(x1)));
                                } else {
                                    const float y0 = x0 * up, y1 = x1 * up;
                                    const _Float16 h0 = (_Float16)y0, h1 = (_Float16)y1;
                                    Bh[ks][q] = h0;
                                    Bh[ks][4 + q] = h1;
                                    Bl[ks][q] = (_Float16)(y0 - (float)h0);
                                    Bl[ks][4 + q] = (_Float16)(y1 - (float)h1);
                                }
                            }
                        }
                    }
                }
            }
        }
        tr(3);
        __syncthreads();                             // S2: first pair staged, O region free (pair buffer 1, exchange)

        for (int i = 0; i <= cnt; ++i) {
            tr(4);
            if (i < cnt) {
                const unsigned char *pair = (i & 1) ? stg1 : stg0;
                const f16x8 *ah_ = reinterpret_cast<const f16x8 *>(pair + RTK_PACK_HDR), *al_ = ah_ + KS * 64;
                const f16x8 *bh_ = reinterpret_cast<const f16x8 *>(pair + TB + RTK_PACK_HDR), *bl_ = bh_ + KS * 64;
                f32x16 accA, accB;
#pragma unroll
                for (int e = 0; e < 16; ++e) accA[e] = accB[e] = 0.f;
                constexpr int PF = KS < 2 ? KS : 2;  // fragments PF k-steps ahead (LDS latency > one k-step)
                f16x8 fah[PF], fal[PF], fbh[PF], fbl[PF];
#pragma unroll
                for (int p = 0; p < PF; ++p) {
                    fah[p] = ah_[p * 64 + lane];
                    fal[p] = al_[p * 64 + lane];
                    fbh[p] = bh_[p * 64 + lane];
                    fbl[p] = bl_[p * 64 + lane];
                }
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const f16x8 a_h = fah[ks % PF], a_l = fal[ks % PF], b_h = fbh[ks % PF], b_l = fbl[ks % PF];
                    if (ks + PF < KS) {
                        fah[ks % PF] = ah_[(ks + PF) * 64 + lane];
                        fal[ks % PF] = al_[(ks + PF) * 64 + lane];
                        fbh[ks % PF] = bh_[(ks + PF) * 64 + lane];
                        fbl[ks % PF] = bl_[(ks + PF) * 64 + lane];
                    }
                    // order pinned: the scheduler otherwise sinks each read next to its MFMA
                    __builtin_amdgcn_sched_barrier(0);
                    accA = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_h, Bh[ks], accA, 0, 0, 0);
                    accB = __builtin_amdgcn_mfma_f32_32x32x16_f16(b_h, Bh[ks], accB, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    accA = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_h, Bl[ks], accA, 0, 0, 0);
                    accB = __builtin_amdgcn_mfma_f32_32x32x16_f16(b_h, Bl[ks], accB, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    accA = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_l, Bh[ks], accA, 0, 0, 0);
                    accB = __builtin_amdgcn_mfma_f32_32x32x16_f16(b_l, Bh[ks], accB, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                tr(5);
                // unscale (row factors from the two tile headers) and hand both tiles to the helper
                const float *sa = reinterpret_cast<const float *>(pair), *sb = reinterpret_cast<const float *>(pair + TB);
                f32x4 za[4], zb[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 va = *reinterpret_cast<const f32x4 *>(sa + 8 * g + 4 * h);
                    const f32x4 vb = *reinterpret_cast<const f32x4 *>(sb + 8 * g + 4 * h);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        za[g][q] = accA[4 * g + q] * va[q] * us_o;
                        zb[g][q] = accB[4 * g + q] * vb[q] * us_o;
                    }
                }
                // the helper has copied pair i-1 out of the slot halves (long since: it does so right
                // after the barrier resp. one epilogue later, this wave has a whole chain behind it)
                f32x4 *ex = reinterpret_cast<f32x4 *>(exb + w4 * 8192);
                if (i > 0)
                    while (flag[2 * w4] < i) {}
#pragma unroll
                for (int g = 0; g < 4; ++g) ex[g * 64 + lane] = za[g];
                if (i > 0)
                    while (flag[2 * w4 + 1] < i) {}
#pragma unroll
                for (int g = 0; g < 4; ++g) ex[256 + g * 64 + lane] = zb[g];
                tr(6);
            }
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------ H role
template <int KS, int SIGMOID, bool STAMP>
__device__ __forceinline__ void h_role(Sched sc, const unsigned char *__restrict__ q_packed,
                                       const float *__restrict__ O, float *__restrict__ out, int64_t ld_out,
                                       unsigned char *lds, int lane, int w4, int ht) {
    constexpr int TB = tile_bytes<KS>();
    constexpr int CHUNKS2 = 2 * TB / 16;               // a pair of tiles is contiguous in q_packed
    constexpr int NLD = (CHUNKS2 + 255) / 256;
    constexpr int NOR = 2 * KS;                        // raw O 16-B pieces per helper thread (32*c/256 <= 2*KS)
    unsigned char *const stg0 = lds, *const oreg = lds + 2 * TB;
    unsigned char *const stg1 = oreg, *const exb = oreg + 2 * TB;
    volatile int *const flag = reinterpret_cast<volatile int *>(oreg + 2 * TB + EXB);
    const int c = sc.c, N = sc.N, B = sc.B;
    Trace<STAMP> tr;
    tr.init(4 + w4, lane);
    u32x4 oraw[NOR];
    auto load_oraw = [&](int ntile) {   // 128 rows = 32*c pieces of 16 B, contiguous in memory (c % 4 == 0)
        const int64_t row0 = (int64_t)ntile * 128;
        const int valid = (int)max((int64_t)0, min((int64_t)128, (int64_t)N - row0)) * c;
        const float *src = O + row0 * c;
#pragma unroll
        for (int i = 0; i < NOR; ++i) {
            const int pc = i * 256 + ht;
            u32x4 x = {0u, 0u, 0u, 0u};
            if (4 * pc + 4 <= valid) x = *reinterpret_cast<const u32x4 *>(src + 4 * pc);
            oraw[i] = x;
        }
    };
    const __amdgpu_buffer_rsrc_t qrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char *>(q_packed), 0, (unsigned)(sc.n_mt * TB), 0x00020000);
    constexpr int NH = (NLD + 1) / 2;                  // staged in two halves: half the registers in flight
    u32x4 sreg[NH];
    auto stage_load = [&](int pt, int half) {   // pair pt = tiles 2*pt, 2*pt+1; bytes past the last tile read as 0
#pragma unroll
        for (int i = 0; i < NH; ++i) {
            const int ci = half * NH + i;
            const unsigned vo = (ci * 256 + 255 < CHUNKS2 || ci * 256 + ht < CHUNKS2) ? (unsigned)(ht * 16) : 0x80000000u;
            sreg[i] = __builtin_amdgcn_raw_buffer_load_b128(qrs, vo, pt * 2 * TB + ci * 4096, 0);
        }
    };
    auto stage_store = [&](unsigned char *dstb, int half) {
        u32x4 *dst = reinterpret_cast<u32x4 *>(dstb);
#pragma unroll
        for (int i = 0; i < NH; ++i) {
            const int ch = (half * NH + i) * 256 + ht;
            if ((half * NH + i) * 256 + 255 < CHUNKS2 || ch < CHUNKS2) dst[ch] = sreg[i];
        }
    };
    {
        int first_tile = 0;
        if (sc.peek(first_tile)) load_oraw(first_tile);
    }
    const unsigned ld4 = (unsigned)(ld_out * 4);
    int ntile, pt0, cnt, next_tile;
    bool more;
    while (sc.next(ntile, pt0, cnt, more, next_tile)) {
#pragma unroll
        for (int i = 0; i < NOR; ++i) {
            const int pc = i * 256 + ht;
            if (pc < 32 * c) reinterpret_cast<u32x4 *>(oreg)[pc] = oraw[i];
        }
        tr(1);
        __syncthreads();                             // S1
        tr(2);
        stage_load(pt0, 0);                          // first pair of the sweep (the M waves are converting)
        stage_store(stg0, 0);
        stage_load(pt0, 1);
        stage_store(stg0, 1);
        if (more) load_oraw(next_tile);              // stays in registers for the whole sweep
        tr(3);
        __syncthreads();                             // S2: the M waves are done with the raw tile
        if (lane < 2) flag[2 * w4 + lane] = 0;       // (flag words live in the O region)
        const int h = lane >> 5, j = ntile * 128 + w4 * 32 + (lane & 31);   // entity: row of O, column of out
        const unsigned voff = (j < N) ? (unsigned)((4 * h * ld_out + j) * 4) : 0x80000000u;
        for (int i = 0; i <= cnt; ++i) {
            // next pair: loads first (older than this iteration's score stores in the in-order vmcnt
            // stream), LDS writes last
            const bool stage = i + 1 < cnt;
            unsigned char *const nbuf = ((i + 1) & 1) ? stg1 : stg0;
            tr(4);
            if (stage) stage_load(pt0 + i + 1, 0);
            tr(5);
            if (i >= 1) {
#pragma unroll
                for (int tsel = 0; tsel < 2; ++tsel) {
                    const int mt = 2 * (pt0 + i - 1) + tsel;
                    const int rows = max(0, min(32, B - mt * 32));
                    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                        out + (int64_t)mt * 32 * ld_out, 0, (unsigned)(rows * ld_out * 4), 0x00020000);
                    // copy this tile's half of the wave's slot to registers, then release that half
                    const f32x4 *ex = reinterpret_cast<const f32x4 *>(exb + w4 * 8192 + tsel * 4096);
                    f32x4 zq[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) zq[g] = ex[g * 64 + lane];
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (lane == 0) flag[2 * w4 + tsel] = i;
                    if (tsel == 1 && stage) {        // first half landed behind tile 0's epilogue: swap halves
                        stage_store(nbuf, 0);
                        stage_load(pt0 + i + 1, 1);
                    }
                    tr(6 + 3 * tsel);
                    float zz[16], dd[16], pp[16];
#pragma unroll
                    for (int it = 0; it < 4; ++it)
#pragma unroll
                        for (int q = 0; q < 4; ++q) zz[4 * it + q] = zq[it][q];
                    if (SIGMOID == 2) {   // stage by stage over 16 values (element by element serialises)
#pragma unroll
                        for (int e = 0; e < 16; ++e) zz[e] = fminf(zz[e] * -1.4426950408889634f, 126.0f);
#pragma unroll
                        for (int e = 0; e < 16; ++e) dd[e] = 1.0f + __builtin_amdgcn_exp2f(zz[e]);
#pragma unroll
                        for (int e = 0; e < 16; ++e) pp[e] = __builtin_amdgcn_rcpf(dd[e]);
#pragma unroll
                        for (int e = 0; e < 16; ++e) pp[e] = fmaf(pp[e], fmaf(-dd[e], pp[e], 1.0f), pp[e]);
                    } else {
#pragma unroll
                        for (int e = 0; e < 16; ++e) pp[e] = (SIGMOID == 1) ? rtk_sigmoid(zz[e]) : zz[e];
                    }
                    tr(7 + 3 * tsel);
                    unsigned off = voff;
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, pp[e]), rs, off, 0, 0);
                        off += ((e & 3) == 3) ? 5u * ld4 : ld4;   // rows 0,1,2,3,8,9,10,11,16,...
                    }
                    tr(8 + 3 * tsel);
                }
            }
            if (stage) {
                if (i == 0) {
                    stage_store(nbuf, 0);
                    stage_load(pt0 + i + 1, 1);
                }
                stage_store(nbuf, 1);
            }
            tr(12);
            __syncthreads();
        }
    }
}

template <int KS, int SIGMOID, bool STAMP = false>
__global__ __launch_bounds__(512, 2) void score_ws2_kernel(
    const unsigned char *__restrict__ q_packed, int B, const float *__restrict__ O, int N, int c,
    float *__restrict__ out, int64_t ld_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    Sched sc;
    sc.init(B, N, c, blockIdx.x, gridDim.x);
    // wave-uniform role split (readfirstlane makes the uniformity visible to the compiler)
    if (__builtin_amdgcn_readfirstlane(wave) < 4) m_role<KS, STAMP>(sc, lds, lane, wave & 3);
    else h_role<KS, SIGMOID, STAMP>(sc, q_packed, O, out, ld_out, lds, lane, wave & 3, t & 255);
}

}  // namespace rtk_ws2
