// Stage 2 for bf16 operands:  out[d, j] = logistic( v[d,:] . O[j,:] ),  v and O in bf16,
// fp32 accumulation on v_mfma_f32_32x32x16_bf16, fp32 scores.   (reference: R_TuckER.py:47-48
// with bf16 parameters -- BASELINE.json configs[2], FB15k-237 symmetric rank (200,200) B 2048.)
//
// Same "entity-stationary" skeleton as the split-fp16 kernel (rtk_score_split_kernel.h) with one
// operand plane and no scaling: a wave keeps the bf16 B fragments of its 32 entity rows in
// registers (loaded straight from O: a fragment IS 16 contiguous bytes of a row), the workgroup
// sweeps 32-query tiles of the packed bf16 plane through a register-staged, double-buffered LDS
// tile, KS MFMAs per tile on two alternating accumulators, logistic + branch-free buffer stores
// of the previous tile in the MFMA gaps.  With a third of the MFMA work of the fp32 path the
// kernel is bound by the fp32 score write (HBM).
#include "rtk_common.h"
#include <stdlib.h>
#include "rtk_pack.h"

#ifndef RTK_BF16_ABL
#define RTK_BF16_ABL 0          // tools/ablate/bf16 builds only: 1 no stores, 2 no LDS fragment reads, 4 no staging, 8 no barrier,
                                // 16 every query tile stores into the first tile's rows (16 MB of scores: no HBM write stream)
#endif

#ifndef RTK_BF16_ST64
#define RTK_BF16_ST64 0         // 1 (tools/ablate/bf16 builds only): fp32 scores of the V2 loop leave as 8-byte stores (two neighbouring
                                // columns of one row per lane, traded between adjacent lanes through DPP): 8 store instructions per
                                // tile instead of 16.  Parity-correct and 4 % SLOWER at the C5 shard (1.283 / 1.279 ms against
                                // 1.234 / 1.235 on one box): the store instruction count is not what the stores cost.
#endif

namespace {

// NW waves per workgroup, 32 entity rows each: the staged query tile is shared by 32*NW entities,
// so its L2 -> CU traffic per score is 2*K / (32*NW) bytes (K = 512, NW = 4: 8 B per 4-B score).
// QB = query tiles per block of the sweep: every workgroup walks the query blocks in the same
// order, so the chip works on one block (QB tiles, <= ~3 MB) at a time and it stays in the L2s.
// V2 (round 3, the 8-wave fp32-score form): ONE accumulator chain per tile whose register set alternates with the
// finished tile's between tiles (no second chain, no v_pk_add merge, no copy), and the 16 store offsets of a lane
// computed once per unit instead of one v_add per store -- ~190 issue cycles less VALU per wave and tile.
template <int KS, int SIGMOID, int MINW, int NW, bool NTS, bool OBF, bool V2 = false>
__global__ __launch_bounds__(64 * NW, MINW) void score_bf16_kernel(
    const unsigned char *__restrict__ q_packed, int B, const rtk_bf16 *__restrict__ O, int N, int c,
    float *__restrict__ out, int64_t ld_out, bool o_vec, int QB) {
    constexpr int TILE_BYTES = RTK_PACK_HDR + KS * 1024;
    constexpr int CHUNKS = TILE_BYTES / 16;
    constexpr int NT = 64 * NW;               // threads
    constexpr int NLD = (CHUNKS + NT - 1) / NT;
    static_assert(NLD <= KS, "one staging load per k-step of the chain");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];  // 2 * TILE_BYTES

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int n_mt = (B + 31) / 32, n_nt = (N + 32 * NW - 1) / (32 * NW);
    const __amdgpu_buffer_rsrc_t qrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char *>(q_packed), 0, (unsigned)(n_mt * TILE_BYTES), 0x00020000);

    for (int qb0 = 0; qb0 < n_mt; qb0 += QB) {
    const int tq = min(QB, n_mt - qb0);       // query tiles of this block
    const int64_t U = (int64_t)n_nt * tq;
#if defined(RTK_BF16_SCHED) && RTK_BF16_SCHED == 1
    // (tools/ablate/bf16) whole entity tiles per workgroup, tile = blockIdx + k * grid: every workgroup is on the same
    // query tile at the same time (one L2-resident query tile, scores written in long runs per row), unbalanced tail
    for (int tile_ = blockIdx.x; tile_ < n_nt; tile_ += gridDim.x) {
    int64_t lin = (int64_t)tile_ * tq;
    const int64_t lin_end = lin + tq;
    while (lin < lin_end) {
#elif defined(RTK_BF16_SCHED) && RTK_BF16_SCHED == 2
    // whole entity tiles first (tile = blockIdx + k * grid, all query tiles: the workgroups move through the query
    // tiles in step), then the remaining tiles' (tile, query tile) units cut evenly over the grid
    const int whole_ = n_nt / (int)gridDim.x, rem0_ = whole_ * (int)gridDim.x;
    const int64_t Ur_ = (int64_t)(n_nt - rem0_) * tq;
    int64_t lin = Ur_ * blockIdx.x / gridDim.x;
    const int64_t lin_end = Ur_ * (blockIdx.x + 1) / gridDim.x;
    int whole_k_ = 0;
    {
    while (whole_k_ < whole_ || lin < lin_end) {
        int ntile_s, mt0_s, cnt_s;
        if (whole_k_ < whole_) {
            ntile_s = blockIdx.x + whole_k_ * gridDim.x; mt0_s = qb0; cnt_s = tq; ++whole_k_;
        } else {
            ntile_s = rem0_ + (int)(lin / tq); mt0_s = qb0 + (int)(lin % tq);
            cnt_s = (int)min((int64_t)(qb0 + tq - mt0_s), lin_end - lin); lin += cnt_s;
        }
#define RTK_SCHED2_UNIT 1
#else
    int64_t lin = U * blockIdx.x / gridDim.x;
    const int64_t lin_end = U * (blockIdx.x + 1) / gridDim.x;
    {
    while (lin < lin_end) {
#endif
#ifdef RTK_SCHED2_UNIT
        const int ntile = ntile_s, mt0 = mt0_s, cnt = cnt_s;
#else
        const int ntile = (int)(lin / tq), mt0 = qb0 + (int)(lin % tq);
        const int cnt = (int)min((int64_t)(qb0 + tq - mt0), lin_end - lin);
        lin += cnt;
#endif
        const int j = ntile * 32 * NW + wave * 32 + r;  // entity (row of O, column of out)

        u32x4 stg[NLD];
        auto stage_load_one = [&](int mt, int i) {
            const unsigned vo = (i + 1 < NLD || i * NT + t < CHUNKS) ? (unsigned)(t * 16) : 0x80000000u;
            stg[i] = __builtin_amdgcn_raw_buffer_load_b128(qrs, vo, mt * TILE_BYTES + i * NT * 16, 0);
        };
        auto stage_load = [&](int mt) {
#pragma unroll
            for (int i = 0; i < NLD; ++i) stage_load_one(mt, i);
        };
        auto stage_store = [&](int buf) {
            u32x4 *dst = reinterpret_cast<u32x4 *>(lds + buf * TILE_BYTES);
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const int ch = i * NT + t;
                if (i + 1 < NLD || ch < CHUNKS) dst[ch] = stg[i];
            }
        };
        // first query tile of the unit: requested ahead of the B fragments (its L2 round trip then runs
        // beside their loads instead of between the two barriers below)
        stage_load(mt0);
        // B fragments: lane (r, h) holds k = 16*ks + 8*h + q, q < 8 of row j = 16 contiguous bytes
        const rtk_bf16 *orow = O + (int64_t)min(j, N - 1) * c;
        bf16x8 Bf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int k = 16 * ks + 8 * h;
            bf16x8 x = {0, 0, 0, 0, 0, 0, 0, 0};
            if (o_vec) {  // c % 8 == 0: a fragment is wholly inside or wholly outside the row
                if (k + 8 <= c) x = *reinterpret_cast<const bf16x8 *>(orow + k);
            } else {
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (k + q < c) x[q] = (short)orow[k + q];
            }
            Bf[ks] = x;
        }

        // OBF: bf16 scores (what the reference's bf16 model returns).  Adjacent lanes (columns j, j + 1)
        // trade one value per row pair through DPP so that each holds two neighbouring columns of ONE
        // row, rounds them with v_cvt_pk_bf16_f32 and stores 4 bytes: the even lane the pair's first
        // row, the odd lane its second.
        constexpr int ES = OBF ? 2 : 4;              // bytes per score
        const int par = lane & 1, c0 = j & ~1;       // OBF: first column of this lane's pair
        const unsigned voff = !OBF ? ((j < N) ? (unsigned)((4 * h * ld_out + j) * 4) : 0x80000000u)
                                   : ((c0 + 1 < N) ? (unsigned)(((4 * h + par) * ld_out + c0) * 2) : 0x80000000u);
        const unsigned voff_last = (OBF && c0 + 1 == N) ? (unsigned)(((4 * h + par) * ld_out + c0) * 2) : 0x80000000u;
        const bool n_odd = OBF && (N & 1);           // then the last column is stored on its own (2 bytes)
        const unsigned ld4 = (unsigned)(ld_out * ES);
        __amdgpu_buffer_rsrc_t ers;
        unsigned ep_off = voff, ep_off_last = voff_last;
        auto epilogue_begin = [&](int mt, bool live) {
            const int rows = live ? min(32, B - mt * 32) : 0;
            ers = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<unsigned char *>(out) + (int64_t)((RTK_BF16_ABL & 16) ? 0 : max(mt, 0)) * 32 * ld_out * ES, 0,
                                                    (unsigned)(rows * ld_out * ES), 0x00020000);
            ep_off = voff;
            ep_off_last = voff_last;
            asm volatile("" : "+v"(ep_off));
        };
        constexpr bool ST64 = RTK_BF16_ST64 && V2 && !OBF;
        unsigned voffs[V2 ? 16 : 1];                 // V2: value e is row 8 (e / 4) + e % 4 (+ 4 h, inside voff) of the tile
        // ST64: pair p = values (2p, 2p + 1) = two consecutive rows; after the trade the even lane holds columns (c0, c0 + 1)
        // of the first row, the odd lane those of the second; pair_base = this lane's offset for pair 0
        const unsigned pair_base = (unsigned)(((4 * h + par) * ld_out + c0) * 4);
        const bool pair_ok = c0 + 1 < N, pair_last = c0 + 1 == N;          // last: N odd, column c0 alone is real
        const bool n_odd4 = ST64 && (N & 1);
        if (V2 && !ST64) {
#pragma unroll
            for (int e = 0; e < (V2 ? 16 : 1); ++e)
                voffs[e] = (voff == 0x80000000u) ? voff : voff + (unsigned)((8 * (e >> 2) + (e & 3)) * ld4);
        }
        if (ST64) {
#pragma unroll
            for (int p = 0; p < 8; ++p)
                voffs[p] = pair_ok ? pair_base + (unsigned)((8 * (p >> 1) + 2 * (p & 1)) * ld4) : 0x80000000u;
        }
        float ep_d = 1.f, ep_p = 1.f, ep_keep = 0.f;
        auto piece = [&](const f32x16 &z, int pc) {   // 32 pieces: 16 values x {exp half, reciprocal half + store}
            const int e = pc >> 1;
            if ((pc & 1) == 0) {
                if (SIGMOID == 2) {
                    ep_d = __builtin_amdgcn_exp2f(z[e] * -1.4426950408889634f);   // 2^t = inf for very negative z: p = 0
                } else if (SIGMOID == 1) {
                    ep_d = 1.0f + expf(-z[e]);
                } else {
                    ep_p = z[e];
                }
            } else {
                float pv = ep_p;
                if (SIGMOID == 2) pv = __builtin_amdgcn_rcpf(1.0f + ep_d);   // v_exp_f32 / v_rcp_f32 are 1-ulp
                if (SIGMOID == 1) pv = 1.0f / ep_d;
                // NTS (128-B aligned rows): nontemporal -- the scores are written once and not re-read here
                if (RTK_BF16_ABL & 1) {
                    if (pv == 12345.678f) out[0] = pv;
                } else if (ST64) {
                    if (!(e & 1)) {
                        ep_keep = pv;                // first row of the pair: wait for the second
                    } else {
                        typedef float f32x2_t __attribute__((ext_vector_type(2)));
                        typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
                        const float send = par ? ep_keep : pv;
                        const float recv = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, send), 0xB1, 0xF, 0xF, true));
                        const f32x2_t pr = par ? f32x2_t{recv, pv} : f32x2_t{ep_keep, recv};   // (column c0, column c0 + 1)
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, pr), ers, voffs[e >> 1], 0, NTS ? 2 : 0);
                        if (n_odd4) {
                            const int p = e >> 1;
                            const unsigned vo = pair_last ? pair_base + (unsigned)((8 * (p >> 1) + 2 * (p & 1)) * ld4) : 0x80000000u;
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, pr[0]), ers, vo, 0, 0);
                        }
                    }
                } else if (!OBF && V2) {
#ifdef RTK_BF16_STORE_AUX       // (tools/ablate/bf16: cache-policy bits of the score stores)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, pv), ers, voffs[V2 ? e : 0], 0, RTK_BF16_STORE_AUX);
#else
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, pv), ers, voffs[V2 ? e : 0], 0, NTS ? 2 : 0);
#endif
                } else if (!OBF) {
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, pv), ers, ep_off, 0, NTS ? 2 : 0);
                    ep_off += ((e & 3) == 3) ? 5u * ld4 : ld4;   // rows 0,1,2,3,8,9,10,11,16,...
                } else if (!(e & 1)) {
                    ep_keep = pv;                    // first row of the pair: wait for the second
                } else {
                    const float send = par ? ep_keep : pv;
                    const float recv = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, send), 0xB1, 0xF, 0xF, true));
                    typedef float f32x2_t __attribute__((ext_vector_type(2)));
                    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
                    const f32x2_t pr = par ? f32x2_t{recv, pv} : f32x2_t{ep_keep, recv};   // (column c0, column c0 + 1)
                    const unsigned pk = __builtin_bit_cast(unsigned, __builtin_convertvector(pr, bf16x2_t));
                    __builtin_amdgcn_raw_buffer_store_b32(pk, ers, ep_off, 0, NTS ? 2 : 0);
                    if (n_odd) {
                        __builtin_amdgcn_raw_buffer_store_b16((unsigned short)(pk & 0xffffu), ers, ep_off_last, 0, 0);
                        ep_off_last += ((e & 3) == 3) ? 6u * ld4 : 2u * ld4;
                    }
                    ep_off += ((e & 3) == 3) ? 6u * ld4 : 2u * ld4;   // row pairs 0/1, 2/3, 8/9, 10/11, 16/17, ...
                }
            }
        };

        __syncthreads();   // previous unit's last tile fully read before restaging buffer 0
        stage_store(0);
        __syncthreads();
        if constexpr (V2) {
            constexpr bool STAGE_IN_CHAIN = NW == 8;
            constexpr int PF = KS < 4 ? KS : (NW == 8 ? 4 : 3);
            f32x16 accA, accB;
#pragma unroll
            for (int e = 0; e < 16; ++e) accB[e] = 0.f;
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            // tile i accumulates in accC while the finished tile i - 1 in accP is turned into probabilities and stored
            auto iteration = [&](int i, f32x16 &accC, const f32x16 &accP) {
                const int cur = i & 1;
                const bool stage = i + 1 < cnt;
                if (!STAGE_IN_CHAIN && stage) stage_load(mt0 + i + 1);
                const bf16x8 *la = reinterpret_cast<const bf16x8 *>(lds + cur * TILE_BYTES + RTK_PACK_HDR);
                epilogue_begin(mt0 + i - 1, i > 0);
                bf16x8 fa[PF];
#pragma unroll
                for (int p = 0; p < PF; ++p) fa[p] = la[p * 64 + lane];
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const bf16x8 a = fa[ks % PF];
                    if (ks + PF < KS && !(RTK_BF16_ABL & 2)) fa[ks % PF] = la[(ks + PF) * 64 + lane];
                    accC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, Bf[ks], ks == 0 ? zero : accC, 0, 0, 0);
                    if (STAGE_IN_CHAIN && ks < NLD && stage && !(RTK_BF16_ABL & 4)) stage_load_one(mt0 + i + 1, ks);
#pragma unroll
                    for (int pc = ks * 32 / KS; pc < (ks + 1) * 32 / KS; ++pc) piece(accP, pc);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (!(RTK_BF16_ABL & 4)) stage_store(cur ^ 1);
                if (!(RTK_BF16_ABL & 8)) __syncthreads();
            };
            int i = 0;
            for (; i + 1 < cnt; i += 2) {
                iteration(i, accA, accB);
                iteration(i + 1, accB, accA);
            }
            if (i < cnt) {
                iteration(i, accA, accB);
                epilogue_begin(mt0 + cnt - 1, true);
#pragma unroll
                for (int pc = 0; pc < 32; ++pc) piece(accA, pc);
            } else {
                epilogue_begin(mt0 + cnt - 1, true);
#pragma unroll
                for (int pc = 0; pc < 32; ++pc) piece(accB, pc);
            }
        } else {
        f32x16 prev;
#pragma unroll
        for (int e = 0; e < 16; ++e) prev[e] = 0.f;
        for (int i = 0; i < cnt; ++i) {
            const int cur = i & 1;
            // One workgroup per CU (NW == 8): the next tile's staging loads are issued one per MFMA gap
            // at the head of the chain, not in front of it -- 8 waves x NLD 1-KiB loads through one
            // address unit hold the first MFMA back by several hundred cycles on every tile (C5: 1.27 ->
            // 1.21 ms).  Two workgroups per CU (NW == 4) cover each other's gaps; there the early issue
            // wins (C3: 32.5 vs 37 us).
            constexpr bool STAGE_IN_CHAIN = NW == 8;
            const bool stage = i + 1 < cnt;
            if (!STAGE_IN_CHAIN && stage) stage_load(mt0 + i + 1);
            const bf16x8 *la = reinterpret_cast<const bf16x8 *>(lds + cur * TILE_BYTES + RTK_PACK_HDR);
            f32x16 acc, acc2;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = acc2[e] = 0.f;
            epilogue_begin(mt0 + i - 1, i > 0);
            constexpr int PF = KS < 4 ? KS : (NW == 8 ? 4 : 3);
            bf16x8 fa[PF];
#pragma unroll
            for (int p = 0; p < PF; ++p) fa[p] = la[p * 64 + lane];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 a = fa[ks % PF];
                if (ks + PF < KS) fa[ks % PF] = la[(ks + PF) * 64 + lane];
                if (ks & 1) acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, Bf[ks], acc2, 0, 0, 0);
                else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, Bf[ks], acc, 0, 0, 0);
                if (STAGE_IN_CHAIN && ks < NLD && stage) stage_load_one(mt0 + i + 1, ks);
#pragma unroll
                for (int pc = ks * 32 / KS; pc < (ks + 1) * 32 / KS; ++pc) piece(prev, pc);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) prev[e] = acc[e] + acc2[e];
            stage_store(cur ^ 1);   // unconditional: a stale tile in the spare buffer is never read
            __syncthreads();
        }
        epilogue_begin(mt0 + cnt - 1, true);
#pragma unroll
        for (int pc = 0; pc < 32; ++pc) piece(prev, pc);
        }
    }
    }
    }
}

template <int KS, int SG, int MINW, int NW, bool NTS, bool OBF, bool V2 = false>
int launch_nt(const unsigned char *qp, int B, const rtk_bf16 *O, int N, int c, float *out, int64_t ld, bool o_vec,
              hipStream_t st) {
    constexpr size_t tile = RTK_PACK_HDR + KS * 1024, smem = 2 * tile;
    static std::atomic<unsigned long long> lds_ok{0};
    if (smem > 64 * 1024) {
        const int rc = rtk_ensure_dynamic_lds(reinterpret_cast<const void *>(&score_bf16_kernel<KS, SG, MINW, NW, NTS, OBF, V2>),
                                              (int)smem, lds_ok, "score_bf16_kernel");
        if (rc != RTK_OK) return rc;
    }
    const int n_mt = (int)rtk_cdiv(B, 32);
    static const int qb_kb = getenv("RTK_BF16_QB_KB") ? atoi(getenv("RTK_BF16_QB_KB")) : 3072;   // A/B: block size of the sweep
    // query tiles per block of the sweep: <= 3 MB of packed planes (C5 shard, score kernel: 0.5 MB 1.47 ms,
    // 1 MB 1.29, 1.5 MB 1.20, 2.5-6 MB 1.15-1.17, unblocked 1.18 -- small blocks reload the B fragments too often)
    int qb = (int)(((size_t)qb_kb << 10) / tile);
    if (qb < 1) qb = 1;
    if (qb >= n_mt) qb = n_mt;
    else qb = (int)rtk_cdiv(n_mt, rtk_cdiv(n_mt, qb));   // equal blocks
    const int64_t units = rtk_cdiv(N, 32 * NW) * (int64_t)qb;
    const unsigned grid = (unsigned)(units < 256 * MINW ? units : 256 * MINW);
    RTK_LAUNCH_SCORE((score_bf16_kernel<KS, SG, MINW, NW, NTS, OBF, V2>), dim3(grid), dim3(64 * NW), smem, st, qp, B, O, N, c, out, ld, o_vec, qb);
    return RTK_OK;
}

template <int KS, int SG, int MINW, int NW>
int launch_one(const unsigned char *qp, int B, const rtk_bf16 *O, int N, int c, float *out, int64_t ld, bool o_vec,
                bool obf, hipStream_t st) {
    static const bool nts_off = getenv("RTK_NO_NT_STORES") != nullptr;
    const bool nts = !nts_off && (ld * (obf ? 2 : 4)) % 128 == 0 && (reinterpret_cast<uintptr_t>(out) & 127) == 0;
    static const bool v1 = getenv("RTK_BF16_V1") != nullptr;     // A/B: round 2's two-chain loop in the deep-K form
    if constexpr (SG == 2) {   // bf16 scores: probabilities only (the caller checks)
        if (obf) {
            // (the same accumulation order as the fp32-score form: bf16 scores = the fp32 ones rounded, bit for bit)
            if constexpr (NW == 8) {
                if (!v1) {
                    if (nts) return launch_nt<KS, SG, MINW, NW, true, true, true>(qp, B, O, N, c, out, ld, o_vec, st);
                    return launch_nt<KS, SG, MINW, NW, false, true, true>(qp, B, O, N, c, out, ld, o_vec, st);
                }
            }
            if (nts) return launch_nt<KS, SG, MINW, NW, true, true>(qp, B, O, N, c, out, ld, o_vec, st);
            return launch_nt<KS, SG, MINW, NW, false, true>(qp, B, O, N, c, out, ld, o_vec, st);
        }
    }
    if constexpr (NW == 8 && SG != 1) {   // deep-K form, fp32 scores: the V2 loop
        if (!v1) {
            if (nts) return launch_nt<KS, SG, MINW, NW, true, false, true>(qp, B, O, N, c, out, ld, o_vec, st);
            return launch_nt<KS, SG, MINW, NW, false, false, true>(qp, B, O, N, c, out, ld, o_vec, st);
        }
    }
    if constexpr (SG != 1) {   // (the exact-logistic variant keeps one form)
        if (nts) return launch_nt<KS, SG, MINW, NW, true, false>(qp, B, O, N, c, out, ld, o_vec, st);
    }
    return launch_nt<KS, SG, MINW, NW, false, false>(qp, B, O, N, c, out, ld, o_vec, st);
}

template <int KS, int MINW, int NW>
int launch_ks(const unsigned char *qp, int B, const rtk_bf16 *O, int N, int c, float *out, int64_t ld, int sg,
              bool o_vec, bool obf, hipStream_t st) {
    if (sg == 0) return launch_one<KS, 0, MINW, NW>(qp, B, O, N, c, out, ld, o_vec, obf, st);
    if (sg == 1) return launch_one<KS, 1, MINW, NW>(qp, B, O, N, c, out, ld, o_vec, obf, st);
    return launch_one<KS, 2, MINW, NW>(qp, B, O, N, c, out, ld, o_vec, obf, st);
}

template <int KS, int MINW>
int launch_shape(bool wide, const unsigned char *qp, int B, const rtk_bf16 *O, int N, int c, float *out, int64_t ld,
                  int sg, bool o_vec, bool obf, hipStream_t st) {
    if constexpr (KS > 16) {
        if (wide) return launch_ks<KS, 1, 8>(qp, B, O, N, c, out, ld, sg, o_vec, obf, st);
    }
    return launch_ks<KS, MINW, 4>(qp, B, O, N, c, out, ld, sg, o_vec, obf, st);
}

}  // namespace

extern "C" int rtk_score_packed_bf16(const void *q_packed, int64_t batch, int c, const void *O, int64_t n_local,
                                     float *out, int64_t ld_out, unsigned flags, void *stream) {
    RTK_REQUIRE(q_packed && O && out, RTK_ERR_BAD_ARG, "rtk_score_packed_bf16: null operand");
    RTK_REQUIRE(batch > 0 && n_local > 0 && c > 0, RTK_ERR_BAD_ARG, "rtk_score_packed_bf16: sizes must be positive");
    RTK_REQUIRE(ld_out >= n_local, RTK_ERR_BAD_ARG, "rtk_score_packed_bf16: ld_out < n_local");
    RTK_REQUIRE(ld_out < (1ll << 24), RTK_ERR_UNSUPPORTED, "rtk_score_packed_bf16: ld_out >= 2^24");
    RTK_REQUIRE(batch < (1ll << 31) && n_local < (1ll << 31) - 256, RTK_ERR_UNSUPPORTED, "rtk_score_packed_bf16: dimension too large");
    RTK_REQUIRE(c <= 512, RTK_ERR_UNSUPPORTED, "rtk_score_packed_bf16: c=%d > 512 not supported", c);
    hipStream_t st = (hipStream_t)stream;
    const int ks = (c + 15) / 16;
    const int sg = !(flags & RTK_SCORE_SIGMOID) ? 0 : ((flags & RTK_SCORE_SIGMOID_FAST) ? 2 : 1);
    const bool obf = (flags & RTK_SCORE_OUT_BF16) != 0;
    RTK_REQUIRE(!obf || sg == 2, RTK_ERR_UNSUPPORTED, "rtk_score_packed_bf16: RTK_SCORE_OUT_BF16 needs RTK_SCORE_SIGMOID | RTK_SCORE_SIGMOID_FAST");
    RTK_REQUIRE(!obf || (reinterpret_cast<uintptr_t>(out) & 3) == 0, RTK_ERR_BAD_ARG, "rtk_score_packed_bf16: bf16 out must be 4-byte aligned");
    const bool o_vec = (c % 8 == 0) && ((reinterpret_cast<uintptr_t>(O) & 15) == 0);
    const unsigned char *qp = (const unsigned char *)q_packed;
    const rtk_bf16 *Ob = (const rtk_bf16 *)O;
    const int B = (int)batch, N = (int)n_local;
    // 8-wave workgroups (256 entities share a staged query tile) once the problem fills the chip that way
    static const bool narrow = getenv("RTK_BF16_NARROW") != nullptr;   // A/B: 4-wave workgroups, two per CU
    const bool wide = !narrow && ks > 16 && rtk_cdiv(N, 256) * rtk_cdiv(B, 32) >= 4 * 256;
#define RTK_KS(K_, W_) case K_: rc = launch_shape<K_, W_>(wide, qp, B, Ob, N, c, out, ld_out, sg, o_vec, obf, st); break;
    int rc = RTK_OK;
    switch (ks) {
#ifdef RTK_BF16_HARNESS_KS
        RTK_KS(RTK_BF16_HARNESS_KS, 2)
#else
        RTK_KS(1, 2) RTK_KS(2, 2) RTK_KS(3, 2) RTK_KS(4, 2) RTK_KS(5, 2) RTK_KS(6, 2) RTK_KS(7, 2) RTK_KS(8, 2)
        RTK_KS(9, 2) RTK_KS(10, 2) RTK_KS(11, 2) RTK_KS(12, 2) RTK_KS(13, 2) RTK_KS(14, 2) RTK_KS(15, 2) RTK_KS(16, 2)
        RTK_KS(17, 2) RTK_KS(18, 2) RTK_KS(19, 2) RTK_KS(20, 2) RTK_KS(21, 2) RTK_KS(22, 2) RTK_KS(23, 2) RTK_KS(24, 2)
        RTK_KS(25, 2) RTK_KS(26, 2) RTK_KS(27, 2) RTK_KS(28, 2) RTK_KS(29, 2) RTK_KS(30, 2) RTK_KS(31, 2) RTK_KS(32, 2)
#endif
        default:
            rtk_set_error("rtk_score_packed_bf16: unsupported k-step count %d", ks);
            return RTK_ERR_UNSUPPORTED;
    }
#undef RTK_KS
    if (rc != RTK_OK) return rc;
    return rtk_check_launch("rtk_score_packed_bf16");
}
