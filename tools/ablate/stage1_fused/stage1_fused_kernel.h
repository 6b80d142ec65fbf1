// RECORD of a rejected variant (round 4), not part of the library: stage 1 as ONE relation-grouped kernel
// (table slab built in LDS, never in memory) + pack_rows_kernel, in place of tables_kernel + contract_kernel.
// It was wired into rtk_query.hip::query_vectors_impl / from_tables_impl (host part at the end of this file) and is
// bit-identical to the two-kernel path (tests/test_gpu_round2.py carried five shapes for it, all green), but SLOWER:
//
//   rocprofv3 --kernel-trace, bench.py at BASELINE.json configs[1] (WN18RR, B = 512), one box, average per launch
//     tables_kernel 7.4 us (4-5 without the group build) + contract_kernel 8.6 us          two-kernel path
//     stage1_fused_kernel<5> 15.5-15.8 us + pack_rows_kernel<false> 5.5 us                  this file (FCS 4, 512 threads)
//     stage1_fused_kernel<5> 18.8 us (FCS 8, 1024 threads, loads behind the scan)           first form
//   step: 51.2-51.8 us against 45.9 us
//   ablations of the kernel (RTK_F1_TUNE; average, minimum): nothing but launch + core loads + scan 7.8 / 7.0 us;
//   without the core loads 12.5; without the subject rows 15.2; without the contraction 13.3; none of the three 9.3.
//
// What it showed: at this size a kernel is worth ~4 us before it does anything (launch, id scan, two barriers), the
// 47 MB of core slices the 286 workgroups pull through their L1s another 3.4 us, and the LDS contraction 2.5 us --
// every phase about twice its estimate from bandwidths and latencies; the two existing kernels are already near what
// a launch costs.  Also tried on the same day: contract_kernel with 16 / 20 / 40 table rows in flight per thread
// instead of 8 (one dependent trip instead of five): no difference (45.6-46.2 us per step) -- that kernel is bound by
// the 82 MB it moves from L2, not by the chain.
#if 0
// ------------------------------------------------- tables + contract, fused ------
// Small relation rank, few relations, a batch of a few hundred queries (WN18RR: a = 10, 22 relations, B = 512): the two
// kernels above are launch- and latency-bound (7.4 + 8.6 us against a 33 us score kernel: a kernel boundary costs ~3 us
// and every DEPENDENT trip to memory 1-1.5 us at this size) and the second re-reads a relation's whole 160 KB table
// once per QUERY (82 MB through the L2).  Here one workgroup owns (relation r, a block of FCS float4 column slots): it
// finds the batch's queries of r itself (a scan of the B relation ids, no sort, no work list), builds its b x 4 FCS
// slab of M_r in LDS -- never in memory -- and contracts every query of r against it.  Two dependent trips: the core
// slices (requested first: they depend on nothing) beside the id scan, then the subject rows.  A relation no query
// asks for costs one scan; the table of a relation with n queries is built once, not read n times.
// Arithmetic: the slab element is tables_kernel's fmaf chain over a; a query's partial sums are contract_kernel's NGRP
// chains over b = g, g + NGRP, ... added in the order g = 0, 1, ... -- v has the bits of the two-kernel path (the
// packed planes then come from pack_rows_kernel, the tail of contract_kernel as a kernel of its own).  With prebuilt
// tables (M != nullptr: rtk_query_vectors_from_tables_*) the slab is copied instead of built; the rest is the same.
constexpr int FCS = 4;           // column slots (float4) per workgroup: 16 columns
constexpr int FNT = 512;
constexpr int FQMAX = 2048;      // most queries in a batch (the grouped kernel takes over there anyway)
constexpr int FAMAX = 16;        // largest relation rank
constexpr int FBMAX = 2 * FNT / FCS;   // largest b: two slab pieces per thread
template <int NGRP>
__global__ __launch_bounds__(FNT) void stage1_fused_kernel(const float *__restrict__ G, const float *__restrict__ M,
                                                           int a, int b, int c,
                                                           const float *__restrict__ R, int n_rel,
                                                           const float *__restrict__ S, int64_t n_sub,
                                                           const int64_t *__restrict__ rel_idx,
                                                           const int64_t *__restrict__ sub_idx, int B, int ncb,
                                                           float *__restrict__ v_out, uint32_t *__restrict__ flags,
                                                           int rel_part, int rel_parts, int tune) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int QC = FNT / (FCS * NGRP);                // queries per pass
    constexpr int CBW = FCS * 4;
    float *slab = smem;                                   // b x CBW
    float *srow = slab + (size_t)b * CBW;                 // QC x b: the subject rows of a pass
    float *part = srow + (size_t)QC * b;                  // NGRP x QC x CBW
    int *qlist = reinterpret_cast<int *>(part + (size_t)NGRP * QC * CBW);   // FQMAX x (query, subject)
    __shared__ int nq_;
    const int t = threadIdx.x;
    const int r = blockIdx.x / ncb, cb = blockIdx.x - r * ncb;
    if (rel_parts > 1 && r % rel_parts != rel_part) return;
    const int cols = c >> 2;                              // float4 column slots of a row (c % 4 == 0)
    const int cs0 = cb * FCS;
    const int ncs = min(FCS, cols - cs0);                 // live slots of this block
    // (1) the slab's sources, all in flight before anything else: piece e = (bi, cs), e = t and t + FNT
    const int64_t bc = (int64_t)b * c;
    f32x4 g[2][FAMAX];
    bool live[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int e = t + u * FNT, bi = e / FCS, cs = e - bi * FCS;
        live[u] = bi < b && cs < ncs;
        if (tune & 1) {
#pragma unroll
            for (int k = 0; k < FAMAX; ++k) g[u][k] = f32x4{1.f, 1.f, 1.f, 1.f};
        } else if (live[u]) {
            if (M) {
                g[u][0] = *reinterpret_cast<const f32x4 *>(M + ((int64_t)r * b + bi) * c + (cs0 + cs) * 4);
            } else {
                const float *g0 = G + (int64_t)bi * c + (cs0 + cs) * 4;
#pragma unroll
                for (int k = 0; k < FAMAX; ++k)
                    if (k < a) g[u][k] = *reinterpret_cast<const f32x4 *>(g0 + (int64_t)k * bc);
            }
        }
    }
    // (2) the batch's queries of relation r (a bad id counts as relation 0, like the per-query kernel) with their subjects
    if (t == 0) nq_ = 0;
    __syncthreads();
    for (int d = t; d < B; d += FNT) {
        int64_t rr = rel_idx[d], h = sub_idx[d];
        const bool bad = rr < 0 || rr >= n_rel;
        if (bad) rr = 0;
        if ((int)rr == r) {
            const bool badh = h < 0 || h >= n_sub;
            if (badh) h = 0;
            if ((bad || badh) && cb == 0) atomicOr(&flags[0], 1u);
            const int at = atomicAdd(&nq_, 1);
            qlist[2 * at] = d;
            qlist[2 * at + 1] = (int)h;
        }
    }
    __syncthreads();
    const int nq = nq_;
    if (nq == 0 || (tune & 8)) return;
    // (3) the subject rows of the first pass: requested before the slab arithmetic waits for (1)
    constexpr int SPT = (QC * (FBMAX / 4) + FNT - 1) / FNT;      // float4 pieces per thread of a pass
    const int b4n = b >> 2;
    f32x4 sx[SPT];
    auto request_rows = [&](int q0) {
        const int nqp = min(QC, nq - q0);
#pragma unroll
        for (int i = 0; i < SPT; ++i) {
            const int e = t + i * FNT;
            if (e < nqp * b4n && !(tune & 2)) {
                const int q = e / b4n, b4 = e - q * b4n;
                sx[i] = *reinterpret_cast<const f32x4 *>(S + (int64_t)qlist[2 * (q0 + q) + 1] * b + b4 * 4);
            }
        }
    };
    request_rows(0);
    // slab: M_r[bi][cs] = sum_ai R[r, ai] * G[ai, bi, cs]  (acc = fmaf(R, G, acc) from zero, ai ascending)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int e = t + u * FNT, bi = e / FCS, cs = e - bi * FCS;
        if (bi < b) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            if (live[u]) {
                if (M) {
                    acc = g[u][0];
                } else {
#pragma unroll
                    for (int k = 0; k < FAMAX; ++k) {
                        if (k < a) {
                            const float rv = R[(int64_t)r * a + k];
#pragma unroll
                            for (int j = 0; j < 4; ++j) acc[j] = fmaf(rv, g[u][k][j], acc[j]);
                        }
                    }
                }
            }
            *reinterpret_cast<f32x4 *>(slab + (size_t)bi * CBW + cs * 4) = acc;
        }
    }
    // thread = (query of the pass, group of b, column slot)
    const int cs = t % FCS, gq = (t / FCS) % NGRP, qi = t / (FCS * NGRP);
    for (int q0 = 0; q0 < nq; q0 += QC) {
        const int nqp = min(QC, nq - q0);
        if (q0 > 0) {
            __syncthreads();                              // the previous pass is done with srow and part
            request_rows(q0);
        }
#pragma unroll
        for (int i = 0; i < SPT; ++i) {
            const int e = t + i * FNT;
            if (e < nqp * b4n) *reinterpret_cast<f32x4 *>(srow + (size_t)e * 4) = sx[i];     // row q at q * b
        }
        __syncthreads();                                  // slab and rows complete
        if (qi < nqp && cs < ncs && !(tune & 4)) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const float *sq = srow + (size_t)qi * b;
#pragma unroll 8
            for (int bi = gq; bi < b; bi += NGRP) {
                const float sv = sq[bi];
                const f32x4 m = *reinterpret_cast<const f32x4 *>(slab + (size_t)bi * CBW + cs * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = fmaf(sv, m[j], acc[j]);
            }
            *reinterpret_cast<f32x4 *>(part + ((size_t)gq * QC + qi) * CBW + cs * 4) = acc;
        }
        __syncthreads();
        // the groups, added in order from zero (contract_kernel's reduction); one thread per (query, column)
        for (int e = t; e < nqp * CBW; e += FNT) {
            const int q = e / CBW, col = e - q * CBW;
            if (col < ncs * 4) {
                float x = 0.f;
#pragma unroll
                for (int gg = 0; gg < NGRP; ++gg) x += part[((size_t)gg * QC + q) * CBW + col];
                v_out[(int64_t)qlist[2 * (q0 + q)] * c + cs0 * 4 + col] = x;
            }
        }
    }
}


// ---- host side (rtk_query.hip) ----
// Stage 1 in one kernel + the pack (see stage1_fused_kernel): fp32, a <= 16, b <= 256, at most 128 relations, fewer than
// 2048 queries, b and c multiples of four with at most 256 column slots and at most eight groups of b.  RTK_ERR_UNSUPPORTED (without an
// error text) when the shape is not this one.  RTK_STAGE1_FUSED=0 turns it off (A/B).
template <typename T>
static int fused_stage1(const T *core, const float *tables, int a, int b, int c, const T *R, int64_t n_rel, const T *S,
                        int64_t n_sub, const int64_t *rel_idx, const int64_t *sub_idx, int64_t batch, float *v_out,
                        void *q_packed, const RtkWorkspace &ws, hipStream_t st, int rel_part = 0, int rel_parts = 1) {
    return RTK_ERR_UNSUPPORTED;
}
template <>
int fused_stage1<float>(const float *core, const float *tables, int a, int b, int c, const float *R, int64_t n_rel,
                        const float *S, int64_t n_sub, const int64_t *rel_idx, const int64_t *sub_idx, int64_t batch,
                        float *v_out, void *q_packed, const RtkWorkspace &ws, hipStream_t st, int rel_part, int rel_parts) {
    static const int on = getenv("RTK_STAGE1_FUSED") ? atoi(getenv("RTK_STAGE1_FUSED")) : 0;
    const int cols = c / 4;
    if (!on || a > FAMAX || n_rel > 128 || n_rel > batch || batch >= FQMAX || c % 4 != 0 || b % 4 != 0 || cols > 256 ||
        b > FBMAX || (reinterpret_cast<uintptr_t>(tables ? tables : core) & 15) != 0 || (reinterpret_cast<uintptr_t>(S) & 15) != 0)
        return RTK_ERR_UNSUPPORTED;
    const int ngrp = 256 / cols;                      // contract_kernel's groups of b (cols <= 256: >= 1)
    if (ngrp < 1 || ngrp > 8) return RTK_ERR_UNSUPPORTED;
    float *v = v_out ? v_out : ws.v;
    if (!v) return RTK_ERR_UNSUPPORTED;
    const int ncb = (cols + FCS - 1) / FCS;
    const int qc = FNT / (FCS * ngrp);
    const size_t smem = ((size_t)b * FCS * 4 + (size_t)qc * b + (size_t)ngrp * qc * FCS * 4) * sizeof(float) + 2 * FQMAX * sizeof(int);
    if (smem > 150 * 1024) return RTK_ERR_UNSUPPORTED;
    const dim3 grid((unsigned)(n_rel * ncb));
    static const int tune = getenv("RTK_F1_TUNE") ? atoi(getenv("RTK_F1_TUNE")) : 0;   // ablations (wrong results)
#define RTK_F1(NG_)                                                                                                     \
    case NG_: {                                                                                                         \
        static std::atomic<unsigned long long> ok{0};                                                                   \
        const int rc = rtk_ensure_dynamic_lds(reinterpret_cast<const void *>(&stage1_fused_kernel<NG_>), 150 * 1024, ok, \
                                              "stage1_fused_kernel");                                                   \
        if (rc != RTK_OK) return rc;                                                                                    \
        hipLaunchKernelGGL((stage1_fused_kernel<NG_>), grid, dim3(FNT), smem, st, core, tables, a, b, c, R, (int)n_rel, S,  \
                           n_sub, rel_idx, sub_idx, (int)batch, ncb, v, ws.flags, rel_part, rel_parts, tune);           \
        break;                                                                                                          \
    }
    switch (ngrp) { RTK_F1(1) RTK_F1(2) RTK_F1(3) RTK_F1(4) RTK_F1(5) RTK_F1(6) RTK_F1(7) RTK_F1(8) }
#undef RTK_F1
    if (q_packed)      // (never with rel_parts > 1: the rows are complete only after the ranks' all-reduce)
        hipLaunchKernelGGL(pack_rows_kernel<false>, dim3((unsigned)batch), dim3(256), 0, st, v, c, (c + 15) / 16,
                           (unsigned char *)q_packed);
    return rtk_check_launch("rtk_query_vectors (fused stage 1)");
}

#endif
