// Stage 1 of the scoring path: query vectors
//     v[d,:] = S[h_d,:] . ( G x_0 R[r_d,:] )            (reference: asymmetric/R_TuckER.py:43-46)
// computed in the reference's summation order (relation mode first, then subject
// mode), but regrouped by DISTINCT relation so the (B,b,c) intermediate of the
// reference's einsum never exists:
//     plan     : distinct relations of the batch -> slots           (only if n_rel > B)
//     tables   : M_u[b,c] = sum_a R[u,a] * G[a,b,c]   for each slot u   (n_u*a*b*c FMAs)
//     contract : v_d[c]   = sum_b S[h_d,b] * M_{slot(r_d)}[b,c]         (B*b*c FMAs)
// The contract kernel owns whole rows of v, so it also emits the "packed query
// planes" the split-fp16 score kernel consumes (per-row power-of-two scaling and
// hi/lo fp16 split in MFMA-fragment order; layout in rtk_pack.h).
#include "rtk_common.h"
#include <stdlib.h>
#include "rtk_pack.h"

namespace {

// ---------------------------------------------------------------- plan ------
// One workgroup.  slot_of_rel[r] = slot or -1; rel_list[slot] = r; flags[1] = n_u.
// Slot order is arbitrary (atomic ticket); nothing downstream depends on it.
__global__ __launch_bounds__(1024) void plan_kernel(const int64_t *__restrict__ rel_idx, int B, int n_rel,
                                                    int32_t *__restrict__ slot_of_rel,
                                                    int32_t *__restrict__ rel_list,
                                                    uint32_t *__restrict__ flags) {
    __shared__ unsigned count;
    const int t = threadIdx.x;
    if (t == 0) count = 0;
    for (int r = t; r < n_rel; r += blockDim.x) slot_of_rel[r] = -1;
    __syncthreads();
    bool bad = false;
    for (int d = t; d < B; d += blockDim.x) {
        const int64_t r = rel_idx[d];
        if (r < 0 || r >= n_rel) bad = true;
        else slot_of_rel[r] = -2;  // benign race: every writer stores the same value
    }
    if (bad) atomicOr(&flags[0], 1u);
    __syncthreads();
    for (int r = t; r < n_rel; r += blockDim.x) {
        if (slot_of_rel[r] == -2) {
            const unsigned s = atomicAdd(&count, 1u);
            slot_of_rel[r] = (int32_t)s;
            rel_list[s] = r;
        }
    }
    __syncthreads();
    if (t == 0) flags[1] = count;
}

// -------------------------------------------------------------- groups ------
// Queries bucketed by table slot, for the grouped contract kernel: order[] = query ids sorted by
// slot, work[w] = (slot, first position in order[], count <= QG), flags[2] = number of work items.
// Runs on ONE workgroup of 256 threads -- as an extra block of the kernel that builds the tables,
// so it costs no launch.  The order inside a slot is arbitrary (atomic tickets); every query's
// row is computed independently, so the result does not depend on it.
constexpr int GROUPS_LDS_SLOTS = 2048;   // counters live in LDS up to this many slots (global atomics beyond)
// CNT: pointer type of the counters -- an LDS array (ds_add_rtn: ~100 cycles a round) when the slots
// fit, the workspace otherwise; as one generic pointer the atomics are flat_atomic_* for both and
// the B = 8192 build took 50 us.
template <typename CNT, int NT>
__device__ __forceinline__ void build_groups_impl(const int64_t *__restrict__ rel_idx, int B, int n_rel,
                                                  const int32_t *__restrict__ slot_of_rel, int n_slots, int QG,
                                                  CNT cnt, int32_t *__restrict__ order,
                                                  int32_t *__restrict__ work, uint32_t *__restrict__ flags,
                                                  const int64_t *__restrict__ sub_idx, int64_t *__restrict__ qinfo,
                                                  int *sc_q, int *sc_w, int &base_q, int &base_w) {
    const int t = threadIdx.x;
    CNT fill = cnt + n_slots;
    for (int s = t; s < 2 * n_slots; s += NT) cnt[s] = 0;
    if (t == 0) { base_q = 0; base_w = 0; }
    __syncthreads();
    // (ids are fetched eight at a time ahead of the atomics: one workgroup walks the whole batch,
    // a load per trip would be a chain of B / 256 exposed latencies; 32 at a time costs the HOST
    // kernel 280 registers and its occupancy -- tables 5.5 -> 11 us)
    constexpr int CH = 8;
    // batches of up to NT * CH queries: slots and subject ids are read ONCE and kept in registers
    // across the scan (the scatter pass then has no load round trip of its own)
    const bool single = B <= NT * CH;
    int sl1[CH];
    int64_t hs1[CH];
    for (int d0 = t; d0 < B; d0 += NT * CH) {
        int sl[CH];
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const int d = d0 + NT * k;
            int64_t r = d < B ? rel_idx[d] : 0;
            r = r < 0 ? 0 : (r >= n_rel ? n_rel - 1 : r);   // bad ids are reported by the contract kernel
            sl[k] = slot_of_rel ? max(slot_of_rel[r], 0) : (int)r;
            if (single) {
                sl1[k] = sl[k];
                hs1[k] = (qinfo && d < B) ? sub_idx[d] : 0;
            }
        }
#pragma unroll
        for (int k = 0; k < CH; ++k)
            if (d0 + NT * k < B) atomicAdd(&cnt[sl[k]], 1);
    }
    __syncthreads();
    // exclusive scans (queries, work items) over the slots, NT at a time; fill[] := first position
    for (int s0 = 0; s0 < n_slots; s0 += NT) {
        const int s = s0 + t;
        const int nq = s < n_slots ? cnt[s] : 0;
        const int nw = (nq + QG - 1) / QG;
        // inclusive scans of (nq, nw) over the NT slots of this chunk: shuffles inside a wave, the NT / 64
        // wave totals through LDS (two barriers instead of the sixteen of a step-by-step LDS scan)
        int iq = nq, iw = nw;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int uq = __shfl_up(iq, o), uw = __shfl_up(iw, o);
            if ((t & 63) >= o) { iq += uq; iw += uw; }
        }
        if ((t & 63) == 63) { sc_q[t >> 6] = iq; sc_w[t >> 6] = iw; }
        __syncthreads();
        for (int w = 0; w < (t >> 6); ++w) { iq += sc_q[w]; iw += sc_w[w]; }
        __syncthreads();
        sc_q[t] = iq;
        sc_w[t] = iw;
        const int q0 = base_q + sc_q[t] - nq, w0 = base_w + sc_w[t] - nw;
        if (s < n_slots) {
            fill[s] = q0;
            for (int k = 0; k < nw; ++k) {
                int32_t *wk = work + 4 * (int64_t)(w0 + k);
                wk[0] = s;
                wk[1] = q0 + k * QG;
                wk[2] = min(QG, nq - k * QG);
                wk[3] = 0;
            }
        }
        __syncthreads();
        if (t == NT - 1) { base_q += sc_q[NT - 1]; base_w += sc_w[NT - 1]; }
        __syncthreads();
    }
    if (t == 0) flags[2] = (uint32_t)base_w;
    for (int d0 = t; d0 < B; d0 += NT * CH) {
        int sl[CH];
        int64_t hs[CH];
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const int d = d0 + NT * k;
            if (single) {
                sl[k] = sl1[k];
                hs[k] = hs1[k];
            } else {
                int64_t r = d < B ? rel_idx[d] : 0;
                r = r < 0 ? 0 : (r >= n_rel ? n_rel - 1 : r);
                sl[k] = slot_of_rel ? max(slot_of_rel[r], 0) : (int)r;
                hs[k] = (qinfo && d < B) ? sub_idx[d] : 0;
            }
        }
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const int d = d0 + NT * k;
            if (d >= B) continue;
            const int pos = atomicAdd(&fill[sl[k]], 1);
            order[pos] = d;
            if (qinfo) {   // what the per-query contract kernel needs about position pos, in one 16-B load
                qinfo[2 * (int64_t)pos] = hs[k];
                qinfo[2 * (int64_t)pos + 1] = (int64_t)(uint32_t)d | ((int64_t)sl[k] << 32);
            }
        }
    }
}

template <int NT>
__device__ void build_groups(const int64_t *__restrict__ rel_idx, int B, int n_rel,
                             const int32_t *__restrict__ slot_of_rel, int n_slots, int QG,
                             int32_t *__restrict__ cnt_g, int32_t *__restrict__ order,
                             int32_t *__restrict__ work, uint32_t *__restrict__ flags,
                             const int64_t *__restrict__ sub_idx, int64_t *__restrict__ qinfo) {
    __shared__ int sc_q[NT], sc_w[NT];
    __shared__ int base_q, base_w;
    __shared__ int cnt_l[2 * GROUPS_LDS_SLOTS];
    if (n_slots <= GROUPS_LDS_SLOTS)
        build_groups_impl<int *, NT>(rel_idx, B, n_rel, slot_of_rel, n_slots, QG, (int *)cnt_l, order, work, flags, sub_idx, qinfo,
                                     sc_q, sc_w, base_q, base_w);
    else
        build_groups_impl<int32_t *, NT>(rel_idx, B, n_rel, slot_of_rel, n_slots, QG, cnt_g, order, work, flags, sub_idx, qinfo,
                                         sc_q, sc_w, base_q, base_w);
}

struct GroupArgs {   // by value to the kernels that host the extra block
    const int64_t *rel_idx;
    const int32_t *slot_of_rel;
    int32_t *cnt, *order, *work;
    uint32_t *flags;
    int B, n_rel, n_slots, QG;
    const int64_t *sub_idx;
    int64_t *qinfo;
};

// on its own: 1024 threads (the build is a chain of dependent round trips per NT * 8 queries; 51 us at B = 8192
// with 256 threads)
__global__ __launch_bounds__(1024) void groups_kernel(GroupArgs ga) {
    build_groups<1024>(ga.rel_idx, ga.B, ga.n_rel, ga.slot_of_rel, ga.n_slots, ga.QG, ga.cnt, ga.order, ga.work, ga.flags, ga.sub_idx, ga.qinfo);
}

// -------------------------------------------------------------- tables ------
// M[u, n] = sum_a R[rel(u), a] * G[a, n],  n in [0, b*c).  Streaming VALU kernel for
// small relation rank (a <= 32: WN18RR has a = 10): each thread owns one float4 of
// n for UT relations, so every G element is loaded once per UT relations.
constexpr int UT = 4;
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void tables_kernel(const T *__restrict__ G, int a, int64_t bc,
                                                     const T *__restrict__ R,
                                                     const int32_t *__restrict__ rel_list, int n_u_max,
                                                     const uint32_t *__restrict__ n_u_dev,
                                                     float *__restrict__ M, GroupArgs ga) {
    __shared__ float Rs[UT * 64];  // UT relations x a (a <= 64 here)
    const int xb = ga.QG > 0 ? 1 : 0;
    if (xb && blockIdx.x == 0) {   // the extra block (dispatched first): query groups for the contract kernel
        if (blockIdx.y == 0) build_groups<256>(ga.rel_idx, ga.B, ga.n_rel, ga.slot_of_rel, ga.n_slots, ga.QG, ga.cnt, ga.order, ga.work, ga.flags, ga.sub_idx, ga.qinfo);
        return;
    }
    const int n_u = n_u_dev ? min(n_u_max, (int)*n_u_dev) : n_u_max;
    const int u0 = blockIdx.y * UT;
    if (u0 >= n_u) return;
    const int t = threadIdx.x;
    constexpr int W = VEC ? 4 : 1;
    constexpr int AB = 16;
    const int64_t n = ((int64_t)(blockIdx.x - xb) * 256 + t) * W;
    const bool live = n < bc;
    // the first AB relation-rank slices of G are requested before anything else: they do not depend
    // on the R rows, whose trip through LDS (load, store, barrier) would otherwise sit in front of
    // them as one more exposed latency of this short kernel
    float g[AB][W];
    auto load_slices = [&](int a0) {
#pragma unroll
        for (int k = 0; k < AB; ++k) {
            if (live && a0 + k < a) {   // wave-uniform in k: the loads of a batch still issue back to back
                if (VEC) {
                    const f32x4 x = rtk_load4(G + (int64_t)(a0 + k) * bc + n);
#pragma unroll
                    for (int j = 0; j < W; ++j) g[k][j] = x[j];
                } else {
                    g[k][0] = rtk_to_f32(G[(int64_t)(a0 + k) * bc + n]);
                }
            }
        }
    };
    load_slices(0);
    for (int i = t; i < UT * a; i += 256) {
        const int u = u0 + i / a, ai = i % a;
        float x = 0.f;
        if (u < n_u) {
            const int rel = rel_list ? rel_list[u] : u;
            x = rtk_to_f32(R[(int64_t)rel * a + ai]);
        }
        Rs[(i / a) * 64 + ai] = x;
    }
    __syncthreads();
    if (!live) return;
    float acc[UT][W];
#pragma unroll
    for (int u = 0; u < UT; ++u)
#pragma unroll
        for (int j = 0; j < W; ++j) acc[u][j] = 0.f;
    for (int a0 = 0; a0 < a; a0 += AB) {
        if (a0 > 0) load_slices(a0);
#pragma unroll
        for (int k = 0; k < AB; ++k) {
            if (a0 + k < a) {
#pragma unroll
                for (int u = 0; u < UT; ++u) {
                    const float rv = Rs[u * 64 + a0 + k];
#pragma unroll
                    for (int j = 0; j < W; ++j) acc[u][j] = fmaf(rv, g[k][j], acc[u][j]);
                }
            }
        }
    }
#pragma unroll
    for (int u = 0; u < UT; ++u) {
        if (u0 + u >= n_u) break;
        float *dst = M + (int64_t)(u0 + u) * bc + n;
        if (VEC) {
            f32x4 x;
#pragma unroll
            for (int j = 0; j < W; ++j) x[j] = acc[u][j];
            *reinterpret_cast<f32x4 *>(dst) = x;
        } else {
            dst[0] = acc[u][0];
        }
    }
}

// ------------------------------------------- large relation rank, bf16 ------
// For a > 32 the tables are a real GEMM, M[u, n] = sum_a R[u,a] * G[a,n] (n = (b,c) flattened).
// It runs on the bf16 score kernel: the batch's relation rows play the queries (packed planes,
// K = a) and the core, transposed to [(b,c)][a], plays the entity matrix -- both operands are
// then K-contiguous, which is what the MFMA fragments want.
__global__ __launch_bounds__(256) void transpose_core_kernel(const rtk_bf16 *__restrict__ G, int a, int64_t bc,
                                                             rtk_bf16 *__restrict__ GT, GroupArgs ga) {
    const int xb = ga.QG > 0 ? 1 : 0;
    if (xb && blockIdx.x == 0) {   // the extra block: query groups for the contract kernel -- block 0, so that it
        // is dispatched first and runs beside the whole transpose (a single workgroup's latency chain,
        // longer than any tile); here, not in the small kernel that packs the relation rows
        if (blockIdx.y == 0) build_groups<256>(ga.rel_idx, ga.B, ga.n_rel, ga.slot_of_rel, ga.n_slots, ga.QG, ga.cnt, ga.order, ga.work, ga.flags, ga.sub_idx, ga.qinfo);
        return;
    }
    // 64 (a) x 64 (n) tile through LDS; 8-byte accesses on both sides when the shapes allow (4 elements
    // along n on the way in, 4 along a on the way out): a wave instruction then moves 512 B, not 128
    __shared__ rtk_bf16 tile[64][68];
    typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
    const int64_t n0 = (int64_t)(blockIdx.x - xb) * 64;
    const int a0 = blockIdx.y * 64;
    const int t = threadIdx.x;
    const bool vin = (bc % 4 == 0) && ((reinterpret_cast<uintptr_t>(G) & 7) == 0);
    const bool vout = (a % 4 == 0) && ((reinterpret_cast<uintptr_t>(GT) & 7) == 0);
    if (vin) {
        for (int i = t; i < 64 * 16; i += 256) {      // rows of G (fixed a), 4 consecutive n per thread
            const int ai = i >> 4, ni = (i & 15) * 4;
            u16x4 x = {0, 0, 0, 0};
            if (a0 + ai < a && n0 + ni < bc) x = *reinterpret_cast<const u16x4 *>(G + (int64_t)(a0 + ai) * bc + n0 + ni);
            *reinterpret_cast<u16x4 *>(&tile[ai][ni]) = x;      // (bc % 4 == 0: a quad is wholly inside or outside)
        }
    } else {
        for (int i = t; i < 64 * 64; i += 256) {
            const int ai = i >> 6, ni = i & 63;
            rtk_bf16 x = 0;
            if (a0 + ai < a && n0 + ni < bc) x = G[(int64_t)(a0 + ai) * bc + n0 + ni];
            tile[ai][ni] = x;
        }
    }
    __syncthreads();
    if (vout) {
        for (int i = t; i < 64 * 16; i += 256) {      // rows of GT (fixed n), 4 consecutive a per thread
            const int ni = i >> 4, ai = (i & 15) * 4;
            if (a0 + ai < a && n0 + ni < bc) {
                const u16x4 x = {tile[ai][ni], tile[ai + 1][ni], tile[ai + 2][ni], tile[ai + 3][ni]};
                *reinterpret_cast<u16x4 *>(GT + (n0 + ni) * a + a0 + ai) = x;
            }
        }
    } else {
        for (int i = t; i < 64 * 64; i += 256) {
            const int ni = i >> 6, ai = i & 63;
            if (a0 + ai < a && n0 + ni < bc) GT[(n0 + ni) * a + a0 + ai] = tile[ai][ni];
        }
    }
}

__global__ __launch_bounds__(256) void pack_rel_rows_kernel(const rtk_bf16 *__restrict__ R, int a, int n_rel,
                                                            const int32_t *__restrict__ rel_list, int n_u_max,
                                                            const uint32_t *__restrict__ n_u_dev,
                                                            unsigned char *__restrict__ planes, int ksteps) {
    const int n_u = n_u_dev ? min(n_u_max, (int)*n_u_dev) : n_u_max;
    const int u = blockIdx.x;                      // one relation slot per block (rows >= n_u: zeros)
    unsigned char *tile = planes + (int64_t)(u >> 5) * rtk_pack_tile_bytes(ksteps, 1);
    const int row = u & 31;
    if (threadIdx.x == 0) reinterpret_cast<float *>(tile)[row] = 1.0f;
    rtk_bf16 *plane = reinterpret_cast<rtk_bf16 *>(tile + RTK_PACK_HDR);
    int rel = 0;
    if (u < n_u) rel = rel_list ? rel_list[u] : u;
    rel = min(max(rel, 0), n_rel - 1);
    for (int k = threadIdx.x; k < ksteps * 16; k += 256)
        plane[rtk_pack_offset(ksteps, k, row)] = (u < n_u && k < a) ? R[(int64_t)rel * a + k] : (rtk_bf16)0;
}

// ------------------------------------------------------------ contract ------
// One workgroup per query d:  v_d[c] = sum_b S[h_d, b] * M_slot[b, c].
// 256 threads = G groups x (c/W) column slots; group g takes b = g, g+G, ...;
// partial sums meet in LDS.  The finished row is written as fp32 and/or packed.
template <typename T, bool VEC, int LB = 8>
__global__ __launch_bounds__(256) void contract_kernel(const float *__restrict__ M, int b, int c,
                                                       const T *__restrict__ S, int64_t n_sub,
                                                       const int64_t *__restrict__ rel_idx,
                                                       const int64_t *__restrict__ sub_idx, int n_rel,
                                                       const int32_t *__restrict__ slot_of_rel,
                                                       float *__restrict__ v_out,
                                                       unsigned char *__restrict__ q_packed, int ksteps,
                                                       uint32_t *__restrict__ flags,
                                                       const int64_t *__restrict__ qinfo, int B,
                                                       int rel_part, int rel_parts) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *part = smem;                       // G * cpad floats, later the finished row
    __shared__ float red[4];
    constexpr int W = VEC ? 4 : 1;
    const int t = threadIdx.x;
    int d = blockIdx.x;
    int64_t h, r = 0;
    int slot = -1;
    if (qinfo) {
        // Workgroups go round-robin to the 8 XCDs; XCD x takes the x-th eighth of the queries in
        // slot order, so its L2 serves a few tables instead of all of them.  (subject, query, slot)
        // of that position come in one 16-B load (build_groups).
        const int n8 = (B + 7) / 8, pos = (blockIdx.x & 7) * n8 + (blockIdx.x >> 3);
        if (pos >= B) return;
        h = qinfo[2 * (int64_t)pos];
        const int64_t ds = qinfo[2 * (int64_t)pos + 1];
        d = (int)(uint32_t)ds;
        slot = (int)(ds >> 32);
        r = rel_idx[d];                       // only to report a bad id (the slot is already clamped)
    } else {
        if (d >= B) return;
        h = sub_idx[d];
        r = rel_idx[d];
    }
    const int cols = (c + W - 1) / W;         // column slots
    const int cpad = cols * W;
    const int ngroups = max(1, 256 / cols);   // groups of b
    const int npass = (cols + 255) / 256;     // > 1 only when cols > 256

    bool bad = false;
    if (h < 0 || h >= n_sub) { bad = true; h = 0; }
    if (r < 0 || r >= n_rel) { bad = true; r = 0; }
    // stage 1 split over ranks BY RELATION (rel_parts > 1): this launch only contracts the queries whose relation id
    // is congruent to rel_part; the rows of the others are left as they are (a bad id counts as relation 0)
    if (rel_parts > 1 && (int)(r % rel_parts) != rel_part) return;
    if (bad && t == 0) atomicOr(&flags[0], 1u);
    if (slot < 0) slot = slot_of_rel ? max(slot_of_rel[r], 0) : (int)r;
    const float *Mq = M + (int64_t)slot * b * c;
    const T *Sh = S + h * b;                  // this query's subject row: read straight from memory
                                              // (group-uniform addresses), not staged through LDS --
                                              // the table loads then do not wait behind a barrier

    for (int pass = 0; pass < npass; ++pass) {
        const int g = (npass == 1) ? t / cols : 0;
        const int col = (npass == 1) ? t % cols : pass * 256 + t;
        const bool active = (npass == 1) ? (g < ngroups) : (col < cols);
        float acc[W];
#pragma unroll
        for (int j = 0; j < W; ++j) acc[j] = 0.f;
        if (active) {
            const int gstep = (npass == 1) ? ngroups : 1;
            // LB table rows are requested before the first one is used: with one load per trip the
            // thread walks a chain of b/gstep exposed L2 latencies (the whole kernel is that chain)
            for (int b0 = g; b0 < b; b0 += gstep * LB) {
                float x[LB][W], sx[LB];
#pragma unroll
                for (int k = 0; k < LB; ++k) {
                    const int bi = b0 + k * gstep;
                    if (bi < b) {
                        sx[k] = rtk_to_f32(Sh[bi]);
                        if (VEC) {
                            const f32x4 y = *reinterpret_cast<const f32x4 *>(Mq + (int64_t)bi * c + col * 4);
#pragma unroll
                            for (int j = 0; j < W; ++j) x[k][j] = y[j];
                        } else {
                            x[k][0] = Mq[(int64_t)bi * c + col];
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < LB; ++k) {
                    const int bi = b0 + k * gstep;
                    if (bi < b) {
                        const float sv = sx[k];
#pragma unroll
                        for (int j = 0; j < W; ++j) acc[j] = fmaf(sv, x[k][j], acc[j]);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < W; ++j) part[g * cpad + col * W + j] = acc[j];
        }
    }
    __syncthreads();
    // reduce the groups; thread k owns element k of the row (k < c), looped if c > 256
    const int ng = (npass == 1) ? ngroups : 1;
    float mx = 0.f;
    for (int k = t; k < c; k += 256) {
        float x = 0.f;
        for (int g = 0; g < ng; ++g) x += part[g * cpad + k];
        part[k] = x;  // slot g = 0 of column k: only this thread touches column k from here on
        mx = fmaxf(mx, fabsf(x));
        if (v_out) v_out[(int64_t)d * c + k] = x;
    }
    if (!q_packed) return;
    if (sizeof(T) == 2) {
        // bf16 path: one plane of bf16 (round to nearest even), no scaling (bf16 has fp32's range)
        unsigned char *tile = q_packed + (int64_t)(d >> 5) * rtk_pack_tile_bytes(ksteps, 1);
        const int row = d & 31;
        if (t == 0) reinterpret_cast<float *>(tile)[row] = 1.0f;
        rtk_bf16 *plane = reinterpret_cast<rtk_bf16 *>(tile + RTK_PACK_HDR);
        for (int k = t; k < ksteps * 16; k += 256) plane[rtk_pack_offset(ksteps, k, row)] = rtk_f32_to_bf16((k < c) ? part[k] : 0.f);
        return;
    }
    // row maximum -> power-of-two scale
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((t & 63) == 0) red[t >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const int sh = rtk_pack_shift(mx);
    const float up = ldexpf(1.0f, sh);
    unsigned char *tile = q_packed + (int64_t)(d >> 5) * rtk_pack_tile_bytes(ksteps, 2);
    const int row = d & 31;
    if (t == 0) reinterpret_cast<float *>(tile)[row] = ldexpf(1.0f, -sh);
    _Float16 *planes = reinterpret_cast<_Float16 *>(tile + RTK_PACK_HDR);
    for (int k = t; k < ksteps * 16; k += 256) {
        const float x = (k < c) ? part[k] * up : 0.f;
        const _Float16 hi = (_Float16)x;
        const _Float16 lo = (_Float16)(x - (float)hi);
        const int off = rtk_pack_offset(ksteps, k, row);
        planes[off] = hi;
        planes[off + ksteps * 512] = lo;  // plane 1 follows plane 0 (ksteps*2*32*8 halves)
    }
}

// ---------------------------------------------------- grouped contract ------
// One workgroup per work item = up to QG queries that share a table slot: every table row is
// loaded once for the QG queries (the per-query kernel above re-reads the whole table for each
// query -- B * b * c * 4 bytes through the L2, the bound of that kernel).
//     v_d[c] = sum_b S[h_d, b] * M_slot[b, c]        d in the group
// 256 threads = G groups of b x (c/W) column slots, partial sums meet in LDS; the finished rows
// go out as fp32 and/or packed planes exactly as in the per-query kernel.
template <typename T, bool VEC, int QG>
__global__ __launch_bounds__(256) void contract_grouped_kernel(
    const float *__restrict__ M, int b, int c, const T *__restrict__ S, int64_t n_sub,
    const int64_t *__restrict__ rel_idx, const int64_t *__restrict__ sub_idx, int n_rel,
    const int32_t *__restrict__ work, const int32_t *__restrict__ order, float *__restrict__ v_out,
    unsigned char *__restrict__ q_packed, int ksteps, uint32_t *__restrict__ flags, int rel_part, int rel_parts) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ float red[4][QG];
    __shared__ int qid[QG];
    constexpr int W = VEC ? 4 : 1;
    const int t = threadIdx.x;
    // work items are in slot order; XCD x (workgroups go round-robin to the 8 XCDs) takes the x-th
    // eighth of them, so a table is read through one L2
    const int n_work = (int)flags[2], n8 = (n_work + 7) / 8;
    const int item = (blockIdx.x & 7) * n8 + (blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= n8 || item >= n_work) return;
    const int32_t *wk = work + 4 * (int64_t)item;
    const int slot = wk[0], q0 = wk[1], nq = wk[2];
    // stage 1 split over ranks by relation (the slot IS the relation id where this is used: prebuilt tables): the work
    // items of the other ranks' relations end here, before their table is touched
    if (rel_parts > 1 && slot % rel_parts != rel_part) return;
    const int bpad = (b + 3) & ~3;
    const int cols = (c + W - 1) / W, cpad = cols * W;   // cols <= 256 (host)
    const int ngroups = 256 / cols;
    const int PG = ngroups > 1 ? ngroups - 1 : 1;
    float *s_rows = smem;                    // QG x bpad
    float *part = smem + QG * bpad;          // QG x PG x cpad: partial sums of groups 1.., later (slot 0) the finished rows
    const float *Mq = M + (int64_t)slot * b * c;

    if (t < QG) {
        int d = -1;
        if (t < nq) {
            d = order[q0 + t];
            const int64_t h = sub_idx[d], r = rel_idx[d];
            if (h < 0 || h >= n_sub || r < 0 || r >= n_rel) atomicOr(&flags[0], 1u);
        }
        qid[t] = d;
    }
    __syncthreads();
    for (int i = t; i < QG * b; i += 256) {
        const int q = i / b, bi = i - q * b;
        float x = 0.f;
        if (q < nq) {
            int64_t h = sub_idx[qid[q]];
            h = h < 0 || h >= n_sub ? 0 : h;
            x = rtk_to_f32(S[h * b + bi]);
        }
        s_rows[q * bpad + bi] = x;
    }
    __syncthreads();

    const int g = t / cols, col = t - g * cols;
    float acc[QG][W];
#pragma unroll
    for (int q = 0; q < QG; ++q)
#pragma unroll
        for (int j = 0; j < W; ++j) acc[q][j] = 0.f;
    if (g < ngroups) {
        constexpr int LB = QG <= 4 ? 20 : 8;  // table rows requested before the first is used (bytes in flight set the rate)
        for (int b0 = g; b0 < b; b0 += ngroups * LB) {
            float x[LB][W];
#pragma unroll
            for (int k = 0; k < LB; ++k) {
                const int bi = b0 + k * ngroups;
                if (bi < b) {
                    if (VEC) {
                        const f32x4 y = *reinterpret_cast<const f32x4 *>(Mq + (int64_t)bi * c + col * 4);
#pragma unroll
                        for (int j = 0; j < W; ++j) x[k][j] = y[j];
                    } else {
                        x[k][0] = Mq[(int64_t)bi * c + col];
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < LB; ++k) {
                const int bi = b0 + k * ngroups;
                if (bi < b) {
#pragma unroll
                    for (int q = 0; q < QG; ++q) {
                        const float sv = s_rows[q * bpad + bi];
#pragma unroll
                        for (int j = 0; j < W; ++j) acc[q][j] = fmaf(sv, x[k][j], acc[q][j]);
                    }
                }
            }
        }
        if (g >= 1) {
#pragma unroll
            for (int q = 0; q < QG; ++q)
#pragma unroll
                for (int j = 0; j < W; ++j) part[(q * PG + g - 1) * cpad + col * W + j] = acc[q][j];
        }
    }
    __syncthreads();
    // Group 0 keeps its sums in registers and adds the other groups' in group order (0 + p0 + p1 + ...: the
    // per-query kernel's summation order); the finished row overwrites group 1's partials, column by column by
    // the thread that has just read them.  (Every group through LDS cost QG x ngroups x c floats: 64 KB at
    // QG = 16, c = 512 -- one workgroup per CU; now 32 KB.)
    float mx[QG];
#pragma unroll
    for (int q = 0; q < QG; ++q) mx[q] = 0.f;
    if (g == 0) {
#pragma unroll
        for (int q = 0; q < QG; ++q) {
            if (q < nq) {
#pragma unroll
                for (int j = 0; j < W; ++j) {
                    const int k = col * W + j;
                    if (k < c) {
                        float x = 0.f;
                        x += acc[q][j];
                        for (int gg = 1; gg < ngroups; ++gg) x += part[(q * PG + gg - 1) * cpad + k];
                        part[q * PG * cpad + k] = x;
                        mx[q] = fmaxf(mx[q], fabsf(x));
                        if (v_out) v_out[(int64_t)qid[q] * c + k] = x;
                    }
                }
            }
        }
    }
    if (!q_packed) return;
    if (sizeof(T) == 2) {   // bf16: one plane, round to nearest even, no scaling
        __syncthreads();
        for (int q = 0; q < nq; ++q) {
            const int d = qid[q];
            unsigned char *tile = q_packed + (int64_t)(d >> 5) * rtk_pack_tile_bytes(ksteps, 1);
            const int row = d & 31;
            if (t == 0) reinterpret_cast<float *>(tile)[row] = 1.0f;
            rtk_bf16 *plane = reinterpret_cast<rtk_bf16 *>(tile + RTK_PACK_HDR);
            const float *rowv = part + q * PG * cpad;
            for (int k = t; k < ksteps * 16; k += 256)
                plane[rtk_pack_offset(ksteps, k, row)] = rtk_f32_to_bf16((k < c) ? rowv[k] : 0.f);
        }
        return;
    }
#pragma unroll
    for (int q = 0; q < QG; ++q) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx[q] = fmaxf(mx[q], __shfl_xor(mx[q], o));
        if ((t & 63) == 0) red[t >> 6][q] = mx[q];
    }
    __syncthreads();
    for (int q = 0; q < nq; ++q) {
        const float m = fmaxf(fmaxf(red[0][q], red[1][q]), fmaxf(red[2][q], red[3][q]));
        const int sh = rtk_pack_shift(m);
        const float up = ldexpf(1.0f, sh);
        const int d = qid[q];
        unsigned char *tile = q_packed + (int64_t)(d >> 5) * rtk_pack_tile_bytes(ksteps, 2);
        const int row = d & 31;
        if (t == 0) reinterpret_cast<float *>(tile)[row] = ldexpf(1.0f, -sh);
        _Float16 *planes = reinterpret_cast<_Float16 *>(tile + RTK_PACK_HDR);
        const float *rowv = part + q * PG * cpad;
        for (int k = t; k < ksteps * 16; k += 256) {
            const float x = (k < c) ? rowv[k] * up : 0.f;
            const _Float16 hi = (_Float16)x;
            const _Float16 lo = (_Float16)(x - (float)hi);
            const int off = rtk_pack_offset(ksteps, k, row);
            planes[off] = hi;
            planes[off + ksteps * 512] = lo;
        }
    }
}

// ---------------------------------------------------------------- pack ------
// Packed query planes from fp32 query vectors that were computed elsewhere (entity-sharded scoring with
// stage 1 split over the ranks: the B x c vectors arrive by all-gather).  Same arithmetic as the tail of
// the contract kernels: bf16 = round to nearest even, one plane; fp32 = row maximum -> power-of-two
// scale -> fp16 hi/lo planes.  One workgroup per query row.
template <bool BF16>
__global__ __launch_bounds__(256) void pack_rows_kernel(const float *__restrict__ v, int c, int ksteps,
                                                        unsigned char *__restrict__ q_packed) {
    __shared__ float red[4];
    const int t = threadIdx.x;
    const int64_t d = blockIdx.x;
    const float *row = v + d * c;
    unsigned char *tile = q_packed + (d >> 5) * rtk_pack_tile_bytes(ksteps, BF16 ? 1 : 2);
    const int r = (int)(d & 31);
    if (BF16) {
        if (t == 0) reinterpret_cast<float *>(tile)[r] = 1.0f;
        rtk_bf16 *plane = reinterpret_cast<rtk_bf16 *>(tile + RTK_PACK_HDR);
        for (int k = t; k < ksteps * 16; k += 256) plane[rtk_pack_offset(ksteps, k, r)] = rtk_f32_to_bf16((k < c) ? row[k] : 0.f);
        return;
    }
    float mx = 0.f;
    for (int k = t; k < c; k += 256) mx = fmaxf(mx, fabsf(row[k]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((t & 63) == 0) red[t >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const int sh = rtk_pack_shift(mx);
    const float up = ldexpf(1.0f, sh);
    if (t == 0) reinterpret_cast<float *>(tile)[r] = ldexpf(1.0f, -sh);
    _Float16 *planes = reinterpret_cast<_Float16 *>(tile + RTK_PACK_HDR);
    for (int k = t; k < ksteps * 16; k += 256) {
        const float x = (k < c) ? row[k] * up : 0.f;
        const _Float16 hi = (_Float16)x;
        const _Float16 lo = (_Float16)(x - (float)hi);
        const int off = rtk_pack_offset(ksteps, k, r);
        planes[off] = hi;
        planes[off + ksteps * 512] = lo;
    }
}

// rows of R in table-slot order (slots past the batch's distinct relations hold an arbitrary valid row: never read)
__global__ __launch_bounds__(64) void gather_rel_rows_kernel(const float *__restrict__ R, int a, int n_rel,
                                                             const int32_t *__restrict__ rel_list, float *__restrict__ out) {
    const int u = blockIdx.x;
    int rel = rel_list[u];
    rel = rel < 0 ? 0 : (rel >= n_rel ? n_rel - 1 : rel);
    for (int k = threadIdx.x; k < a; k += 64) out[(int64_t)u * a + k] = R[(int64_t)rel * a + k];
}

}  // namespace

int rtk_gemm_f32_ex(const void *A, int a_kmajor, int64_t lda, const int32_t *a_rows, const void *B,
                    int b_kmajor, int64_t ldb, float *C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                    unsigned flags, const uint32_t *m_dev, int in_bf16, hipStream_t st);

// ---- host side ---------------------------------------------------------------------------------
// Stage 1 is two steps with a clean interface between them, the relation tables M (n_slots x b x c,
// fp32): (1) build_tables: M_u = G x_0 R[rel(u)]  -- a function of the PARAMETERS and of which relations
// are asked for; (2) contract_stage: v_d = S[h_d] . M_{slot(r_d)}.  rtk_query_vectors_* runs both per
// call (tables of the batch's distinct relations); rtk_relation_tables_* runs (1) once for all
// relations (slot == relation id) and rtk_query_vectors_from_tables_* runs (2) per batch -- at
// evaluation time the tables only change when the parameters do.

// (1)  rel_list / n_u_dev: slot -> relation id and device-side slot count (planned batches), or
// nullptr / nullptr for "slot == relation id, n_u_max slots".  ga.QG > 0: one extra workgroup of the
// first kernel builds the query groups (costs no launch).  core_t / r_packed: scratch of the bf16
// large-relation-rank path (transposed core, packed relation rows), may be null otherwise.
template <typename T>
static int build_tables(const T *core, int a, int b, int c, const T *R, int64_t n_rel, const int32_t *rel_list,
                        int n_u_max, const uint32_t *n_u_dev, float *tables, void *core_t, void *r_packed,
                        const GroupArgs &ga, hipStream_t st) {
    const int64_t bc = (int64_t)b * c;
    constexpr int VA = rtk_vec4_align<T>();
    const unsigned xb = ga.QG > 0 ? 1u : 0u;   // the extra block that builds the groups
    if (a <= 32) {
        const bool vec = (bc % 4 == 0) && ((reinterpret_cast<uintptr_t>(core) & (VA - 1)) == 0) &&
                         ((reinterpret_cast<uintptr_t>(tables) & 15) == 0);
        const int W = vec ? 4 : 1;
        dim3 grid((unsigned)rtk_cdiv(bc, 256 * W) + xb, (unsigned)rtk_cdiv(n_u_max, UT));
        RTK_REQUIRE(grid.y <= 65535, RTK_ERR_UNSUPPORTED, "relation tables: more than %d relations per launch", 65535 * UT);
        if (vec) hipLaunchKernelGGL((tables_kernel<T, true>), grid, dim3(256), 0, st, core, a, bc, R, rel_list, n_u_max, n_u_dev, tables, ga);
        else hipLaunchKernelGGL((tables_kernel<T, false>), grid, dim3(256), 0, st, core, a, bc, R, rel_list, n_u_max, n_u_dev, tables, ga);
    } else if (sizeof(T) == 2 && core_t && r_packed) {
        // bf16, a <= 512: transpose the core, pack the relation rows, run the bf16 MFMA score kernel
        // with (queries, entities, K) := (relation slots, (b,c) pairs, a); raw fp32 output = the tables
        const int ks_a = (a + 15) / 16;
        dim3 tg((unsigned)rtk_cdiv(bc, 64) + xb, (unsigned)rtk_cdiv(a, 64));
        hipLaunchKernelGGL(transpose_core_kernel, tg, dim3(256), 0, st, (const rtk_bf16 *)core, a, bc, (rtk_bf16 *)core_t, ga);
        const int rows_padded = (int)rtk_cdiv(n_u_max, 32) * 32;
        hipLaunchKernelGGL(pack_rel_rows_kernel, dim3((unsigned)rows_padded), dim3(256), 0, st, (const rtk_bf16 *)R, a,
                           (int)n_rel, rel_list, n_u_max, n_u_dev, (unsigned char *)r_packed, ks_a);
        int rc = rtk_score_packed_bf16(r_packed, n_u_max, a, core_t, bc, tables, bc, 0, (void *)st);
        if (rc != RTK_OK) return rc;
    } else if (sizeof(T) == 4 && r_packed) {
        // fp32, a > 32 (FB15k: a = 200, 2 690 relations): M[u, n] = sum_a R[rel(u), a] * G[a, n] on the split-fp16
        // MFMA GEMM of the backward (three f16 MFMAs per k-step on hi/lo halves, one power-of-two scale per operand
        // from its magnitude bound: the accuracy class the score kernel already works in) -- 2.2x the exact-fp32
        // MFMA GEMM that built these tables in round 2 (494 us at C4).  The bounds are taken over ALL of R and G,
        // so a relation's table has the same bits whether it is built for one batch or for every relation.
        if (xb) hipLaunchKernelGGL(groups_kernel, dim3(1), dim3(1024), 0, st, ga);
        float *bounds = (float *)r_packed;
        float *rg = (float *)((unsigned char *)r_packed + 256);
        int rc = rtk_absmax_f32((const float *)R, n_rel, a, a, bounds, (void *)st);
        if (rc != RTK_OK) return rc;
        rc = rtk_absmax_f32((const float *)core, a, bc, bc, bounds + 1, (void *)st);
        if (rc != RTK_OK) return rc;
        const float *A = (const float *)R;
        if (rel_list) {
            hipLaunchKernelGGL(gather_rel_rows_kernel, dim3((unsigned)n_u_max), dim3(64), 0, st, (const float *)R, a, (int)n_rel,
                               rel_list, rg);
            A = rg;
        }
        rc = rtk_gemm_sf16_splitk(A, 1, a, bounds, (const float *)core, 0, bc, bounds + 1, tables, bc, n_u_max, bc, a, 1,
                                  nullptr, 0, (void *)st);
        if (rc != RTK_OK) return rc;
    } else {
        // M[u, n] = sum_a R[rel(u), a] * G[a, n]  as an fp32 MFMA GEMM (bf16 operands widen on load)
        if (xb) hipLaunchKernelGGL(groups_kernel, dim3(1), dim3(1024), 0, st, ga);
        int rc = rtk_gemm_f32_ex(R, 1, a, rel_list, core, 0, bc, tables, bc, n_u_max, bc, a, 0, n_u_dev,
                                 sizeof(T) == 2, st);
        if (rc != RTK_OK) return rc;
    }
    return rtk_check_launch("relation tables");
}

struct ContractPlan {   // host-side choices of step (2), needed before step (1) (the groups ride in its first kernel)
    bool grouped, cvec;
    int QG;
    size_t smem_grouped;
};
constexpr size_t GROUPED16_LDS_MAX = 78 * 1024;
static ContractPlan plan_contract(int b, int c, int64_t batch, const float *tables, int64_t n_slots) {
    ContractPlan p;
    p.cvec = (c % 4 == 0) && ((reinterpret_cast<uintptr_t>(tables) & 15) == 0);
    const int cW = p.cvec ? 4 : 1;
    const int ccols = (c + cW - 1) / cW;
    // queries per work item (a work item reads its whole b x c table once).  16 is built for A/B only
    // (RTK_CONTRACT_QG=16): at BASELINE configs[4] -- 8192 queries over 1000 relations, 43 % of the relations with more
    // than eight queries, 1430 items of 8 against ~1000 of 16, 1 GB of tables -- the step was 2 % SLOWER with 16
    // (1.441 / 1.461 ms against 1.413 / 1.428 on one box: 212 registers and 64 KB of LDS leave two workgroups per CU)
    static const int force_qg = getenv("RTK_CONTRACT_QG") ? atoi(getenv("RTK_CONTRACT_QG")) : 0;
    p.QG = batch >= 2048 ? 8 : 4;
    if (force_qg == 4 || force_qg == 8 || force_qg == 16) p.QG = force_qg;
    const int ngroups = ccols <= 256 ? 256 / ccols : 1;
    const int pg = ngroups > 1 ? ngroups - 1 : 1;
    p.smem_grouped = ccols <= 256
        ? (size_t)((size_t)p.QG * ((b + 3) & ~3) + (size_t)p.QG * pg * ccols * cW) * sizeof(float) : (size_t)-1;
    static const int force = [] {   // RTK_CONTRACT=perquery|grouped: A/B comparisons
        const char *e = getenv("RTK_CONTRACT");
        return !e ? 0 : (e[0] == 'p' ? 1 : (e[0] == 'g' ? 2 : 0));
    }();
    // (16 queries: up to 78 KB of dynamic LDS, two workgroups per CU -- opted in per instantiation in contract_stage)
    if (p.smem_grouped != (size_t)-1 && p.smem_grouped > GROUPED16_LDS_MAX && p.QG == 16) {
        p.QG = 8;
        p.smem_grouped = (size_t)((size_t)p.QG * ((b + 3) & ~3) + (size_t)p.QG * pg * ccols * cW) * sizeof(float);
    }
    p.grouped = p.smem_grouped <= (p.QG == 16 ? GROUPED16_LDS_MAX : 64 * 1024 - 1024) && force != 1 && (batch >= 2048 || force == 2);
    return p;
}

// (2)  slot_of_rel: relation id -> table slot (planned batches) or nullptr for slot == relation id.
// have_groups: the slot order of the queries (grp_order / grp_work / grp_qinfo, flags[2]) was built.
template <typename T>
static int contract_stage(const float *tables, int b, int c, const T *S, int64_t n_sub, const int64_t *rel_idx,
                          const int64_t *sub_idx, int64_t n_rel, int64_t batch, const int32_t *slot_of_rel,
                          int n_slots, const ContractPlan &cp, bool have_groups, const int32_t *grp_work,
                          const int32_t *grp_order, const int64_t *grp_qinfo, uint32_t *flags, float *v_out,
                          void *q_packed, hipStream_t st, int rel_part = 0, int rel_parts = 1) {
    const int ksteps = (c + 15) / 16;
    if (cp.grouped) {
        RTK_REQUIRE(have_groups, RTK_ERR_BAD_ARG, "rtk_query_vectors: grouped contract without groups");
        const unsigned nwg = (unsigned)(batch / cp.QG + (n_slots < batch ? n_slots : batch) + 8);   // upper bound on the work items, rounded up to 8 (flags[2] holds the count)
#define RTK_CG(V_, Q_) hipLaunchKernelGGL((contract_grouped_kernel<T, V_, Q_>), dim3(nwg), dim3(256), cp.smem_grouped, st, tables, b, c, S, n_sub, rel_idx, sub_idx, (int)n_rel, grp_work, grp_order, v_out, (unsigned char *)q_packed, ksteps, flags, rel_part, rel_parts)
        if (cp.QG == 16) {
            static std::atomic<unsigned long long> ok_v{0}, ok_s{0};
            const int rc = cp.cvec ? rtk_ensure_dynamic_lds(reinterpret_cast<const void *>(&contract_grouped_kernel<T, true, 16>), (int)GROUPED16_LDS_MAX, ok_v, "contract_grouped_kernel")
                                   : rtk_ensure_dynamic_lds(reinterpret_cast<const void *>(&contract_grouped_kernel<T, false, 16>), (int)GROUPED16_LDS_MAX, ok_s, "contract_grouped_kernel");
            if (rc != RTK_OK) return rc;
        }
        if (cp.cvec) { if (cp.QG == 16) RTK_CG(true, 16); else if (cp.QG == 8) RTK_CG(true, 8); else RTK_CG(true, 4); }
        else { if (cp.QG == 16) RTK_CG(false, 16); else if (cp.QG == 8) RTK_CG(false, 8); else RTK_CG(false, 4); }
#undef RTK_CG
        return rtk_check_launch("rtk_query_vectors");
    }
    const int W = cp.cvec ? 4 : 1;
    const int cols = (c + W - 1) / W;
    const int ngroups = cols >= 256 ? 1 : 256 / cols;
    const size_t smem = (size_t)ngroups * cols * W * sizeof(float);
    RTK_REQUIRE(smem <= 64 * 1024, RTK_ERR_UNSUPPORTED, "rtk_query_vectors: rank too large for the contract kernel (b=%d c=%d)", b, c);
    const int64_t *pq_order = have_groups ? grp_qinfo : nullptr;
    const unsigned pq_grid = (unsigned)(have_groups ? rtk_cdiv(batch, 8) * 8 : batch);
    if (cp.cvec) hipLaunchKernelGGL((contract_kernel<T, true>), dim3(pq_grid), dim3(256), smem, st, tables, b, c, S, n_sub, rel_idx, sub_idx, (int)n_rel, slot_of_rel, v_out, (unsigned char *)q_packed, ksteps, flags, pq_order, (int)batch, rel_part, rel_parts);
    else hipLaunchKernelGGL((contract_kernel<T, false>), dim3(pq_grid), dim3(256), smem, st, tables, b, c, S, n_sub, rel_idx, sub_idx, (int)n_rel, slot_of_rel, v_out, (unsigned char *)q_packed, ksteps, flags, pq_order, (int)batch, rel_part, rel_parts);
    return rtk_check_launch("rtk_query_vectors");
}

// Enqueue stage 1.  `ws` already carved (rtk_abi.hip).  T = float or rtk_bf16 (operands);
// tables, accumulation and v_out are fp32 either way.
template <typename T>
static int query_vectors_impl(const T *core, int a, int b, int c, const T *R, int64_t n_rel, const T *S,
                              int64_t n_sub, const int64_t *rel_idx, const int64_t *sub_idx, int64_t batch,
                              float *v_out, void *q_packed, const RtkWorkspace &ws, hipStream_t st) {
    const bool planned = n_rel > batch;  // otherwise: one table per relation id, slot == id
    const int n_u_max = (int)(planned ? batch : n_rel);
    // the error word (flags[0]) is sticky and owned by the caller: zeroed at workspace creation
    // and by rtk_read_error_flag, not per call (a memset node costs ~4 us per batch)
    if (planned) {
        hipLaunchKernelGGL(plan_kernel, dim3(1), dim3(1024), 0, st, rel_idx, (int)batch, (int)n_rel,
                           ws.slot_of_rel, ws.rel_list, ws.flags);
    }
    const ContractPlan cp = plan_contract(b, c, batch, ws.tables, n_u_max);
    // the slot order is built either way: the per-query kernel uses it to keep a table in one XCD's L2
    // (the per-position (subject, query, slot) records are only read by the per-query contract kernel)
    // The slot order of the queries rides in the first kernel of the table build (one extra workgroup, no launch).  For
    // the per-query contract kernel it is only worth L2 locality (~1.8 us at B = 512), and at a small relation rank the
    // one-workgroup sort IS the tables kernel's critical path (7.4 us against ~4 without it): built there only when the
    // table build is a GEMM (a > 32).  WN18RR per-batch step 49.0 -> 46.9 us on one box (RTK_PB_GROUPS=1 / 0 force it).
    static const int pb_env = getenv("RTK_PB_GROUPS") ? atoi(getenv("RTK_PB_GROUPS")) : -1;
    const bool groups = cp.grouped || (pb_env < 0 ? a > 32 : pb_env != 0);
    GroupArgs ga{rel_idx, planned ? ws.slot_of_rel : nullptr, ws.grp_cnt, ws.grp_order, ws.grp_work, ws.flags,
                 (int)batch, (int)n_rel, n_u_max, groups ? cp.QG : 0, sub_idx, cp.grouped ? nullptr : ws.grp_qinfo};
    int rc = build_tables<T>(core, a, b, c, R, n_rel, planned ? ws.rel_list : nullptr, n_u_max,
                             planned ? ws.flags + 1 : nullptr, ws.tables, ws.core_t, ws.r_packed, ga, st);
    if (rc != RTK_OK) return rc;
    return contract_stage<T>(ws.tables, b, c, S, n_sub, rel_idx, sub_idx, n_rel, batch,
                             planned ? ws.slot_of_rel : nullptr, n_u_max, cp, groups, ws.grp_work, ws.grp_order,
                             ws.grp_qinfo, ws.flags, v_out, q_packed, st);
}

// Tables of ALL relations, slot == relation id (rtk_relation_tables_*).
template <typename T>
static int relation_tables_impl(const T *core, int a, int b, int c, const T *R, int64_t n_rel, float *tables,
                                void *core_t, void *r_packed, hipStream_t st) {
    GroupArgs ga{};   // QG = 0: no groups
    return build_tables<T>(core, a, b, c, R, n_rel, nullptr, (int)n_rel, nullptr, tables, core_t, r_packed, ga, st);
}

// Step (2) alone against prebuilt tables (rtk_query_vectors_from_tables_*).  ws: the small "from
// tables" workspace (rtk_abi.hip::carve_ft).  The grouped kernel (batch >= 2048) needs the slot order
// of the queries: one single-workgroup launch; the per-query kernel runs without it (one launch
// instead of two: at B = 512 the order is worth 1.8 us of L2 locality, a launch costs more).
template <typename T>
static int from_tables_impl(const float *tables, int64_t n_rel, int b, int c, const T *S, int64_t n_sub,
                            const int64_t *rel_idx, const int64_t *sub_idx, int64_t batch, float *v_out,
                            void *q_packed, const RtkWorkspace &ws, hipStream_t st, int rel_part = 0, int rel_parts = 1) {
    const ContractPlan cp = plan_contract(b, c, batch, tables, n_rel);
    static const bool order_small = getenv("RTK_FT_ORDER") != nullptr;   // A/B: slot order for the per-query kernel too
    const bool groups = cp.grouped || order_small;
    if (groups) {
        GroupArgs ga{rel_idx, nullptr, ws.grp_cnt, ws.grp_order, ws.grp_work, ws.flags,
                     (int)batch, (int)n_rel, (int)n_rel, cp.QG, sub_idx, cp.grouped ? nullptr : ws.grp_qinfo};
        hipLaunchKernelGGL(groups_kernel, dim3(1), dim3(1024), 0, st, ga);
    }
    return contract_stage<T>(tables, b, c, S, n_sub, rel_idx, sub_idx, n_rel, batch, nullptr, (int)n_rel, cp, groups,
                             ws.grp_work, ws.grp_order, ws.grp_qinfo, ws.flags, v_out, q_packed, st, rel_part, rel_parts);
}

int rtk_query_vectors_f32_impl(const float *core, int a, int b, int c, const float *R, int64_t n_rel,
                               const float *S, int64_t n_sub, const int64_t *rel_idx,
                               const int64_t *sub_idx, int64_t batch, float *v_out, void *q_packed,
                               const RtkWorkspace &ws, hipStream_t st) {
    return query_vectors_impl<float>(core, a, b, c, R, n_rel, S, n_sub, rel_idx, sub_idx, batch, v_out, q_packed, ws, st);
}

int rtk_query_vectors_bf16_impl(const void *core, int a, int b, int c, const void *R, int64_t n_rel,
                                const void *S, int64_t n_sub, const int64_t *rel_idx,
                                const int64_t *sub_idx, int64_t batch, float *v_out, void *q_packed,
                                const RtkWorkspace &ws, hipStream_t st) {
    return query_vectors_impl<rtk_bf16>((const rtk_bf16 *)core, a, b, c, (const rtk_bf16 *)R, n_rel,
                                        (const rtk_bf16 *)S, n_sub, rel_idx, sub_idx, batch, v_out, q_packed, ws, st);
}

int rtk_relation_tables_f32_impl(const float *core, int a, int b, int c, const float *R, int64_t n_rel, float *tables,
                                 void *r_scratch, hipStream_t st) {
    return relation_tables_impl<float>(core, a, b, c, R, n_rel, tables, nullptr, r_scratch, st);
}

int rtk_relation_tables_bf16_impl(const void *core, int a, int b, int c, const void *R, int64_t n_rel, float *tables,
                                  void *core_t, void *r_packed, hipStream_t st) {
    return relation_tables_impl<rtk_bf16>((const rtk_bf16 *)core, a, b, c, (const rtk_bf16 *)R, n_rel, tables, core_t,
                                          r_packed, st);
}

int rtk_from_tables_f32_impl(const float *tables, int64_t n_rel, int b, int c, const float *S, int64_t n_sub,
                             const int64_t *rel_idx, const int64_t *sub_idx, int64_t batch, float *v_out,
                             void *q_packed, const RtkWorkspace &ws, hipStream_t st, int rel_part, int rel_parts) {
    return from_tables_impl<float>(tables, n_rel, b, c, S, n_sub, rel_idx, sub_idx, batch, v_out, q_packed, ws, st, rel_part,
                                   rel_parts);
}

int rtk_from_tables_bf16_impl(const float *tables, int64_t n_rel, int b, int c, const void *S, int64_t n_sub,
                              const int64_t *rel_idx, const int64_t *sub_idx, int64_t batch, float *v_out,
                              void *q_packed, const RtkWorkspace &ws, hipStream_t st, int rel_part, int rel_parts) {
    return from_tables_impl<rtk_bf16>(tables, n_rel, b, c, (const rtk_bf16 *)S, n_sub, rel_idx, sub_idx, batch, v_out,
                                      q_packed, ws, st, rel_part, rel_parts);
}

extern "C" int rtk_pack_query_vectors(const float *v, int64_t batch, int c, int dtype, void *q_packed, void *stream) {
    RTK_REQUIRE(v && q_packed, RTK_ERR_BAD_ARG, "rtk_pack_query_vectors: null operand");
    RTK_REQUIRE(batch > 0 && c > 0 && batch < (1ll << 31), RTK_ERR_BAD_ARG, "rtk_pack_query_vectors: bad sizes");
    RTK_REQUIRE(dtype == RTK_F32 || dtype == RTK_BF16, RTK_ERR_BAD_ARG, "rtk_pack_query_vectors: dtype");
    RTK_REQUIRE(c <= 512, RTK_ERR_UNSUPPORTED, "rtk_pack_query_vectors: c=%d > 512 has no packed score kernel", c);
    hipStream_t st = (hipStream_t)stream;
    const int ks = (c + 15) / 16;
    // rows of the last tile beyond `batch` stay unwritten: the score kernels never store them
    if (dtype == RTK_BF16) hipLaunchKernelGGL(pack_rows_kernel<true>, dim3((unsigned)batch), dim3(256), 0, st, v, c, ks, (unsigned char *)q_packed);
    else hipLaunchKernelGGL(pack_rows_kernel<false>, dim3((unsigned)batch), dim3(256), 0, st, v, c, ks, (unsigned char *)q_packed);
    return rtk_check_launch("rtk_pack_query_vectors");
}
