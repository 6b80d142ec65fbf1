// Filtered ranking on the device (the eval tail of the reference, train.py:113-117):
//   filter_predictions (src/utils/utils.py:15-22): scores of the OTHER known-true objects of a
//     query are replaced by 0, the queried object keeps its score;
//   metrics (src/utils/metrics.py:4-22): rank of the queried object in the descending sort.
// The reference does a full sort of B x N (150 ms per batch on CPU).  Here one
// workgroup per query counts, in a single pass over its score row,
//     rank = 1 + #{ j : p'_j > p_t } + #{ j < t : p'_j == p_t }
// i.e. the position in a STABLE descending sort (what torch's CUDA radix sort and
// sort(stable=True) produce; torch's default CPU sort is unstable, so the reference's own
// tie order on CPU is unspecified).  Optionally it also returns the row's BCE sum
// (nn.BCELoss with 0/1 targets, logs clamped at -100 like torch), train.py:113.
//
// Entity-sharded form ("better than gather"): a rank holds the columns
// [col0, col0 + N) of the score matrix.  The count above is a sum over columns, so each rank counts
// over its own block against the target's score (owned by one rank: rtk_target_scores_f32 +
// all-reduce MAX) and the B partial counts are all-reduced -- B x 12 bytes cross the links instead
// of B x N x 4.  The single-device entry point is the col0 = 0 case of the same kernel.
#include "rtk_common.h"

namespace {

// ln x on v_log_f32 (log2, 1 ulp); torch clamps BCE's logs at -100
__device__ __forceinline__ float clog(float x) { return fmaxf(__builtin_amdgcn_logf(x) * 0.6931471805599453f, -100.0f); }

// PARTIAL: target scores come from pt_in, the "+1" is left to the caller, ids are global
template <bool PARTIAL>
__global__ __launch_bounds__(256) void filtered_rank_kernel(
    const float *__restrict__ P, int B, int N, int64_t ld, int64_t col0, const float *__restrict__ pt_in,
    const int64_t *__restrict__ obj_idx,
    const int64_t *__restrict__ pair_slot, const int64_t *__restrict__ pair_ptr,
    const int64_t *__restrict__ pair_obj, int32_t *__restrict__ ranks, double *__restrict__ bce_rows) {
    __shared__ int s_gt[4], s_eq[4];
    __shared__ float s_bce[4];
    const int d = blockIdx.x, t = threadIdx.x;
    const float *row = P + (int64_t)d * ld;
    int64_t tgt = obj_idx[d] - col0;            // local column of the queried object (outside [0, N): another rank's)
    if (!PARTIAL) tgt = tgt < 0 ? 0 : (tgt >= N ? N - 1 : tgt);
    const bool own = tgt >= 0 && tgt < N;
    const float pt = PARTIAL ? pt_in[d] : row[tgt];
    int gt = 0, eq = 0;
    float bce = 0.f;
    const bool want_bce = bce_rows != nullptr;
    constexpr int U = 8;                               // loads in flight per thread (the pass is latency-bound otherwise)
    int j = t;
    for (; j + 256 * (U - 1) < N; j += 256 * U) {
        float p[U];
#pragma unroll
        for (int u = 0; u < U; ++u) p[u] = row[j + 256 * u];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            gt += p[u] > pt;
            eq += (p[u] == pt) & (j + 256 * u < tgt);
            if (want_bce) bce += clog(1.0f - p[u]);
        }
    }
    for (; j < N; j += 256) {
        const float p = row[j];
        gt += p > pt;
        eq += (p == pt) & (j < tgt);
        if (want_bce) bce += clog(1.0f - p);
    }
    // the query's other true objects count as score 0 (and, for BCE, as positives)
    const int64_t s = pair_slot ? pair_slot[d] : -1;
    if (s >= 0) {
        for (int64_t i = pair_ptr[s] + t; i < pair_ptr[s + 1]; i += 256) {
            const int64_t j = pair_obj[i] - col0;
            if (j < 0 || j >= N) continue;
            const float p = row[j];
            if (want_bce) bce += clog(p) - clog(1.0f - p);
            if (j == tgt) continue;
            gt -= p > pt;
            eq -= (p == pt) & (j < tgt);
            eq += (0.0f == pt) & (j < tgt);     // now a 0: ties only with a zero target score
        }
    } else if (want_bce && t == 0 && own) {     // no filter list: the queried object is the only positive
        bce += clog(pt) - clog(1.0f - pt);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        gt += __shfl_xor(gt, o);
        eq += __shfl_xor(eq, o);
        bce += __shfl_xor(bce, o);
    }
    if ((t & 63) == 0) {
        s_gt[t >> 6] = gt;
        s_eq[t >> 6] = eq;
        s_bce[t >> 6] = bce;
    }
    __syncthreads();
    if (t == 0) {
        ranks[d] = (PARTIAL ? 0 : 1) + s_gt[0] + s_gt[1] + s_gt[2] + s_gt[3] + s_eq[0] + s_eq[1] + s_eq[2] + s_eq[3];
        if (want_bce) bce_rows[d] = -((double)s_bce[0] + (double)s_bce[1] + (double)s_bce[2] + (double)s_bce[3]);
    }
}

__global__ __launch_bounds__(256) void target_scores_kernel(const float *__restrict__ P, int B, int N, int64_t ld,
                                                            int64_t col0, const int64_t *__restrict__ obj_idx,
                                                            float *__restrict__ pt_out) {
    const int d = blockIdx.x * 256 + threadIdx.x;
    if (d >= B) return;
    const int64_t j = obj_idx[d] - col0;
    pt_out[d] = (j >= 0 && j < N) ? P[(int64_t)d * ld + j] : -INFINITY;
}

// acc[0] += sum 1/rank, acc[1..3] += #(rank <= 1 / 3 / 10), acc[4] += sum of the BCE row sums
// (src/utils/metrics.py:4-22 returns exactly these batch sums; train.py:118-121 adds them up)
__global__ __launch_bounds__(256) void rank_metrics_kernel(const int32_t *__restrict__ ranks,
                                                           const double *__restrict__ bce_rows, int B,
                                                           double *__restrict__ acc, double bce_scale) {
    __shared__ double s[4][5];
    const int t = threadIdx.x;
    double v[5] = {0, 0, 0, 0, 0};
    for (int d = t; d < B; d += 256) {
        const int rk = ranks[d];
        v[0] += 1.0 / (double)rk;
        v[1] += rk <= 1;
        v[2] += rk <= 3;
        v[3] += rk <= 10;
        if (bce_rows) v[4] += bce_rows[d] * bce_scale;
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_xor(v[k], o);
        if ((t & 63) == 0) s[t >> 6][k] = v[k];
    }
    __syncthreads();
    if (t < 5) acc[t] += s[0][t] + s[1][t] + s[2][t] + s[3][t];   // one workgroup, launches are stream-ordered: no atomics needed
}

}  // namespace

extern "C" int rtk_rank_metrics_f64(const int32_t *ranks, const double *bce_rows, int64_t batch, double *acc5,
                                    void *stream) {
    RTK_REQUIRE(ranks && acc5 && batch > 0 && batch < (1ll << 31), RTK_ERR_BAD_ARG, "rtk_rank_metrics_f64: bad argument");
    hipLaunchKernelGGL(rank_metrics_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, ranks, bce_rows, (int)batch, acc5, 1.0);
    return rtk_check_launch("rtk_rank_metrics_f64");
}

extern "C" int rtk_rank_metrics_scaled_f64(const int32_t *ranks, const double *bce_rows, int64_t batch, double bce_scale,
                                           double *acc5, void *stream) {
    RTK_REQUIRE(ranks && acc5 && batch > 0 && batch < (1ll << 31), RTK_ERR_BAD_ARG, "rtk_rank_metrics_scaled_f64: bad argument");
    hipLaunchKernelGGL(rank_metrics_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, ranks, bce_rows, (int)batch, acc5, bce_scale);
    return rtk_check_launch("rtk_rank_metrics_scaled_f64");
}

extern "C" int rtk_target_scores_f32(const float *P, int64_t batch, int64_t n_local, int64_t ld, int64_t col0,
                                     const int64_t *obj_idx, float *pt_out, void *stream) {
    RTK_REQUIRE(P && obj_idx && pt_out, RTK_ERR_BAD_ARG, "rtk_target_scores_f32: null operand");
    RTK_REQUIRE(batch > 0 && n_local > 0 && ld >= n_local && col0 >= 0, RTK_ERR_BAD_ARG, "rtk_target_scores_f32: bad sizes");
    RTK_REQUIRE(batch < (1ll << 31) && n_local < (1ll << 31), RTK_ERR_UNSUPPORTED, "rtk_target_scores_f32: dimension too large");
    hipLaunchKernelGGL(target_scores_kernel, dim3((unsigned)rtk_cdiv(batch, 256)), dim3(256), 0, (hipStream_t)stream, P,
                       (int)batch, (int)n_local, ld, col0, obj_idx, pt_out);
    return rtk_check_launch("rtk_target_scores_f32");
}

extern "C" int rtk_filtered_rank_partial_f32(const float *P, int64_t batch, int64_t n_local, int64_t ld, int64_t col0,
                                             const float *target_scores, const int64_t *obj_idx,
                                             const int64_t *pair_slot, const int64_t *pair_ptr, const int64_t *pair_obj,
                                             int32_t *counts_out, double *bce_rows_out, void *stream) {
    RTK_REQUIRE(P && obj_idx && counts_out && target_scores, RTK_ERR_BAD_ARG, "rtk_filtered_rank_partial_f32: null operand");
    RTK_REQUIRE(batch > 0 && n_local > 0 && ld >= n_local && col0 >= 0, RTK_ERR_BAD_ARG, "rtk_filtered_rank_partial_f32: bad sizes");
    RTK_REQUIRE(!pair_slot || (pair_ptr && pair_obj), RTK_ERR_BAD_ARG, "rtk_filtered_rank_partial_f32: pair_slot without the CSR arrays");
    RTK_REQUIRE(batch < (1ll << 31) && n_local < (1ll << 31), RTK_ERR_UNSUPPORTED, "rtk_filtered_rank_partial_f32: dimension too large");
    hipLaunchKernelGGL(filtered_rank_kernel<true>, dim3((unsigned)batch), dim3(256), 0, (hipStream_t)stream, P, (int)batch,
                       (int)n_local, ld, col0, target_scores, obj_idx, pair_slot, pair_ptr, pair_obj, counts_out, bce_rows_out);
    return rtk_check_launch("rtk_filtered_rank_partial_f32");
}

extern "C" int rtk_filtered_rank_f32(const float *P, int64_t batch, int64_t n_ent, int64_t ld,
                                     const int64_t *obj_idx, const int64_t *pair_slot, const int64_t *pair_ptr,
                                     const int64_t *pair_obj, int32_t *ranks_out, double *bce_rows_out,
                                     void *stream) {
    RTK_REQUIRE(P && obj_idx && ranks_out, RTK_ERR_BAD_ARG, "rtk_filtered_rank_f32: null operand");
    RTK_REQUIRE(batch > 0 && n_ent > 0 && ld >= n_ent, RTK_ERR_BAD_ARG, "rtk_filtered_rank_f32: bad sizes");
    RTK_REQUIRE(!pair_slot || (pair_ptr && pair_obj), RTK_ERR_BAD_ARG, "rtk_filtered_rank_f32: pair_slot without the CSR arrays");
    RTK_REQUIRE(batch < (1ll << 31) && n_ent < (1ll << 31), RTK_ERR_UNSUPPORTED, "rtk_filtered_rank_f32: dimension too large");
    hipLaunchKernelGGL(filtered_rank_kernel<false>, dim3((unsigned)batch), dim3(256), 0, (hipStream_t)stream, P, (int)batch,
                       (int)n_ent, ld, (int64_t)0, (const float *)nullptr, obj_idx, pair_slot, pair_ptr, pair_obj, ranks_out,
                       bce_rows_out);
    return rtk_check_launch("rtk_filtered_rank_f32");
}
