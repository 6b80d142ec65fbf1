// Training loss without dense targets.  The reference builds, per item, a dense
// N-vector of label-smoothed targets on the host (src/data/Dataset.py:43-53:
// y = (1 - eps) * multi_hot + eps / N), ships B x N floats per batch to the device and lets
// nn.BCELoss read scores and targets again (train.py:76-82, :136).  Here the targets exist only as
// the CSR of known objects per (subject, relation) pair, uploaded once:
//   rtk_bce_rows_f32 : per-row sum of BCE terms           (forward:  loss = sum / (B * N))
//   rtk_bce_grad_f32 : P <- (P - y) * g * scale, in place (backward: d loss / d logits, because
//                      d BCE / d z = p - y for p = sigmoid(z)), ready for the dO / dv GEMMs.
//                      Where the fp32 score is SATURATED (p == 1.0f for z >~ 16.64, p == 0.0f) the result is 0:
//                      that is what the reference's autograd returns there (BCELoss' backward divides
//                      by max(p (1 - p), 1e-12) and the logistic's backward multiplies by p (1 - p) = 0),
//                      and a trained model does saturate.
// Both are one pass over the B x N scores plus a pass over the few positives.
// The CSR must hold every object of a pair ONCE (evaluation.DeviceFilter de-duplicates).
#include "rtk_common.h"

namespace {

// ln x on v_log_f32 (log2, 1 ulp) -- the row sums are compute-bound on ocml's logf otherwise (62 us vs the
// 15 us the 84 MB read takes at C2); torch clamps BCE's logs at -100
__device__ __forceinline__ float clog(float x) { return fmaxf(__builtin_amdgcn_logf(x) * 0.6931471805599453f, -100.0f); }

// one workgroup per row
__global__ __launch_bounds__(256) void bce_rows_kernel(const float *__restrict__ P, int N, int64_t ld, float t0, float dt,
                                                       const int64_t *__restrict__ pair_slot,
                                                       const int64_t *__restrict__ pair_ptr,
                                                       const int64_t *__restrict__ pair_obj,
                                                       double *__restrict__ rows) {
    __shared__ double s_sum[4];
    const int d = blockIdx.x, t = threadIdx.x;
    const float *row = P + (int64_t)d * ld;
    // every entity as a smoothed negative: y = t0 = eps / N
    float acc = 0.f;
    constexpr int U = 8;                               // loads in flight per thread (the pass is latency-bound otherwise)
    int j = t;
    for (; j + 256 * (U - 1) < N; j += 256 * U) {
        float p[U];
#pragma unroll
        for (int u = 0; u < U; ++u) p[u] = row[j + 256 * u];
#pragma unroll
        for (int u = 0; u < U; ++u) acc += t0 * clog(p[u]) + (1.0f - t0) * clog(1.0f - p[u]);
    }
    for (; j < N; j += 256) {
        const float p = row[j];
        acc += t0 * clog(p) + (1.0f - t0) * clog(1.0f - p);
    }
    // the pair's known objects: y = t0 + dt, dt = 1 - eps
    const int64_t s = pair_slot[d];
    for (int64_t i = pair_ptr[s] + t; i < pair_ptr[s + 1]; i += 256) {
        const int64_t j = pair_obj[i];
        if (j < 0 || j >= N) continue;
        const float p = row[j];
        acc += dt * (clog(p) - clog(1.0f - p));
    }
    double a = (double)acc;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
    if ((t & 63) == 0) s_sum[t >> 6] = a;
    __syncthreads();
    if (t == 0) rows[d] = -(s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3]);
}

// Pass 1, the positives (their target is t0 + dt): p <- p - dt unless p is saturated (p == 1.0f stays 1.0f and
// is then zeroed by pass 2 like a saturated negative; an unsaturated positive lands in (-dt, 1 - dt), never
// on 1.0f).  One workgroup per row.
__global__ __launch_bounds__(64) void bce_grad_pos_kernel(float *__restrict__ P, int N, int64_t ld, float dt,
                                                          const int64_t *__restrict__ pair_slot,
                                                          const int64_t *__restrict__ pair_ptr,
                                                          const int64_t *__restrict__ pair_obj) {
    const int d = blockIdx.x;
    float *row = P + (int64_t)d * ld;
    const int64_t sl = pair_slot[d];
    for (int64_t i = pair_ptr[sl] + threadIdx.x; i < pair_ptr[sl + 1]; i += 64) {
        const int64_t j = pair_obj[i];
        if (j >= 0 && j < N) {
            const float p = row[j];
            if (p != 1.0f && p != 0.0f) row[j] = p - dt;
        }
    }
}

// Pass 2, every element: x <- (x - t0) * g * scale, or 0 where the score was saturated (x == 1.0f / 0.0f)
// (grid-stride over rows x column chunks)
__device__ __forceinline__ float bce_grad_one(float x, float t0, float s) {
    return (x == 1.0f || x == 0.0f) ? 0.0f : (x - t0) * s;
}
template <bool VEC>
__global__ __launch_bounds__(256) void bce_grad_all_kernel(float *__restrict__ P, int B, int N, int64_t ld, float t0,
                                                           const float *__restrict__ g, float scale) {
    const float s = g[0] * scale;
    const int d = blockIdx.y;
    float *row = P + (int64_t)d * ld;
    if (VEC) {
        const int n4 = N >> 2;
        for (int q = blockIdx.x * 256 + threadIdx.x; q < n4; q += gridDim.x * 256) {
            f32x4 x = reinterpret_cast<f32x4 *>(row)[q];
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e] = bce_grad_one(x[e], t0, s);
            reinterpret_cast<f32x4 *>(row)[q] = x;
        }
        for (int j = (n4 << 2) + blockIdx.x * 256 + threadIdx.x; j < N; j += gridDim.x * 256) row[j] = bce_grad_one(row[j], t0, s);
    } else {
        for (int j = blockIdx.x * 256 + threadIdx.x; j < N; j += gridDim.x * 256) row[j] = bce_grad_one(row[j], t0, s);
    }
}

// The positives of the fused training forward (rtk_score_packed_bce_f32 wrote x = p - t0, or a signed zero where p was
// saturated, and summed every entry's BCE term as a negative): x <- x - dt, and the row's correction of the loss,
// -dt (ln p - ln(1 - p)).  p cannot be read back from x (p - t0 has lost a p far below t0, and a positive with a very
// negative logit is exactly where ln p matters), so the logit of each positive is recomputed from the fp32 query vector
// and entity row -- one wave per row, the 64 lanes share each dot product -- and p from it by the kernel's own formula.  rows_pos holds PATCH_Y = 4 partial sums per row.
// A stored zero is a score the kernel saw saturated (+0: 1.0f, -0: 0.0f): it stays zero (zero logit gradient, like the
// reference's autograd) and its correction uses torch's clamp, ln 0 = -100.
constexpr int PATCH_Y = 4;      // workgroups per row; 4 waves each: 16 positives of a row in flight (hub pairs have hundreds)
__global__ __launch_bounds__(256) void bce_patch_pos_kernel(float *__restrict__ X, int N, int64_t ld, float t0, float dt,
                                                            const int64_t *__restrict__ pair_slot,
                                                            const int64_t *__restrict__ pair_ptr,
                                                            const int64_t *__restrict__ pair_obj,
                                                            const float *__restrict__ v, const float *__restrict__ O, int c,
                                                            double *__restrict__ rows_pos) {
    __shared__ double s_sum[4];
    const int d = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *row = X + (int64_t)d * ld;
    const float *vd = v + (int64_t)d * c;
    const int64_t sl = pair_slot[d];
    float acc = 0.f;
    for (int64_t i = pair_ptr[sl] + blockIdx.y * 4 + wave; i < pair_ptr[sl + 1]; i += 4 * PATCH_Y) {
        const int64_t j = pair_obj[i];
        if (j < 0 || j >= N) continue;                   // (uniform over the wave)
        const float *oj = O + j * c;
        float z = 0.f;
        for (int k = lane; k < c; k += 64) z += vd[k] * oj[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) z += __shfl_xor(z, o);
        if (lane == 0) {
            const float x = row[j];
            if (x == 0.0f) {                             // saturated in the kernel: +0 = score 1.0f, -0 = score 0.0f
                acc += (__builtin_bit_cast(unsigned, x) >> 31) ? -dt * 100.0f : dt * 100.0f;
            } else {
                const float p = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z * -1.4426950408889634f));
                acc += dt * (clog(p) - clog(1.0f - p));
                row[j] = x - dt;
            }
        }
    }
    if (lane == 0) s_sum[wave] = (double)acc;
    __syncthreads();
    if (threadIdx.x == 0) rows_pos[(int64_t)d * PATCH_Y + blockIdx.y] = -(s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3]);
}

int check(const char *fn, const float *P, int64_t batch, int64_t n_ent, int64_t ld, const int64_t *pair_slot,
          const int64_t *pair_ptr, const int64_t *pair_obj, float eps) {
    RTK_REQUIRE(P && pair_slot && pair_ptr && pair_obj, RTK_ERR_BAD_ARG, "%s: null operand", fn);
    RTK_REQUIRE(batch > 0 && n_ent > 0 && ld >= n_ent, RTK_ERR_BAD_ARG, "%s: bad sizes", fn);
    RTK_REQUIRE(eps >= 0.f && eps < 1.f, RTK_ERR_BAD_ARG, "%s: label smoothing %g outside [0, 1)", fn, (double)eps);
    RTK_REQUIRE(batch < 65536 * 16384ll && n_ent < (1ll << 31), RTK_ERR_UNSUPPORTED, "%s: dimension too large", fn);
    return RTK_OK;
}

}  // namespace

extern "C" int rtk_bce_rows_f32(const float *P, int64_t batch, int64_t n_ent, int64_t ld, const int64_t *pair_slot,
                                const int64_t *pair_ptr, const int64_t *pair_obj, float label_smoothing,
                                double *rows_out, void *stream) {
    int rc = check("rtk_bce_rows_f32", P, batch, n_ent, ld, pair_slot, pair_ptr, pair_obj, label_smoothing);
    if (rc != RTK_OK) return rc;
    RTK_REQUIRE(rows_out, RTK_ERR_BAD_ARG, "rtk_bce_rows_f32: null output");
    RTK_REQUIRE(batch < (1ll << 31), RTK_ERR_UNSUPPORTED, "rtk_bce_rows_f32: batch too large");
    const float t0 = label_smoothing / (float)n_ent, dt = 1.0f - label_smoothing;
    hipLaunchKernelGGL(bce_rows_kernel, dim3((unsigned)batch), dim3(256), 0, (hipStream_t)stream, P, (int)n_ent, ld, t0, dt,
                       pair_slot, pair_ptr, pair_obj, rows_out);
    return rtk_check_launch("rtk_bce_rows_f32");
}

extern "C" int rtk_bce_grad_f32(float *P, int64_t batch, int64_t n_ent, int64_t ld, const int64_t *pair_slot,
                                const int64_t *pair_ptr, const int64_t *pair_obj, float label_smoothing,
                                const float *grad_loss, float scale, void *stream) {
    int rc = check("rtk_bce_grad_f32", P, batch, n_ent, ld, pair_slot, pair_ptr, pair_obj, label_smoothing);
    if (rc != RTK_OK) return rc;
    RTK_REQUIRE(grad_loss, RTK_ERR_BAD_ARG, "rtk_bce_grad_f32: null grad_loss");
    RTK_REQUIRE(batch <= 65535, RTK_ERR_UNSUPPORTED, "rtk_bce_grad_f32: batch > 65535");
    const float t0 = label_smoothing / (float)n_ent, dt = 1.0f - label_smoothing;
    hipStream_t st = (hipStream_t)stream;
    const bool vec = (ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(P) & 15) == 0);
    const unsigned gx = (unsigned)(rtk_cdiv(n_ent, 1024 * 4) < 1 ? 1 : rtk_cdiv(n_ent, 1024 * 4));
    dim3 grid(gx < 64 ? gx : 64, (unsigned)batch);
    hipLaunchKernelGGL(bce_grad_pos_kernel, dim3((unsigned)batch), dim3(64), 0, st, P, (int)n_ent, ld, dt, pair_slot, pair_ptr,
                       pair_obj);
    if (vec) hipLaunchKernelGGL((bce_grad_all_kernel<true>), grid, dim3(256), 0, st, P, (int)batch, (int)n_ent, ld, t0, grad_loss, scale);
    else hipLaunchKernelGGL((bce_grad_all_kernel<false>), grid, dim3(256), 0, st, P, (int)batch, (int)n_ent, ld, t0, grad_loss, scale);
    return rtk_check_launch("rtk_bce_grad_f32");
}

extern "C" int rtk_bce_patch_pos_f32(float *X, int64_t batch, int64_t n_ent, int64_t ld, const int64_t *pair_slot,
                                     const int64_t *pair_ptr, const int64_t *pair_obj, float label_smoothing,
                                     const float *v, const float *O, int c, double *rows_pos_out, void *stream) {
    int rc = check("rtk_bce_patch_pos_f32", X, batch, n_ent, ld, pair_slot, pair_ptr, pair_obj, label_smoothing);
    if (rc != RTK_OK) return rc;
    RTK_REQUIRE(rows_pos_out && v && O && c > 0, RTK_ERR_BAD_ARG, "rtk_bce_patch_pos_f32: null operand");
    RTK_REQUIRE(batch < (1ll << 31) / 4, RTK_ERR_UNSUPPORTED, "rtk_bce_patch_pos_f32: batch too large");
    const float t0 = label_smoothing / (float)n_ent, dt = 1.0f - label_smoothing;
    hipLaunchKernelGGL(bce_patch_pos_kernel, dim3((unsigned)batch, PATCH_Y), dim3(256), 0, (hipStream_t)stream, X, (int)n_ent, ld,
                       t0, dt, pair_slot, pair_ptr, pair_obj, v, O, c, rows_pos_out);
    return rtk_check_launch("rtk_bce_patch_pos_f32");
}
