/*
 * rtucker_hip.h -- C ABI of librtucker_hip.so: R-TuckER's 1-vs-all Tucker scoring
 * path for AMD MI355X (gfx950 / CDNA4), hand-written HIP kernels.
 *
 * The reference (johanDDC/R-TuckER) has no native boundary: the path is the
 * Python closure  score_fn = model(subject_idx, relation_idx);  P = score_fn(T)
 * (src/model/asymmetric/R_TuckER.py:41-50, src/model/symmetric/R_TuckER.py:38-47).
 * This header is the boundary a maintainer binds instead of the five torch ops
 * inside that closure (INTEGRATION.md shows the ctypes stub).  Each entry point
 * below cites the reference lines it replaces.
 *
 * Conventions
 *  - plain C symbols, no C++/torch types; every pointer is a DEVICE pointer
 *    (hipMalloc'd / torch tensor .data_ptr()) unless the name says "host";
 *  - row-major, contiguous operands; sizes are element counts;
 *  - core axis order is (relation a, subject b, object c)  [train.py:41,
 *    asymmetric/R_TuckER.py:20-23];  factors R:(nR,a)  S:(nS,b)  O:(N,c);
 *  - `stream` is a hipStream_t passed as void* (0 = the null stream); every call
 *    only ENQUEUES work on it: no allocation, no synchronisation, no hidden
 *    state besides a thread-local last-error string -> graph-capturable;
 *  - scratch memory is caller-owned: query its size with rtk_workspace_bytes(),
 *    pass it in; the same workspace must not be used by two in-flight calls;
 *  - return value: 0 = RTK_OK, negative = rtk_status (nothing was enqueued).
 *    Index values are range-checked ON DEVICE: an out-of-range id never reads
 *    out of bounds (it is clamped) and raises bit 0 of the 32-bit word at the
 *    start of the workspace, which rtk_read_error_flag() fetches (synchronising).
 *    (Reference behaviour: torch raises IndexError / device assert.)
 */
#ifndef RTUCKER_HIP_H
#define RTUCKER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum rtk_status {
    RTK_OK = 0,
    RTK_ERR_BAD_ARG = -1,      /* null pointer, non-positive size, b != c ...        */
    RTK_ERR_WORKSPACE = -2,    /* workspace missing or smaller than required        */
    RTK_ERR_UNSUPPORTED = -3,  /* shape outside what the kernels implement          */
    RTK_ERR_LAUNCH = -4        /* hipGetLastError() after a launch was not success  */
} rtk_status;

typedef enum rtk_dtype {
    RTK_F32 = 0,   /* fp32 operands, fp32 scores (reference dtype, R_TuckER.py:23)   */
    RTK_BF16 = 1   /* bf16 operands, fp32 accumulate, fp32 scores                    */
} rtk_dtype;

/* flags for the score stage */
#define RTK_SCORE_SIGMOID 1u      /* apply sigmoid (R_TuckER.py:48); else raw logits  */
#define RTK_SCORE_EXACT_F32 2u    /* fp32 operands: force the exact-fp32 MFMA kernel   */
                                  /* instead of the split-fp16 (hi/lo) MFMA kernel     */
/*
 * Precision of the default fp32 score kernel (rtk_score_packed_f32, "split-fp16"): every operand row
 * (a query vector v_d, an entity row O[j]) is scaled by a power of two so that its LARGEST element
 * lands in [2^14, 2^15) and each element is split x = hi + lo into two fp16 values (~22 significand
 * bits relative to the ROW MAXIMUM); v.o ~= vh.oh + vh.ol + vl.oh, accumulated in fp32 (the vl.ol
 * term, 2^-22 relative, is dropped).  The guarantee is therefore NORMWISE PER ROW, not element-wise:
 *     |dz[d,j]|  <~  2^-21 * K * max|v_d| * max|O_j|
 * -- elements much smaller than their row's maximum lose RELATIVE precision: an element keeps all ~22
 * bits down to 2^-17 of the row maximum, then one bit less per factor of two (fp16 subnormals) to 11
 * bits at 2^-28, and is dropped entirely below 2^-39 of the row maximum -- which
 * does not matter for a dot product dominated by the large elements but is not what an exact fp32
 * product gives for rows of extreme dynamic range.  Measured: the same error against float64 as the
 * reference's CPU sgemm at rank (10,200,200) (3-5e-6 relative; tests/test_gpu_parity.py,
 * test_wide_dynamic_range_rows).  RTK_SCORE_EXACT_F32 selects v_mfma_f32_32x32x2_f32 (bitwise an
 * fmaf chain) at 1/5 of the throughput when element-wise fp32 behaviour is required.
 */
#define RTK_SCORE_SIGMOID_FAST 4u /* with RTK_SCORE_SIGMOID: 1 / (1 + 2^(-z log2 e)) on  */
                                  /* v_exp_f32 + v_rcp_f32 (1 ulp each) instead of expf */
                                  /* + IEEE divide                                      */
#define RTK_SCORE_OUT_BF16 8u     /* bf16 operand entry points, with SIGMOID | SIGMOID_FAST:   */
                                  /* `out` holds bf16 scores (what the reference's bf16 model */
                                  /* returns), ld_out counts bf16 elements; pass the pointer  */
                                  /* through the float* parameter                              */

/* Which fp32 split-fp16 score kernel runs (rtk_score_packed_f32, rtk_score_1vN_f32); 0 = by shape.  The kernels
 * give the same scores up to the summation order inside a dot product; the hints exist for A/B runs and so that
 * the tests can put every kernel on every shape.  A hinted kernel that does not cover the shape falls through to
 * the next one (cg: c <= 208, c % 4 == 0; ws: the same; v3: c <= 512). */
#define RTK_SCORE_KERNEL_MASK 0x300u
#define RTK_SCORE_KERNEL_CG 0x100u  /* column-group kernel (csrc/rtk_score_cg_kernel.h), on any entity count    */
#define RTK_SCORE_KERNEL_WS 0x200u  /* wave-specialised persistent kernel (csrc/rtk_score_ws_kernel.h)           */
#define RTK_SCORE_KERNEL_V3 0x300u  /* two workgroups per CU, every wave does everything (rtk_score_split_kernel.h) */

int rtk_version(void);
const char *rtk_last_error_string(void);

/* Bytes of workspace the calls below need for these sizes (upper bound, 256-B aligned). */
size_t rtk_workspace_bytes(int dtype, int64_t batch, int64_t n_rel, int a, int b, int c);

/* Fetches and clears the device error word of a workspace (hipStreamSynchronize
 * on `stream` first).  Bit 0: relation/subject index out of range. */
int rtk_read_error_flag(void *workspace, void *stream, uint32_t *host_flag_out);

/*
 * Stage 1: query vectors  v[d,:] = S[h_d,:] . ( G x_0 R[r_d,:] )        (B x c)
 * replaces asymmetric/R_TuckER.py:43-46 (two gathers, einsum "abc,da->dbc", bmm)
 * and symmetric/R_TuckER.py:40-43 (pass E as S).  Requires b == c like the
 * reference's .view(-1, b) -> RTK_ERR_BAD_ARG otherwise.
 *   v_out    : fp32 (B x c), may be NULL
 *   q_packed : "query planes" consumed by rtk_score_packed_*, may be NULL;
 *              size rtk_packed_query_bytes(dtype, B, c)
 * Internally: relation tables M_u = G x_0 R[u] for the distinct relations of the
 * batch, then v_d = S[h_d] . M_{r_d}  (the reference's summation order).
 */
int rtk_query_vectors_f32(const float *core, int a, int b, int c,
                          const float *R, int64_t n_rel,
                          const float *S, int64_t n_sub,
                          const int64_t *rel_idx, const int64_t *sub_idx, int64_t batch,
                          float *v_out, void *q_packed,
                          void *workspace, size_t workspace_bytes, void *stream);

size_t rtk_packed_query_bytes(int dtype, int64_t batch, int c);

/*
 * Packed query planes from fp32 query vectors computed elsewhere (entity-sharded scoring with stage 1
 * split over the ranks: each rank contracts a slice of the batch, the (batch x c) vectors are
 * all-gathered, every rank packs them and scores its entity shard with rtk_score_packed_*).  Same
 * arithmetic as the packing inside rtk_query_vectors_* (bit-identical planes).  dtype: rtk_dtype of
 * the score kernel that will consume them.
 */
int rtk_pack_query_vectors(const float *v, int64_t batch, int c, int dtype, void *q_packed, void *stream);

/*
 * Stage 1 split at the relation tables, for callers whose parameters stay fixed over many batches
 * (the evaluation loop, train.py:107-121: extract_tensor(model) is the same tensor for every batch).
 * The einsum of asymmetric/R_TuckER.py:45 applied to ALL relation rows,
 *     tables[u, :, :] = sum_a R[u, a] * G[a, :, :]          (n_rel x b x c, fp32)
 * depends on the parameters only:
 *   rtk_relation_tables_{f32,bf16}            build it once per parameter version
 *                                             (bytes: rtk_relation_tables_bytes; scratch:
 *                                             rtk_relation_tables_workspace_bytes, >= 256)
 *   rtk_query_vectors_from_tables_{f32,bf16}  per batch: v_d = S[h_d] . tables[r_d]   (R_TuckER.py:43-46
 *                                             given the tables); same outputs as rtk_query_vectors_*;
 *                                             workspace rtk_from_tables_workspace_bytes, whose first
 *                                             word is the sticky error word like the main workspace's.
 * Same summation order as rtk_query_vectors_*: the results are bit-identical to the uncached path.
 * `tables` must be 256-byte aligned.
 */
size_t rtk_relation_tables_bytes(int64_t n_rel, int b, int c);
size_t rtk_relation_tables_workspace_bytes(int dtype, int64_t n_rel, int a, int b, int c);
size_t rtk_from_tables_workspace_bytes(int64_t batch, int64_t n_rel);

int rtk_relation_tables_f32(const float *core, int a, int b, int c, const float *R, int64_t n_rel,
                            float *tables, void *workspace, size_t workspace_bytes, void *stream);
int rtk_relation_tables_bf16(const void *core, int a, int b, int c, const void *R, int64_t n_rel,
                             float *tables, void *workspace, size_t workspace_bytes, void *stream);

int rtk_query_vectors_from_tables_f32(const float *tables, int64_t n_rel, int b, int c,
                                      const float *S, int64_t n_sub,
                                      const int64_t *rel_idx, const int64_t *sub_idx, int64_t batch,
                                      float *v_out, void *q_packed,
                                      void *workspace, size_t workspace_bytes, void *stream);
int rtk_query_vectors_from_tables_bf16(const float *tables, int64_t n_rel, int b, int c,
                                       const void *S, int64_t n_sub,
                                       const int64_t *rel_idx, const int64_t *sub_idx, int64_t batch,
                                       float *v_out, void *q_packed,
                                       void *workspace, size_t workspace_bytes, void *stream);

/*
 * The same step restricted to the queries whose relation id is congruent to `part` modulo `n_parts`: their rows of
 * v_out (B x c fp32, required) are written, every other row is left untouched.  This is stage 1 split over the ranks
 * of an entity-sharded run BY RELATION: a rank then streams only the tables of its own relations
 * (1/n_parts of rtk_relation_tables_bytes instead of nearly all of it when the batch is split by position: at
 * BASELINE.json configs[4], 8192 queries over 1000 relations, a batch slice of 1024 queries still touches ~640
 * tables of 1 MB); the B x c vectors are completed by one all-reduce(SUM) over buffers zeroed before the call --
 * every row is non-zero on exactly one rank, so the sum is exact -- and packed with rtk_pack_query_vectors.
 */
int rtk_query_vectors_from_tables_part_f32(const float *tables, int64_t n_rel, int b, int c,
                                           const float *S, int64_t n_sub,
                                           const int64_t *rel_idx, const int64_t *sub_idx, int64_t batch,
                                           int part, int n_parts, float *v_out,
                                           void *workspace, size_t workspace_bytes, void *stream);
int rtk_query_vectors_from_tables_part_bf16(const float *tables, int64_t n_rel, int b, int c,
                                            const void *S, int64_t n_sub,
                                            const int64_t *rel_idx, const int64_t *sub_idx, int64_t batch,
                                            int part, int n_parts, float *v_out,
                                            void *workspace, size_t workspace_bytes, void *stream);

/*
 * Stage 2: scores  out[d, j] = sigmoid( v[d,:] . O[j,:] )   for j < n_local
 * replaces asymmetric/R_TuckER.py:47-48 ( @ T.factors[2].T ; sigmoid ) and
 * symmetric/R_TuckER.py:44-45.  `O` is the (shard of the) entity matrix,
 * (n_local x c) row-major; `out` has leading dimension ld_out >= n_local.
 * rtk_score_f32        : exact fp32 MFMA (v_mfma_f32_32x32x2_f32), v in fp32.
 * rtk_score_packed_f32 : split-fp16 MFMA path, v given as packed query planes.
 */
int rtk_score_f32(const float *v, int64_t batch, int c,
                  const float *O, int64_t n_local,
                  float *out, int64_t ld_out, unsigned flags, void *stream);

int rtk_score_packed_f32(const void *q_packed, int64_t batch, int c,
                         const float *O, int64_t n_local,
                         float *out, int64_t ld_out, unsigned flags, void *stream);

/*
 * Both stages: the whole closure body, asymmetric/R_TuckER.py:43-48.
 * out (B x ld_out) receives sigmoid scores (or logits without RTK_SCORE_SIGMOID)
 * against the n_local rows of O.  For the symmetric model pass S == O == E.
 */
int rtk_score_1vN_f32(const float *core, int a, int b, int c,
                      const float *R, int64_t n_rel,
                      const float *S, int64_t n_sub,
                      const float *O, int64_t n_local,
                      const int64_t *rel_idx, const int64_t *sub_idx, int64_t batch,
                      float *out, int64_t ld_out, unsigned flags,
                      void *workspace, size_t workspace_bytes, void *stream);

/*
 * bf16 operand path (BASELINE.json configs[2], [4]): core, R, S, O are bf16 (raw 16-bit
 * storage, torch.bfloat16), all accumulation is fp32, the query vectors are rounded to bf16
 * before the score product (v_mfma_f32_32x32x16_bf16), scores are fp32.  Same argument
 * meaning, workspace rules (query sizes with dtype = RTK_BF16) and reference lines as the
 * _f32 entry points above.
 */
int rtk_query_vectors_bf16(const void *core, int a, int b, int c,
                           const void *R, int64_t n_rel,
                           const void *S, int64_t n_sub,
                           const int64_t *rel_idx, const int64_t *sub_idx, int64_t batch,
                           float *v_out, void *q_packed,
                           void *workspace, size_t workspace_bytes, void *stream);

int rtk_score_packed_bf16(const void *q_packed, int64_t batch, int c,
                          const void *O, int64_t n_local,
                          float *out, int64_t ld_out, unsigned flags, void *stream);

int rtk_score_1vN_bf16(const void *core, int a, int b, int c,
                       const void *R, int64_t n_rel,
                       const void *S, int64_t n_sub,
                       const void *O, int64_t n_local,
                       const int64_t *rel_idx, const int64_t *sub_idx, int64_t batch,
                       float *out, int64_t ld_out, unsigned flags,
                       void *workspace, size_t workspace_bytes, void *stream);

/*
 * General fp32 MFMA GEMM used by the backward pass and as the exact fallback:
 *   C[m,n] (+)= sum_k A(m,k) * B(n,k)
 * A(m,k) = A[m*lda + k] if a_kmajor else A[k*lda + m]; likewise B(n,k).
 * flags: RTK_SCORE_SIGMOID applies sigmoid to C.
 */
int rtk_gemm_f32(const float *A, int a_kmajor, int64_t lda,
                 const float *B, int b_kmajor, int64_t ldb,
                 float *C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                 unsigned flags, void *stream);

/* Split-K variant for short-and-wide products (K >> M, N), e.g. the backward product
 * dv = dZ . O (K = number of entities; autograd of asymmetric/R_TuckER.py:47): the K chunks are
 * computed by separate workgroups into slabs of the caller's workspace
 * (rtk_gemm_f32_splitk_workspace_bytes) and added in chunk order by a second kernel -- a fixed
 * summation order, bit-identical from run to run (no float atomics).  C contiguous (ldc == N). */
size_t rtk_gemm_f32_splitk_workspace_bytes(int64_t M, int64_t N, int splits);
int rtk_gemm_f32_splitk(const float *A, int a_kmajor, int64_t lda,
                        const float *B, int b_kmajor, int64_t ldb,
                        float *C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                        int splits, void *workspace, size_t workspace_bytes, void *stream);

/* The same product on the split-fp16 path of the forward (three f16 MFMAs per k-step on hi/lo halves,
 * fp32 accumulation) for the two B x N sized backward products of asymmetric/R_TuckER.py:47,
 * dO = dZ^T v and dv = dZ O.  amax_a / amax_b: DEVICE pointers to one float each, an UPPER BOUND of
 * max|A| and max|B| (for the BCE gradient |dZ| <= |g| / (B N) needs no pass over dZ); each operand is
 * scaled by the power of two that puts its bound just below 2^15, so an element larger than the bound
 * overflows to inf (loud), and an element keeps max(2^-22 |x|, 2^-40 bound) of absolute accuracy: the
 * result is accurate normwise, relative to max|A| max|B| K -- right for a gradient, not for an
 * orthogonalisation (use rtk_gemm_f32 there).  splits == 1: C may have any ldc >= N, no workspace;
 * splits > 1: as rtk_gemm_f32_splitk (slabs added in chunk order: deterministic). */
int rtk_gemm_sf16_splitk(const float *A, int a_kmajor, int64_t lda, const float *amax_a,
                         const float *B, int b_kmajor, int64_t ldb, const float *amax_b,
                         float *C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                         int splits, void *workspace, size_t workspace_bytes, void *stream);

/* max |x| over a rows x cols fp32 matrix with row pitch ld -> *out (one float on the device): the operand
 * bounds of rtk_gemm_sf16_splitk when the caller has no analytic one.  Order-independent (integer atomicMax
 * on the bit patterns), NaNs skipped. */
int rtk_absmax_f32(const float *x, int64_t rows, int64_t cols, int64_t ld, float *out, void *stream);

/*
 * Backward of stage 1 (autograd of asymmetric/R_TuckER.py:43-46, which the Riemannian gradient
 * differentiates through loss_fn, train.py:79-82): given dv = d loss / d v (batch x c, fp32),
 *   g_core (a,b,c) = sum_d R[r_d] (x) S[h_d] (x) dv[d]
 *   g_R (n_rel,a)  : row u = sum over the queries with r_d = u of  (G x_1 S[h_d] x_2 dv[d])
 *   g_S (n_sub,b)  : row j = sum over the queries with h_d = j of  (G x_0 R[r_d] x_2 dv[d])
 * Each output may be NULL (skipped); the others are written in full (untouched rows = 0).
 * Deterministic: fixed summation order everywhere (the row scatter adds the queries of an id in
 * increasing query order; no float atomics).  Ranks up to 1024.  Workspace:
 * rtk_query_bwd_workspace_bytes (2 * batch * a * b floats + the per-query rows).
 */
size_t rtk_query_bwd_workspace_bytes(int64_t batch, int a, int b, int c);
int rtk_query_vectors_bwd_f32(const float *core, int a, int b, int c,
                              const float *R, int64_t n_rel,
                              const float *S, int64_t n_sub,
                              const int64_t *rel_idx, const int64_t *sub_idx, int64_t batch,
                              const float *dv, float *g_core, float *g_R, float *g_S,
                              void *workspace, size_t workspace_bytes, void *stream);

/* Backward of the logistic (R_TuckER.py:48): dZ = dP * P * (1 - P), n contiguous elements. */
int rtk_sigmoid_grad_f32(const float *dP, const float *P, float *dZ, int64_t n, void *stream);

/* the same with a row pitch (in elements) per array: dZ on 128-byte aligned rows for the GEMMs */
int rtk_sigmoid_grad_rows_f32(const float *dP, int64_t ld_dp, const float *P, int64_t ld_p, float *dZ,
                              int64_t ld_dz, int64_t batch, int64_t n, void *stream);

/*
 * Filtered rank of the queried object, on the device -- replaces the full B x N sort of the
 * eval tail (train.py:115-117; src/utils/utils.py:15-22 filter_predictions + src/utils/metrics.py:5-8).
 *   P          (batch x ld) scores, as written by the score stage (not modified)
 *   obj_idx    queried object id per row (features[:, 2])
 *   pair_slot  per row, index into the CSR of known-true objects of its (subject, relation)
 *              pair, or NULL for "no filtering"; pair_ptr / pair_obj: that CSR (int64)
 *   ranks_out  int32: 1 + #{j: p'_j > p_t} + #{j < t: p'_j == p_t}, p' = scores with the other
 *              true objects set to 0  (= position in a stable descending sort)
 *   bce_rows_out  optional double[batch]: per-row sum of nn.BCELoss terms against the 0/1 targets
 *              (train.py:113), logs clamped at -100 like torch; NULL to skip
 */
int rtk_filtered_rank_f32(const float *P, int64_t batch, int64_t n_ent, int64_t ld,
                          const int64_t *obj_idx, const int64_t *pair_slot,
                          const int64_t *pair_ptr, const int64_t *pair_obj,
                          int32_t *ranks_out, double *bce_rows_out, void *stream);

/*
 * Batch sums of the metrics (src/utils/metrics.py:4-22: mrr = sum 1/rank, hits@k = #(rank <= k)) and of
 * the BCE row sums, ADDED to acc5[0..4] = (sum 1/rank, hits@1, hits@3, hits@10, bce) -- the running
 * totals train.py:118-121 keeps per evaluation, without a device -> host copy per batch.
 */
int rtk_rank_metrics_f64(const int32_t *ranks, const double *bce_rows, int64_t batch, double *acc5, void *stream);
/* The same with every BCE row sum multiplied by bce_scale first: the reference averages the per-batch MEAN losses
 * (train.py:113,125), i.e. bce_scale = 1 / (batch * n_ent), without a pass over the row sums in between. */
int rtk_rank_metrics_scaled_f64(const int32_t *ranks, const double *bce_rows, int64_t batch, double bce_scale,
                                double *acc5, void *stream);

/*
 * Kernel timer: the duration of ONE score-kernel launch as the device saw it -- begin and end of the kernel itself, the
 * figure a rocprofv3 kernel trace reports -- without a profiler.  (A pair of events recorded on the stream around a
 * launch also carries the event records' own stream time and the dispatch gap in front of the kernel: 2.5-3 us of a
 * 31 us kernel.)  rtk_timer_arm(t): the NEXT score kernel this thread launches through rtk_score_packed_* /
 * rtk_score_1vN_* (the column-group, wave-specialised and bf16 kernels; not the v3 split kernel or the exact-fp32 GEMM: there the timer reports an error) is launched
 * with the timer's two events (hipExtLaunchKernelGGL); rtk_timer_elapsed_ms waits for that kernel and returns its
 * duration.  Arming is per thread and consumed by one launch -- arm it directly in front of the stage-2 call
 * (rtk_score_packed_*): with bf16 operands and a relation rank above 32 stage 1 builds its tables on the same bf16
 * kernel and would take the timer.  bench.py's roofline.kernel_ms.
 */
int rtk_timer_create(void **timer);
int rtk_timer_arm(void *timer);
int rtk_timer_elapsed_ms(void *timer, float *ms);
int rtk_timer_destroy(void *timer);

/*
 * Training forward with the loss fused into the score kernel's epilogue (reference train.py:79,136:
 * nn.BCELoss(mean) of sigmoid scores against label-smoothed targets y = (1 - eps) multi_hot + eps / N,
 * src/data/Dataset.py:51-52).  fp32 operands, c <= 512, packed query planes from rtk_query_vectors_f32.
 *   rtk_score_packed_bce_f32: x_out[d, j] = p - eps / N  (0 where the fp32 score saturated to 1.0f / 0.0f: the
 *       reference's autograd gives a zero logit gradient there) = d BCE / d logit of a NEGATIVE entry, up to the
 *       factor g / (B N); partials_out[0 .. rtk_score_bce_partials()) receive per-workgroup sums of the entries' BCE
 *       terms taken as negatives (unused slots are zeroed).  The B x N matrix is written once, never re-read.
 *   rtk_bce_patch_pos_f32: the known objects of every pair (CSR, as rtk_bce_rows_f32): x <- x - (1 - eps), and
 *       rows_pos_out[4 d .. 4 d + 3] = four partial sums of the correction of row d's BCE sum (4 * batch doubles); the positives' logits are recomputed from the fp32
 *       query vectors v (B, c) and entity rows O (N, c) (x = p - eps / N cannot give back a p far below eps / N).
 *   loss = (sum(partials) + sum(rows_pos)) / (B N);   d loss / d logits = x * g / (B N).
 */
int rtk_score_bce_partials(void);
int rtk_score_packed_bce_f32(const void *q_packed, int64_t batch, int c, const float *O, int64_t n_local,
                             float *x_out, int64_t ld_out, float label_smoothing, double *partials_out, void *stream);
int rtk_bce_patch_pos_f32(float *X, int64_t batch, int64_t n_ent, int64_t ld, const int64_t *pair_slot,
                          const int64_t *pair_ptr, const int64_t *pair_obj, float label_smoothing,
                          const float *v, const float *O, int c, double *rows_pos_out, void *stream);

/*
 * The exchange step of the entity-sharded path (BASELINE.json north_star: "RCCL all-gather of
 * per-shard scores over xGMI"): one process per GPU, rank p holds entity rows [p * n_loc, (p + 1) * n_loc) and
 * writes its (B, pitch) score block into slot p of a (world, B, pitch) buffer; rtk_allgather_scores completes the
 * buffer IN PLACE on every rank (ncclAllGather's in-place form, enqueued on `stream`).  RCCL is bound at run time
 * (the copy already in the process, else librccl.so of the ROCm install); without it these return
 * RTK_ERR_UNSUPPORTED.  rtk_comm_unique_id: 128 bytes for rank 0 to hand to its peers by the host's own means.
 * rtk_comm_init uses the calling thread's current device; one communicator per process and GPU.
 * No counterpart in the reference (single device): the Python mirror is r_tucker_amd.sharded.
 */
int rtk_comm_unique_id(void *id_out_128_bytes);
int rtk_comm_init(int rank, int world, const void *unique_id_128_bytes, void **comm_out);
int rtk_allgather_scores(void *comm, void *buf, size_t bytes_per_rank, void *stream);
int rtk_comm_destroy(void *comm);

/*
 * Batched small Cholesky-QR factor step, float64: for each of `batch` symmetric positive semidefinite k x k Gram
 * matrices S = W^T W (row-major, contiguous, k <= 256), with D = sqrt(diag S) if `equilibrate` (else I) and
 *     A = D^-1 S D^-1 + (shift_diag + shift_trace * trace(S)) I = L L^T,
 * the upper triangular  R = L^T D  and  X = D^-1 L^-T :  W X has orthonormal columns, W = (W X) R, and
 * X X^T = (S + shift)^-1 when not equilibrated.  One workgroup per matrix, one launch, no workspace, no host
 * synchronisation, no failure status (a pivot that cancelled below 1e-14 of its diagonal entry is floored there;
 * trace(S) <= 0 gives zero outputs).  The Riemannian optimizer step orthonormalises and inverts the core's Gram
 * matrices with it (replaces tucker_riemopt's QR / SVD / solve calls reached from
 * src/model/asymmetric/optim.py:86-89,107-108 and src/model/symmetric/optim.py:80-83,101-103).
 * The three buffers must be distinct.
 */
int rtk_gram_factor_f64(const void *S, int64_t batch, int k, int equilibrate, double shift_diag, double shift_trace,
                        void *R_out, void *X_out, void *stream);

/*
 * The same ranking with the entity dimension sharded over GPUs (no gather of the scores): a rank
 * holds columns [col0, col0 + n_local) of the score matrix; the count is a sum over columns.
 *   1. rtk_target_scores_f32: pt_out[d] = P[d, obj_idx[d] - col0] where this rank owns the queried
 *      object, -inf elsewhere;  all-reduce MAX over the ranks gives every rank the target scores;
 *   2. rtk_filtered_rank_partial_f32: counts_out[d] = this block's share of
 *      #{j: p'_j > p_t} + #{j < t: p'_j == p_t}  (obj_idx and pair_obj hold GLOBAL entity ids), and
 *      optionally its share of the row's BCE sum;  all-reduce SUM, then rank = 1 + count.
 * B x 12 bytes cross the links instead of B x N x 4 (train.py:113-117 on a sharded entity matrix).
 */
int rtk_target_scores_f32(const float *P, int64_t batch, int64_t n_local, int64_t ld, int64_t col0,
                          const int64_t *obj_idx, float *pt_out, void *stream);

int rtk_filtered_rank_partial_f32(const float *P, int64_t batch, int64_t n_local, int64_t ld, int64_t col0,
                                  const float *target_scores, const int64_t *obj_idx,
                                  const int64_t *pair_slot, const int64_t *pair_ptr, const int64_t *pair_obj,
                                  int32_t *counts_out, double *bce_rows_out, void *stream);

/*
 * Training loss without dense targets (train.py:76-82 with criterion = nn.BCELoss, train.py:136;
 * targets as src/data/Dataset.py:43-53 builds them: y = (1 - eps) * multi_hot + eps / N).
 * The multi-hot part is the CSR of known objects per (subject, relation) pair (pair_slot per row;
 * every object of a pair listed once).
 *   rtk_bce_rows_f32  rows_out[d] = sum_j BCE(P[d,j], y[d,j])  (double; logs clamped at -100 like
 *                     torch); the reference's mean loss is sum(rows_out) / (batch * n_ent)
 *   rtk_bce_grad_f32  in place  P[d,j] <- (P[d,j] - y[d,j]) * grad_loss[0] * scale : with
 *                     scale = 1 / (batch * n_ent) this is d loss / d logits (P = sigmoid(logits)),
 *                     the left operand of the dO / dv GEMMs.  grad_loss is a DEVICE scalar.
 *                     Entries whose fp32 score is saturated (exactly 1.0f or 0.0f) become 0, as in the
 *                     reference's autograd (BCELoss backward x logistic backward = (p - y) * [p(1-p) / max(p(1-p), 1e-12)]).
 */
int rtk_bce_rows_f32(const float *P, int64_t batch, int64_t n_ent, int64_t ld, const int64_t *pair_slot,
                     const int64_t *pair_ptr, const int64_t *pair_obj, float label_smoothing,
                     double *rows_out, void *stream);

int rtk_bce_grad_f32(float *P, int64_t batch, int64_t n_ent, int64_t ld, const int64_t *pair_slot,
                     const int64_t *pair_ptr, const int64_t *pair_obj, float label_smoothing,
                     const float *grad_loss, float scale, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* RTUCKER_HIP_H */
