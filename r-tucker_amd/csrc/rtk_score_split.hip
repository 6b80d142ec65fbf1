// Stage 2 of the scoring path on the bf16/fp16-rate matrix cores:
//     out[d, j] = sigmoid( v[d,:] . O[j,:] )         (reference: asymmetric/R_TuckER.py:47-48)
//
// fp32 operands are evaluated as a split-precision product on v_mfma_f32_32x32x16_f16:
// x = hi + lo (two fp16 halves of a power-of-two-scaled row, ~22 significand bits) and
//     v.o  ~=  vh.oh + vh.ol + vl.oh           (fp32 accumulation inside the MFMA)
// -- 3 MFMAs at 16x the fp32-MFMA rate, i.e. 5.3x the throughput of an exact fp32 MFMA
// GEMM at about fp32 accuracy (the dropped vl.ol term is 2^-22 relative).  That moves
// this kernel from matrix-core-bound (fp32) to the HBM write of the B x N scores.
//
// Work decomposition ("entity-stationary"): a workgroup of 4 waves owns 128 entity
// rows of O; each wave converts ITS 32 rows once into register-resident B fragments
// (row max -> power-of-two scale -> hi/lo split; O is read from HBM exactly once and
// never written back), then the workgroup sweeps the query tiles: each 32-query tile
// of packed planes (rtk_pack.h) is copied linearly into LDS, every wave reads the same
// A fragments (conflict-free ds_read_b128) and issues 3*KS MFMAs into one 32x32
// accumulator, and the epilogue unscales, applies the logistic and stores
// 128-B-contiguous row segments (lane = entity) straight from the accumulator.
#include <stdlib.h>

#include "rtk_common.h"
#include "rtk_pack.h"

#include "rtk_score_split_kernel.h"
using rtk_split::score_split_kernel;
namespace {

template <int KS, int SG, int MINW>
int launch_one(const unsigned char *qp, int B, const float *O, int N, int c, float *out, int64_t ld,
                bool o_vec, unsigned grid, hipStream_t st) {
    constexpr size_t smem = 2 * (size_t)(RTK_PACK_HDR + 2 * KS * 1024);
    static std::atomic<unsigned long long> lds_ok{0};
    if (smem > 64 * 1024) {
        const int rc = rtk_ensure_dynamic_lds(reinterpret_cast<const void *>(&score_split_kernel<KS, SG, MINW, 0>), (int)smem,
                                              lds_ok, "score_split_kernel");
        if (rc != RTK_OK) return rc;
    }
    hipLaunchKernelGGL((score_split_kernel<KS, SG, MINW, 0>), dim3(grid), dim3(256), smem, st, qp, B, O, N, c,
                       out, ld, o_vec);
    return RTK_OK;
}

template <int KS, int MINW>
int launch_ks(const unsigned char *qp, int B, const float *O, int N, int c, float *out, int64_t ld,
               int sigmoid, bool o_vec, hipStream_t st) {
    // one block per resident slot (256 CUs x MINW workgroups); the kernel splits the
    // linearised (entity tile, query tile) space evenly over them
    const int64_t units = rtk_cdiv(N, 128) * rtk_cdiv(B, 32);
    const unsigned grid = (unsigned)(units < 256 * MINW ? units : 256 * MINW);
    if (sigmoid == 0) return launch_one<KS, 0, MINW>(qp, B, O, N, c, out, ld, o_vec, grid, st);
    if (sigmoid == 1) return launch_one<KS, 1, MINW>(qp, B, O, N, c, out, ld, o_vec, grid, st);
    return launch_one<KS, 2, MINW>(qp, B, O, N, c, out, ld, o_vec, grid, st);
}

// training forward: x = p - t0 and the negatives' BCE terms (score_split_kernel<..., LOSS = true>)
template <int KS, int MINW>
int launch_loss(const unsigned char *qp, int B, const float *O, int N, int c, float *out, int64_t ld, bool o_vec, float t0,
                double *partials, hipStream_t st) {
    constexpr size_t smem = 2 * (size_t)(RTK_PACK_HDR + 2 * KS * 1024);
    static std::atomic<unsigned long long> lds_ok{0};
    if (smem > 64 * 1024) {
        const int rc = rtk_ensure_dynamic_lds(reinterpret_cast<const void *>(&score_split_kernel<KS, 2, MINW, 0, true>), (int)smem,
                                              lds_ok, "score_split_kernel (loss)");
        if (rc != RTK_OK) return rc;
    }
    const int64_t units = rtk_cdiv(N, 128) * rtk_cdiv(B, 32);
    const unsigned grid = (unsigned)(units < 256 * MINW ? units : 256 * MINW);
    hipLaunchKernelGGL((score_split_kernel<KS, 2, MINW, 0, true>), dim3(grid), dim3(256), smem, st, qp, B, O, N, c, out, ld,
                       o_vec, t0, partials);
    return RTK_OK;
}

}  // namespace

int rtk_score_ws_launch(const unsigned char *qp, int B, const float *O, int N, int c, float *out, int64_t ld,
                        int sg, bool o_vec, hipStream_t st);

int rtk_score_cg_launch(const unsigned char *qp, int B, const float *O, int N, int c, float *out, int64_t ld,
                        int sg, bool o_vec, bool force, hipStream_t st);

// Default: the column-group kernel (cg) where its schedule fills the chip (rtk_score_cg.hip), else the
// persistent wave-specialised kernel (ws), then the two-workgroups-per-CU kernel (v3) for the shapes
// neither covers.  RTK_SCORE_KERNEL=v3 forces v3, =ws skips cg, =cg runs cg on every shape it can
// (c <= 208) -- A/B comparisons and tests.  (The two-tiles-per-barrier variant "ws2" was measured
// slower, 53.9 vs 48.5 us at the WN18RR shape, and lives in tools/ablate/ only.)
// 0 = v3 only, 1 = ws then v3, 2 = cg (by shape) then ws then v3, 3 = cg (forced) then ws then v3
static int kernel_choice() {
    static int v = -1;
    if (v < 0) {
        const char *e = getenv("RTK_SCORE_KERNEL");
        if (e && e[0] == 'v' && e[1] == '3') v = 0;
        else if (e && e[0] == 'w' && e[1] == 's') v = 1;
        else if (e && e[0] == 'c' && e[1] == 'g') v = 3;
        else v = 2;
    }
    return v;
}

int rtk_split_ksteps_supported(int c) {
    const int ks = (c + 15) / 16;
    return ks >= 1 && ks <= 32;   // two fp16 planes of B fragments must fit the register file next to the pipeline state
}

extern "C" int rtk_score_packed_f32(const void *q_packed, int64_t batch, int c, const float *O,
                                    int64_t n_local, float *out, int64_t ld_out, unsigned flags,
                                    void *stream) {
    RTK_REQUIRE(q_packed && O && out, RTK_ERR_BAD_ARG, "rtk_score_packed_f32: null operand");
    RTK_REQUIRE(batch > 0 && n_local > 0 && c > 0, RTK_ERR_BAD_ARG, "rtk_score_packed_f32: sizes must be positive");
    RTK_REQUIRE(ld_out >= n_local, RTK_ERR_BAD_ARG, "rtk_score_packed_f32: ld_out < n_local");
    RTK_REQUIRE(ld_out < (1ll << 24), RTK_ERR_UNSUPPORTED, "rtk_score_packed_f32: ld_out >= 2^24 (32 rows must fit a 2 GiB buffer window)");
    RTK_REQUIRE(batch < (1ll << 31) && n_local < (1ll << 31) - 256, RTK_ERR_UNSUPPORTED, "rtk_score_packed_f32: dimension too large");
    RTK_REQUIRE(rtk_split_ksteps_supported(c), RTK_ERR_UNSUPPORTED, "rtk_score_packed_f32: c=%d > 512 not supported by the split-fp16 kernel (use rtk_score_f32)", c);
    hipStream_t st = (hipStream_t)stream;
    const int ks = (c + 15) / 16;
    const int sg = !(flags & RTK_SCORE_SIGMOID) ? 0 : ((flags & RTK_SCORE_SIGMOID_FAST) ? 2 : 1);
    const bool o_vec = (c % 4 == 0) && ((reinterpret_cast<uintptr_t>(O) & 15) == 0);
    const int B = (int)batch, N = (int)n_local;
    const unsigned char *qp = (const unsigned char *)q_packed;
    const unsigned hint = flags & RTK_SCORE_KERNEL_MASK;
    const int choice = hint == RTK_SCORE_KERNEL_CG ? 3 : hint == RTK_SCORE_KERNEL_WS ? 1 : hint == RTK_SCORE_KERNEL_V3 ? 0
                                                                                                        : kernel_choice();
    if (choice >= 2) {
        const int took = rtk_score_cg_launch(qp, B, O, N, c, out, ld_out, sg, o_vec, choice == 3, st);
        if (took < 0) return took;
        if (took) return rtk_check_launch("rtk_score_packed_f32");
    }
    if (choice >= 1) {
        const int took = rtk_score_ws_launch(qp, B, O, N, c, out, ld_out, sg, o_vec, st);
        if (took < 0) return took;
        if (took) return rtk_check_launch("rtk_score_packed_f32");
    }
    int rc = RTK_OK;
    // the packed planes were written for exactly `ks` k-steps (tile stride), so the
    // instantiation must match exactly.
#define RTK_KS(K_, W_) \
    case K_: rc = launch_ks<K_, W_>(qp, B, O, N, c, out, ld_out, sg, o_vec, st); break;
    switch (ks) {
        RTK_KS(1, 2) RTK_KS(2, 2) RTK_KS(3, 2) RTK_KS(4, 2) RTK_KS(5, 2) RTK_KS(6, 2) RTK_KS(7, 2) RTK_KS(8, 2)
        RTK_KS(9, 2) RTK_KS(10, 2) RTK_KS(11, 2) RTK_KS(12, 2) RTK_KS(13, 2) RTK_KS(14, 2) RTK_KS(15, 2) RTK_KS(16, 2)
        // 256 < c <= 512 (the doubled-rank tensors the Riemannian gradient scores): one
        // 4-wave workgroup per CU, the hi/lo B fragments take up to 256 of a wave's 512 registers
        // (the compiler places them in the accumulation half of the unified file)
        RTK_KS(17, 1) RTK_KS(18, 1) RTK_KS(19, 1) RTK_KS(20, 1) RTK_KS(21, 1) RTK_KS(22, 1) RTK_KS(23, 1) RTK_KS(24, 1)
        RTK_KS(25, 1) RTK_KS(26, 1) RTK_KS(27, 1) RTK_KS(28, 1) RTK_KS(29, 1) RTK_KS(30, 1) RTK_KS(31, 1) RTK_KS(32, 1)
        default:
            rtk_set_error("rtk_score_packed_f32: unsupported k-step count %d", ks);
            return RTK_ERR_UNSUPPORTED;
    }
#undef RTK_KS
    if (rc != RTK_OK) return rc;
    return rtk_check_launch("rtk_score_packed_f32");
}

extern "C" int rtk_score_bce_partials(void) { return 512; }      // >= the largest grid of the loss kernel

extern "C" int rtk_score_packed_bce_f32(const void *q_packed, int64_t batch, int c, const float *O, int64_t n_local,
                                        float *x_out, int64_t ld_out, float label_smoothing, double *partials_out,
                                        void *stream) {
    RTK_REQUIRE(q_packed && O && x_out && partials_out, RTK_ERR_BAD_ARG, "rtk_score_packed_bce_f32: null operand");
    RTK_REQUIRE(batch > 0 && n_local > 0 && c > 0, RTK_ERR_BAD_ARG, "rtk_score_packed_bce_f32: sizes must be positive");
    RTK_REQUIRE(ld_out >= n_local, RTK_ERR_BAD_ARG, "rtk_score_packed_bce_f32: ld_out < n_local");
    RTK_REQUIRE(ld_out < (1ll << 24), RTK_ERR_UNSUPPORTED, "rtk_score_packed_bce_f32: ld_out >= 2^24");
    RTK_REQUIRE(batch < (1ll << 31) && n_local < (1ll << 31) - 256, RTK_ERR_UNSUPPORTED, "rtk_score_packed_bce_f32: dimension too large");
    RTK_REQUIRE(rtk_split_ksteps_supported(c), RTK_ERR_UNSUPPORTED, "rtk_score_packed_bce_f32: c=%d > 512 not supported", c);
    RTK_REQUIRE(label_smoothing >= 0.f && label_smoothing < 1.f, RTK_ERR_BAD_ARG, "rtk_score_packed_bce_f32: label smoothing %g outside [0, 1)", (double)label_smoothing);
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(partials_out, 0, 512 * sizeof(double), st) != hipSuccess) {
        rtk_set_error("rtk_score_packed_bce_f32: clearing the partial sums failed");
        return RTK_ERR_LAUNCH;
    }
    const int ks = (c + 15) / 16;
    const bool o_vec = (c % 4 == 0) && ((reinterpret_cast<uintptr_t>(O) & 15) == 0);
    const int B = (int)batch, N = (int)n_local;
    const float t0 = label_smoothing / (float)n_local;
    const unsigned char *qp = (const unsigned char *)q_packed;
    int rc = RTK_OK;
#define RTK_KS(K_, W_) \
    case K_: rc = launch_loss<K_, W_>(qp, B, O, N, c, x_out, ld_out, o_vec, t0, partials_out, st); break;
    switch (ks) {
        RTK_KS(1, 2) RTK_KS(2, 2) RTK_KS(3, 2) RTK_KS(4, 2) RTK_KS(5, 2) RTK_KS(6, 2) RTK_KS(7, 2) RTK_KS(8, 2)
        RTK_KS(9, 2) RTK_KS(10, 2) RTK_KS(11, 2) RTK_KS(12, 2) RTK_KS(13, 2) RTK_KS(14, 2) RTK_KS(15, 2) RTK_KS(16, 2)
        RTK_KS(17, 1) RTK_KS(18, 1) RTK_KS(19, 1) RTK_KS(20, 1) RTK_KS(21, 1) RTK_KS(22, 1) RTK_KS(23, 1) RTK_KS(24, 1)
        RTK_KS(25, 1) RTK_KS(26, 1) RTK_KS(27, 1) RTK_KS(28, 1) RTK_KS(29, 1) RTK_KS(30, 1) RTK_KS(31, 1) RTK_KS(32, 1)
        default:
            rtk_set_error("rtk_score_packed_bce_f32: unsupported k-step count %d", ks);
            return RTK_ERR_UNSUPPORTED;
    }
#undef RTK_KS
    if (rc != RTK_OK) return rc;
    return rtk_check_launch("rtk_score_packed_bce_f32");
}
