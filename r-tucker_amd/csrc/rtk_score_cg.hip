// Launcher of the column-group split-fp16 score kernel (rtk_score_cg_kernel.h).
// Compiled three times (build.sh), once per logistic variant RTK_CG_SG = 0 (logits), 1 (exact), 2 (fast):
// 13 k-step counts x 4 M-wave roles x 2 sweeps each; the RTK_CG_SG = 2 object also holds the dispatcher.
#include <stdlib.h>

#include "rtk_score_cg_kernel.h"

namespace {

template <int KS, int SG>
bool launch_v(const unsigned char *qp, int B, const float *O, int N, int c, float *out, int64_t ld, int W, int U,
              hipStream_t st) {
    const size_t smem = rtk_cg::lds_bytes<KS>(c);
    static std::atomic<unsigned long long> lds_ok{0};
    if (rtk_ensure_dynamic_lds(reinterpret_cast<const void *>(&rtk_cg::score_cg_kernel<KS, SG>), 160 * 1024, lds_ok,
                               "score_cg_kernel") != RTK_OK)
        return false;
    static const int nt_env = getenv("RTK_WS_NT") ? atoi(getenv("RTK_WS_NT")) : 1;   // A/B: nontemporal score stores
    const int nts = nt_env && (ld * 4) % 128 == 0 && (reinterpret_cast<uintptr_t>(out) & 127) == 0;
    static const int tune = getenv("RTK_CG_TUNE") ? atoi(getenv("RTK_CG_TUNE")) : 0;   // A/B: wave priorities
    RTK_LAUNCH_SCORE((rtk_cg::score_cg_kernel<KS, SG>), dim3(W), dim3(512), smem, st, qp, B, O, N, c, out, ld, U, nts, tune);
    return true;
}

}  // namespace

#ifndef RTK_CG_SG
#error "compile with -DRTK_CG_SG=0|1|2"
#endif
#define RTK_CG_CAT2(a, b) a##b
#define RTK_CG_CAT(a, b) RTK_CG_CAT2(a, b)
int rtk_score_cg_launch_sg0(int ks, const unsigned char *qp, int B, const float *O, int N, int c, float *out, int64_t ld, int W, int U, hipStream_t st);
int rtk_score_cg_launch_sg1(int ks, const unsigned char *qp, int B, const float *O, int N, int c, float *out, int64_t ld, int W, int U, hipStream_t st);
int rtk_score_cg_launch_sg2(int ks, const unsigned char *qp, int B, const float *O, int N, int c, float *out, int64_t ld, int W, int U, hipStream_t st);

int RTK_CG_CAT(rtk_score_cg_launch_sg, RTK_CG_SG)(int ks, const unsigned char *qp, int B, const float *O, int N, int c,
                                                  float *out, int64_t ld, int W, int U, hipStream_t st) {
#define RTK_KS(K_) case K_: return launch_v<K_, RTK_CG_SG>(qp, B, O, N, c, out, ld, W, U, st) ? 1 : RTK_ERR_LAUNCH;
    switch (ks) {
        RTK_KS(1) RTK_KS(2) RTK_KS(3) RTK_KS(4) RTK_KS(5) RTK_KS(6) RTK_KS(7) RTK_KS(8) RTK_KS(9) RTK_KS(10)
        RTK_KS(11) RTK_KS(12) RTK_KS(13)
        default: return 0;
    }
#undef RTK_KS
}

#if RTK_CG_SG == 2

#ifdef RTK_CG_STAMPS
// tools/ablate: copy out (dst != NULL) or clear (clear != 0) the timeline of the fast-logistic instantiation
extern "C" int rtk_cg_timeline(unsigned long long *dst, int n, int clear) {
    static unsigned long long zeros[256 * 2 * 64];
    if (clear && hipMemcpyToSymbol(HIP_SYMBOL(rtk_cg::g_cg_tl), zeros, sizeof(zeros)) != hipSuccess) return -1;
    if (dst && hipMemcpyFromSymbol(dst, HIP_SYMBOL(rtk_cg::g_cg_tl), (size_t)n * 8) != hipSuccess) return -2;
    return 0;
}
#endif

// The set schedule: G = ceil(N/32) column groups in U = W * P sets of <= 5 consecutive groups, P sets per
// workgroup (one after the other).  `force`: run this kernel whatever the shape (tests, A/B).
// Chosen by default only where it is the better schedule: one set per workgroup (P = 1) and at least ~4.4
// groups in it on a full grid -- the four M waves all busy, the fifth group filling most sets -- i.e.
// 36 000 < N <= 40 960 on 256 CUs (WN18RR: 40 943 entities = 1280 groups = 5 per CU exactly).  Shallower or
// deeper shapes keep the ws kernel's tile schedule (several passes here would each pay an exposed load of
// the next set).
// 1 = launched, 0 = not this kernel's shape (the caller goes on to the next kernel), < 0 = rtk_status
int rtk_score_cg_launch(const unsigned char *qp, int B, const float *O, int N, int c, float *out, int64_t ld,
                        int sg, bool o_vec, bool force, hipStream_t st) {
    const int ks = (c + 15) / 16;
    if (!o_vec || ks > 13) return 0;   // c % 4 != 0 or unaligned O: the two-workgroup kernel has the scalar paths
    const int64_t G = rtk_cdiv(N, 32);
    int64_t sets_min = rtk_cdiv(G, rtk_cg::NG);
    // (A/B, RTK_CG_SPREAD=1: a shape with fewer than 256 full sets is spread over min(G, 256) workgroups of 1-4 groups
    // instead of ceil(G / 5) workgroups of five)
    static const int spread = getenv("RTK_CG_SPREAD") ? atoi(getenv("RTK_CG_SPREAD")) : 0;
    if (spread && sets_min < 256) sets_min = G < 256 ? G : 256;
    const int W = (int)(sets_min < 256 ? sets_min : 256);
    const int64_t P = rtk_cdiv(sets_min, W);
    if (P * W > (1 << 30)) return 0;
    if (!force && !(P == 1 && W == 256 && 10 * G >= 44 * W)) return 0;
    const int U = (int)(P * W);
    if (sg == 0) return rtk_score_cg_launch_sg0(ks, qp, B, O, N, c, out, ld, W, U, st);
    if (sg == 1) return rtk_score_cg_launch_sg1(ks, qp, B, O, N, c, out, ld, W, U, st);
    return rtk_score_cg_launch_sg2(ks, qp, B, O, N, c, out, ld, W, U, st);
}
#endif
