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
    RTK_LAUNCH_SCORE((rtk_cg::score_cg_kernel<KS, SG>), dim3(W), dim3(512), smem, st, qp, B, O, N, c, out, ld, U,
                     (int)(rtk_cdiv(N, 32) / U), (int)(rtk_cdiv(N, 32) % U), nts, tune);
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
// workgroup (one after the other): ceil(G / 5) sets when that fills the chip, else min(G, 256) sets of 1-5 groups
// (all CUs busy; up to 1024 groups no set has a fifth group and every score has the ws kernel's bits).
// `force`: run this kernel whatever the shape (tests, A/B).  Chosen by default for one set per workgroup (P = 1)
// with at least two groups in it: 18 432 <= N <= 40 960 on 256 CUs (tools/ab_cg_shapes.py, c = 200, B = 512, back to
// back: N = 20 000 22.3 us against 24.5 for the ws kernel, 26 000 25.0 / 28.8, 32 000 25.9 / 32.7, 36 000 31-33 / 40,
// 40 943 (WN18RR: 1280 groups = 5 per CU exactly) 32-34 / 36-41; at 14 951 the two tie, at 16 384 (the ws kernel's tiles divide evenly) it is 10 % ahead, from 46 000 on -- two
// passes here, each with its own exposed prologue -- the ws kernel's tile schedule is 1-10 % ahead).
// 1 = launched, 0 = not this kernel's shape (the caller goes on to the next kernel), < 0 = rtk_status
int rtk_score_cg_launch(const unsigned char *qp, int B, const float *O, int N, int c, float *out, int64_t ld,
                        int sg, bool o_vec, bool force, hipStream_t st) {
    const int ks = (c + 15) / 16;
    if (!o_vec || ks > 13) return 0;   // c % 4 != 0 or unaligned O: the two-workgroup kernel has the scalar paths
    const int64_t G = rtk_cdiv(N, 32);
    int64_t sets = rtk_cdiv(G, rtk_cg::NG);
    if (sets < 256) sets = G < 256 ? G : 256;
    const int W = (int)(sets < 256 ? sets : 256);
    const int64_t P = rtk_cdiv(sets, W);
    if (P * W > (1 << 30)) return 0;
    if (!force && !(P == 1 && G >= 576)) return 0;
    const int U = (int)(P * W);
    if (sg == 0) return rtk_score_cg_launch_sg0(ks, qp, B, O, N, c, out, ld, W, U, st);
    if (sg == 1) return rtk_score_cg_launch_sg1(ks, qp, B, O, N, c, out, ld, W, U, st);
    return rtk_score_cg_launch_sg2(ks, qp, B, O, N, c, out, ld, W, U, st);
}
#endif
