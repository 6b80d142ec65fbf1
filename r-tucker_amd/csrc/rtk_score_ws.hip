// Launcher of the wave-specialised persistent split-fp16 score kernel (rtk_score_ws_kernel.h).
#include <stdlib.h>

#include "rtk_score_ws_kernel.h"

namespace {

template <int KS, int SG, bool OV>
bool launch_v(const unsigned char *qp, int B, const float *O, int N, int c, float *out, int64_t ld, hipStream_t st) {
    const size_t smem = rtk_ws::lds_bytes<KS>(c);
    static std::atomic<unsigned long long> lds_ok{0};
    if (rtk_ensure_dynamic_lds(reinterpret_cast<const void *>(&rtk_ws::score_ws_kernel<KS, SG, OV>), 160 * 1024, lds_ok,
                               "score_ws_kernel") != RTK_OK)
        return false;
    // one resident workgroup per CU; the kernel cuts the (entity tile x query tile) space evenly
    const int64_t units = rtk_cdiv(N, 128) * rtk_cdiv(B, 32);
    const unsigned grid = (unsigned)(units < 256 ? units : 256);
    static const int xcd_remap = getenv("RTK_WS_XCD") ? atoi(getenv("RTK_WS_XCD")) : 2;   // A/B: XCD-aware schedule (0 off, 1 both phases, 2 remainder tiles only)
    static const int nt_env = getenv("RTK_WS_NT") ? atoi(getenv("RTK_WS_NT")) : 1;   // A/B: nontemporal score stores
    const int nts = nt_env && (ld * 4) % 128 == 0 && (reinterpret_cast<uintptr_t>(out) & 127) == 0;
    RTK_LAUNCH_SCORE((rtk_ws::score_ws_kernel<KS, SG, OV>), dim3(grid), dim3(512), smem, st, qp, B, O, N, c, out, ld,
                       xcd_remap, nts);
    return true;
}

template <int KS, int SG>
bool launch_one(const unsigned char *qp, int B, const float *O, int N, int c, float *out, int64_t ld,
                bool o_vec, hipStream_t st) {
    (void)o_vec;
    return launch_v<KS, SG, true>(qp, B, O, N, c, out, ld, st);   // the caller guarantees o_vec
}

template <int KS>
bool launch_ks(const unsigned char *qp, int B, const float *O, int N, int c, float *out, int64_t ld, int sg,
               bool o_vec, hipStream_t st) {
    if (sg == 0) return launch_one<KS, 0>(qp, B, O, N, c, out, ld, o_vec, st);
    if (sg == 1) return launch_one<KS, 1>(qp, B, O, N, c, out, ld, o_vec, st);
    return launch_one<KS, 2>(qp, B, O, N, c, out, ld, o_vec, st);
}

}  // namespace

// 1 = launched, 0 = this kernel does not cover the shape (the caller falls back), < 0 = rtk_status
int rtk_score_ws_launch(const unsigned char *qp, int B, const float *O, int N, int c, float *out, int64_t ld,
                         int sg, bool o_vec, hipStream_t st) {
    const int ks = (c + 15) / 16;
    if (!o_vec) return 0;   // c % 4 != 0 or unaligned O: the two-workgroup kernel has the scalar paths
#define RTK_KS(K_) case K_: return launch_ks<K_>(qp, B, O, N, c, out, ld, sg, o_vec, st) ? 1 : RTK_ERR_LAUNCH;
    switch (ks) {
        RTK_KS(1) RTK_KS(2) RTK_KS(3) RTK_KS(4) RTK_KS(5) RTK_KS(6) RTK_KS(7) RTK_KS(8) RTK_KS(9) RTK_KS(10)
        RTK_KS(11) RTK_KS(12) RTK_KS(13)
        default: return 0;
    }
#undef RTK_KS
}
