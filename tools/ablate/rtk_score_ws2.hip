// Launcher of the two-tiles-per-barrier persistent split-fp16 score kernel (rtk_score_ws2_kernel.h).
#include "rtk_score_ws2_kernel.h"

namespace {

template <int KS, int SG>
void launch_one(const unsigned char *qp, int B, const float *O, int N, int c, float *out, int64_t ld, hipStream_t st) {
    const size_t smem = rtk_ws2::lds_bytes<KS>(c);
    static bool attr_set = false;   // > 64 KiB of dynamic LDS needs the attribute (idempotent)
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&rtk_ws2::score_ws2_kernel<KS, SG>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    // one resident workgroup per CU; the kernel cuts the (entity tile x query-tile pair) space evenly
    const int64_t units = rtk_cdiv(N, 128) * rtk_cdiv(rtk_cdiv(B, 32), 2);
    const unsigned grid = (unsigned)(units < 256 ? units : 256);
    hipLaunchKernelGGL((rtk_ws2::score_ws2_kernel<KS, SG>), dim3(grid), dim3(512), smem, st, qp, B, O, N, c, out, ld);
}

template <int KS>
void launch_ks(const unsigned char *qp, int B, const float *O, int N, int c, float *out, int64_t ld, int sg,
               hipStream_t st) {
    if (sg == 0) launch_one<KS, 0>(qp, B, O, N, c, out, ld, st);
    else if (sg == 1) launch_one<KS, 1>(qp, B, O, N, c, out, ld, st);
    else launch_one<KS, 2>(qp, B, O, N, c, out, ld, st);
}

}  // namespace

// returns false if this kernel does not cover the shape (caller falls back)
bool rtk_score_ws2_launch(const unsigned char *qp, int B, const float *O, int N, int c, float *out, int64_t ld,
                          int sg, bool o_vec, hipStream_t st) {
    const int ks = (c + 15) / 16;
    if (!o_vec || B <= 32) return false;   // a single query tile gains nothing from pairing
#define RTK_KS(K_) case K_: launch_ks<K_>(qp, B, O, N, c, out, ld, sg, st); return true;
    switch (ks) {
        RTK_KS(1) RTK_KS(2) RTK_KS(3) RTK_KS(4) RTK_KS(5) RTK_KS(6) RTK_KS(7) RTK_KS(8) RTK_KS(9) RTK_KS(10)
        RTK_KS(11) RTK_KS(12) RTK_KS(13)
        default: return false;
    }
#undef RTK_KS
}
