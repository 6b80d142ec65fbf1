// Ablation build of the split-fp16 score kernel (NOT part of librtucker_hip.so): the same
// kernel source instantiated with compile-time ablation masks, to see where the time goes.
// Build: tools/ablate/build.sh -> tools/ablate/librtk_ablate.so
#include <stdlib.h>
#include "rtk_score_split_kernel.h"
#include "rtk_score_ws_kernel.h"

void rtk_set_error(const char *, ...) {}

template <int SG, unsigned ABL>
static void go(const void *qp, int64_t B, int c, const float *O, int64_t N, float *out, int64_t ld, unsigned grid,
               hipStream_t st) {
    constexpr size_t smem = 2 * (size_t)(RTK_PACK_HDR + 2 * 13 * 1024);
    hipLaunchKernelGGL((rtk_split::score_split_kernel<13, SG, 2, ABL>), dim3(grid), dim3(256), smem, st,
                       (const unsigned char *)qp, (int)B, O, (int)N, c, out, ld, c % 4 == 0);
}

extern "C" int rtk_ablate_read_istamps(unsigned long long *host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(rtk_split::g_istamps), sizeof(unsigned long long) * n);
}

extern "C" int rtk_ablate_read_stamps(unsigned long long *host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(rtk_split::g_stamps), sizeof(unsigned long long) * n);
}

extern "C" int rtk_ablate_score_packed_f32(const void *qp, int64_t B, int c, const float *O, int64_t N,
                                           float *out, int64_t ld, int sigmoid, int grid, unsigned abl,
                                           void *stream) {
    if ((c + 15) / 16 != 13) return -3;
    hipStream_t st = (hipStream_t)stream;
#define CASE(SG, A) if (sigmoid == SG && abl == A) { go<SG, A>(qp, B, c, O, N, out, ld, grid, st); return (int)hipGetLastError(); }
    CASE(2, 0) CASE(1, 0) CASE(0, 0)
    CASE(2, 1) CASE(2, 2) CASE(2, 4) CASE(2, 8) CASE(2, 16) CASE(2, 20) CASE(2, 32) CASE(2, 36) CASE(2, 21) CASE(2, 53)
    CASE(2, 64) CASE(2, 128) CASE(0, 128)
    CASE(0, 21) CASE(0, 22) CASE(2, 19) CASE(0, 53) CASE(2, 5) CASE(2, 37)
    return -5;
}

template <unsigned XP>
static int ws_go(const void *qp, int64_t B, int c, const float *O, int64_t N, float *out, int64_t ld, int grid, void *stream) {
    const size_t smem = rtk_ws::lds_bytes<13>(c);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&rtk_ws::score_ws_kernel<13, 2, true, true, XP>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((rtk_ws::score_ws_kernel<13, 2, true, true, XP>), dim3(grid), dim3(512), smem, (hipStream_t)stream,
                       (const unsigned char *)qp, (int)B, O, (int)N, c, out, ld, getenv("RTK_WS_XCD") ? atoi(getenv("RTK_WS_XCD")) : 0, getenv("RTK_WS_NT") ? atoi(getenv("RTK_WS_NT")) : 0);
    return (int)hipGetLastError();
}
extern "C" int rtk_ablate_ws(const void *qp, int64_t B, int c, const float *O, int64_t N, float *out, int64_t ld,
                             int grid, int xp, void *stream) {
    if ((c + 15) / 16 != 13) return -3;
    switch (xp) {
        case 0: return ws_go<0>(qp, B, c, O, N, out, ld, grid, stream);
        case 1: return ws_go<1>(qp, B, c, O, N, out, ld, grid, stream);
        case 2: return ws_go<2>(qp, B, c, O, N, out, ld, grid, stream);
        case 3: return ws_go<3>(qp, B, c, O, N, out, ld, grid, stream);
        case 4: return ws_go<4>(qp, B, c, O, N, out, ld, grid, stream);
        case 8: return ws_go<8>(qp, B, c, O, N, out, ld, grid, stream);
        case 12: return ws_go<12>(qp, B, c, O, N, out, ld, grid, stream);
        case 16: return ws_go<16>(qp, B, c, O, N, out, ld, grid, stream);
        case 20: return ws_go<20>(qp, B, c, O, N, out, ld, grid, stream);
        case 24: return ws_go<24>(qp, B, c, O, N, out, ld, grid, stream);
        case 28: return ws_go<28>(qp, B, c, O, N, out, ld, grid, stream);
    }
    return -5;
}
extern "C" int rtk_ablate_ws_stamps(unsigned long long *host, int n, int clear) {
    if (clear) {
        static unsigned long long zeros[256 * 64];
        return (int)hipMemcpyToSymbol(HIP_SYMBOL(rtk_ws::g_ws_stamps), zeros, sizeof(zeros));
    }
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(rtk_ws::g_ws_stamps), sizeof(unsigned long long) * n);
}

extern "C" int rtk_ablate_ws_timeline(unsigned long long *host, int n, int clear) {
    if (clear) {
        void *p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(rtk_ws::g_ws_tl)) != hipSuccess) return -1;
        return (int)hipMemset(p, 0, sizeof(unsigned long long) * 256 * 2 * 64);
    }
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(rtk_ws::g_ws_tl), sizeof(unsigned long long) * n);
}

// ---- two-tiles-per-barrier kernel (rtk_score_ws2_kernel.h), STAMP build
#include "rtk_score_ws2_kernel.h"
extern "C" int rtk_ablate_ws2(const void *qp, int64_t B, int c, const float *O, int64_t N, float *out, int64_t ld,
                              int grid, void *stream) {
    if ((c + 15) / 16 != 13) return -3;
    const size_t smem = rtk_ws2::lds_bytes<13>(c);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&rtk_ws2::score_ws2_kernel<13, 2, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((rtk_ws2::score_ws2_kernel<13, 2, true>), dim3(grid), dim3(512), smem, (hipStream_t)stream,
                       (const unsigned char *)qp, (int)B, O, (int)N, c, out, ld);
    return (int)hipGetLastError();
}
extern "C" int rtk_ablate_ws2_stamps(unsigned long long *host, int n, int clear) {
    if (clear) {
        void *p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(rtk_ws2::g_ws2_stamps)) != hipSuccess) return -1;
        return (int)hipMemset(p, 0, sizeof(unsigned long long) * 256 * 8 * rtk_ws2::STAMPS_PER_WAVE);
    }
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(rtk_ws2::g_ws2_stamps), sizeof(unsigned long long) * n);
}
