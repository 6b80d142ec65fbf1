// "Packed query planes": the layout stage 1 writes and the split-fp16 / bf16 score
// kernels stream through LDS.  One block per tile of 32 queries:
//
//   [ 32 x float : per-row unscale factor 2^-sh_d            ]   RTK_PACK_HDR = 128 B
//   [ plane 0 (hi) : ksteps x 2 (k-half) x 32 (row) x 8 x 16-bit ]   1024*ksteps B
//   [ plane 1 (lo) : same                                       ]   (fp32 path only)
//
// i.e. exactly the A-operand fragment order of v_mfma_f32_32x32x16_{f16,bf16}
// (lane l holds A[row = l & 31][k = 16*ks + 8*(l >> 5) + j], j = 0..7 -- 16 contiguous
// bytes), so a wave's fragment read is one linear, conflict-free 1-KiB ds_read_b128.
// Element v[d,k] of the fp32 path is stored as  hi = fp16(x), lo = fp16(x - hi) with
// x = v[d,k] * 2^sh_d and sh_d chosen so the row maximum lands in [2^14, 2^15): both
// halves stay inside fp16's normal range, hi + lo carries ~22 significand bits.
#pragma once
#include <stdint.h>
#include <math.h>

#define RTK_PACK_HDR 128

#if defined(__HIPCC__)
#define RTK_HD __host__ __device__ __forceinline__
#else
#define RTK_HD static inline
#endif

RTK_HD int64_t rtk_pack_tile_bytes(int ksteps, int planes) { return RTK_PACK_HDR + (int64_t)planes * ksteps * 1024; }

// index (in 16-bit elements, within a plane) of element k of row `row`
RTK_HD int rtk_pack_offset(int ksteps, int k, int row) {
    (void)ksteps;
    return (((k >> 4) * 2 + ((k >> 3) & 1)) * 32 + row) * 8 + (k & 7);
}

// power-of-two shift that brings a row maximum `mx` (>= 0) into [2^14, 2^15)
RTK_HD int rtk_pack_shift(float mx) {
    if (!(mx > 0.f) || !(mx < INFINITY)) return 0;
    int e;
    (void)frexpf(mx, &e);        // mx = m * 2^e, m in [0.5, 1)  ->  floor(log2 mx) = e - 1
    int sh = 14 - (e - 1);
    return sh > 100 ? 100 : (sh < -100 ? -100 : sh);
}
