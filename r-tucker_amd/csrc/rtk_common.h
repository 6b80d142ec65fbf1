// Shared helpers for the gfx950 kernels of librtucker_hip.so (internal header).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stddef.h>

#include "rtucker_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define RTK_WAVE 64

// host side ---------------------------------------------------------------
void rtk_set_error(const char *fmt, ...);
#define RTK_REQUIRE(cond, code, ...)            \
    do {                                        \
        if (!(cond)) {                          \
            rtk_set_error(__VA_ARGS__);         \
            return (code);                      \
        }                                       \
    } while (0)

// Kernel timer (rtk_timer_*, rtk_abi.hip): a score-kernel launcher asks whether the calling thread armed a timer; if so
// the launch goes through hipExtLaunchKernelGGL with the timer's two events, which the runtime stamps with the kernel's
// own begin / end times (what a rocprofv3 kernel trace reports), not with the stream's progress around the launch.
bool rtk_take_launch_events(hipEvent_t *start, hipEvent_t *stop);
#define RTK_LAUNCH_SCORE(kernel, grid, block, smem, st, ...)                                                      \
    do {                                                                                                          \
        hipEvent_t ev0_, ev1_;                                                                                    \
        if (rtk_take_launch_events(&ev0_, &ev1_))                                                                 \
            hipExtLaunchKernelGGL(kernel, grid, block, smem, st, ev0_, ev1_, 0, __VA_ARGS__);                     \
        else                                                                                                      \
            hipLaunchKernelGGL(kernel, grid, block, smem, st, __VA_ARGS__);                                       \
    } while (0)

static inline int rtk_check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        rtk_set_error("%s: %s", what, hipGetErrorString(e));
        return RTK_ERR_LAUNCH;
    }
    return RTK_OK;
}

// Kernels that need more than 64 KiB of dynamic LDS must raise hipFuncAttributeMaxDynamicSharedMemorySize,
// and the attribute is kept PER DEVICE: a process that scores on a second GPU has to set it there too.
// `done` is a per-instantiation bitmask of device ordinals already configured (atomic: callers may be
// multi-threaded); devices >= 64 set the attribute on every launch (idempotent).
#include <atomic>
static inline int rtk_ensure_dynamic_lds(const void *kernel, int bytes, std::atomic<unsigned long long> &done,
                                         const char *what) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess) {
        const unsigned long long bit = (dev >= 0 && dev < 64) ? (1ull << dev) : 0ull;
        if (bit && (done.load(std::memory_order_acquire) & bit)) return RTK_OK;
        e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e == hipSuccess) {
            if (bit) done.fetch_or(bit, std::memory_order_release);
            return RTK_OK;
        }
    }
    rtk_set_error("%s: cannot reserve %d bytes of dynamic LDS: %s", what, bytes, hipGetErrorString(e));
    return RTK_ERR_LAUNCH;
}

static inline size_t rtk_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
static inline int64_t rtk_cdiv(int64_t x, int64_t y) { return (x + y - 1) / y; }

// Workspace layout (all offsets 256-B aligned), see rtk_abi.hip::ws_layout
struct RtkWorkspace {
    uint32_t *flags;      // [0] error word, [1] number of distinct relations (n_u)
    int32_t *slot_of_rel; // n_rel : relation id -> table slot (or -1)
    int32_t *rel_list;    // min(n_rel, B) : slot -> relation id
    float *tables;        // n_u_max * b * c : M_u = G x_0 R[u]
    float *v;             // B * c fp32 query vectors
    void *q_packed;       // packed query planes
    void *core_t;         // bf16, relation rank > 32: core transposed to [(b,c)][a]
    void *r_packed;       // bf16, relation rank > 32: packed planes of the batch's relation rows
    int32_t *grp_cnt;     // 2 * n_u_max : queries per table slot; scatter cursor
    int32_t *grp_order;   // B : query ids sorted by table slot
    int32_t *grp_work;    // 4 * (B / 4 + n_u_max) : contract work items (slot, first, count, -)
    int64_t *grp_qinfo;   // 2 * B : per position in slot order (subject id, query id | slot << 32)
    size_t total;
};

// device side -------------------------------------------------------------
#ifdef __HIPCC__
// Operand element types: float, or bf16 carried as its raw 16 bits (torch.bfloat16 storage).
typedef unsigned short rtk_bf16;
__device__ __forceinline__ float rtk_to_f32(float x) { return x; }
__device__ __forceinline__ float rtk_to_f32(rtk_bf16 x) { return __builtin_bit_cast(float, (unsigned)x << 16); }
// round-to-nearest-even fp32 -> bf16 bits (NaN stays NaN: plain cast semantics of v_cvt_pk_bf16_f32)
__device__ __forceinline__ rtk_bf16 rtk_f32_to_bf16(float x) {
    return __builtin_bit_cast(rtk_bf16, (__bf16)x);
}
// 4 consecutive elements -> f32x4 (p must be 16-B aligned for float, 8-B aligned for bf16)
__device__ __forceinline__ f32x4 rtk_load4(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }
__device__ __forceinline__ f32x4 rtk_load4(const rtk_bf16 *p) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 w = *reinterpret_cast<const u32x2 *>(p);
    f32x4 r;
    r[0] = __builtin_bit_cast(float, w[0] << 16);
    r[1] = __builtin_bit_cast(float, w[0] & 0xffff0000u);
    r[2] = __builtin_bit_cast(float, w[1] << 16);
    r[3] = __builtin_bit_cast(float, w[1] & 0xffff0000u);
    return r;
}
template <typename T> __host__ __device__ constexpr int rtk_vec4_align() { return sizeof(T) == 4 ? 16 : 8; }
// Correctly behaving fp32 logistic: 1/(1+exp(-x)) with ocml expf (<= 1 ulp) and an
// IEEE division, the same formula torch's CPU kernel evaluates; saturates to
// exactly 1.0f for x >~ 16.64 like the reference.
__device__ __forceinline__ float rtk_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
#endif
