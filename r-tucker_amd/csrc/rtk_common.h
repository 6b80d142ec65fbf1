// Shared helpers for the gfx950 kernels of librtucker_hip.so (internal header).
#pragma once
#include <hip/hip_runtime.h>
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

static inline int rtk_check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        rtk_set_error("%s: %s", what, hipGetErrorString(e));
        return RTK_ERR_LAUNCH;
    }
    return RTK_OK;
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
    size_t total;
};

// device side -------------------------------------------------------------
#ifdef __HIPCC__
// Correctly behaving fp32 logistic: 1/(1+exp(-x)) with ocml expf (<= 1 ulp) and an
// IEEE division, the same formula torch's CPU kernel evaluates; saturates to
// exactly 1.0f for x >~ 16.64 like the reference.
__device__ __forceinline__ float rtk_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
#endif
