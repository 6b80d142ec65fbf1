// Micro-benchmark: cycles per dependent v_mfma_f32_32x32x16_{f16,bf16} on one SIMD (s_memtime).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ void k(unsigned long long *out, float *sink, int waves_per_simd) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
    bf16x8 ab, bb;
    for (int i = 0; i < 8; ++i) { ab[i] = (short)(0x3f80 + i); bb[i] = (short)(0x3f00 + threadIdx.x); }
    f32x16 acc0 = {0}, acc1 = {0};
    unsigned long long t0, t1;
    asm volatile("" : "+v"(a), "+v"(b));
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 64; ++i) {
        if (MODE == 0) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
        if (MODE == 1) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc0, 0, 0, 0);
        if (MODE == 2) { acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc1, 0, 0, 0); }
    }
    asm volatile("" : "+v"(acc0), "+v"(acc1));
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    __builtin_amdgcn_sched_barrier(0);
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    if (threadIdx.x % 64 == 0) out[blockIdx.x * 16 + threadIdx.x / 64] = t1 - t0;
    if (s == 12345.f) sink[0] = s;
}
int main() {
    unsigned long long *d; float *sink;
    (void)hipMalloc(&d, 8 * 4096 * 16); (void)hipMalloc(&sink, 4);
    unsigned long long h[16];
    for (int mode = 0; mode < 3; ++mode)
        for (int threads : {256, 512}) {
            for (int rep = 0; rep < 2; ++rep) {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(threads), 0, 0, d, sink, 1);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(threads), 0, 0, d, sink, 1);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(threads), 0, 0, d, sink, 1);
                (void)hipDeviceSynchronize();
            }
            (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
            printf("mode %d (%s) threads/block %d: cycles per MFMA by wave:", mode,
                   mode == 0 ? "f16 dependent chain" : mode == 1 ? "bf16 dependent chain" : "f16 two independent chains", threads);
            for (int w = 0; w < threads / 64; ++w) printf(" %.1f", (double)h[w] / (mode == 2 ? 128 : 64));
            printf("\n");
        }
    return 0;
}
