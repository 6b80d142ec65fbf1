// Micro-benchmark: what fits into the gap between two v_mfma_f32_32x32x16_f16 of one wave (one wave per SIMD)?
// Cycles per MFMA (s_memtime) for a chain of 64 MFMAs on two alternating accumulators with, in every gap,
//   mode 0: nothing            1: v_mul + v_exp           2: v_mul + v_add         3: v_exp only
//   mode 4: v_mul + v_exp, accumulators in AGPRs (inline-asm MFMA, "a" constraint)
//   mode 5: nothing, accumulators in AGPRs       6: v_mul + v_rcp     7: two v_exp
//   mode 8: v_mul + v_exp + ds_read_b128 (LDS)   9: as 1 but fillers in front of the MFMA (hoisted order)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(512) void k(unsigned long long *out, float *sink) {
    __shared__ f32x4 lds[1024];
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
    lds[threadIdx.x] = f32x4{1.f, 2.f, 3.f, 4.f};
    __syncthreads();
    f32x16 acc0 = {0}, acc1 = {0};
    float x[8], y[8];
    for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 0.01f + i; y[i] = 0.f; }
    f32x4 lv = {0, 0, 0, 0};
    unsigned long long t0, t1;
    asm volatile("" : "+v"(a), "+v"(b));
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    __builtin_amdgcn_sched_barrier(0);
    auto filler = [&](int i) {
        const int s = i & 7;
        if (MODE == 1 || MODE == 4 || MODE == 8 || MODE == 9 || MODE == 11 || MODE == 13) { asm volatile("v_mul_f32 %0, %1, %2" : "=v"(y[s]) : "v"(x[s]), "v"(x[(s + 1) & 7])); asm volatile("v_exp_f32 %0, %1" : "=v"(x[s]) : "v"(y[s])); }
        if (MODE == 2) { asm volatile("v_mul_f32 %0, %1, %2" : "=v"(y[s]) : "v"(x[s]), "v"(x[(s + 1) & 7])); asm volatile("v_add_f32 %0, %1, %2" : "=v"(x[s]) : "v"(y[s]), "v"(x[(s + 2) & 7])); }
        if (MODE == 3) asm volatile("v_exp_f32 %0, %1" : "=v"(x[s]) : "v"(x[(s + 1) & 7]));
        if (MODE == 6) { asm volatile("v_mul_f32 %0, %1, %2" : "=v"(y[s]) : "v"(x[s]), "v"(x[(s + 1) & 7])); asm volatile("v_rcp_f32 %0, %1" : "=v"(x[s]) : "v"(y[s])); }
        if (MODE == 7) { asm volatile("v_exp_f32 %0, %1" : "=v"(y[s]) : "v"(x[(s + 1) & 7])); asm volatile("v_exp_f32 %0, %1" : "=v"(x[s]) : "v"(x[(s + 2) & 7])); }
        if (MODE == 8) lv = lds[(threadIdx.x + i) & 1023];
    };
#pragma unroll
    for (int i = 0; i < 64; ++i) {
        if (MODE == 9) { filler(i); __builtin_amdgcn_sched_barrier(0); }
        if (MODE == 4 || MODE == 5 || MODE == 10 || MODE == 11) {
            if (MODE >= 10) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc0) : "v"(a), "v"(b));
            else if (i & 1) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc1) : "v"(b), "v"(a));
            else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc0) : "v"(a), "v"(b));
        } else if (MODE == 12 || MODE == 13) {   // ONE chain, VGPR accumulator
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
        } else {
            if (i & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc1, 0, 0, 0);
            else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (MODE != 9) { filler(i); __builtin_amdgcn_sched_barrier(0); }
    }
    if (MODE == 4 || MODE == 5 || MODE == 10 || MODE == 11) asm volatile("" : "+a"(acc0), "+a"(acc1));
    else asm volatile("" : "+v"(acc0), "+v"(acc1));
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    __builtin_amdgcn_sched_barrier(0);
    float s = lv[0];
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    for (int i = 0; i < 8; ++i) s += x[i] + y[i];
    if (threadIdx.x % 64 == 0) out[blockIdx.x * 16 + threadIdx.x / 64] = t1 - t0;
    if (s == 12345.f) sink[0] = s;
}
template <int MODE>
void run(unsigned long long *d, float *sink, const char *what) {
    unsigned long long h[16];
    for (int threads : {256, 512}) {
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, d, sink);
            (void)hipDeviceSynchronize();
        }
        (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d (%s), %d waves/SIMD: cycles per MFMA+gap by wave:", MODE, what, threads / 256);
        for (int w = 0; w < threads / 64; ++w) printf(" %.1f", (double)h[w] / 64);
        printf("\n");
    }
}
int main() {
    unsigned long long *d; float *sink;
    (void)hipMalloc(&d, 8 * 4096 * 16); (void)hipMalloc(&sink, 4);
    run<0>(d, sink, "bare chain, VGPR accumulators");
    run<5>(d, sink, "bare chain, AGPR accumulators");
    run<2>(d, sink, "v_mul + v_add per gap");
    run<3>(d, sink, "v_exp per gap");
    run<1>(d, sink, "v_mul + v_exp per gap");
    run<6>(d, sink, "v_mul + v_rcp per gap");
    run<7>(d, sink, "two v_exp per gap");
    run<4>(d, sink, "v_mul + v_exp per gap, AGPR accumulators");
    run<8>(d, sink, "v_mul + v_exp + ds_read_b128 per gap");
    run<9>(d, sink, "v_mul + v_exp in FRONT of each MFMA");
    run<12>(d, sink, "ONE chain, VGPR accumulator, bare");
    run<13>(d, sink, "ONE chain, VGPR accumulator, v_mul + v_exp per gap");
    run<10>(d, sink, "ONE chain, AGPR accumulator, bare");
    run<11>(d, sink, "ONE chain, AGPR accumulator, v_mul + v_exp per gap");
    return 0;
}
