// Micro-benchmark: does data read by one kernel stay in the XCDs' L2 for the next kernel?
// Each of 256 workgroups (one per CU) reads its own 100 KB slice of a 26 MB buffer once -- the
// score kernel's first-tile ingest.  Launched back to back on the same buffer (same workgroup ->
// XCD mapping), vs alternating between two buffers, vs a 400 MB flush in between.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void ingest(const u32x4 *__restrict__ src, unsigned *sink) {
    extern __shared__ unsigned char lds[];
    const u32x4 *p = src + (size_t)blockIdx.x * 6400;   // 100 KB = 6400 x 16 B
    u32x4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 25; ++i) acc ^= p[i * 256 + threadIdx.x];
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x1234567u) sink[0] = 1;
    if (lds[threadIdx.x] == 77 && sink[1] == 99) sink[2] = 1;
}
__global__ void flush(u32x4 *p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = u32x4{1, 2, 3, 4};
}
int main() {
    u32x4 *a, *b, *big; unsigned *sink;
    const size_t bytes = 256 * 102400;
    (void)hipMalloc(&a, bytes); (void)hipMalloc(&b, bytes); (void)hipMalloc(&big, 400u << 20); (void)hipMalloc(&sink, 16);
    (void)hipMemset(a, 1, bytes); (void)hipMemset(b, 2, bytes); (void)hipMemset(sink, 0, 16);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&ingest), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto timed = [&](const char *name, int mode) {
        float best = 1e9f, sum = 0;
        for (int rep = 0; rep < 20; ++rep) {
            const u32x4 *src = (mode == 1 && (rep & 1)) ? b : a;
            if (mode == 2) hipLaunchKernelGGL(flush, dim3(1024), dim3(256), 0, 0, big, (size_t)(400u << 20) / 16);
            (void)hipEventRecord(e0, 0);
            hipLaunchKernelGGL(ingest, dim3(256), dim3(256), 100 * 1024, 0, src, sink);
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep >= 4) { sum += ms; if (ms < best) best = ms; }
        }
        printf("%-52s mean %.2f us  min %.2f us\n", name, sum / 16 * 1e3, best * 1e3);
    };
    timed("same 26 MB buffer, back to back", 0);
    timed("two buffers alternating (each re-read after 26 MB)", 1);
    timed("same buffer, 400 MB written in between", 2);
    return 0;
}
