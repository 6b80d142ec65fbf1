// Micro-benchmark: per-CU vector-memory throughput of the score kernel's traffic mix on gfx950:
// L2-resident 1-KiB wave loads (packed query tiles) and HBM-bound row-segment stores (the score matrix),
// issued by 4 waves of one resident workgroup per CU, all 256 CUs at once.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int N = 40943, B = 512, NT = 320, QBYTES = 16 * 26752;

// MODE bit 0: loads, bit 1: b32 row-segment stores, bit 2: x4 stores (8 rows x 128 B), bit 3: linear 1-KiB stores,
// bit 4: stores nontemporal
template <int MODE>
__global__ __launch_bounds__(256) void k(const unsigned char *__restrict__ q, float *__restrict__ out, int64_t ld,
                                         int iters, unsigned *sink) {
    extern __shared__ unsigned char lds[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, b = blockIdx.x;
    u32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        const int mt = it % 16, ntile = (b + 256 * (it / 16)) % NT;
        if (MODE & 1) {
            u32x4 x[7];
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                const int piece = (i * 4 + w) % 26;
                x[i] = *reinterpret_cast<const u32x4 *>(q + (size_t)mt * 26752 + piece * 1024 + lane * 16);
            }
#pragma unroll
            for (int i = 0; i < 7; ++i) acc ^= x[i];
        }
        const float v = __builtin_bit_cast(float, 0x3f000000u + (unsigned)it);
        if (MODE & 2) {
            const int j = ntile * 128 + w * 32 + (lane & 31);
            if (j < N) {
                float *p = out + (int64_t)(mt * 32 + 4 * (lane >> 5)) * ld + j;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    if (MODE & 16) __builtin_nontemporal_store(v, p);
                    else *p = v;
                    p += ((e & 3) == 3) ? 5 * ld : ld;
                }
            }
        }
        if (MODE & 4) {
            const int j0 = ntile * 128 + w * 32 + 4 * (lane & 7);
            if (j0 + 4 <= N) {
                float *p = out + (int64_t)(mt * 32 + (lane >> 3)) * ld + j0;
                const u32x4 vv = {__builtin_bit_cast(unsigned, v), 1u, 2u, 3u};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (MODE & 16) __builtin_nontemporal_store(vv, reinterpret_cast<u32x4 *>(p));
                    else *reinterpret_cast<u32x4 *>(p) = vv;
                    p += 8 * ld;
                }
            }
        }
        if (MODE & 8) {   // same bytes per iteration (4 KiB per wave), fully linear
            unsigned char *p = reinterpret_cast<unsigned char *>(out) + ((size_t)(it % 20) * 1024 + b * 4 + w) * 4096 + lane * 16;
            const u32x4 vv = {__builtin_bit_cast(unsigned, v), 1u, 2u, 3u};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (MODE & 16) __builtin_nontemporal_store(vv, reinterpret_cast<u32x4 *>(p + e * 1024));
                else *reinterpret_cast<u32x4 *>(p + e * 1024) = vv;
            }
        }
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) sink[0] = 1;
    if (lds[threadIdx.x] == 77 && iters < 0) sink[1] = 1;
}

template <int MODE>
void run(const char *name, const unsigned char *q, float *out, int64_t ld, unsigned *sink) {
    const int iters = 320;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 100 * 1024, 0, q, out, ld, iters, sink);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double ld_bytes = (MODE & 1) ? 256.0 * iters * 4 * 7 * 1024 : 0, st_bytes = (MODE & 14) ? 256.0 * iters * 4 * 4096 : 0;
    printf("%-44s ld=%lld: %.1f us  per CU: %.2f B/ns (loads %.2f, stores %.2f)  chip %.2f TB/s  per iteration %.0f ns\n", name, (long long)ld,
           best * 1e3, (ld_bytes + st_bytes) / 256 / (best * 1e6), ld_bytes / 256 / (best * 1e6), st_bytes / 256 / (best * 1e6),
           (ld_bytes + st_bytes) / (best * 1e9), best * 1e6 / iters);
}

// one pass over the score matrix (every line written once), same buffer every launch vs 12 rotating buffers
template <int MODE>
void run_once(const char *name, const unsigned char *q, float *base, size_t stride_floats, int nbuf, int64_t ld, unsigned *sink) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int launches = 24;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0, 0);
        for (int l = 0; l < launches; ++l)
            hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 100 * 1024, 0, q, base + (size_t)(l % nbuf) * stride_floats, ld, 20, sink);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep == 1) printf("%-40s ld=%lld buffers=%2d: %.1f us per launch (84 MB written once per launch) = %.2f TB/s\n", name, (long long)ld, nbuf,
               ms * 1e3 / launches, 256.0 * 20 * 4 * 4096 / (ms / launches * 1e9));
    }
}

int main() {
    unsigned char *q; float *out; unsigned *sink;
    const int64_t ldmax = 40960;
    (void)hipMalloc(&q, QBYTES + 4096); (void)hipMemset(q, 1, QBYTES + 4096);
    (void)hipMalloc(&out, (size_t)B * ldmax * 4 + (1 << 20)); (void)hipMalloc(&sink, 8);
    for (int64_t ld : {(int64_t)N, ldmax}) {
        run<1>("loads only (7 x 1 KiB per wave, L2-resident)", q, out, ld, sink);
        run<2>("b32 row-segment stores only", q, out, ld, sink);
        run<4>("x4 stores (8 rows x 128 B) only", q, out, ld, sink);
        run<8>("linear x4 stores only", q, out, ld, sink);
        run<3>("loads + b32 stores", q, out, ld, sink);
        run<5>("loads + x4 stores", q, out, ld, sink);
        run<9>("loads + linear stores", q, out, ld, sink);
        run<18>("b32 stores, nontemporal", q, out, ld, sink);
        run<19>("loads + b32 stores, nontemporal", q, out, ld, sink);
        run<21>("loads + x4 stores, nontemporal", q, out, ld, sink);
    }
    float *big;
    const size_t stride = (size_t)B * ldmax + (1 << 18);
    (void)hipMalloc(&big, stride * 4 * 12);
    for (int64_t ld : {(int64_t)N, ldmax})
        for (int nbuf : {1, 12}) {
            run_once<2>("one pass, b32 row-segment stores", q, big, stride, nbuf, ld, sink);
            run_once<4>("one pass, x4 stores", q, big, stride, nbuf, ld, sink);
            run_once<8>("one pass, linear stores", q, big, stride, nbuf, ld, sink);
            run_once<3>("one pass, loads + b32 stores", q, big, stride, nbuf, ld, sink);
            run_once<18>("one pass, b32 stores nontemporal", q, big, stride, nbuf, ld, sink);
        }
    {   // reference: device fill of the same 84 MB
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        for (int nbuf : {1, 12}) {
            (void)hipEventRecord(e0, 0);
            for (int l = 0; l < 24; ++l) (void)hipMemsetAsync(big + (size_t)(l % nbuf) * stride, 0, (size_t)B * N * 4, 0);
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            printf("hipMemsetAsync 84 MB, buffers=%2d: %.1f us per call = %.2f TB/s\n", nbuf, ms * 1e3 / 24, (double)B * N * 4 / (ms / 24 * 1e9));
        }
    }
    return 0;
}
