// Standalone correctness + timing harness for the deep-K bf16 score kernel (rtk_score_bf16_w1.h): random bf16
// operands, packed query planes built on the host, every variant timed with HIP events and checked on sampled
// entries against a float64 dot product of the same bf16 values.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../../include -I../../../r-tucker_amd/csrc -I. w1_bench.hip -o w1_bench
//   ./w1_bench N B c [reps]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
void rtk_set_error(const char *, ...) {}
#include "rtk_score_bf16_w1.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); u = (u + 0x7fff + ((u >> 16) & 1)) >> 16; return (unsigned short)u; }
static float bf2f(unsigned short b) { unsigned u = (unsigned)b << 16; float f; memcpy(&f, &u, 4); return f; }

template <int KS, int SG, bool NTS, bool TR, unsigned ABL = 0>
double run(const char *name, const unsigned char *qp, int B, const rtk_bf16 *O, int N, int c, float *out, int64_t ld, int qb, int reps,
           const std::vector<unsigned short> &hO, const std::vector<unsigned short> &hV, std::vector<float> &hout) {
    constexpr size_t smem = 3 * (size_t)KS * 1024;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&rtk_w1::score_bf16_w1_kernel<KS, SG, NTS, TR, ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    const int n_mt = (B + 31) / 32;
    if (qb > n_mt) qb = n_mt;
    const long units = (long)((N + 255) / 256) * qb;
    const unsigned grid = (unsigned)(units < 256 ? units : 256);
    CK(hipMemset(out, 0xff, (size_t)B * ld * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((rtk_w1::score_bf16_w1_kernel<KS, SG, NTS, TR, ABL>), dim3(grid), dim3(256), smem, 0, qp, B, O, N, c, out, ld, c % 8 == 0, qb);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((rtk_w1::score_bf16_w1_kernel<KS, SG, NTS, TR, ABL>), dim3(grid), dim3(256), smem, 0, qp, B, O, N, c, out, ld, c % 8 == 0, qb);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(hout.data(), out, (size_t)B * ld * 4, hipMemcpyDeviceToHost));
    // check: rows {0, 1, 31, 32, B/2, B-1} x every 97th column and the last 300 columns
    double worst = 0; long bad = 0, checked = 0;
    int rows[] = {0, 1, 31 % B, 32 % B, B / 2, B - 1};
    for (int d : rows)
        for (int j = 0; j < N; j += (j < N - 300 ? 97 : 1)) {
            double z = 0, az = 0;
            for (int k = 0; k < c; ++k) { const double a = bf2f(hV[(size_t)d * c + k]), b = bf2f(hO[(size_t)j * c + k]); z += a * b; az += fabs(a * b); }
            const double want = SG ? 1.0 / (1.0 + exp(-z)) : z;
            const double got = hout[(size_t)d * ld + j];
            const double tol = (SG ? 0.25 : 1.0) * 1e-5 * az + 1e-6;
            const double err = fabs(got - want);
            if (!(err <= tol)) { if (bad < 5) printf("   MISMATCH d=%d j=%d got %.7g want %.7g\n", d, j, got, want); ++bad; }
            worst = fmax(worst, err / tol); ++checked;
        }
    const double bytes = (double)N * c * 2 + (double)B * N * 4 + (double)B * c * 2;
    printf("%-28s %8.4f ms  %6.2f TB/s  frac %.3f   checked %ld bad %ld worst/tol %.3f\n", name, ms / reps, bytes / (ms / reps * 1e-3) / 1e12,
           bytes / (ms / reps * 1e-3) / 8e12, checked, bad, worst);
    return ms / reps;
}

int main(int argc, char **argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 125000, B = argc > 2 ? atoi(argv[2]) : 8192, c = argc > 3 ? atoi(argv[3]) : 512;
    const int reps = argc > 4 ? atoi(argv[4]) : 20;
    const int KS = (c + 15) / 16, n_mt = (B + 31) / 32;
    const int64_t ld = ((N + 31) / 32) * 32;
    srand(1);
    std::vector<unsigned short> hO((size_t)N * c), hV((size_t)B * c);
    for (auto &x : hO) x = f2bf((rand() / (float)RAND_MAX - 0.5f) * 2.f);
    for (auto &x : hV) x = f2bf((rand() / (float)RAND_MAX - 0.5f) * 0.4f);
    const size_t tile = RTK_PACK_HDR + (size_t)KS * 1024;
    std::vector<unsigned char> hq(n_mt * tile, 0);
    for (int d = 0; d < B; ++d)
        for (int k = 0; k < c; ++k) {
            unsigned short *plane = reinterpret_cast<unsigned short *>(hq.data() + (d / 32) * tile + RTK_PACK_HDR);
            plane[rtk_pack_offset(KS, k, d % 32)] = hV[(size_t)d * c + k];
        }
    unsigned char *qp; rtk_bf16 *O; float *out;
    CK(hipMalloc(&qp, hq.size())); CK(hipMalloc(&O, hO.size() * 2)); CK(hipMalloc(&out, (size_t)B * ld * 4));
    CK(hipMemcpy(qp, hq.data(), hq.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(O, hO.data(), hO.size() * 2, hipMemcpyHostToDevice));
    std::vector<float> hout((size_t)B * ld);
    int qb = (int)((3072u << 10) / tile);
    printf("N %d B %d c %d (KS %d) ld %ld  qb %d\n", N, B, c, KS, (long)ld, qb);
#define RUNKS(K_)                                                                                                    \
    if (KS == K_) {                                                                                                  \
        run<K_, 2, true, false>("std stores, nt", qp, B, O, N, c, out, ld, qb, reps, hO, hV, hout);                  \
        run<K_, 2, false, false>("std stores, plain", qp, B, O, N, c, out, ld, qb, reps, hO, hV, hout);              \
        run<K_, 2, false, true>("transposed 16B, plain", qp, B, O, N, c, out, ld, qb, reps, hO, hV, hout);           \
        run<K_, 0, true, false>("std stores, nt, logits", qp, B, O, N, c, out, ld, qb, reps, hO, hV, hout);          \
        run<K_, 2, true, false>("std, nt, unblocked", qp, B, O, N, c, out, ld, 1 << 20, reps, hO, hV, hout);         \
        if (K_ == 32) {                                                                                              \
            run<K_, 2, true, false, 1>("ABL no stores", qp, B, O, N, c, out, ld, qb, reps, hO, hV, hout);            \
            run<K_, 2, true, false, 2>("ABL no fragment reads", qp, B, O, N, c, out, ld, qb, reps, hO, hV, hout);    \
            run<K_, 2, true, false, 4>("ABL no staging", qp, B, O, N, c, out, ld, qb, reps, hO, hV, hout);           \
            run<K_, 2, true, false, 8>("ABL no barrier", qp, B, O, N, c, out, ld, qb, reps, hO, hV, hout);           \
            run<K_, 0, true, false, 15>("ABL all four, logits", qp, B, O, N, c, out, ld, qb, reps, hO, hV, hout);    \
            run<K_, 0, true, false, 7>("ABL 1+2+4, logits", qp, B, O, N, c, out, ld, qb, reps, hO, hV, hout);        \
        }                                                                                                            \
    }
    RUNKS(32) RUNKS(24) RUNKS(20)
    return 0;
}
