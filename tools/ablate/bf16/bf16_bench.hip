// Standalone timing + spot-check harness for the library's bf16 score kernel (rtk_score_bf16.hip, one k-step count
// instantiated: fast to build): random bf16 operands, packed planes built on the host, HIP events, float64 checks.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DRTK_BF16_HARNESS_KS=32 -I../../../include -I../../../r-tucker_amd/csrc bf16_bench.hip -o bf16_bench
//   ./bf16_bench N B c reps          (RTK_BF16_V1=1: round 2's loop)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
void rtk_set_error(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
#include "rtk_score_bf16.hip"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); u = (u + 0x7fff + ((u >> 16) & 1)) >> 16; return (unsigned short)u; }
static float bf2f(unsigned short b) { unsigned u = (unsigned)b << 16; float f; memcpy(&f, &u, 4); return f; }
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
        for (int k = 0; k < c; ++k)
            reinterpret_cast<unsigned short *>(hq.data() + (d / 32) * tile + RTK_PACK_HDR)[rtk_pack_offset(KS, k, d % 32)] = hV[(size_t)d * c + k];
    unsigned char *qp; rtk_bf16 *O; float *out;
    CK(hipMalloc(&qp, hq.size())); CK(hipMalloc(&O, hO.size() * 2)); CK(hipMalloc(&out, (size_t)B * ld * 4));
    CK(hipMemcpy(qp, hq.data(), hq.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(O, hO.data(), hO.size() * 2, hipMemcpyHostToDevice));
    std::vector<float> hout((size_t)B * ld);
    for (unsigned flags : {RTK_SCORE_SIGMOID | RTK_SCORE_SIGMOID_FAST, 0u}) {
        CK(hipMemset(out, 0xff, (size_t)B * ld * 4));
        for (int w = 0; w < 3; ++w) if (rtk_score_packed_bf16(qp, B, c, O, N, out, ld, flags, nullptr) != 0) return 1;
        CK(hipDeviceSynchronize());
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        for (int w = 0; w < reps; ++w) rtk_score_packed_bf16(qp, B, c, O, N, out, ld, flags, nullptr);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(hout.data(), out, (size_t)B * ld * 4, hipMemcpyDeviceToHost));
        long bad = 0, checked = 0; double worst = 0;
        int rows[] = {0, 1, 31 % B, 32 % B, 33 % B, B / 2, B - 1};
        for (int d : rows)
            for (int j = 0; j < N; j += (j < N - 300 ? 97 : 1)) {
                double z = 0, az = 0;
                for (int k = 0; k < c; ++k) { const double a = bf2f(hV[(size_t)d * c + k]), b = bf2f(hO[(size_t)j * c + k]); z += a * b; az += fabs(a * b); }
                const double want = flags ? 1.0 / (1.0 + exp(-z)) : z, got = hout[(size_t)d * ld + j];
                const double tol = (flags ? 0.25 : 1.0) * 1e-5 * az + 1e-6, err = fabs(got - want);
                if (!(err <= tol)) { if (bad < 3) printf("   MISMATCH d=%d j=%d got %.7g want %.7g\n", d, j, got, want); ++bad; }
                worst = fmax(worst, err / tol); ++checked;
            }
        const double bytes = (double)N * c * 2 + (double)B * N * 4 + (double)B * c * 2;
        printf("%-10s N %d B %d c %d: %8.4f ms  frac %.3f   checked %ld bad %ld worst/tol %.3f\n", flags ? "logistic" : "logits", N, B, c, ms / reps,
               bytes / (ms / reps * 1e-3) / 8e12, checked, bad, worst);
    }
    return 0;
}
