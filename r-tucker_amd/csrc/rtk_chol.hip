// Batched small Cholesky factor AND its inverse in one launch (float64, k <= 256), with the equilibration, the
// diagonal shift and the transposed / rescaled outputs of a Cholesky-QR step fused in: rtk_gram_factor_f64.
//
// The Riemannian optimizer step (r-tucker_amd/{tucker,riemannian,smalllinalg}.py; reference call sites
// src/model/asymmetric/optim.py:86-89,107-108 through tucker_riemopt's round / grad / project) orthonormalises
// by Cholesky QR and inverts the core's r x r Gram matrices ~25 times per step.  rocSOLVER's potrf + rocBLAS trsm
// cost 170 us + 5 launches each and syevd ~15 000 launches; a training step is launch- and latency-bound, not
// flop-bound (k = 200: 5 MFLOP).  Here one 1024-thread workgroup per matrix runs a right-looking blocked
// factorisation (32-wide blocks) on the matrix in its output buffer (L2-resident) with the diagonal block, the
// panel and the pivot rows of the inverse in LDS, and applies the same row operations to an identity, so
// Linv falls out of the same sweep (no second triangular pass, no host round trip, no workspace): the
// explicit inverse turns every "divide by L" of the step into a plain GEMM.  No failure path: a pivot that
// cancelled to (or below) half the diagonal shift (1e-14 of its original diagonal entry without a shift) is replaced by
// that floor -- exact arithmetic on S >= 0 keeps every pivot above the shift, so the floor only acts on a Gram matrix
// whose rounding noise exceeds it (an fp32 Gram matrix of 40 943 rows with dependent columns: noise 1e-5 against a
// shift of 3e-6 gave pivots floored at 1e-14, L entries of 100 behind them and |L^-1| = 1e107) -- the callers
// equilibrate and shift their matrices, so this only triggers on numerically rank-deficient input -- and
// nothing is reported to the host (the training driver reads one health word per epoch).
#include "rtk_common.h"
#include <stdlib.h>

namespace {

constexpr int NB = 32;          // block width
constexpr int KMAX = 256;       // largest matrix (LDS: two k x 32 panels + two 32 x 32 blocks)
constexpr int NT = 1024;
constexpr int GRP = 2;        // global read-modify-writes in flight per thread in the trailing updates (8 spills)

// S: k x k Gram matrix.  A = D^-1 S D^-1 + (shift_diag + shift_trace * trace(S)) I with D = sqrt(diag S) when
// `equil` (else D = I) is factored A = L L^T; outputs (upper triangular)  R = L^T D  and  X = D^-1 L^-T, so that for
// S = W^T W:  W X has orthonormal columns, W = (W X) R, and S^-1 ~ X X^T.  trace(S) <= 0: both outputs are zero.
// Lo / Ti are the output buffers, used as working storage (lower triangles) until the final transposition.
__global__ __launch_bounds__(NT) void chol_inv_kernel(const double *__restrict__ S, int k, int equil, double shift_diag,
                                                      double shift_trace, double *__restrict__ Lo,
                                                      double *__restrict__ Ti, int tune) {
    __shared__ double dd[NB][NB + 1];        // diagonal block of A -> L11
    __shared__ double ti[NB][NB + 1];        // identity -> T11 = L11^-1
    __shared__ double pan[KMAX][NB + 1];     // panel L[i, j-block] for i below the block
    __shared__ double bj[KMAX][NB + 1];      // bj[c][m] = B[j0 + m, c]: the block's rows of the inverse, transposed
    __shared__ double dg[KMAX];              // original diagonal (pivot floor)
    __shared__ double pv[NB];                // pivots = diagonal of L11
    const int t = threadIdx.x;
    const size_t off = (size_t)blockIdx.x * k * k;
    S += off; Lo += off; Ti += off;
    // column scales and the trace
    __shared__ double dsc[KMAX];
    __shared__ double trace_;
    if (t < 64) {
        double part = 0.0;
        for (int i = t; i < k; i += 64) {
            const double sii = S[(size_t)i * k + i];
            part += sii;
            const double d = sqrt(fmax(sii, 0.0));
            dsc[i] = (equil && d > 0.0) ? d : 1.0;
        }
        for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
        if (t == 0) trace_ = part;
    }
    __syncthreads();
    const double tr = trace_;
    if (!(tr > 0.0)) {                       // zero (or not finite-positive) matrix: zero outputs
        for (int e = t; e < k * k; e += NT) Lo[e] = 0.0, Ti[e] = 0.0;
        return;
    }
    const double shift = shift_diag + shift_trace * tr;
    // working copies: Lo = lower triangle of the scaled, shifted matrix (zero above), Ti = identity
    for (int e = t; e < k * k; e += NT) {
        const int i = e / k, c = e - i * k;
        const double a = S[e] / (dsc[i] * dsc[c]) + ((c == i) ? shift : 0.0);
        Lo[e] = (c <= i) ? a : 0.0;
        Ti[e] = (c == i) ? 1.0 : 0.0;
        if (c == i) dg[i] = a;
    }
    __syncthreads();
    const int r = t >> 5, cc = t & 31;       // (row, column) inside a 32 x 32 block
    for (int j0 = 0; j0 < k; j0 += NB) {
        const int w = min(NB, k - j0);
        const int below = k - j0 - w;        // rows under the block
        const int left = j0 + w;             // live columns of the inverse's rows j0 .. j0 + w - 1
        // --- A) diagonal block and an identity into LDS
        dd[r][cc] = (r < w && cc <= r) ? Lo[(size_t)(j0 + r) * k + j0 + cc] : 0.0;
        ti[r][cc] = (r == cc) ? 1.0 : 0.0;
        __syncthreads();
        for (int c = 0; c < ((tune & 1) ? 0 : w); ++c) {
            __syncthreads();                 // the previous column's updates are visible
            const double floor_ = fmax(1e-14 * fabs(dg[j0 + c]), 0.5 * shift) + 1e-300;
            const double piv = sqrt(fmax(dd[c][c], floor_));     // dd[c][c] itself is left alone (pv holds L[c][c])
            const double ip = 1.0 / piv;
            if (cc == c && r > c && r < w) dd[r][c] *= ip;
            if (r == c && cc <= c) ti[c][cc] *= ip;
            if (r == c && cc == c) pv[c] = piv;
            __syncthreads();
            if (r > c && r < w) {
                const double l = dd[r][c];
                if (cc > c && cc <= r) dd[r][cc] -= l * dd[cc][c];
                if (cc <= c) ti[r][cc] -= l * ti[c][cc];
            }
        }
        __syncthreads();
        // L11 out; rows of the inverse: B[j-block, 0:left] <- T11 * B[j-block, 0:left]  (their old diagonal
        // block is the identity, columns < j0 hold the updates of the earlier steps)
        if (r < w && cc <= r) Lo[(size_t)(j0 + r) * k + j0 + cc] = (cc == r) ? pv[r] : dd[r][cc];
        for (int e = t; e < ((tune & 2) ? 0 : left * NB); e += NT) {   // stage the old rows: bj[c][m] = B[j0 + m][c]
            const int m = e / left, c = e - m * left;
            bj[c][m] = (m < w) ? Ti[(size_t)(j0 + m) * k + c] : 0.0;
        }
        __syncthreads();
        {
            // new row m of the block: sum_{m' <= m} T11[m][m'] * old[m'][c]
            constexpr int NA = (KMAX * NB + NT - 1) / NT;
            double acc[NA];
#pragma unroll
            for (int n = 0; n < NA; ++n) {
                const int e = t + n * NT;
                const int c = e >> 5, m = e & 31;
                double s = 0.0;
                if (e < left * NB && m < w)
                    for (int mp = 0; mp <= m; ++mp) s += ti[m][mp] * bj[c][mp];
                acc[n] = s;
            }
            __syncthreads();
#pragma unroll
            for (int n = 0; n < NA; ++n) {
                const int e = t + n * NT;
                const int c = e >> 5, m = e & 31;
                if (e < left * NB) {
                    bj[c][m] = acc[n];
                    if (m < w) Ti[(size_t)(j0 + m) * k + c] = acc[n];
                }
            }
        }
        // --- B) panel: L[i, j-block] = A[i, j-block] * T11^T   (i below the block)
        for (int e = t; e < ((tune & 4) ? 0 : below * NB); e += NT) {
            const int i = e >> 5, c = e & 31;
            double s = 0.0;
            if (c < w) {
                const double *arow = Lo + (size_t)(j0 + w + i) * k + j0;
                for (int m = 0; m <= c; ++m) s += arow[m] * ti[c][m];
            }
            pan[i][c] = s;
        }
        __syncthreads();
        for (int e = t; e < below * NB; e += NT) {
            const int i = e >> 5, c = e & 31;
            if (c < w) Lo[(size_t)(j0 + w + i) * k + j0 + c] = pan[i][c];
        }
        // --- C) trailing updates with the panel: the rest of A (lower triangle) and the inverse's rows below.
        // Read-modify-writes of global memory in groups of eight: the loads of a group are in flight together
        // (one after the other, each waiting for the previous store, they were 80 % of the kernel).
        // A[i, c2] -= sum_m pan[i][m] pan[c2][m],  j0 + w <= c2 <= i
        for (int e0 = t; e0 < ((tune & 8) ? 0 : below * below); e0 += GRP * NT) {
            double old[GRP], s8[GRP];
            size_t at[GRP];
            bool on[GRP];
#pragma unroll
            for (int u = 0; u < GRP; ++u) {
                const int e = e0 + u * NT;
                const int i = e / below, c2 = e - i * below;
                on[u] = e < below * below && c2 <= i;
                at[u] = (size_t)(j0 + w + i) * k + j0 + w + c2;
                old[u] = on[u] ? Lo[at[u]] : 0.0;
                double sacc = 0.0;
                if (on[u]) {
#pragma unroll 8
                    for (int m = 0; m < NB; ++m) sacc += pan[i][m] * pan[c2][m];
                }
                s8[u] = sacc;
            }
#pragma unroll
            for (int u = 0; u < GRP; ++u)
                if (on[u]) Lo[at[u]] = old[u] - s8[u];
        }
        // B[i, c] -= sum_m pan[i][m] * B[j0 + m, c],  c < left
        for (int e0 = t; e0 < ((tune & 16) ? 0 : below * left); e0 += GRP * NT) {
            double old[GRP], s8[GRP];
            size_t at[GRP];
            bool on[GRP];
#pragma unroll
            for (int u = 0; u < GRP; ++u) {
                const int e = e0 + u * NT;
                const int i = e / left, c = e - i * left;
                on[u] = e < below * left;
                at[u] = (size_t)(j0 + w + i) * k + c;
                old[u] = on[u] ? Ti[at[u]] : 0.0;
                double sacc = 0.0;
                if (on[u]) {
#pragma unroll 8
                    for (int m = 0; m < NB; ++m) sacc += pan[i][m] * bj[c][m];
                }
                s8[u] = sacc;
            }
#pragma unroll
            for (int u = 0; u < GRP; ++u)
                if (on[u]) Ti[at[u]] = old[u] - s8[u];
        }
        __syncthreads();
    }
    // R = L^T D (upper), X = D^-1 L^-T (upper): transpose in place, one thread per (i <= c) pair
    for (int e = t; e < k * k; e += NT) {
        const int i = e / k, c = e - i * k;
        if (i > c) continue;
        const double l = Lo[(size_t)c * k + i], x = Ti[(size_t)c * k + i];
        if (i != c) Lo[(size_t)c * k + i] = 0.0, Ti[(size_t)c * k + i] = 0.0;
        Lo[(size_t)i * k + c] = l * dsc[c];
        Ti[(size_t)i * k + c] = x / dsc[i];
    }
}

}  // namespace

extern "C" int rtk_gram_factor_f64(const void *S, int64_t batch, int k, int equilibrate, double shift_diag,
                                   double shift_trace, void *R_out, void *X_out, void *stream) {
    RTK_REQUIRE(S && R_out && X_out, RTK_ERR_BAD_ARG, "rtk_gram_factor_f64: null operand");
    RTK_REQUIRE(batch > 0 && batch < (1ll << 20), RTK_ERR_BAD_ARG, "rtk_gram_factor_f64: batch must be in [1, 2^20)");
    RTK_REQUIRE(k > 0 && k <= KMAX, RTK_ERR_UNSUPPORTED, "rtk_gram_factor_f64: k=%d not in [1, %d]", k, KMAX);
    RTK_REQUIRE(S != R_out && S != X_out && R_out != X_out, RTK_ERR_BAD_ARG, "rtk_gram_factor_f64: buffers must be distinct");
    RTK_REQUIRE(shift_diag >= 0.0 && shift_trace >= 0.0, RTK_ERR_BAD_ARG, "rtk_gram_factor_f64: shifts must be >= 0");
    static const int tune = getenv("RTK_CHOL_TUNE") ? atoi(getenv("RTK_CHOL_TUNE")) : 0;   // timing ablations (wrong results)
    hipLaunchKernelGGL(chol_inv_kernel, dim3((unsigned)batch), dim3(NT), 0, (hipStream_t)stream, (const double *)S, k,
                       equilibrate ? 1 : 0, shift_diag, shift_trace, (double *)R_out, (double *)X_out, tune);
    return rtk_check_launch("rtk_gram_factor_f64");
}
