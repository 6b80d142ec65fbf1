// Batched small Cholesky factor AND its inverse in one launch (float64, k <= 256), with the equilibration, the
// diagonal shift and the transposed / rescaled outputs of a Cholesky-QR step fused in: rtk_gram_factor_f64.
//
// The Riemannian optimizer step (r-tucker_amd/{tucker,riemannian,smalllinalg}.py; reference call sites
// src/model/asymmetric/optim.py:86-89,107-108 through tucker_riemopt's round / grad / project) orthonormalises
// by Cholesky QR and inverts the core's r x r Gram matrices ~25 times per step.  rocSOLVER's potrf + rocBLAS trsm
// cost 170 us + 5 launches each and syevd ~15 000 launches; a training step is launch- and latency-bound, not
// flop-bound (k = 200: 5 MFLOP).  Here one 1024-thread workgroup per matrix runs a right-looking blocked
// factorisation (32-wide blocks) on the matrix in its output buffer (L2-resident) with the
// panel and the pivot rows of the inverse in LDS, and applies the same row operations to an identity, so
// Linv falls out of the same sweep (no second triangular pass, no host round trip, no workspace): the
// explicit inverse turns every "divide by L" of the step into a plain GEMM.  No failure path: a pivot that
// cancelled to (or below) half the diagonal shift (1e-14 of its original diagonal entry without a shift) is replaced by
// that floor -- exact arithmetic on S >= 0 keeps every pivot above the shift, so the floor only acts on a Gram matrix
// whose rounding noise exceeds it (an fp32 Gram matrix of 40 943 rows with dependent columns: noise 1e-5 against a
// shift of 3e-6 gave pivots floored at 1e-14, L entries of 100 behind them and |L^-1| = 1e107) -- the callers
// equilibrate and shift their matrices, so this only triggers on numerically rank-deficient input -- and
// nothing is reported to the host (the training driver reads one health word per epoch).
//
// Round 4 (513 -> 248 us per k = 200 matrix, 263 -> 127 us averaged over a training step's 25 launches;
// profiles/r04_chol_phase_ablations.txt has the phase-by-phase ablations of the old form).  What the time was: with one
// output element per thread and both operands of every multiply-add read from LDS, the panel, trailing and inverse-row
// updates (6 M multiply-adds at k = 200) moved 100 MB through LDS -- the launch was LDS-bound, not latency-bound (eight
// global loads in flight per thread instead of two changed nothing).  All of them are "rows of X times rows of Y,
// transposed" over the block's 32 columns and now run on the f64 matrix pipe, v_mfma_f64_16x16x4_f64, 16 x 16 tiles
// dealt round-robin to the sixteen waves (one LDS read per operand per instruction: 1/16 of the traffic); the panel's
// rows come through LDS in one batch of loads (a dot product over `arow[m]` in memory was 32 dependent trips to L2).
// The diagonal block is factored with one element per thread in REGISTERS, one barrier per column instead of two, and
// the pivot's reciprocal square root comes from v_rsq_f64 + two Newton steps (every thread computes it; the IEEE sqrt and
// division sequences were most of a column step).  The 32 x 7 sequential column steps are what is left (106 us).
#include "rtk_common.h"
#include <stdlib.h>

namespace {

typedef double f64x4 __attribute__((ext_vector_type(4)));

constexpr int NB = 32;          // block width
constexpr int KMAX = 256;       // largest matrix (LDS: two k x 32 panels + two 32 x 32 blocks)
constexpr int NT = 1024;
constexpr int IB = 8;         // loads in flight per thread in the copy-in / copy-out passes

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
    __shared__ double colb[2][NB], rowb[2][NB];   // a column step's column of the block / row of the inverse
    const int t = threadIdx.x;
    const size_t off = (size_t)blockIdx.x * k * k;
    S += off; Lo += off; Ti += off;
    // column scales and the trace
    __shared__ double dsc[KMAX], idsc[KMAX];     // column scales and their reciprocals
    __shared__ double trace_;
    if (t < 64) {
        double part = 0.0;
        for (int i = t; i < k; i += 64) {
            const double sii = S[(size_t)i * k + i];
            part += sii;
            const double d = sqrt(fmax(sii, 0.0));
            dsc[i] = (equil && d > 0.0) ? d : 1.0;
            idsc[i] = (equil && d > 0.0) ? 1.0 / d : 1.0;
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
    // (IB loads of a thread in flight together: one element per trip was k*k/1024 dependent trips to L2)
    for (int e0 = t; e0 < k * k; e0 += IB * NT) {
        double sv[IB];
#pragma unroll
        for (int u = 0; u < IB; ++u) {
            const int e = e0 + u * NT;
            sv[u] = e < k * k ? S[e] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < IB; ++u) {
            const int e = e0 + u * NT;
            if (e < k * k) {
                const int i = e / k, c = e - i * k;
                const double a = sv[u] * idsc[i] * idsc[c] + ((c == i) ? shift : 0.0);
                Lo[e] = (c <= i) ? a : 0.0;
                Ti[e] = (c == i) ? 1.0 : 0.0;
                if (c == i) dg[i] = a;
            }
        }
    }
    __syncthreads();
    const int r = t >> 5, cc = t & 31;       // (row, column) inside a 32 x 32 block
    for (int j0 = 0; j0 < k; j0 += NB) {
        const int w = min(NB, k - j0);
        const int below = k - j0 - w;        // rows under the block
        const int left = j0 + w;             // live columns of the inverse's rows j0 .. j0 + w - 1
        // --- A) the diagonal block, one element of it and of the identity beside it per thread, IN REGISTERS: a column
        // step publishes the block's column c and the inverse's row c (two 32-vectors, double-buffered), one barrier, and
        // every thread updates its own two numbers (with the block in LDS a step was two barriers and a read-modify-write
        // of shared memory: 129 us of a 513 us launch at k = 200).  Same operations on the same values as before.
        {
            double d = (r < w && cc <= r) ? Lo[(size_t)(j0 + r) * k + j0 + cc] : 0.0;
            double tv = (r == cc) ? 1.0 : 0.0;
            for (int c = 0; c < ((tune & 1) ? 0 : w); ++c) {
                const int par = c & 1;
                if (cc == c) colb[par][r] = d;
                if (r == c) rowb[par][cc] = tv;
                __syncthreads();
                const double floor_ = fmax(1e-14 * fabs(dg[j0 + c]), 0.5 * shift) + 1e-300;
                // pivot and its reciprocal from v_rsq_f64 + two Newton steps (every thread of the block computes them: the
                // IEEE sqrt and division sequences were most of a column step); piv gets one correction of its own
                const double x = fmax(colb[par][c], floor_);             // the pivot d(c, c) itself is left alone (pv holds L[c][c])
                const double hx = 0.5 * x;
                double ip = __builtin_amdgcn_rsq(x);
                ip = fma(ip, fma(-hx * ip, ip, 0.5), ip);
                ip = fma(ip, fma(-hx * ip, ip, 0.5), ip);
                double piv = x * ip;
                piv = fma(fma(-piv, piv, x), 0.5 * ip, piv);
                const double lr = colb[par][r] * ip, lc = colb[par][cc] * ip, tc = rowb[par][cc] * ip;
                if (cc == c && r > c && r < w) d = lr;
                if (r == c && cc <= c) tv = tc;
                if (r == c && cc == c) pv[c] = piv;
                if (r > c && r < w) {
                    if (cc > c && cc <= r) d -= lr * lc;
                    if (cc <= c) tv -= lr * tc;
                }
            }
            dd[r][cc] = d;
            ti[r][cc] = tv;
        }
        __syncthreads();
        // L11 out
        if (r < w && cc <= r) Lo[(size_t)(j0 + r) * k + j0 + cc] = (cc == r) ? pv[r] : dd[r][cc];
        // Everything below is "rows of X times rows of Y, transposed": D[i][j] = sum_m X[i][m] * Y[j][m] over the 32 columns
        // of the block, on the f64 matrix pipe (v_mfma_f64_16x16x4_f64: 16 x 16 tiles, eight instructions per tile; one
        // LDS read per operand per instruction instead of two per multiply-add -- with one output element per thread the
        // 6 M multiply-adds of a k = 200 factorisation were bound by 100 MB of LDS reads: profiles/r04_chol_phase_ablations.txt).
        const int lane = t & 63, wv = t >> 6, li = lane & 15, lq = lane >> 4;
        auto tile = [&](const double (*X)[NB + 1], int x0, const double (*Y)[NB + 1], int y0) {
            f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kk = 0; kk < NB / 4; ++kk)
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(X[x0 + li][4 * kk + lq], Y[y0 + li][4 * kk + lq], acc, 0, 0, 0);
            return acc;                                  // element e: row lq + 4 e, column li of the tile
        };
        // --- A2) rows of the inverse: B[j-block, 0:left] <- T11 * B[j-block, 0:left]  (their old diagonal block is the
        // identity, columns < j0 hold the updates of the earlier steps); bj[c][m] = B[j0 + m][c]
        const int nlt = (left + 15) >> 4;                // column tiles of the live part of the inverse's rows
        if (!(tune & 2)) {
            constexpr int NA = (KMAX * NB + NT - 1) / NT;
            double ov[NA];
#pragma unroll
            for (int n = 0; n < NA; ++n) {
                const int e = t + n * NT, m = e / left, c = e - m * left;
                ov[n] = (e < left * NB && m < w) ? Ti[(size_t)(j0 + m) * k + c] : 0.0;
            }
#pragma unroll
            for (int n = 0; n < NA; ++n) {
                const int e = t + n * NT, m = e / left, c = e - m * left;
                if (e < left * NB) bj[c][m] = ov[n];
            }
            __syncthreads();
            f64x4 d2[2];                                 // 2 x nlt <= 32 tiles over sixteen waves
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int id = wv + 16 * q, tm = id & 1, tn = id >> 1;
                d2[q] = tn < nlt ? tile(ti, 16 * tm, bj, 16 * tn) : f64x4{0.0, 0.0, 0.0, 0.0};
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int id = wv + 16 * q, tm = id & 1, tn = id >> 1;
                const int c = 16 * tn + li;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int m = 16 * tm + lq + 4 * e;
                    const double x = d2[q][e];
                    if (tn < nlt && c < left) {
                        bj[c][m] = x;
                        if (m < w) Ti[(size_t)(j0 + m) * k + c] = x;
                    }
                }
            }
        }
        // --- B) panel: L[i, j-block] = A[i, j-block] * T11^T   (i below the block)
        const int nbt = (below + 15) >> 4;               // row tiles below the block
        if (!(tune & 4)) {
            constexpr int NP = (KMAX * NB + NT - 1) / NT;
            double av[NP];
#pragma unroll
            for (int n = 0; n < NP; ++n) {
                const int e = t + n * NT, i = e >> 5, c = e & 31;
                av[n] = (e < below * NB && c < w) ? Lo[(size_t)(j0 + w + i) * k + j0 + c] : 0.0;
            }
#pragma unroll
            for (int n = 0; n < NP; ++n) {
                const int e = t + n * NT;
                if (e < below * NB) pan[e >> 5][e & 31] = av[n];
            }
            __syncthreads();
            f64x4 d2[2];                                 // nbt x 2 <= 32 tiles
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int id = wv + 16 * q, tn = id & 1, tm = id >> 1;
                d2[q] = tm < nbt ? tile(pan, 16 * tm, ti, 16 * tn) : f64x4{0.0, 0.0, 0.0, 0.0};
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int id = wv + 16 * q, tn = id & 1, tm = id >> 1;
                const int c = 16 * tn + li;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = 16 * tm + lq + 4 * e;
                    if (tm < nbt && i < below) {
                        const double x = c < w ? d2[q][e] : 0.0;
                        pan[i][c] = x;
                        if (c < w) Lo[(size_t)(j0 + w + i) * k + j0 + c] = x;
                    }
                }
            }
        }
        __syncthreads();
        // --- C) trailing updates with the panel, 16 x 16 tiles dealt round-robin to the sixteen waves: the rest of A
        // (tiles on and under the diagonal; inside a diagonal tile only c2 <= i is stored) and the inverse's rows below.
        if (!(tune & 8)) {
            const int ntri = nbt * (nbt + 1) / 2;
            for (int id = wv; id < ntri; id += 16) {
                int tm = 0, rem = id;                    // id -> (tm, tn <= tm)
                while (rem > tm) rem -= ++tm;
                const int tn = rem;
                const int c2 = 16 * tn + li;
                double *base = Lo + (size_t)(j0 + w) * k + j0 + w;
                f64x4 old;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = 16 * tm + lq + 4 * e;
                    old[e] = (i < below && c2 <= i) ? base[(size_t)i * k + c2] : 0.0;
                }
                const f64x4 d = tile(pan, 16 * tm, pan, 16 * tn);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = 16 * tm + lq + 4 * e;
                    if (i < below && c2 <= i) base[(size_t)i * k + c2] = old[e] - d[e];
                }
            }
        }
        if (!(tune & 16)) {
            for (int id = wv; id < nbt * nlt; id += 16) {
                const int tm = id / nlt, tn = id - tm * nlt;
                const int c = 16 * tn + li;
                double *base = Ti + (size_t)(j0 + w) * k;
                f64x4 old;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = 16 * tm + lq + 4 * e;
                    old[e] = (i < below && c < left) ? base[(size_t)i * k + c] : 0.0;
                }
                const f64x4 d = tile(pan, 16 * tm, bj, 16 * tn);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = 16 * tm + lq + 4 * e;
                    if (i < below && c < left) base[(size_t)i * k + c] = old[e] - d[e];
                }
            }
        }
        __syncthreads();
    }
    // R = L^T D (upper), X = D^-1 L^-T (upper): transpose in place, one thread per (i <= c) pair
    for (int e0 = t; e0 < k * k; e0 += (IB / 2) * NT) {
        double lv[IB / 2], xv[IB / 2];
#pragma unroll
        for (int u = 0; u < IB / 2; ++u) {
            const int e = e0 + u * NT, i = e / k, c = e - i * k;
            const bool on = e < k * k && i <= c;
            lv[u] = on ? Lo[(size_t)c * k + i] : 0.0;
            xv[u] = on ? Ti[(size_t)c * k + i] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < IB / 2; ++u) {
            const int e = e0 + u * NT, i = e / k, c = e - i * k;
            if (e < k * k && i <= c) {
                if (i != c) Lo[(size_t)c * k + i] = 0.0, Ti[(size_t)c * k + i] = 0.0;
                Lo[(size_t)i * k + c] = lv[u] * dsc[c];
                Ti[(size_t)i * k + c] = xv[u] * idsc[i];
            }
        }
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
