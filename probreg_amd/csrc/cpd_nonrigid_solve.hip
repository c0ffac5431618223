// Non-rigid CPD M-step on MI355X (gfx950): dense fp64 solve on the f64 matrix cores.
//
// Reference (neka-nat/probreg v0.3.7, probreg/cpd.py:284-303):
//     W = solve(diag(p1) G + lmd*sigma2_prev*I,  px - diag(p1) Y)        (LAPACK gesv, float64)
//     T = Y + G W ;  sigma2 = (tr(X^T diag(pt1) X) - 2 tr(px^T T) + tr(T^T diag(p1) T)) / (n_p D)
//
// The system matrix A = D G + c I (D = diag(p1), c = lmd*sigma2_prev) is not symmetric, but with
// U = D^1/2 and V = D^1/2 G the push-through identity
//     (c I + U V)^-1 = (1/c) (I - U (c I + V U)^-1 V)
// turns it into one SPD system in S = c I + D^1/2 G D^1/2 (eigenvalues >= c > 0, legal for p1 >= 0):
//     W = (B - D^1/2 S^-1 D^1/2 (G B)) / c ,   B = px - D Y.
// S is factored by a right-looking blocked Cholesky (NB = 128) whose trailing update and panel
// solve are NT-GEMMs on v_mfma_f64_16x16x4_f64; everything is fp64 because cond(S) reaches 1e7-1e8
// (c ~ 1e-3, lambda_max(G) ~ M) which rules out an fp32 factorisation at the 1e-4 parity target.
#include <math.h>

#include <algorithm>

#include "cpd_plan.h"

namespace {

constexpr int kBlock = 256;
constexpr int NB = 128;          // Cholesky block size == GEMM tile edge
constexpr int LDP = NB + 1;      // padded LDS row stride (doubles)
typedef double d4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// ---- right-hand side and S --------------------------------------------------------------------
// B = px - p1 * y   (cpd.py:296, right-hand side), rows >= m zero.  sp = sqrt(p1).
// With correspondence priors (ConstrainedNonRigidCPD, cpd.py:391-396): p1 -> p1 + sigma2/alpha * p1_tilde on the
// left-hand side and B += sigma2/alpha * (px_tilde - p1_tilde * y).
__global__ __launch_bounds__(kBlock) void k_rhs(const double* __restrict__ rowacc, int64_t mcap,
                                                const float4* __restrict__ src4, int64_t m, int64_t mp,
                                                const double* __restrict__ prior, double alpha,
                                                const double* __restrict__ params, double* __restrict__ b3,
                                                double* __restrict__ sp) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= mp) return;
    if (i < m) {
        double p1 = rowacc[i];
        const float4 y = src4[i];
        double b0 = rowacc[mcap + i] - p1 * (double)y.x;
        double b1 = rowacc[2 * mcap + i] - p1 * (double)y.y;
        double b2 = rowacc[3 * mcap + i] - p1 * (double)y.z;
        if (prior) {
            const double f = params[13] / alpha, pt = prior[i];
            b0 += f * (prior[m + i] - pt * (double)y.x);
            b1 += f * (prior[2 * m + i] - pt * (double)y.y);
            b2 += f * (prior[3 * m + i] - pt * (double)y.z);
            p1 += f * pt;
        }
        b3[i * 3 + 0] = b0;
        b3[i * 3 + 1] = b1;
        b3[i * 3 + 2] = b2;
        sp[i] = sqrt(fmax(p1, 0.0));
    } else {
        b3[i * 3] = b3[i * 3 + 1] = b3[i * 3 + 2] = 0.0;
        sp[i] = 0.0;
    }
}

// S = c I + diag(sp) G diag(sp) on the lower 128-tiles (diagonal tiles full); identity in the pad.
__global__ __launch_bounds__(kBlock) void k_build_s(const float* __restrict__ g, int64_t m, int64_t mp,
                                                    const double* __restrict__ sp, const double* __restrict__ params,
                                                    double lmd, double* __restrict__ s) {
    const int by = blockIdx.y, bx = blockIdx.x;
    if (bx > by) return;
    const double c = params ? lmd * params[13] : lmd;  // params == nullptr: lmd carries c itself (BCPD)
    const int tx = threadIdx.x & 127, ty = threadIdx.x >> 7;  // 128 columns x 2 rows per pass
    const int64_t j = (int64_t)bx * NB + tx;
    const double spj = j < m ? sp[j] : 0.0;
    for (int r = ty; r < NB; r += 2) {
        const int64_t i = (int64_t)by * NB + r;
        double v;
        if (i < m && j < m)
            v = sp[i] * (double)g[i * m + j] * spj + (i == j ? c : 0.0);
        else
            v = (i == j) ? 1.0 : 0.0;
        s[i * mp + j] = v;
    }
}

// ---- diagonal block: Cholesky + triangular inverse in LDS ---------------------------------------
// One workgroup (256 threads), a = S[k0:k0+128, k0:k0+128] in 132 KB of LDS -> L (written back, lower) and
// L^-1 (to linv, full 128 x 128 row-major with an explicit zero upper triangle).
// Both phases are blocked by 8 columns so that the sequential part is 16 steps, not 128: the 8 x 8 diagonal
// block is factored / inverted redundantly by every thread in registers (36 doubles, static indexing), each
// thread then owns one row below it, and the rank-8 trailing update is register tiled 4 x 4.
constexpr int PB = 8;

// 1 / sqrt(d) for d > 0 to ~1 ulp: v_rsq_f64 (2^-26) + two Newton-Raphson steps, scaled against under/overflow of d y^2
__device__ __forceinline__ double rsqrt_newton(double d) {
    double y = __builtin_amdgcn_rsq(d);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const double e = fma(-d * y, y, 1.0);  // 1 - d y^2
        y = fma(0.5 * y, e, y);
    }
    return y;
}

__device__ __forceinline__ void load_diag8(const double* a, int jb, double (&l)[PB][PB]) {
#pragma unroll
    for (int r = 0; r < PB; ++r)
#pragma unroll
        for (int c = 0; c < PB; ++c) l[r][c] = (c <= r) ? a[(jb + r) * LDP + jb + c] : 0.0;
}

// nreal: rows / columns of S that are not identity padding (>= k0 + 1): the block's trailing identity part is skipped.
__global__ __launch_bounds__(kBlock) void k_potrf_inv(double* __restrict__ s, int64_t ld, int64_t k0,
                                                      double* __restrict__ linv, int* __restrict__ info,
                                                      int64_t nreal) {
    // linv == nullptr: factor only (the small-system path solves with the factor itself, k_trsm_rows / k_tri_solve3)
    extern __shared__ double a[];  // [NB][LDP]
    __shared__ double rdiag[NB];   // 1 / L_jj
    const int tid = threadIdx.x;
    double* sblk = s + k0 * ld + k0;
    // active part of the block, in whole panels of 8 (the rest is identity and stays identity)
    const int nb = (int)((nreal - k0 >= NB) ? NB : ((nreal - k0 + PB - 1) / PB) * PB);
    // block -> LDS, 16 loads in flight per thread (a load / LDS-store pair per trip would pay one global round trip each)
    for (int base = 0; base < NB * NB; base += 16 * kBlock) {
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int idx = base + u * kBlock + tid, i = idx >> 7, j = idx & 127;
            v[u] = (i < nb && j < nb) ? sblk[(int64_t)i * ld + j] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int idx = base + u * kBlock + tid, i = idx >> 7, j = idx & 127;
            if (i < nb && j < nb) a[i * LDP + j] = v[u];
        }
    }
    // ---------------- Cholesky, 16 panels of 8 columns ----------------
    for (int jb = 0; jb < nb; jb += PB) {
        __syncthreads();
        double l[PB][PB];
        load_diag8(a, jb, l);
        const int i = jb + PB + tid;  // the row below the diagonal block this thread owns (if any)
        const bool has_row = i < nb;
        double x[PB];
        if (has_row) {
#pragma unroll
            for (int c = 0; c < PB; ++c) x[c] = a[i * LDP + jb + c];
        }
        __syncthreads();  // everybody holds the old diagonal block / its row in registers
        double inv[PB];
        bool bad = false;
#pragma unroll
        for (int c = 0; c < PB; ++c) {  // Cholesky-Crout on the 8 x 8 block
            double d = l[c][c];
#pragma unroll
            for (int k = 0; k < c; ++k) d -= l[c][k] * l[c][k];
            if (!(d > 0.0)) { bad = true; d = fabs(d) > 0.0 ? fabs(d) : 1.0; }
            // this chain is the sequential part of the whole factorisation: 1 / sqrt(d) from the hardware estimate and
            // two Newton steps (a dozen dependent operations) instead of a correctly rounded sqrt followed by a divide
            inv[c] = rsqrt_newton(d);
            l[c][c] = d * inv[c];
#pragma unroll
            for (int r = c + 1; r < PB; ++r) {
                double t = l[r][c];
#pragma unroll
                for (int k = 0; k < c; ++k) t -= l[r][k] * l[c][k];
                l[r][c] = t * inv[c];
            }
        }
        if (bad && tid == 0) atomicMax(info, (int)(k0 + jb + 1));
        if (has_row) {  // x := x * L11^-T  (forward substitution along the row)
#pragma unroll
            for (int c = 0; c < PB; ++c) {
                double t = x[c];
#pragma unroll
                for (int k = 0; k < c; ++k) t -= x[k] * l[c][k];
                x[c] = t * inv[c];
                a[i * LDP + jb + c] = x[c];
            }
        }
        if (tid == kBlock - 1) {
#pragma unroll
            for (int r = 0; r < PB; ++r) {
                rdiag[jb + r] = inv[r];
#pragma unroll
                for (int c = 0; c <= r; ++c) a[(jb + r) * LDP + jb + c] = l[r][c];
            }
        }
        __syncthreads();
        // rank-8 update of the trailing lower triangle, 4 x 4 register tiles
        const int t0 = jb + PB, nt = (nb - t0 + 3) >> 2;
        for (int t = tid; t < nt * nt; t += kBlock) {
            const int ti = t / nt, tk = t % nt;
            if (tk > ti) continue;
            const int i0 = t0 + 4 * ti, kk0 = t0 + 4 * tk;
            double ai[4][PB], ak[4][PB];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < PB; ++c) {
                    ai[r][c] = (i0 + r < nb) ? a[(i0 + r) * LDP + jb + c] : 0.0;
                    ak[r][c] = (kk0 + r < nb) ? a[(kk0 + r) * LDP + jb + c] : 0.0;
                }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (i0 + r < nb && kk0 + q <= i0 + r) {
                        double acc = 0.0;
#pragma unroll
                        for (int c = 0; c < PB; ++c) acc += ai[r][c] * ak[q][c];
                        a[(i0 + r) * LDP + kk0 + q] -= acc;
                    }
                }
        }
    }
    __syncthreads();
    for (int idx = tid; idx < NB * NB; idx += kBlock) {
        const int i = idx >> 7, j = idx & 127;
        if (j <= i && i < nb) sblk[(int64_t)i * ld + j] = a[i * LDP + j];
    }
    if (!linv) return;
    // ---------------- in-place inverse of the lower triangle, block columns from last to first ----------------
    // with A11 the 8 x 8 diagonal block, A21 the rows below it and X22 = inv(A22) already in place:
    //   new A21 = -X22 * A21 * inv(A11),  new A11 = inv(A11)
    for (int jb = nb - PB; jb >= 0; jb -= PB) {
        __syncthreads();
        double l[PB][PB], li[PB][PB], dinv[PB];
        load_diag8(a, jb, l);
#pragma unroll
        for (int r = 0; r < PB; ++r) dinv[r] = rdiag[jb + r];
#pragma unroll
        for (int c = 0; c < PB; ++c) {  // li = inv(l), column by column (forward substitution)
#pragma unroll
            for (int r = 0; r < PB; ++r) {
                if (r < c) { li[r][c] = 0.0; continue; }
                double t = (r == c) ? 1.0 : 0.0;
#pragma unroll
                for (int k = 0; k < PB; ++k)
                    if (k >= c && k < r) t -= l[r][k] * li[k][c];
                li[r][c] = t * dinv[r];
            }
        }
        const int i = jb + PB + tid;
        const bool has_row = i < nb;
        double z[PB];
        if (has_row) {
            double y[PB];
#pragma unroll
            for (int c = 0; c < PB; ++c) y[c] = 0.0;
            // y = X22[i, :] * A21 (X22 lower triangular); four rows per trip so that their LDS reads are in flight together
            int k = jb + PB;
            for (; k + 3 <= i; k += 4) {
                double xk[4], ar[4][PB];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    xk[u] = a[i * LDP + k + u];
#pragma unroll
                    for (int c = 0; c < PB; ++c) ar[u][c] = a[(k + u) * LDP + jb + c];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int c = 0; c < PB; ++c) y[c] = fma(xk[u], ar[u][c], y[c]);
            }
            for (; k <= i; ++k) {
                const double xik = a[i * LDP + k];
#pragma unroll
                for (int c = 0; c < PB; ++c) y[c] = fma(xik, a[k * LDP + jb + c], y[c]);
            }
#pragma unroll
            for (int c = 0; c < PB; ++c) {  // z = -y * inv(A11)
                double t = 0.0;
#pragma unroll
                for (int q = 0; q < PB; ++q)
                    if (q >= c) t += y[q] * li[q][c];
                z[c] = -t;
            }
        }
        __syncthreads();  // every read of the old A21 / A11 is done
        if (has_row) {
#pragma unroll
            for (int c = 0; c < PB; ++c) a[i * LDP + jb + c] = z[c];
        }
        if (tid == kBlock - 1) {
#pragma unroll
            for (int r = 0; r < PB; ++r)
#pragma unroll
                for (int c = 0; c <= r; ++c) a[(jb + r) * LDP + jb + c] = li[r][c];
        }
    }
    __syncthreads();
    for (int idx = tid; idx < NB * NB; idx += kBlock) {
        const int i = idx >> 7, j = idx & 127;
        linv[idx] = (i < nb) ? ((j <= i) ? a[i * LDP + j] : 0.0) : (i == j ? 1.0 : 0.0);
    }
}

// ---- NT GEMM on the f64 matrix cores ---------------------------------------------------------------
// C[i][j] (op)= sum_{k < K} A[i][k] * B[j][k] for one 128 x 128 tile per workgroup, K a multiple of 16.
//   MODE 0 (panel solve, X = A21 * Linv^T, K = 128): C = A B^T, C aliases A (all loads finish before any store).
//   MODE 1 (trailing update, A22 -= L21 L21^T): C -= A B^T on the lower tiles, 1-D triangular grid.
//   MODE 2 (update inside the outer panel): C -= A B^T on a (rows x few column tiles) grid, tiles above the
//           diagonal skipped.
// The outer panel is 512 columns wide (4 Cholesky blocks): the big MODE 1 update then runs with K = 512, i.e.
// 4x fewer read-modify-write passes over the trailing matrix than with K = 128 - at K = 128 the update is
// HBM bound (16 flop per byte of C traffic), at K = 512 it is bound by the matrix cores.
// Wave w owns the 64 x 64 quadrant (w>>1, w&1) as 4 x 4 MFMA tiles of 16 x 16 (64 accumulator f64
// per lane).  v_mfma_f64_16x16x4_f64 operand map: lane l supplies A[i = l&15][k = l>>4] and
// B[k = l>>4][j = l&15]; the k index inside a 16-chunk is permuted (lane group kq holds
// k = 4 kq + s at step s) identically for A and B, so every lane fetches 4 consecutive doubles
// (two 16-byte loads) per 16-row strip.  C/D map: col = l&15, row = (l>>4) + 4 reg.
template <int MODE>
__global__ __launch_bounds__(kBlock, 2) void k_gemm_nt_f64(const double* abase, const double* bbase,
                                                        int64_t lda, int64_t ldb, double* cbase, int64_t ldc,
                                                        int kdim) {
    int by, bx;
    if (MODE == 2 || MODE == 3) {  // MODE 3: rectangular C -= A B^T, every tile (triangular solve with many RHS)
        by = blockIdx.x;
        bx = blockIdx.y;
        if (MODE == 2 && bx > by) return;
    } else if (MODE == 1) {
        const int t = blockIdx.x;
        by = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
        while ((int64_t)(by + 1) * (by + 2) / 2 <= t) ++by;
        while ((int64_t)by * (by + 1) / 2 > t) --by;
        bx = t - (int)((int64_t)by * (by + 1) / 2);
    } else {
        by = blockIdx.x;
        bx = 0;
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wr = wv >> 1, wc = wv & 1;
    const int li = lane & 15, kq = lane >> 4;
    const double* ap = abase + ((int64_t)by * NB + wr * 64 + li) * lda + 4 * kq;
    const double* bp = bbase + ((int64_t)bx * NB + wc * 64 + li) * ldb + 4 * kq;
    d4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};
    for (int kc = 0; kc < kdim; kc += 16) {
        // operands straight from global / L2 into registers; with two workgroups per CU the loads of one wave
        // overlap the 64 MFMAs (64 cycles each) of the other wave on the same SIMD
        d4 af[4], bf[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            af[t] = *reinterpret_cast<const d4*>(ap + (int64_t)t * 16 * lda + kc);
            bf[t] = *reinterpret_cast<const d4*>(bp + (int64_t)t * 16 * ldb + kc);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
    }
    if (MODE == 0) __syncthreads();  // C aliases A: every wave's loads are done before anyone stores
    // epilogue: the read-modify-write of the C tile is batched (16 independent loads in flight, then 16
    // stores) - a load/sub/store chain per element would serialise 64 global round trips per lane
    double* __restrict__ cp = cbase + ((int64_t)by * NB + wr * 64 + kq) * ldc + (int64_t)bx * NB + wc * 64 + li;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (MODE != 0) {
            double cv[4][4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) cv[j][r] = cp[(int64_t)(16 * i + 4 * r) * ldc + 16 * j];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) cp[(int64_t)(16 * i + 4 * r) * ldc + 16 * j] = cv[j][r] - acc[i][j][r];
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) cp[(int64_t)(16 * i + 4 * r) * ldc + 16 * j] = acc[i][j][r];
        }
    }
}

// ---- small systems: solves with the factor itself (no block inverses) ------------------------------------------------
// The reduced system of the low-rank M-step has a few hundred rows: the triangular inverse that feeds the MFMA panel
// solve of the big factorisation costs more than it saves there (it is 60 % of k_potrf_inv).
//
// Panel solve X L11^T = A21 for the rows below a factored diagonal block, in place.  8 lanes share a row (lane s owns the
// columns j = s mod 8); right-looking: x_j is finished by its owner, broadcast inside the 8-lane group, and every lane
// retires it from its later columns.  L11 sits in LDS (k * LDP + j: conflict free across the 8 x 8 lanes of a wave).
__global__ __launch_bounds__(kBlock) void k_trsm_rows(double* __restrict__ s, int64_t ld, int64_t k0, int64_t row_begin,
                                                      int64_t row_end) {
    extern __shared__ double a[];  // [NB][LDP] lower triangle of L11
    __shared__ double rdiag[NB];
    const int tid = threadIdx.x;
    const double* lblk = s + k0 * ld + k0;
    for (int base = 0; base < NB * NB; base += 16 * kBlock) {
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int idx = base + u * kBlock + tid, i = idx >> 7, j = idx & 127;
            v[u] = j <= i ? lblk[(int64_t)i * ld + j] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int idx = base + u * kBlock + tid, i = idx >> 7, j = idx & 127;
            a[i * LDP + j] = v[u];
        }
    }
    __syncthreads();
    if (tid < NB) rdiag[tid] = 1.0 / a[tid * LDP + tid];
    __syncthreads();
    const int sub = tid & 7;
    const int64_t row = row_begin + (int64_t)blockIdx.x * (kBlock / 8) + (tid >> 3);
    const bool live = row < row_end;
    double* __restrict__ arow = s + (live ? row : row_begin) * ld + k0;
    double x[16];  // columns sub + 8 q
#pragma unroll
    for (int q = 0; q < 16; ++q) x[q] = live ? arow[sub + 8 * q] : 0.0;
    const int lane = tid & 63;
#pragma unroll
    for (int jq = 0; jq < 16; ++jq) {
#pragma unroll
        for (int js = 0; js < 8; ++js) {
            const int j = 8 * jq + js;
            const double mine = x[jq] * rdiag[j];  // only lane js of the group holds column j
            const double xj = __shfl(mine, (lane & ~7) | js, 64);
            if (sub == js) x[jq] = xj;
#pragma unroll
            for (int q = jq; q < 16; ++q) {
                const int k = sub + 8 * q;
                if (k > j) x[q] = fma(-xj, a[k * LDP + j], x[q]);
            }
        }
    }
    if (live) {
#pragma unroll
        for (int q = 0; q < 16; ++q) arow[sub + 8 * q] = x[q];
    }
}

// L11 u = v (TRANS 0) or L11^T u = v (TRANS 1) for one diagonal block and 3 right-hand sides, in place: one wave, two rows
// per lane, column oriented - u_j is finished by its owner, broadcast, and retired from the rows that still wait.
template <int TRANS>
__global__ __launch_bounds__(kBlock) void k_tri_solve3(const double* __restrict__ s, int64_t ld, int64_t k0,
                                                       double* __restrict__ v) {
    extern __shared__ double a[];  // [NB][LDP]
    const int tid = threadIdx.x;
    const double* lblk = s + k0 * ld + k0;
    for (int base = 0; base < NB * NB; base += 16 * kBlock) {
        double t[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int idx = base + u * kBlock + tid, i = idx >> 7, j = idx & 127;
            t[u] = j <= i ? lblk[(int64_t)i * ld + j] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int idx = base + u * kBlock + tid, i = idx >> 7, j = idx & 127;
            a[i * LDP + j] = t[u];
        }
    }
    __syncthreads();
    if (tid >= 64) return;
    const int lane = tid;
    double r[2][3], dinv[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int k = lane + 64 * h;
        dinv[h] = 1.0 / a[k * LDP + k];
#pragma unroll
        for (int c = 0; c < 3; ++c) r[h][c] = v[(k0 + k) * 3 + c];
    }
#pragma unroll
    for (int h0 = 0; h0 < 2; ++h0) {
        const int hj = TRANS ? 1 - h0 : h0;  // the half the pivots of this sweep live in
        for (int step = 0; step < 64; ++step) {
            const int jl = TRANS ? 63 - step : step, j = jl + 64 * hj;
            double u[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) u[c] = __shfl(r[hj][c] * dinv[hj], jl, 64);
            if (lane == jl) {
#pragma unroll
                for (int c = 0; c < 3; ++c) r[hj][c] = u[c];
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int k = lane + 64 * h;
                const bool pending = TRANS ? k < j : k > j;
                const double l = pending ? (TRANS ? a[j * LDP + k] : a[k * LDP + j]) : 0.0;
#pragma unroll
                for (int c = 0; c < 3; ++c) r[h][c] = fma(-l, u[c], r[h][c]);
            }
        }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int c = 0; c < 3; ++c) v[(k0 + lane + 64 * h) * 3 + c] = r[h][c];
}

// ---- block triangular solves with 3 right-hand sides ----------------------------------------------
// x_k = Linv_k * v_k (TRANS = 0) or Linv_k^T * v_k (TRANS = 1); v, x are [mp][3] row-major.
template <int TRANS>
__global__ __launch_bounds__(384) void k_diag_solve(const double* __restrict__ linv, double* __restrict__ v,
                                                    int64_t k0) {
    __shared__ double vin[NB][3];
    const int r = threadIdx.x / 3, c = threadIdx.x % 3;
    vin[r][c] = v[(k0 + r) * 3 + c];
    __syncthreads();
    double s = 0.0;
    // the upper triangle of linv is stored as explicit zeros, so both loops can run over all 128 terms
    // with independent loads (unrolled, pipelined) instead of a data-dependent trip count
    if (TRANS == 0) {
#pragma unroll 16
        for (int k = 0; k < NB; ++k) s += linv[r * NB + k] * vin[k][c];
    } else {
#pragma unroll 16
        for (int k = 0; k < NB; ++k) s += linv[k * NB + r] * vin[k][c];
    }
    v[(k0 + r) * 3 + c] = s;
}

// forward (right-looking): v[i] -= sum_c L[i][k0 + c] * x_k[c] for rows i >= k0 + 128.
// 4 threads per row, each covering 32 of the 128 columns in 16-byte pieces; x_k in LDS.
__global__ __launch_bounds__(kBlock) void k_fwd_update(const double* __restrict__ s, int64_t ld, int64_t k0,
                                                       int64_t mp, double* __restrict__ v) {
    __shared__ double xk[NB][3];
    for (int idx = threadIdx.x; idx < NB * 3; idx += kBlock) xk[idx / 3][idx % 3] = v[k0 * 3 + idx];
    __syncthreads();
    const int part = threadIdx.x & 3;
    const int64_t i = k0 + NB + (int64_t)blockIdx.x * 64 + (threadIdx.x >> 2);
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    if (i < mp) {
        const double* row = s + i * ld + k0;
#pragma unroll 4
        for (int q = 0; q < 16; ++q) {
            const int c = q * 8 + part * 2;
            const double2 l2 = *reinterpret_cast<const double2*>(row + c);
            a0 += l2.x * xk[c][0] + l2.y * xk[c + 1][0];
            a1 += l2.x * xk[c][1] + l2.y * xk[c + 1][1];
            a2 += l2.x * xk[c][2] + l2.y * xk[c + 1][2];
        }
    }
    a0 += __shfl_xor(a0, 1, 64); a0 += __shfl_xor(a0, 2, 64);
    a1 += __shfl_xor(a1, 1, 64); a1 += __shfl_xor(a1, 2, 64);
    a2 += __shfl_xor(a2, 1, 64); a2 += __shfl_xor(a2, 2, 64);
    if (i < mp && part == 0) {
        v[i * 3] -= a0;
        v[i * 3 + 1] -= a1;
        v[i * 3 + 2] -= a2;
    }
}

// backward (right-looking): v[c] -= sum_r L[k0 + r][c] * x_k[r] for columns c < k0 (coalesced along c).
__global__ __launch_bounds__(kBlock) void k_bwd_update(const double* __restrict__ s, int64_t ld, int64_t k0,
                                                       double* __restrict__ v) {
    __shared__ double xk[NB][3];
    for (int idx = threadIdx.x; idx < NB * 3; idx += kBlock) xk[idx / 3][idx % 3] = v[k0 * 3 + idx];
    __syncthreads();
    const int64_t c = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (c >= k0) return;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    const double* col = s + k0 * ld + c;
#pragma unroll 8
    for (int r = 0; r < NB; ++r) {
        const double l = col[(int64_t)r * ld];
        a0 += l * xk[r][0];
        a1 += l * xk[r][1];
        a2 += l * xk[r][2];
    }
    v[c * 3] -= a0;
    v[c * 3 + 1] -= a1;
    v[c * 3 + 2] -= a2;
}

// v = sp .* gb     |   w = (b - sp .* u) / c
__global__ __launch_bounds__(kBlock) void k_scale_rows(const double* __restrict__ sp, const double* __restrict__ in,
                                                       int64_t mp, double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= mp) return;
    const double f = sp[i];
    out[i * 3] = f * in[i * 3];
    out[i * 3 + 1] = f * in[i * 3 + 1];
    out[i * 3 + 2] = f * in[i * 3 + 2];
}
__global__ __launch_bounds__(kBlock) void k_form_w(const double* __restrict__ b3, const double* __restrict__ sp,
                                                   const double* __restrict__ u, int64_t m,
                                                   const double* __restrict__ params, double lmd,
                                                   double* __restrict__ w) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= m) return;
    const double rc = 1.0 / (lmd * params[13]);
    const double f = sp[i];
    w[i * 3] = (b3[i * 3] - f * u[i * 3]) * rc;
    w[i * 3 + 1] = (b3[i * 3 + 1] - f * u[i * 3 + 1]) * rc;
    w[i * 3 + 2] = (b3[i * 3 + 2] - f * u[i * 3 + 2]) * rc;
}

// iterative refinement: r = b - (sp^2 .* (G w) + c w)   (residual of (D G + c I) w = b, fp64)
__global__ __launch_bounds__(kBlock) void k_residual(const double* __restrict__ b3, const double* __restrict__ sp,
                                                     const double* __restrict__ gw, const double* __restrict__ w,
                                                     int64_t m, int64_t mp, const double* __restrict__ params,
                                                     double lmd, double* __restrict__ r3) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= mp) return;
    if (i < m) {
        const double c = lmd * params[13], d = sp[i] * sp[i];
        for (int k = 0; k < 3; ++k) r3[i * 3 + k] = b3[i * 3 + k] - (d * gw[i * 3 + k] + c * w[i * 3 + k]);
    } else {
        r3[i * 3] = r3[i * 3 + 1] = r3[i * 3 + 2] = 0.0;
    }
}
__global__ __launch_bounds__(kBlock) void k_axpy3(const double* __restrict__ x, int64_t m, double* __restrict__ y) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < m * 3) y[i] += x[i];
}

// partial sums of tr(px^T T) and tr(T^T diag(p1) T), T = y + gw   (cpd.py:298-300)
__global__ __launch_bounds__(kBlock) void k_traces(const double* __restrict__ rowacc, int64_t mcap,
                                                   const float4* __restrict__ src4, const double* __restrict__ gw,
                                                   int64_t m, double* __restrict__ part) {
    __shared__ double sh[4][2];
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    double a = 0.0, b = 0.0;
    if (i < m) {
        const float4 y = src4[i];
        const double t0 = (double)y.x + gw[i * 3], t1 = (double)y.y + gw[i * 3 + 1], t2 = (double)y.z + gw[i * 3 + 2];
        a = rowacc[mcap + i] * t0 + rowacc[2 * mcap + i] * t1 + rowacc[3 * mcap + i] * t2;
        b = rowacc[i] * (t0 * t0 + t1 * t1 + t2 * t2);
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    a = wave_sum(a);
    b = wave_sum(b);
    if (lane == 0) { sh[wv][0] = a; sh[wv][1] = b; }
    __syncthreads();
    if (threadIdx.x < 2)
        part[(int64_t)blockIdx.x * 2 + threadIdx.x] = sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] +
                                                     sh[3][threadIdx.x];
}

__global__ void k_nonrigid_finish(const double* __restrict__ part, int nblk, const double* __restrict__ mom,
                                  double* __restrict__ params, int dim) {
    __shared__ double sh[2][64];
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 64) {
        a += part[2 * i];
        b += part[2 * i + 1];
    }
    sh[0][threadIdx.x] = a;
    sh[1][threadIdx.x] = b;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tr_pxt = 0.0, tr_tpt = 0.0;
        for (int i = 0; i < 64; ++i) { tr_pxt += sh[0][i]; tr_tpt += sh[1][i]; }
        const double n_p = mom[0], tr_xp1x = mom[22];
        const double sigma2 = (tr_xp1x - 2.0 * tr_pxt + tr_tpt) / (n_p * dim);  // cpd.py:301 (no eps clamp)
        params[13] = sigma2;
        params[14] = sigma2;  // q := sigma2, cpd.py:303
        params[15] = n_p;
        params[16] += 1.0;
    }
}

// ---- low-rank M-step (G = F F^T, cpd_nonrigid.hip) ---------------------------------------------------------------
// (D G + c I) W = B with G = F F^T:   W = (B - D F Z) / c,   (c I + F^T D F) Z = F^T B     (push-through identity again,
// now with an r x r SPD system, r = rank of the factor: M r^2 flop instead of M^3 / 3 and no M x M matrix at all).
// T = F^T D F on 64 x 64 tiles of the lower triangle; the sum over the points is split over workgroups, each of which
// writes its partial tile; k_lr_gram_reduce adds them up in a fixed order (reproducible) and puts c on the diagonal.
constexpr int GT = 64;    // tile edge
constexpr int GS = 32;    // points per LDS slab
constexpr int GLD = GT + 1;

// The tile products run on the f64 matrix cores (v_mfma_f64_16x16x4_f64: wave w owns the 32 x 32 quadrant (w >> 1, w & 1)
// as 2 x 2 MFMA tiles, operands from the LDS slab).  Workgroups of the first tile column (tb == 0) also form their rows of
// u = F^T B with one more MFMA per k-step (B = [b | 0] as a 16-column operand), so the right-hand side of the reduced
// system costs no pass of its own.
__global__ __launch_bounds__(kBlock) void k_lr_gram(const double* __restrict__ f, int64_t ld, int64_t m, int rank,
                                                    const double* __restrict__ sp, const double* __restrict__ b3,
                                                    int64_t chunk, double* __restrict__ part, double* __restrict__ upart) {
    __shared__ double as[GS][GLD], bs[GS][GLD], xs[GS][16];
    // tile (ta >= tb) of the lower triangle from the linear index
    int ta = 0, rem = blockIdx.x;
    while (rem > ta) {
        rem -= ta + 1;
        ++ta;
    }
    const int tb = rem;
    const int a0 = ta * GT, b0 = tb * GT;
    const int64_t i_begin = (int64_t)blockIdx.y * chunk, i_end = i_begin + chunk < m ? i_begin + chunk : m;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wr = wv >> 1, wc = wv & 1, li = lane & 15, kq = lane >> 4;
    const int lk = threadIdx.x & 31, lr = threadIdx.x >> 5;  // staging: point lk of the slab, factor rows lr + 8 q
    const bool with_u = tb == 0;
    d4 acc[2][2], uacc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        uacc[i] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};
    }
    (&xs[0][0])[threadIdx.x] = 0.0;  // columns 3..15 of the right-hand-side operand stay zero
    (&xs[0][0])[threadIdx.x + kBlock] = 0.0;
    // software pipeline: the global loads of slab t + 1 are in flight while the matrix cores work on slab t (all loads of
    // a slab are issued before the first LDS store - interleaved, every store would wait for its own load's round trip)
    double ra[GT / 8], rb[GT / 8], rx = 0.0;
    auto fetch = [&](int64_t i0) {
        const int64_t i = i0 + lk;
        const bool in = i < i_end;
        const double pw = in ? sp[i] * sp[i] : 0.0;  // D_ii (k_rhs left sqrt(D) in sp)
#pragma unroll
        for (int q = 0; q < GT / 8; ++q) {
            const int r = lr + 8 * q;
            ra[q] = (in && a0 + r < rank) ? f[(int64_t)(a0 + r) * ld + i] : 0.0;
            rb[q] = (in && b0 + r < rank) ? f[(int64_t)(b0 + r) * ld + i] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < GT / 8; ++q) rb[q] *= pw;
        if (with_u && lr < 3) rx = in ? b3[i * 3 + lr] : 0.0;
    };
    fetch(i_begin);
    for (int64_t i0 = i_begin; i0 < i_end; i0 += GS) {
        __syncthreads();  // the previous slab has been consumed
#pragma unroll
        for (int q = 0; q < GT / 8; ++q) {
            as[lk][lr + 8 * q] = ra[q];
            bs[lk][lr + 8 * q] = rb[q];
        }
        if (with_u && lr < 3) xs[lk][lr] = rx;
        __syncthreads();
        if (i0 + GS < i_end) fetch(i0 + GS);
#pragma unroll
        for (int k4 = 0; k4 < GS; k4 += 4) {
            double av[2], bv[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                av[t] = as[k4 + kq][wr * 32 + 16 * t + li];
                bv[t] = bs[k4 + kq][wc * 32 + 16 * t + li];
            }
#pragma unroll
            for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
                for (int j2 = 0; j2 < 2; ++j2)
                    acc[i2][j2] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[i2], bv[j2], acc[i2][j2], 0, 0, 0);
            if (with_u && wc == 0) {
                const double xv = xs[k4 + kq][li];
                uacc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[0], xv, uacc[0], 0, 0, 0);
                uacc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[1], xv, uacc[1], 0, 0, 0);
            }
        }
    }
    // C/D map of the MFMA: row = (lane >> 4) + 4 reg, column = lane & 15
    double* __restrict__ out = part + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * (GT * GT);
#pragma unroll
    for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
        for (int j2 = 0; j2 < 2; ++j2)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                out[(wr * 32 + 16 * i2 + kq + 4 * r) * GT + wc * 32 + 16 * j2 + li] = acc[i2][j2][r];
    if (with_u && wc == 0 && li < 3) {  // upart[split][row][3]
        double* __restrict__ uo = upart + (int64_t)blockIdx.y * ((int64_t)rank * 3);
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = a0 + wr * 32 + 16 * i2 + kq + 4 * r;
                if (row < rank) uo[(int64_t)row * 3 + li] = uacc[i2][r];
            }
    }
}

// S[a][b] = c [a == b] + sum over the splits of T's tile entry (a, b) for a, b < rank; identity in the pad (rp x rp).
// Four lanes share an output element: each sums every fourth split, the four partial sums are combined in a fixed order
// (reproducible; the loads of the four chains run in parallel - one thread walking all splits is a chain of dependent adds
// behind ~80 loads).
__global__ __launch_bounds__(kBlock) void k_lr_gram_reduce(const double* __restrict__ part, const double* __restrict__ upart,
                                                           int ntile, int nsplit, int rank, int64_t rp,
                                                           const double* __restrict__ params, double lmd,
                                                           double* __restrict__ s, double* __restrict__ u) {
    const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t e = t >> 2;
    const int sub = (int)(t & 3);
    double v = 0.0, uv = 0.0;
    const bool in_u = e < (int64_t)rank * 3;
    const int a = (int)(e / rp), b = (int)(e % rp);
    const bool in_s = e < rp * rp, real = in_s && a < rank && b < rank;
    if (in_u) {  // u = F^T B
        const double* __restrict__ src = upart + e;
        const int64_t st = (int64_t)rank * 3;
        for (int q = sub; q < nsplit; q += 4) uv += src[q * st];
    }
    if (real) {
        const int hi = a > b ? a : b, lo = a > b ? b : a;  // lower triangle holds (hi, lo)
        const int ta = hi / GT, tb = lo / GT;
        const int tile = ta * (ta + 1) / 2 + tb;
        const int r = hi - ta * GT, c = lo - tb * GT;
        const double* __restrict__ src = part + (int64_t)tile * (GT * GT) + r * GT + c;
        const int64_t st = (int64_t)ntile * (GT * GT);
        for (int q = sub; q < nsplit; q += 4) v += src[q * st];
    }
    // (lanes 4k .. 4k+3 hold the four chains of one element)
    v = (v + __shfl_xor(v, 1, 64));
    v = (v + __shfl_xor(v, 2, 64));
    uv = (uv + __shfl_xor(uv, 1, 64));
    uv = (uv + __shfl_xor(uv, 2, 64));
    if (sub != 0) return;
    if (e < rp * 3) u[e] = in_u ? uv : 0.0;  // zero rows in the pad
    if (!in_s) return;
    if (real) {
        if (a == b) v += lmd * params[13];
    } else {
        v = a == b ? 1.0 : 0.0;
    }
    s[e] = v;
}

// w = (b - D F z) / c
__global__ __launch_bounds__(kBlock) void k_lr_form_w(const double* __restrict__ b3, const double* __restrict__ sp,
                                                      const double* __restrict__ fz, int64_t m,
                                                      const double* __restrict__ params, double lmd,
                                                      double* __restrict__ w) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= m) return;
    const double rc = 1.0 / (lmd * params[13]);
    const double d = sp[i] * sp[i];
    w[i * 3] = (b3[i * 3] - d * fz[i * 3]) * rc;
    w[i * 3 + 1] = (b3[i * 3 + 1] - d * fz[i * 3 + 1]) * rc;
    w[i * 3 + 2] = (b3[i * 3 + 2] - d * fz[i * 3 + 2]) * rc;
}

inline dim3 grid1(int64_t n) { return dim3((unsigned)prg::ceil_div(n, kBlock)); }

// blocked Cholesky S = L L^T (lower, in place): outer panels of 512 columns, inner blocks of 128; the inverses
// of the diagonal blocks go to linv.  Enqueued on the plan stream (+ the side stream of the look-ahead); on
// return the plan stream waits for everything.
int cholesky_lookahead(prg_cpd* h, double* S, int64_t mp, double* linv, int* info, int64_t nreal) {
    hipStream_t st = h->stream;
    // Look-ahead over two streams: the trailing update of outer panel J is split into U1 (the 512 columns of
    // the next panel, plan stream) and U2 (everything to the right, side stream), so the latency-bound panel
    // factorisation J+1 (potrf + panel solves) runs underneath U2(J) instead of leaving the GPU idle.
    const size_t lds = (size_t)NB * LDP * sizeof(double);
    constexpr int64_t NBO = 512;
    const int64_t nouter = prg::ceil_div(mp, NBO);
    if (!h->nr_stream2) PRG_HIP(hipStreamCreateWithFlags(&h->nr_stream2, hipStreamNonBlocking));
    while ((int64_t)h->nr_events.size() < 2 * nouter + 1) {
        hipEvent_t e;
        PRG_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        h->nr_events.push_back(e);
    }
    hipStream_t sb = h->nr_stream2;
    PRG_HIP(hipEventRecord(h->nr_events[2 * nouter], st));
    PRG_HIP(hipStreamWaitEvent(sb, h->nr_events[2 * nouter], 0));
    int64_t last_u2 = -1;
    for (int64_t J = 0; J < nouter; ++J) {
        const int64_t K0 = J * NBO;
        const int64_t kend = std::min<int64_t>(K0 + NBO, mp);
        for (int64_t k0 = K0; k0 < kend; k0 += NB) {
            double* lk = linv + (size_t)(k0 / NB) * NB * NB;
            k_potrf_inv<<<1, kBlock, lds, st>>>(S, mp, k0, lk, info, nreal);
            const int64_t below = (mp - k0 - NB) / NB;
            if (below > 0) {
                double* a21 = S + (k0 + NB) * mp + k0;
                k_gemm_nt_f64<0><<<(unsigned)below, kBlock, 0, st>>>(a21, lk, mp, NB, a21, mp, NB);
                const int64_t inner_cols = (kend - k0 - NB) / NB;  // remaining block columns of the outer panel
                if (inner_cols > 0)
                    k_gemm_nt_f64<2><<<dim3((unsigned)below, (unsigned)inner_cols), kBlock, 0, st>>>(
                        a21, a21, mp, mp, S + (k0 + NB) * mp + (k0 + NB), mp, NB);
            }
        }
        const int64_t rem = (mp - kend) / NB;
        if (rem <= 0) break;
        const int kd = (int)(kend - K0);
        double* apan = S + kend * mp + K0;  // rows >= kend of the outer panel: L[kend:, K0:kend]
        PRG_HIP(hipEventRecord(h->nr_events[2 * J], st));
        // U2(J): the sub-triangle right of the next panel, on the side stream
        const int64_t r2 = rem - NBO / NB;
        if (r2 > 0) {
            PRG_HIP(hipStreamWaitEvent(sb, h->nr_events[2 * J], 0));
            double* ap2 = apan + (NBO / NB) * NB * mp;
            k_gemm_nt_f64<1><<<(unsigned)(r2 * (r2 + 1) / 2), kBlock, 0, sb>>>(
                ap2, ap2, mp, mp, S + (kend + NBO) * mp + (kend + NBO), mp, kd);
            PRG_HIP(hipEventRecord(h->nr_events[2 * J + 1], sb));
        }
        // U1(J): the next panel's columns; it must see U2(J-1), which updated the same tiles
        if (last_u2 >= 0) PRG_HIP(hipStreamWaitEvent(st, h->nr_events[2 * last_u2 + 1], 0));
        k_gemm_nt_f64<2><<<dim3((unsigned)rem, (unsigned)std::min<int64_t>(NBO / NB, rem)), kBlock, 0, st>>>(
            apan, apan, mp, mp, S + kend * mp + kend, mp, kd);
        last_u2 = (r2 > 0) ? J : -1;
    }
    if (last_u2 >= 0) PRG_HIP(hipStreamWaitEvent(st, h->nr_events[2 * last_u2 + 1], 0));
    return PRG_OK;
}

// L L^T u = v in place for 3 right-hand sides (v [mp][3]) with the factor cholesky_lookahead left in S / linv
int cholesky_solve3(prg_cpd* h, const double* S, int64_t mp, const double* linv, double* v) {
    hipStream_t st = h->stream;
    const int64_t nblk = mp / NB;
    for (int64_t kb = 0; kb < nblk; ++kb) {  // L u' = v
        const int64_t k0 = kb * NB;
        k_diag_solve<0><<<1, 384, 0, st>>>(linv + (size_t)kb * NB * NB, v, k0);
        const int64_t rows = mp - k0 - NB;
        if (rows > 0) k_fwd_update<<<(unsigned)prg::ceil_div(rows, 64), kBlock, 0, st>>>(S, mp, k0, mp, v);
    }
    for (int64_t kb = nblk - 1; kb >= 0; --kb) {  // L^T u = u'
        const int64_t k0 = kb * NB;
        k_diag_solve<1><<<1, 384, 0, st>>>(linv + (size_t)kb * NB * NB, v, k0);
        if (k0 > 0) k_bwd_update<<<grid1(k0), kBlock, 0, st>>>(S, mp, k0, v);
    }
    PRG_HIP(hipGetLastError());
    return PRG_OK;
}

// Small SPD systems (the r x r system of the low-rank M-step): plain right-looking blocked Cholesky on one stream, the
// factor is used as it is.  nreal: rows that are not identity padding.
int cholesky_small(prg_cpd* h, double* S, int64_t mp, int* info, int64_t nreal) {
    hipStream_t st = h->stream;
    const size_t lds = (size_t)NB * LDP * sizeof(double);
    for (int64_t k0 = 0; k0 < mp && k0 < nreal; k0 += NB) {
        k_potrf_inv<<<1, kBlock, lds, st>>>(S, mp, k0, nullptr, info, nreal);
        const int64_t row_end = std::min<int64_t>(mp, prg::round_up(nreal, NB));  // identity rows need nothing
        const int64_t rows = row_end - (k0 + NB);
        if (rows <= 0) break;
        k_trsm_rows<<<(unsigned)prg::ceil_div(rows, kBlock / 8), kBlock, lds, st>>>(S, mp, k0, k0 + NB, row_end);
        const int64_t below = rows / NB;
        double* a21 = S + (k0 + NB) * mp + k0;
        k_gemm_nt_f64<2><<<dim3((unsigned)below, (unsigned)below), kBlock, 0, st>>>(a21, a21, mp, mp,
                                                                                  S + (k0 + NB) * mp + (k0 + NB), mp, NB);
    }
    PRG_HIP(hipGetLastError());
    return PRG_OK;
}

// L L^T u = v in place, 3 right-hand sides, with the factor of cholesky_small
int cholesky_small_solve3(prg_cpd* h, const double* S, int64_t mp, double* v, int64_t nreal) {
    hipStream_t st = h->stream;
    const size_t lds = (size_t)NB * LDP * sizeof(double);
    const int64_t nblk = prg::ceil_div(std::min<int64_t>(mp, nreal), NB);  // identity blocks: u = v
    for (int64_t kb = 0; kb < nblk; ++kb) {
        const int64_t k0 = kb * NB;
        k_tri_solve3<0><<<1, kBlock, lds, st>>>(S, mp, k0, v);
        const int64_t rows = nblk * NB - k0 - NB;
        if (rows > 0) k_fwd_update<<<(unsigned)prg::ceil_div(rows, 64), kBlock, 0, st>>>(S, mp, k0, nblk * NB, v);
    }
    for (int64_t kb = nblk - 1; kb >= 0; --kb) {
        const int64_t k0 = kb * NB;
        k_tri_solve3<1><<<1, kBlock, lds, st>>>(S, mp, k0, v);
        if (k0 > 0) k_bwd_update<<<grid1(k0), kBlock, 0, st>>>(S, mp, k0, v);
    }
    PRG_HIP(hipGetLastError());
    return PRG_OK;
}

int ensure_solve_workspace(prg_cpd* h, size_t need) {
    // kernels that keep a 128 x 129 fp64 block in LDS need the opt-in above 64 KB (once per process and device context)
    for (const void* fn : {reinterpret_cast<const void*>(k_potrf_inv), reinterpret_cast<const void*>(k_trsm_rows),
                           reinterpret_cast<const void*>(k_tri_solve3<0>), reinterpret_cast<const void*>(k_tri_solve3<1>)})
        PRG_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, NB * LDP * (int)sizeof(double)));
    if (h->nr_solve_bytes >= need) return PRG_OK;
    if (h->nr_solve) {
        PRG_HIP(hipStreamSynchronize(h->stream));
        (void)hipFree(h->nr_solve);
    }
    h->nr_solve = nullptr;
    h->nr_solve_bytes = 0;
    PRG_HIP(hipMalloc((void**)&h->nr_solve, need));
    PRG_HIP(hipMemsetAsync(h->nr_solve, 0, need, h->stream));  // (the sticky pivot flag of the low-rank M-step lives at its end)
    h->nr_solve_bytes = need;
    h->nr_info = nullptr;
    return PRG_OK;
}

int mstep_nonrigid_lowrank(prg_cpd* h, double lmd) {
    const int64_t m = h->M, ld = h->f_ld;
    const int rank = h->f_rank;
    const int64_t rp = prg::round_up(rank, NB), nblk = rp / NB;
    const int tiles1 = (int)prg::ceil_div(rank, GT), ntile = tiles1 * (tiles1 + 1) / 2;
    // split the sum over the points so that the chip is full (~512 workgroups of >= 8 slabs), in whole slabs
    int nsplit = (int)std::min<int64_t>(std::max(1, 512 / ntile), prg::ceil_div(m, 8 * GS));
    const int64_t chunk = prg::round_up(prg::ceil_div(m, nsplit), GS);
    nsplit = (int)prg::ceil_div(m, chunk);
    const size_t n_s = (size_t)rp * rp, n_linv = (size_t)nblk * NB * NB, n_vec = (size_t)ld * 3,
                 n_part = (size_t)nsplit * ntile * GT * GT + (size_t)nsplit * rank * 3;
    const int tr_blk = (int)prg::ceil_div(m, kBlock);
    PRG_TRY(ensure_solve_workspace(h, (n_s + n_linv + n_part + 6 * n_vec + (size_t)rp * 3 + 2 * (size_t)tr_blk + 16) * sizeof(double)));
    double* S = h->nr_solve;
    double* linv = S + n_s;
    double* part = linv + n_linv;
    double* upart = part + (size_t)nsplit * ntile * GT * GT;
    double* b3 = part + n_part;
    double* sp = b3 + n_vec;
    double* fz = sp + n_vec;
    double* gb = fz + n_vec;
    double* r3 = gb + n_vec;
    double* dw = r3 + n_vec;
    double* z = dw + n_vec;  // [rp][3]
    double* trpart = z + (size_t)rp * 3;
    int* info = reinterpret_cast<int*>(trpart + 2 * (size_t)tr_blk);
    hipStream_t st = h->stream;
    const dim3 ggrid((unsigned)ntile, (unsigned)nsplit);
    const dim3 rgrid = grid1(4 * std::max<int64_t>((int64_t)rp * rp, rp * 3));  // four lanes per element

    // (info is sticky: cleared when the workspace is allocated and when prg_cpd_get_params has reported it)
    k_rhs<<<grid1(ld), kBlock, 0, st>>>(h->rowacc, h->Mcap, h->src4, m, ld, h->nr_alpha > 0.0 ? h->nr_prior : nullptr,
                                        h->nr_alpha, h->params, b3, sp);
    // S = c I + F^T D F and z = F^T B in one pass over the factor
    k_lr_gram<<<ggrid, kBlock, 0, st>>>(h->F, ld, m, rank, sp, b3, chunk, part, upart);
    k_lr_gram_reduce<<<rgrid, kBlock, 0, st>>>(part, upart, ntile, nsplit, rank, rp, h->params, lmd, S, z);
    const bool small = rp <= 1024;  // (beyond, the MFMA panel solves of the big factorisation pay for their inverses)
    // (one workgroup doing the whole r x r solve out of LDS was built in round 3: with the packed triangle's irregular LDS
    // addressing it took 270 us at r = 176, more than the ten launches below - 240 us - and was removed)
    if (small) {
        PRG_TRY(cholesky_small(h, S, rp, info, rank));
        PRG_TRY(cholesky_small_solve3(h, S, rp, z, rank));  // z = (c I + F^T D F)^-1 F^T B
    } else {
        PRG_TRY(cholesky_lookahead(h, S, rp, linv, info, rank));
        PRG_TRY(cholesky_solve3(h, S, rp, linv, z));
    }
    double* gw = h->nr_work;                      // [M][3]: G W, kept for the next E-step's transform
    // W = (B - D F z) / c ... and G W = F (F^T W) = F z exactly: F^T W = (F^T B - F^T D F z) / c = (S z - (S - c I) z) / c
    PRG_TRY(prg::lowrank_apply(h, z, gw));
    k_lr_form_w<<<grid1(m), kBlock, 0, st>>>(b3, sp, gw, m, h->params, lmd, h->W);
    // With correspondence priors the diagonal scaling spans sigma2/alpha ~ 1e7 and the push-through form loses digits
    // to cancellation: two steps of fp64 iterative refinement on the original system (see prg_cpd_mstep_nonrigid)
    const int nrefine = h->nr_alpha > 0.0 ? 2 : 0;
    for (int it = 0; it < nrefine; ++it) {
        PRG_TRY(prg::nonrigid_gw(h, h->W, gb));
        k_residual<<<grid1(ld), kBlock, 0, st>>>(b3, sp, gb, h->W, m, ld, h->params, lmd, r3);
        if (rp > rank) PRG_HIP(hipMemsetAsync(z + (size_t)rank * 3, 0, (size_t)(rp - rank) * 3 * sizeof(double), st));
        PRG_TRY(prg::lowrank_ft3(h, r3, z));
        if (small)
            PRG_TRY(cholesky_small_solve3(h, S, rp, z, rank));
        else
            PRG_TRY(cholesky_solve3(h, S, rp, linv, z));
        PRG_TRY(prg::lowrank_apply(h, z, fz));
        k_lr_form_w<<<grid1(m), kBlock, 0, st>>>(r3, sp, fz, m, h->params, lmd, dw);
        k_axpy3<<<grid1(m * 3), kBlock, 0, st>>>(dw, m, h->W);
    }
    if (nrefine > 0) PRG_TRY(prg::nonrigid_gw(h, h->W, gw));
    k_traces<<<tr_blk, kBlock, 0, st>>>(h->rowacc, h->Mcap, h->src4, gw, m, trpart);
    k_nonrigid_finish<<<1, 64, 0, st>>>(trpart, tr_blk, h->moments, h->params, h->D);
    PRG_HIP(hipGetLastError());
    h->gw_valid = true;
    // The pivot check does not stall the stream: the flag is looked at the next time the host synchronises anyway
    // (prg_cpd_get_params / the next M-step), where a failure of this step is reported.
    h->nr_info = info;
    return PRG_OK;
}

}  // namespace

extern "C" int prg_cpd_mstep_nonrigid(prg_cpd* h, double lmd) {
    PRG_REQUIRE(h && (h->G || h->F) && h->W && h->have_estep, PRG_ERR_STATE,
                "prg_cpd_mstep_nonrigid: needs build_g and an E-step first");
    PRG_REQUIRE(lmd > 0.0, PRG_ERR_INVALID, "prg_cpd_mstep_nonrigid: lmd must be > 0 (got %g)", lmd);
    prg::DeviceGuard g(h->device);
    if (h->F) return mstep_nonrigid_lowrank(h, lmd);
    const int64_t m = h->M, mp = prg::round_up(m, NB), nblk = mp / NB;
    // workspace: S [mp*mp] | Linv [nblk*128*128] | b3, gb, v, sp (each <= 3 mp) | trace partials | info
    const size_t n_s = (size_t)mp * mp, n_linv = (size_t)nblk * NB * NB, n_vec = (size_t)mp * 3;
    const int tr_blk = (int)prg::ceil_div(m, kBlock);
    const size_t need = (n_s + n_linv + 6 * n_vec + 2 * (size_t)tr_blk + 16) * sizeof(double);
    if (h->nr_solve_bytes < need) {
        if (h->nr_solve) (void)hipFree(h->nr_solve);
        h->nr_solve = nullptr;
        h->nr_solve_bytes = 0;
        PRG_HIP(hipMalloc((void**)&h->nr_solve, need));
        h->nr_solve_bytes = need;
        PRG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_potrf_inv),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, NB * LDP * (int)sizeof(double)));
    }
    double* S = h->nr_solve;
    double* linv = S + n_s;
    double* b3 = linv + n_linv;
    double* gb = b3 + n_vec;
    double* v = gb + n_vec;
    double* sp = v + n_vec;
    double* r3 = sp + n_vec;
    double* dw = r3 + n_vec;
    double* trpart = dw + n_vec;
    int* info = reinterpret_cast<int*>(trpart + 2 * (size_t)tr_blk);
    hipStream_t st = h->stream;

    PRG_HIP(hipMemsetAsync(info, 0, sizeof(int), st));
    k_rhs<<<grid1(mp), kBlock, 0, st>>>(h->rowacc, h->Mcap, h->src4, m, mp, h->nr_alpha > 0.0 ? h->nr_prior : nullptr,
                                        h->nr_alpha, h->params, b3, sp);
    k_build_s<<<dim3((unsigned)nblk, (unsigned)nblk), kBlock, 0, st>>>(h->G, m, mp, sp, h->params, lmd, S);

    PRG_TRY(cholesky_lookahead(h, S, mp, linv, info, m));
    // w = (rhs - D^1/2 S^-1 D^1/2 (G rhs)) / c with the factor above: two blocked triangular sweeps, 3 RHS
    auto solve_with_factor = [&](const double* rhs, double* wout) -> int {
        PRG_TRY(prg::nonrigid_gw(h, rhs, gb));                    // G rhs
        k_scale_rows<<<grid1(m), kBlock, 0, st>>>(sp, gb, m, v);  // v = D^1/2 G rhs
        if (mp > m) PRG_HIP(hipMemsetAsync(v + m * 3, 0, (size_t)(mp - m) * 3 * sizeof(double), st));
        for (int64_t kb = 0; kb < nblk; ++kb) {                   // L u' = v
            const int64_t k0 = kb * NB;
            k_diag_solve<0><<<1, 384, 0, st>>>(linv + (size_t)kb * NB * NB, v, k0);
            const int64_t rows = mp - k0 - NB;
            if (rows > 0) k_fwd_update<<<(unsigned)prg::ceil_div(rows, 64), kBlock, 0, st>>>(S, mp, k0, mp, v);
        }
        for (int64_t kb = nblk - 1; kb >= 0; --kb) {              // L^T u = u'
            const int64_t k0 = kb * NB;
            k_diag_solve<1><<<1, 384, 0, st>>>(linv + (size_t)kb * NB * NB, v, k0);
            if (k0 > 0) k_bwd_update<<<grid1(k0), kBlock, 0, st>>>(S, mp, k0, v);
        }
        k_form_w<<<grid1(m), kBlock, 0, st>>>(rhs, sp, v, m, h->params, lmd, wout);
        PRG_HIP(hipGetLastError());
        return PRG_OK;
    };
    PRG_TRY(solve_with_factor(b3, h->W));
    // With correspondence priors the diagonal scaling spans sigma2/alpha ~ 1e7 and the push-through form
    // loses digits to cancellation; two steps of fp64 iterative refinement on the ORIGINAL system
    // (D G + c I) w = b bring the solution back to the backward-stable level LAPACK gesv delivers.
    const int nrefine = h->nr_alpha > 0.0 ? 2 : 0;
    for (int it = 0; it < nrefine; ++it) {
        PRG_TRY(prg::nonrigid_gw(h, h->W, gb));
        k_residual<<<grid1(mp), kBlock, 0, st>>>(b3, sp, gb, h->W, m, mp, h->params, lmd, r3);
        PRG_TRY(solve_with_factor(r3, dw));
        k_axpy3<<<grid1(m * 3), kBlock, 0, st>>>(dw, m, h->W);
    }
    PRG_TRY(prg::nonrigid_gw(h, h->W, h->nr_work));            // G W, kept for the next E-step's transform
    k_traces<<<tr_blk, kBlock, 0, st>>>(h->rowacc, h->Mcap, h->src4, h->nr_work, m, trpart);
    k_nonrigid_finish<<<1, 64, 0, st>>>(trpart, tr_blk, h->moments, h->params, h->D);
    PRG_HIP(hipGetLastError());
    h->gw_valid = true;
    int host_info = 0;
    PRG_HIP(hipMemcpyAsync(&host_info, info, sizeof(int), hipMemcpyDeviceToHost, st));
    PRG_HIP(hipStreamSynchronize(st));
    PRG_REQUIRE(host_info == 0, PRG_ERR_STATE,
                "prg_cpd_mstep_nonrigid: S is not positive definite at pivot %d (sigma2 or lmd <= 0?)", host_info - 1);
    return PRG_OK;
}

// =================================================================================================
// BCPD M-step core (reference bcpd.py:123-133): with nu = row sums of P, c = s^2 / sigma2_prev^2,
//     Sigma = (lmd G^-1 + c diag(nu))^-1,   v_hat = c Sigma diag(nu) R,   R = T^-1(x_hat) - y.
// The reference inverts G and then the M x M sum explicitly.  Here G^-1 never appears: by Woodbury, with
// D = diag(nu) and the SPD matrix S = (lmd / c) I + D^1/2 G D^1/2 = L L^T,
//     Sigma = (G - B^T S^-1 B) / lmd,   B = D^1/2 G,
// so  diag(Sigma)_m = (g_mm - |L^-1 B e_m|^2) / lmd   (a triangular solve with M right-hand sides, on the
// matrix cores, M^3 flop) and  v_hat = (c / lmd) (G b - B^T S^-1 B b),  b = D R  (3 right-hand sides).
// The form stays finite for nu_m -> 0 (points without support), where Sigma_mm -> g_mm / lmd.
// =================================================================================================
namespace {

__global__ __launch_bounds__(kBlock) void k_bcpd_rhs(const double* __restrict__ rowacc, const double* __restrict__ nu_ext,
                                                     const double* __restrict__ resid, const int* __restrict__ perm,
                                                     int64_t m, int64_t mp, int dim, double* __restrict__ b3,
                                                     double* __restrict__ sp) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= mp) return;
    double nu = 0.0, r[3] = {0.0, 0.0, 0.0};
    if (i < m) {
        const int64_t j = perm ? perm[i] : i;
        nu = fmax(nu_ext ? nu_ext[j] : rowacc[i], 0.0);  // nu_ext is in the caller's order, rowacc in kernel order
        for (int k = 0; k < dim; ++k) r[k] = resid[j * dim + k];
    }
    b3[i * 3] = nu * r[0];
    b3[i * 3 + 1] = nu * r[1];
    b3[i * 3 + 2] = nu * r[2];
    sp[i] = sqrt(nu);
}

// Wt[c][j] = g[c][j] * sp[j]  (= (D^1/2 G)^T, G symmetric); zero in the pad
__global__ __launch_bounds__(kBlock) void k_build_bt(const float* __restrict__ g, int64_t m, int64_t mp,
                                                     const double* __restrict__ sp, double* __restrict__ wt) {
    const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= mp) return;
    const double spj = j < m ? sp[j] : 0.0;
    for (int64_t c = blockIdx.y; c < mp; c += gridDim.y)
        wt[c * mp + j] = (c < m && j < m) ? (double)g[c * m + j] * spj : 0.0;
}

// diag[i] = (g_ii - sum_j Wt[i][j]^2) / lmd ; one wave per row
__global__ __launch_bounds__(kBlock) void k_sigma_diag(const double* __restrict__ wt, const float* __restrict__ g,
                                                       int64_t m, int64_t mp, double lmd, double* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= m) return;
    const double* row = wt + i * mp;
    double a = 0.0;
    for (int64_t j = lane * 2; j < mp; j += 128) {
        const double2 v = *reinterpret_cast<const double2*>(row + j);
        a += v.x * v.x + v.y * v.y;
    }
    a = wave_sum(a);
    if (lane == 0) out[i] = ((double)g[i * m + i] - a) / lmd;
}

__global__ __launch_bounds__(kBlock) void k_bcpd_vhat(const double* __restrict__ gb, const double* __restrict__ gt,
                                                      int64_t m, double f, double* __restrict__ v) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < m * 3) v[i] = f * (gb[i] - gt[i]);
}

__global__ __launch_bounds__(kBlock) void k_unsort_rows(const double* __restrict__ in, int stride_in, int dim,
                                                        int64_t m, const int* __restrict__ perm,
                                                        double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= m) return;
    const int64_t j = perm ? perm[i] : i;
    for (int k = 0; k < dim; ++k) out[j * dim + k] = in[i * stride_in + k];
}

}  // namespace

extern "C" int prg_cpd_bcpd_solve(prg_cpd* h, double lmd, double cfac, const double* nu_hd, const double* resid_hd,
                                  double* vhat_hd, double* sigma_diag_hd) {
    PRG_REQUIRE(h && h->bcpd && h->G && h->W, PRG_ERR_STATE, "prg_cpd_bcpd_solve: needs prg_cpd_bcpd_build_g first");
    PRG_REQUIRE(nu_hd || h->have_estep, PRG_ERR_STATE, "prg_cpd_bcpd_solve: no E-step has run and no nu was given");
    PRG_REQUIRE(resid_hd && vhat_hd && sigma_diag_hd, PRG_ERR_INVALID, "prg_cpd_bcpd_solve: NULL argument");
    PRG_REQUIRE(lmd > 0.0 && cfac > 0.0, PRG_ERR_INVALID, "prg_cpd_bcpd_solve: lmd and c must be > 0 (got %g, %g)", lmd,
                cfac);
    prg::DeviceGuard g(h->device);
    const int64_t m = h->M, mp = prg::round_up(m, NB), nblk = mp / NB;
    const size_t n_s = (size_t)mp * mp, n_linv = (size_t)nblk * NB * NB, n_vec = (size_t)mp * 3;
    const size_t need = (2 * n_s + n_linv + 6 * n_vec + (size_t)mp + 16) * sizeof(double);
    if (h->nr_solve_bytes < need) {
        if (h->nr_solve) (void)hipFree(h->nr_solve);
        h->nr_solve = nullptr;
        h->nr_solve_bytes = 0;
        PRG_HIP(hipMalloc((void**)&h->nr_solve, need));
        h->nr_solve_bytes = need;
        PRG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_potrf_inv),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, NB * LDP * (int)sizeof(double)));
    }
    double* S = h->nr_solve;
    double* wt = S + n_s;
    double* linv = wt + n_s;
    double* b3 = linv + n_linv;
    double* gb = b3 + n_vec;
    double* v = gb + n_vec;
    double* sp = v + n_vec;
    double* gt = sp + n_vec;
    double* tv = gt + n_vec;
    double* diag = tv + n_vec;
    int* info = reinterpret_cast<int*>(diag + mp);
    hipStream_t st = h->stream;

    PRG_TRY(prg::ensure_stage(h, (size_t)m * 4 * sizeof(double)));
    double* nu_dev = nullptr;
    PRG_HIP(hipMemcpyAsync(h->stage, resid_hd, (size_t)m * h->D * sizeof(double), hipMemcpyDefault, st));
    if (nu_hd) {
        nu_dev = (double*)h->stage + (size_t)m * 3;
        PRG_HIP(hipMemcpyAsync(nu_dev, nu_hd, (size_t)m * sizeof(double), hipMemcpyDefault, st));
    }
    PRG_HIP(hipMemsetAsync(info, 0, sizeof(int), st));
    k_bcpd_rhs<<<grid1(mp), kBlock, 0, st>>>(h->rowacc, nu_dev, (const double*)h->stage, h->perm_src, m, mp, h->D, b3,
                                            sp);
    k_build_s<<<dim3((unsigned)nblk, (unsigned)nblk), kBlock, 0, st>>>(h->G, m, mp, sp, nullptr, lmd / cfac, S);
    k_build_bt<<<dim3((unsigned)prg::ceil_div(mp, kBlock), (unsigned)std::min<int64_t>(mp, 32768)), kBlock, 0, st>>>(h->G, m, mp, sp, wt);
    PRG_TRY(cholesky_lookahead(h, S, mp, linv, info, m));

    // Wt <- Wt L^-T, left-looking over panels of k 128-column blocks: one K-deep rectangular update with everything
    // to the left of the panel, then k (update inside the panel, multiply by the inverted diagonal block) steps.
    // The update runs nblk x k workgroups, 512 at a time (two per CU): k is chosen so that this is close to a whole
    // number of rounds - at M = 20k (157 block rows) four blocks per panel would leave the second round 23 % full.
    int kpanel = 4;
    {
        double best = 1e30;
        for (int k = 4; k <= 16; ++k) {
            const double wgs = (double)nblk * k;
            const double waste = ceil(wgs / 512.0) * 512.0 / wgs;
            if (waste <= best + 1e-9) {  // ties: the wider panel (fewer, larger launches)
                best = waste;
                kpanel = k;
            }
        }
    }
    const int64_t NBO = (int64_t)kpanel * NB;
    for (int64_t K0 = 0; K0 < mp; K0 += NBO) {
        const int64_t kend = std::min<int64_t>(K0 + NBO, mp);
        if (K0 > 0)
            k_gemm_nt_f64<3><<<dim3((unsigned)nblk, (unsigned)((kend - K0) / NB)), kBlock, 0, st>>>(
                wt, S + K0 * mp, mp, mp, wt + K0, mp, (int)K0);
        for (int64_t k0 = K0; k0 < kend; k0 += NB) {
            if (k0 > K0)
                k_gemm_nt_f64<3><<<dim3((unsigned)nblk, 1u), kBlock, 0, st>>>(wt + K0, S + k0 * mp + K0, mp, mp, wt + k0, mp,
                                                                             (int)(k0 - K0));
            k_gemm_nt_f64<0><<<(unsigned)nblk, kBlock, 0, st>>>(wt + k0, linv + (size_t)(k0 / NB) * NB * NB, mp, NB,
                                                               wt + k0, mp, NB);
        }
    }
    k_sigma_diag<<<(unsigned)prg::ceil_div(m, 4), kBlock, 0, st>>>(wt, h->G, m, mp, lmd, diag);

    // v_hat = (c / lmd) (G b - G D^1/2 S^-1 D^1/2 G b)
    PRG_TRY(prg::nonrigid_gw(h, b3, gb));
    k_scale_rows<<<grid1(m), kBlock, 0, st>>>(sp, gb, m, v);
    if (mp > m) PRG_HIP(hipMemsetAsync(v + m * 3, 0, (size_t)(mp - m) * 3 * sizeof(double), st));
    for (int64_t kb = 0; kb < nblk; ++kb) {
        const int64_t k0 = kb * NB;
        k_diag_solve<0><<<1, 384, 0, st>>>(linv + (size_t)kb * NB * NB, v, k0);
        const int64_t rows = mp - k0 - NB;
        if (rows > 0) k_fwd_update<<<(unsigned)prg::ceil_div(rows, 64), kBlock, 0, st>>>(S, mp, k0, mp, v);
    }
    for (int64_t kb = nblk - 1; kb >= 0; --kb) {
        const int64_t k0 = kb * NB;
        k_diag_solve<1><<<1, 384, 0, st>>>(linv + (size_t)kb * NB * NB, v, k0);
        if (k0 > 0) k_bwd_update<<<grid1(k0), kBlock, 0, st>>>(S, mp, k0, v);
    }
    k_scale_rows<<<grid1(m), kBlock, 0, st>>>(sp, v, m, tv);
    PRG_TRY(prg::nonrigid_gw(h, tv, gt));
    k_bcpd_vhat<<<grid1(m * 3), kBlock, 0, st>>>(gb, gt, m, cfac / lmd, h->W);

    // results back in the caller's point order
    double* stage = (double*)h->stage;
    k_unsort_rows<<<grid1(m), kBlock, 0, st>>>(h->W, 3, h->D, m, h->perm_src, stage);
    PRG_HIP(hipMemcpyAsync(vhat_hd, stage, (size_t)m * h->D * sizeof(double), hipMemcpyDefault, st));
    k_unsort_rows<<<grid1(m), kBlock, 0, st>>>(diag, 1, 1, m, h->perm_src, gb);
    PRG_HIP(hipMemcpyAsync(sigma_diag_hd, gb, (size_t)m * sizeof(double), hipMemcpyDefault, st));
    PRG_HIP(hipGetLastError());
    int host_info = 0;
    PRG_HIP(hipMemcpyAsync(&host_info, info, sizeof(int), hipMemcpyDeviceToHost, st));
    PRG_HIP(hipStreamSynchronize(st));
    PRG_REQUIRE(host_info == 0, PRG_ERR_STATE, "prg_cpd_bcpd_solve: S is not positive definite at pivot %d",
                host_info - 1);
    return PRG_OK;
}
