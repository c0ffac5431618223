// Tiny fp64 dense linear algebra used by the single-thread M-step kernels (device only).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

namespace prg {

// One-sided Jacobi SVD of a d x d (d <= 3) matrix: a = U diag(sv) V^T.  Returns U, V, sv (unsorted).
__device__ inline void jacobi_svd(const double a[3][3], int d, double U[3][3], double V[3][3], double sv[3]) {
    double g[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            g[i][j] = (i < d && j < d) ? a[i][j] : 0.0;
            V[i][j] = (i == j) ? 1.0 : 0.0;
        }
    // p, q loops are fully unrolled with compile-time indices so that g / V stay in registers
    // (run-time column indices would push the 3x3 arrays to scratch memory: 30 us instead of ~5)
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
#pragma unroll
            for (int q = p + 1; q < 3; ++q) {
                if (q >= d) continue;
                double alpha = 0, beta = 0, gamma = 0;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    alpha += g[i][p] * g[i][p];
                    beta += g[i][q] * g[i][q];
                    gamma += g[i][p] * g[i][q];
                }
                const double ab = sqrt(alpha * beta);
                if (fabs(gamma) <= 1e-300 + 1e-17 * ab) continue;
                off = fmax(off, fabs(gamma) / ab);
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const double gp = g[i][p], gq = g[i][q];
                    g[i][p] = c * gp - s * gq;
                    g[i][q] = s * gp + c * gq;
                    const double vp = V[i][p], vq = V[i][q];
                    V[i][p] = c * vp - s * vq;
                    V[i][q] = s * vp + c * vq;
                }
            }
        }
        if (off < 1e-15) break;
    }
    // Everything below uses compile-time array indices only (fully unrolled loops with predicates): a single
    // run-time index into g / U / V would move the 3 x 3 arrays to scratch memory (~1 us per access).
    double nmax = 0.0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        double nn = 0;
#pragma unroll
        for (int i = 0; i < 3; ++i) nn += g[i][j] * g[i][j];
        sv[j] = (j < d) ? sqrt(nn) : 0.0;
        nmax = fmax(nmax, sv[j]);
    }
    bool ok[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        ok[j] = (j < d) && sv[j] > 1e-14 * nmax && sv[j] > 0.0;
        const double inv = ok[j] ? 1.0 / sv[j] : 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i) U[i][j] = ok[j] ? g[i][j] * inv : ((i == j) ? 1.0 : 0.0);
    }
    // complete U to an orthonormal basis where singular values vanish (rank-deficient `a`)
    if (d == 2) {
        if (ok[0] && !ok[1]) { U[0][1] = -U[1][0]; U[1][1] = U[0][0]; }
        else if (!ok[0] && ok[1]) { U[0][0] = U[1][1]; U[1][0] = -U[0][1]; }
        else if (!ok[0] && !ok[1]) { U[0][0] = U[1][1] = 1.0; U[0][1] = U[1][0] = 0.0; }
    } else if (d == 3) {
        const int nbad = (!ok[0]) + (!ok[1]) + (!ok[2]);
        if (nbad == 1) {
#pragma unroll
            for (int bcol = 0; bcol < 3; ++bcol) {
                if (ok[bcol]) continue;
                constexpr int nxt[3] = {1, 2, 0}, nx2[3] = {2, 0, 1};
                const int p = nxt[bcol], q = nx2[bcol];  // compile-time after unrolling
                U[0][bcol] = U[1][p] * U[2][q] - U[2][p] * U[1][q];
                U[1][bcol] = U[2][p] * U[0][q] - U[0][p] * U[2][q];
                U[2][bcol] = U[0][p] * U[1][q] - U[1][p] * U[0][q];
            }
        } else if (nbad == 3) {
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) U[i][j] = (i == j) ? 1.0 : 0.0;
        } else if (nbad == 2) {
#pragma unroll
            for (int gcol = 0; gcol < 3; ++gcol) {
                if (!ok[gcol]) continue;
                constexpr int nxt[3] = {1, 2, 0}, nx2[3] = {2, 0, 1};
                const int p = nxt[gcol], q = nx2[gcol];
                // coordinate axis least aligned with the good column, Gram-Schmidt, cross product
                const double a0 = fabs(U[0][gcol]), a1 = fabs(U[1][gcol]), a2 = fabs(U[2][gcol]);
                const int ax = (a0 <= a1 && a0 <= a2) ? 0 : ((a1 <= a2) ? 1 : 2);
                const double dp = (ax == 0) ? U[0][gcol] : ((ax == 1) ? U[1][gcol] : U[2][gcol]);
                double v[3], nn = 0.0;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    v[i] = ((i == ax) ? 1.0 : 0.0) - dp * U[i][gcol];
                    nn += v[i] * v[i];
                }
                nn = 1.0 / sqrt(nn);
#pragma unroll
                for (int i = 0; i < 3; ++i) U[i][p] = v[i] * nn;
                U[0][q] = U[1][gcol] * U[2][p] - U[2][gcol] * U[1][p];
                U[1][q] = U[2][gcol] * U[0][p] - U[0][gcol] * U[2][p];
                U[2][q] = U[0][gcol] * U[1][p] - U[1][gcol] * U[0][p];
            }
        }
    }
}

__device__ inline double det3(const double a[3][3], int d) {
    if (d == 2) return a[0][0] * a[1][1] - a[0][1] * a[1][0];
    return a[0][0] * (a[1][1] * a[2][2] - a[1][2] * a[2][1]) - a[0][1] * (a[1][0] * a[2][2] - a[1][2] * a[2][0]) +
           a[0][2] * (a[1][0] * a[2][1] - a[1][1] * a[2][0]);
}

}  // namespace prg
