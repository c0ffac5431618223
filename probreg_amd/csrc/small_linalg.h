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
    double nmax = 0.0;
    for (int j = 0; j < d; ++j) {
        double nn = 0;
        for (int i = 0; i < d; ++i) nn += g[i][j] * g[i][j];
        sv[j] = sqrt(nn);
        nmax = fmax(nmax, sv[j]);
    }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) U[i][j] = (i == j) ? 1.0 : 0.0;
    bool ok[3] = {false, false, false};
    for (int j = 0; j < d; ++j) {
        ok[j] = sv[j] > 1e-14 * nmax && sv[j] > 0.0;
        if (ok[j])
            for (int i = 0; i < d; ++i) U[i][j] = g[i][j] / sv[j];
    }
    // complete U to an orthonormal basis where singular values vanish (rank-deficient `a`)
    if (d == 2) {
        if (ok[0] && !ok[1]) { U[0][1] = -U[1][0]; U[1][1] = U[0][0]; }
        else if (!ok[0] && ok[1]) { U[0][0] = U[1][1]; U[1][0] = -U[0][1]; }
        else if (!ok[0] && !ok[1]) { U[0][0] = U[1][1] = 1.0; U[0][1] = U[1][0] = 0.0; }
    } else if (d == 3) {
        int nbad = (!ok[0]) + (!ok[1]) + (!ok[2]);
        if (nbad == 1) {
            int b = !ok[0] ? 0 : (!ok[1] ? 1 : 2);
            int p = (b + 1) % 3, q = (b + 2) % 3;
            U[0][b] = U[1][p] * U[2][q] - U[2][p] * U[1][q];
            U[1][b] = U[2][p] * U[0][q] - U[0][p] * U[2][q];
            U[2][b] = U[0][p] * U[1][q] - U[1][p] * U[0][q];
        } else if (nbad >= 2) {
            int gidx = ok[0] ? 0 : (ok[1] ? 1 : (ok[2] ? 2 : -1));
            if (gidx < 0) {
                for (int i = 0; i < 3; ++i)
                    for (int j = 0; j < 3; ++j) U[i][j] = (i == j) ? 1.0 : 0.0;
            } else {
                // pick the coordinate axis least aligned with the good column, Gram-Schmidt, cross
                int ax = 0;
                double best = fabs(U[0][gidx]);
                for (int i = 1; i < 3; ++i)
                    if (fabs(U[i][gidx]) < best) { best = fabs(U[i][gidx]); ax = i; }
                double v[3] = {0, 0, 0};
                v[ax] = 1.0;
                double dp = U[ax][gidx];
                double nn = 0;
                for (int i = 0; i < 3; ++i) { v[i] -= dp * U[i][gidx]; nn += v[i] * v[i]; }
                nn = sqrt(nn);
                int p = (gidx + 1) % 3, q = (gidx + 2) % 3;
                for (int i = 0; i < 3; ++i) U[i][p] = v[i] / nn;
                U[0][q] = U[1][gidx] * U[2][p] - U[2][gidx] * U[1][p];
                U[1][q] = U[2][gidx] * U[0][p] - U[0][gidx] * U[2][p];
                U[2][q] = U[0][gidx] * U[1][p] - U[1][gidx] * U[0][p];
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
