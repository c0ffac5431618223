/*
 * C restatement of probreg's CPD E-step + rigid-moment pass (TEST INFRASTRUCTURE / CPU baseline).
 *
 * Follows /root/reference/probreg/cpd.py:71-88 line by line in float64:
 *   pmat = exp(-cdist(t_source, target, 'sqeuclidean') / (2 sigma2))        :74-76
 *   c    = (2 pi sigma2)^(D/2) * w/(1-w) * M/N                              :78-79
 *   den  = pmat.sum(axis=0); den[den == 0] = eps32; den += c                :80-82
 *   pmat /= den; pt1 = pmat.sum(0); p1 = pmat.sum(1); px = pmat @ target    :84-87
 * but never stores the M x N matrix: sweep 1 owns columns (den), sweep 2 owns rows (p1, px) and
 * recomputes the exponentials.  OpenMP over the owned axis; summation order differs from numpy's
 * pairwise reductions only at the 1e-16 level.  Checked against oracle/cpd_numpy.py in
 * tests/test_oracle_c.py.  Parity status: PINNED (via cpd_numpy, which is pinned to the reference).
 *
 * Build: gcc -O3 -march=native -fopenmp -shared -fPIC cpd_estep_c.c -o libcpd_oracle.so -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define EPS32 1.1920928955078125e-07

int cpd_oracle_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* exp(a) for a < -745.2 is exactly 0.0 in IEEE double (the smallest subnormal is exp(-744.44); everything
 * below -745.14 rounds to 0), so the call is skipped there - bit-identical to calling it, and what keeps a
 * late-EM-iteration E-step (sigma2 ~ 1e-5: almost every pair underflows) affordable at N = M = 1e5. */
static inline double exp_or_zero(double a) { return a < -745.2 ? 0.0 : exp(a); }

#define TILE 512 /* streamed points kept in L1 while a block of owned points sweeps over them */
#define OWN 64   /* owned points per work item */

/* t_source: M x D, target: N x D (row-major float64); outputs pt1[N], p1[M], px[M x D]; returns n_p */
double cpd_oracle_estep(const double* ts, int64_t m, const double* x, int64_t n, int d, double sigma2, double w,
                        double* pt1, double* p1, double* px) {
    const double inv = -1.0 / (2.0 * sigma2);
    double c = pow(2.0 * M_PI * sigma2, d * 0.5);
    c *= w / (1.0 - w) * (double)m / (double)n;
    double* den = (double*)malloc(sizeof(double) * (size_t)n);
    /* sweep 1 (cpd.py:74-82): den_j = sum_i exp(-|ts_i - x_j|^2 / 2 sigma2), columns owned, sources streamed in tiles */
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t j0 = 0; j0 < n; j0 += OWN) {
        const int64_t j1 = j0 + OWN < n ? j0 + OWN : n;
        double s[OWN];
        for (int q = 0; q < OWN; ++q) s[q] = 0.0;
        for (int64_t i0 = 0; i0 < m; i0 += TILE) {
            const int64_t i1 = i0 + TILE < m ? i0 + TILE : m;
            for (int64_t j = j0; j < j1; ++j) {
                double acc = s[j - j0];
                for (int64_t i = i0; i < i1; ++i) {
                    double d2 = 0.0;
                    for (int k = 0; k < d; ++k) {
                        const double df = ts[i * d + k] - x[j * d + k];
                        d2 += df * df;
                    }
                    acc += exp_or_zero(d2 * inv);
                }
                s[j - j0] = acc;
            }
        }
        for (int64_t j = j0; j < j1; ++j) {
            const double kmass = s[j - j0];
            double sj = kmass;
            if (sj == 0.0) sj = EPS32; /* cpd.py:81 */
            den[j] = sj + c;           /* cpd.py:82 */
            pt1[j] = kmass / den[j];   /* column sum of P (cpd.py:85) */
        }
    }
    /* sweep 2 (cpd.py:84-87): P = K / den, p1 = row sums, px = P @ target; rows owned, targets streamed in tiles */
    double n_p = 0.0;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : n_p)
    for (int64_t i0 = 0; i0 < m; i0 += OWN) {
        const int64_t i1 = i0 + OWN < m ? i0 + OWN : m;
        double s[OWN], acc[OWN][3];
        for (int q = 0; q < OWN; ++q) s[q] = acc[q][0] = acc[q][1] = acc[q][2] = 0.0;
        for (int64_t jt = 0; jt < n; jt += TILE) {
            const int64_t je = jt + TILE < n ? jt + TILE : n;
            for (int64_t i = i0; i < i1; ++i) {
                double si = s[i - i0], a0 = acc[i - i0][0], a1 = acc[i - i0][1], a2 = acc[i - i0][2];
                for (int64_t j = jt; j < je; ++j) {
                    double d2 = 0.0;
                    for (int k = 0; k < d; ++k) {
                        const double df = ts[i * d + k] - x[j * d + k];
                        d2 += df * df;
                    }
                    const double e = exp_or_zero(d2 * inv);
                    if (e == 0.0) continue; /* contributes exact zeros */
                    const double p = e / den[j];
                    si += p;
                    a0 += p * x[j * d];
                    a1 += p * x[j * d + 1];
                    if (d > 2) a2 += p * x[j * d + 2];
                }
                s[i - i0] = si;
                acc[i - i0][0] = a0;
                acc[i - i0][1] = a1;
                acc[i - i0][2] = a2;
            }
        }
        for (int64_t i = i0; i < i1; ++i) {
            p1[i] = s[i - i0];
            for (int k = 0; k < d; ++k) px[i * d + k] = acc[i - i0][k];
            n_p += s[i - i0];
        }
    }
    free(den);
    return n_p;
}

/* ---- non-rigid M-step pieces at sizes numpy cannot hold (tests/test_fullsize_gpu.py, C3 at N = M = 50 000) ----------
 * The reference's G (transformation.py:91-99 -> cc/math_utils.cc:17-19) is a float32 matrix: squared distance and exp in
 * float32 (un-fused: this file is built for x86-64-v2, which has no FMA) as in oracle/cpd_numpy.py:rbf_kernel; the squared
 * distances are bit-identical to that function's, the exponential is glibc's expf (correctly rounded in > 99.9 % of the cases)
 * where numpy and Eigen use their own vectorised float32 exp (each within 1 ulp): entries agree to 1 float32 ulp.  The two
 * functions below evaluate the entries on the fly instead of storing M x M floats. */
static inline float rbf32(const float* a, const float* b, int d, float den) {
    float s = 0.f;
    for (int k = 0; k < d; ++k) {
        const float t = a[k] - b[k];
        s += t * t;
    }
    return expf(-s / den);
}

/* a (column-major M x M float64, LAPACK layout) = (p1 * g).T + c I of cpd.py:297, i.e. a[i][j] = p1[i] g[i][j] + c delta_ij */
void cpd_oracle_nonrigid_lhs(const float* y32, int64_t m, int d, double beta, const double* p1, double c, double* a) {
    const float den = (float)(2.0 * beta);
#pragma omp parallel for schedule(static)
    for (int64_t j = 0; j < m; ++j) {
        double* col = a + j * m;
        const float* yj = y32 + j * d;
        for (int64_t i = 0; i < m; ++i) col[i] = p1[i] * (double)rbf32(y32 + i * d, yj, d, den);
        col[j] += c;
    }
}

/* out (M x d float64) = g @ w with float64 accumulation (numpy upcasts the float32 g for the product, cpd.py:298) */
void cpd_oracle_nonrigid_gw(const float* y32, int64_t m, int d, double beta, const double* w, double* out) {
    const float den = (float)(2.0 * beta);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < m; ++i) {
        double acc[3] = {0.0, 0.0, 0.0};
        const float* yi = y32 + i * d;
        for (int64_t j = 0; j < m; ++j) {
            const double g = (double)rbf32(yi, y32 + j * d, d, den);
            for (int k = 0; k < d; ++k) acc[k] += g * w[j * d + k];
        }
        for (int k = 0; k < d; ++k) out[i * d + k] = acc[k];
    }
}
