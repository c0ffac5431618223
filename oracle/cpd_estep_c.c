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

/* t_source: M x D, target: N x D (row-major float64); outputs pt1[N], p1[M], px[M x D]; returns n_p */
double cpd_oracle_estep(const double* ts, int64_t m, const double* x, int64_t n, int d, double sigma2, double w,
                        double* pt1, double* p1, double* px) {
    const double inv = -1.0 / (2.0 * sigma2);
    double c = pow(2.0 * M_PI * sigma2, d * 0.5);
    c *= w / (1.0 - w) * (double)m / (double)n;
    double* den = (double*)malloc(sizeof(double) * (size_t)n);
    double* kmass = (double*)malloc(sizeof(double) * (size_t)n);
#pragma omp parallel for schedule(static)
    for (int64_t j = 0; j < n; ++j) {
        double s = 0.0;
        for (int64_t i = 0; i < m; ++i) {
            double d2 = 0.0;
            for (int k = 0; k < d; ++k) {
                const double df = ts[i * d + k] - x[j * d + k];
                d2 += df * df;
            }
            s += exp(d2 * inv);
        }
        kmass[j] = s;
        if (s == 0.0) s = EPS32;
        den[j] = s + c;
        pt1[j] = kmass[j] / den[j];
    }
    double n_p = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : n_p)
    for (int64_t i = 0; i < m; ++i) {
        double s = 0.0, acc[3] = {0.0, 0.0, 0.0};
        for (int64_t j = 0; j < n; ++j) {
            double d2 = 0.0;
            for (int k = 0; k < d; ++k) {
                const double df = ts[i * d + k] - x[j * d + k];
                d2 += df * df;
            }
            const double p = exp(d2 * inv) / den[j];
            s += p;
            for (int k = 0; k < d; ++k) acc[k] += p * x[j * d + k];
        }
        p1[i] = s;
        for (int k = 0; k < d; ++k) px[i * d + k] = acc[k];
        n_p += s;
    }
    free(den);
    free(kmass);
    return n_p;
}
