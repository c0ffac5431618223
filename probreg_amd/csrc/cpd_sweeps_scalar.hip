// E-step pair sweeps, one-fp32-op-per-pair form (this file is compiled with -fno-slp-vectorize so
// that hipcc does not pack the arithmetic): the A/B partner of cpd_sweeps_packed.hip, selected
// with negative points-per-lane in prg_cpd_set_tuning.  Same structure, same outputs.
#include <math.h>

#include "cpd_sweeps.h"

namespace {
constexpr double kLog2e = 1.4426950408889634;
constexpr int kBlock = prg::kSweepBlock;
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// ---------------------------------------------------------------------------------------------
// E-step sweep 1: column pass (den of cpd.py:80 in (min d^2, sum exp2) form)
// ---------------------------------------------------------------------------------------------
// grid = (Ncap / (256 R), S); block b.y streams source segment [b.y*seg_len, +seg_len).
// The source pointer is wave-uniform and read-only for the kernel, so hipcc emits s_load_dwordx16
// (4 points per scalar load); each point then feeds R/2 packed-fp32 VALU chains.
template <int R, int CH>
__global__ __launch_bounds__(kBlock) void k_colpass(const float4* __restrict__ tgt4, const float4* __restrict__ z4,
                                                    int seg_len, const double* __restrict__ params,
                                                    float2* __restrict__ colpart, int64_t ncap) {
    const float kk = (float)(-kLog2e / (2.0 * params[13]));
    const int64_t n0 = (int64_t)blockIdx.x * (kBlock * R) + threadIdx.x;
    float x[R], y[R], z[R], run[R], off[R], s[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const float4 v = tgt4[n0 + r * kBlock];
        x[r] = v.x; y[r] = v.y; z[r] = v.z;
        run[r] = INFINITY;
        off[r] = INFINITY;
        s[r] = 0.f;
    }
    const float4* __restrict__ zp = z4 + (int64_t)blockIdx.y * seg_len;
    for (int m = 0; m < seg_len; m += CH) {
        float d2[CH][R];
        float cm[R];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const float4 q = zp[m + c];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float dx = x[r] - q.x, dy = y[r] - q.y, dz = z[r] - q.z;
                float d = fmaf(dx, dx, q.w);  // q.w = per-source weight term (0 for plain CPD)
                d = fmaf(dy, dy, d);
                d = fmaf(dz, dz, d);
                d2[c][r] = d;
            }
        }
        bool lower = false;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float v = d2[0][r];
#pragma unroll
            for (int c = 1; c < CH; ++c) v = fminf(v, d2[c][r]);
            cm[r] = v;
            lower |= v < run[r];
        }
        if (lower) {  // rare after the first few chunks: rescale the running sums to the new minimum
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float nm = fminf(run[r], cm[r]);
                const float noff = prg::col_offset(kk, nm);
                s[r] *= fast_exp2(noff - off[r]);  // off == +inf on first use: exp2(-inf) = 0 and s == 0 anyway
                run[r] = nm;
                off[r] = noff;
            }
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) {
#pragma unroll
            for (int r = 0; r < R; ++r) s[r] += fast_exp2(fmaf(d2[c][r], kk, off[r]));
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
        colpart[(int64_t)blockIdx.y * ncap + n0 + r * kBlock] = make_float2(run[r], s[r]);
}

// ---------------------------------------------------------------------------------------------
// E-step sweep 2: row pass (cpd.py:84-87 in residual form)
// ---------------------------------------------------------------------------------------------
// grid = (Mcap / (256 R), S); block b.y streams target segment [b.y*seg_len, +seg_len) of (x_n, b_n).
// Output plane layout: rowpart[(seg*5 + comp) * Mcap + m], comp = p1, ux, uy, uz, e.
template <int R>
__global__ __launch_bounds__(kBlock) void k_rowpass(const float4* __restrict__ z4, const float4* __restrict__ tgt4,
                                                    int seg_len, const double* __restrict__ params,
                                                    float* __restrict__ rowpart, int64_t mcap) {
    const float kk = (float)(-kLog2e / (2.0 * params[13]));
    const int64_t m0 = (int64_t)blockIdx.x * (kBlock * R) + threadIdx.x;
    float zx[R], zy[R], zz[R], zq[R], p1[R], ux[R], uy[R], uz[R], e[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const float4 v = z4[m0 + r * kBlock];
        zx[r] = v.x; zy[r] = v.y; zz[r] = v.z; zq[r] = v.w;
        p1[r] = ux[r] = uy[r] = uz[r] = e[r] = 0.f;
    }
    const float4* __restrict__ tp = tgt4 + (int64_t)blockIdx.y * seg_len;
    for (int n = 0; n < seg_len; n += 4) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float4 q = tp[n + c];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float dx = zx[r] - q.x, dy = zy[r] - q.y, dz = zz[r] - q.z;
                float d = fmaf(dx, dx, zq[r]);
                d = fmaf(dy, dy, d);
                d = fmaf(dz, dz, d);
                const float p = fast_exp2(fmaf(d, kk, q.w));
                p1[r] += p;
                ux[r] = fmaf(p, dx, ux[r]);
                uy[r] = fmaf(p, dy, uy[r]);
                uz[r] = fmaf(p, dz, uz[r]);
                e[r] = fmaf(p, d, e[r]);
            }
        }
    }
    float* __restrict__ o = rowpart + (int64_t)blockIdx.y * 5 * mcap;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t m = m0 + r * kBlock;
        o[m] = p1[r];
        o[mcap + m] = -ux[r];  // u = sum P (x - z) = -sum P (z - x)
        o[2 * mcap + m] = -uy[r];
        o[3 * mcap + m] = -uz[r];
        o[4 * mcap + m] = fmaf(-zq[r], p1[r], e[r]);
    }
}

}  // namespace

namespace prg {

void launch_colpass_scalar(prg_cpd* h, int R, int S, int seg_len) {
    dim3 grid((unsigned)ceil_div(h->N, kBlock * R), (unsigned)S);
    h->wg_col = 0;
    h->dense_pairs_col = (double)grid.x * (kBlock * R) * (double)S * seg_len;
    if (R == 2)
        k_colpass<2, 8><<<grid, kBlock, 0, h->stream>>>(h->tgt4, h->z4, seg_len, h->params, h->colpart, h->Ncap);
    else
        k_colpass<4, 4><<<grid, kBlock, 0, h->stream>>>(h->tgt4, h->z4, seg_len, h->params, h->colpart, h->Ncap);
}

void launch_rowpass_scalar(prg_cpd* h, int R, int S, int seg_len) {
    dim3 grid((unsigned)ceil_div(h->M, kBlock * R), (unsigned)S);
    h->wg_row = 0;
    h->dense_pairs_row = (double)grid.x * (kBlock * R) * (double)S * seg_len;
    if (R == 2)
        k_rowpass<2><<<grid, kBlock, 0, h->stream>>>(h->z4, h->tgt4, seg_len, h->params, h->rowpart, h->Mcap);
    else
        k_rowpass<4><<<grid, kBlock, 0, h->stream>>>(h->z4, h->tgt4, seg_len, h->params, h->rowpart, h->Mcap);
}

}  // namespace prg
