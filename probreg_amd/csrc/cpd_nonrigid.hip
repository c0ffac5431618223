// Non-rigid CPD pieces for MI355X (gfx950).
//
// Reference behaviour (neka-nat/probreg v0.3.7):
//   G build        probreg/transformation.py:91-99 -> probreg/cc/math_utils.cc:17-19
//   T(Y) = Y + G W probreg/transformation.py:101-102
//   M-step         probreg/cpd.py:284-303
#include <math.h>

#include <algorithm>
#include <vector>

#include "cpd_plan.h"

namespace {
constexpr int kBlock = 256;

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// G[i][j] = exp(-|y_i - y_j|^2 / (2 beta)) in float32: squared distance in float32 without FMA
// contraction (as the reference's Eigen expression), exponential in fp64 rounded once.
// KIND 1: inverse multiquadric 1 / sqrt(d2 + c), all float32 (cc/math_utils.cc:32-34; BCPD's kernel, bcpd.py:107).
template <int KIND>
__global__ __launch_bounds__(kBlock) void k_build_g(const float4* __restrict__ src4, int64_t m, float two_beta,
                                                    float* __restrict__ g) {
    const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t i0 = (int64_t)blockIdx.y * 16;
    if (j >= m) return;
    const float4 yj = src4[j];
    for (int64_t i = i0; i < i0 + 16 && i < m; ++i) {
        const float4 yi = src4[i];  // wave-uniform -> scalar load
        const float dx = __fsub_rn(yi.x, yj.x), dy = __fsub_rn(yi.y, yj.y), dz = __fsub_rn(yi.z, yj.z);
        const float d2 = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
        if (KIND == 0)
            g[i * m + j] = (float)exp((double)__fdiv_rn(-d2, two_beta));
        else  // two_beta carries c
            g[i * m + j] = __fdiv_rn(1.f, __fsqrt_rn(__fadd_rn(d2, two_beta)));
    }
}

// out[i][0..2] = sum_j G[i][j] W[j][0..2] in fp64 (the reference upcasts the float32 G, numpy dot).
// Block = 8 rows; thread t owns columns {4t..4t+3} + 1024 k of every row, W values held in registers
// across the 8 rows so G (HBM-bound, read once) dominates the traffic.
constexpr int kGwRows = 8;
__global__ __launch_bounds__(kBlock) void k_gw(const float* __restrict__ g, int64_t m, const double* __restrict__ w,
                                               double* __restrict__ out) {
    __shared__ double sh[4][kGwRows][3];
    const int64_t r0 = (int64_t)blockIdx.x * kGwRows;
    double acc[kGwRows][3];
#pragma unroll
    for (int r = 0; r < kGwRows; ++r) acc[r][0] = acc[r][1] = acc[r][2] = 0.0;
    const bool vec_ok = (m % 4) == 0;
    for (int64_t j0 = (int64_t)threadIdx.x * 4; j0 < m; j0 += kBlock * 4) {
        double wv[4][3];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int64_t j = j0 + c;
#pragma unroll
            for (int k = 0; k < 3; ++k) wv[c][k] = j < m ? w[j * 3 + k] : 0.0;
        }
#pragma unroll
        for (int r = 0; r < kGwRows; ++r) {
            const int64_t i = r0 + r;
            if (i >= m) continue;
            float gv[4];
            if (vec_ok) {
                const float4 t = *reinterpret_cast<const float4*>(g + i * m + j0);
                gv[0] = t.x; gv[1] = t.y; gv[2] = t.z; gv[3] = t.w;
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) gv[c] = (j0 + c) < m ? g[i * m + j0 + c] : 0.f;
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const double gd = gv[c];
#pragma unroll
                for (int k = 0; k < 3; ++k) acc[r][k] = fma(gd, wv[c][k], acc[r][k]);
            }
        }
    }
    const int lane = threadIdx.x & 63, wvid = threadIdx.x >> 6;
#pragma unroll
    for (int r = 0; r < kGwRows; ++r)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double s = wave_sum(acc[r][k]);
            if (lane == 0) sh[wvid][r][k] = s;
        }
    __syncthreads();
    if (threadIdx.x < kGwRows * 3) {
        const int r = threadIdx.x / 3, k = threadIdx.x % 3;
        if (r0 + r < m) out[(r0 + r) * 3 + k] = sh[0][r][k] + sh[1][r][k] + sh[2][r][k] + sh[3][r][k];
    }
}

// caller's [m][dim] (original point order) <-> the plan's [m][3] (kernel order: sorted position i = original perm[i])
__global__ __launch_bounds__(kBlock) void k_pack_w(const double* __restrict__ in, int64_t m, int dim,
                                                   double* __restrict__ w3, int to_w3, const int* __restrict__ perm) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= m) return;
    const int64_t j = perm ? perm[i] : i;
    if (to_w3) {
        for (int k = 0; k < 3; ++k) w3[i * 3 + k] = k < dim ? in[j * dim + k] : 0.0;
    } else {
        double* out = const_cast<double*>(in);
        for (int k = 0; k < dim; ++k) out[j * dim + k] = w3[i * 3 + k];
    }
}

// caller's p1 [m] and px [m][dim] -> 4 planes of m in kernel order (missing dims zero)
__global__ __launch_bounds__(kBlock) void k_unpack_planes(const double* __restrict__ p1, const double* __restrict__ in,
                                                          int64_t m, int dim, double* __restrict__ planes,
                                                          const int* __restrict__ perm) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= m) return;
    const int64_t j = perm ? perm[i] : i;
    planes[i] = p1[j];
    for (int k = 0; k < 3; ++k) planes[(int64_t)(k + 1) * m + i] = k < dim ? in[j * dim + k] : 0.0;
}

// kernel-order points back in the caller's order (float4 layout kept)
__global__ __launch_bounds__(kBlock) void k_unsort4(const float4* __restrict__ in, int64_t m, const int* __restrict__ perm,
                                                    float4* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < m) out[perm ? perm[i] : i] = in[i];
}

// ---- low-rank factor of the Gaussian kernel matrix -------------------------------------------------------------
// G_ij = exp(-|y_i - y_j|^2 / (2 beta)) is numerically low rank for the smooth kernels CPD uses (C3: beta = 2 on a unit-sized
// cloud -> rank ~180 at 1e-14, whatever M).  Greedy pivoted Cholesky with the columns evaluated on the fly in fp64:
//     step j: p = argmax_i d_i ;  F[j][i] = (G_ip - sum_{k<j} F[k][i] F[k][p]) / sqrt(d_p) ;  d_i -= F[j][i]^2
// G - F F^T is positive semi-definite with diagonal d, so max d bounds every entry of the remainder and sum d its norm.
// One launch per step: every workgroup re-derives the pivot from the per-workgroup maxima the previous step left.
struct PcholState {
    int rank;
    int done;
    double dmax;
};
constexpr int kMaxRank = 2048;

__device__ __forceinline__ void argmax_merge(double& v, int& i, double ov, int oi) {
    if (ov > v || (ov == v && oi < i)) {
        v = ov;
        i = oi;
    }
}

// workgroup argmax of (v, i) (ties: smaller index); result valid in every thread
__device__ __forceinline__ void block_argmax(double& v, int& i) {
    __shared__ double sv[kBlock / 64];
    __shared__ int si[kBlock / 64];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double ov = __shfl_xor(v, o, 64);
        const int oi = __shfl_xor(i, o, 64);
        argmax_merge(v, i, ov, oi);
    }
    __syncthreads();  // (the arrays may still be read from a previous call)
    if ((threadIdx.x & 63) == 0) {
        sv[threadIdx.x >> 6] = v;
        si[threadIdx.x >> 6] = i;
    }
    __syncthreads();
    v = sv[0];
    i = si[0];
#pragma unroll
    for (int k = 1; k < kBlock / 64; ++k) argmax_merge(v, i, sv[k], si[k]);
}

__global__ __launch_bounds__(kBlock) void k_pchol_init(int64_t m, int64_t mp, double* __restrict__ d,
                                                       double2* __restrict__ part, PcholState* __restrict__ state) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < mp) d[i] = i < m ? 1.0 : 0.0;  // G_ii = 1
    if (threadIdx.x == 0) {
        const int64_t first = (int64_t)blockIdx.x * kBlock;
        part[blockIdx.x] = first < m ? make_double2(1.0, (double)first) : make_double2(-1.0, 0.0);
        if (blockIdx.x == 0) {
            state->rank = 0;
            state->done = 0;
            state->dmax = 1.0;
        }
    }
}

__global__ __launch_bounds__(kBlock) void k_pchol_step(const float4* __restrict__ src4, int64_t m, int64_t ld,
                                                       double inv_two_beta, double* __restrict__ f,
                                                       double* __restrict__ d, const double2* __restrict__ part_in,
                                                       double2* __restrict__ part_out, int nblk, int j, double tol,
                                                       PcholState* __restrict__ state, int* __restrict__ piv) {
    __shared__ double prow[kMaxRank];
    // the pivot: largest remaining diagonal entry (every workgroup derives the same one)
    double dmax = -1.0;
    int p = 0x7fffffff;
    for (int b = threadIdx.x; b < nblk; b += kBlock) {
        const double2 q = part_in[b];
        argmax_merge(dmax, p, q.x, (int)q.y);
    }
    block_argmax(dmax, p);
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const bool active = dmax > tol;
    double dn = -1.0;
    if (active) {
        for (int k = threadIdx.x; k < j; k += kBlock) prow[k] = f[(int64_t)k * ld + p];
        __syncthreads();
        double v = 0.0;
        dn = 0.0;
        if (i < m) {
            const float4 yi = src4[i], yp = src4[p];
            const double dx = (double)yi.x - (double)yp.x, dy = (double)yi.y - (double)yp.y, dz = (double)yi.z - (double)yp.z;
            const double g = exp(-(dx * dx + dy * dy + dz * dz) * inv_two_beta);
            double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
            int k = 0;
            for (; k + 4 <= j; k += 4) {
                s0 = fma(f[(int64_t)k * ld + i], prow[k], s0);
                s1 = fma(f[(int64_t)(k + 1) * ld + i], prow[k + 1], s1);
                s2 = fma(f[(int64_t)(k + 2) * ld + i], prow[k + 2], s2);
                s3 = fma(f[(int64_t)(k + 3) * ld + i], prow[k + 3], s3);
            }
            for (; k < j; ++k) s0 = fma(f[(int64_t)k * ld + i], prow[k], s0);
            const double root = sqrt(dmax);
            if (i == p) {
                v = root;
            } else {
                v = (g - ((s0 + s1) + (s2 + s3))) / root;
                dn = fmax(d[i] - v * v, 0.0);
            }
            d[i] = dn;
        }
        if (i < ld) f[(int64_t)j * ld + i] = v;
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            state->rank = j + 1;
            state->dmax = dmax;
            piv[j] = p;
        }
    } else {  // converged in an earlier step of this batch: keep the maxima alive, change nothing
        if (i < m) dn = d[i];
        if (blockIdx.x == 0 && threadIdx.x == 0) state->done = 1;
    }
    int bi = i < m ? (int)i : 0x7fffffff;
    if (i >= m) dn = -1.0;
    block_argmax(dn, bi);
    if (threadIdx.x == 0) part_out[blockIdx.x] = make_double2(dn, (double)bi);
}

// out[k][0..2] = sum_i F[k][i] x[i][0..2]: one workgroup per factor column (fixed summation order)
__global__ __launch_bounds__(kBlock) void k_lr_ft3(const double* __restrict__ f, int64_t ld, int64_t m,
                                                   const double* __restrict__ x3, double* __restrict__ out) {
    __shared__ double sh[kBlock / 64][3];
    const double* __restrict__ col = f + (int64_t)blockIdx.x * ld;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
#pragma unroll 4
    for (int64_t i = threadIdx.x; i < m; i += kBlock) {
        const double c = col[i];
        a0 = fma(c, x3[i * 3], a0);
        a1 = fma(c, x3[i * 3 + 1], a1);
        a2 = fma(c, x3[i * 3 + 2], a2);
    }
    a0 = wave_sum(a0);
    a1 = wave_sum(a1);
    a2 = wave_sum(a2);
    if ((threadIdx.x & 63) == 0) {
        sh[threadIdx.x >> 6][0] = a0;
        sh[threadIdx.x >> 6][1] = a1;
        sh[threadIdx.x >> 6][2] = a2;
    }
    __syncthreads();
    if (threadIdx.x < 3)
        out[(int64_t)blockIdx.x * 3 + threadIdx.x] =
            (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
}

// out[i][0..2] = sum_k F[k][i] v[k][0..2]
__global__ __launch_bounds__(kBlock) void k_lr_apply(const double* __restrict__ f, int64_t ld, int64_t m, int rank,
                                                     const double* __restrict__ v, double* __restrict__ out3) {
    __shared__ double sv[kMaxRank * 3];
    for (int k = threadIdx.x; k < rank * 3; k += kBlock) sv[k] = v[k];
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= m) return;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    for (int k = 0; k < rank; ++k) {
        const double c = f[(int64_t)k * ld + i];
        a0 = fma(c, sv[3 * k], a0);
        a1 = fma(c, sv[3 * k + 1], a1);
        a2 = fma(c, sv[3 * k + 2], a2);
    }
    out3[i * 3] = a0;
    out3[i * 3 + 1] = a1;
    out3[i * 3 + 2] = a2;
}

inline dim3 grid1(int64_t n) { return dim3((unsigned)prg::ceil_div(n, kBlock)); }

}  // namespace

namespace prg {

// workspace layout inside h->nr_work (doubles): [0, 3M) GW / scratch
// displacement field G W of the current W -> nr_work[0, 3M); the E-step's transform kernel adds it to the source
int nonrigid_displacement(prg_cpd* h, const double** gw_out) {
    PRG_REQUIRE((h->G || h->F) && h->W, PRG_ERR_STATE, "non-rigid transform: G has not been built");
    if (!h->gw_valid) {
        PRG_TRY(nonrigid_gw(h, h->W, h->nr_work));
        h->gw_valid = true;
    }
    *gw_out = h->nr_work;
    return PRG_OK;
}

int lowrank_ft3(prg_cpd* h, const double* x3, double* out) {
    k_lr_ft3<<<(unsigned)h->f_rank, kBlock, 0, h->stream>>>(h->F, h->f_ld, h->M, x3, out);
    PRG_HIP(hipGetLastError());
    return PRG_OK;
}

int lowrank_apply(prg_cpd* h, const double* v, double* out3) {
    k_lr_apply<<<grid1(h->M), kBlock, 0, h->stream>>>(h->F, h->f_ld, h->M, h->f_rank, v, out3);
    PRG_HIP(hipGetLastError());
    return PRG_OK;
}

int nonrigid_gw(prg_cpd* h, const double* w3, double* out3) {
    PRG_REQUIRE(h->G || h->F, PRG_ERR_STATE, "non-rigid: G has not been built");
    if (h->F) {  // G w = F (F^T w)
        double* v = h->nr_work + (size_t)h->M * 3;  // [rank][3] behind the GW block of the workspace
        PRG_TRY(lowrank_ft3(h, w3, v));
        return lowrank_apply(h, v, out3);
    }
    k_gw<<<(unsigned)prg::ceil_div(h->M, kGwRows), kBlock, 0, h->stream>>>(h->G, h->M, w3, out3);
    PRG_HIP(hipGetLastError());
    return PRG_OK;
}

// G = F F^T by pivoted Cholesky (see k_pchol_step).  On success h->F / f_rank are set; when the rank would exceed
// max_rank the factor is dropped and *ok = false (the caller falls back to the dense matrix).
static int build_lowrank_factor(prg_cpd* h, double beta, int max_rank, double tol, bool* ok) {
    *ok = false;
    // rows of the factor are 256 bytes out of step with each other: kernels that walk many columns of F at the same
    // point offset (k_lr_gram's staging, k_lr_ft3, this file's k_pchol_step) would otherwise hit the same memory
    // channels with every row (row stride = a multiple of 8 KB)
    const int64_t m = h->M, mp = round_up(m, kBlock), ld = mp + 32;
    const int nblk = (int)(mp / kBlock);
    int cap = std::min(max_rank, 256);
    double* f = nullptr;
    double* work = nullptr;  // d [ld] | part [2][nblk] double2 | state | piv [kMaxRank]
    const size_t work_bytes = (size_t)ld * 8 + (size_t)nblk * 2 * 16 + 64 + (size_t)kMaxRank * 4;
    PRG_HIP(hipMalloc((void**)&f, (size_t)cap * ld * sizeof(double)));
    if (hipMalloc((void**)&work, work_bytes) != hipSuccess) {
        (void)hipFree(f);
        PRG_REQUIRE(false, PRG_ERR_HIP, "non-rigid: out of device memory for the kernel factor");
    }
    double* d = work;
    double2* part = reinterpret_cast<double2*>(work + ld);
    PcholState* state = reinterpret_cast<PcholState*>(part + 2 * (size_t)nblk);
    int* piv = reinterpret_cast<int*>(reinterpret_cast<char*>(state) + 64);
    auto fail = [&](const char* what) {
        (void)hipFree(f);
        (void)hipFree(work);
        prg::set_error("non-rigid kernel factor: %s", what);
        return PRG_ERR_HIP;
    };
    k_pchol_init<<<nblk, kBlock, 0, h->stream>>>(m, mp, d, part, state);
    PcholState host = {0, 0, 1.0};
    int j = 0;
    const int limit = (int)std::min<int64_t>(max_rank, m);
    cap = std::min(cap, limit);
    while (true) {
        if (j == cap) {  // grow the factor
            if (cap >= limit) break;  // not low rank enough
            const int ncap = std::min(limit, cap * 2);
            double* nf = nullptr;
            if (hipMalloc((void**)&nf, (size_t)ncap * ld * sizeof(double)) != hipSuccess) return fail("out of device memory");
            if (hipMemcpyAsync(nf, f, (size_t)cap * ld * sizeof(double), hipMemcpyDeviceToDevice, h->stream) != hipSuccess ||
                hipStreamSynchronize(h->stream) != hipSuccess) {
                (void)hipFree(nf);
                return fail("copy failed");
            }
            (void)hipFree(f);
            f = nf;
            cap = ncap;
        }
        const int batch = std::min(32, cap - j);
        for (int b = 0; b < batch; ++b, ++j)
            k_pchol_step<<<nblk, kBlock, 0, h->stream>>>(h->src4, m, ld, 1.0 / (2.0 * beta), f, d, part + (size_t)(j & 1) * nblk,
                                                         part + (size_t)((j + 1) & 1) * nblk, nblk, j, tol, state, piv);
        if (hipMemcpyAsync(&host, state, sizeof(PcholState), hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
            hipStreamSynchronize(h->stream) != hipSuccess || hipGetLastError() != hipSuccess)
            return fail("device error");
        if (host.done) break;  // a step of this batch found the remaining diagonal below tol
    }
    if (!host.done) {  // out of rank: converged only if the very last step reached the tolerance
        std::vector<double2> hp((size_t)nblk);
        if (hipMemcpyAsync(hp.data(), part + (size_t)(j & 1) * nblk, (size_t)nblk * sizeof(double2), hipMemcpyDeviceToHost,
                           h->stream) != hipSuccess ||
            hipStreamSynchronize(h->stream) != hipSuccess)
            return fail("device error");
        double mx = -1.0;
        for (const double2& q : hp) mx = std::max(mx, q.x);
        host.done = mx <= tol;
    }
    (void)hipFree(work);
    if (!host.done) {
        (void)hipFree(f);
        return PRG_OK;
    }
    h->F = f;
    h->f_ld = ld;
    h->f_rank = host.rank;
    h->f_cap = cap;
    *ok = true;
    return PRG_OK;
}

int build_kernel_matrix(prg_cpd* h, int kind, double param) {
    nonrigid_free(h);
    const int64_t m = h->M;
    PRG_HIP(hipMalloc((void**)&h->W, (size_t)m * 3 * sizeof(double)));
    h->nr_work_bytes = ((size_t)m * 16 + (size_t)kMaxRank * 3) * sizeof(double);
    PRG_HIP(hipMalloc((void**)&h->nr_work, h->nr_work_bytes));
    PRG_HIP(hipMemsetAsync(h->W, 0, (size_t)m * 3 * sizeof(double), h->stream));
    if (kind == 0 && h->nr_solver == 1) {
        int max_rank = h->nr_max_rank > 0 ? h->nr_max_rank : (int)std::min<int64_t>(kMaxRank, m / 2);
        max_rank = std::min(max_rank, kMaxRank);
        bool ok = false;
        if (max_rank >= 1) PRG_TRY(build_lowrank_factor(h, param, max_rank, h->nr_tol, &ok));
        if (ok) return PRG_OK;
    }
    PRG_HIP(hipMalloc((void**)&h->G, (size_t)m * m * sizeof(float)));
    dim3 grid((unsigned)prg::ceil_div(m, kBlock), (unsigned)prg::ceil_div(m, 16));
    if (kind == 0)
        k_build_g<0><<<grid, kBlock, 0, h->stream>>>(h->src4, m, (float)(2.0 * param), h->G);
    else
        k_build_g<1><<<grid, kBlock, 0, h->stream>>>(h->src4, m, (float)param, h->G);
    PRG_HIP(hipGetLastError());
    return PRG_OK;
}

int nonrigid_free(prg_cpd* h) {
    for (hipEvent_t e : h->nr_events) (void)hipEventDestroy(e);
    h->nr_events.clear();
    if (h->nr_stream2) (void)hipStreamDestroy(h->nr_stream2);
    h->nr_stream2 = nullptr;
    if (h->nr_prior) (void)hipFree(h->nr_prior);
    h->nr_prior = nullptr;
    h->nr_alpha = 0.0;
    if (h->nr_solve) (void)hipFree(h->nr_solve);
    h->nr_solve = nullptr;
    h->nr_solve_bytes = 0;
    h->nr_info = nullptr;  // the sticky pivot flag lives at the end of nr_solve: it goes with it
    if (h->G) (void)hipFree(h->G);
    if (h->F) (void)hipFree(h->F);
    h->F = nullptr;
    h->f_rank = h->f_cap = 0;
    h->f_ld = 0;
    if (h->W) (void)hipFree(h->W);
    if (h->nr_work) (void)hipFree(h->nr_work);
    h->G = nullptr;
    h->W = nullptr;
    h->nr_work = nullptr;
    h->nr_work_bytes = 0;
    h->nonrigid = false;
    h->bcpd = false;
    h->gw_valid = false;
    return PRG_OK;
}

}  // namespace prg

extern "C" {

int prg_cpd_nonrigid_build_g(prg_cpd* h, double beta) {
    PRG_REQUIRE(h && h->have_source, PRG_ERR_STATE, "prg_cpd_nonrigid_build_g: source not set");
    PRG_REQUIRE(beta > 0.0, PRG_ERR_INVALID, "prg_cpd_nonrigid_build_g: beta must be > 0 (got %g)", beta);
    prg::DeviceGuard g(h->device);
    PRG_TRY(prg::build_kernel_matrix(h, 0, beta));
    h->beta = beta;
    h->nonrigid = true;
    return PRG_OK;
}

int prg_cpd_nonrigid_set_solver(prg_cpd* h, int mode, int max_rank, double tol) {
    PRG_REQUIRE(h, PRG_ERR_INVALID, "prg_cpd_nonrigid_set_solver: NULL plan");
    PRG_REQUIRE((mode == 0 || mode == 1) && max_rank >= 0 && tol >= 0.0, PRG_ERR_INVALID,
                "prg_cpd_nonrigid_set_solver: mode must be 0 or 1, max_rank and tol non-negative");
    h->nr_solver = mode;
    h->nr_max_rank = max_rank;
    if (tol > 0.0) h->nr_tol = tol;
    return PRG_OK;
}

int prg_cpd_nonrigid_rank(prg_cpd* h, int* rank) {
    PRG_REQUIRE(h && rank && (h->G || h->F), PRG_ERR_STATE, "prg_cpd_nonrigid_rank: G has not been built");
    *rank = h->F ? h->f_rank : 0;
    return PRG_OK;
}

int prg_cpd_bcpd_build_g(prg_cpd* h, double c) {
    PRG_REQUIRE(h && h->have_source, PRG_ERR_STATE, "prg_cpd_bcpd_build_g: source not set");
    PRG_REQUIRE(c > 0.0, PRG_ERR_INVALID, "prg_cpd_bcpd_build_g: c must be > 0 (got %g)", c);
    prg::DeviceGuard g(h->device);
    PRG_TRY(prg::build_kernel_matrix(h, 1, c));
    h->beta = c;
    h->bcpd = true;  // the linear transform kernel adds the displacement W (= v_hat) before s R . + t
    return PRG_OK;
}

int prg_cpd_nonrigid_get_g(prg_cpd* h, float* g_hd) {
    PRG_REQUIRE(h && (h->G || h->F) && !h->bcpd && g_hd, PRG_ERR_STATE, "prg_cpd_nonrigid_get_g: G has not been built");
    prg::DeviceGuard g(h->device);
    const size_t bytes = (size_t)h->M * h->M * sizeof(float);
    if (h->G && !h->perm_src) {
        PRG_HIP(hipMemcpyAsync(g_hd, h->G, bytes, hipMemcpyDefault, h->stream));
        PRG_HIP(hipStreamSynchronize(h->stream));
        return PRG_OK;
    }
    // the plan keeps the factor only, or the matrix of the SORTED source: evaluate the float32 matrix of the caller's
    // order on the spot (entry (i, j) depends on the two points alone, so this is the same matrix)
    float* tmp = nullptr;
    float4* pts = nullptr;
    PRG_HIP(hipMalloc((void**)&tmp, bytes));
    if (hipMalloc((void**)&pts, (size_t)h->M * sizeof(float4)) != hipSuccess) {
        (void)hipFree(tmp);
        PRG_REQUIRE(false, PRG_ERR_HIP, "prg_cpd_nonrigid_get_g: out of device memory");
    }
    k_unsort4<<<grid1(h->M), kBlock, 0, h->stream>>>(h->src4, h->M, h->perm_src, pts);
    dim3 grid((unsigned)prg::ceil_div(h->M, kBlock), (unsigned)prg::ceil_div(h->M, 16));
    k_build_g<0><<<grid, kBlock, 0, h->stream>>>(pts, h->M, (float)(2.0 * h->beta), tmp);
    hipError_t e = hipMemcpyAsync(g_hd, tmp, bytes, hipMemcpyDefault, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    (void)hipFree(tmp);
    (void)hipFree(pts);
    PRG_HIP(e);
    return PRG_OK;
}

int prg_cpd_nonrigid_set_w(prg_cpd* h, const double* w_hd) {
    PRG_REQUIRE(h && h->W && w_hd, PRG_ERR_STATE, "prg_cpd_nonrigid_set_w: G has not been built");
    prg::DeviceGuard g(h->device);
    PRG_TRY(prg::ensure_stage(h, (size_t)h->M * h->D * sizeof(double)));
    PRG_HIP(hipMemcpyAsync(h->stage, w_hd, (size_t)h->M * h->D * sizeof(double), hipMemcpyDefault, h->stream));
    k_pack_w<<<grid1(h->M), kBlock, 0, h->stream>>>((const double*)h->stage, h->M, h->D, h->W, 1, h->perm_src);
    h->gw_valid = false;
    PRG_HIP(hipGetLastError());
    PRG_HIP(hipStreamSynchronize(h->stream));
    return PRG_OK;
}

int prg_cpd_nonrigid_get_w(prg_cpd* h, double* w_hd) {
    PRG_REQUIRE(h && h->W && w_hd, PRG_ERR_STATE, "prg_cpd_nonrigid_get_w: G has not been built");
    prg::DeviceGuard g(h->device);
    PRG_TRY(prg::ensure_stage(h, (size_t)h->M * h->D * sizeof(double)));
    k_pack_w<<<grid1(h->M), kBlock, 0, h->stream>>>((const double*)h->stage, h->M, h->D, h->W, 0, h->perm_src);
    PRG_HIP(hipGetLastError());
    PRG_HIP(hipMemcpyAsync(w_hd, h->stage, (size_t)h->M * h->D * sizeof(double), hipMemcpyDefault, h->stream));
    PRG_HIP(hipStreamSynchronize(h->stream));
    return PRG_OK;
}

int prg_cpd_nonrigid_set_priors(prg_cpd* h, const double* p1_tilde_hd, const double* px_tilde_hd, double alpha) {
    PRG_REQUIRE(h && (h->G || h->F), PRG_ERR_STATE, "prg_cpd_nonrigid_set_priors: G has not been built");
    prg::DeviceGuard g(h->device);
    if (!p1_tilde_hd || !px_tilde_hd) {  // clear
        h->nr_alpha = 0.0;
        return PRG_OK;
    }
    PRG_REQUIRE(alpha > 0.0, PRG_ERR_INVALID, "prg_cpd_nonrigid_set_priors: alpha must be > 0 (got %g)", alpha);
    const int64_t m = h->M;
    if (!h->nr_prior) PRG_HIP(hipMalloc((void**)&h->nr_prior, (size_t)m * 4 * sizeof(double)));
    PRG_TRY(prg::ensure_stage(h, (size_t)m * (h->D + 1) * sizeof(double)));
    double* st_p1 = (double*)h->stage;
    double* st_px = st_p1 + m;
    PRG_HIP(hipMemcpyAsync(st_p1, p1_tilde_hd, (size_t)m * sizeof(double), hipMemcpyDefault, h->stream));
    PRG_HIP(hipMemcpyAsync(st_px, px_tilde_hd, (size_t)m * h->D * sizeof(double), hipMemcpyDefault, h->stream));
    k_unpack_planes<<<grid1(m), kBlock, 0, h->stream>>>(st_p1, st_px, m, h->D, h->nr_prior, h->perm_src);
    PRG_HIP(hipGetLastError());
    PRG_HIP(hipStreamSynchronize(h->stream));
    h->nr_alpha = alpha;
    return PRG_OK;
}

int prg_cpd_rowacc_ptr(prg_cpd* h, double** rowacc_dev, int64_t* count) {
    PRG_REQUIRE(h && h->have_source && rowacc_dev && count, PRG_ERR_STATE, "prg_cpd_rowacc_ptr: source not set");
    *rowacc_dev = h->rowacc;
    *count = 4 * h->Mcap;
    return PRG_OK;
}

}  // extern "C"

namespace {
__global__ __launch_bounds__(kBlock) void k_apply_out(const float4* __restrict__ src4, const double* __restrict__ gw,
                                                      int64_t m, int dim, double* __restrict__ out,
                                                      const int* __restrict__ perm) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= m) return;
    const float4 y = src4[i];
    const double yy[3] = {y.x, y.y, y.z};
    const int64_t j = perm ? perm[i] : i;
    for (int k = 0; k < dim; ++k) out[j * dim + k] = yy[k] + gw[i * 3 + k];
}
}  // namespace

extern "C" int prg_cpd_nonrigid_apply(prg_cpd* h, double* t_hd) {
    PRG_REQUIRE(h && (h->G || h->F) && h->W && t_hd, PRG_ERR_STATE, "prg_cpd_nonrigid_apply: G has not been built");
    prg::DeviceGuard g(h->device);
    double* gw = h->nr_work;
    PRG_TRY(prg::nonrigid_gw(h, h->W, gw));
    PRG_TRY(prg::ensure_stage(h, (size_t)h->M * h->D * sizeof(double)));
    k_apply_out<<<grid1(h->M), kBlock, 0, h->stream>>>(h->src4, gw, h->M, h->D, (double*)h->stage, h->perm_src);
    PRG_HIP(hipGetLastError());
    PRG_HIP(hipMemcpyAsync(t_hd, h->stage, (size_t)h->M * h->D * sizeof(double), hipMemcpyDefault, h->stream));
    PRG_HIP(hipStreamSynchronize(h->stream));
    return PRG_OK;
}
