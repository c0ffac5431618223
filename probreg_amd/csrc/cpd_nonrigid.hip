// Non-rigid CPD pieces for MI355X (gfx950).
//
// Reference behaviour (neka-nat/probreg v0.3.7):
//   G build        probreg/transformation.py:91-99 -> probreg/cc/math_utils.cc:17-19
//   T(Y) = Y + G W probreg/transformation.py:101-102
//   M-step         probreg/cpd.py:284-303
#include <math.h>

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

// z4 = y + GW (float32 result of an fp64 sum), pads to sentinel
__global__ __launch_bounds__(kBlock) void k_add_disp(const float4* __restrict__ src4, const double* __restrict__ gw,
                                                     int64_t m, int64_t cap, float4* __restrict__ z4) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= cap) return;
    float4 o;
    if (i < m) {
        const float4 y = src4[i];
        o.x = (float)((double)y.x + gw[i * 3]);
        o.y = (float)((double)y.y + gw[i * 3 + 1]);
        o.z = (float)((double)y.z + gw[i * 3 + 2]);
        o.w = 0.f;
    } else {
        o.x = o.y = o.z = prg::kSrcPad;
        o.w = 0.f;
    }
    z4[i] = o;
}

__global__ __launch_bounds__(kBlock) void k_pack_w(const double* __restrict__ in, int64_t m, int dim,
                                                   double* __restrict__ w3, int to_w3) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= m) return;
    if (to_w3) {
        for (int k = 0; k < 3; ++k) w3[i * 3 + k] = k < dim ? in[i * dim + k] : 0.0;
    } else {
        double* out = const_cast<double*>(in);
        for (int k = 0; k < dim; ++k) out[i * dim + k] = w3[i * 3 + k];
    }
}

// [m][dim] row-major -> 3 planes of m (missing dims zero)
__global__ __launch_bounds__(kBlock) void k_unpack_planes(const double* __restrict__ in, int64_t m, int dim,
                                                          double* __restrict__ planes) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= m) return;
    for (int k = 0; k < 3; ++k) planes[(int64_t)k * m + i] = k < dim ? in[i * dim + k] : 0.0;
}

inline dim3 grid1(int64_t n) { return dim3((unsigned)prg::ceil_div(n, kBlock)); }

}  // namespace

namespace prg {

// workspace layout inside h->nr_work (doubles): [0, 3M) GW / scratch
int nonrigid_transform(prg_cpd* h) {
    PRG_REQUIRE(h->G && h->W, PRG_ERR_STATE, "non-rigid transform: G has not been built");
    double* gw = h->nr_work;
    k_gw<<<(unsigned)prg::ceil_div(h->M, kGwRows), kBlock, 0, h->stream>>>(h->G, h->M, h->W, gw);
    k_add_disp<<<grid1(h->Mcap), kBlock, 0, h->stream>>>(h->src4, gw, h->M, h->Mcap, h->z4);
    PRG_HIP(hipGetLastError());
    return PRG_OK;
}

int nonrigid_gw(prg_cpd* h, const double* w3, double* out3) {
    PRG_REQUIRE(h->G, PRG_ERR_STATE, "non-rigid: G has not been built");
    k_gw<<<(unsigned)prg::ceil_div(h->M, kGwRows), kBlock, 0, h->stream>>>(h->G, h->M, w3, out3);
    PRG_HIP(hipGetLastError());
    return PRG_OK;
}

int build_kernel_matrix(prg_cpd* h, int kind, double param) {
    nonrigid_free(h);
    const int64_t m = h->M;
    PRG_HIP(hipMalloc((void**)&h->G, (size_t)m * m * sizeof(float)));
    PRG_HIP(hipMalloc((void**)&h->W, (size_t)m * 3 * sizeof(double)));
    h->nr_work_bytes = (size_t)m * 16 * sizeof(double);
    PRG_HIP(hipMalloc((void**)&h->nr_work, h->nr_work_bytes));
    PRG_HIP(hipMemsetAsync(h->W, 0, (size_t)m * 3 * sizeof(double), h->stream));
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
    if (h->G) (void)hipFree(h->G);
    if (h->W) (void)hipFree(h->W);
    if (h->nr_work) (void)hipFree(h->nr_work);
    h->G = nullptr;
    h->W = nullptr;
    h->nr_work = nullptr;
    h->nr_work_bytes = 0;
    h->nonrigid = false;
    h->bcpd = false;
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
    PRG_REQUIRE(h && h->G && g_hd, PRG_ERR_STATE, "prg_cpd_nonrigid_get_g: G has not been built");
    prg::DeviceGuard g(h->device);
    PRG_HIP(hipMemcpyAsync(g_hd, h->G, (size_t)h->M * h->M * sizeof(float), hipMemcpyDefault, h->stream));
    PRG_HIP(hipStreamSynchronize(h->stream));
    return PRG_OK;
}

int prg_cpd_nonrigid_set_w(prg_cpd* h, const double* w_hd) {
    PRG_REQUIRE(h && h->W && w_hd, PRG_ERR_STATE, "prg_cpd_nonrigid_set_w: G has not been built");
    prg::DeviceGuard g(h->device);
    PRG_TRY(prg::ensure_stage(h, (size_t)h->M * h->D * sizeof(double)));
    PRG_HIP(hipMemcpyAsync(h->stage, w_hd, (size_t)h->M * h->D * sizeof(double), hipMemcpyDefault, h->stream));
    k_pack_w<<<grid1(h->M), kBlock, 0, h->stream>>>((const double*)h->stage, h->M, h->D, h->W, 1);
    PRG_HIP(hipGetLastError());
    PRG_HIP(hipStreamSynchronize(h->stream));
    return PRG_OK;
}

int prg_cpd_nonrigid_get_w(prg_cpd* h, double* w_hd) {
    PRG_REQUIRE(h && h->W && w_hd, PRG_ERR_STATE, "prg_cpd_nonrigid_get_w: G has not been built");
    prg::DeviceGuard g(h->device);
    PRG_TRY(prg::ensure_stage(h, (size_t)h->M * h->D * sizeof(double)));
    k_pack_w<<<grid1(h->M), kBlock, 0, h->stream>>>((const double*)h->stage, h->M, h->D, h->W, 0);
    PRG_HIP(hipGetLastError());
    PRG_HIP(hipMemcpyAsync(w_hd, h->stage, (size_t)h->M * h->D * sizeof(double), hipMemcpyDefault, h->stream));
    PRG_HIP(hipStreamSynchronize(h->stream));
    return PRG_OK;
}

int prg_cpd_nonrigid_set_priors(prg_cpd* h, const double* p1_tilde_hd, const double* px_tilde_hd, double alpha) {
    PRG_REQUIRE(h && h->G, PRG_ERR_STATE, "prg_cpd_nonrigid_set_priors: G has not been built");
    prg::DeviceGuard g(h->device);
    if (!p1_tilde_hd || !px_tilde_hd) {  // clear
        h->nr_alpha = 0.0;
        return PRG_OK;
    }
    PRG_REQUIRE(alpha > 0.0, PRG_ERR_INVALID, "prg_cpd_nonrigid_set_priors: alpha must be > 0 (got %g)", alpha);
    const int64_t m = h->M;
    if (!h->nr_prior) PRG_HIP(hipMalloc((void**)&h->nr_prior, (size_t)m * 4 * sizeof(double)));
    PRG_TRY(prg::ensure_stage(h, (size_t)m * h->D * sizeof(double)));
    PRG_HIP(hipMemcpyAsync(h->nr_prior, p1_tilde_hd, (size_t)m * sizeof(double), hipMemcpyDefault, h->stream));
    PRG_HIP(hipMemcpyAsync(h->stage, px_tilde_hd, (size_t)m * h->D * sizeof(double), hipMemcpyDefault, h->stream));
    k_unpack_planes<<<grid1(m), kBlock, 0, h->stream>>>((const double*)h->stage, m, h->D, h->nr_prior + m);
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
                                                      int64_t m, int dim, double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= m) return;
    const float4 y = src4[i];
    const double yy[3] = {y.x, y.y, y.z};
    for (int k = 0; k < dim; ++k) out[i * dim + k] = yy[k] + gw[i * 3 + k];
}
}  // namespace

extern "C" int prg_cpd_nonrigid_apply(prg_cpd* h, double* t_hd) {
    PRG_REQUIRE(h && h->G && h->W && t_hd, PRG_ERR_STATE, "prg_cpd_nonrigid_apply: G has not been built");
    prg::DeviceGuard g(h->device);
    double* gw = h->nr_work;
    k_gw<<<(unsigned)prg::ceil_div(h->M, kGwRows), kBlock, 0, h->stream>>>(h->G, h->M, h->W, gw);
    PRG_TRY(prg::ensure_stage(h, (size_t)h->M * h->D * sizeof(double)));
    k_apply_out<<<grid1(h->M), kBlock, 0, h->stream>>>(h->src4, gw, h->M, h->D, (double*)h->stage);
    PRG_HIP(hipGetLastError());
    PRG_HIP(hipMemcpyAsync(t_hd, h->stage, (size_t)h->M * h->D * sizeof(double), hipMemcpyDefault, h->stream));
    PRG_HIP(hipStreamSynchronize(h->stream));
    return PRG_OK;
}
