// Error plumbing and the small stand-alone kernels of libprobreg_hip.so:
//   prg_squared_kernel_sum     (probreg/math_utils.py:28-29 -> cc/math_utils.cc:5-15)
//   prg_rbf_kernel             (probreg/math_utils.py:36-37 -> cc/math_utils.cc:17-19)
//   prg_gauss_transform_direct (probreg/gauss_transform.py:10-25, 46-60)
#include <math.h>

#include <algorithm>
#include <stdarg.h>

#include "prg_common.h"

namespace prg {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace prg

namespace {
constexpr int kBlock = 256;

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// partial sums (sum x, sum y, sum z, sum |p|^2) in fp64 over a row-major n x dim float cloud
__global__ __launch_bounds__(kBlock) void k_sums_rowmajor(const float* __restrict__ p, int64_t n, int dim,
                                                          double* __restrict__ part) {
    __shared__ double sh[4][4];
    double a[4] = {0, 0, 0, 0};
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        double x = p[i * dim], y = p[i * dim + 1], z = dim > 2 ? p[i * dim + 2] : 0.0;
        a[0] += x; a[1] += y; a[2] += z;
        a[3] += x * x + y * y + z * z;
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        double s = wave_sum(a[c]);
        if (lane == 0) sh[wv][c] = s;
    }
    __syncthreads();
    if (threadIdx.x < 4)
        part[(int64_t)blockIdx.x * 4 + threadIdx.x] = sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] +
                                                     sh[3][threadIdx.x];
}

// any dimension (feature clouds, d <= 64): out[0..dim) += column sums, out[dim] += sum of squared norms (fp64 atomics)
__global__ __launch_bounds__(kBlock) void k_sums_generic(const float* __restrict__ p, int64_t n, int dim,
                                                         double* __restrict__ out) {
    double sq = 0.0;
    for (int k = 0; k < dim; ++k) {
        double a = 0.0;
        for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
            const double v = p[i * dim + k];
            a += v;
            sq += v * v;
        }
        a = wave_sum(a);
        if ((threadIdx.x & 63) == 0 && a != 0.0) atomicAdd(&out[k], a);
    }
    sq = wave_sum(sq);
    if ((threadIdx.x & 63) == 0) atomicAdd(&out[dim], sq);
}
__global__ void k_sks_final_generic(const double* __restrict__ sx, const double* __restrict__ sy, double m, double n,
                                    int dim, double* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double cross = 0.0;
    for (int k = 0; k < dim; ++k) cross += sx[k] * sy[k];
    out[0] = (n * sx[dim] + m * sy[dim] - 2.0 * cross) / (m * dim * n);
}

__global__ void k_sks_final(const double* __restrict__ px, int nbx, const double* __restrict__ py, int nby, double m,
                            double n, int dim, double* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double sx[4] = {0, 0, 0, 0}, sy[4] = {0, 0, 0, 0};
    for (int b = 0; b < nbx; ++b)
        for (int c = 0; c < 4; ++c) sx[c] += px[b * 4 + c];
    for (int b = 0; b < nby; ++b)
        for (int c = 0; c < 4; ++c) sy[c] += py[b * 4 + c];
    const double cross = sx[0] * sy[0] + sx[1] * sy[1] + sx[2] * sy[2];
    out[0] = (n * sx[3] + m * sy[3] - 2.0 * cross) / (m * dim * n);
}

// K[i][j] = exp(-|x_i - y_j|^2 / (2 beta)), float32.  The squared distance is evaluated in
// float32 without fused multiply-add, like the reference's Eigen expression
// (cc/math_utils.cc:9-10), and the exponential in fp64 rounded once to float32.
__global__ __launch_bounds__(kBlock) void k_rbf(const float* __restrict__ x, int64_t m, const float* __restrict__ y,
                                                int64_t n, int dim, float two_beta, float* __restrict__ out) {
    const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t i0 = (int64_t)blockIdx.y * 16;
    if (j >= n) return;
    const float yx = y[j * dim], yy = y[j * dim + 1], yz = dim > 2 ? y[j * dim + 2] : 0.f;
    for (int64_t i = i0; i < i0 + 16 && i < m; ++i) {
        const float dx = __fsub_rn(x[i * dim], yx), dy = __fsub_rn(x[i * dim + 1], yy),
                    dz = dim > 2 ? __fsub_rn(x[i * dim + 2], yz) : 0.f;
        const float d2 = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
        const float arg = __fdiv_rn(-d2, two_beta);
        out[i * n + j] = (float)exp((double)arg);
    }
}

// out[c][i] = sum_j w[c][j] exp2(kk |t_i - s_j|^2) for C weight rows in ONE sweep (one exponential per pair whatever C is).
// The reference's direct path works on float64 arrays (gauss_transform.py:10-16), and a narrow kernel (h ~ 1e-2 of the cloud)
// is sensitive to the coordinates' rounding: the differences are formed in fp64 (exact to 1e-16 of the coordinates) and only
// then rounded to fp32 - squared distance and exponential in fp32 (relative error of a term ~ (d/h)^2 x 1.2e-7), sums and
// weights in fp64.  A lane owns a target point; the source (x, y, z as doubles) and the weight rows are read through
// wave-uniform addresses (scalar loads), four points per trip.
template <int C>
__global__ __launch_bounds__(kBlock) void k_gauss_direct(const double* __restrict__ src3, int64_t s_cap,
                                                         const double* __restrict__ wrows, const double* __restrict__ tgt,
                                                         int64_t t, int dim, float kk, double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    double tx = 0.0, ty = 0.0, tz = 0.0;
    if (i < t) {
        tx = tgt[i * dim];
        ty = tgt[i * dim + 1];
        tz = dim > 2 ? tgt[i * dim + 2] : 0.0;
    }
    double acc[C];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = 0.0;
    for (int64_t j0 = 0; j0 < s_cap; j0 += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t j = j0 + u;
            const float dx = (float)(tx - src3[3 * j]), dy = (float)(ty - src3[3 * j + 1]), dz = (float)(tz - src3[3 * j + 2]);
            float d = dx * dx;
            d = fmaf(dy, dy, d);
            d = fmaf(dz, dz, d);
            const double e = (double)__builtin_amdgcn_exp2f(kk * d);
#pragma unroll
            for (int c = 0; c < C; ++c) acc[c] = fma(wrows[(int64_t)c * s_cap + j], e, acc[c]);
        }
    }
    if (i < t) {
#pragma unroll
        for (int c = 0; c < C; ++c) out[(int64_t)c * t + i] = acc[c];
    }
}

// source rows (dim doubles each) -> [cap][3] doubles, pads far away; weight rows -> [rows][cap], pads 0
__global__ __launch_bounds__(kBlock) void k_pack_gauss(const double* __restrict__ s, const double* __restrict__ w, int64_t n,
                                                       int dim, int rows, int64_t cap, double* __restrict__ src3,
                                                       double* __restrict__ wrows) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= cap) return;
    const bool in = i < n;
    src3[3 * i] = in ? s[i * dim] : (double)prg::kSrcPad;
    src3[3 * i + 1] = in ? s[i * dim + 1] : (double)prg::kSrcPad;
    src3[3 * i + 2] = in ? (dim > 2 ? s[i * dim + 2] : 0.0) : (double)prg::kSrcPad;
    for (int c = 0; c < rows; ++c) wrows[(int64_t)c * cap + i] = in ? w[(int64_t)c * n + i] : 0.0;
}

// points (+ optional per-point weight in .w) -> float4, pads far away
__global__ __launch_bounds__(kBlock) void k_pack_weighted(const float* __restrict__ s, const double* __restrict__ w,
                                                          int64_t n, int dim, int64_t cap,
                                                          float4* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= cap) return;
    float4 v;
    if (i < n) {
        v.x = s[i * dim];
        v.y = s[i * dim + 1];
        v.z = dim > 2 ? s[i * dim + 2] : 0.f;
        v.w = w ? (float)w[i] : 0.f;
    } else {
        v.x = v.y = v.z = prg::kSrcPad;
        v.w = 0.f;
    }
    out[i] = v;
}

// K[i][j] = 1 / sqrt(|x_i - y_j|^2 + c), all float32 (cc/math_utils.cc:32-34)
__global__ __launch_bounds__(kBlock) void k_imq(const float* __restrict__ x, int64_t m, const float* __restrict__ y,
                                                int64_t n, int dim, float c, float* __restrict__ out) {
    const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t i0 = (int64_t)blockIdx.y * 16;
    if (j >= n) return;
    const float yx = y[j * dim], yy = y[j * dim + 1], yz = dim > 2 ? y[j * dim + 2] : 0.f;
    for (int64_t i = i0; i < i0 + 16 && i < m; ++i) {
        const float dx = __fsub_rn(x[i * dim], yx), dy = __fsub_rn(x[i * dim + 1], yy),
                    dz = dim > 2 ? __fsub_rn(x[i * dim + 2], yz) : 0.f;
        const float d2 = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
        out[i * n + j] = __fdiv_rn(1.f, __fsqrt_rn(__fadd_rn(d2, c)));
    }
}

// Brute-force nearest neighbour: dmin[i] = min_j |a_i - b_j|^2 over the b segment of blockIdx.y, merged across
// segments with atomicMin on the float bits (non-negative floats order like unsigned integers).
__global__ __launch_bounds__(kBlock) void k_nn_min(const float* __restrict__ a, int64_t m, int dim,
                                                   const float4* __restrict__ b4, int seg_len,
                                                   unsigned* __restrict__ dmin) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    float ax = 0.f, ay = 0.f, az = 0.f;
    if (i < m) {
        ax = a[i * dim];
        ay = a[i * dim + 1];
        az = dim > 2 ? a[i * dim + 2] : 0.f;
    }
    const float4* __restrict__ bp = b4 + (int64_t)blockIdx.y * seg_len;
    float best = INFINITY;
    for (int j = 0; j < seg_len; j += 4) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float4 q = bp[j + c];  // wave-uniform address: scalar loads
            const float dx = ax - q.x, dy = ay - q.y, dz = az - q.z;
            best = fminf(best, fmaf(dz, dz, fmaf(dy, dy, dx * dx)));
        }
    }
    if (i < m) atomicMin(dmin + i, __float_as_uint(best));
}

__global__ __launch_bounds__(kBlock) void k_nn_fill(unsigned* __restrict__ dmin, int64_t m) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < m) dmin[i] = 0x7f800000u;
}

// out[0] = mean_i sqrt(dmin[i]) ; one workgroup
__global__ __launch_bounds__(kBlock) void k_nn_mean(const unsigned* __restrict__ dmin, int64_t m,
                                                    double* __restrict__ out) {
    __shared__ double sh[kBlock];
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < m; i += kBlock) acc += sqrt((double)__uint_as_float(dmin[i]));
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int o = kBlock / 2; o > 0; o >>= 1) {
        if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = sh[0] / (double)m;
}

struct TmpBuf {
    void* p = nullptr;
    ~TmpBuf() {
        if (p) (void)hipFree(p);
    }
};

}  // namespace

extern "C" {

const char* prg_last_error(void) { return prg::g_err; }

int prg_version(void) { return 100; }

int prg_device_count(int* count) {
    PRG_REQUIRE(count != nullptr, PRG_ERR_INVALID, "prg_device_count: NULL argument");
    *count = 0;
    hipError_t e = hipGetDeviceCount(count);
    if (e != hipSuccess) {
        *count = 0;
        prg::set_error("hipGetDeviceCount failed: %s", hipGetErrorString(e));
        return PRG_ERR_HIP;
    }
    return PRG_OK;
}

int prg_squared_kernel_sum(int device, void* hip_stream, const float* x_hd, int64_t m, const float* y_hd, int64_t n,
                           int dim, double* out_host) {
    PRG_REQUIRE(x_hd && y_hd && out_host, PRG_ERR_INVALID, "prg_squared_kernel_sum: NULL argument");
    PRG_REQUIRE(m > 0 && n > 0 && dim >= 1 && dim <= 64, PRG_ERR_INVALID,
                "prg_squared_kernel_sum: need m, n > 0 and dim in [1, 64]");
    prg::DeviceGuard g(device);
    PRG_REQUIRE(g.ok, PRG_ERR_HIP, "prg_squared_kernel_sum: hipSetDevice(%d) failed", device);
    hipStream_t st = (hipStream_t)hip_stream;
    if (dim != 2 && dim != 3) {  // feature clouds (FilterReg with a feature_fn, filterreg.py:126-128)
        TmpBuf fx, fy, fs;
        PRG_HIP(hipMalloc(&fx.p, (size_t)m * dim * sizeof(float)));
        PRG_HIP(hipMalloc(&fy.p, (size_t)n * dim * sizeof(float)));
        PRG_HIP(hipMalloc(&fs.p, (size_t)(2 * (dim + 1) + 1) * sizeof(double)));
        PRG_HIP(hipMemcpyAsync(fx.p, x_hd, (size_t)m * dim * sizeof(float), hipMemcpyDefault, st));
        PRG_HIP(hipMemcpyAsync(fy.p, y_hd, (size_t)n * dim * sizeof(float), hipMemcpyDefault, st));
        PRG_HIP(hipMemsetAsync(fs.p, 0, (size_t)(2 * (dim + 1) + 1) * sizeof(double), st));
        double* sx = (double*)fs.p;
        double* sy = sx + dim + 1;
        k_sums_generic<<<(unsigned)std::min<int64_t>(prg::ceil_div(m, kBlock), 512), kBlock, 0, st>>>((const float*)fx.p, m, dim, sx);
        k_sums_generic<<<(unsigned)std::min<int64_t>(prg::ceil_div(n, kBlock), 512), kBlock, 0, st>>>((const float*)fy.p, n, dim, sy);
        k_sks_final_generic<<<1, 64, 0, st>>>(sx, sy, (double)m, (double)n, dim, sy + dim + 1);
        PRG_HIP(hipGetLastError());
        PRG_HIP(hipMemcpyAsync(out_host, sy + dim + 1, sizeof(double), hipMemcpyDeviceToHost, st));
        PRG_HIP(hipStreamSynchronize(st));
        return PRG_OK;
    }
    const int nbx = (int)(prg::ceil_div(m, kBlock) < 256 ? prg::ceil_div(m, kBlock) : 256);
    const int nby = (int)(prg::ceil_div(n, kBlock) < 256 ? prg::ceil_div(n, kBlock) : 256);
    TmpBuf bx, by, bp;
    PRG_HIP(hipMalloc(&bx.p, (size_t)m * dim * sizeof(float)));
    PRG_HIP(hipMalloc(&by.p, (size_t)n * dim * sizeof(float)));
    PRG_HIP(hipMalloc(&bp.p, (size_t)(nbx + nby) * 4 * sizeof(double) + sizeof(double)));
    PRG_HIP(hipMemcpyAsync(bx.p, x_hd, (size_t)m * dim * sizeof(float), hipMemcpyDefault, st));
    PRG_HIP(hipMemcpyAsync(by.p, y_hd, (size_t)n * dim * sizeof(float), hipMemcpyDefault, st));
    double* px = (double*)bp.p;
    double* py = px + (size_t)nbx * 4;
    double* res = py + (size_t)nby * 4;
    k_sums_rowmajor<<<nbx, kBlock, 0, st>>>((const float*)bx.p, m, dim, px);
    k_sums_rowmajor<<<nby, kBlock, 0, st>>>((const float*)by.p, n, dim, py);
    k_sks_final<<<1, 64, 0, st>>>(px, nbx, py, nby, (double)m, (double)n, dim, res);
    PRG_HIP(hipGetLastError());
    PRG_HIP(hipMemcpyAsync(out_host, res, sizeof(double), hipMemcpyDeviceToHost, st));
    PRG_HIP(hipStreamSynchronize(st));
    return PRG_OK;
}

int prg_rbf_kernel(int device, void* hip_stream, const float* x_hd, int64_t m, const float* y_hd, int64_t n, int dim,
                   double beta, float* out_hd) {
    PRG_REQUIRE(x_hd && y_hd && out_hd, PRG_ERR_INVALID, "prg_rbf_kernel: NULL argument");
    PRG_REQUIRE(m > 0 && n > 0 && (dim == 2 || dim == 3) && beta > 0, PRG_ERR_INVALID,
                "prg_rbf_kernel: need m, n > 0, dim in {2,3}, beta > 0");
    prg::DeviceGuard g(device);
    PRG_REQUIRE(g.ok, PRG_ERR_HIP, "prg_rbf_kernel: hipSetDevice(%d) failed", device);
    hipStream_t st = (hipStream_t)hip_stream;
    TmpBuf bx, by, bo;
    PRG_HIP(hipMalloc(&bx.p, (size_t)m * dim * sizeof(float)));
    PRG_HIP(hipMalloc(&by.p, (size_t)n * dim * sizeof(float)));
    PRG_HIP(hipMalloc(&bo.p, (size_t)m * n * sizeof(float)));
    PRG_HIP(hipMemcpyAsync(bx.p, x_hd, (size_t)m * dim * sizeof(float), hipMemcpyDefault, st));
    PRG_HIP(hipMemcpyAsync(by.p, y_hd, (size_t)n * dim * sizeof(float), hipMemcpyDefault, st));
    dim3 grid((unsigned)prg::ceil_div(n, kBlock), (unsigned)prg::ceil_div(m, 16));
    k_rbf<<<grid, kBlock, 0, st>>>((const float*)bx.p, m, (const float*)by.p, n, dim, (float)(2.0 * beta),
                                   (float*)bo.p);
    PRG_HIP(hipGetLastError());
    PRG_HIP(hipMemcpyAsync(out_hd, bo.p, (size_t)m * n * sizeof(float), hipMemcpyDefault, st));
    PRG_HIP(hipStreamSynchronize(st));
    return PRG_OK;
}

int prg_inverse_multiquadric_kernel(int device, void* hip_stream, const float* x_hd, int64_t m, const float* y_hd,
                                    int64_t n, int dim, double c, float* out_hd) {
    PRG_REQUIRE(x_hd && y_hd && out_hd, PRG_ERR_INVALID, "prg_inverse_multiquadric_kernel: NULL argument");
    PRG_REQUIRE(m > 0 && n > 0 && (dim == 2 || dim == 3) && c > 0, PRG_ERR_INVALID,
                "prg_inverse_multiquadric_kernel: need m, n > 0, dim in {2,3}, c > 0");
    prg::DeviceGuard g(device);
    PRG_REQUIRE(g.ok, PRG_ERR_HIP, "prg_inverse_multiquadric_kernel: hipSetDevice(%d) failed", device);
    hipStream_t st = (hipStream_t)hip_stream;
    TmpBuf bx, by, bo;
    PRG_HIP(hipMalloc(&bx.p, (size_t)m * dim * sizeof(float)));
    PRG_HIP(hipMalloc(&by.p, (size_t)n * dim * sizeof(float)));
    PRG_HIP(hipMalloc(&bo.p, (size_t)m * n * sizeof(float)));
    PRG_HIP(hipMemcpyAsync(bx.p, x_hd, (size_t)m * dim * sizeof(float), hipMemcpyDefault, st));
    PRG_HIP(hipMemcpyAsync(by.p, y_hd, (size_t)n * dim * sizeof(float), hipMemcpyDefault, st));
    dim3 grid((unsigned)prg::ceil_div(n, kBlock), (unsigned)prg::ceil_div(m, 16));
    k_imq<<<grid, kBlock, 0, st>>>((const float*)bx.p, m, (const float*)by.p, n, dim, (float)c, (float*)bo.p);
    PRG_HIP(hipGetLastError());
    PRG_HIP(hipMemcpyAsync(out_hd, bo.p, (size_t)m * n * sizeof(float), hipMemcpyDefault, st));
    PRG_HIP(hipStreamSynchronize(st));
    return PRG_OK;
}

int prg_nn_mean_distance(int device, void* hip_stream, const float* a_hd, int64_t m, const float* b_hd, int64_t n,
                         int dim, double* out_host) {
    PRG_REQUIRE(a_hd && b_hd && out_host, PRG_ERR_INVALID, "prg_nn_mean_distance: NULL argument");
    PRG_REQUIRE(m > 0 && n > 0 && (dim == 2 || dim == 3), PRG_ERR_INVALID,
                "prg_nn_mean_distance: need m, n > 0 and dim in {2,3}");
    prg::DeviceGuard g(device);
    PRG_REQUIRE(g.ok, PRG_ERR_HIP, "prg_nn_mean_distance: hipSetDevice(%d) failed", device);
    hipStream_t st = (hipStream_t)hip_stream;
    // enough (a block, b segment) workgroups to fill 256 CUs even for a few thousand points
    const int64_t nbx = prg::ceil_div(m, kBlock);
    int nseg = (int)std::min<int64_t>(std::max<int64_t>(1, 2048 / nbx), prg::ceil_div(n, 256));
    const int seg_len = (int)prg::round_up(prg::ceil_div(n, nseg), 4);
    const int64_t cap = (int64_t)seg_len * nseg;
    TmpBuf ba, bb, b4, bd;
    PRG_HIP(hipMalloc(&ba.p, (size_t)m * dim * sizeof(float)));
    PRG_HIP(hipMalloc(&bb.p, (size_t)n * dim * sizeof(float)));
    PRG_HIP(hipMalloc(&b4.p, (size_t)cap * sizeof(float4)));
    PRG_HIP(hipMalloc(&bd.p, (size_t)m * sizeof(unsigned) + sizeof(double) * 2));
    PRG_HIP(hipMemcpyAsync(ba.p, a_hd, (size_t)m * dim * sizeof(float), hipMemcpyDefault, st));
    PRG_HIP(hipMemcpyAsync(bb.p, b_hd, (size_t)n * dim * sizeof(float), hipMemcpyDefault, st));
    double* res = reinterpret_cast<double*>((char*)bd.p + prg::round_up((int64_t)m * sizeof(unsigned), 8));
    k_pack_weighted<<<(unsigned)prg::ceil_div(cap, kBlock), kBlock, 0, st>>>((const float*)bb.p, nullptr, n, dim, cap,
                                                                            (float4*)b4.p);
    k_nn_fill<<<(unsigned)nbx, kBlock, 0, st>>>((unsigned*)bd.p, m);
    k_nn_min<<<dim3((unsigned)nbx, (unsigned)nseg), kBlock, 0, st>>>((const float*)ba.p, m, dim, (const float4*)b4.p,
                                                                    seg_len, (unsigned*)bd.p);
    k_nn_mean<<<1, kBlock, 0, st>>>((const unsigned*)bd.p, m, res);
    PRG_HIP(hipGetLastError());
    PRG_HIP(hipMemcpyAsync(out_host, res, sizeof(double), hipMemcpyDeviceToHost, st));
    PRG_HIP(hipStreamSynchronize(st));
    return PRG_OK;
}

int prg_gauss_transform_direct(int device, void* hip_stream, const double* source_hd, int64_t s, const double* target_hd,
                               int64_t t, int dim, const double* weights_hd, int n_weight_rows, double h,
                               double* out_hd) {
    PRG_REQUIRE(source_hd && target_hd && weights_hd && out_hd, PRG_ERR_INVALID,
                "prg_gauss_transform_direct: NULL argument");
    PRG_REQUIRE(s > 0 && t > 0 && (dim == 2 || dim == 3) && n_weight_rows > 0 && h > 0, PRG_ERR_INVALID,
                "prg_gauss_transform_direct: need s, t, rows > 0, dim in {2,3}, h > 0");
    prg::DeviceGuard g(device);
    PRG_REQUIRE(g.ok, PRG_ERR_HIP, "prg_gauss_transform_direct: hipSetDevice(%d) failed", device);
    hipStream_t st = (hipStream_t)hip_stream;
    const int64_t cap = prg::round_up(s, 256);
    const int rows = n_weight_rows;
    TmpBuf bs, bt, bw, b3, bwr, bo;
    PRG_HIP(hipMalloc(&bs.p, (size_t)s * dim * sizeof(double)));
    PRG_HIP(hipMalloc(&bt.p, (size_t)t * dim * sizeof(double)));
    PRG_HIP(hipMalloc(&bw.p, (size_t)s * rows * sizeof(double)));
    PRG_HIP(hipMalloc(&b3.p, (size_t)cap * 3 * sizeof(double)));
    PRG_HIP(hipMalloc(&bwr.p, (size_t)cap * rows * sizeof(double)));
    PRG_HIP(hipMalloc(&bo.p, (size_t)t * rows * sizeof(double)));
    PRG_HIP(hipMemcpyAsync(bs.p, source_hd, (size_t)s * dim * sizeof(double), hipMemcpyDefault, st));
    PRG_HIP(hipMemcpyAsync(bt.p, target_hd, (size_t)t * dim * sizeof(double), hipMemcpyDefault, st));
    PRG_HIP(hipMemcpyAsync(bw.p, weights_hd, (size_t)s * rows * sizeof(double), hipMemcpyDefault, st));
    const float kk = (float)(-1.4426950408889634 / (h * h));
    k_pack_gauss<<<(unsigned)prg::ceil_div(cap, kBlock), kBlock, 0, st>>>((const double*)bs.p, (const double*)bw.p, s, dim, rows, cap,
                                                                         (double*)b3.p, (double*)bwr.p);
    const unsigned grid = (unsigned)prg::ceil_div(t, kBlock);
    for (int c = 0; c < rows;) {  // four weight rows per sweep (compute_l2_dist: 1 + D rows = one sweep), then 2, then 1
        const double* w = (const double*)bwr.p + (size_t)c * cap;
        double* o = (double*)bo.p + (size_t)c * t;
        if (rows - c >= 4) {
            k_gauss_direct<4><<<grid, kBlock, 0, st>>>((const double*)b3.p, cap, w, (const double*)bt.p, t, dim, kk, o);
            c += 4;
        } else if (rows - c >= 2) {
            k_gauss_direct<2><<<grid, kBlock, 0, st>>>((const double*)b3.p, cap, w, (const double*)bt.p, t, dim, kk, o);
            c += 2;
        } else {
            k_gauss_direct<1><<<grid, kBlock, 0, st>>>((const double*)b3.p, cap, w, (const double*)bt.p, t, dim, kk, o);
            c += 1;
        }
    }
    PRG_HIP(hipGetLastError());
    PRG_HIP(hipMemcpyAsync(out_hd, bo.p, (size_t)t * rows * sizeof(double), hipMemcpyDefault, st));
    PRG_HIP(hipStreamSynchronize(st));
    return PRG_OK;
}

}  // extern "C"
