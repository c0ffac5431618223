// VALU micro-benchmarks for gfx950: issue rates of the instructions the CPD pair sweeps are made of.
// Build: hipcc -O3 --offload-arch=gfx950 ubench.hip -o ubench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITER = 4096;

__global__ __launch_bounds__(256) void k_fma(float* out, float a, float b) {
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 1e-3f + i;
    for (int it = 0; it < ITER; ++it)
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaf(v[i], a, b);
    float s = 0; for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_pkfma(float* out, float a, float b) {
    f2 v[8];
    for (int i = 0; i < 8; ++i) v[i] = (f2){threadIdx.x * 1e-3f + i, threadIdx.x * 2e-3f + i};
    const f2 av = {a, a}, bv = {b, b};
    for (int it = 0; it < ITER; ++it)
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = __builtin_elementwise_fma(v[i], av, bv);
    float s = 0; for (int i = 0; i < 8; ++i) s += v[i].x + v[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_exp(float* out, float a) {
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = -(threadIdx.x * 1e-3f + i);
    for (int it = 0; it < ITER; ++it)
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = __builtin_amdgcn_exp2f(v[i]) - a;
    float s = 0; for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
// exp only (dependent chains of exp, 8 independent)
__global__ __launch_bounds__(256) void k_exp_only(float* out) {
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = -(threadIdx.x * 1e-3f + i) * 1e-3f;
    for (int it = 0; it < ITER; ++it)
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = __builtin_amdgcn_exp2f(v[i]);
    float s = 0; for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
// mix like the row pass: 6 pk + 1 exp-pair per 2 "pairs"
__global__ __launch_bounds__(256) void k_mix(float* out, float a, float b) {
    f2 v[4], acc[4];
    for (int i = 0; i < 4; ++i) { v[i] = (f2){threadIdx.x * 1e-3f + i, threadIdx.x * 2e-3f + i}; acc[i] = (f2){0, 0}; }
    const f2 av = {a, a}, bv = {b, b};
    for (int it = 0; it < ITER; ++it)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f2 d = v[i] - bv; f2 q = d * d; q = __builtin_elementwise_fma(d, d, q); q = __builtin_elementwise_fma(d, d, q);
            f2 e = __builtin_elementwise_fma(q, av, bv);
            f2 p = (f2){__builtin_amdgcn_exp2f(e.x), __builtin_amdgcn_exp2f(e.y)};
            acc[i] += p; v[i] = __builtin_elementwise_fma(p, d, v[i]);
        }
    float s = 0; for (int i = 0; i < 4; ++i) s += acc[i].x + acc[i].y + v[i].x;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
// f32 MFMA 4x4x1 16 blocks interleaved with pk fma: do the two pipes overlap inside one wave?
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_mfma4(float* out, float a, float b) {
    f4 c[4];
    for (int i = 0; i < 4; ++i) c[i] = (f4){0, 0, 0, 0};
    float x = threadIdx.x * 1e-3f;
    for (int it = 0; it < ITER; ++it)
#pragma unroll
        for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(x, a, c[i], 0, 0, 0);
    float s = 0; for (int i = 0; i < 4; ++i) s += c[i].x + c[i].y + c[i].z + c[i].w;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_mfma4_pk(float* out, float a, float b) {
    f4 c[4]; f2 v[8];
    for (int i = 0; i < 4; ++i) c[i] = (f4){0, 0, 0, 0};
    for (int i = 0; i < 8; ++i) v[i] = (f2){threadIdx.x * 1e-3f + i, threadIdx.x * 2e-3f + i};
    const f2 av = {a, a}, bv = {b, b};
    float x = threadIdx.x * 1e-3f;
    for (int it = 0; it < ITER; ++it)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            c[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(x, a, c[i], 0, 0, 0);
            v[2 * i] = __builtin_elementwise_fma(v[2 * i], av, bv);
            v[2 * i + 1] = __builtin_elementwise_fma(v[2 * i + 1], av, bv);
        }
    float s = 0; for (int i = 0; i < 4; ++i) s += c[i].x + c[i].y + c[i].z + c[i].w;
    for (int i = 0; i < 8; ++i) s += v[i].x + v[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

typedef double d4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_mfma_f64(double* out, double a, double b) {
    d4 c[8];
    for (int i = 0; i < 8; ++i) c[i] = (d4){0, 0, 0, 0};
    double x = threadIdx.x * 1e-3, y = threadIdx.x * 2e-3;
    for (int it = 0; it < ITER / 4; ++it)
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c[i], 0, 0, 0);
    double s = 0; for (int i = 0; i < 8; ++i) s += c[i].x + c[i].y + c[i].z + c[i].w;
    out[blockIdx.x * 256 + threadIdx.x] = s + a + b;
}
__global__ __launch_bounds__(256) void k_fma_f64(double* out, double a, double b) {
    double v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 1e-3 + i;
    for (int it = 0; it < ITER; ++it)
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = __builtin_fma(v[i], a, b);
    double s = 0; for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename F>
float time_kernel(F launch) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    launch(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < 5; ++i) launch(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / 5;
}

int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s  CUs=%d  clock=%d kHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate);
    const int blocks = prop.multiProcessorCount * 8;  // 8 waves/SIMD
    float* out; CHECK(hipMalloc(&out, blocks * 256 * sizeof(float)));
    const double lanes = (double)blocks * 256;
    float ms;
    ms = time_kernel([&] { k_fma<<<blocks, 256>>>(out, 1.0001f, 0.5f); });
    printf("v_fma_f32      : %.3f ms  %.1f Glane-op/s  (%.1f TFLOP/s)\n", ms, lanes * ITER * 8 / ms / 1e6, lanes * ITER * 8 * 2 / ms / 1e9);
    ms = time_kernel([&] { k_pkfma<<<blocks, 256>>>(out, 1.0001f, 0.5f); });
    printf("v_pk_fma_f32   : %.3f ms  %.1f Glane-instr/s (%.1f TFLOP/s)\n", ms, lanes * ITER * 8 / ms / 1e6, lanes * ITER * 8 * 4 / ms / 1e9);
    ms = time_kernel([&] { k_exp<<<blocks, 256>>>(out, 0.5f); });
    printf("v_exp+v_sub    : %.3f ms  %.1f Glane-pair/s\n", ms, lanes * ITER * 8 / ms / 1e6);
    ms = time_kernel([&] { k_exp_only<<<blocks, 256>>>(out); });
    printf("v_exp_f32 only : %.3f ms  %.1f Glane-op/s\n", ms, lanes * ITER * 8 / ms / 1e6);
    ms = time_kernel([&] { k_mix<<<blocks, 256>>>(out, -1.0001f, 0.5f); });
    printf("mix 7pk+2exp   : %.3f ms  %.1f Gpair/s (2 pairs per group)\n", ms, lanes * ITER * 4 * 2 / ms / 1e6);
    ms = time_kernel([&] { k_mfma4<<<blocks, 256>>>(out, 1.0001f, 0.5f); });
    printf("mfma_4x4x1 f32 : %.3f ms  %.1f Gwave-instr/s  (%.1f TFLOP/s)\n", ms, lanes / 64 * ITER * 4 / ms / 1e6, lanes / 64 * ITER * 4 * 512 / ms / 1e9);
    ms = time_kernel([&] { k_mfma4_pk<<<blocks, 256>>>(out, 1.0001f, 0.5f); });
    printf("mfma4x4 + 2pk  : %.3f ms  (same mfma count as above + 2 pk_fma per mfma)\n", ms);
    double* outd; CHECK(hipMalloc(&outd, blocks * 256 * sizeof(double)));
    for (int wpc = 1; wpc <= 2; ++wpc) {   // 1 or 2 workgroups (4 / 8 waves) per CU
        const int nb = prop.multiProcessorCount * wpc;
        ms = time_kernel([&] { k_mfma_f64<<<nb, 256>>>(outd, 1.0001, 0.5); });
        printf("mfma_f64 16x16x4 (%d WG/CU): %.3f ms  %.2f TFLOP/s  (%.1f cycles per MFMA per SIMD at 2.4 GHz)\n", wpc, ms,
               (double)nb * 4 * (ITER / 4) * 8 * 2048.0 / ms / 1e9, ms * 1e-3 * 2.4e9 / ((ITER / 4) * 8.0 * wpc));
    }
    ms = time_kernel([&] { k_fma_f64<<<blocks, 256>>>(outd, 1.0001, 0.5); });
    printf("v_fma_f64      : %.3f ms  %.1f TFLOP/s\n", ms, lanes * ITER * 8 * 2 / ms / 1e9);
    hipFree(outd);
    hipFree(out);
    return 0;
}
