// Does a wave that is alone on its SIMD reach the SIMD's VALU throughput?  The row pass' per-point arithmetic
// (12 packed ops + 2 exp per two pairs) on registers only, launched with k waves per SIMD (k = 1, 2, 4, 8).
// hipcc --offload-arch=gfx950 -O3 -o tools/bin/lone_wave tools/lone_wave.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 splat(float v) { return (f2){v, v}; }
__device__ __forceinline__ f2 fmav(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 exp2v(f2 a) { return (f2){__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)}; }

__global__ __launch_bounds__(256) void k_math(float* out, int iters, float kk, float sx, float sy, float sz, float sb) {
    const float t = (float)threadIdx.x * 1e-3f;
    f2 zx = {t, t + 0.5f}, zy = {t * 2.f, t}, zz = {t, -t}, zq = {0.f, 0.f};
    f2 p1 = splat(0.f), ux = splat(0.f), uy = splat(0.f), uz = splat(0.f), e = splat(0.f);
    float qx = sx, qy = sy, qz = sz, qb = sb;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f2 dx = zx - splat(qx), dy = zy - splat(qy), dz = zz - splat(qz);
            const f2 d = fmav(dz, dz, fmav(dy, dy, fmav(dx, dx, zq)));
            const f2 pr = exp2v(fmav(d, splat(kk), splat(qb)));
            p1 += pr;
            ux = fmav(pr, dx, ux);
            uy = fmav(pr, dy, uy);
            uz = fmav(pr, dz, uz);
            e = fmav(pr, d, e);
            qx += 0.001f; qy -= 0.002f; qz += 0.0005f;   // scalar ALU work (wave-uniform), like new streamed points
        }
    }
    const f2 s = p1 + ux + uy + uz + e;
    if (s.x + s.y == 123.456f) out[blockIdx.x * 256 + threadIdx.x] = s.x;
}

int main() {
    float* out;
    (void)hipMalloc(&out, 1 << 24);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000;  // x 4 points
    for (int k : {1, 2, 3, 4, 8}) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0, 0);
            k_math<<<256 * k, 256>>>(out, iters, -1.3f, 0.1f, 0.2f, 0.3f, -2.f);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        const double points = (double)iters * 4;
        printf("%d wave(s) per SIMD: %.3f ms  -> %.1f cycles per streamed point per wave at 2.4 GHz, %.1f per SIMD\n", k, best,
               best * 1e-3 * 2.4e9 / points, best * 1e-3 * 2.4e9 / points / k);
    }
    return 0;
}
