// As tools/mfma_overlap.hip, with the distance block on the bf16 matrix pipe (v_mfma_f32_16x16x32_bf16): does THAT pipe
// overlap with the vector pipe's exponentials?
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form -o tools/bin/mfma_overlap_bf16 tools/mfma_overlap_bf16.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
constexpr int ITER = 2048, U = 8;

__device__ __forceinline__ void valu_block(const f4 d, float& tm, float& s) {
    tm = fmaxf(fmaxf(tm, d[0]), d[1]);
    tm = fmaxf(fmaxf(tm, d[2]), d[3]);
    s += (__builtin_amdgcn_exp2f(d[0]) + __builtin_amdgcn_exp2f(d[1])) + (__builtin_amdgcn_exp2f(d[2]) + __builtin_amdgcn_exp2f(d[3]));
}

__global__ __launch_bounds__(256) void k(float* out, float a, int mode) {
    float tm[U], s[U];
    f4 acc[U];
    bf8 b[U];
    for (int u = 0; u < U; ++u) {
        tm[u] = -1e30f; s[u] = 0.f; acc[u] = (f4){0, 0, 0, 0};
        for (int q = 0; q < 8; ++q) b[u][q] = (__bf16)(a * (u + 1) * 1e-3f + threadIdx.x * 1e-5f + q * 1e-4f);
    }
    bf8 x;
    for (int q = 0; q < 8; ++q) x[q] = (__bf16)(threadIdx.x * 1e-3f - 1.f + q * 1e-3f);
    const f4 c = {-1.f, -2.f, -3.f, -4.f};
    if (mode == 0) {
        for (int it = 0; it < ITER; ++it)
#pragma unroll
            for (int u = 0; u < U; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, b[u], acc[u], 0, 0, 0);
    } else if (mode == 1) {
        f4 d = {(float)x[0], (float)x[1], (float)x[2], (float)x[3]};
        for (int it = 0; it < ITER; ++it)
#pragma unroll
            for (int u = 0; u < U; ++u) {
                valu_block(d, tm[u], s[u]);
                d[u & 3] = tm[u] * 1e-30f - s[u] * 1e-30f - 1.f;
            }
    } else if (mode == 3) {  // half the waves only MFMAs, the other half only vector blocks
        if (((threadIdx.x >> 6) & 1) == 0) {
            for (int it = 0; it < ITER; ++it)
#pragma unroll
                for (int u = 0; u < U; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, b[u], acc[u], 0, 0, 0);
        } else {
            f4 d = {(float)x[0], (float)x[1], (float)x[2], (float)x[3]};
            for (int it = 0; it < ITER; ++it)
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    valu_block(d, tm[u], s[u]);
                    d[u & 3] = tm[u] * 1e-30f - s[u] * 1e-30f - 1.f;
                }
        }
    } else if (mode == 4) {  // phases: U MFMAs back to back, then the U vector blocks (waves drift out of phase)
        for (int it = 0; it < ITER; ++it) {
            f4 d[U];
#pragma unroll
            for (int u = 0; u < U; ++u) d[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, b[u], c, 0, 0, 0);
#pragma unroll
            for (int u = 0; u < U; ++u) valu_block(d[u], tm[u], s[u]);
            x[0] = (__bf16)(s[0] * 1e-30f);
        }
    } else {
        f4 d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, b[0], c, 0, 0, 0);
        for (int it = 0; it < ITER; ++it)
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const f4 dn = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, b[(u + 1) % U], c, 0, 0, 0);
                valu_block(d, tm[u], s[u]);
                d = dn;
            }
    }
    float r = 0.f;
    for (int u = 0; u < U; ++u) r += tm[u] + s[u] + acc[u][0] + acc[u][1] + acc[u][2] + acc[u][3];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    float* out;
    const char* names[5] = {"bf16 mfma 16x16x32 only", "valu only", "both, every wave (pipelined)",
                            "both, split over waves (half each)", "both, phases of 8 MFMAs then 8 vector blocks"};
    for (int wpsimd = 2; wpsimd <= 8; wpsimd *= 2) {
        const int blocks = prop.multiProcessorCount * wpsimd;
        hipMalloc(&out, (size_t)blocks * 256 * 4);
        for (int mode = 0; mode < 5; ++mode) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            k<<<blocks, 256>>>(out, 1.0001f, mode);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int r = 0; r < 5; ++r) k<<<blocks, 256>>>(out, 1.0001f, mode);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            ms /= 5;
            printf("%d waves/SIMD  %-46s %.3f ms  %.1f cycles per block per SIMD at 2.4 GHz\n", wpsimd, names[mode], ms,
                   ms * 1e-3 * 2.4e9 / ((double)ITER * U * wpsimd * (mode == 3 ? 0.5 : 1.0)));
        }
        hipFree(out);
    }
    return 0;
}
