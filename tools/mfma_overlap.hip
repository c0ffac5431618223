// Do the f32 MFMA pipe and the vector pipe of one gfx950 SIMD overlap?  Times, per SIMD and per "block" of work
// (one v_mfma_f32_16x16x4_f32 + 4 v_exp_f32 + 4 v_add_f32 + 2 v_max3_f32 = the column pass' 16 x 16 block of pairs):
//   mfma only / valu only / both in one wave (software pipelined) / both, in different waves of the same SIMD.
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form -o tools/bin/mfma_overlap tools/mfma_overlap.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int ITER = 2048, U = 8;

__device__ __forceinline__ void valu_block(const f4 d, float& tm, float& s) {
    tm = fmaxf(fmaxf(tm, d[0]), d[1]);
    tm = fmaxf(fmaxf(tm, d[2]), d[3]);
    s += (__builtin_amdgcn_exp2f(d[0]) + __builtin_amdgcn_exp2f(d[1])) + (__builtin_amdgcn_exp2f(d[2]) + __builtin_amdgcn_exp2f(d[3]));
}

// mode 0: mfma only, 1: valu only, 2: both pipelined in every wave, 3: even waves mfma only / odd waves valu only
__global__ __launch_bounds__(256) void k(float* out, float a, int mode) {
    const int wv = threadIdx.x >> 6;
    float b[U], tm[U], s[U];
    f4 acc[U];
    for (int u = 0; u < U; ++u) { b[u] = a * (u + 1) + threadIdx.x * 1e-3f; tm[u] = -1e30f; s[u] = 0.f; acc[u] = (f4){0, 0, 0, 0}; }
    const f4 c = {-1.f, -2.f, -3.f, -4.f};
    float x = threadIdx.x * 1e-3f - 1.f;
    const bool do_m = mode == 0 || mode == 2 || (mode == 3 && (wv & 1) == 0);
    const bool do_v = mode == 1 || mode == 2 || (mode == 3 && (wv & 1) == 1);
    if (mode == 2) {
        f4 d = __builtin_amdgcn_mfma_f32_16x16x4f32(x, b[0], c, 0, 0, 0);
        for (int it = 0; it < ITER; ++it)
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const f4 dn = __builtin_amdgcn_mfma_f32_16x16x4f32(x, b[(u + 1) % U], c, 0, 0, 0);
                valu_block(d, tm[u], s[u]);
                d = dn;
            }
    } else if (mode == 4) {  // batched: U MFMAs back to back, then the U vector blocks
        for (int it = 0; it < ITER; ++it) {
            f4 d[U];
#pragma unroll
            for (int u = 0; u < U; ++u) d[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, b[u], c, 0, 0, 0);
#pragma unroll
            for (int u = 0; u < U; ++u) valu_block(d[u], tm[u], s[u]);
            x += s[0] * 1e-30f;
        }
    } else if (mode == 5) {  // batched and pipelined: the next batch of MFMAs is issued before this batch's vector blocks
        f4 d[U];
#pragma unroll
        for (int u = 0; u < U; ++u) d[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, b[u], c, 0, 0, 0);
        for (int it = 0; it < ITER; ++it) {
            f4 dn[U];
#pragma unroll
            for (int u = 0; u < U; ++u) dn[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, b[u], c, 0, 0, 0);
#pragma unroll
            for (int u = 0; u < U; ++u) valu_block(d[u], tm[u], s[u]);
#pragma unroll
            for (int u = 0; u < U; ++u) d[u] = dn[u];
            x += s[0] * 1e-30f;
        }
    } else if (do_m) {
        for (int it = 0; it < ITER; ++it)
#pragma unroll
            for (int u = 0; u < U; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, b[u], acc[u], 0, 0, 0);
    } else if (do_v) {
        f4 d = {x, x * 0.5f, x * 0.25f, x * 0.125f};
        for (int it = 0; it < ITER; ++it)
#pragma unroll
            for (int u = 0; u < U; ++u) {
                valu_block(d, tm[u], s[u]);
                d[u & 3] = tm[u] * 1e-30f - s[u] * 1e-30f - 1.f;  // keep a dependence so nothing is hoisted
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
    const char* names[6] = {"mfma only", "valu only", "both, every wave (pipelined)", "both, split over waves (half the waves each)",
                            "both, batches of 8 MFMAs then 8 vector blocks", "both, batched + next batch's MFMAs issued first"};
    for (int wpsimd = 1; wpsimd <= 4; wpsimd *= 2) {
        const int blocks = prop.multiProcessorCount * wpsimd;  // 256-thread WGs: wpsimd waves per SIMD
        hipMalloc(&out, (size_t)blocks * 256 * 4);
        for (int mode = 0; mode < 6; ++mode) {
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
            // blocks of work per SIMD: every wave does ITER * U blocks (mode 3: half the waves do each kind)
            const double per_simd = (double)ITER * U * wpsimd * (mode == 3 ? 0.5 : 1.0);
            printf("%d waves/SIMD  %-46s %.3f ms  %.1f cycles per block per SIMD at 2.4 GHz\n", wpsimd, names[mode], ms,
                   ms * 1e-3 * 2.4e9 / per_simd);
        }
        hipFree(out);
    }
    return 0;
}
