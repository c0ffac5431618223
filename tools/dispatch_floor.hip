// How long does the GPU need just to run N workgroups of 256 threads that do (almost) nothing - and what makes
// that floor rise?  hipcc --offload-arch=gfx950 -O3 -o tools/bin/dispatch_floor tools/dispatch_floor.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ __launch_bounds__(256) void k_empty(float* out) {
    if (threadIdx.x == 0 && blockIdx.x == 0x7fffffff) out[0] = 1.f;
}
__global__ __launch_bounds__(256) void k_touch(const float4* __restrict__ in, float2* __restrict__ out, int n) {
    const int i = (blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x;
    const float4 a = in[(i * 2) % n], b = in[(i * 2 + 1) % n];
    float v = a.x + b.y;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    if (v == 123.f) out[i] = make_float2(v, a.z);
}
// many live registers at wave start (forces a large VGPR allocation), otherwise empty
template <int NV>
__global__ __launch_bounds__(256) void k_fat(const float* __restrict__ in, float* __restrict__ out, int never) {
    float acc[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) acc[k] = (float)k;
    if (never) {
        for (int it = 0; it < never; ++it)
#pragma unroll
            for (int k = 0; k < NV; ++k) acc[k] = fmaf(acc[k], in[(it + k) & 1023], acc[(k + 1) % NV]);
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k) s += acc[k];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    }
}
// every lane stores 16 bytes (what an all-culled column-pass wave still writes)
__global__ __launch_bounds__(256) void k_store(float4* __restrict__ out) {
    const size_t i = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x;
    out[i] = make_float4(INFINITY, 0.f, INFINITY, 0.f);
}
// 32-byte vector load per lane from a small hot table + ballot
__global__ __launch_bounds__(256) void k_meta(const float4* __restrict__ meta, float* __restrict__ out, float thr) {
    const int lane = threadIdx.x & 63;
    const float4 a = meta[(blockIdx.y * 56 + lane) * 2], b = meta[(blockIdx.y * 56 + lane) * 2 + 1];
    const bool need = a.x * b.y + a.z > thr;
    if (__ballot(need) != 0ull) out[blockIdx.x * 256 + threadIdx.x] = a.w;
}

// LDS allocation + a barrier at launch + an LDS counter at the end (the skeleton of the culled sweeps' merge)
__global__ __launch_bounds__(256) void k_lds(const float4* __restrict__ meta, float* __restrict__ out, float thr) {
    __shared__ float4 part[4][5][64];
    __shared__ int arrived;
    if (threadIdx.x == 0) arrived = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const float4 a = meta[((blockIdx.y * 4 + wv) * 16 + (lane & 15)) * 2];
    const bool need = a.x * a.y + a.z > thr;
    if (__ballot(need) != 0ull) part[wv][0][lane] = a;
    int last = 0;
    if (lane == 0) last = atomicAdd(&arrived, 1) == 3;
    if (__builtin_amdgcn_readfirstlane(last) && thr < 0.f) out[blockIdx.x * 256 + threadIdx.x] = part[0][0][lane].x;
}

template <typename F>
static void timeit(const char* what, int gx, int gy, F launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(e0, 0);
        launch();
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("grid %3d x %3d = %6d workgroups  %-28s %7.1f us\n", gx, gy, gx * gy, what, best * 1e3f);
}

int main() {
    float4* in;
    float2* out;
    const int n = 1 << 20;
    (void)hipMalloc(&in, n * sizeof(float4));
    (void)hipMalloc(&out, (size_t)64 * n * sizeof(float2));
    (void)hipMemset(in, 0, n * sizeof(float4));
    const int grids[][2] = {{196, 64}, {25, 196}, {782, 49}, {782, 16}};
    for (auto& g : grids) {
        dim3 gr(g[0], g[1]);
        timeit("empty", g[0], g[1], [&] { k_empty<<<gr, 256>>>((float*)out); });
        timeit("2 loads + wave min", g[0], g[1], [&] { k_touch<<<gr, 256>>>(in, out, n); });
        timeit("24 live VGPRs", g[0], g[1], [&] { k_fat<24><<<gr, 256>>>((const float*)in, (float*)out, 0); });
        timeit("48 live VGPRs", g[0], g[1], [&] { k_fat<48><<<gr, 256>>>((const float*)in, (float*)out, 0); });
        timeit("96 live VGPRs", g[0], g[1], [&] { k_fat<96><<<gr, 256>>>((const float*)in, (float*)out, 0); });
        timeit("16-byte store per lane", g[0], g[1], [&] { k_store<<<gr, 256>>>((float4*)out); });
        timeit("32-byte meta load + ballot", g[0], g[1], [&] { k_meta<<<gr, 256>>>(in, (float*)out, 1.f); });
        timeit("LDS 20 KB + barrier + counter", g[0], g[1], [&] { k_lds<<<gr, 256>>>(in, (float*)out, 1.f); });
    }
    return 0;
}
