// Prints the lane/register -> matrix element maps of the two f32 MFMA shapes the dense CPD sweeps use
// (v_mfma_f32_16x16x4_f32 and v_mfma_f32_4x4x1_16b_f32), found by feeding one-hot operands.  gfx950 only.
//   hipcc --offload-arch=gfx950 -O2 -o tools/bin/mfma_layout_probe tools/mfma_layout_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// out[(la * 64 + lb) ...] too big; instead: A = lane id + 1 in lane la only..  we use value encoding:
// A(lane) = 1 + lane, B(lane) = 1000 * (1 + lane) as exact integers; with one-hot masks chosen by the host.
__global__ void probe16(const float* a, const float* b, float* d) {
    const int l = threadIdx.x;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[l], b[l], c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) d[l * 4 + r] = c[r];
}
__global__ void probe4(const float* a, const float* b, float* d) {
    const int l = threadIdx.x;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) d[l * 4 + r] = c[r];
}

int main() {
    float *a, *b, *d;
    hipMallocManaged(&a, 64 * 4);
    hipMallocManaged(&b, 64 * 4);
    hipMallocManaged(&d, 256 * 4);
    int bad = 0;
    // 16x16x4: expect A lane l = A[i=l%16][k=l/16], B lane l = B[k=l/16][j=l%16], D lane l reg r = D[i=4*(l/16)+r][j=l%16]
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            for (int l = 0; l < 64; ++l) { a[l] = l == la ? 1.f : 0.f; b[l] = l == lb ? 1.f : 0.f; }
            hipLaunchKernelGGL(probe16, 1, 64, 0, 0, a, b, d);
            hipDeviceSynchronize();
            const int i = la % 16, ka = la / 16, kb = lb / 16, j = lb % 16;
            for (int l = 0; l < 64; ++l)
                for (int r = 0; r < 4; ++r) {
                    const float want = (ka == kb && 4 * (l / 16) + r == i && l % 16 == j) ? 1.f : 0.f;
                    if (d[l * 4 + r] != want) ++bad;
                }
        }
    printf("16x16x4 layout: %s (%d mismatches)\n", bad ? "UNEXPECTED" : "as documented", bad);
    bad = 0;
    // 4x4x1 16 blocks: expect A lane l = A_b[i=l%4], b=l/4; B lane l = B_b[j=l%4]; D lane l reg r = D_b[i=r][j=l%4], b=l/4
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            for (int l = 0; l < 64; ++l) { a[l] = l == la ? 1.f : 0.f; b[l] = l == lb ? 1.f : 0.f; }
            hipLaunchKernelGGL(probe4, 1, 64, 0, 0, a, b, d);
            hipDeviceSynchronize();
            for (int l = 0; l < 64; ++l)
                for (int r = 0; r < 4; ++r) {
                    const float want = (la / 4 == lb / 4 && l / 4 == la / 4 && r == la % 4 && l % 4 == lb % 4) ? 1.f : 0.f;
                    if (d[l * 4 + r] != want) {
                        if (bad < 8) printf("  4x4x1: a-lane %d b-lane %d -> d[lane %d][reg %d] = %g (expected %g)\n", la, lb, l, r, d[l * 4 + r], want);
                        ++bad;
                    }
                }
        }
    printf("4x4x1_16b layout: %s (%d mismatches)\n", bad ? "UNEXPECTED" : "as documented", bad);
    return 0;
}
