// E-step pair sweeps on the matrix cores (f32 MFMA), MI355X gfx950 - the DENSE regime of CPD's E-step (cpd.py:71-88).
//
// While sigma2 is large every source-target pair contributes and the VALU sweeps of cpd_sweeps_packed.hip are bound by
// their ~13 vector instructions per pair.  Here the squared distances come from the matrix pipe instead:
//
//   kk |x - z|^2 + b  =  [x'_x x'_y x'_z 1] . [-2kk z'_x, -2kk z'_y, -2kk z'_z, kk |z'|^2]^T  +  (kk |x'|^2 + b)
//
// is one v_mfma_f32_16x16x4_f32 per 16 x 16 block of pairs (K = 4 holds the three coordinates and the owned point's
// constant; the streamed point's constant rides in the C operand), the vector pipe only exponentiates (4 v_exp_f32 per
// lane per block), and the row pass' contraction P @ [x' 1] is four v_mfma_f32_4x4x1_16b_f32 whose A operand IS the
// exponentiated accumulator of the first MFMA and whose B operand is a 16-byte LDS read - no cross-lane traffic
// (lane maps: tools/mfma_layout_probe.hip).  f32 MFMA is an exact k-ordered fmaf chain and runs at the vector rate
// (157 TFLOP/s) CONCURRENTLY with the vector pipe: a block of 256 pairs costs 32 (column pass) / 64 (row pass) matrix
// cycles and 4 transcendentals per lane, against ~100 vector cycles in the VALU sweeps.
//
// Work mapping.  A workgroup (4 waves) owns 4 x OWN tiles of 16 consecutive points of one cloud (512 points: one
// spatially compact patch, the clouds are sorted along a space-filling curve) and streams a segment of the other
// cloud through LDS in chunks of 512 points: every thread loads two float4 points, shifts them by the workgroup's
// origin, forms the per-point constant and writes planes x' y' z' 1 c |x'|^2; the waves then read MFMA operands
// straight out of those planes (one ds_read_b32 + three ds_read_b128 per 16 streamed points, conflict-free).  The
// next chunk's global loads are in flight while the current one is multiplied; one barrier per chunk.
//
// Precision.  The expanded form cancels: its rounding error is eps * |kk| * (|x'|^2 + |z'|^2), not eps * |exponent|.  Every
// workgroup therefore shifts both clouds by ITS OWN origin o (the first of the points it owns): for the pairs that
// matter (within a few sigma of the patch) |x'| and |z'| are patch-sized.  Measured along a C1 registration
// (tools/mfma_vs_valu.py): sigma2 after the M-step within 8e-7 of the VALU sweeps' at every sigma2 down to 5e-5,
// rotation within 1e-8.  The row pass' sums come out relative to o - u' = sum P (x - o), e' = sum P |x - o|^2 - which is
// the residual form k_row_moments already consumes (it takes o instead of z_m as the reference point).
//
// Range.  The column pass needs an exponent offset per column BEFORE it sees the data: the previous E-step's column
// minimum and the source motion since bracket this E-step's minimum (triangle inequality), prg::col_seed_offset turns
// the bracket into an offset that neither overflows nor underflows for brackets up to 180 exponent units wide (the host
// checks the widest bracket and falls back to the VALU sweeps beyond); k_colfinal reproduces the offset bit for bit.
#include <math.h>

#include "cpd_sweeps.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr double kLog2e = 1.4426950408889634;
constexpr int kBlock = prg::kSweepBlock;
constexpr int kOwn = prg::kMfmaOwn;          // tiles of 16 points a wave owns
constexpr int kWgPoints = 4 * 16 * kOwn;     // points a workgroup owns (512)
constexpr int kChunk = 512;                  // streamed points staged in LDS at a time (2 per thread)
constexpr int kChunkTiles = kChunk / 16;
constexpr int kPlane = 96;                   // floats per staged tile: x'[16] y'[16] z'[16] 1[16] c[16] |x'|^2[16]

__device__ __forceinline__ float exp2r(float a) { return __builtin_amdgcn_exp2f(a); }
__device__ __forceinline__ float sel4(int k, float a, float b, float c, float d) {
    return k == 0 ? a : (k == 1 ? b : (k == 2 ? c : d));
}
__device__ __forceinline__ float xor_sum(float v) {  // sum over the four 16-lane rows of a wave (lanes l, l^16, l^32, l^48)
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
__device__ __forceinline__ float xor_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    v = fmaxf(v, __shfl_xor(v, 32, 64));
    return v;
}

// one streamed point -> its planes in the staging buffer.  `w` is added to the constant (the row pass' b_n; the source's
// additive weight is 0 for plain CPD).  Pads (1e18 away) give c = -huge or -inf and every exponential of theirs is 0.
__device__ __forceinline__ void stage_point(float* __restrict__ buf, int p, const float4 v, const float4 o, float kk) {
    const float dx = v.x - o.x, dy = v.y - o.y, dz = v.z - o.z;
    const float sq = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
    float* t = buf + (p >> 4) * kPlane + (p & 15);
    t[0] = dx;
    t[16] = dy;
    t[32] = dz;
    t[48] = 1.f;
    t[64] = fmaf(kk, sq, v.w);
    t[80] = sq;
}

// ---- sweep 1 on the matrix cores: den_n of cpd.py:80 --------------------------------------------------------------
// grid = (ceil(N / 512), S); plane blockIdx.y receives (min d^2 over the segment, sum of exp2(kk d^2 + L_n)) with
// L_n = prg::col_seed_offset - the same for every segment of a column.
__global__ __launch_bounds__(kBlock) void k_colpass_mfma(const float4* __restrict__ tgt4, const float4* __restrict__ z4,
                                                         const float* __restrict__ colmin_prev,
                                                         const unsigned* __restrict__ motion, int chunks_per_seg,
                                                         int64_t m_total, const double* __restrict__ params,
                                                         float2* __restrict__ colpart, int64_t ncap) {
    __shared__ float stage[2][kChunkTiles * kPlane];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t n0wg = (int64_t)blockIdx.x * kWgPoints, n0 = n0wg + wv * (16 * kOwn);
    const float kk = (float)(-kLog2e / (2.0 * params[13]));
    const float mo = __uint_as_float(*motion);
    float4 o = tgt4[n0wg];  // the workgroup's origin: a real point of its patch
    o.w = 0.f;
    const int k = lane >> 4, j = lane & 15;
    float bx[kOwn], s[kOwn], tm[kOwn], off[kOwn];
#pragma unroll
    for (int t = 0; t < kOwn; ++t) {
        const float4 x = tgt4[n0 + 16 * t + j];
        const float xx = x.x - o.x, xy = x.y - o.y, xz = x.z - o.z;
        const float xsq = fmaf(xz, xz, fmaf(xy, xy, xx * xx));
        off[t] = prg::col_seed_offset(kk, colmin_prev[n0 + 16 * t + j], mo);
        const float m2 = -2.f * kk;
        bx[t] = sel4(k, m2 * xx, m2 * xy, m2 * xz, fmaf(kk, xsq, off[t]));
        s[t] = 0.f;
        tm[t] = -INFINITY;
    }
    const int64_t c0 = (int64_t)blockIdx.y * chunks_per_seg;
    const int64_t nchunks = (m_total + kChunk - 1) / kChunk;
    const int64_t c1 = c0 + chunks_per_seg < nchunks ? c0 + chunks_per_seg : nchunks;
    float4 ra = z4[c0 * kChunk + threadIdx.x], rb = z4[c0 * kChunk + kBlock + threadIdx.x];
    ra.w = rb.w = 0.f;  // (weighted sources do not take this path)
    stage_point(stage[0], threadIdx.x, ra, o, kk);
    stage_point(stage[0], kBlock + threadIdx.x, rb, o, kk);
    __syncthreads();
    for (int64_t c = c0; c < c1; ++c) {
        const float* __restrict__ buf = stage[(c - c0) & 1];
        const bool more = c + 1 < c1;
        if (more) {
            ra = z4[(c + 1) * kChunk + threadIdx.x];
            rb = z4[(c + 1) * kChunk + kBlock + threadIdx.x];
            ra.w = rb.w = 0.f;
        }
        // software pipeline: the MFMA of the NEXT (tile, column tile) pair is issued before the current pair's
        // accumulator is exponentiated, so the matrix pipe's 40-cycle latency hides under the transcendentals
        float a1 = buf[lane];
        f32x4 cz = *reinterpret_cast<const f32x4*>(buf + 64 + 4 * k);
        f32x4 d = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bx[0], cz, 0, 0, 0);
        for (int t = 0; t < kChunkTiles; ++t) {
            const float* __restrict__ tn = buf + (t + 1 < kChunkTiles ? t + 1 : t) * kPlane;
            const float a1n = tn[lane];
            const f32x4 czn = *reinterpret_cast<const f32x4*>(tn + 64 + 4 * k);
#pragma unroll
            for (int u = 0; u < kOwn; ++u) {
                const f32x4 dn = u + 1 < kOwn ? __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bx[u + 1], cz, 0, 0, 0)
                                              : __builtin_amdgcn_mfma_f32_16x16x4f32(a1n, bx[0], czn, 0, 0, 0);
                tm[u] = fmaxf(fmaxf(tm[u], d[0]), d[1]);
                tm[u] = fmaxf(fmaxf(tm[u], d[2]), d[3]);
                s[u] += (exp2r(d[0]) + exp2r(d[1])) + (exp2r(d[2]) + exp2r(d[3]));
                d = dn;
            }
            a1 = a1n;
            cz = czn;
        }
        if (more) {
            float* __restrict__ nb = stage[(c + 1 - c0) & 1];
            stage_point(nb, threadIdx.x, ra, o, kk);
            stage_point(nb, kBlock + threadIdx.x, rb, o, kk);
        }
        __syncthreads();
    }
    float2* __restrict__ out = colpart + (int64_t)blockIdx.y * ncap + n0;
#pragma unroll
    for (int u = 0; u < kOwn; ++u) {
        const float st = xor_sum(s[u]);
        const float tt = xor_max(tm[u]);
        if (lane < 16) {
            // t = kk d^2 + L  ->  d^2 = (t - L) / kk; kept slightly high: the next E-step's cull bound wants an upper bound
            float dmin = fmaxf((tt - off[u]) / kk, 0.f);
            dmin = fmaf(dmin, 2.0e-4f, dmin) + 1.0e-12f;
            out[16 * u + lane] = make_float2(tt == -INFINITY ? INFINITY : dmin, st);
        }
    }
}

// ---- sweep 2 on the matrix cores: p1, px and the sigma2 residual of cpd.py:84-87 ----------------------------------
// Per (row tile, target tile):
//   D = mfma16x16x4([x' 1], [-2kk z'; kk|z'|^2], kk|x'|^2 + b)      lane l, reg r: pair (n = 4 (l/16) + r, m = l % 16)
//   P = exp2(D)
//   acc += mfma4x4x1(P[r], [x' 1][n][c = l % 4])  r = 0..3          block l/4: rows m = 4 ((l/4) % 4) + i, columns c
//   e   += P[r] |x'_n|^2
// Output plane blockIdx.y, relative to the workgroup's origin o (stored in rorig[row block of 512]): p1,
// u' = sum P (x - o), e' = sum P |x - o|^2 - k_row_moments' residual form with o as the reference point.
__global__ __launch_bounds__(kBlock) void k_rowpass_mfma(const float4* __restrict__ z4, const float4* __restrict__ tgt4,
                                                         int chunks_per_seg, int64_t n_total,
                                                         const double* __restrict__ params, float* __restrict__ rowpart,
                                                         int64_t mcap, float4* __restrict__ rorig) {
    __shared__ float stage[2][kChunkTiles * kPlane];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t m0wg = (int64_t)blockIdx.x * kWgPoints, m0 = m0wg + wv * (16 * kOwn);
    const float kk = (float)(-kLog2e / (2.0 * params[13]));
    float4 o = z4[m0wg];
    o.w = 0.f;
    const int k = lane >> 4, j = lane & 15, cc = lane & 3;
    float bz[kOwn], e[kOwn];
    f32x4 acc[kOwn];
#pragma unroll
    for (int t = 0; t < kOwn; ++t) {
        const float4 z = z4[m0 + 16 * t + j];
        const float zx = z.x - o.x, zy = z.y - o.y, zz = z.z - o.z;
        const float zsq = fmaf(zz, zz, fmaf(zy, zy, zx * zx));
        const float m2 = -2.f * kk;
        bz[t] = sel4(k, m2 * zx, m2 * zy, m2 * zz, kk * zsq);
        e[t] = 0.f;
        acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const int64_t c0 = (int64_t)blockIdx.y * chunks_per_seg;
    const int64_t nchunks = (n_total + kChunk - 1) / kChunk;
    const int64_t c1 = c0 + chunks_per_seg < nchunks ? c0 + chunks_per_seg : nchunks;
    float4 ra = tgt4[c0 * kChunk + threadIdx.x], rb = tgt4[c0 * kChunk + kBlock + threadIdx.x];
    stage_point(stage[0], threadIdx.x, ra, o, kk);
    stage_point(stage[0], kBlock + threadIdx.x, rb, o, kk);
    __syncthreads();
    for (int64_t c = c0; c < c1; ++c) {
        const float* __restrict__ buf = stage[(c - c0) & 1];
        const bool more = c + 1 < c1;
        if (more) {
            ra = tgt4[(c + 1) * kChunk + threadIdx.x];
            rb = tgt4[(c + 1) * kChunk + kBlock + threadIdx.x];
        }
        // software pipeline as in the column pass: the distance MFMA of the next pair runs under this pair's
        // exponentials and contraction MFMAs
        float a1 = buf[lane];
        f32x4 cx = *reinterpret_cast<const f32x4*>(buf + 64 + 4 * k);
        f32x4 d = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bz[0], cx, 0, 0, 0);
        for (int t = 0; t < kChunkTiles; ++t) {
            const float* __restrict__ tb = buf + t * kPlane;
            const float* __restrict__ tn = buf + (t + 1 < kChunkTiles ? t + 1 : t) * kPlane;
            const float a1n = tn[lane];
            const f32x4 cxn = *reinterpret_cast<const f32x4*>(tn + 64 + 4 * k);
            const f32x4 xs = *reinterpret_cast<const f32x4*>(tb + 80 + 4 * k);
            const f32x4 b2 = *reinterpret_cast<const f32x4*>(tb + 16 * cc + 4 * k);  // plane cc: x', y', z' or ones
#pragma unroll
            for (int u = 0; u < kOwn; ++u) {
                const f32x4 dn = u + 1 < kOwn ? __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bz[u + 1], cx, 0, 0, 0)
                                              : __builtin_amdgcn_mfma_f32_16x16x4f32(a1n, bz[0], cxn, 0, 0, 0);
                const float p0 = exp2r(d[0]), p1 = exp2r(d[1]), p2 = exp2r(d[2]), p3 = exp2r(d[3]);
                acc[u] = __builtin_amdgcn_mfma_f32_4x4x1f32(p0, b2[0], acc[u], 0, 0, 0);
                acc[u] = __builtin_amdgcn_mfma_f32_4x4x1f32(p1, b2[1], acc[u], 0, 0, 0);
                acc[u] = __builtin_amdgcn_mfma_f32_4x4x1f32(p2, b2[2], acc[u], 0, 0, 0);
                acc[u] = __builtin_amdgcn_mfma_f32_4x4x1f32(p3, b2[3], acc[u], 0, 0, 0);
                e[u] = fmaf(p3, xs[3], fmaf(p2, xs[2], fmaf(p1, xs[1], fmaf(p0, xs[0], e[u]))));
                d = dn;
            }
            a1 = a1n;
            cx = cxn;
        }
        if (more) {
            float* __restrict__ nb = stage[(c + 1 - c0) & 1];
            stage_point(nb, threadIdx.x, ra, o, kk);
            stage_point(nb, kBlock + threadIdx.x, rb, o, kk);
        }
        __syncthreads();
    }
    float* __restrict__ out = rowpart + (int64_t)blockIdx.y * 5 * mcap + m0;
#pragma unroll
    for (int u = 0; u < kOwn; ++u) {
        f32x4 a = acc[u];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = xor_sum(a[i]);
        const float et = xor_sum(e[u]);
        if (lane < 16) {
            const int comp = cc == 3 ? 0 : 1 + cc;  // column 3 of [x' 1] is the row sum p1
#pragma unroll
            for (int i = 0; i < 4; ++i) out[(int64_t)comp * mcap + 16 * u + 4 * (lane >> 2) + i] = a[i];
            out[4 * mcap + 16 * u + lane] = et;
        }
    }
    if (blockIdx.y == 0 && threadIdx.x == 0) rorig[blockIdx.x] = o;
}

}  // namespace

namespace prg {

// segments of whole 512-point chunks: as many as fill the chip's workgroup slots once (4 workgroups per CU)
static int mfma_chunks_per_seg(int64_t owned_points, int64_t streamed_points, int S) {
    const int64_t chunks = ceil_div(streamed_points, kChunk);
    if (S <= 0) {
        const int64_t blocks = ceil_div(owned_points, kWgPoints);
        S = (int)std::max<int64_t>(1, std::min<int64_t>(1024 / std::max<int64_t>(blocks, 1), chunks));
    }
    return (int)ceil_div(chunks, std::min<int64_t>(S, chunks));
}

int mfma_planes(int64_t owned_points, int64_t streamed_points, int S) {
    const int cps = mfma_chunks_per_seg(owned_points, streamed_points, S);
    return (int)ceil_div(ceil_div(streamed_points, kChunk), cps);
}

void launch_colpass_mfma(prg_cpd* h, int S) {
    const int cps = mfma_chunks_per_seg(h->N, h->M, S);
    dim3 grid((unsigned)ceil_div(h->N, kWgPoints), (unsigned)ceil_div(ceil_div(h->M, kChunk), cps));
    k_colpass_mfma<<<grid, kBlock, 0, h->stream>>>(h->tgt4, h->z4, h->colmin, h->motion + ((h->estep_count - 1) & 1), cps,
                                                   h->M, h->params, h->colpart, h->Ncap);
    h->wg_col = 0;
    h->dense_pairs_col = (double)grid.x * kWgPoints * (double)ceil_div(h->M, kChunk) * kChunk;
}

void launch_rowpass_mfma(prg_cpd* h, int S) {
    const int cps = mfma_chunks_per_seg(h->M, h->N, S);
    dim3 grid((unsigned)ceil_div(h->M, kWgPoints), (unsigned)ceil_div(ceil_div(h->N, kChunk), cps));
    k_rowpass_mfma<<<grid, kBlock, 0, h->stream>>>(h->z4, h->tgt4, cps, h->N, h->params, h->rowpart, h->Mcap, h->rorig);
    h->wg_row = 0;
    h->dense_pairs_row = (double)grid.x * kWgPoints * (double)ceil_div(h->N, kChunk) * kChunk;
}

}  // namespace prg
