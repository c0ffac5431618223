// CPD EM iteration for MI355X (gfx950): E-step pair sweeps, fp64 moment reduction, device M-step.
//
// Reference behaviour (neka-nat/probreg v0.3.7):
//   E-step   probreg/cpd.py:71-88          M-step rigid  probreg/cpd.py:160-192
//   M-step affine probreg/cpd.py:219-244   transforms    probreg/transformation.py:49-50, 77-78
//
// Design: the M x N responsibility matrix is never stored.
//   k_colpass  lane owns R target columns, streams a segment of the transformed source through
//              SGPRs (scalar loads, wave-uniform), keeps an online (min d^2, sum exp2) pair.
//   k_colfinal merges the segment partials in fp64 -> b_n = -log2(den_n + c), pt1_n.
//   k_rowpass  lane owns R source rows, streams a segment of (x_n, b_n) through SGPRs and
//              accumulates p1, u = sum P (x - z), e = sum P |x - z|^2 (residual form, fp32).
//   k_row_moments  sums the segment partials per row in fp64, rebuilds px = u + p1 z and the
//              23 fp64 moments the rigid / affine M-step needs (the RCCL all-reduce payload).
//   k_mstep    one thread, fp64: 3x3 one-sided Jacobi SVD / 3x3 solve, sigma2, q.
// A RIGID iteration (prg_cpd_iterate, prg_cpd_set_moments_only) runs ONE sweep instead of two: the column
// pass carries per-column sums of the source side as well - on the matrix cores relative to a block origin (k_colpass_mfma<FUSED> ->
// k_colfinal_fused), on the vector pipe as residuals against the column's own x_n (k_colpass_cull<true> / k_colpass_queue<true> ->
// k_colfinal_resid) - and k_fused_final maps the z-side sums back to the source's frame: no row pass, no per-point block.
#include <math.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "cpd_plan.h"
#include "cpd_sweeps.h"
#include "morton.h"
#include "small_linalg.h"

#include <atomic>
#include <cstring>
#include <numeric>
#include <vector>

namespace {

constexpr double kLog2e = 1.4426950408889634;
constexpr double kEps32 = 1.1920928955078125e-07;  // np.finfo(np.float32).eps, cpd.py:81,189
constexpr int kBlock = 256;
constexpr int kMomComp = 24;

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// ---------------------------------------------------------------------------------------------
// layout / upload helpers
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_pack_cloud(const float* __restrict__ in, int64_t n, int dim,
                                                       float4* __restrict__ out, int64_t cap, float pad,
                                                       float aux, const int* __restrict__ perm) {
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= cap) return;
    float4 v;
    if (i < n) {
        const int64_t j = perm ? perm[i] : i;  // sorted position i holds original point perm[i]
        v.x = in[j * dim];
        v.y = in[j * dim + 1];
        v.z = dim > 2 ? in[j * dim + 2] : 0.f;
        v.w = aux;
    } else {
        v.x = v.y = v.z = pad;
        v.w = 0.f;
    }
    out[i] = v;
}

// sum of coordinates and of squared norms: partials [nblk][4]
__global__ __launch_bounds__(kBlock) void k_cloud_sums(const float4* __restrict__ pts, int64_t n,
                                                       double* __restrict__ part) {
    __shared__ double sh[4][4];
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    double a[4] = {0, 0, 0, 0};
    if (i < n) {
        float4 v = pts[i];
        a[0] = v.x; a[1] = v.y; a[2] = v.z;
        a[3] = (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z;
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        double s = wave_sum(a[c]);
        if (lane == 0) sh[wv][c] = s;
    }
    __syncthreads();
    if (threadIdx.x < 4) part[(int64_t)blockIdx.x * 4 + threadIdx.x] =
        sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
}

// out[off + c] = sum_b part[b][ncomp] ; one block of 1024 threads (32 slices x 32 components), ncomp <= 32
constexpr int kRedBlock = 1024;
__global__ __launch_bounds__(kRedBlock) void k_reduce_partials(const double* __restrict__ part, int nblk, int ncomp,
                                                               double* __restrict__ out, int off) {
    __shared__ double sh[32][33];
    const int c = threadIdx.x & 31, slice = threadIdx.x >> 5;
    double s = 0.0;
    if (c < ncomp)
        for (int b = slice; b < nblk; b += 32) s += part[(int64_t)b * ncomp + c];
    sh[slice][c] = s;
    __syncthreads();
    if (threadIdx.x < ncomp) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 32; ++k) t += sh[k][threadIdx.x];
        out[off + threadIdx.x] = t;
    }
}

__global__ void k_zero_doubles(double* p, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0.0;
}

// sigma2_0 = [M sum|x|^2 + N sum|y|^2 - 2 (sum x).(sum y)] / (D M N)   (math_utils.py:28-29 in closed form)
// q0 = 1 + N D / 2 log(sigma2_0)                                           (cpd.py:148)
__global__ void k_init_params(double* __restrict__ moments, const double* __restrict__ srcsum,
                              double* __restrict__ params, double m, double nglobal, int dim,
                              const double* __restrict__ init /* 16 doubles or null */) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const double* ts = moments + 24;
    double cross = ts[0] * srcsum[0] + ts[1] * srcsum[1] + ts[2] * srcsum[2];
    double total = m * ts[3] + nglobal * srcsum[3] - 2.0 * cross;
    if (init) {
        // the caller subtracted different origins from the two clouds (delta = origin_target - origin_source):
        // sum |x' - y' + delta|^2 = sum |x' - y'|^2 + 2 delta.(M sum x' - N sum y') + M N |delta|^2
        const double* dl = init + 13;
        double lin = 0.0, d2 = 0.0;
        for (int k = 0; k < 3; ++k) {
            lin += dl[k] * (m * ts[k] - nglobal * srcsum[k]);
            d2 += dl[k] * dl[k];
        }
        total += 2.0 * lin + m * nglobal * d2;
    }
    double sigma2 = total / (dim * m * nglobal);
    for (int i = 0; i < PRG_NPARAMS; ++i) params[i] = 0.0;
    if (init) {
        for (int i = 0; i < 13; ++i) params[i] = init[i];
    } else {
        params[0] = params[4] = params[8] = 1.0;
        params[12] = 1.0;
    }
    params[13] = sigma2;
    params[14] = 1.0 + nglobal * dim * 0.5 * log(sigma2);
    // the target sums have served their purpose: keep the all-reduced block bounded over the EM iterations
    for (int i = 24; i < PRG_NMOMENTS; ++i) moments[i] = 0.0;
}

// half-wave (32-lane) reductions: a cull group is 32 consecutive points = one half of a wave
__device__ __forceinline__ float half_min(float v) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float half_max(float v) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Writes the boxes of the 8 groups of 32 points this workgroup holds (one point per thread): lo.xyz, hi.xyz, max aux,
// min aux.  write_boxes == false: only the aux range is refreshed (the boxes of a static cloud were written at upload).
__device__ __forceinline__ void block_group_meta(float x, float y, float z, float wmax_in, float wmin_in,
                                                 bool write_boxes, float* __restrict__ gmeta, float* box_out = nullptr,
                                                 bool real = true) {
    float v[8];
    // (real == false: a pad; a group of real points and pads gets the box of its real points, an all-pad group the pads' own)
    v[0] = half_min(real ? x : INFINITY); v[1] = half_min(real ? y : INFINITY); v[2] = half_min(real ? z : INFINITY);
    v[3] = half_max(real ? x : -INFINITY); v[4] = half_max(real ? y : -INFINITY); v[5] = half_max(real ? z : -INFINITY);
    if (v[0] == INFINITY) {
        v[0] = v[3] = x; v[1] = v[4] = y; v[2] = v[5] = z;
    }
    v[6] = half_max(wmax_in);
    v[7] = half_min(wmin_in);
    if ((threadIdx.x & 31) == 0) {
        float* o = gmeta + ((int64_t)blockIdx.x * 8 + (threadIdx.x >> 5)) * 8;
#pragma unroll
        for (int c = 0; c < 8; ++c)
            if (write_boxes || c >= 6) o[c] = v[c];
    }
    if (box_out)
#pragma unroll
        for (int c = 0; c < 6; ++c) box_out[c] = v[c];
}

// z = scale * L y + t in fp64, rounded once to fp32 (transformation.py:49-50 / 77-78).  The same kernel measures
// how far the source moved since the previous E-step (cull bound of k_colpass_cull) and writes the group
// boxes of the transformed cloud.  grid = ceil(M / 256), one point per thread (pad-only blocks keep their static boxes).
__global__ __launch_bounds__(kBlock) void k_transform_linear(const float4* __restrict__ src4, float4* __restrict__ z4,
                                                             int64_t m, const double* __restrict__ params,
                                                             unsigned* __restrict__ motion, int slot,
                                                             float* __restrict__ gmeta,
                                                             const float* __restrict__ srcw,
                                                             const double* __restrict__ disp,
                                                             float* __restrict__ cmeta) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    float moved = 0.f;
    float4 o;
    if (i < m) {
        const double s = params[12];
        float4 y = src4[i];
        double yx = y.x, yy = y.y, yz = y.z;
        if (disp) {  // BCPD: z = s R (y + v_hat) + t  (CombinedTransformation, transformation.py)
            yx += disp[i * 3];
            yy += disp[i * 3 + 1];
            yz += disp[i * 3 + 2];
        }
        o.x = (float)(s * (params[0] * yx + params[1] * yy + params[2] * yz) + params[9]);
        o.y = (float)(s * (params[3] * yx + params[4] * yy + params[5] * yz) + params[10]);
        o.z = (float)(s * (params[6] * yx + params[7] * yy + params[8] * yz) + params[11]);
        // weight a_m as an extra squared distance: a_m exp(-d2 / 2 sigma2) = exp(-(d2 + q_m) / 2 sigma2)
        o.w = srcw ? (float)(-2.0 * params[13] * (double)srcw[i]) : 0.f;
        const float4 old = z4[i];
        const float dx = o.x - old.x, dy = o.y - old.y, dz = o.z - old.z;
        moved = sqrtf(dx * dx + dy * dy + dz * dz) * 1.000001f;
    } else {
        o.x = o.y = o.z = prg::kSrcPad;
        o.w = 0.f;
    }
    z4[i] = o;
    // non-negative floats order like their bit patterns: one atomicMax per wave into this E-step's slot; the other
    // slot (next E-step's) is cleared here - nobody touches it until the next launch of this kernel
    // (one atomic per workgroup: ~1600 same-address atomics from every wave cost more than the rest of the kernel)
    __shared__ float wave_moved[kBlock / 64];
    __shared__ float half_box[kBlock / 32][6];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) moved = fmaxf(moved, __shfl_xor(moved, off, 64));
    if ((threadIdx.x & 63) == 0) wave_moved[threadIdx.x >> 6] = moved;
    // the boxes of the block's 8 groups of 32 points, and - the block IS one 256-point chunk of the stream - their union: the box
    // of the chunk (zchunk; level 1 of the owner sweep's hierarchy, cpd_sweeps_owner.hip, in every regime)
    float gb[6];
    block_group_meta(o.x, o.y, o.z, 0.f, 0.f, true, gmeta, gb, i < m);
    if ((threadIdx.x & 31) == 0)
#pragma unroll
        for (int c = 0; c < 6; ++c) half_box[threadIdx.x >> 5][c] = gb[c];
    __syncthreads();
    if (threadIdx.x == 0) {
        float mv = wave_moved[0];
#pragma unroll
        for (int k = 1; k < kBlock / 64; ++k) mv = fmaxf(mv, wave_moved[k]);
        if (mv > 0.f) atomicMax(motion + slot, __float_as_uint(mv));
    }
    if (threadIdx.x < 6 && cmeta) {
        float v = half_box[0][threadIdx.x];
#pragma unroll
        for (int k = 1; k < kBlock / 32; ++k) v = threadIdx.x < 3 ? fminf(v, half_box[k][threadIdx.x]) : fmaxf(v, half_box[k][threadIdx.x]);
        cmeta[(int64_t)blockIdx.x * 8 + threadIdx.x] = v;
    }
    if (i == 0) {
        motion[slot ^ 1] = 0u;
        motion[4 + slot] = 0u;  // k_colfinal of THIS E-step collects the largest column minimum here
    }
}

// bounding box (+ range of .w) of every group of 32 consecutive points -> meta[g][8] = lo.xyz, hi.xyz, max w, min w
__global__ __launch_bounds__(kBlock) void k_group_meta(const float4* __restrict__ pts, int64_t ngroups,
                                                       float* __restrict__ meta) {
    const int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (g >= ngroups) return;
    const float4* p = pts + g * prg::kGroup;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY}, wmax = -INFINITY,
          wmin = INFINITY;
    // a group that holds real points AND pads (the last real group of a cloud) gets the box of its real points: a pad never
    // contributes to any sum, and a box that reaches out to the pads (1e18 away) makes its owner need every cell of the other cloud
    const bool mixed = fabsf(p[0].x) < 1e17f && !(fabsf(p[prg::kGroup - 1].x) < 1e17f);  // (pads fill the tail)
    for (int k = 0; k < prg::kGroup; ++k) {
        const float4 v = p[k];
        if (mixed && !(fabsf(v.x) < 1e17f)) continue;
        lo[0] = fminf(lo[0], v.x); hi[0] = fmaxf(hi[0], v.x);
        lo[1] = fminf(lo[1], v.y); hi[1] = fmaxf(hi[1], v.y);
        lo[2] = fminf(lo[2], v.z); hi[2] = fmaxf(hi[2], v.z);
        wmax = fmaxf(wmax, v.w);
        wmin = fminf(wmin, v.w);
    }
    float* o = meta + g * 8;
    o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2];
    o[3] = hi[0]; o[4] = hi[1]; o[5] = hi[2];
    o[6] = wmax;
    o[7] = wmin;
}

// (the two pair sweeps live in cpd_sweeps_packed.hip / cpd_sweeps_scalar.hip / cpd_sweeps_mfma.hip)

// Consumers of a sweep over the work queue (cpd_sweeps_queue.hip): the partial results of a block of 128 owned points sit
// in the slots of its units; chunk[b][c] = (first slot, units) for every chunk of 32 stream segments.  Walking the chunks
// and their units in order gives every block a fixed summation order, wherever the atomics placed the units.
struct QueueView {
    const int2* chunk;  // null: the sweep did not run over the queue
    int nchunk;
    int* ctrl;          // reset for the next E-step by the consumer's first thread
    int pop_start, cap_soft;
};
__device__ __forceinline__ void queue_reset(const QueueView& q) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const int fine = q.ctrl[0] < q.cap_soft ? q.ctrl[0] : q.cap_soft;
        q.ctrl[8] = fine;                   // fine / coarse units of this sweep (prg_cpd_pair_counts)
        q.ctrl[9] = q.ctrl[7];
        q.ctrl[2] = q.ctrl[0];              // the next build sizes its units from this sweep's count ...
        q.ctrl[3] = q.ctrl[6];              // ... and unit size
        q.ctrl[0] = 0;                      // it appends from slot 0 ...
        q.ctrl[7] = 0;
        q.ctrl[1] = q.pop_start;            // ... and its waves take the first `pop_start` units without asking
    }
}

// Merge the S partial (min, sum) pairs of each column in fp64; apply cpd.py:78-82:
//   den == 0 -> eps32 (then the whole column of P is 0/eps = 0), den += c.
// Writes b_n = -log2(den_n) into tgt4[n].w so that P_mn = exp2(kk d2 + b_n), and pt1_n = den/(den+c).
__global__ __launch_bounds__(kBlock) void k_colfinal(float4* __restrict__ tgt4, const float2* __restrict__ colpart,
                                                     int nseg, int64_t ncap, int64_t n, float* __restrict__ pt1,
                                                     const double* __restrict__ params, double w, double m_over_n,
                                                     int dim, float* __restrict__ colmin, float* __restrict__ colmin_g,
                                                     float* __restrict__ gmeta, int seed_mode,
                                                     unsigned* __restrict__ stat, int slot, const QueueView qv,
                                                     double* __restrict__ xpart) {
    const int64_t i_own = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (qv.chunk) queue_reset(qv);
    float b = 0.f;  // pads keep b = 0
    float cmin = 0.f;  // pads do not widen the seed
    // Lanes past the end redo the last column and store nothing: the wave stays whole, which the queue consumer below
    // (entries handed round with readlane) relies on.
    const bool valid = i_own < n;
    const int64_t i = valid ? i_own : n - 1;
    {
    const double sigma2 = params[13];
    const float kkf = (float)(-kLog2e / (2.0 * sigma2));
    // Online merge of the segment partials (dmin_s, sum_s), 8 in flight per lane.  sum_s is relative to the
    // exponent offset off_s = col_offset(kk, dmin_s) the column pass used (reproduced bit for bit), i.e. the true
    // segment sum is sum_s * 2^(-off_s).  The rescale factors are <= 1 and go through v_exp_f32 like the sweeps'
    // own (their 1-ulp error is far below the fp32 sums they multiply); the running sum is fp64.
    float gmin = INFINITY, goff = INFINITY;  // goff = smallest offset seen = offset of the column minimum
    double ssum = 0.0;
    if (seed_mode) {
        // matrix-core column pass: every segment's sum is relative to the SAME offset, known before the sweep
        // (prg::col_seed_offset from the previous E-step's minimum, still in colmin[i], and this E-step's motion)
        // (seed_mode 2: the first E-step's sweep ran without offsets)
        goff = seed_mode == 2 ? 0.f : prg::col_seed_offset(kkf, colmin[i], __uint_as_float(stat[slot]));
        for (int s0 = 0; s0 < nseg; ++s0) {
            const float2 p = colpart[(int64_t)s0 * ncap + i];
            gmin = fminf(gmin, p.x);
            ssum += (double)p.y;
        }
    } else if (qv.chunk) {
        // the slots of the column's block of 128, [unit][128] (min, sum) pairs, chunk by chunk, unit by unit: the wave's
        // 64 columns share the block, so lane c fetches chunk c's (first slot, units) entry once and the walk hands them
        // round with readlane; four units (four loads) are in flight per trip
        const int2* __restrict__ cb = qv.chunk + (i >> 7) * qv.nchunk;
        const int lane = threadIdx.x & 63;
        for (int c0 = 0; c0 < qv.nchunk; c0 += 64) {
            const int2 mine = c0 + lane < qv.nchunk ? cb[c0 + lane] : make_int2(0, 0);
            const int lim = qv.nchunk - c0 < 64 ? qv.nchunk - c0 : 64;
            int c = -1, left = 0, next = 0;
            auto next_slot = [&]() -> int {  // (wave-uniform)
                while (left == 0) {
                    if (++c >= lim) return -1;
                    next = __builtin_amdgcn_readlane(mine.x, c);
                    left = __builtin_amdgcn_readlane(mine.y, c);
                }
                --left;
                return next++;
            };
            for (;;) {
                int sl[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) sl[q] = next_slot();
                if (sl[0] < 0) break;
                float2 p[4];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    p[q] = sl[q] < 0 ? make_float2(INFINITY, 0.f) : colpart[(int64_t)sl[q] * 128 + (i & 127)];
                const float cm = fminf(fminf(p[0].x, p[1].x), fminf(p[2].x, p[3].x));
                if (cm < gmin) {
                    const float noff = prg::col_offset(kkf, cm);
                    ssum *= (double)__builtin_amdgcn_exp2f(noff - goff);
                    gmin = cm;
                    goff = noff;
                }
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (p[q].y != 0.f) ssum += (double)(p[q].y * __builtin_amdgcn_exp2f(goff - prg::col_offset(kkf, p[q].x)));
                if (sl[3] < 0) break;
            }
        }
    } else
    for (int s0 = 0; s0 < nseg; s0 += 8) {
        float2 p[8];
#pragma unroll
        for (int k = 0; k < 8; ++k)
            p[k] = (s0 + k < nseg) ? colpart[(int64_t)(s0 + k) * ncap + i] : make_float2(INFINITY, 0.f);
        float cm = p[0].x;
#pragma unroll
        for (int k = 1; k < 8; ++k) cm = fminf(cm, p[k].x);
        if (cm < gmin) {
            const float noff = prg::col_offset(kkf, cm);
            ssum *= (double)__builtin_amdgcn_exp2f(noff - goff);  // first chunk: 0 * exp2(-inf) = 0
            gmin = cm;
            goff = noff;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)  // culled segments are empty (sum 0, min = inf or their seed bound)
            if (p[k].y != 0.f) ssum += (double)(p[k].y * __builtin_amdgcn_exp2f(goff - prg::col_offset(kkf, p[k].x)));
    }
    const double den = ssum * exp2(-(double)goff);  // underflows to 0 exactly where fp64 exp() does
    double c = 0.0;  // uniform (outlier) term of cpd.py:78-79; w == 0 is the common case and fp64 pow() is not free
    if (w > 0.0) c = pow(2.0 * M_PI * sigma2, dim * 0.5) * (w / (1.0 - w) * m_over_n);
    float p;
    if (den == 0.0) {
        b = -INFINITY;
        p = 0.f;
    } else {
        const double tot = den + c;
        b = (float)(-log2(tot));
        p = (float)(den / tot);
    }
    if (valid) {
        reinterpret_cast<float*>(tgt4 + i)[3] = b;
        pt1[i] = p;
        colmin[i] = gmin;  // min_m |x_n - z_m|^2 of this E-step: seed of the next column pass' cull bound
        cmin = gmin;
    } else {
        b = 0.f;
    }
    // (sum_n pt1_n |x_n|^2, sum_n pt1_n) of this workgroup's columns, for a row pass that does not carry the residual sums
    if (xpart) {
        const float4 xf = tgt4[i];
        const double ps = valid ? (double)p : 0.0;
        const double xs = ps * ((double)xf.x * xf.x + (double)xf.y * xf.y + (double)xf.z * xf.z);
        __shared__ double xsum[kBlock / 64][2];
        const double wx = wave_sum(xs), wp = wave_sum(ps);
        if ((threadIdx.x & 63) == 0) {
            xsum[threadIdx.x >> 6][0] = wx;
            xsum[threadIdx.x >> 6][1] = wp;
        }
        __syncthreads();
        if (threadIdx.x < 2) {
            double t = xsum[0][threadIdx.x];
#pragma unroll
            for (int k = 1; k < kBlock / 64; ++k) t += xsum[k][threadIdx.x];
            xpart[2 * (int64_t)blockIdx.x + threadIdx.x] = t;
        }
    }
    }
    // per group of 32 columns: the largest of these minima - what a wave of the next column pass needs for its seed
    {
        const float gm = half_max(cmin);
        if ((threadIdx.x & 31) == 0) colmin_g[(int64_t)blockIdx.x * (kBlock / 32) + (threadIdx.x >> 5)] = gm;
        // largest column minimum of the whole shard: the host's bracket check for the next matrix-core column pass
        __shared__ float wg_max[kBlock / 32];
        if ((threadIdx.x & 31) == 0) wg_max[threadIdx.x >> 5] = gm;
        __syncthreads();
        if (threadIdx.x == 0) {
            float mx = wg_max[0];
#pragma unroll
            for (int k = 1; k < kBlock / 32; ++k) mx = fmaxf(mx, wg_max[k]);
            if (mx > 0.f) atomicMax(stat + 4 + slot, __float_as_uint(mx));  // (+inf orders above every finite value)
        }
    }
    // refresh the b_n range of this workgroup's 8 groups (their boxes are static)
    if (gmeta) block_group_meta(0.f, 0.f, 0.f, b, b, false, gmeta);
}

// ---------------------------------------------------------------------------------------------
// fp64 moment reduction: per row, then per block -> mompart[nblk][24]
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void block_reduce_store(double (&a)[kMomComp], double* __restrict__ mompart) {
    __shared__ double sh[4][kMomComp];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < kMomComp; ++c) {
        const double s = wave_sum(a[c]);
        if (lane == 0) sh[wv][c] = s;
    }
    __syncthreads();
    if (threadIdx.x < kMomComp)
        mompart[(int64_t)blockIdx.x * kMomComp + threadIdx.x] =
            sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
}

// ---------------------------------------------------------------------------------------------
// Fused single sweep of a rigid EM iteration (cpd_sweeps_mfma.hip, k_colpass_mfma<FUSED>): per column n the planes hold
// (min d^2, A, Bx, By, Bz, E) with A = sum_m K, B = sum_m K (z_m - o), E = sum_m K |z_m - o|^2, K = exp2(kk d^2 + L_n), o the
// origin of the column's 512-block.  This kernel is k_colfinal (den_n, pt1_n, b_n, the seeds of the next column pass) AND the
// moment kernel: with q_n = pt1_n / A_n the column contributes
//   [0] pt1   [1..3] pt1 x   [4..6] pz = q B + pt1 o   [7..15] x pz^T   [16] q (E + 2 o.B) + pt1 |o|^2   [22] pt1 |x|^2
// (block partials in mompart; k_fused_final sums them and maps the z-side sums back to the source's own frame).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_colfinal_fused(float4* __restrict__ tgt4, const float* __restrict__ fpart, int nseg,
                                                           int64_t ncap, int64_t n, float* __restrict__ pt1,
                                                           const double* __restrict__ params, double w, double m_over_n, int dim,
                                                           float* __restrict__ colmin, float* __restrict__ colmin_g,
                                                           float* __restrict__ gmeta, int seed_mode, unsigned* __restrict__ stat,
                                                           int slot, const float4* __restrict__ corig,
                                                           double* __restrict__ mompart) {
    const int64_t i_own = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const bool valid = i_own < n;
    const int64_t i = valid ? i_own : n - 1;
    const double sigma2 = params[13];
    const float kkf = (float)(-kLog2e / (2.0 * sigma2));
    // every plane's sums are relative to the SAME exponent offset, known before the sweep (as in k_colfinal's seed mode)
    const float goff = seed_mode == 2 ? 0.f : prg::col_seed_offset(kkf, colmin[i], __uint_as_float(stat[slot]));
    float gmin = INFINITY;
    double A = 0.0, B[3] = {0.0, 0.0, 0.0}, E = 0.0;
    for (int s0 = 0; s0 < nseg; ++s0) {
        const float* __restrict__ q = fpart + (int64_t)s0 * 6 * ncap + i;
        gmin = fminf(gmin, q[0]);
        A += (double)q[ncap];
        B[0] += (double)q[2 * ncap];
        B[1] += (double)q[3 * ncap];
        B[2] += (double)q[4 * ncap];
        E += (double)q[5 * ncap];
    }
    const double den = A * exp2(-(double)goff);  // underflows to 0 exactly where fp64 exp() does
    double c = 0.0;
    if (w > 0.0) c = pow(2.0 * M_PI * sigma2, dim * 0.5) * (w / (1.0 - w) * m_over_n);
    float b, p;
    double pd = 0.0, qn = 0.0;
    if (den == 0.0) {  // cpd.py:81: den = eps32, the column of P is all zero
        b = -INFINITY;
        p = 0.f;
    } else {
        const double tot = den + c;
        b = (float)(-log2(tot));
        pd = den / tot;
        p = (float)pd;
        qn = pd / A;
    }
    float cmin = 0.f;
    double a[kMomComp];
#pragma unroll
    for (int k = 0; k < kMomComp; ++k) a[k] = 0.0;
    if (valid) {
        reinterpret_cast<float*>(tgt4 + i)[3] = b;
        pt1[i] = p;
        colmin[i] = gmin;
        cmin = gmin;
        const float4 xf = tgt4[i], of = corig[i / prg::kMfmaWgPoints];
        const double x[3] = {xf.x, xf.y, xf.z}, o[3] = {of.x, of.y, of.z};
        double pz[3], ob = 0.0, oo = 0.0, xx = 0.0;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            pz[k] = qn * B[k] + pd * o[k];
            ob += o[k] * B[k];
            oo += o[k] * o[k];
            xx += x[k] * x[k];
        }
        a[0] = pd;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            a[1 + r] = pd * x[r];
            a[4 + r] = pz[r];
#pragma unroll
            for (int k = 0; k < 3; ++k) a[7 + 3 * r + k] = x[r] * pz[k];
        }
        a[16] = qn * (E + 2.0 * ob) + pd * oo;
        a[22] = pd * xx;
    } else {
        b = 0.f;
    }
    block_reduce_store(a, mompart);
    {  // per group of 32 columns the largest of the minima, and the shard's largest (exactly as k_colfinal)
        const float gm = half_max(cmin);
        if ((threadIdx.x & 31) == 0) colmin_g[(int64_t)blockIdx.x * (kBlock / 32) + (threadIdx.x >> 5)] = gm;
        __shared__ float wg_max[kBlock / 32];
        if ((threadIdx.x & 31) == 0) wg_max[threadIdx.x >> 5] = gm;
        __syncthreads();
        if (threadIdx.x == 0) {
            float mx = wg_max[0];
#pragma unroll
            for (int k = 1; k < kBlock / 32; ++k) mx = fmaxf(mx, wg_max[k]);
            if (mx > 0.f) atomicMax(stat + 4 + slot, __float_as_uint(mx));
        }
    }
    if (gmeta) block_group_meta(0.f, 0.f, 0.f, b, b, false, gmeta);
}

// ---------------------------------------------------------------------------------------------
// Residual-form single sweep of a rigid EM iteration on the vector pipe (k_colpass_cull<true> / k_colpass_queue<true>,
// DESIGN.md 3.1f): per column n the partials hold (min d^2, A, Ux, Uy, Uz, R) with A = sum_m K, U = sum_m K (x_n - z_m),
// R = sum_m K |x_n - z_m|^2, K = exp2(kk d^2 + off), off the offset of the partial's OWN minimum (prg::col_offset).  This
// kernel merges them online in fp64 (k_colfinal's merge with five channels), applies cpd.py:78-82 (den == 0 -> eps32, + c),
// writes b_n / pt1_n / the next column pass' seeds AND the column's share of the rigid M-step's moments - k_colfinal_fused's
// terms with the column's own x_n as the origin:  sum_m K z = x A - U,  sum_m K |z|^2 = |x|^2 A - 2 x.U + R:
//   [0] pt1   [1..3] pt1 x   [4..6] pz = pt1 x - q U   [7..15] x pz^T   [16] pt1 |x|^2 + q (R - 2 x.U)   [22] pt1 |x|^2,
// q = pt1 / A.  The sums are residuals against the CURRENT transformation (small where P is not), so sigma2 keeps the accuracy of
// the row pass' residual form at any amplification mean|x|^2 / (sigma2 D) - unlike the matrix-core fused sweep, whose
// origin is a 512-column block's.  k_fused_final maps the z-side sums back to the source's frame.
// QUEUE: the partials are the slots of the block's units, [unit][6][128], walked chunk by chunk, unit by unit (fixed order);
// otherwise planes [plane][6][ncap] with one touched flag per (128-column block, plane) behind them.
// ---------------------------------------------------------------------------------------------
template <bool QUEUE>
__global__ __launch_bounds__(kBlock) void k_colfinal_resid(float4* __restrict__ tgt4, const float* __restrict__ fpart, int nseg,
                                                           int64_t ncap, int64_t n, float* __restrict__ pt1,
                                                           const double* __restrict__ params, double w, double m_over_n, int dim,
                                                           float* __restrict__ colmin, float* __restrict__ colmin_g,
                                                           float* __restrict__ gmeta, unsigned* __restrict__ stat, int slot,
                                                           const unsigned char* __restrict__ colflag, const QueueView qv,
                                                           double* __restrict__ mompart, int flag_shift) {
    const int64_t i_own = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (QUEUE) queue_reset(qv);
    const bool valid = i_own < n;
    const int64_t i = valid ? i_own : n - 1;  // lanes past the end redo the last column and store nothing: the wave stays whole
    const int lane = threadIdx.x & 63;
    const double sigma2 = params[13];
    const float kkf = (float)(-kLog2e / (2.0 * sigma2));
    float gmin = INFINITY, goff = INFINITY;
    double A = 0.0, U[3] = {0.0, 0.0, 0.0}, R = 0.0;
    auto merge = [&](float pm, float a, float u0, float u1, float u2, float r) {
        if (pm < gmin) {
            const float noff = prg::col_offset(kkf, pm);
            const double f = (double)__builtin_amdgcn_exp2f(noff - goff);  // first partial: 0 * exp2(-inf) = 0
            A *= f; U[0] *= f; U[1] *= f; U[2] *= f; R *= f;
            gmin = pm;
            goff = noff;
        }
        if (a != 0.f) {
            const float f = __builtin_amdgcn_exp2f(goff - prg::col_offset(kkf, pm));
            A += (double)(a * f);
            U[0] += (double)(u0 * f);
            U[1] += (double)(u1 * f);
            U[2] += (double)(u2 * f);
            R += (double)(r * f);
        }
    };
    if (QUEUE) {
        const int2* __restrict__ cb = qv.chunk + (i >> 7) * qv.nchunk;
        for (int c0 = 0; c0 < qv.nchunk; c0 += 64) {
            const int2 mine = c0 + lane < qv.nchunk ? cb[c0 + lane] : make_int2(0, 0);
            const int lim = qv.nchunk - c0 < 64 ? qv.nchunk - c0 : 64;
            int c = -1, left = 0, next = 0;
            auto next_slot = [&]() -> int {  // (wave-uniform)
                while (left == 0) {
                    if (++c >= lim) return -1;
                    next = __builtin_amdgcn_readlane(mine.x, c);
                    left = __builtin_amdgcn_readlane(mine.y, c);
                }
                --left;
                return next++;
            };
            for (;;) {  // four units (24 loads) in flight per trip, merged in order
                int sl[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) sl[q] = next_slot();
                if (sl[0] < 0) break;
                float v[4][6];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float* __restrict__ o = fpart + (int64_t)(sl[q] < 0 ? sl[0] : sl[q]) * 768 + (i & 127);
#pragma unroll
                    for (int k = 0; k < 6; ++k) v[q][k] = o[128 * k];
                }
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (sl[q] >= 0) merge(v[q][0], v[q][1], v[q][2], v[q][3], v[q][4], v[q][5]);
                if (sl[3] < 0) break;
            }
        }
    } else {
        // the wave's 64 columns lie in one 128-column block: lane l looks at the flag of plane p0 + l, a ballot gives the live planes
        // (flag_shift: log2 of the columns a flag stands for - 7, or 6 when the owner sweep ran with one column per lane)
        const unsigned char* __restrict__ fl = colflag + (i >> flag_shift) * nseg;
        for (int p0 = 0; p0 < nseg; p0 += 64) {
            unsigned long long live = __ballot(p0 + lane < nseg && fl[p0 + lane] != 0);
            while (live) {
                const int s0 = p0 + __builtin_ctzll(live);
                live &= live - 1;
                const int s1 = live ? p0 + __builtin_ctzll(live) : -1;
                live &= live - 1;  // (0 stays 0)
                const float* __restrict__ o0 = fpart + (int64_t)s0 * 6 * ncap + i;
                const float* __restrict__ o1 = fpart + (int64_t)(s1 < 0 ? s0 : s1) * 6 * ncap + i;
                float v0[6], v1[6];
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    v0[k] = o0[k * ncap];
                    v1[k] = o1[k * ncap];
                }
                merge(v0[0], v0[1], v0[2], v0[3], v0[4], v0[5]);
                if (s1 >= 0) merge(v1[0], v1[1], v1[2], v1[3], v1[4], v1[5]);
            }
        }
    }
    const double den = A * exp2(-(double)goff);  // underflows to 0 exactly where fp64 exp() does
    double c = 0.0;
    if (w > 0.0) c = pow(2.0 * M_PI * sigma2, dim * 0.5) * (w / (1.0 - w) * m_over_n);
    float b, p;
    double pd = 0.0, qn = 0.0;
    if (den == 0.0) {  // cpd.py:81: den = eps32, the column of P is all zero
        b = -INFINITY;
        p = 0.f;
    } else {
        const double tot = den + c;
        b = (float)(-log2(tot));
        pd = den / tot;
        p = (float)pd;
        qn = pd / A;
    }
    float cmin = 0.f;
    double a[kMomComp];
#pragma unroll
    for (int k = 0; k < kMomComp; ++k) a[k] = 0.0;
    if (valid) {
        reinterpret_cast<float*>(tgt4 + i)[3] = b;
        pt1[i] = p;
        colmin[i] = gmin;
        cmin = gmin;
        const float4 xf = tgt4[i];
        const double x[3] = {xf.x, xf.y, xf.z};
        double pz[3], xu = 0.0, xx = 0.0;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            pz[k] = pd * x[k] - qn * U[k];
            xu += x[k] * U[k];
            xx += x[k] * x[k];
        }
        a[0] = pd;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            a[1 + r] = pd * x[r];
            a[4 + r] = pz[r];
#pragma unroll
            for (int k = 0; k < 3; ++k) a[7 + 3 * r + k] = x[r] * pz[k];
        }
        a[16] = pd * xx + qn * (R - 2.0 * xu);
        a[22] = pd * xx;
    } else {
        b = 0.f;
    }
    block_reduce_store(a, mompart);
    {  // per group of 32 columns the largest of the minima, and the shard's largest (exactly as k_colfinal)
        const float gm = half_max(cmin);
        if ((threadIdx.x & 31) == 0) colmin_g[(int64_t)blockIdx.x * (kBlock / 32) + (threadIdx.x >> 5)] = gm;
        __shared__ float wg_max[kBlock / 32];
        if ((threadIdx.x & 31) == 0) wg_max[threadIdx.x >> 5] = gm;
        __syncthreads();
        if (threadIdx.x == 0) {
            float mx = wg_max[0];
#pragma unroll
            for (int k = 1; k < kBlock / 32; ++k) mx = fmaxf(mx, wg_max[k]);
            if (mx > 0.f) atomicMax(stat + 4 + slot, __float_as_uint(mx));
        }
    }
    if (gmeta) block_group_meta(0.f, 0.f, 0.f, b, b, false, gmeta);
}

// block partials of k_colfinal_fused -> MOMENTS in the layout k_mstep reads.  The sweep saw the TRANSFORMED source
// z = s R y + t; the rigid M-step wants sums over y: y = R^T (z - t) / s, so
//   Sy = R^T (Sz - S0 t) / s,   Sxy = (Sxz - Sx t^T) R / s,   tr Syy = (tr Szz - 2 t.Sz + S0 |t|^2) / s^2
// (R orthonormal: a rotation - checked on the host for the initial one, true by construction afterwards; the M-step only
// takes the trace of Syy for a rigid fit, cpd.py:179-182, so it goes to [16] and the other five entries stay 0).
__global__ __launch_bounds__(kRedBlock) void k_fused_final(const double* __restrict__ part, int nblk,
                                                           const double* __restrict__ params, double* __restrict__ moments) {
    __shared__ double sh[32][33];
    __shared__ double m[32];
    const int c = threadIdx.x & 31, slice = threadIdx.x >> 5;
    double sum = 0.0;
    if (c < kMomComp)
        for (int b = slice; b < nblk; b += 32) sum += part[(int64_t)b * kMomComp + c];
    sh[slice][c] = sum;
    __syncthreads();
    if (threadIdx.x < 32) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 32; ++k) t += sh[k][threadIdx.x];
        m[threadIdx.x] = t;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    const double S0 = m[0], s = params[12];
    const double t[3] = {params[9], params[10], params[11]};
    double tsz = 0.0, tt = 0.0;
    moments[0] = S0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        moments[1 + i] = m[1 + i];
        tsz += t[i] * m[4 + i];
        tt += t[i] * t[i];
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        double sy = 0.0;
#pragma unroll
        for (int k = 0; k < 3; ++k) sy += params[3 * k + j] * (m[4 + k] - S0 * t[k]);  // (R^T)[j][k] = R[k][j]
        moments[4 + j] = sy / s;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            double v = 0.0;
#pragma unroll
            for (int k = 0; k < 3; ++k) v += (m[7 + 3 * i + k] - m[1 + i] * t[k]) * params[3 * k + j];
            moments[7 + 3 * i + j] = v / s;
        }
    moments[16] = (m[16] - 2.0 * tsz + S0 * tt) / (s * s);
#pragma unroll
    for (int k = 17; k < 22; ++k) moments[k] = 0.0;
    moments[22] = m[22];
    moments[23] = 0.0;
}

__device__ __forceinline__ void row_moment_terms(double (&a)[kMomComp], double p1, const double (&px)[3],
                                                 const double (&y)[3]) {
    a[0] = p1;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        a[1 + i] = px[i];
        a[4 + i] = p1 * y[i];
#pragma unroll
        for (int j = 0; j < 3; ++j) a[7 + 3 * i + j] = px[i] * y[j];
    }
    a[16] = p1 * y[0] * y[0];
    a[17] = p1 * y[0] * y[1];
    a[18] = p1 * y[0] * y[2];
    a[19] = p1 * y[1] * y[1];
    a[20] = p1 * y[1] * y[2];
    a[21] = p1 * y[2] * y[2];
}

__global__ __launch_bounds__(kBlock) void k_row_moments(const float* __restrict__ rowpart, int nseg, int64_t mcap,
                                                        int64_t m, const float4* __restrict__ src4,
                                                        const float4* __restrict__ z4, double* __restrict__ rowacc,
                                                        double* __restrict__ mompart,
                                                        const unsigned char* __restrict__ rowflag,
                                                        const float4* __restrict__ rorig, const QueueView qv, int lean) {
    if (qv.chunk) queue_reset(qv);
    // lean: the row pass left no residual sums e (k_rowpass_mfma<LEAN>: planes p1, ux, uy, uz only) - component 22,
    // sum_n pt1_n |x_n|^2, is filled in from the column side afterwards (k_xpx_columns)
    const bool has_e = !lean;
    double a[kMomComp];
#pragma unroll
    for (int c = 0; c < kMomComp; ++c) a[c] = 0.0;
    // grid-stride over the rows: few workgroups -> few partials for the single-block final reduction.  The trip count is the
    // same for the 64 lanes of a wave (`valid` masks the rows past the end): the queue consumer below talks across lanes.
    const int lane = threadIdx.x & 63;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i - lane < m; i += (int64_t)gridDim.x * kBlock) {
        const bool valid = i < m;
        double p1 = 0, u[3] = {0, 0, 0}, e = 0;
        if (qv.chunk) {
            // The slots of the row's block of 128, [unit][5][128]: the wave's 64 rows share the block, so lane c fetches chunk
            // c's table entry once and the (first slot, units) pairs are handed round with readlane; the units are then
            // taken FOUR at a time (20 loads in flight), in order: chunk by chunk, unit by unit.
            const int2* __restrict__ cb = qv.chunk + (i >> 7) * qv.nchunk;
            for (int c0 = 0; c0 < qv.nchunk; c0 += 64) {
                const int2 mine = c0 + lane < qv.nchunk ? cb[c0 + lane] : make_int2(0, 0);
                const int lim = qv.nchunk - c0 < 64 ? qv.nchunk - c0 : 64;
                int c = -1, left = 0, next = 0;
                auto next_slot = [&]() -> int {  // (wave-uniform)
                    while (left == 0) {
                        if (++c >= lim) return -1;
                        next = __builtin_amdgcn_readlane(mine.x, c);
                        left = __builtin_amdgcn_readlane(mine.y, c);
                    }
                    --left;
                    return next++;
                };
                for (;;) {
                    int sl[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) sl[q] = next_slot();
                    if (sl[0] < 0) break;
                    float v[4][5];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float* __restrict__ o = rowpart + (int64_t)(sl[q] < 0 ? sl[0] : sl[q]) * 640 + (i & 127);
#pragma unroll
                        for (int k = 0; k < 5; ++k) v[q][k] = o[128 * k];
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (sl[q] < 0) continue;
                        p1 += (double)v[q][0];
                        u[0] += (double)v[q][1];
                        u[1] += (double)v[q][2];
                        u[2] += (double)v[q][3];
                        e += (double)v[q][4];
                    }
                    if (sl[3] < 0) break;
                }
            }
        }
        // (128-row wave block, segment) partials the culled row pass never touched are absent (neither written nor
        // read): the wave fetches its block's 64 flag bytes once and walks the set bits (<= 64 planes)
        uint64_t live = qv.chunk ? 0ull : (nseg >= 64 ? ~0ull : ((1ull << nseg) - 1ull));
        if (rowflag && !qv.chunk) {
            const int wb = __builtin_amdgcn_readfirstlane((int)(i >> 7));
            const uint4* __restrict__ f = reinterpret_cast<const uint4*>(rowflag + (int64_t)wb * 64);
            uint64_t bits = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint4 v = f[q];
                const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int bb = 0; bb < 4; ++bb)
                        bits |= (uint64_t)((w4[k] >> (8 * bb)) & 1u) << (q * 16 + k * 4 + bb);
            }
            live &= bits;
        }
        while (live) {
            // two live segments per trip: ten independent loads in flight
            const int s = __builtin_ctzll(live);
            live &= live - 1;
            const int s2 = live ? __builtin_ctzll(live) : s;
            const double k2 = live ? 1.0 : 0.0;
            live &= live - 1;
            const float* __restrict__ o = rowpart + (int64_t)s * 5 * mcap + i;
            const float* __restrict__ o2 = rowpart + (int64_t)s2 * 5 * mcap + i;
            const float v0 = o[0], v1 = o[mcap], v2 = o[2 * mcap], v3 = o[3 * mcap], v4 = has_e ? o[4 * mcap] : 0.f;
            const float w0 = o2[0], w1 = o2[mcap], w2 = o2[2 * mcap], w3 = o2[3 * mcap], w4 = has_e ? o2[4 * mcap] : 0.f;
            p1 += (double)v0 + k2 * (double)w0;
            u[0] += (double)v1 + k2 * (double)w1;
            u[1] += (double)v2 + k2 * (double)w2;
            u[2] += (double)v3 + k2 * (double)w3;
            e += (double)v4 + k2 * (double)w4;
        }
        // reference point of the residual sums: the row's own z_m (VALU sweeps) or the origin of its 512-row block
        // (matrix-core sweeps) - the identities below hold for any reference
        const float4 zf = rorig ? rorig[i / prg::kMfmaWgPoints] : z4[i], yf = src4[i];
        const double z[3] = {zf.x, zf.y, zf.z};
        const double y[3] = {yf.x, yf.y, yf.z};
        double px[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) px[k] = u[k] + p1 * z[k];  // exact identity: sum P x = sum P (x - z) + p1 z
        double t[kMomComp];
#pragma unroll
        for (int c = 0; c < kMomComp; ++c) t[c] = 0.0;
        row_moment_terms(t, p1, px, y);
        // sum_n pt1_n |x_n|^2 restricted to this row: sum_n P |x|^2 = p1 |z|^2 + 2 z.u + e
        t[22] = has_e ? p1 * (z[0] * z[0] + z[1] * z[1] + z[2] * z[2]) + 2.0 * (z[0] * u[0] + z[1] * u[1] + z[2] * u[2]) + e : 0.0;
        if (valid) {
#pragma unroll
            for (int c = 0; c < kMomComp; ++c) a[c] += t[c];
            rowacc[i] = p1;
            rowacc[mcap + i] = px[0];
            rowacc[2 * mcap + i] = px[1];
            rowacc[3 * mcap + i] = px[2];
        }
    }
    block_reduce_store(a, mompart);
}

// Lean matrix-core row pass: moments[22] = sum_n pt1_n |x_n|^2 from k_colfinal's per-workgroup partials (fixed order), scaled
// by (sum of the ROW sums p1) / (sum of the column sums pt1): the two differ by ~1e-6 (fp32 accumulation drops the far tail
// of a long sum), and the M-step's sigma2 subtracts quantities built from the row sums from this one.
__global__ __launch_bounds__(kBlock) void k_xpx_columns(const double* __restrict__ xpart, int nblk, double* __restrict__ moments) {
    __shared__ double sh[kBlock / 64][2];
    double x = 0.0, p = 0.0;
    for (int b = threadIdx.x; b < nblk; b += kBlock) {
        x += xpart[2 * (int64_t)b];
        p += xpart[2 * (int64_t)b + 1];
    }
    const double wx = wave_sum(x), wp = wave_sum(p);
    if ((threadIdx.x & 63) == 0) {
        sh[threadIdx.x >> 6][0] = wx;
        sh[threadIdx.x >> 6][1] = wp;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double tx = 0.0, tp = 0.0;
#pragma unroll
        for (int k = 0; k < kBlock / 64; ++k) {
            tx += sh[k][0];
            tp += sh[k][1];
        }
        moments[22] = tp > 0.0 ? tx * (moments[0] / tp) : tx;
    }
}

// Moments from explicit EstepResult arrays (public maximization_step path, cpd.py:90-93):
// rows (p1, px) -> comps 0..21 ; columns (pt1, x) -> comp 22.  grid covers max(M, N) items.
__global__ __launch_bounds__(kBlock) void k_moments_from_arrays(const double* __restrict__ pt1,
                                                                const double* __restrict__ p1,
                                                                const double* __restrict__ px, int dim, int64_t m,
                                                                int64_t n, const float4* __restrict__ src4,
                                                                const float4* __restrict__ tgt4,
                                                                const int* __restrict__ perm_src,
                                                                const int* __restrict__ perm_tgt,
                                                                double* __restrict__ mompart) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    double a[kMomComp];
#pragma unroll
    for (int c = 0; c < kMomComp; ++c) a[c] = 0.0;
    if (i < m) {
        const float4 yf = src4[i];
        const double y[3] = {yf.x, yf.y, yf.z};
        const int64_t j = perm_src ? perm_src[i] : i;  // the caller's arrays are in the original point order
        double pxi[3] = {px[j * dim], px[j * dim + 1], dim > 2 ? px[j * dim + 2] : 0.0};
        row_moment_terms(a, p1[j], pxi, y);
    }
    if (i < n) {
        const float4 xf = tgt4[i];
        const int64_t j = perm_tgt ? perm_tgt[i] : i;
        a[22] = pt1[j] * ((double)xf.x * xf.x + (double)xf.y * xf.y + (double)xf.z * xf.z);
    }
    block_reduce_store(a, mompart);
}

// The same arrays as the per-point fp64 block [4][Mcap] (p1, px) the non-rigid solve reads (kernel order).
__global__ __launch_bounds__(kBlock) void k_rowacc_from_arrays(const double* __restrict__ pt1,
                                                               const double* __restrict__ p1,
                                                               const double* __restrict__ px, int dim, int64_t m,
                                                               int64_t n, int64_t mcap,
                                                               const int* __restrict__ perm_src,
                                                               const int* __restrict__ perm_tgt,
                                                               double* __restrict__ rowacc, float* __restrict__ pt1f) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) pt1f[i] = (float)pt1[perm_tgt ? perm_tgt[i] : i];
    if (i >= m) return;
    const int64_t j = perm_src ? perm_src[i] : i;
    rowacc[i] = p1[j];
    rowacc[mcap + i] = px[j * dim];
    rowacc[2 * mcap + i] = px[j * dim + 1];
    rowacc[3 * mcap + i] = dim > 2 ? px[j * dim + 2] : 0.0;
}

// ---------------------------------------------------------------------------------------------
// device M-step (fp64, one thread)
// ---------------------------------------------------------------------------------------------
// kind: PRG_TF_RIGID (cpd.py:160-192) or PRG_TF_AFFINE (cpd.py:219-244).
__global__ void k_mstep(const double* __restrict__ mom, double* __restrict__ params, int kind, int update_scale,
                        int dim) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    // NB: every array index below is a compile-time constant after unrolling (run-time indices would push the
    // 3 x 3 arrays into scratch memory and cost ~1 us per access); the D = 2 case lives in the upper-left block
    // of the same 3 x 3 problem (z = 0 makes the third row / column of every moment vanish).
    const int d = dim;
    const double S0 = mom[0];
    double mu_x[3], mu_y[3], A[3][3], YPY[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        mu_x[i] = mom[1 + i] / S0;  // cpd.py:169
        mu_y[i] = mom[4 + i] / S0;  // cpd.py:170
    }
    // a = px^T (Y - mu_y) - mu_x (p1^T (Y - mu_y)) ; the second term is identically 0   (cpd.py:173-175)
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) A[i][j] = mom[7 + 3 * i + j] - mom[1 + i] * mu_y[j];
    const double syy[3][3] = {{mom[16], mom[17], mom[18]}, {mom[17], mom[19], mom[20]}, {mom[18], mom[20], mom[21]}};
    double tr_yp1y = 0.0, mux2 = 0.0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) YPY[i][j] = syy[i][j] - S0 * mu_y[i] * mu_y[j];  // (Y-mu)^T diag(p1) (Y-mu)
        tr_yp1y += YPY[i][i];
        mux2 += mu_x[i] * mu_x[i];
    }
    const double tr_xp1x = mom[22] - S0 * mux2;  // cpd.py:183 / 237
    double L[3][3], t[3];
    double scale = 1.0, sigma2, q;
    if (kind == PRG_TF_RIGID) {
        double U[3][3], V[3][3], sv[3];
        prg::jacobi_svd(A, d, U, V, sv);
        // rot = U diag(1,..,det(U V^T)) V^T with the correction on the smallest singular value (cpd.py:176-179)
        const double dd = prg::det3(U, d) * prg::det3(V, d);
        double c[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            bool is_min = k < d;
#pragma unroll
            for (int j = 0; j < 3; ++j)
                if (j < d && j != k && (sv[j] < sv[k] || (sv[j] == sv[k] && j < k))) is_min = false;
            c[k] = is_min ? dd : 1.0;
        }
        double tr_atr = 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                double r = 0;
#pragma unroll
                for (int k = 0; k < 3; ++k) r += c[k] * U[i][k] * V[j][k];
                L[i][j] = r;
                tr_atr += (i < d && j < d) ? A[i][j] * r : 0.0;  // trace(a^T rot), cpd.py:180
            }
        scale = update_scale ? tr_atr / tr_yp1y : 1.0;  // cpd.py:182
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            double r = 0;
#pragma unroll
            for (int j = 0; j < 3; ++j) r += L[i][j] * mu_y[j];
            t[i] = (i < d) ? mu_x[i] - scale * r : 0.0;  // cpd.py:183
        }
        if (update_scale)
            sigma2 = (tr_xp1x - scale * tr_atr) / (S0 * d);  // cpd.py:186
        else
            sigma2 = (tr_xp1x + tr_yp1y - scale * tr_atr) / (S0 * d);  // cpd.py:188 (sic)
        sigma2 = fmax(sigma2, kEps32);                                  // cpd.py:189
        q = (tr_xp1x - 2.0 * scale * tr_atr + scale * scale * tr_yp1y) / (2.0 * sigma2);
        q += d * S0 * 0.5 * log(sigma2);  // cpd.py:190-191
    } else {
        // b = solve(yp1y^T, a^T)^T : Gaussian elimination with partial pivoting (cpd.py:235) on the 3 x 3 embedding
        // [yp1y^T | a^T] with a unit diagonal in the unused dimension
        double Mx[3][6];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const bool in = i < d && j < d;
                Mx[i][j] = in ? YPY[j][i] : ((i == j) ? 1.0 : 0.0);
                Mx[i][3 + j] = in ? A[j][i] : 0.0;
            }
        double ypy_scale = 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) ypy_scale = fmax(ypy_scale, (i < d && j < d) ? fabs(YPY[i][j]) : 0.0);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
#pragma unroll
            for (int r = c + 1; r < 3; ++r) {  // bring the largest pivot candidate up by conditional row swaps
                if (fabs(Mx[r][c]) > fabs(Mx[c][c])) {
#pragma unroll
                    for (int j = 0; j < 6; ++j) { const double tmp = Mx[c][j]; Mx[c][j] = Mx[r][j]; Mx[r][j] = tmp; }
                }
            }
            // np.linalg.solve raises on a singular matrix (cpd.py:237).  The moments are fp64 sums, so a rank-deficient
            // Y^T diag(p1) Y (fewer than D + 1 supported points, coplanar ones) shows up as a pivot at round-off
            // level of the matrix scale: turn it into the non-finite result the host maps to LinAlgError
            if (c < d && !(fabs(Mx[c][c]) > 1e-12 * ypy_scale)) Mx[c][c] = 0.0;
#pragma unroll
            for (int r = c + 1; r < 3; ++r) {
                const double f = Mx[r][c] / Mx[c][c];
#pragma unroll
                for (int j = 0; j < 6; ++j) Mx[r][j] -= (j >= c) ? f * Mx[c][j] : 0.0;
            }
        }
        double Xs[3][3];
#pragma unroll
        for (int col = 0; col < 3; ++col)
#pragma unroll
            for (int r = 2; r >= 0; --r) {
                double v = Mx[r][3 + col];
#pragma unroll
                for (int j = 0; j < 3; ++j) v -= (j > r) ? Mx[r][j] * Xs[j][col] : 0.0;
                Xs[r][col] = v / Mx[r][r];
            }
        double tr_ab = 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            double r = 0;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const bool in = i < d && j < d;
                L[i][j] = in ? Xs[j][i] : ((i == j) ? 1.0 : 0.0);
                r += in ? L[i][j] * mu_y[j] : 0.0;
                tr_ab += in ? A[i][j] * L[i][j] : 0.0;  // trace(a b^T), cpd.py:238,240
            }
            t[i] = (i < d) ? mu_x[i] - r : 0.0;  // cpd.py:236
        }
        sigma2 = (tr_xp1x - tr_ab) / (S0 * d);  // cpd.py:239
        sigma2 = fmax(sigma2, kEps32);
        q = (tr_xp1x - 2.0 * tr_ab + tr_ab) / (2.0 * sigma2) + d * S0 * 0.5 * log(sigma2);  // cpd.py:242-243
    }
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) params[3 * i + j] = L[i][j];
        params[9 + i] = t[i];
    }
    params[12] = scale;
    params[13] = sigma2;
    params[14] = q;
    params[15] = S0;
    params[16] += 1.0;
}

// EstepResult materialisation helpers
// Outputs go back to the caller's point order: sorted position i holds original point perm[i].
__global__ __launch_bounds__(kBlock) void k_float_to_double(const float* __restrict__ in, double* __restrict__ out,
                                                            int64_t n, const int* __restrict__ perm) {
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) out[perm ? perm[i] : i] = in[i];
}
__global__ __launch_bounds__(kBlock) void k_scatter_double(const double* __restrict__ in, double* __restrict__ out,
                                                           int64_t n, const int* __restrict__ perm) {
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) out[perm ? perm[i] : i] = in[i];
}
// srcw[i] = (float) lw[perm[i]]  (sorted position i holds original point perm[i])
__global__ __launch_bounds__(kBlock) void k_gather_weights(const double* __restrict__ lw, int64_t m,
                                                           const int* __restrict__ perm, float* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < m) out[i] = (float)lw[perm ? perm[i] : i];
}
__global__ __launch_bounds__(kBlock) void k_pack_px(const double* __restrict__ rowacc, int64_t mcap, int64_t m,
                                                    int dim, double* __restrict__ out, const int* __restrict__ perm) {
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= m) return;
    const int64_t j = perm ? perm[i] : i;
    for (int k = 0; k < dim; ++k) out[j * dim + k] = rowacc[(int64_t)(1 + k) * mcap + i];
}
__global__ __launch_bounds__(kBlock) void k_unpack_points(const float4* __restrict__ in, int64_t m, int dim,
                                                          float* __restrict__ out, const int* __restrict__ perm) {
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= m) return;
    const float4 v = in[i];
    const int64_t j = perm ? perm[i] : i;
    out[j * dim] = v.x;
    out[j * dim + 1] = v.y;
    if (dim > 2) out[j * dim + 2] = v.z;
}

inline dim3 grid1(int64_t n) { return dim3((unsigned)prg::ceil_div(n, kBlock)); }

// Segment count for the streamed axis.  The grid should hold ~12.5k workgroups (~50k waves: enough to hide the
// scalar-load latency of the streams and, in the culled regime, short enough per-wave chains) whatever the size
// of the lane-owned cloud: C1 on one GPU -> 64 x 196 workgroups (best both dense and culled, tools/cull_floor.py);
// an 8-way target shard (12.5k local columns, 25 workgroups wide) -> ~200 short column-pass segments instead of 8
// long ones, which is what keeps a shard's E-step near 1/8 of the single-GPU time (tools/shard_profile2.py).
// Large clouds keep segments of >= 2048 streamed points (one ballot of group tests), at most 64 of them.
int auto_segments(int64_t nblk_x, int64_t stream_len, int quantum, int cap) {
    int64_t target = std::max<int64_t>(prg::ceil_div(12544, std::max<int64_t>(nblk_x, 1)),
                                       std::min<int64_t>(stream_len / 2048, 64));
    target = std::min<int64_t>(std::max<int64_t>(target, 1), cap);
    const int64_t seg = prg::round_up(prg::ceil_div(stream_len, target), quantum);
    return (int)prg::ceil_div(stream_len, seg);
}

static QueueView queue_view(const SweepQueue& q, bool active) {
    QueueView v;
    v.chunk = active ? q.chunk : nullptr;
    v.nchunk = q.nchunk;
    v.ctrl = q.ctrl;
    v.pop_start = prg::kQueueWorkgroups * (prg::kSweepBlock / 64);
    v.cap_soft = q.cap_soft;
    return v;
}

static void free_queue(SweepQueue& q) {
    for (void* p : {(void*)q.masks, (void*)q.chunk, (void*)q.units, (void*)q.ctrl, (void*)q.ucount})
        if (p) (void)hipFree(p);
    q = SweepQueue();
}

int free_plan_buffers(prg_cpd* h) {
    free_queue(h->qcol);
    free_queue(h->qrow);
    if (h->src4) (void)hipFree(h->src4);
    if (h->z4) (void)hipFree(h->z4);
    if (h->tgt4) (void)hipFree(h->tgt4);
    if (h->pt1) (void)hipFree(h->pt1);
    if (h->colpart) (void)hipFree(h->colpart);
    if (h->rowpart) (void)hipFree(h->rowpart);
    if (h->rowacc) (void)hipFree(h->rowacc);
    if (h->mompart) (void)hipFree(h->mompart);
    if (h->stage) (void)hipFree(h->stage);
    for (void* q : {(void*)h->perm_src, (void*)h->perm_tgt, (void*)h->zmeta, (void*)h->tmeta, (void*)h->colmin,
                    (void*)h->motion, (void*)h->srcw, (void*)h->wgcount, (void*)h->rorig, (void*)h->corig, (void*)h->zchunk, (void*)h->tchunk})
        if (q) (void)hipFree(q);
    h->rorig = nullptr;
    h->corig = nullptr;
    h->zchunk = h->tchunk = nullptr;
    h->wgcount = nullptr;
    h->wg_cap = 0;
    h->perm_src = h->perm_tgt = nullptr;
    h->zmeta = h->tmeta = h->colmin = nullptr;
    h->motion = nullptr;
    h->src4 = h->z4 = h->tgt4 = nullptr;
    h->pt1 = nullptr;
    h->colpart = nullptr;
    h->rowpart = nullptr;
    h->rowacc = nullptr;
    h->mompart = nullptr;
    h->stage = nullptr;
    h->stage_bytes = 0;
    return PRG_OK;
}

template <typename T>
int ensure_buffer(T** p, int64_t* have, int64_t need) {
    if (*p && *have >= need) return PRG_OK;
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    *have = 0;
    PRG_HIP(hipMalloc((void**)p, (size_t)need * sizeof(T)));
    *have = need;
    return PRG_OK;
}

// capacity: the cloud + room for the segments' rounding to 256-point multiples + prefetch slack
int cap_for(int64_t n) { return (int)prg::round_up(n + 64 * prg::kSuper + 1024, 1024); }

// Spatial order of a cloud: sorted position -> original index, as the plan's device permutation.  [r6] The kd-tree order of
// morton.h, built on the device (spatial_order.hip: ~2 ms per 100k points; PRG_SPATIAL_ORDER=kd_host: the host build, 17 ms;
// =morton: the Z-curve of rounds 1 - 5).
int morton_permutation(prg_cpd* h, const float* pts_hd, int64_t n, int dim, int** perm_dev, double* ext2 = nullptr,
                       float* box = nullptr) {
    std::vector<float> host((size_t)n * dim);
    PRG_HIP(hipMemcpy(host.data(), pts_hd, host.size() * sizeof(float), hipMemcpyDefault));
    if (ext2) {  // squared diagonal of the cloud's bounding box (scale of the dense-regime criterion); box = lo.xyz, hi.xyz
        *ext2 = 0.0;
        for (int k = 0; k < 3; ++k) {
            float lo = INFINITY, hi = -INFINITY;
            for (int64_t i = 0; i < n && k < dim; ++i) {
                lo = std::min(lo, host[(size_t)i * dim + k]);
                hi = std::max(hi, host[(size_t)i * dim + k]);
            }
            if (k >= dim) lo = hi = 0.f;
            *ext2 += (double)(hi - lo) * (double)(hi - lo);
            if (box) {
                box[k] = lo;
                box[3 + k] = hi;
            }
        }
    }
    if (*perm_dev) (void)hipFree(*perm_dev);
    *perm_dev = nullptr;
    PRG_HIP(hipMalloc((void**)perm_dev, (size_t)n * sizeof(int)));
    static const std::string order = getenv("PRG_SPATIAL_ORDER") ? getenv("PRG_SPATIAL_ORDER") : "";
    if (order != "morton" && order != "kd_host") {
        // the caller's layout, [n][dim] floats, goes to the staging buffer (where k_pack_cloud reads it anyway) and is ordered there
        PRG_TRY(prg::ensure_stage(h, (size_t)n * dim * sizeof(float)));
        PRG_HIP(hipMemcpyAsync(h->stage, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice, h->stream));
        return prg::device_kd_order((const float*)h->stage, n, dim, *perm_dev, h->stream);
    }
    const std::vector<int> perm = order == "morton" ? prg::morton_order(host.data(), n, dim) : prg::kd_order(host.data(), n, dim);
    PRG_HIP(hipMemcpy(*perm_dev, perm.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice));
    return PRG_OK;
}

template <typename T>
int ensure_exact(T** p, size_t count) {
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    PRG_HIP(hipMalloc((void**)p, count * sizeof(T)));
    return PRG_OK;
}


int mom_blocks(const prg_cpd* h) {
    int64_t items = h->M > h->N ? h->M : h->N;
    return (int)prg::ceil_div(items, kBlock);
}

int ensure_mompart(prg_cpd* h) {
    // [mom_blocks][24] block partials of the moment kernels, then [mom_blocks][2] partials of (sum pt1 |x|^2, sum pt1) (k_colfinal), then
    // 64 doubles of scratch at the very end
    int64_t need = (int64_t)mom_blocks(h) * (kMomComp + 2) + 64;
    return ensure_buffer(&h->mompart, &h->mompart_elems, need);
}

}  // namespace

namespace prg {
int ensure_stage(prg_cpd* h, size_t bytes) {
    if (h->stage && h->stage_bytes >= bytes) return PRG_OK;
    if (h->stage) {
        PRG_HIP(hipStreamSynchronize(h->stream));
        (void)hipFree(h->stage);
    }
    h->stage = nullptr;
    h->stage_bytes = 0;
    PRG_HIP(hipMalloc(&h->stage, bytes));
    h->stage_bytes = bytes;
    return PRG_OK;
}
}  // namespace prg

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

int prg_cpd_create(prg_cpd** out, int device, void* hip_stream) {
    PRG_REQUIRE(out != nullptr, PRG_ERR_INVALID, "prg_cpd_create: out is NULL");
    int count = 0;
    PRG_HIP(hipGetDeviceCount(&count));
    PRG_REQUIRE(device >= 0 && device < count, PRG_ERR_INVALID, "prg_cpd_create: device %d out of range (%d devices)",
                device, count);
    prg::DeviceGuard g(device);
    PRG_REQUIRE(g.ok, PRG_ERR_HIP, "prg_cpd_create: hipSetDevice(%d) failed", device);
    prg_cpd* h = new (std::nothrow) prg_cpd();
    PRG_REQUIRE(h != nullptr, PRG_ERR_NOMEM, "prg_cpd_create: out of host memory");
    h->device = device;
    h->stream = (hipStream_t)hip_stream;
    if (const char* eng = getenv("PRG_DENSE_ENGINE")) h->dense_engine = std::max(0, std::min(2, atoi(eng)));  // experiments
    if (const char* eng = getenv("PRG_SPARSE_ENGINE")) h->sparse_engine = std::max(0, std::min(2, atoi(eng)));
    if (const char* eng = getenv("PRG_RESID_SWEEP")) h->resid_sweep = atoi(eng) != 0;  // (A/B runs of the two-sweep sparse regime)
    hipError_t e = hipMalloc((void**)&h->state, (PRG_NMOMENTS + PRG_NPARAMS) * sizeof(double));
    if (e != hipSuccess) {
        delete h;
        prg::set_error("prg_cpd_create: hipMalloc failed: %s", hipGetErrorString(e));
        return PRG_ERR_HIP;
    }
    h->moments = h->state;
    h->params = h->state + PRG_NMOMENTS;
    k_zero_doubles<<<1, 64, 0, h->stream>>>(h->state, PRG_NMOMENTS + PRG_NPARAMS);
    *out = h;
    return PRG_OK;
}

int prg_cpd_destroy(prg_cpd* h) {
    if (!h) return PRG_OK;
    prg::DeviceGuard g(h->device);
    (void)hipStreamSynchronize(h->stream);
    free_plan_buffers(h);
    prg::nonrigid_free(h);
    if (h->state) (void)hipFree(h->state);
    if (h->pinned) (void)hipHostFree(h->pinned);
    if (h->eng_host) (void)hipHostFree(h->eng_host);
    if (h->eng_dev) (void)hipFree(h->eng_dev);
    delete h;
    return PRG_OK;
}

int prg_cpd_set_source(prg_cpd* h, const float* source_hd, int64_t m, int dim) {
    PRG_REQUIRE(h && source_hd, PRG_ERR_INVALID, "prg_cpd_set_source: NULL argument");
    PRG_REQUIRE(m > 0 && (dim == 2 || dim == 3), PRG_ERR_INVALID,
                "prg_cpd_set_source: need m > 0 and dim in {2,3} (got m=%lld dim=%d)", (long long)m, dim);
    PRG_REQUIRE(!h->have_target || h->D == dim, PRG_ERR_INVALID,
                "prg_cpd_set_source: dim %d does not match target dim %d", dim, h->D);
    prg::DeviceGuard g(h->device);
    PRG_HIP(hipStreamSynchronize(h->stream));
    const int64_t cap = cap_for(m);
    if (cap != h->Mcap) {
        if (h->src4) (void)hipFree(h->src4);
        if (h->z4) (void)hipFree(h->z4);
        if (h->rowacc) (void)hipFree(h->rowacc);
        h->src4 = h->z4 = nullptr;
        h->rowacc = nullptr;
        PRG_HIP(hipMalloc((void**)&h->src4, cap * sizeof(float4)));
        PRG_HIP(hipMalloc((void**)&h->z4, cap * sizeof(float4)));
        PRG_HIP(hipMalloc((void**)&h->rowacc, 4 * cap * sizeof(double)));
    }
    if (cap != h->Mcap || !h->zmeta) {
        PRG_TRY(ensure_exact(&h->zmeta, (size_t)(cap / prg::kGroup) * 8));
        if (!h->motion) {
            PRG_TRY(ensure_exact(&h->motion, 16));
            PRG_HIP(hipMemsetAsync(h->motion, 0, 16 * sizeof(unsigned), h->stream));
        }
        PRG_TRY(ensure_exact(&h->rorig, (size_t)(cap / prg::kMfmaWgPoints) + 4));
        PRG_TRY(ensure_exact(&h->zchunk, (size_t)(cap / prg::kSuper) * 8));
    }
    h->M = m;
    h->D = dim;
    h->Mcap = cap;
    if (h->opt_sort_src) {
        PRG_TRY(morton_permutation(h, source_hd, m, dim, &h->perm_src, &h->sext2));
    } else if (h->perm_src) {
        (void)hipFree(h->perm_src);
        h->perm_src = nullptr;
    }
    PRG_TRY(prg::ensure_stage(h, (size_t)m * dim * sizeof(float)));
    PRG_HIP(hipMemcpyAsync(h->stage, source_hd, (size_t)m * dim * sizeof(float), hipMemcpyDefault, h->stream));
    k_pack_cloud<<<grid1(cap), kBlock, 0, h->stream>>>((const float*)h->stage, m, dim, h->src4, cap, prg::kSrcPad,
                                                       0.f, h->perm_src);
    k_pack_cloud<<<grid1(cap), kBlock, 0, h->stream>>>((const float*)h->stage, m, dim, h->z4, cap, prg::kSrcPad, 0.f,
                                                       h->perm_src);
    // boxes of the whole padded array once; the per-iteration transform kernel refreshes the blocks with real points
    k_group_meta<<<grid1(cap / prg::kGroup), kBlock, 0, h->stream>>>(h->z4, cap / prg::kGroup, h->zmeta);
    PRG_HIP(hipGetLastError());
    PRG_HIP(hipStreamSynchronize(h->stream));  // the caller's buffer may be pageable host memory
    h->have_colmin = false;
    h->have_source = true;
    h->have_estep = false;
    prg::nonrigid_free(h);
    if (h->srcw) (void)hipFree(h->srcw);  // weights belong to the previous source
    h->srcw = nullptr;
    h->uniform_ratio = 0.0;
    return PRG_OK;
}

int prg_cpd_set_source_weights(prg_cpd* h, const double* log_weights_hd, double uniform_ratio) {
    PRG_REQUIRE(h && h->have_source, PRG_ERR_STATE, "prg_cpd_set_source_weights: source not set");
    PRG_REQUIRE(uniform_ratio >= 0.0, PRG_ERR_INVALID, "prg_cpd_set_source_weights: uniform_ratio must be >= 0");
    prg::DeviceGuard g(h->device);
    h->uniform_ratio = uniform_ratio;
    if (!log_weights_hd) {
        PRG_HIP(hipStreamSynchronize(h->stream));
        if (h->srcw) (void)hipFree(h->srcw);
        h->srcw = nullptr;
        return PRG_OK;
    }
    hipPointerAttribute_t attr;
    const bool on_host = hipPointerGetAttributes(&attr, log_weights_hd) != hipSuccess || attr.type != hipMemoryTypeDevice;
    (void)hipGetLastError();
    if (on_host)  // a_m > 1 would make q_m negative and break the cull bounds: normalise by the largest weight first
        for (int64_t i = 0; i < h->M; ++i)
            PRG_REQUIRE(log_weights_hd[i] <= 0.0, PRG_ERR_INVALID,
                        "prg_cpd_set_source_weights: log-weight %lld is %g, must be <= 0 (and not NaN)", (long long)i,
                        log_weights_hd[i]);
    if (!h->srcw) PRG_HIP(hipMalloc((void**)&h->srcw, (size_t)h->Mcap * sizeof(float)));
    PRG_TRY(prg::ensure_stage(h, (size_t)h->M * sizeof(double)));
    PRG_HIP(hipMemcpyAsync(h->stage, log_weights_hd, (size_t)h->M * sizeof(double), hipMemcpyDefault, h->stream));
    k_gather_weights<<<grid1(h->M), kBlock, 0, h->stream>>>((const double*)h->stage, h->M, h->perm_src, h->srcw);
    PRG_HIP(hipGetLastError());
    PRG_HIP(hipStreamSynchronize(h->stream));
    h->have_colmin = false;
    return PRG_OK;
}

int prg_cpd_set_target(prg_cpd* h, const float* target_hd, int64_t n_local, int dim, int64_t n_global) {
    PRG_REQUIRE(h && target_hd, PRG_ERR_INVALID, "prg_cpd_set_target: NULL argument");
    PRG_REQUIRE(n_local > 0 && n_global >= n_local && (dim == 2 || dim == 3), PRG_ERR_INVALID,
                "prg_cpd_set_target: need 0 < n_local <= n_global and dim in {2,3}");
    PRG_REQUIRE(!h->have_source || h->D == dim, PRG_ERR_INVALID,
                "prg_cpd_set_target: dim %d does not match source dim %d", dim, h->D);
    prg::DeviceGuard g(h->device);
    PRG_HIP(hipStreamSynchronize(h->stream));
    // (the sums of the previous target no longer describe this one: no lean row pass until prg_cpd_init_sums has run)
    if (h->tsum_local) PRG_HIP(hipMemsetAsync(h->tsum_local, 0, 4 * sizeof(double), h->stream));
    const int64_t cap = cap_for(n_local);
    if (cap != h->Ncap) {
        if (h->tgt4) (void)hipFree(h->tgt4);
        if (h->pt1) (void)hipFree(h->pt1);
        h->tgt4 = nullptr;
        h->pt1 = nullptr;
        PRG_HIP(hipMalloc((void**)&h->tgt4, cap * sizeof(float4)));
        PRG_HIP(hipMalloc((void**)&h->pt1, cap * sizeof(float)));
    }
    if (cap != h->Ncap || !h->tmeta) {
        PRG_TRY(ensure_exact(&h->tmeta, (size_t)(cap / prg::kGroup) * 8));
        PRG_TRY(ensure_exact(&h->tchunk, (size_t)(cap / prg::kSuper) * 8));
        PRG_TRY(ensure_exact(&h->corig, (size_t)(cap / prg::kMfmaWgPoints) + 4));
        PRG_TRY(ensure_exact(&h->colmin, (size_t)cap + (size_t)cap / prg::kGroup));  // + per-group maxima
        PRG_HIP(hipMemsetAsync(h->colmin, 0, ((size_t)cap + (size_t)cap / prg::kGroup) * sizeof(float), h->stream));
    }
    if (!h->motion) {
        PRG_TRY(ensure_exact(&h->motion, 16));
        PRG_HIP(hipMemsetAsync(h->motion, 0, 16 * sizeof(unsigned), h->stream));
    }
    h->N = n_local;
    h->Nglobal = n_global;
    h->D = dim;
    h->Ncap = cap;
    if (h->opt_sort_tgt) {
        PRG_TRY(morton_permutation(h, target_hd, n_local, dim, &h->perm_tgt, &h->text2, h->tbox));
    } else if (h->perm_tgt) {
        (void)hipFree(h->perm_tgt);
        h->perm_tgt = nullptr;
    }
    PRG_TRY(prg::ensure_stage(h, (size_t)n_local * dim * sizeof(float)));
    PRG_HIP(hipMemcpyAsync(h->stage, target_hd, (size_t)n_local * dim * sizeof(float), hipMemcpyDefault, h->stream));
    k_pack_cloud<<<grid1(cap), kBlock, 0, h->stream>>>((const float*)h->stage, n_local, dim, h->tgt4, cap,
                                                       prg::kTgtPad, 0.f, h->perm_tgt);
    // the target never moves: its group boxes are written once (k_colfinal refreshes the b_n range)
    k_group_meta<<<grid1(cap / prg::kGroup), kBlock, 0, h->stream>>>(h->tgt4, cap / prg::kGroup, h->tmeta);
    PRG_HIP(hipGetLastError());
    PRG_HIP(hipStreamSynchronize(h->stream));
    h->have_colmin = false;
    h->have_target = true;
    h->have_estep = false;
    return PRG_OK;
}

int prg_cpd_bind_moments(prg_cpd* h, double* moments_dev) {
    PRG_REQUIRE(h, PRG_ERR_INVALID, "prg_cpd_bind_moments: NULL handle");
    h->moments = moments_dev ? moments_dev : h->state;
    return PRG_OK;
}

int prg_cpd_moments_ptr(prg_cpd* h, double** moments_dev) {
    PRG_REQUIRE(h && moments_dev, PRG_ERR_INVALID, "prg_cpd_moments_ptr: NULL argument");
    *moments_dev = h->moments;
    return PRG_OK;
}

int prg_cpd_params_ptr(prg_cpd* h, double** params_dev) {
    PRG_REQUIRE(h && params_dev, PRG_ERR_INVALID, "prg_cpd_params_ptr: NULL argument");
    *params_dev = h->params;
    return PRG_OK;
}

int prg_cpd_set_tuning(prg_cpd* h, int r_col, int seg_col, int r_row, int seg_row) {
    PRG_REQUIRE(h, PRG_ERR_INVALID, "prg_cpd_set_tuning: NULL handle");
    auto ok_r = [](int r) { return r == 0 || r == 2 || r == 4 || r == -2 || r == -4; };
    PRG_REQUIRE(ok_r(r_col) && ok_r(r_row), PRG_ERR_INVALID,
                "prg_cpd_set_tuning: points per lane must be 0 (auto), 2, 4 (packed) or -2, -4 (scalar form)");
    PRG_REQUIRE(seg_col >= 0 && seg_col <= 1024 && seg_row >= 0 && seg_row <= 256, PRG_ERR_INVALID,
                "prg_cpd_set_tuning: segment counts must be in [0, 1024] (column pass) / [0, 256] (row pass)");
    h->r_col = r_col;
    h->seg_col = seg_col;
    h->r_row = r_row;
    h->seg_row = seg_row;
    return PRG_OK;
}

int prg_cpd_set_dense_engine(prg_cpd* h, int mode, double bound) {
    PRG_REQUIRE(h, PRG_ERR_INVALID, "prg_cpd_set_dense_engine: NULL handle");
    PRG_REQUIRE(mode >= 0 && mode <= 2, PRG_ERR_INVALID, "prg_cpd_set_dense_engine: mode must be 0, 1 or 2");
    PRG_REQUIRE(bound >= 0.0, PRG_ERR_INVALID, "prg_cpd_set_dense_engine: bound must be >= 0 (0 keeps the default)");
    h->dense_engine = mode;
    if (bound > 0.0) h->dense_bound = bound;
    h->mfma_off = false;
    h->pred_col = 1;
    h->pred_fine = 0;
    h->eng_reset = true;
    return PRG_OK;
}

int prg_cpd_last_estep_engine(prg_cpd* h, int* engine) {
    PRG_REQUIRE(h && engine, PRG_ERR_INVALID, "prg_cpd_last_estep_engine: NULL argument");
    *engine = h->last_estep_mfma ? 1 : 0;
    return PRG_OK;
}

int prg_cpd_set_sparse_engine(prg_cpd* h, int mode) {
    PRG_REQUIRE(h, PRG_ERR_INVALID, "prg_cpd_set_sparse_engine: NULL handle");
    PRG_REQUIRE(mode >= 0 && mode <= 3, PRG_ERR_INVALID, "prg_cpd_set_sparse_engine: mode must be 0 (grid of culled waves), 1 (default: owner sweep for single-sweep "
                "iterations, work queue for the two-sweep E-steps of large clouds), 2 (work queue always) or 3 (round 5's default: queue for large clouds, no owner sweep)");
    h->sparse_engine = mode;
    return PRG_OK;
}

int prg_cpd_last_estep_engines(prg_cpd* h, int* col_engine, int* row_engine) {
    PRG_REQUIRE(h && col_engine && row_engine, PRG_ERR_INVALID, "prg_cpd_last_estep_engines: NULL argument");
    *col_engine = h->last_estep_mfma ? 1 : 0;
    *row_engine = h->last_estep_row_mfma ? 1 : 0;
    return PRG_OK;
}

int prg_cpd_last_estep_lean(prg_cpd* h, int* lean) {
    PRG_REQUIRE(h && lean, PRG_ERR_INVALID, "prg_cpd_last_estep_lean: NULL argument");
    *lean = h->last_estep_row_lean ? 1 : 0;
    return PRG_OK;
}

int prg_cpd_set_moments_only(prg_cpd* h, int mode) {
    PRG_REQUIRE(h, PRG_ERR_INVALID, "prg_cpd_set_moments_only: NULL handle");
    PRG_REQUIRE(mode >= 0 && mode <= 2, PRG_ERR_INVALID, "prg_cpd_set_moments_only: mode must be 0, 1 or 2");
    h->moments_only = mode == 1;
    h->fused_in_iterate = mode != 2;
    return PRG_OK;
}

int prg_cpd_set_resid_sweep(prg_cpd* h, int on) {
    PRG_REQUIRE(h, PRG_ERR_INVALID, "prg_cpd_set_resid_sweep: NULL handle");
    h->resid_sweep = on != 0;
    return PRG_OK;
}

int prg_cpd_set_fused_factor(prg_cpd* h, double factor) {
    PRG_REQUIRE(h && factor >= 0.0, PRG_ERR_INVALID, "prg_cpd_set_fused_factor: need a handle and a factor >= 0");
    h->fused_factor = factor;
    return PRG_OK;
}

int prg_cpd_last_estep_fused(prg_cpd* h, int* fused) {
    PRG_REQUIRE(h && fused, PRG_ERR_INVALID, "prg_cpd_last_estep_fused: NULL argument");
    *fused = h->last_estep_fused ? 1 : 0;
    return PRG_OK;
}

int prg_cpd_set_stream_mode(prg_cpd* h, int on) {
    PRG_REQUIRE(h, PRG_ERR_INVALID, "prg_cpd_set_stream_mode: NULL handle");
    h->mfma_stream = on != 0;
    return PRG_OK;
}

int prg_cpd_set_lean_factor(prg_cpd* h, double factor) {
    PRG_REQUIRE(h, PRG_ERR_INVALID, "prg_cpd_set_lean_factor: NULL handle");
    h->lean_factor = factor;
    return PRG_OK;
}

int prg_cpd_set_options(prg_cpd* h, int sort_source, int sort_target, int cull) {
    PRG_REQUIRE(h, PRG_ERR_INVALID, "prg_cpd_set_options: NULL handle");
    PRG_REQUIRE(!h->have_source && !h->have_target, PRG_ERR_STATE,
                "prg_cpd_set_options: must be called before the clouds are uploaded");
    h->opt_sort_src = sort_source != 0;
    h->opt_sort_tgt = sort_target != 0;
    h->opt_cull = cull != 0;
    return PRG_OK;
}

static int ensure_engine_state(prg_cpd* h);

int prg_cpd_init_sums(prg_cpd* h) {
    PRG_REQUIRE(h && h->have_target, PRG_ERR_STATE, "prg_cpd_init_sums: target not set");
    prg::DeviceGuard g(h->device);
    PRG_TRY(ensure_mompart(h));
    const int nblk = (int)prg::ceil_div(h->N, kBlock);
    k_zero_doubles<<<1, 64, 0, h->stream>>>(h->moments, PRG_NMOMENTS);
    k_cloud_sums<<<nblk, kBlock, 0, h->stream>>>(h->tgt4, h->N, h->mompart);
    k_reduce_partials<<<1, kRedBlock, 0, h->stream>>>(h->mompart, nblk, 4, h->moments, 24);
    PRG_HIP(hipGetLastError());
    PRG_TRY(ensure_engine_state(h));
    PRG_HIP(hipMemcpyAsync(h->tsum_local, h->moments + 24, 4 * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    if (h->comm) PRG_TRY(prg::comm_all_reduce_f64(h->comm, h->moments, PRG_NMOMENTS, h->stream));  // (after the LOCAL sums were kept)
    return PRG_OK;
}

int prg_cpd_set_comm(prg_cpd* h, prg_comm* comm) {
    PRG_REQUIRE(h, PRG_ERR_INVALID, "prg_cpd_set_comm: NULL handle");
    PRG_REQUIRE(!comm || comm->device == h->device, PRG_ERR_INVALID, "prg_cpd_set_comm: the communicator lives on device %d, the plan on %d",
                comm ? comm->device : -1, h->device);
    h->comm = comm;
    return PRG_OK;
}

// R^T R = I to 1e-12 for the row-major 3 x 3 block at `lin` (what the fused single sweep's frame change assumes)
static bool is_rotation(const double* lin) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double d = 0.0;
            for (int k = 0; k < 3; ++k) d += lin[3 * k + i] * lin[3 * k + j];
            if (!(fabs(d - (i == j ? 1.0 : 0.0)) <= 1.0e-12)) return false;
        }
    return true;
}

int prg_cpd_init_params(prg_cpd* h, const double* init_params_host) {
    PRG_REQUIRE(h && h->have_source && h->have_target, PRG_ERR_STATE, "prg_cpd_init_params: clouds not set");
    prg::DeviceGuard g(h->device);
    PRG_TRY(ensure_mompart(h));
    const int nblk = (int)prg::ceil_div(h->M, kBlock);
    double* srcsum = h->mompart + (h->mompart_elems - 64);
    double* init_dev = nullptr;
    if (init_params_host) {
        init_dev = srcsum + 8;
        PRG_HIP(hipMemcpyAsync(init_dev, init_params_host, 16 * sizeof(double), hipMemcpyHostToDevice, h->stream));
    }
    k_cloud_sums<<<nblk, kBlock, 0, h->stream>>>(h->src4, h->M, h->mompart);
    k_reduce_partials<<<1, kRedBlock, 0, h->stream>>>(h->mompart, nblk, 4, srcsum, 0);
    k_init_params<<<1, 64, 0, h->stream>>>(h->moments, srcsum, h->params, (double)h->M, (double)h->Nglobal, h->D,
                                           init_dev);
    PRG_HIP(hipGetLastError());
    if (init_params_host) PRG_HIP(hipStreamSynchronize(h->stream));  // host buffer may be reused by the caller
    h->have_colmin = false;  // a new registration starts: its first column pass takes no seed from the previous one
    h->mfma_off = false;     // ... and it starts in the dense regime
    h->pred_col = 1;
    h->pred_fine = 0;
    h->pred_fused = 0;
    h->mfma_grid_fine = false;
    h->eng_reset = true;
    // the fused single sweep maps its column-side sums back through s R: R has to be a rotation (the M-step's own results are)
    h->init_rot_orthonormal = !init_params_host || is_rotation(init_params_host);
    return PRG_OK;
}

// Below how many evaluated pairs per owned point does a matrix-core sweep lose to the vector-pipe sweep?  The two-line cost
// model of DESIGN.md 3.1c (see estep_impl): segment chain of a workgroup + late start on grids deeper than the chip, against
// evenly shared blocks.  Host arithmetic only.
// Segments of a matrix-core launch once its chunk / tile masks skip work ([r5]).  The default grid fills the
// chip's 768 workgroup slots about once when there are few owned blocks (a target shard's column pass: 25 blocks x 30 segments
// of 14 chunks at 1/8 of C1) - fine while every chunk is needed, but a culled sweep then lasts as long as its busiest
// workgroup, which still needs its whole segment: rank 0 of 8 stayed at 0.31 ms from EM iteration 6 to 10 while one GPU went
// 1.78 -> 0.87 ms (profiles/r4_shard_window.log).  With >= 3 rounds of shorter segments the slots even the load out.
// 0: the default grid is already that deep (C1 on one GPU: 4.85 rounds), or PRG_MFMA_SEG pins the count.
static int mfma_fine_segments(int64_t owned, int64_t streamed, int max_planes) {
    static const int seg = getenv("PRG_MFMA_SEG") ? atoi(getenv("PRG_MFMA_SEG")) : 0;
    static const bool off = getenv("PRG_MFMA_FINE_GRID") && atoi(getenv("PRG_MFMA_FINE_GRID")) == 0;
    if (seg || off) return seg;
    const int64_t blocks = prg::ceil_div(owned, prg::kMfmaWgPoints), chunks = prg::ceil_div(streamed, 256);
    int64_t want = prg::ceil_div(3 * 768 + 256, blocks);
    want = std::min<int64_t>(std::min<int64_t>(want, chunks / 4), max_planes);
    return want > prg::mfma_planes(owned, streamed, 0) ? (int)want : 0;
}

// (the model and its constants describe the DEFAULT grid, which is what the crossovers were measured on; the finer grid of a
// culling sweep only makes the matrix cores faster near the crossover - leaving at this bound is then slightly early, never late)
static double engine_leave_below(int64_t owned, int64_t streamed, double tau, double delta, double c_v) {
    static const int seg = getenv("PRG_MFMA_SEG") ? atoi(getenv("PRG_MFMA_SEG")) : 0;
    // (segments as round 4 cut them: the rule the constants were fitted with, see mfma_chunks_per_seg_model)
    const int cps_i = seg ? prg::mfma_chunks_per_seg(owned, streamed, seg) : prg::mfma_chunks_per_seg_model(owned, streamed);
    const double cps = (double)cps_i;
    const double wgs = (double)prg::ceil_div(owned, prg::kMfmaWgPoints) * (double)prg::ceil_div(prg::ceil_div(streamed, 256), cps_i);
    const double late = std::max(0.0, 1.0 - 768.0 / wgs) * tau / (768.0 * 512.0 * 256.0);
    return std::max(0.0, cps * tau + delta) / (c_v - late) / (double)owned;  // pairs per owned point
}
static double engine_col_bound(int64_t m, int64_t n_local) { return engine_leave_below(n_local, m, 12.0e-6, -15.0e-6, 0.200e-12); }
// Row pass, round 4: what competes near the crossover is the LEAN matrix-core row pass (no residual sums: 14.5 us per chunk
// instead of 19.2) against vector-pipe sweeps that skip at 2^-48 - re-measured from identical states on the surface at
// 30k / 50k / 100k / 250k points and on rank 0 of 2 / 4 / 8 at 100k (profiles/r4_engine_switch_*.log): the two cross at
// 18.6k / 13.4k / 18k / 37.5k and 12.4k / 6.3k / 5.3k evaluated targets per source point; tau 14.5 us, c_v 0.25 ps, delta -20 us
// put the bound within x1.24 of every one of them (round 3's constants left 2-2.5x too early after those two changes).  Where the
// row pass cannot run lean (amplification above the lean factor, prg_cpd_set_lean_factor(0), no prg_cpd_init_sums) round 3's
// constants apply: the decision kernel, which knows, picks between the two bounds (EngineArgs::r_row_bound / r_row_bound_full).
static double engine_row_bound(int64_t m, int64_t n_local, bool lean = true) {
    return lean ? engine_leave_below(m, n_local, 14.5e-6, -20.0e-6, 0.250e-12) : engine_leave_below(m, n_local, 19.2e-6, 8.0e-6, 0.233e-12);
}

int prg_cpd_engine_bounds(int64_t m, int64_t n_local, double* col_bound, double* row_bound) {
    PRG_REQUIRE(m > 0 && n_local > 0 && col_bound && row_bound, PRG_ERR_INVALID, "prg_cpd_engine_bounds: need m, n_local > 0 and two outputs");
    *col_bound = engine_col_bound(m, n_local);
    *row_bound = engine_row_bound(m, n_local);
    return PRG_OK;
}

// Device / mapped-host state of the engine decision: the decision itself, the matrix-core sweeps' tile counters and the LOCAL
// target's sum |x|^2 (prg_cpd_init_sums keeps a copy here: the caller all-reduces the moments block it also writes it to).
static int ensure_engine_state(prg_cpd* h) {
    if (h->eng_host) return PRG_OK;
    PRG_HIP(hipHostMalloc((void**)&h->eng_host, sizeof(EngineDecision), hipHostMallocMapped | hipHostMallocCoherent));
    memset(h->eng_host, 0, sizeof(EngineDecision));
    PRG_HIP(hipHostGetDevicePointer((void**)&h->eng_host_dev, h->eng_host, 0));
    const size_t bytes = sizeof(EngineDecision) + 2 * sizeof(unsigned long long) + 4 * sizeof(double);
    PRG_HIP(hipMalloc((void**)&h->eng_dev, bytes));
    PRG_HIP(hipMemsetAsync(h->eng_dev, 0, bytes, h->stream));
    h->eng_work = reinterpret_cast<unsigned long long*>(h->eng_dev + 1);
    h->tsum_local = reinterpret_cast<double*>(h->eng_work + 2);
    return PRG_OK;
}

static int estep_impl(prg_cpd* h, double w, hipEvent_t* ev) {
    PRG_REQUIRE(h && h->have_source && h->have_target, PRG_ERR_STATE, "prg_cpd_estep: clouds not set");
    PRG_REQUIRE(w >= 0.0 && w < 1.0, PRG_ERR_INVALID, "prg_cpd_estep: w must be in [0, 1) (got %g)", w);
    prg::DeviceGuard g(h->device);
    const int ra = h->r_col ? h->r_col : 2, rb = h->r_row ? h->r_row : 2;
    const int RA = ra < 0 ? -ra : ra, RB = rb < 0 ? -rb : rb;
    const int64_t nblkA = prg::ceil_div(h->N, kBlock * RA), nblkB = prg::ceil_div(h->M, kBlock * RB);
    // Culled sweeps need both clouds Morton-sorted (compact waves / groups); they walk the stream in groups of 32.
    const bool use_cull = h->opt_cull && h->perm_src && h->perm_tgt && h->r_col == 0 && h->r_row == 0;  // (segment counts stay tunable)
    // segment lengths are multiples of the loop trip (8 points, or 256 points = 8 groups); the pads absorb the
    // overshoot and the prefetch over-read of the last segment
    const int quantum = use_cull ? prg::kSuper : 8;
    int SA, SB;
    if (use_cull) {
        // Culled sweeps: a workgroup = 128 lane points x 4 consecutive segments (one per wave, merged in LDS), so
        // S segments cost S/4 partial planes.  Segments of 512 streamed points keep a wave's chain of dependent
        // scalar loads short - the bound of a sparse E-step (tools/wave_trace.py) - and 256 segments (64 planes,
        // the width of the row pass' touched-flag rows) are the cap; small problems get 256-point segments.
        // [r3] Few owned blocks (a target shard's column pass, the row pass against a short shard) leave the chip with
        // too few waves to hide a chain of 16 groups behind: below 65536 waves the segments are 256 points (8 groups) -
        // 8 ranks at C1: E-step 0.245 -> 0.213 ms (mid), 0.124 -> 0.105 ms (late), tools/shard_segments.py.
        auto cull_segments = [](int64_t lane_points, int64_t stream_len, int64_t cap) {
            int64_t s = std::min<int64_t>(prg::ceil_div(stream_len, 512), cap);
            if (s * prg::ceil_div(lane_points, 128) < 65536) s = std::min<int64_t>(prg::ceil_div(stream_len, 256), cap);
            return (int)std::max<int64_t>(s, 1);
        };
        SA = h->seg_col ? h->seg_col : cull_segments(h->N, h->M, 1024);
        SB = h->seg_row ? h->seg_row : cull_segments(h->M, h->N, 256);  // (64 planes: the width of the touched-flag rows)
    } else {
        SA = h->seg_col ? h->seg_col : auto_segments(nblkA, h->M, quantum, 256);
        SB = h->seg_row ? h->seg_row : auto_segments(nblkB, h->N, quantum, 64);
    }
    // equal segments of a whole number of quanta: seg = round_up(len / S), S = ceil(len / seg)
    auto seg_of = [quantum](int64_t len, int s) { return (int)prg::round_up(prg::ceil_div(len, s), quantum); };
    int segA = seg_of(h->M, SA), segB = seg_of(h->N, SB);
    SA = (int)prg::ceil_div(h->M, segA);
    SB = (int)prg::ceil_div(h->N, segB);
    while (SA > 1 && (int64_t)SA * segA + prg::kOverRead > h->Mcap) --SA, segA = seg_of(h->M, SA);
    while (SB > 1 && (int64_t)SB * segB + prg::kOverRead > h->Ncap) --SB, segB = seg_of(h->N, SB);
    PRG_REQUIRE((int64_t)SA * segA + prg::kOverRead <= h->Mcap && (int64_t)SB * segB + prg::kOverRead <= h->Ncap,
                PRG_ERR_STATE, "prg_cpd_estep: internal segmenting failure");
    // partial planes in HBM: one per segment, or one per four segments for the culled sweeps
    const int PA = use_cull ? (int)prg::ceil_div(SA, 4) : SA, PB = use_cull ? (int)prg::ceil_div(SB, 4) : SB;
    PRG_REQUIRE(!use_cull || PB <= 64, PRG_ERR_INVALID, "prg_cpd_estep: at most 256 row-pass segments with culling");
    // matrix-core sweeps (dense regime, decided per E-step below): segments of whole 512-point chunks, one plane each
    // ... for clouds large enough that the sweeps are worth it: below ~8k points an E-step is launch bound whatever the
    // engine, and a 512-point patch of a small cloud spans most of it (the patch-local origin buys no precision)
    const bool mfma_possible = use_cull && h->dense_engine > 0 && !h->srcw &&
                               (h->dense_engine >= 2 || (h->M >= 8192 && h->N >= 8192));
    // the fused single sweep needs: a caller that wants nothing but a rigid M-step's moments, unweighted sources, a rotation to map
    // the column-side sums back through
    const bool allow_fused = mfma_possible && h->moments_only && !h->nonrigid && !h->bcpd && h->init_rot_orthonormal;
    static const int mfma_seg = getenv("PRG_MFMA_SEG") ? atoi(getenv("PRG_MFMA_SEG")) : 0;  // 0: fill the chip once
    // ... and, once the previous sweep skipped a tenth of its pairs, in >= 3 rounds of shorter segments (mfma_fine_segments)
    const int seg_col_fine = mfma_possible ? mfma_fine_segments(h->N, h->M, 256) : 0, seg_row_fine = mfma_possible ? mfma_fine_segments(h->M, h->N, 64) : 0;
    // (buffers are sized for whichever way a matrix-core launch is cut: grid of segments - default or fine - or stream mode)
    const int PAm = mfma_possible ? std::max(std::max(prg::mfma_planes(h->N, h->M, mfma_seg), prg::mfma_planes(h->N, h->M, seg_col_fine)),
                                             prg::mfma_stream_planes(h->N, h->M)) : 0,
              PBm = mfma_possible ? std::max(std::max(prg::mfma_planes(h->M, h->N, mfma_seg), prg::mfma_planes(h->M, h->N, seg_row_fine)),
                                             prg::mfma_stream_planes(h->M, h->N)) : 0;
    // sparse regime: sweeps over a device-built work queue (cpd_sweeps_queue.hip) - partial results per unit, not per plane
    // ... when both clouds are large: the queue costs a build pass and leaves more partial results than the grid of culled
    // waves, which only pays off while a sweep is long (measured at C1: ahead with the target on 1 or 2 ranks, behind on 4 and 8)
    const bool use_queue = use_cull && (h->sparse_engine == 2 || ((h->sparse_engine == 1 || h->sparse_engine == 3) && h->M >= 32768 && h->N >= 32768));
    const int64_t qcol_elems = use_queue ? prg::queue_max_units(h->N, h->M) * 128 : 0,
                  qrow_elems = use_queue ? prg::queue_max_units(h->M, h->N) * 640 : 0;
    const int64_t fused_elems = allow_fused ? (int64_t)3 * std::max(prg::mfma_planes(h->N, h->M, mfma_seg), prg::mfma_planes(h->N, h->M, seg_col_fine)) * h->Ncap : 0;  // 6 floats per (plane, column)
    // the residual-form single sweep on the vector pipe: the same callers as the fused sweep, any sigma2, no
    // matrix cores needed - 6 floats per (plane, column) + a touched flag per (128-column block, plane), or 6 x 128 floats per unit
    const bool allow_resid = use_cull && h->resid_sweep && h->moments_only && !h->nonrigid && !h->bcpd && !h->srcw && h->init_rot_orthonormal;
    // ... which the column block's owner runs ([r6] cpd_sweeps_owner.hip: the stream dealt out over PO parts x 8 waves per 128-column
    // block, cells found through the chunk / group hierarchy; prg_cpd_set_sparse_engine(0 / 2 / 3): round 5's grid / queue instead)
    static const bool owner_env_off = getenv("PRG_OWNER_SWEEP") && atoi(getenv("PRG_OWNER_SWEEP")) == 0;
    const bool use_owner = allow_resid && h->sparse_engine == 1 && !owner_env_off;
    const int PO = use_owner ? prg::owner_planes(h->N, h->M) : 0;
    // (sized for the engine that can run: with the work queue the vector pipe's column pass never goes through the grid of planes)
    const int64_t resid_elems = !allow_resid ? 0 : use_owner ? (int64_t)3 * PO * h->Ncap + (prg::ceil_div(h->N, 64) * PO + 64) / 8 + 8
                                : use_queue ? 3 * qcol_elems : (int64_t)3 * PA * h->Ncap + (prg::ceil_div(h->N, 128) * PA + 64) / 8 + 8;
    PRG_TRY(ensure_buffer(&h->colpart, &h->colpart_elems,
                          std::max<int64_t>(std::max<int64_t>(std::max<int64_t>((int64_t)std::max(PA, PAm) * h->Ncap, qcol_elems), fused_elems), resid_elems)));
    PRG_TRY(ensure_buffer(&h->rowpart, &h->rowpart_elems,
                          std::max<int64_t>((int64_t)std::max(PB, PBm) * 5 * h->Mcap + (h->Mcap >> 7) * 16, qrow_elems)));  // + touched flags: 64 bytes per 128 rows
    PRG_TRY(ensure_mompart(h));
    if (use_queue) PRG_TRY(prg::prepare_queues(h));
    if (use_cull) {  // per-workgroup counters of evaluated (wave, group) blocks (prg_cpd_pair_counts)
        const int64_t need = std::max<int64_t>(
            std::max<int64_t>(std::max<int64_t>(prg::ceil_div(h->N, 128) * PA, prg::ceil_div(h->N, 64) * PO), prg::ceil_div(h->M, 128) * PB),
            std::max<int64_t>(prg::ceil_div(h->N, prg::kMfmaWgPoints) * PAm, prg::ceil_div(h->M, prg::kMfmaWgPoints) * PBm));
        if (need > h->wg_cap) {
            if (h->wgcount) {
                PRG_HIP(hipStreamSynchronize(h->stream));
                (void)hipFree(h->wgcount);
            }
            h->wgcount = nullptr;
            h->wg_cap = 0;
            PRG_HIP(hipMalloc((void**)&h->wgcount, (size_t)need * 2 * sizeof(unsigned)));
            h->wg_cap = need;
        }
    }

    if (ev) PRG_HIP(hipEventRecord(ev[0], h->stream));
    const int slot = (int)(h->estep_count & 1);
    ++h->estep_count;
    // one fused kernel: transform, source motion, group boxes of the transformed cloud.  Non-rigid: z = y + G W with the
    // parameter block's identity linear part (the fp64 sum is rounded once, transformation.py:101-102)
    const double* disp = h->bcpd ? h->W : nullptr;
    if (h->nonrigid) PRG_TRY(prg::nonrigid_displacement(h, &disp));
    k_transform_linear<<<(unsigned)prg::ceil_div(h->M, kBlock), kBlock, 0, h->stream>>>(
        h->src4, h->z4, h->M, h->params, h->motion, slot, h->zmeta, h->srcw, disp, h->zchunk);  // pad-only blocks are static
    // Dense regime on the matrix cores?  Decided per E-step from numbers only the device has at this point - sigma2, the
    // source motion of this transform, the largest column minimum of the previous E-step - so the device
    // decides (last thread of k_chunk_meta_bbox) and the host neither reads back nor synchronises: it launches the column
    // pass of the engine the PREVIOUS E-step used right behind the decision kernel (guarded: the launch returns at once if
    // the decision names the other engine), then polls the mapped mailbox while that launch runs and enqueues the rest of
    // the E-step behind it - the queue never drains.  Once sigma2 has fallen to where the culled vector sweeps skip most
    // of the pairs the registration stays on them and nothing is asked any more.
    bool use_mfma = false, row_mfma = false;  // column pass / row pass on the matrix cores
    bool first_mfma = false;                  // ... column pass without seeds (first E-step of a registration)
    bool fine_cull = false;                   // ... with the per-wave group tests (some groups can be skipped by now)
    bool row_lean = false;                    // ... row pass without its residual sums (k_rowpass_mfma<LEAN>)
    bool col_launched = false;
    bool fused = false;                       // ... ONE sweep for the whole E-step (rigid M-step moments from the column side)
    bool resid = false;                       // ... ONE sweep on the vector pipe: the residual-form column pass (k_colpass_cull<true> / k_colpass_queue<true>)
    const bool cull_seed = h->have_colmin && !h->srcw;  // the seed bound assumes unweighted distances
    const bool ask = mfma_possible && !h->mfma_off;
    if (ev && !ask) PRG_HIP(hipEventRecord(ev[1], h->stream));
    h->wg_col_pairs = h->wg_row_pairs = 128.0 * prg::kGroup;  // a (wave, group) block of the culled vector-pipe sweeps
    if (ask) {
        PRG_TRY(ensure_engine_state(h));
        EngineArgs ea;
        // size of the problem: the (replicated) source's bounding box or the local target's, whichever is larger - a
        // target shard is a small patch, and every rank should leave the dense regime at the same sigma2
        ea.ext2 = std::max(h->sext2, h->text2);
        // The matrix-core sweeps stay while they evaluate enough pairs per owned point - counted by the sweeps themselves,
        // one E-step back, so the switch follows the clouds' shape, their density and the size of this rank's shard
        // instead of a fit in sigma2.  The count P (pairs) is compared with a two-line cost model of the engines
        // (tools/mfma_vs_valu.py, profiles/r3_engine_switch_*.log):
        //   matrix cores:  a workgroup owns 512 points and a segment of `cps` 256-point chunks of the other cloud, tau per
        //                  chunk; in the dense regime some workgroup still needs its whole segment - cps x tau however
        //                  much the others cull - and when the grid is deeper than the chip's 768 workgroup slots that
        //                  workgroup may start late: + (1 - 768 / workgroups) x P x tau / (768 x 512 x 256)
        //   vector pipe:   the evaluated 128 x 32 blocks are shared out evenly: c_v x P
        // plus a difference `delta` of the fixed costs.  Leave when the vector pipe is shorter:
        //   P < (cps x tau + delta) / (c_v - (1 - 768 / workgroups) x tau / (768 x 512 x 256))
        // with  column pass  tau 12 us    c_v 0.200 ps  delta -15 us
        //       row pass     tau 19.2 us  c_v 0.233 ps  delta  +8 us
        // - per-kernel constants of this chip, the same for every cloud: surface, volume and 10:1:1 clouds of 12k ... 400k
        // points and 1/2, 1/4, 1/8 shards of 100k all cross over within one EM iteration of what this predicts.
        static const double r_col_env = getenv("PRG_ENGINE_RCOL") ? atof(getenv("PRG_ENGINE_RCOL")) : 0.0;
        static const double r_row_env = getenv("PRG_ENGINE_RROW") ? atof(getenv("PRG_ENGINE_RROW")) : 0.0;
        ea.r_col_bound = r_col_env > 0.0 ? r_col_env : h->dense_bound > 0.0 ? h->dense_bound : engine_col_bound(h->M, h->N);
        // one fused sweep against the vector pipe's two: it stays ahead further down than the matrix-core column pass alone does
        // (measured at C1, profiles/r4_fused_lower_bound.log: with the dense regime's lower end at 1.0 / 0.7 / 0.5 / 0.35 / 0.25 of the
        // column pass' own bound the window runs at 792 / 815 / 840 / 831 / 830 it/s (+-2 %): half of that bound is where the
        // gain levels off; with it, and the fused factor of 256, C1 runs fused through EM iteration 14)
        // [r5] what the fused sweep competes with below the dense regime is ONE vector-pipe sweep too (the residual-form column pass,
        // DESIGN.md 3.1f), whose per-pair cost is above the plain column pass' the bound was fitted on - as the fused sweep's is
        // above the matrix-core column pass': same command, lower end at 0.5 / 0.75 / 1.0 / 1.4 of the column pass' bound:
        // 828 / 841 / 855 / 859 it/s (profiles/r5_fused_lower_bound.log); 1.4 hands over at EM iteration 12 of C1 (0.64 -> 0.57 ms)
        static const double fused_scale = getenv("PRG_FUSED_RCOL_SCALE") ? atof(getenv("PRG_FUSED_RCOL_SCALE")) : -1.0;
        // [r6] with the clouds in kd-tree order and the owner sweep below it the C1 window is flat from 1.0 to 2.8 (906 / 909 / 912 / 910
        // it/s at 1.0 / 1.4 / 2.0 / 2.8, profiles/r6_fused_lower_bound.log: the optimum is bracketed); target shards, whose ranks leave
        // the matrix cores at their own iteration, do better the later they leave: 8 ranks 4.10 -> 3.99 ms per window at 1.4 -> 1.0
        // (0.7: 4.02), 4 ranks 6.51 -> 6.42.  1.0 it is.
        ea.r_col_bound_fused = ea.r_col_bound * (fused_scale > 0.0 ? fused_scale : allow_resid ? 1.0 : 0.5);
        ea.r_row_bound = r_row_env > 0.0 ? r_row_env : engine_row_bound(h->M, h->N, true);
        ea.r_row_bound_full = r_row_env > 0.0 ? r_row_env : engine_row_bound(h->M, h->N, false);  // (the device knows which applies)
        ea.streamed_col = (double)h->M;
        ea.streamed_row = (double)h->N;
        // the first sweep over the work queue after the matrix cores has no previous build to size its units from: about
        // `bound` pairs per owned point are needed then - 32 groups per unit unless that overfills the queue (>= 250k points)
        auto first_unit = [](double bound, int64_t owned, int64_t streamed) {
            const double groups = std::min(bound, (double)streamed) * (double)owned / (128.0 * prg::kGroup);
            int q = 32;
            while (groups / q > 0.75 * prg::kQueueMaxUnits && q < 256) q *= 2;
            return q;
        };
        h->q_first_col = first_unit(ea.r_col_bound, h->N, h->M);
        h->q_first_row = first_unit(ea.r_row_bound, h->M, h->N);
        ea.owned_col = (double)h->N;
        ea.owned_row = (double)h->M;
        ea.work = h->eng_work;
        ea.tsum = h->tsum_local;  // (prg_cpd_init_sums: sums of the LOCAL target; zeros if it was never called: not lean)
        ea.dim = h->D;
        static const double lean_env = getenv("PRG_LEAN_FACTOR") ? atof(getenv("PRG_LEAN_FACTOR")) : -1.0;
        // (tools/lean_error.py, profiles/r4_lean_error_rigid_100k_*.log: with the row-sum scaling of k_xpx_columns sigma2 stays
        // within 2.7e-6 of the oracle's up to an amplification of 190 - 1.5e-6 at 56, 2.0e-6 at 85; tests/test_lean_gpu.py holds
        // the forced pass to 1e-5 up to 128 with w = 0 / 0.1 and on a 2-rank shard.  64 makes every matrix-core row pass of C1 lean.)
        ea.lean_factor = h->lean_factor >= 0.0 ? h->lean_factor : lean_env >= 0.0 ? lean_env : 64.0;
        ea.fused_allowed = allow_fused ? 1 : 0;
        ea.resid_allowed = allow_resid ? 1 : 0;
        static const double fused_factor_env = getenv("PRG_FUSED_FACTOR") ? atof(getenv("PRG_FUSED_FACTOR")) : -1.0;
        ea.fused_factor = fused_factor_env >= 0.0 ? fused_factor_env : h->fused_factor;
        ea.reset = h->eng_reset ? 1 : 0;
        h->eng_reset = false;
        for (int k = 0; k < 6; ++k) ea.tbox[k] = h->tbox[k];
        ea.slot = slot;
        ea.have_colmin = h->have_colmin ? 1 : 0;
        ea.forced = h->dense_engine >= 2 ? 1 : 0;
        ea.seq = (unsigned)h->estep_count;  // (already incremented: never 0, the mailbox's initial value)
        ea.dev = h->eng_dev;
        ea.host = h->eng_host_dev;
        // chunk boxes of this E-step's transformed source (the matrix-core sweeps cull with them), its bounding box, and
        // the decision
        prg::launch_chunk_meta_bbox(h, &ea);
        if (ev) PRG_HIP(hipEventRecord(ev[1], h->stream));
        const bool pred = h->pred_col != 0, pred_fused = allow_fused && h->pred_fused != 0;
        int seg_col = h->mfma_grid_fine && seg_col_fine ? seg_col_fine : mfma_seg;  // (from the count the previous decision saw)
        if (pred_fused)  // (the single sweep of a rigid iteration, if the previous E-step ran it)
            prg::launch_fused_mfma(h, seg_col, !h->have_colmin, false, h->eng_dev);
        else if (pred)  // (stream mode if the previous decision found nothing to skip: the dense regime)
            prg::launch_colpass_mfma(h, seg_col, !h->have_colmin, false, h->eng_dev, h->mfma_stream && h->pred_fine == 0 && !h->mfma_grid_fine);
        else if (!use_queue && !use_owner)
            prg::launch_colpass_cull(h, SA, segA, cull_seed, h->eng_dev, allow_resid);
        // (pred == vector pipe with the work queue: nothing goes out ahead - inside the dense regime that engine only runs
        // when the bracket of the column minima is too wide for the matrix-core offsets, a handful of E-steps at most)
        PRG_HIP(hipGetLastError());
        // the answer: a few microseconds after the transform has finished, long before the column pass has
        volatile EngineDecision* mb = h->eng_host;
        {
            hipError_t werr;
            const bool got = prg::wait_mailbox(&mb->seq, ea.seq, h->stream, &werr);
            PRG_HIP(werr);
            PRG_REQUIRE(got, PRG_ERR_HIP, "prg_cpd_estep: the engine decision never reached the host");
        }
        use_mfma = mb->col != 0;
        first_mfma = mb->first != 0;
        row_mfma = mb->row != 0;
        fine_cull = mb->fine != 0;
        row_lean = row_mfma && mb->lean != 0;
        fused = allow_fused && mb->fused != 0;
        if (!mb->dense) h->mfma_off = true;
        // the previous matrix-core column pass skipped a tenth of its pairs: its masks are at work, cut the grid finer from here on
        h->mfma_grid_fine = !first_mfma && (double)mb->r_col < 0.9 * (double)h->M;
        static const bool debug_engine = getenv("PRG_DEBUG_ENGINE") != nullptr;
        if (debug_engine)
            fprintf(stderr, "[engine] sigma2 %.4e nk*ext2 %.1f pairs per owned point col %.0f (bound %.0f) row %.0f (bound %.0f) motion %.3e cmax %.3e nk*width %.1f "
                            "nk*far2 %.1f have_colmin %d -> col %d (first %d, launched ahead: %s) row %d fine %d\n",
                    (double)mb->sigma2, (double)mb->nk_ext2, (double)mb->r_col, ea.r_col_bound, (double)mb->r_row, ea.r_row_bound,
                    (double)mb->motion, (double)mb->cmax,
                    (double)mb->nk_width, (double)mb->nk_far2, (int)h->have_colmin, (int)use_mfma, (int)first_mfma,
                    pred == use_mfma ? "yes" : "NO", (int)row_mfma, (int)fine_cull);
        // which of the three guarded launches (fused sweep / matrix-core column pass / culled column pass) went out ahead, and
        // was it the one the decision names?
        if (pred_fused)
            col_launched = fused;
        else if (pred)
            col_launched = use_mfma && !fused;
        else
            col_launched = !use_mfma && !use_queue && !use_owner;
        h->pred_col = use_mfma ? 1 : 0;
        h->pred_fine = fine_cull ? 1 : 0;
        h->pred_fused = fused ? 1 : 0;
        seg_col = h->mfma_grid_fine && seg_col_fine ? seg_col_fine : mfma_seg;
        if (!col_launched && fused) {  // (the guarded launch has returned at once; rare: the engine changes a few times per registration)
            prg::launch_fused_mfma(h, seg_col, first_mfma, fine_cull, h->eng_dev);
            col_launched = true;
        } else if (!col_launched && use_mfma) {
            prg::launch_colpass_mfma(h, seg_col, first_mfma, fine_cull, h->eng_dev, h->mfma_stream && !fine_cull && !h->mfma_grid_fine);
            col_launched = true;
        }
    }
    // the vector pipe's column pass of an E-step that feeds nothing but a rigid M-step is the residual-form single sweep
    resid = allow_resid && !use_mfma;
    h->last_estep_mfma = use_mfma;
    // (a single-sweep E-step has no row pass: nothing to report for it)
    h->last_estep_row_mfma = row_mfma && !fused && !resid;
    h->last_estep_row_lean = row_lean && !fused && !resid;
    const bool col_owner = !col_launched && resid && use_owner;
    const bool col_queue = !col_launched && use_queue && !col_owner, row_queue = !row_mfma && use_queue;
    if (col_launched) {
    } else if (col_owner)
        prg::launch_colpass_owner(h, cull_seed, PO);
    else if (col_queue)
        PRG_TRY(prg::launch_colpass_queue(h, cull_seed, h->qcol_live ? 0 : h->q_first_col, resid));
    else if (use_cull)
        prg::launch_colpass_cull(h, SA, segA, cull_seed, nullptr, resid);
    else if (ra < 0)
        prg::launch_colpass_scalar(h, RA, SA, segA);
    else
        prg::launch_colpass_packed(h, RA, SA, segA);
    if (ev) PRG_HIP(hipEventRecord(ev[2], h->stream));
    h->last_estep_fused = fused || resid;
    if (resid) {
        // (A, U, R) per column -> den_n / pt1_n / seeds AND the moments in one merge kernel; no row pass, no per-point block
        const int nblk_f = (int)prg::ceil_div(h->N, kBlock);
        const double m_over_n = h->uniform_ratio > 0.0 ? h->uniform_ratio : (double)h->M / (double)h->Nglobal;
        if (col_queue)
            k_colfinal_resid<true><<<nblk_f, kBlock, 0, h->stream>>>(h->tgt4, reinterpret_cast<const float*>(h->colpart), 0, h->Ncap, h->N, h->pt1,
                                                                     h->params, w, m_over_n, h->D, h->colmin, h->colmin + h->Ncap, h->tmeta,
                                                                     h->motion, slot, nullptr, queue_view(h->qcol, true), h->mompart, 7);
        else
            k_colfinal_resid<false><<<nblk_f, kBlock, 0, h->stream>>>(h->tgt4, reinterpret_cast<const float*>(h->colpart), col_owner ? PO : PA, h->Ncap, h->N, h->pt1,
                                                                      h->params, w, m_over_n, h->D, h->colmin, h->colmin + h->Ncap, h->tmeta,
                                                                      h->motion, slot, prg::resid_flags(h, col_owner ? PO : PA), queue_view(h->qcol, false),
                                                                      h->mompart, col_owner && prg::owner_cols_per_lane() == 1 ? 6 : 7);
        if (ev) {
            PRG_HIP(hipEventRecord(ev[3], h->stream));
            PRG_HIP(hipEventRecord(ev[4], h->stream));
        }
        k_fused_final<<<1, kRedBlock, 0, h->stream>>>(h->mompart, nblk_f, h->params, h->moments);
        if (ev) PRG_HIP(hipEventRecord(ev[5], h->stream));
        PRG_HIP(hipGetLastError());
        if (h->comm) PRG_TRY(prg::comm_all_reduce_f64(h->comm, h->moments, kMomComp, h->stream));
        h->wg_row = 0;
        h->dense_pairs_row = 0.0;
        h->qcol_live = col_queue;
        h->qrow_live = false;
        h->have_estep = true;
        h->rowacc_valid = false;
        h->have_colmin = true;
        h->last_w = w;
        return PRG_OK;
    }
    if (fused) {
        // the single sweep has left per-column (A, B, E): den_n / pt1_n / the next E-step's seeds AND the moments come out of
        // one merge kernel; no row pass, no per-point block
        const int nblk_f = (int)prg::ceil_div(h->N, kBlock);
        k_colfinal_fused<<<nblk_f, kBlock, 0, h->stream>>>(h->tgt4, reinterpret_cast<const float*>(h->colpart), h->mfma_col_planes, h->Ncap,
                                                           h->N, h->pt1, h->params, w,
                                                           h->uniform_ratio > 0.0 ? h->uniform_ratio : (double)h->M / (double)h->Nglobal,
                                                           h->D, h->colmin, h->colmin + h->Ncap, h->tmeta, first_mfma ? 2 : 1, h->motion,
                                                           slot, h->corig, h->mompart);
        if (ev) {
            PRG_HIP(hipEventRecord(ev[3], h->stream));
            PRG_HIP(hipEventRecord(ev[4], h->stream));
        }
        k_fused_final<<<1, kRedBlock, 0, h->stream>>>(h->mompart, nblk_f, h->params, h->moments);
        if (ev) PRG_HIP(hipEventRecord(ev[5], h->stream));
        PRG_HIP(hipGetLastError());
        // (the E-step's 24 sums only: [24..27] hold the target sums prg_cpd_init_sums has already made global)
        if (h->comm) PRG_TRY(prg::comm_all_reduce_f64(h->comm, h->moments, kMomComp, h->stream));
        h->wg_row = 0;
        h->dense_pairs_row = 0.0;
        h->qcol_live = h->qrow_live = false;
        h->have_estep = true;
        h->rowacc_valid = false;
        h->have_colmin = true;
        h->last_w = w;
        return PRG_OK;
    }
    // (lean matrix-core row pass: no residual sums - sum pt1 |x|^2 goes from k_colfinal's partials to k_xpx_columns)
    double* xpart = h->mompart + (int64_t)mom_blocks(h) * kMomComp;
    k_colfinal<<<grid1(h->N), kBlock, 0, h->stream>>>(h->tgt4, h->colpart, use_mfma ? h->mfma_col_planes : PA, h->Ncap, h->N, h->pt1, h->params, w,
                                                      h->uniform_ratio > 0.0 ? h->uniform_ratio : (double)h->M / (double)h->Nglobal, h->D, h->colmin,
                                                      h->colmin + h->Ncap,
                                                      use_cull ? h->tmeta : nullptr, use_mfma ? (first_mfma ? 2 : 1) : 0, h->motion, slot,
                                                      queue_view(h->qcol, col_queue), row_lean ? xpart : nullptr);
    if (ev) PRG_HIP(hipEventRecord(ev[3], h->stream));
    if (row_mfma)
        prg::launch_rowpass_mfma(h, h->mfma_grid_fine && seg_row_fine ? seg_row_fine : mfma_seg, fine_cull, row_lean, h->mfma_stream && !fine_cull && !h->mfma_grid_fine);
    else if (row_queue)
        PRG_TRY(prg::launch_rowpass_queue(h, h->qrow_live ? 0 : h->q_first_row));
    else if (use_cull)
        prg::launch_rowpass_cull(h, SB, segB);
    else if (rb < 0)
        prg::launch_rowpass_scalar(h, RB, SB, segB);
    else
        prg::launch_rowpass_packed(h, RB, SB, segB);
    if (ev) PRG_HIP(hipEventRecord(ev[4], h->stream));
    const int nblk = (int)std::min<int64_t>(prg::ceil_div(h->M, kBlock), 1024);
    const int row_planes = row_mfma ? h->mfma_row_planes : PB;
    k_row_moments<<<nblk, kBlock, 0, h->stream>>>(h->rowpart, row_planes, h->Mcap, h->M, h->src4, h->z4, h->rowacc,
                                                  h->mompart,
                                                  use_cull ? reinterpret_cast<const unsigned char*>(h->rowpart + (int64_t)row_planes * 5 * h->Mcap)
                                                           : nullptr,
                                                  row_mfma ? h->rorig : nullptr, queue_view(h->qrow, row_queue), row_lean ? 1 : 0);
    // (folding this single-block reduction into the last-finishing workgroup of k_row_moments was measured in round 3:
    // +25 us - that workgroup's 256 threads read the ~400 partial rows through L2 in a few dependent rounds, the 1024
    // threads of this launch do it in 5 us including the launch)
    k_reduce_partials<<<1, kRedBlock, 0, h->stream>>>(h->mompart, nblk, kMomComp, h->moments, 0);
    if (row_lean) k_xpx_columns<<<1, kBlock, 0, h->stream>>>(xpart, (int)prg::ceil_div(h->N, kBlock), h->moments);
    if (ev) PRG_HIP(hipEventRecord(ev[5], h->stream));
    PRG_HIP(hipGetLastError());
    // target sharded over ranks: the one exchange step of the path - partial moments -> moments, on this stream
    if (h->comm) {
        PRG_TRY(prg::comm_all_reduce_f64(h->comm, h->moments, kMomComp, h->stream));  // (the E-step's 24 sums; see above)
        if (h->nonrigid) PRG_TRY(prg::comm_all_reduce_f64(h->comm, h->rowacc, 4 * h->Mcap, h->stream));
    }
    h->qcol_live = col_queue;
    h->qrow_live = row_queue;
    h->have_estep = true;
    h->rowacc_valid = true;
    h->have_colmin = true;  // colmin now describes the z4 of this E-step (motion is measured against it)
    h->last_w = w;
    return PRG_OK;
}


int prg_cpd_estep(prg_cpd* h, double w) { return estep_impl(h, w, nullptr); }

int prg_cpd_estep_timed(prg_cpd* h, double w, float* ms_out) {
    PRG_REQUIRE(h && ms_out, PRG_ERR_INVALID, "prg_cpd_estep_timed: NULL argument");
    prg::DeviceGuard g(h->device);
    hipEvent_t ev[6];
    for (int i = 0; i < 6; ++i) PRG_HIP(hipEventCreate(&ev[i]));
    int st = estep_impl(h, w, ev);
    if (st == PRG_OK) {
        hipError_t e = hipEventSynchronize(ev[5]);
        if (e != hipSuccess) {
            prg::set_error("prg_cpd_estep_timed: hipEventSynchronize failed: %s", hipGetErrorString(e));
            st = PRG_ERR_HIP;
        }
    }
    if (st == PRG_OK) {
        for (int i = 0; i < 5; ++i) (void)hipEventElapsedTime(&ms_out[i], ev[i], ev[i + 1]);
        (void)hipEventElapsedTime(&ms_out[5], ev[0], ev[5]);
    }
    for (int i = 0; i < 6; ++i) (void)hipEventDestroy(ev[i]);
    return st;
}

int prg_cpd_pair_counts(prg_cpd* h, double* col_pairs, double* row_pairs) {
    PRG_REQUIRE(h && h->have_estep && col_pairs && row_pairs, PRG_ERR_STATE, "prg_cpd_pair_counts: no E-step has been run");
    prg::DeviceGuard g(h->device);
    *col_pairs = h->dense_pairs_col;
    *row_pairs = h->dense_pairs_row;
    for (int pass = 0; pass < 2; ++pass) {  // sweeps over the work queue: one count of (128 x 32) blocks per unit
        const SweepQueue& q = pass ? h->qrow : h->qcol;
        if ((pass ? h->wg_row : h->wg_col) != -1) continue;
        int nu[2] = {0, 0};  // fine units [0, nu[0]), coarse units [cap_soft, cap_soft + nu[1])
        PRG_HIP(hipMemcpyAsync(nu, q.ctrl + 8, 2 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
        PRG_HIP(hipStreamSynchronize(h->stream));
        double sum = 0.0;
        for (int part = 0; part < 2; ++part) {
            if (nu[part] <= 0) continue;
            std::vector<unsigned> host((size_t)nu[part]);
            PRG_HIP(hipMemcpyAsync(host.data(), q.ucount + (part ? q.cap_soft : 0), (size_t)nu[part] * sizeof(unsigned),
                                   hipMemcpyDeviceToHost, h->stream));
            PRG_HIP(hipStreamSynchronize(h->stream));
            for (int i = 0; i < nu[part]; ++i) sum += host[(size_t)i];
        }
        *(pass ? row_pairs : col_pairs) = sum * 128.0 * prg::kGroup;
    }
    if (h->wg_col > 0 || h->wg_row > 0) {
        std::vector<unsigned> host((size_t)h->wg_cap * 2);
        PRG_HIP(hipMemcpyAsync(host.data(), h->wgcount, host.size() * sizeof(unsigned), hipMemcpyDeviceToHost, h->stream));
        PRG_HIP(hipStreamSynchronize(h->stream));
        if (h->wg_col > 0) {
            double s = 0.0;
            for (int64_t i = 0; i < h->wg_col; ++i) s += host[(size_t)i];
            *col_pairs = s * h->wg_col_pairs;
        }
        if (h->wg_row > 0) {
            double s = 0.0;
            for (int64_t i = 0; i < h->wg_row; ++i) s += host[(size_t)(h->wg_cap + i)];
            *row_pairs = s * h->wg_row_pairs;
        }
    }
    // counted blocks include the pad points that fill the last block of either cloud (C1: 1.00448e10 counted for 1e10 real
    // pairs in a dense sweep): never report more than the pairs there are
    const double all_pairs = (double)h->M * (double)h->N;
    *col_pairs = std::min(*col_pairs, all_pairs);
    *row_pairs = std::min(*row_pairs, all_pairs);
    return PRG_OK;
}

int prg_cpd_mstep(prg_cpd* h, int kind, int update_scale) {
    PRG_REQUIRE(h && h->have_source, PRG_ERR_STATE, "prg_cpd_mstep: source not set");
    PRG_REQUIRE(kind == PRG_TF_RIGID || kind == PRG_TF_AFFINE, PRG_ERR_INVALID,
                "prg_cpd_mstep: kind must be PRG_TF_RIGID or PRG_TF_AFFINE (use prg_cpd_mstep_nonrigid)");
    // a single-sweep E-step (fused / residual form) leaves tr(Y^T P1 Y) in MOMENTS[16] and zeros in [17..21]: enough for the rigid fit,
    // not for the affine one (cpd.py:230-235 needs all of Y^T diag(p1) Y)
    PRG_REQUIRE(kind == PRG_TF_RIGID || !(h->have_estep && h->last_estep_fused), PRG_ERR_STATE,
                "prg_cpd_mstep: the last E-step ran as the single sweep of a rigid iteration (prg_cpd_set_moments_only(1)): its "
                "moments do not hold Y^T diag(p1) Y, which the affine M-step needs");
    prg::DeviceGuard g(h->device);
    k_mstep<<<1, 64, 0, h->stream>>>(h->moments, h->params, kind, update_scale, h->D);
    PRG_HIP(hipGetLastError());
    // the single sweeps map their column-side sums back through s R: only a rigid fit leaves a rotation in the parameter block
    h->init_rot_orthonormal = kind == PRG_TF_RIGID;
    return PRG_OK;
}

int prg_cpd_iterate(prg_cpd* h, int kind, int update_scale, double w, int n_iter) {
    PRG_REQUIRE(h && h->have_source && h->have_target, PRG_ERR_STATE, "prg_cpd_iterate: clouds not set");
    PRG_REQUIRE(kind == PRG_TF_RIGID || kind == PRG_TF_AFFINE, PRG_ERR_INVALID,
                "prg_cpd_iterate: kind must be PRG_TF_RIGID or PRG_TF_AFFINE");
    PRG_REQUIRE(n_iter >= 0, PRG_ERR_INVALID, "prg_cpd_iterate: n_iter must be >= 0");
    prg::DeviceGuard g(h->device);
    // a rigid iteration wants nothing of its E-step but the moments: the dense regime may run the fused single sweep
    const bool keep = h->moments_only;
    if (kind == PRG_TF_RIGID && h->fused_in_iterate) h->moments_only = true;
    if (kind != PRG_TF_RIGID) h->moments_only = false;  // (an affine M-step needs all of Y^T diag(p1) Y: two sweeps, whatever the caller left set)
    int st = PRG_OK;
    for (int it = 0; it < n_iter && st == PRG_OK; ++it) {
        st = estep_impl(h, w, nullptr);  // (ends with the all-reduce when a communicator is attached)
        if (st == PRG_OK) {
            k_mstep<<<1, 64, 0, h->stream>>>(h->moments, h->params, kind, update_scale, h->D);
            h->init_rot_orthonormal = kind == PRG_TF_RIGID;  // (an affine fit leaves a general B: no single sweeps from it)
        }
    }
    h->moments_only = keep;
    PRG_TRY(st);
    PRG_HIP(hipGetLastError());
    return PRG_OK;
}

int prg_cpd_get_params(prg_cpd* h, double* params_host) {
    PRG_REQUIRE(h && params_host, PRG_ERR_INVALID, "prg_cpd_get_params: NULL argument");
    prg::DeviceGuard g(h->device);
    // through pinned memory: a device->host copy into pageable memory pays ~100 us of staging, and the EM driver
    // reads the parameter block every iteration when it has a tolerance to test
    if (!h->pinned) PRG_HIP(hipHostMalloc((void**)&h->pinned, 64 * sizeof(double), hipHostMallocDefault));
    PRG_HIP(hipMemcpyAsync(h->pinned, h->params, PRG_NPARAMS * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    int* host_info = reinterpret_cast<int*>(h->pinned + 48);
    *host_info = 0;
    if (h->nr_info) PRG_HIP(hipMemcpyAsync(host_info, h->nr_info, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    PRG_HIP(hipStreamSynchronize(h->stream));
    for (int i = 0; i < PRG_NPARAMS; ++i) params_host[i] = h->pinned[i];
    if (*host_info != 0) {  // a non-rigid M-step since the last read-back hit a non-positive pivot (it does not stall to say so)
        const int pivot = *host_info - 1;
        PRG_HIP(hipMemsetAsync(h->nr_info, 0, sizeof(int), h->stream));
        prg::set_error("prg_cpd_mstep_nonrigid: the reduced system is not positive definite at pivot %d (sigma2 or lmd <= 0?)", pivot);
        return PRG_ERR_STATE;
    }
    return PRG_OK;
}

int prg_cpd_set_params(prg_cpd* h, const double* params_host) {
    PRG_REQUIRE(h && params_host, PRG_ERR_INVALID, "prg_cpd_set_params: NULL argument");
    prg::DeviceGuard g(h->device);
    PRG_HIP(hipMemcpyAsync(h->params, params_host, PRG_NPARAMS * sizeof(double), hipMemcpyHostToDevice, h->stream));
    PRG_HIP(hipStreamSynchronize(h->stream));
    h->init_rot_orthonormal = is_rotation(params_host);  // (a caller-set linear part that is no rotation keeps the two sweeps)
    return PRG_OK;
}

int prg_cpd_get_moments(prg_cpd* h, double* moments_host) {
    PRG_REQUIRE(h && moments_host, PRG_ERR_INVALID, "prg_cpd_get_moments: NULL argument");
    prg::DeviceGuard g(h->device);
    PRG_HIP(hipMemcpyAsync(moments_host, h->moments, PRG_NMOMENTS * sizeof(double), hipMemcpyDeviceToHost,
                           h->stream));
    PRG_HIP(hipStreamSynchronize(h->stream));
    return PRG_OK;
}

int prg_cpd_get_estep(prg_cpd* h, double* pt1_hd, double* p1_hd, double* px_hd) {
    PRG_REQUIRE(h && h->have_estep, PRG_ERR_STATE, "prg_cpd_get_estep: no E-step has been run");
    PRG_REQUIRE(h->rowacc_valid || (!p1_hd && !px_hd), PRG_ERR_STATE,
                "prg_cpd_get_estep: the last E-step ran as the fused single sweep of a rigid iteration (prg_cpd_iterate / "
                "prg_cpd_set_moments_only): it leaves MOMENTS and pt1, no per-point p1 / px");
    prg::DeviceGuard g(h->device);
    const size_t need = (size_t)(h->N > h->M * h->D ? h->N : h->M * h->D) * sizeof(double);
    PRG_TRY(prg::ensure_stage(h, need));
    if (pt1_hd) {
        k_float_to_double<<<grid1(h->N), kBlock, 0, h->stream>>>(h->pt1, (double*)h->stage, h->N, h->perm_tgt);
        PRG_HIP(hipMemcpyAsync(pt1_hd, h->stage, h->N * sizeof(double), hipMemcpyDefault, h->stream));
        PRG_HIP(hipStreamSynchronize(h->stream));
    }
    if (p1_hd) {
        k_scatter_double<<<grid1(h->M), kBlock, 0, h->stream>>>(h->rowacc, (double*)h->stage, h->M, h->perm_src);
        PRG_HIP(hipMemcpyAsync(p1_hd, h->stage, h->M * sizeof(double), hipMemcpyDefault, h->stream));
        PRG_HIP(hipStreamSynchronize(h->stream));
    }
    if (px_hd) {
        k_pack_px<<<grid1(h->M), kBlock, 0, h->stream>>>(h->rowacc, h->Mcap, h->M, h->D, (double*)h->stage, h->perm_src);
        PRG_HIP(hipMemcpyAsync(px_hd, h->stage, h->M * h->D * sizeof(double), hipMemcpyDefault, h->stream));
        PRG_HIP(hipStreamSynchronize(h->stream));
    }
    PRG_HIP(hipGetLastError());
    return PRG_OK;
}

int prg_cpd_get_tsource(prg_cpd* h, float* tsource_hd) {
    PRG_REQUIRE(h && h->have_source && tsource_hd, PRG_ERR_STATE, "prg_cpd_get_tsource: source not set");
    prg::DeviceGuard g(h->device);
    PRG_TRY(prg::ensure_stage(h, (size_t)h->M * h->D * sizeof(float)));
    k_unpack_points<<<grid1(h->M), kBlock, 0, h->stream>>>(h->z4, h->M, h->D, (float*)h->stage, h->perm_src);
    PRG_HIP(hipMemcpyAsync(tsource_hd, h->stage, h->M * h->D * sizeof(float), hipMemcpyDefault, h->stream));
    PRG_HIP(hipStreamSynchronize(h->stream));
    return PRG_OK;
}

int prg_cpd_moments_from_estep(prg_cpd* h, const double* pt1_hd, const double* p1_hd, const double* px_hd) {
    PRG_REQUIRE(h && h->have_source && h->have_target, PRG_ERR_STATE, "prg_cpd_moments_from_estep: clouds not set");
    PRG_REQUIRE(pt1_hd && p1_hd && px_hd, PRG_ERR_INVALID, "prg_cpd_moments_from_estep: NULL array");
    prg::DeviceGuard g(h->device);
    const size_t nb_pt1 = (size_t)h->N * sizeof(double), nb_p1 = (size_t)h->M * sizeof(double),
                 nb_px = (size_t)h->M * h->D * sizeof(double);
    PRG_TRY(prg::ensure_stage(h, nb_pt1 + nb_p1 + nb_px));
    PRG_TRY(ensure_mompart(h));
    double* d_pt1 = (double*)h->stage;
    double* d_p1 = d_pt1 + h->N;
    double* d_px = d_p1 + h->M;
    PRG_HIP(hipMemcpyAsync(d_pt1, pt1_hd, nb_pt1, hipMemcpyDefault, h->stream));
    PRG_HIP(hipMemcpyAsync(d_p1, p1_hd, nb_p1, hipMemcpyDefault, h->stream));
    PRG_HIP(hipMemcpyAsync(d_px, px_hd, nb_px, hipMemcpyDefault, h->stream));
    const int nblk = mom_blocks(h);
    k_moments_from_arrays<<<nblk, kBlock, 0, h->stream>>>(d_pt1, d_p1, d_px, h->D, h->M, h->N, h->src4, h->tgt4,
                                                          h->perm_src, h->perm_tgt, h->mompart);
    k_reduce_partials<<<1, kRedBlock, 0, h->stream>>>(h->mompart, nblk, kMomComp, h->moments, 0);
    k_rowacc_from_arrays<<<nblk, kBlock, 0, h->stream>>>(d_pt1, d_p1, d_px, h->D, h->M, h->N, h->Mcap, h->perm_src,
                                                         h->perm_tgt, h->rowacc, h->pt1);
    PRG_HIP(hipGetLastError());
    PRG_HIP(hipStreamSynchronize(h->stream));
    h->have_estep = true;  // rowacc / MOMENTS now hold an E-step result (the caller's): all 24 moments and the per-point block
    h->last_estep_fused = false;
    h->rowacc_valid = true;
    return PRG_OK;
}

}  // extern "C"
