// The single pair sweep of a rigid EM iteration on the VECTOR pipe (cpd.py:71-88 + what cpd.py:160-192 consumes of it), round 6:
// the column block's OWNER finds its own work, MI355X gfx950.
//
// What the round-5 sweeps of this regime cost beside their arithmetic (profiles/r6_shard_trace_*.txt): a build pass that tests
// every (128-column block, 32-point group) pair of boxes - 2.4 M tests, 35 us at C1 - to fill a queue, persistent waves that walk
// units of 16 groups in a row at the latency of their scalar loads, one slot of partial sums per unit and a merge kernel that
// chases them (40 us); on a target shard a grid of 38k waves of which a few hundred find work (52 us for 7 us of arithmetic)
// and a merge over 98 flagged planes (23 us).
//
// Here a workgroup of 8 waves owns 128 columns (2 per lane, as before) and the whole streamed cloud - or 1/S of it on a shard,
// where 128-column blocks are too few to fill the chip - and finds its pairs through the hierarchy the kd-tree order of the
// clouds (morton.h) provides: every aligned 256-point chunk and every 32-point group of the stream is ONE axis-aligned cell.
//   level 1   lane l of wave w tests the box of chunk ((64 r + l) S + s) 8 + w against the box of the owned columns: chunks are
//             dealt out round-robin over the S x 8 waves that share a column block, so a run of neighbouring cells - what a block
//             needs in the sparse regime - is spread evenly over them; one ballot = 64 chunks = 16 384 streamed points
//   level 2   the 8 groups of 8 needed chunks at a time, one per lane, a second ballot
//   sweep     the needed groups with the arithmetic of k_colpass_cull<true>: the streamed points arrive through
//             SGPRs, a lane keeps (min d^2, A, Ux, Uy, Uz, R) of its two columns under the online rescaling
//   merge     the 8 waves' sums in LDS, by whichever wave arrives last (no barrier); one plane [6][ncap] per s and a touched flag per
//             (block, s) - the layout k_colfinal_resid<false> already reads.  S = 1 on one GPU at C1: ONE partial per column.
// No queue, no build pass, no units; the box tests fall from 2.4 M to ~0.3 M per sweep; nobody walks more than its share of the
// block's groups.  The pairs evaluated are those of the round-5 sweeps (same boxes, same 2^-48 bound), the sums differ in their
// fp32 summation order only.
#include <math.h>

#include <algorithm>

#include "cpd_sweeps.h"

namespace {

typedef float f2 __attribute__((ext_vector_type(2)));
struct alignas(64) Quad { float4 q[4]; };
struct alignas(32) GroupMeta { float lo[3]; float hi[3]; float aux; float pad; };

constexpr double kLog2e = 1.4426950408889634;
constexpr float kCullLog2 = -prg::kCullExp;
constexpr int kWaves = prg::kOwnerWaves;
constexpr int kThreads = 64 * kWaves;
#ifndef PRG_OWNER_SUB
#define PRG_OWNER_SUB 2
#endif
constexpr int kOwnerSub = PRG_OWNER_SUB;  // streamed points per rescale check (2: 64 VGPRs without spills, 8 waves per SIMD)

__device__ __forceinline__ f2 splat(float a) { return (f2){a, a}; }
__device__ __forceinline__ f2 exp2v(f2 a) { return (f2){__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)}; }
__device__ __forceinline__ f2 col_offset2(float kk, f2 run) { return (f2){prg::col_offset(kk, run.x), prg::col_offset(kk, run.y)}; }
__device__ __forceinline__ f2 fmav(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 minv(f2 a, f2 b) { return __builtin_elementwise_min(a, b); }

__device__ __forceinline__ float box_dist2(const float (&alo)[3], const float (&ahi)[3], const GroupMeta& g) {
    float d2 = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float gap = fmaxf(fmaxf(alo[k] - g.hi[k], g.lo[k] - ahi[k]), 0.f);
        d2 = fmaf(gap, gap, d2);
    }
    return d2;
}

// grid = (ceil(N / 128), S), 512 threads.  nchunks / ngroups: 256-point chunks / 32-point groups of the stream that hold real points.
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(7, 8))) void k_colpass_owner(const float4* __restrict__ tgt4, const float4* __restrict__ z4,
                                                            const GroupMeta* __restrict__ zchunk,
                                                            const GroupMeta* __restrict__ zmeta,
                                                            const GroupMeta* __restrict__ tmeta, int64_t n, int nchunks, int ngroups,
                                                            const double* __restrict__ params,
                                                            const float* __restrict__ colmin_g,
                                                            const unsigned* __restrict__ motion, float* __restrict__ colpart,
                                                            int64_t ncap, unsigned* __restrict__ wgcount,
                                                            unsigned char* __restrict__ colflag) {
    __shared__ float2 partr[kWaves][6][64];
    __shared__ int arrived, wave_groups[kWaves];
    if (threadIdx.x == 0) arrived = 0;
    __syncthreads();  // (at launch, before any wave waits for memory; there is no barrier at the end)
    const float kk = (float)(-kLog2e / (2.0 * params[13]));
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t n0 = (int64_t)blockIdx.x * 128 + 2 * lane;
    const int stride = (int)gridDim.y * kWaves, first = (int)blockIdx.y * kWaves + wv;  // this wave's chunks: first + i * stride
    // the box of the owned columns and how far a needed cell may be from it (k_colpass_cull has the derivation)
    // (the last block's pad-only groups stay out of it: their boxes sit 1e18 away and would make the block need every cell)
    const GroupMeta* __restrict__ own = tmeta + (int64_t)blockIdx.x * 4;
    const int real_groups = (int)(((n - (int64_t)blockIdx.x * 128 < 128 ? n - (int64_t)blockIdx.x * 128 : 128) + 31) >> 5);  // 1 .. 4
    float lo[3], hi[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        lo[k] = own[0].lo[k];
        hi[k] = own[0].hi[k];
#pragma unroll
        for (int q = 1; q < 4; ++q)
            if (q < real_groups) {
                lo[k] = fminf(lo[k], own[q].lo[k]);
                hi[k] = fmaxf(hi[k], own[q].hi[k]);
            }
    }
    float thr = INFINITY;
    if (colmin_g) {
        const int64_t gw = (int64_t)blockIdx.x * 4;
        const float cmax = fmaxf(fmaxf(colmin_g[gw], colmin_g[gw + 1]), fmaxf(colmin_g[gw + 2], colmin_g[gw + 3]));
        const float r = sqrtf(cmax) + __uint_as_float(*motion);
        thr = r * r * 1.00001f + kCullLog2 / kk;
    }
    f2 x = splat(0.f), y = splat(0.f), z = splat(0.f);
    bool have_points = false;
    f2 run = splat(INFINITY), off = splat(INFINITY), s = splat(0.f);
    f2 ux = splat(0.f), uy = splat(0.f), uz = splat(0.f), rr = splat(0.f);
    int ngrp = 0;  // (128 x 32) blocks of pairs this wave evaluates (wave-uniform)
    const Quad* __restrict__ zp = reinterpret_cast<const Quad*>(z4);
    for (int i0 = 0; first + (int64_t)i0 * stride < nchunks; i0 += 64) {
        // ---- level 1: one chunk per lane ----
        const int64_t c = first + (int64_t)(i0 + lane) * stride;
        const GroupMeta cm = zchunk[c < nchunks ? c : nchunks - 1];
        unsigned long long mask1 = __ballot(c < nchunks && !(box_dist2(lo, hi, cm) > thr));
        while (mask1) {
            // ---- level 2: the 8 groups of the next (up to) 8 needed chunks, one per lane ----
            unsigned long long m = mask1;
            const int k = lane >> 3;
#pragma unroll
            for (int j = 0; j < 7; ++j)
                if (j < k) m &= m - 1;
            const bool vk = m != 0ull;
            const int g_mine = (first + (i0 + (vk ? __builtin_ctzll(m) : 0)) * stride) * 8 + (lane & 7);
            const bool in = vk && g_mine < ngroups;
            const GroupMeta gm = zmeta[in ? g_mine : 0];
            unsigned long long mask2 = __ballot(in && !(box_dist2(lo, hi, gm) > thr));
#pragma unroll
            for (int j = 0; j < 8; ++j) mask1 &= mask1 - 1;  // (0 stays 0)
            if (mask2 == 0ull) continue;
            ngrp += __builtin_popcountll(mask2);
            if (!have_points) {
                const float4 a = tgt4[n0], b = tgt4[n0 + 1];
                x = (f2){a.x, b.x};
                y = (f2){a.y, b.y};
                z = (f2){a.z, b.z};
                have_points = true;
            }
            int g = __builtin_amdgcn_readlane(g_mine, __builtin_ctzll(mask2));
            mask2 &= mask2 - 1;
            Quad qa = zp[(int64_t)g * 8];
            for (;;) {
                const int jn = mask2 ? __builtin_ctzll(mask2) : -1;
                mask2 &= mask2 - 1;  // (0 stays 0)
                const int gnext = jn >= 0 ? __builtin_amdgcn_readlane(g_mine, jn) : -1;
                const Quad* __restrict__ q = zp + (int64_t)g * 8;
                const Quad* __restrict__ qn = zp + (int64_t)(gnext >= 0 ? gnext : g) * 8;
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const Quad nq = (t < 7) ? q[t + 1] : qn[0];  // prefetch: next quad, or the next needed group's first
#pragma unroll
                    for (int h2 = 0; h2 < 4; h2 += kOwnerSub) {  // kOwnerSub streamed points at a time: the differences stay in registers
                        f2 dx[kOwnerSub], dy[kOwnerSub], dz[kOwnerSub], d2[kOwnerSub];
#pragma unroll
                        for (int cc = 0; cc < kOwnerSub; ++cc) {
                            dx[cc] = x - splat(qa.q[h2 + cc].x);
                            dy[cc] = y - splat(qa.q[h2 + cc].y);
                            dz[cc] = z - splat(qa.q[h2 + cc].z);
                            d2[cc] = fmav(dz[cc], dz[cc], fmav(dy[cc], dy[cc], fmav(dx[cc], dx[cc], splat(qa.q[h2 + cc].w))));
                        }
                        f2 cmn = d2[0];
#pragma unroll
                        for (int cc = 1; cc < kOwnerSub; ++cc) cmn = minv(cmn, d2[cc]);
                        if ((cmn.x < run.x) | (cmn.y < run.y)) {  // rare after the first trips: all five sums move to the new minimum
                            const f2 nm = minv(run, cmn);
                            const f2 noff = col_offset2(kk, nm);
                            const f2 f = exp2v(noff - off);  // first use: off == +inf -> 0, and the sums are 0 anyway
                            s *= f; ux *= f; uy *= f; uz *= f; rr *= f;
                            run = nm;
                            off = noff;
                        }
#pragma unroll
                        for (int cc = 0; cc < kOwnerSub; ++cc) {
                            const f2 pr = exp2v(fmav(d2[cc], splat(kk), off));
                            s += pr;
                            ux = fmav(pr, dx[cc], ux);
                            uy = fmav(pr, dy[cc], uy);
                            uz = fmav(pr, dz[cc], uz);
                            rr = fmav(pr, d2[cc], rr);
                        }
                    }
                    qa = nq;
                }
                if (gnext < 0) break;
                g = gnext;
            }
        }
    }
    // ---- merge of the workgroup's waves in LDS by the last one to arrive (as k_colpass_cull) ----
    if (ngrp) {
        partr[wv][0][lane] = make_float2(run.x, run.y);
        partr[wv][1][lane] = make_float2(s.x, s.y);
        partr[wv][2][lane] = make_float2(ux.x, ux.y);
        partr[wv][3][lane] = make_float2(uy.x, uy.y);
        partr[wv][4][lane] = make_float2(uz.x, uz.y);
        partr[wv][5][lane] = make_float2(rr.x, rr.y);
    }
    int last = 0;
    if (lane == 0) {
        wave_groups[wv] = ngrp;
        last = atomicAdd(&arrived, 1) == kWaves - 1;  // LDS ops of a wave execute in order: its sums are visible
    }
    if (!__builtin_amdgcn_readfirstlane(last)) return;
    int total = 0;
    run = splat(INFINITY);
    off = splat(INFINITY);
    s = ux = uy = uz = rr = splat(0.f);
#pragma unroll
    for (int k = 0; k < kWaves; ++k) {
        const int tk = wave_groups[k];
        total += tk;
        if (!tk) continue;
        const float2 q0 = partr[k][0][lane], q1 = partr[k][1][lane], q2 = partr[k][2][lane], q3 = partr[k][3][lane],
                     q4 = partr[k][4][lane], q5 = partr[k][5][lane];
        const f2 orun = {q0.x, q0.y};
        const f2 nm = minv(run, orun);
        const f2 noff = col_offset2(kk, nm);
        const f2 fa = exp2v(noff - off), fb = exp2v(noff - col_offset2(kk, orun));
        s = s * fa + (f2){q1.x, q1.y} * fb;
        ux = ux * fa + (f2){q2.x, q2.y} * fb;
        uy = uy * fa + (f2){q3.x, q3.y} * fb;
        uz = uz * fa + (f2){q4.x, q4.y} * fb;
        rr = rr * fa + (f2){q5.x, q5.y} * fb;
        run = nm;
        off = noff;
    }
    if (lane == 0) {
        wgcount[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = (unsigned)total;
        colflag[(int64_t)blockIdx.x * gridDim.y + blockIdx.y] = total ? 1 : 0;
    }
    if (!total) return;  // an untouched (block, plane) leaves nothing to read
    float* __restrict__ o = colpart + (int64_t)blockIdx.y * 6 * ncap + n0;
    *reinterpret_cast<float2*>(o) = make_float2(run.x, run.y);
    *reinterpret_cast<float2*>(o + ncap) = make_float2(s.x, s.y);
    *reinterpret_cast<float2*>(o + 2 * ncap) = make_float2(ux.x, ux.y);
    *reinterpret_cast<float2*>(o + 3 * ncap) = make_float2(uy.x, uy.y);
    *reinterpret_cast<float2*>(o + 4 * ncap) = make_float2(uz.x, uz.y);
    *reinterpret_cast<float2*>(o + 5 * ncap) = make_float2(rr.x, rr.y);
}

}  // namespace

namespace prg {

// Parts S the stream is dealt out over per column block.  The sweep is bound by its arithmetic in the mid regime and the chip
// takes whole workgroups as they come: ~2.5 rounds of its 8192 wave slots (256 CUs x 4 SIMDs x 8) even the load out; far more
// parts than work cost their launch in the late regime.  Measured (whole EM iterations, ms; round 5's queue / grid beside it):
//   C1, one GPU      iteration 12: S = 1 0.606, 4 0.541, 8 0.544, 16 0.561, queue 0.547 | iteration 19: 0.180, 0.180, 0.196, 0.220, queue 0.203
//   rank 3 of 8      iteration 7: S = 6 0.438, 16 0.375, 32 0.350, 64 0.367, grid 0.363 | iteration 19: 0.076, 0.086, 0.090, 0.090, grid 0.092
// C1 on one GPU: 782 blocks -> S = 3; a 1/8 shard: 98 blocks -> S = 26.
int owner_planes(int64_t owned, int64_t streamed) {
    static const int env = getenv("PRG_OWNER_PLANES") ? atoi(getenv("PRG_OWNER_PLANES")) : 0;
    const int64_t blocks = ceil_div(owned, 128), chunks = ceil_div(streamed, kSuper);
    int64_t s = env > 0 ? env : (20000 + blocks * kOwnerWaves / 2) / (blocks * kOwnerWaves);
    if (env <= 0) s = std::min<int64_t>(s, std::max<int64_t>(1, chunks / kOwnerWaves));  // (at least one chunk per wave)
    return (int)std::max<int64_t>(1, std::min<int64_t>(s, kOwnerMaxPlanes));
}

void launch_colpass_owner(prg_cpd* h, bool use_seed, int planes) {
    dim3 grid((unsigned)ceil_div(h->N, 128), (unsigned)planes);
    k_colpass_owner<<<grid, kThreads, 0, h->stream>>>(h->tgt4, h->z4, reinterpret_cast<const GroupMeta*>(h->zchunk),
                                                      reinterpret_cast<const GroupMeta*>(h->zmeta),
                                                      reinterpret_cast<const GroupMeta*>(h->tmeta), h->N, (int)ceil_div(h->M, kSuper),
                                                      (int)ceil_div(h->M, kGroup), h->params, use_seed ? h->colmin + h->Ncap : nullptr,
                                                      h->motion + ((h->estep_count - 1) & 1), reinterpret_cast<float*>(h->colpart),
                                                      h->Ncap, h->wgcount, resid_flags(h, planes));
    h->wg_col = (int64_t)grid.x * grid.y;
    h->wg_col_pairs = 128.0 * kGroup;
    h->dense_pairs_col = 0.0;
}

}  // namespace prg
