// E-step pair sweeps for the SPARSE regime of CPD's E-step (cpd.py:71-88), MI355X gfx950: a device-built work queue.
//
// Once sigma2 is small most (128 owned points x 32 streamed points) blocks of P are exact zeros.  The
// culled sweeps of cpd_sweeps_packed.hip still LAUNCH a wave for every (128-point block, 512-point segment) pair - 153k
// waves at C1, 94 % of which find nothing to do - and the few busy ones, spread thinly over the chip, run at the latency of
// their scalar loads (5500 cycles per block against 2100 of arithmetic) for up to 16 blocks in a row.  Here the work is found
// first and packed:
//
//   k_queue_build   one box test per (owned block, streamed group of 32), a ballot per 64 of them = one 64-bit word of the
//                   block's need-mask; the wave that tested a chunk of 512 groups cuts the needed ones, in order, into UNITS
//                   of Q groups (Q = 8 while the work is sparse; it doubles when the previous E-step produced more units than
//                   the chip has wave slots for) and appends them to the queue (one atomicAdd per workgroup);
//   k_rowpass_queue / k_colpass_queue
//                   persistent waves (8 per SIMD) take one unit each, then pop further ones with an atomicAdd, and evaluate
//                   them with the arithmetic of the culled sweeps - every wave busy, eight to a SIMD to hide the scalar
//                   loads, nobody with more than Q blocks in a row; a unit leaves its partial sums in ITS OWN slot;
//   consumers       k_colfinal / k_row_moments (cpd.hip) walk a block's chunks and their units IN ORDER and add the slots up.
//
// Results: the same pairs are evaluated with the same arithmetic as in the culled sweeps; the partials of a block are
// combined in a fixed order (chunk, then unit) wherever the atomics placed them in the queue, so the sweep is reproducible
// bit for bit whatever wave appended or popped what.
#include <math.h>

#include <algorithm>

#include "cpd_sweeps.h"

namespace {

typedef float f2 __attribute__((ext_vector_type(2)));
struct alignas(64) Quad { float4 q[4]; };
struct alignas(32) GroupMeta { float lo[3]; float hi[3]; float aux; float pad; };

constexpr double kLog2e = 1.4426950408889634;
constexpr int kBlock = prg::kSweepBlock;
constexpr float kCullLog2 = -prg::kCullExp;  // (cpd_sweeps.h)
constexpr int kChunkGroups = prg::kQueueChunkGroups;  // 512 streamed groups = 8 mask words per (block, chunk)
constexpr int kChunkWords = kChunkGroups / 64;
constexpr int kBuildWaves = 16;                       // waves (= chunks) per workgroup of the build kernel
#ifndef PRG_QMIN
#define PRG_QMIN 16
#endif
constexpr int kQmax = 256;                             // largest unit: large clouds early in the sparse regime (250k points: Q = 32 overflowed the queue above 20 % needed groups)
constexpr int kQmin = PRG_QMIN;                       // smallest unit (groups): fewer, larger partial results for the consumers

__device__ __forceinline__ f2 splat(float a) { return (f2){a, a}; }
__device__ __forceinline__ f2 exp2v(f2 a) { return (f2){__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)}; }
__device__ __forceinline__ f2 col_offset2(float kk, f2 run) {
    return (f2){prg::col_offset(kk, run.x), prg::col_offset(kk, run.y)};
}
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

// unit = (owned block, first group | count << 22): `count` needed groups starting at bit `first` of the block's need-mask
// (count 0: every needed group up to the end of the chunk - the coarse unit of an overfull queue)
__device__ __forceinline__ int2 make_unit(int b, int first_group, int count) { return make_int2(b, first_group | (int)((unsigned)count << 22)); }

// ---- 1. need-masks and units in one pass ------------------------------------------------------------------------------------
// grid = (owned blocks of 128 points, ceil(chunks / 16)); one WAVE per (block, chunk of 512 streamed groups).
// ROW: a group is needed unless kk dist2(boxes) + max b_n < -127 (every P of the 128 x 32 block is an exact zero);
// !ROW (column pass): unless dist2(boxes) > (sqrt(largest column minimum of the block) + source motion)^2 + 127 / |kk|.
// WHERE in the queue a chunk's units land does not matter for the result: chunk[b][c] = (first slot, units) records it, and
// the consumers walk a block's chunks and their units in order.
// ctrl: [0] units appended, [1] pop counter, [2] units of the previous sweep, [3] Q of the previous build, [6] Q of this one,
//       [7] coarse units appended (queue overfull: the chunk becomes one unit in the reserve behind `cap_soft`).
template <bool ROW>
__global__ __launch_bounds__(kBuildWaves * 64) void k_queue_build(const GroupMeta* __restrict__ own_meta,
                                                                  const GroupMeta* __restrict__ str_meta, int ngroups,
                                                                  int nchunk, const double* __restrict__ params,
                                                                  const float* __restrict__ colmin_g,
                                                                  const unsigned* __restrict__ motion, int target_units,
                                                                  int q_init, int cap_soft, unsigned long long* __restrict__ masks,
                                                                  int2* __restrict__ chunk, int2* __restrict__ units,
                                                                  int* __restrict__ ctrl) {
    __shared__ int wave_units[kBuildWaves], wg_base;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t b = blockIdx.x;
    const int c = blockIdx.y * kBuildWaves + wv;
    // groups per unit: the previous build's, doubled / halved until the queue holds about `target_units` (kQmin ... kQmax)
    int Q = ctrl[3] < kQmin ? kQmin : ctrl[3];
    const int prev_units = ctrl[2];
    if (q_init) Q = q_init;  // (first sweep over the queue after a dense engine: nothing to adapt from)
    else {
        // prev_units counts what the previous build WANTED to append (also past the end of an overfull queue).  Up to 32
        // groups per unit the size follows the tuned band around `target_units`; beyond, units only grow as far as it takes
        // to keep the queue from overflowing into coarse units (large clouds early in the sparse regime) and shrink back as
        // soon as it fits.
        const int guard = cap_soft / 4 * 3;
        int want = prev_units;
        while (Q > 32 && 2 * want <= guard) Q /= 2, want *= 2;
        if (Q <= 32) {
            while (want > 2 * target_units && Q < 32) Q *= 2, want /= 2;
            while (want < target_units / 2 && Q > kQmin) Q /= 2, want *= 2;
        }
        while (want > guard && Q < kQmax) Q *= 2, want /= 2;
    }
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) ctrl[6] = Q;
    unsigned long long word[kChunkWords];  // the chunk's need-mask (wave-uniform)
    int needed = 0;
#pragma unroll
    for (int r = 0; r < kChunkWords; ++r) word[r] = 0ull;
    if (c < nchunk) {
        const float kk = (float)(-kLog2e / (2.0 * params[13]));
        const GroupMeta* __restrict__ own = own_meta + b * 4;
        float lo[3], hi[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            lo[k] = fminf(fminf(own[0].lo[k], own[1].lo[k]), fminf(own[2].lo[k], own[3].lo[k]));
            hi[k] = fmaxf(fmaxf(own[0].hi[k], own[1].hi[k]), fmaxf(own[2].hi[k], own[3].hi[k]));
        }
        float thr = INFINITY;
        if (!ROW && colmin_g) {
            const float cmax = fmaxf(fmaxf(colmin_g[b * 4], colmin_g[b * 4 + 1]), fmaxf(colmin_g[b * 4 + 2], colmin_g[b * 4 + 3]));
            const float r = sqrtf(cmax) + __uint_as_float(*motion);
            thr = r * r * 1.00001f + kCullLog2 / kk;
        }
        const int g0 = c * kChunkGroups;
        GroupMeta gm[kChunkWords];
#pragma unroll
        for (int r = 0; r < kChunkWords; ++r) {  // (all eight loads go out before the first test)
            const int g = g0 + r * 64 + lane;
            gm[r] = str_meta[g < ngroups ? g : ngroups - 1];
        }
#pragma unroll
        for (int r = 0; r < kChunkWords; ++r) {
            const int g = g0 + r * 64 + lane;
            const float d2 = box_dist2(lo, hi, gm[r]);
            const bool need = g < ngroups && (ROW ? !(fmaf(d2, kk, gm[r].aux) < kCullLog2) : !(d2 > thr));
            word[r] = __ballot(need);
            needed += __popcll(word[r]);
        }
        if (lane < kChunkWords) {
            unsigned long long mine = word[0];
#pragma unroll
            for (int r = 1; r < kChunkWords; ++r)
                if (lane == r) mine = word[r];
            masks[(b * nchunk + c) * kChunkWords + lane] = mine;
        }
    }
    const int k = (needed + Q - 1) / Q;
    if (lane == 0) wave_units[wv] = k;
    __syncthreads();
    if (threadIdx.x == 0) {
        int ku = 0;
#pragma unroll
        for (int w = 0; w < kBuildWaves; ++w) ku += wave_units[w];
        wg_base = ku ? atomicAdd(ctrl, ku) : 0;
    }
    __syncthreads();
    if (c >= nchunk || lane != 0) return;
    int base = wg_base;
    for (int w = 0; w < wv; ++w) base += wave_units[w];
    if (k == 0) {
        chunk[b * nchunk + c] = make_int2(0, 0);
        return;
    }
    if (base + k > cap_soft) {
        // the queue is overfull (a dense E-step right after a sparse one, before Q has caught up): this chunk becomes ONE
        // coarse unit in the reserve; the slots it was given stay empty
        for (int u = base; u < base + k && u < cap_soft; ++u) units[u] = make_int2(-1, 0);
        const int slot = cap_soft + atomicAdd(ctrl + 7, 1);
        units[slot] = make_unit((int)b, c * kChunkGroups, 0);
        chunk[b * nchunk + c] = make_int2(slot, 1);
        return;
    }
    chunk[b * nchunk + c] = make_int2(base, k);
    // a unit every Q needed groups, in order
    int u = base, cnt = 0, first = 0;
#pragma unroll
    for (int r = 0; r < kChunkWords; ++r) {
        unsigned long long bits = word[r];
        while (bits) {
            const int bit = __builtin_ctzll(bits);
            bits &= bits - 1;
            if (cnt == 0) first = c * kChunkGroups + r * 64 + bit;
            if (++cnt == Q) {
                units[u++] = make_unit((int)b, first, Q);
                cnt = 0;
            }
        }
    }
    if (cnt) units[u] = make_unit((int)b, first, cnt);
}

// ---- 2. persistent sweeps -------------------------------------------------------------------------------------------------
__device__ __forceinline__ int pop_unit(int* __restrict__ ctrl, int lane) {
    int v = 0;
    if (lane == 0) v = atomicAdd(ctrl + 1, 1);
    return __builtin_amdgcn_readfirstlane(v);
}

// The needed groups of a unit, in order, with one group of look-ahead (the first quad of the NEXT needed group is fetched
// while the current one is evaluated, as in the culled sweeps).  Lane l < 8 holds word l of the unit's chunk.
struct UnitWalk {
    unsigned lo, hi;  // this lane's mask word (two halves)
    int chunk_g0;     // first group of the chunk
    int w;            // current word
    unsigned long long cur;  // remaining bits of the current word (wave-uniform)
    int left;         // needed groups still to deliver
    __device__ __forceinline__ void init(const unsigned long long* __restrict__ mrow, int first_group, int count, int lane) {
        chunk_g0 = first_group & ~(kChunkGroups - 1);
        const unsigned long long mine = lane < kChunkWords ? mrow[chunk_g0 / 64 + lane] : 0ull;
        lo = (unsigned)mine;
        hi = (unsigned)(mine >> 32);
        const int off = first_group - chunk_g0;
        w = off >> 6;
        cur = word(w) & (~0ull << (off & 63));
        left = count ? count : kChunkGroups;  // (count 0: to the end of the chunk)
    }
    __device__ __forceinline__ unsigned long long word(int i) const {
        return (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)lo, i) |
               ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)hi, i) << 32);
    }
    __device__ __forceinline__ int next() {  // the next needed group, or -1
        if (left <= 0) return -1;
        while (!cur) {
            if (++w >= kChunkWords) return -1;
            cur = word(w);
        }
        const int bit = __builtin_ctzll(cur);
        cur &= cur - 1;
        --left;
        return chunk_g0 + w * 64 + bit;
    }
};

// Row pass over the queue: lane owns rows b * 128 + 2 * lane, +1; slot u receives p1, ux, uy, uz, e as [5][128] floats.
__global__ __launch_bounds__(kBlock) void k_rowpass_queue(const float4* __restrict__ z4, const float4* __restrict__ tgt4,
                                                          const unsigned long long* __restrict__ masks, int nwords,
                                                          const int2* __restrict__ units, int* __restrict__ ctrl,
                                                          int cap_soft, const double* __restrict__ params,
                                                          float* __restrict__ slots, unsigned* __restrict__ ucount) {
    const int lane = threadIdx.x & 63;
    const int n_fine = ctrl[0] < cap_soft ? ctrl[0] : cap_soft, total = n_fine + ctrl[7];
    const float kk = (float)(-kLog2e / (2.0 * params[13]));
    // the first unit of a wave is its own number (no atomic storm at the start: the pop counter starts behind the grid)
    int u = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)));  // (wave-uniform: scalar loads below)
    while (u < total) {
        const int slot = u < n_fine ? u : cap_soft + (u - n_fine);
        const int2 un = units[slot];
        const int b = un.x;
        if (b >= 0) {  // (an empty slot of an overfull queue otherwise)
            const int64_t m0 = (int64_t)b * 128 + 2 * lane;
            const float4 za = z4[m0], zb = z4[m0 + 1];
            const f2 zx = {za.x, zb.x}, zy = {za.y, zb.y}, zz = {za.z, zb.z}, zq = {za.w, zb.w};
            f2 p1 = splat(0.f), ux = splat(0.f), uy = splat(0.f), uz = splat(0.f), e = splat(0.f);
            int ngrp = 0;
            UnitWalk walk;
            walk.init(masks + (int64_t)b * nwords, un.y & 0x3FFFFF, (int)((unsigned)un.y >> 22), lane);
            int g = walk.next();
            if (g >= 0) {
                const Quad* __restrict__ tp = reinterpret_cast<const Quad*>(tgt4);
                Quad cq = tp[(int64_t)g * 8];
                while (g >= 0) {
                    const int gnext = walk.next();
                    ++ngrp;
                    const Quad* __restrict__ q = tp + (int64_t)g * 8;
                    const Quad* __restrict__ qn = tp + (int64_t)(gnext >= 0 ? gnext : g) * 8;
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        const Quad nq = (t < 7) ? q[t + 1] : qn[0];
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const f2 dx = zx - splat(cq.q[c].x), dy = zy - splat(cq.q[c].y), dz = zz - splat(cq.q[c].z);
                            const f2 d = fmav(dz, dz, fmav(dy, dy, fmav(dx, dx, zq)));
                            const f2 pr = exp2v(fmav(d, splat(kk), splat(cq.q[c].w)));
                            p1 += pr;
                            ux = fmav(pr, dx, ux);
                            uy = fmav(pr, dy, uy);
                            uz = fmav(pr, dz, uz);
                            e = fmav(pr, d, e);
                        }
                        cq = nq;
                    }
                    g = gnext;
                }
            }
            float* __restrict__ o = slots + (int64_t)slot * (5 * 128) + 2 * lane;
            *reinterpret_cast<float2*>(o) = make_float2(p1.x, p1.y);
            *reinterpret_cast<float2*>(o + 128) = make_float2(-ux.x, -ux.y);  // u = sum P (x - z) = -sum P (z - x)
            *reinterpret_cast<float2*>(o + 256) = make_float2(-uy.x, -uy.y);
            *reinterpret_cast<float2*>(o + 384) = make_float2(-uz.x, -uz.y);
            *reinterpret_cast<float2*>(o + 512) = make_float2(fmaf(-za.w, p1.x, e.x), fmaf(-zb.w, p1.y, e.y));
            if (lane == 0) ucount[slot] = (unsigned)ngrp;
        } else if (lane == 0) {
            ucount[slot] = 0u;
        }
        u = pop_unit(ctrl, lane);
    }
}

// Column pass over the queue: lane owns columns b * 128 + 2 * lane, +1; slot u receives (min d^2, sum exp2) pairs as
// [128] float2 (sums relative to the offset of their own minimum, as the culled column pass leaves them).
// RESID: the residual-form single sweep of a rigid iteration (k_colpass_cull<true> in cpd_sweeps_packed.hip has the algebra):
// slot u receives (min d^2, A, Ux, Uy, Uz, R) as [6][128] floats.
template <bool RESID>
__global__ __launch_bounds__(kBlock) void k_colpass_queue(const float4* __restrict__ tgt4, const float4* __restrict__ z4,
                                                          const unsigned long long* __restrict__ masks, int nwords,
                                                          const int2* __restrict__ units, int* __restrict__ ctrl,
                                                          int cap_soft, const double* __restrict__ params,
                                                          float2* __restrict__ slots, unsigned* __restrict__ ucount) {
    const int lane = threadIdx.x & 63;
    const int n_fine = ctrl[0] < cap_soft ? ctrl[0] : cap_soft, total = n_fine + ctrl[7];
    const float kk = (float)(-kLog2e / (2.0 * params[13]));
    int u = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)));  // (wave-uniform: scalar loads below)
    while (u < total) {
        const int slot = u < n_fine ? u : cap_soft + (u - n_fine);
        const int2 un = units[slot];
        const int b = un.x;
        if (b >= 0) {
            const int64_t n0 = (int64_t)b * 128 + 2 * lane;
            const float4 xa = tgt4[n0], xb = tgt4[n0 + 1];
            const f2 x = {xa.x, xb.x}, y = {xa.y, xb.y}, z = {xa.z, xb.z};
            f2 run = splat(INFINITY), off = splat(INFINITY), sm = splat(0.f);
            f2 ux = splat(0.f), uy = splat(0.f), uz = splat(0.f), rr = splat(0.f);
            int ngrp = 0;
            UnitWalk walk;
            walk.init(masks + (int64_t)b * nwords, un.y & 0x3FFFFF, (int)((unsigned)un.y >> 22), lane);
            int g = walk.next();
            if (g >= 0) {
                const Quad* __restrict__ zp = reinterpret_cast<const Quad*>(z4);
                Quad qa = zp[(int64_t)g * 8];
                while (g >= 0) {
                    const int gnext = walk.next();
                    ++ngrp;
                    const Quad* __restrict__ q = zp + (int64_t)g * 8;
                    const Quad* __restrict__ qn = zp + (int64_t)(gnext >= 0 ? gnext : g) * 8;
                    if constexpr (RESID) {
#pragma unroll
                        for (int t = 0; t < 8; ++t) {
                            const Quad nq = (t < 7) ? q[t + 1] : qn[0];
                            f2 dx[4], dy[4], dz[4], d2[4];
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                dx[c] = x - splat(qa.q[c].x);
                                dy[c] = y - splat(qa.q[c].y);
                                dz[c] = z - splat(qa.q[c].z);
                                d2[c] = fmav(dz[c], dz[c], fmav(dy[c], dy[c], fmav(dx[c], dx[c], splat(qa.q[c].w))));
                            }
                            const f2 cm = minv(minv(d2[0], d2[1]), minv(d2[2], d2[3]));
                            if ((cm.x < run.x) | (cm.y < run.y)) {
                                const f2 nm = minv(run, cm);
                                const f2 noff = col_offset2(kk, nm);
                                const f2 f = exp2v(noff - off);
                                sm *= f; ux *= f; uy *= f; uz *= f; rr *= f;
                                run = nm;
                                off = noff;
                            }
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                const f2 pr = exp2v(fmav(d2[c], splat(kk), off));
                                sm += pr;
                                ux = fmav(pr, dx[c], ux);
                                uy = fmav(pr, dy[c], uy);
                                uz = fmav(pr, dz[c], uz);
                                rr = fmav(pr, d2[c], rr);
                            }
                            qa = nq;
                        }
                    } else {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const Quad qb = q[2 * t + 1];
                        f2 d2[8];
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const f2 dx = x - splat(qa.q[c].x), dy = y - splat(qa.q[c].y), dz = z - splat(qa.q[c].z);
                            d2[c] = fmav(dz, dz, fmav(dy, dy, fmav(dx, dx, splat(qa.q[c].w))));
                        }
                        qa = (t < 3) ? q[2 * t + 2] : qn[0];
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const f2 dx = x - splat(qb.q[c].x), dy = y - splat(qb.q[c].y), dz = z - splat(qb.q[c].z);
                            d2[4 + c] = fmav(dz, dz, fmav(dy, dy, fmav(dx, dx, splat(qb.q[c].w))));
                        }
                        f2 cm = d2[0];
#pragma unroll
                        for (int c = 1; c < 8; ++c) cm = minv(cm, d2[c]);
                        if ((cm.x < run.x) | (cm.y < run.y)) {
                            const f2 nm = minv(run, cm);
                            const f2 noff = col_offset2(kk, nm);
                            sm *= exp2v(noff - off);
                            run = nm;
                            off = noff;
                        }
#pragma unroll
                        for (int c = 0; c < 8; ++c) sm += exp2v(fmav(d2[c], splat(kk), off));
                    }
                    }
                    g = gnext;
                }
            }
            if constexpr (RESID) {
                float* __restrict__ o = reinterpret_cast<float*>(slots) + (int64_t)slot * (6 * 128) + 2 * lane;
                *reinterpret_cast<float2*>(o) = make_float2(run.x, run.y);
                *reinterpret_cast<float2*>(o + 128) = make_float2(sm.x, sm.y);
                *reinterpret_cast<float2*>(o + 256) = make_float2(ux.x, ux.y);
                *reinterpret_cast<float2*>(o + 384) = make_float2(uy.x, uy.y);
                *reinterpret_cast<float2*>(o + 512) = make_float2(uz.x, uz.y);
                *reinterpret_cast<float2*>(o + 640) = make_float2(rr.x, rr.y);
            } else {
                *reinterpret_cast<float4*>(slots + (int64_t)slot * 128 + 2 * lane) = make_float4(run.x, sm.x, run.y, sm.y);
            }
            if (lane == 0) ucount[slot] = (unsigned)ngrp;
        } else if (lane == 0) {
            ucount[slot] = 0u;
        }
        u = pop_unit(ctrl, lane);
    }
}

}  // namespace

namespace prg {

static int queue_chunks(int64_t streamed) { return (int)ceil_div(ceil_div(streamed, kGroup), kQueueChunkGroups); }

// fine units the queue holds before chunks fall back to one coarse unit each (the reserve behind it has a slot per chunk)
static int64_t queue_cap_soft(int64_t owned, int64_t streamed) {
    const int64_t worst = ceil_div(owned, 128) * ceil_div(ceil_div(streamed, kGroup), 8);
    return std::min<int64_t>(worst, kQueueMaxUnits) + 64;
}

int64_t queue_max_units(int64_t owned, int64_t streamed) {
    return queue_cap_soft(owned, streamed) + ceil_div(owned, 128) * queue_chunks(streamed) + 64;
}

static int ensure_queue(prg_cpd* h, SweepQueue& q, int64_t nblocks, int nchunk, int64_t max_units) {
    if (nblocks > q.cap_blocks || nchunk > q.cap_chunk || max_units > q.cap_units) {
        PRG_HIP(hipStreamSynchronize(h->stream));
        for (void* p : {(void*)q.masks, (void*)q.chunk, (void*)q.units, (void*)q.ctrl, (void*)q.ucount})
            if (p) (void)hipFree(p);
        q = SweepQueue();
        PRG_HIP(hipMalloc((void**)&q.masks, (size_t)nblocks * nchunk * (kQueueChunkGroups / 64) * sizeof(unsigned long long)));
        PRG_HIP(hipMalloc((void**)&q.chunk, (size_t)nblocks * nchunk * sizeof(int2)));
        PRG_HIP(hipMalloc((void**)&q.units, (size_t)max_units * sizeof(int2)));
        PRG_HIP(hipMalloc((void**)&q.ctrl, 16 * sizeof(int)));
        PRG_HIP(hipMalloc((void**)&q.ucount, (size_t)max_units * sizeof(unsigned)));
        int init[16] = {0};
        init[1] = kQueueWorkgroups * (kSweepBlock / 64);
        PRG_HIP(hipMemcpyAsync(q.ctrl, init, sizeof(init), hipMemcpyHostToDevice, h->stream));
        PRG_HIP(hipStreamSynchronize(h->stream));
        q.cap_blocks = nblocks;
        q.cap_chunk = nchunk;
        q.cap_units = max_units;
    }
    return PRG_OK;
}

template <bool ROW>
static int build_queue(prg_cpd* h, SweepQueue& q, int64_t owned, int64_t streamed, const float* own_meta,
                       const float* str_meta, bool use_seed, int q_init) {
    const int64_t nblocks = ceil_div(owned, 128);
    const int64_t ngroups = ceil_div(streamed, kGroup);
    const int nchunk = queue_chunks(streamed);
    PRG_REQUIRE(ngroups < ((int64_t)1 << 22) && nblocks < ((int64_t)1 << 31) && queue_max_units(owned, streamed) < ((int64_t)1 << 30),
                PRG_ERR_INVALID, "prg_cpd_estep: cloud too large for the sweep queue");
    PRG_TRY(ensure_queue(h, q, nblocks, nchunk, queue_max_units(owned, streamed)));
    q.nblocks = nblocks;
    q.nchunk = nchunk;
    q.cap_soft = (int)queue_cap_soft(owned, streamed);
    dim3 grid((unsigned)nblocks, (unsigned)ceil_div(nchunk, kBuildWaves));
    k_queue_build<ROW><<<grid, kBuildWaves * 64, 0, h->stream>>>(
        reinterpret_cast<const GroupMeta*>(own_meta), reinterpret_cast<const GroupMeta*>(str_meta), (int)ngroups, nchunk, h->params,
        (!ROW && use_seed) ? h->colmin + h->Ncap : nullptr, h->motion + ((h->estep_count - 1) & 1), kQueueTargetUnits, q_init, q.cap_soft,
        q.masks, q.chunk, q.units, q.ctrl);
    return PRG_OK;
}

// allocate both queues now (first E-step of a plan) rather than in the middle of a registration, where the first sparse
// E-step would stall on the allocations
int prepare_queues(prg_cpd* h) {
    PRG_TRY(ensure_queue(h, h->qcol, ceil_div(h->N, 128), queue_chunks(h->M), queue_max_units(h->N, h->M)));
    PRG_TRY(ensure_queue(h, h->qrow, ceil_div(h->M, 128), queue_chunks(h->N), queue_max_units(h->M, h->N)));
    return PRG_OK;
}

int launch_colpass_queue(prg_cpd* h, bool use_seed, int q_init, bool resid) {
    PRG_TRY(build_queue<false>(h, h->qcol, h->N, h->M, h->tmeta, h->zmeta, use_seed, q_init));
    SweepQueue& q = h->qcol;
    if (resid)
        k_colpass_queue<true><<<kQueueWorkgroups, kBlock, 0, h->stream>>>(h->tgt4, h->z4, q.masks, q.nchunk * (kQueueChunkGroups / 64), q.units,
                                                                         q.ctrl, q.cap_soft, h->params, reinterpret_cast<float2*>(h->colpart),
                                                                         q.ucount);
    else
    k_colpass_queue<false><<<kQueueWorkgroups, kBlock, 0, h->stream>>>(h->tgt4, h->z4, q.masks, q.nchunk * (kQueueChunkGroups / 64), q.units,
                                                               q.ctrl, q.cap_soft, h->params, reinterpret_cast<float2*>(h->colpart),
                                                               q.ucount);
    h->wg_col = -1;  // counted per unit (prg_cpd_pair_counts reads the unit counts and ucount)
    h->dense_pairs_col = 0.0;
    return PRG_OK;
}

int launch_rowpass_queue(prg_cpd* h, int q_init) {
    PRG_TRY(build_queue<true>(h, h->qrow, h->M, h->N, h->zmeta, h->tmeta, false, q_init));
    SweepQueue& q = h->qrow;
    k_rowpass_queue<<<kQueueWorkgroups, kBlock, 0, h->stream>>>(h->z4, h->tgt4, q.masks, q.nchunk * (kQueueChunkGroups / 64), q.units,
                                                               q.ctrl, q.cap_soft, h->params, h->rowpart, q.ucount);
    h->wg_row = -1;
    h->dense_pairs_row = 0.0;
    return PRG_OK;
}

}  // namespace prg
