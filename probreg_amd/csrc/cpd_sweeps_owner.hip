// The single pair sweep of a rigid EM iteration on the VECTOR pipe (cpd.py:71-88 + what cpd.py:160-192 consumes of it), round 6:
// the column block's OWNER finds its own work, MI355X gfx950.
//
// What the round-5 sweeps of this regime cost beside their arithmetic (profiles/r6_shard_trace_*.txt): a build pass that tests
// every (128-column block, 32-point group) pair of boxes - 2.4 M tests, 35 us at C1 - to fill a queue, persistent waves that walk
// units of 16 groups in a row at the latency of their scalar loads, one slot of partial sums per unit and a merge kernel that
// chases them (40 us); on a target shard a grid of 38k waves of which a few hundred find work (52 us for 7 us of arithmetic)
// and a merge over 98 flagged planes (23 us).
//
// Here a workgroup of 8 waves owns 64 columns (one per lane; or 128, two per lane as the other sweeps) and the whole streamed
// cloud - or 1/S of it where column blocks are too few to fill the chip - and finds its pairs through the hierarchy the kd-tree order of the
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
// streamed points per rescale check (their differences stay in registers between the check and the accumulation): 4 with one
// column per lane (60 VGPRs), 2 with two (72; 4 would spill).  Measured with one column per lane, C1's sweeps of iterations 12 / 19:
// 1 point 0.534 / 0.128 ms, 2 points 0.530 / 0.114, 4 points 0.490 / 0.105 (late 9 150 / 9 900 / 10 500 it/s); two quads per
// scalar-load round trip (128-byte loads, 34 SGPRs spilled to lanes): 0.521 / 0.112 - no gain, removed.
#ifdef PRG_OWNER_SUB
template <int CPL> struct OwnerSub { static constexpr int value = PRG_OWNER_SUB; };
#else
template <int CPL> struct OwnerSub { static constexpr int value = CPL == 1 ? 4 : 2; };
#endif

__device__ __forceinline__ float box_dist2(const float (&alo)[3], const float (&ahi)[3], const GroupMeta& g) {
    float d2 = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float gap = fmaxf(fmaxf(alo[k] - g.hi[k], g.lo[k] - ahi[k]), 0.f);
        d2 = fmaf(gap, gap, d2);
    }
    return d2;
}

// One or two columns per lane: the arithmetic on float or on float2 (v_pk_*_f32; packed fp32 issues at the rate of two plain
// instructions on gfx950, so neither is cheaper per pair - what differs is the GRANULARITY: a wave owns 64 or 128 columns).
template <int CPL> struct Cols;
template <> struct Cols<2> {
    typedef f2 T;
    typedef float2 S;
    static __device__ __forceinline__ T splat(float a) { return (f2){a, a}; }
    static __device__ __forceinline__ T exp2(T a) { return (f2){__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)}; }
    static __device__ __forceinline__ T coff(float kk, T run) { return (f2){prg::col_offset(kk, run.x), prg::col_offset(kk, run.y)}; }
    static __device__ __forceinline__ T fma(T a, T b, T c) { return __builtin_elementwise_fma(a, b, c); }
    static __device__ __forceinline__ T min(T a, T b) { return __builtin_elementwise_min(a, b); }
    static __device__ __forceinline__ bool any_less(T a, T b) { return (a.x < b.x) | (a.y < b.y); }
    static __device__ __forceinline__ S store(T a) { return make_float2(a.x, a.y); }
    static __device__ __forceinline__ T load(S a) { return (f2){a.x, a.y}; }
    static __device__ __forceinline__ void points(const float4* __restrict__ p, T& x, T& y, T& z) {
        const float4 a = p[0], b = p[1];
        x = (f2){a.x, b.x};
        y = (f2){a.y, b.y};
        z = (f2){a.z, b.z};
    }
};
template <> struct Cols<1> {
    typedef float T;
    typedef float S;
    static __device__ __forceinline__ T splat(float a) { return a; }
    static __device__ __forceinline__ T exp2(T a) { return __builtin_amdgcn_exp2f(a); }
    static __device__ __forceinline__ T coff(float kk, T run) { return prg::col_offset(kk, run); }
    static __device__ __forceinline__ T fma(T a, T b, T c) { return fmaf(a, b, c); }
    static __device__ __forceinline__ T min(T a, T b) { return fminf(a, b); }
    static __device__ __forceinline__ bool any_less(T a, T b) { return a < b; }
    static __device__ __forceinline__ S store(T a) { return a; }
    static __device__ __forceinline__ T load(S a) { return a; }
    static __device__ __forceinline__ void points(const float4* __restrict__ p, T& x, T& y, T& z) {
        const float4 a = p[0];
        x = a.x;
        y = a.y;
        z = a.z;
    }
};

// grid = (ceil(N / (64 CPL)), S), 512 threads.  nchunks / ngroups: 256-point chunks / 32-point groups of the stream that hold real points.
template <int CPL>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(7, 8))) void k_colpass_owner(const float4* __restrict__ tgt4, const float4* __restrict__ z4,
                                                            const GroupMeta* __restrict__ zchunk,
                                                            const GroupMeta* __restrict__ zmeta,
                                                            const GroupMeta* __restrict__ tmeta, int64_t n, int nchunks, int ngroups,
                                                            const double* __restrict__ params,
                                                            const float* __restrict__ colmin_g,
                                                            const unsigned* __restrict__ motion, float* __restrict__ colpart,
                                                            int64_t ncap, unsigned* __restrict__ wgcount,
                                                            unsigned char* __restrict__ colflag) {
    typedef Cols<CPL> C;
    typedef typename C::T T;
    constexpr int kCols = 64 * CPL, kOwnGroups = kCols / 32, kOwnerSub = OwnerSub<CPL>::value;
    __shared__ typename C::S partr[kWaves][6][64];
    __shared__ int arrived, wave_groups[kWaves];
    if (threadIdx.x == 0) arrived = 0;
    __syncthreads();  // (at launch, before any wave waits for memory; there is no barrier at the end)
    const float kk = (float)(-kLog2e / (2.0 * params[13]));
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t n0 = (int64_t)blockIdx.x * kCols + CPL * lane;
    const int stride = (int)gridDim.y * kWaves, first = (int)blockIdx.y * kWaves + wv;  // this wave's chunks: first + i * stride
    // the box of the owned columns and how far a needed cell may be from it (k_colpass_cull has the derivation)
    // (the last block's pad-only groups stay out of it: their boxes sit 1e18 away and would make the block need every cell)
    const GroupMeta* __restrict__ own = tmeta + (int64_t)blockIdx.x * kOwnGroups;
    const int64_t left = n - (int64_t)blockIdx.x * kCols;
    const int real_groups = (int)(((left < kCols ? left : kCols) + 31) >> 5);  // 1 .. kOwnGroups
    float lo[3], hi[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        lo[k] = own[0].lo[k];
        hi[k] = own[0].hi[k];
#pragma unroll
        for (int q = 1; q < kOwnGroups; ++q)
            if (q < real_groups) {
                lo[k] = fminf(lo[k], own[q].lo[k]);
                hi[k] = fmaxf(hi[k], own[q].hi[k]);
            }
    }
    float thr = INFINITY;
    if (colmin_g) {
        const int64_t gw = (int64_t)blockIdx.x * kOwnGroups;
        float cmax = colmin_g[gw];
#pragma unroll
        for (int q = 1; q < kOwnGroups; ++q) cmax = fmaxf(cmax, colmin_g[gw + q]);
        const float r = sqrtf(cmax) + __uint_as_float(*motion);
        thr = r * r * 1.00001f + kCullLog2 / kk;
    }
    T x = C::splat(0.f), y = C::splat(0.f), z = C::splat(0.f);
    bool have_points = false;
    T run = C::splat(INFINITY), off = C::splat(INFINITY), s = C::splat(0.f);
    T ux = C::splat(0.f), uy = C::splat(0.f), uz = C::splat(0.f), rr = C::splat(0.f);
    int ngrp = 0;  // (64 CPL x 32) blocks of pairs this wave evaluates (wave-uniform)
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
                C::points(tgt4 + n0, x, y, z);
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
                        T dx[kOwnerSub], dy[kOwnerSub], dz[kOwnerSub], d2[kOwnerSub];
#pragma unroll
                        for (int cc = 0; cc < kOwnerSub; ++cc) {
                            dx[cc] = x - C::splat(qa.q[h2 + cc].x);
                            dy[cc] = y - C::splat(qa.q[h2 + cc].y);
                            dz[cc] = z - C::splat(qa.q[h2 + cc].z);
                            d2[cc] = C::fma(dz[cc], dz[cc], C::fma(dy[cc], dy[cc], C::fma(dx[cc], dx[cc], C::splat(qa.q[h2 + cc].w))));
                        }
                        T cmn = d2[0];
#pragma unroll
                        for (int cc = 1; cc < kOwnerSub; ++cc) cmn = C::min(cmn, d2[cc]);
                        if (C::any_less(cmn, run)) {  // rare after the first trips: all five sums move to the new minimum
                            const T nm = C::min(run, cmn);
                            const T noff = C::coff(kk, nm);
                            const T f = C::exp2(noff - off);  // first use: off == +inf -> 0, and the sums are 0 anyway
                            s *= f; ux *= f; uy *= f; uz *= f; rr *= f;
                            run = nm;
                            off = noff;
                        }
#pragma unroll
                        for (int cc = 0; cc < kOwnerSub; ++cc) {
                            const T pr = C::exp2(C::fma(d2[cc], C::splat(kk), off));
                            s += pr;
                            ux = C::fma(pr, dx[cc], ux);
                            uy = C::fma(pr, dy[cc], uy);
                            uz = C::fma(pr, dz[cc], uz);
                            rr = C::fma(pr, d2[cc], rr);
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
        partr[wv][0][lane] = C::store(run);
        partr[wv][1][lane] = C::store(s);
        partr[wv][2][lane] = C::store(ux);
        partr[wv][3][lane] = C::store(uy);
        partr[wv][4][lane] = C::store(uz);
        partr[wv][5][lane] = C::store(rr);
    }
    int last = 0;
    if (lane == 0) {
        wave_groups[wv] = ngrp;
        last = atomicAdd(&arrived, 1) == kWaves - 1;  // LDS ops of a wave execute in order: its sums are visible
    }
    if (!__builtin_amdgcn_readfirstlane(last)) return;
    int total = 0;
    run = C::splat(INFINITY);
    off = C::splat(INFINITY);
    s = ux = uy = uz = rr = C::splat(0.f);
#pragma unroll
    for (int k = 0; k < kWaves; ++k) {
        const int tk = wave_groups[k];
        total += tk;
        if (!tk) continue;
        const T orun = C::load(partr[k][0][lane]);
        const T nm = C::min(run, orun);
        const T noff = C::coff(kk, nm);
        const T fa = C::exp2(noff - off), fb = C::exp2(noff - C::coff(kk, orun));
        s = s * fa + C::load(partr[k][1][lane]) * fb;
        ux = ux * fa + C::load(partr[k][2][lane]) * fb;
        uy = uy * fa + C::load(partr[k][3][lane]) * fb;
        uz = uz * fa + C::load(partr[k][4][lane]) * fb;
        rr = rr * fa + C::load(partr[k][5][lane]) * fb;
        run = nm;
        off = noff;
    }
    if (lane == 0) {
        wgcount[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = (unsigned)total;
        colflag[(int64_t)blockIdx.x * gridDim.y + blockIdx.y] = total ? 1 : 0;
    }
    if (!total) return;  // an untouched (block, plane) leaves nothing to read
    float* __restrict__ o = colpart + (int64_t)blockIdx.y * 6 * ncap + n0;
    typedef typename C::S S;
    *reinterpret_cast<S*>(o) = C::store(run);
    *reinterpret_cast<S*>(o + ncap) = C::store(s);
    *reinterpret_cast<S*>(o + 2 * ncap) = C::store(ux);
    *reinterpret_cast<S*>(o + 3 * ncap) = C::store(uy);
    *reinterpret_cast<S*>(o + 4 * ncap) = C::store(uz);
    *reinterpret_cast<S*>(o + 5 * ncap) = C::store(rr);
}

}  // namespace

namespace prg {

// Parts S the stream is dealt out over per column block.  The sweep is bound by its arithmetic in the mid regime and the chip
// takes whole workgroups as they come: ~2.5 rounds of its 8192 wave slots (256 CUs x 4 SIMDs x 8) even the load out; far more
// parts than work cost their launch in the late regime.  Measured (whole EM iterations, ms; round 5's queue / grid beside it):
//   C1, one GPU      iteration 12: S = 1 0.606, 4 0.541, 8 0.544, 16 0.561, queue 0.547 | iteration 19: 0.180, 0.180, 0.196, 0.220, queue 0.203
//   rank 3 of 8      iteration 7: S = 6 0.438, 16 0.375, 32 0.350, 64 0.367, grid 0.363 | iteration 19: 0.076, 0.086, 0.090, 0.090, grid 0.092
// C1 on one GPU: 782 blocks -> S = 3; a 1/8 shard: 98 blocks -> S = 26.
// columns a lane owns: 1 (a wave owns 64 columns: with kd-ordered clouds 25 - 35 % fewer pairs evaluated in the late regime than
// with 128, twice the workgroups to even the load out) or 2.  Measured, whole EM iterations (ms), 2 | 1 columns per lane:
//   C1, one GPU   iteration 12: 0.538 | 0.535   iteration 19: 0.171 | 0.144   iterations 45..49: 8 250 | 9 980 it/s   window 901 | 906 it/s
//   rank 3 of 8   iteration 7: 0.361 | 0.373    iteration 9: 0.208 | 0.200    iteration 19: 0.082 | 0.066
int owner_cols_per_lane() {
    static const int cpl = getenv("PRG_OWNER_CPL") ? atoi(getenv("PRG_OWNER_CPL")) : kOwnerColsPerLane;
    return cpl == 1 ? 1 : 2;
}

int owner_planes(int64_t owned, int64_t streamed) {
    static const int env = getenv("PRG_OWNER_PLANES") ? atoi(getenv("PRG_OWNER_PLANES")) : 0;
    const int64_t blocks = ceil_div(owned, 64 * owner_cols_per_lane()), chunks = ceil_div(streamed, kSuper);
    int64_t s = env > 0 ? env : (20000 + blocks * kOwnerWaves / 2) / (blocks * kOwnerWaves);
    if (env <= 0) s = std::min<int64_t>(s, std::max<int64_t>(1, chunks / kOwnerWaves));  // (at least one chunk per wave)
    return (int)std::max<int64_t>(1, std::min<int64_t>(s, kOwnerMaxPlanes));
}

void launch_colpass_owner(prg_cpd* h, bool use_seed, int planes) {
    const int cpl = owner_cols_per_lane();
    dim3 grid((unsigned)ceil_div(h->N, 64 * cpl), (unsigned)planes);
#define PRG_OWNER_ARGS                                                                                                              \
    h->tgt4, h->z4, reinterpret_cast<const GroupMeta*>(h->zchunk), reinterpret_cast<const GroupMeta*>(h->zmeta),                    \
        reinterpret_cast<const GroupMeta*>(h->tmeta), h->N, (int)ceil_div(h->M, kSuper), (int)ceil_div(h->M, kGroup), h->params,    \
        use_seed ? h->colmin + h->Ncap : nullptr, h->motion + ((h->estep_count - 1) & 1), reinterpret_cast<float*>(h->colpart),     \
        h->Ncap, h->wgcount, resid_flags(h, planes)
    if (cpl == 1)
        k_colpass_owner<1><<<grid, kThreads, 0, h->stream>>>(PRG_OWNER_ARGS);
    else
        k_colpass_owner<2><<<grid, kThreads, 0, h->stream>>>(PRG_OWNER_ARGS);
#undef PRG_OWNER_ARGS
    h->wg_col = (int64_t)grid.x * grid.y;
    h->wg_col_pairs = 64.0 * cpl * kGroup;
    h->dense_pairs_col = 0.0;
}

}  // namespace prg
