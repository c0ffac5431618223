// E-step pair sweeps, packed-fp32 form (CDNA4 v_pk_add/mul/fma_f32), MI355X gfx950.
//
// Both kernels follow one pattern: a lane OWNS R points of one cloud (registers) and the other
// cloud is streamed through SGPRs: the stream pointer is wave-uniform and read-only, so each
// 64-byte struct load becomes one s_load_dwordx16 (4 points) that is prefetched one trip ahead,
// and every VALU instruction takes the streamed coordinate as a scalar (broadcast) operand - no
// LDS traffic, no VGPRs for the streamed tile.  The M x N matrix P (cpd.py:74-84) never exists.
#include <math.h>

#include "cpd_sweeps.h"

namespace {

typedef float f2 __attribute__((ext_vector_type(2)));
struct alignas(64) Quad { float4 q[4]; };

constexpr double kLog2e = 1.4426950408889634;
constexpr int kBlock = prg::kSweepBlock;

__device__ __forceinline__ f2 splat(float a) { return (f2){a, a}; }
__device__ __forceinline__ f2 exp2v(f2 a) { return (f2){__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)}; }
// Exponent offset of the running column sums: s = sum exp2(kk d2 + off) with off = -kk * run (>= 0), so a term
// costs one FMA + one exp; k_colfinal recomputes the offset bit for bit from the stored minimum
// (prg::col_offset in cpd_sweeps.h).
__device__ __forceinline__ f2 col_offset2(float kk, f2 run) {
    return (f2){prg::col_offset(kk, run.x), prg::col_offset(kk, run.y)};
}
__device__ __forceinline__ f2 fmav(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 minv(f2 a, f2 b) { return __builtin_elementwise_min(a, b); }

// ---- sweep 1: den_n of cpd.py:80 as an online (min d^2, sum exp2(kk (d^2 - min))) pair ---------
// grid = (ceil(N / (256 R)), S); block b.y streams source points [b.y*seg_len, +seg_len).
template <int RP>
__global__ __launch_bounds__(kBlock) void k_colpass(const float4* __restrict__ tgt4, const float4* __restrict__ z4,
                                                    int seg_len, const double* __restrict__ params,
                                                    float2* __restrict__ colpart, int64_t ncap) {
    constexpr int R = 2 * RP;
    const float kk = (float)(-kLog2e / (2.0 * params[13]));
    const int64_t n0 = (int64_t)blockIdx.x * (kBlock * R) + threadIdx.x;
    f2 x[RP], y[RP], z[RP], run[RP], off[RP], s[RP];
#pragma unroll
    for (int p = 0; p < RP; ++p) {
        const float4 a = tgt4[n0 + (2 * p) * kBlock], b = tgt4[n0 + (2 * p + 1) * kBlock];
        x[p] = (f2){a.x, b.x};
        y[p] = (f2){a.y, b.y};
        z[p] = (f2){a.z, b.z};
        run[p] = splat(INFINITY);
        off[p] = splat(INFINITY);
        s[p] = splat(0.f);
    }
    const Quad* __restrict__ zp = reinterpret_cast<const Quad*>(z4 + (int64_t)blockIdx.y * seg_len);
    const int nq = seg_len >> 2;
    Quad cur = zp[0];
    for (int m = 0; m < nq; m += 2) {
        const Quad nx0 = zp[m + 1];
        f2 d2[8][RP];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int p = 0; p < RP; ++p) {
                const f2 dx = x[p] - splat(cur.q[c].x), dy = y[p] - splat(cur.q[c].y), dz = z[p] - splat(cur.q[c].z);
                d2[c][p] = fmav(dz, dz, fmav(dy, dy, fmav(dx, dx, splat(cur.q[c].w))));
            }
        const Quad nx1 = zp[m + 2];  // prefetch for the next trip (pads make the over-read safe)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int p = 0; p < RP; ++p) {
                const f2 dx = x[p] - splat(nx0.q[c].x), dy = y[p] - splat(nx0.q[c].y), dz = z[p] - splat(nx0.q[c].z);
                d2[4 + c][p] = fmav(dz, dz, fmav(dy, dy, fmav(dx, dx, splat(nx0.q[c].w))));
            }
        cur = nx1;
        bool lower = false;
        f2 cm[RP];
#pragma unroll
        for (int p = 0; p < RP; ++p) {
            f2 v = d2[0][p];
#pragma unroll
            for (int c = 1; c < 8; ++c) v = minv(v, d2[c][p]);
            cm[p] = v;
            lower |= (v.x < run[p].x) | (v.y < run[p].y);
        }
        if (lower) {  // rare after the first trips: move the running sums to the new minimum
#pragma unroll
            for (int p = 0; p < RP; ++p) {
                const f2 nm = minv(run[p], cm[p]);
                const f2 noff = col_offset2(kk, nm);
                s[p] *= exp2v(noff - off[p]);  // first use: off == +inf -> exp2(-inf) = 0, and s == 0 anyway
                run[p] = nm;
                off[p] = noff;
            }
        }
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
            for (int p = 0; p < RP; ++p) s[p] += exp2v(fmav(d2[c][p], splat(kk), off[p]));
    }
#pragma unroll
    for (int p = 0; p < RP; ++p) {
        colpart[(int64_t)blockIdx.y * ncap + n0 + (2 * p) * kBlock] = make_float2(run[p].x, s[p].x);
        colpart[(int64_t)blockIdx.y * ncap + n0 + (2 * p + 1) * kBlock] = make_float2(run[p].y, s[p].y);
    }
}

// ---- sweep 2: p1, px, and the sigma2 residual of cpd.py:84-87 in residual form ------------------
// grid = (ceil(M / (256 R)), S); block b.y streams target points [b.y*seg_len, +seg_len) as
// (x, y, z, b_n) with b_n = -log2(den_n + c), so P_mn = exp2(kk d2 + b_n) costs one FMA + one exp.
// Output planes: rowpart[(seg*5 + comp) * Mcap + m], comp = p1, ux, uy, uz, e with
// u = sum_n P (x_n - z_m), e = sum_n P |x_n - z_m|^2.
template <int RP>
__global__ __launch_bounds__(kBlock) void k_rowpass(const float4* __restrict__ z4, const float4* __restrict__ tgt4,
                                                    int seg_len, const double* __restrict__ params,
                                                    float* __restrict__ rowpart, int64_t mcap) {
    constexpr int R = 2 * RP;
    const float kk = (float)(-kLog2e / (2.0 * params[13]));
    const int64_t m0 = (int64_t)blockIdx.x * (kBlock * R) + threadIdx.x;
    f2 zx[RP], zy[RP], zz[RP], zq[RP], p1[RP], ux[RP], uy[RP], uz[RP], e[RP];
#pragma unroll
    for (int p = 0; p < RP; ++p) {
        const float4 a = z4[m0 + (2 * p) * kBlock], b = z4[m0 + (2 * p + 1) * kBlock];
        zx[p] = (f2){a.x, b.x};
        zy[p] = (f2){a.y, b.y};
        zz[p] = (f2){a.z, b.z};
        zq[p] = (f2){a.w, b.w};  // q_m >= 0: per-source weight as an additive squared distance (0 for plain CPD)
        p1[p] = ux[p] = uy[p] = uz[p] = e[p] = splat(0.f);
    }
    const Quad* __restrict__ tp = reinterpret_cast<const Quad*>(tgt4 + (int64_t)blockIdx.y * seg_len);
    const int nq = seg_len >> 2;
    Quad cur = tp[0];
    for (int n = 0; n < nq; ++n) {
        const Quad nxt = tp[n + 1];  // prefetch
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int p = 0; p < RP; ++p) {
                const f2 dx = zx[p] - splat(cur.q[c].x), dy = zy[p] - splat(cur.q[c].y),
                         dz = zz[p] - splat(cur.q[c].z);
                const f2 d = fmav(dz, dz, fmav(dy, dy, fmav(dx, dx, zq[p])));
                const f2 pr = exp2v(fmav(d, splat(kk), splat(cur.q[c].w)));
                p1[p] += pr;
                ux[p] = fmav(pr, dx, ux[p]);
                uy[p] = fmav(pr, dy, uy[p]);
                uz[p] = fmav(pr, dz, uz[p]);
                e[p] = fmav(pr, d, e[p]);
            }
        }
        cur = nxt;
    }
    float* __restrict__ o = rowpart + (int64_t)blockIdx.y * 5 * mcap;
#pragma unroll
    for (int p = 0; p < RP; ++p) {
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int64_t m = m0 + (2 * p + hh) * kBlock;
            o[m] = p1[p][hh];
            o[mcap + m] = -ux[p][hh];  // u = sum P (x - z) = -sum P (z - x)
            o[2 * mcap + m] = -uy[p][hh];
            o[3 * mcap + m] = -uz[p][hh];
            o[4 * mcap + m] = fmaf(-zq[p][hh], p1[p][hh], e[p][hh]);  // e accumulated d2 + q: take q p1 back out
        }
    }
}

}  // namespace

namespace prg {

void launch_colpass_packed(prg_cpd* h, int R, int S, int seg_len) {
    dim3 grid((unsigned)ceil_div(h->N, kBlock * R), (unsigned)S);
    h->wg_col = 0;
    h->dense_pairs_col = (double)grid.x * (kBlock * R) * (double)S * seg_len;
    if (R == 2)
        k_colpass<1><<<grid, kBlock, 0, h->stream>>>(h->tgt4, h->z4, seg_len, h->params, h->colpart, h->Ncap);
    else
        k_colpass<2><<<grid, kBlock, 0, h->stream>>>(h->tgt4, h->z4, seg_len, h->params, h->colpart, h->Ncap);
}

void launch_rowpass_packed(prg_cpd* h, int R, int S, int seg_len) {
    dim3 grid((unsigned)ceil_div(h->M, kBlock * R), (unsigned)S);
    h->wg_row = 0;
    h->dense_pairs_row = (double)grid.x * (kBlock * R) * (double)S * seg_len;
    if (R == 2)
        k_rowpass<1><<<grid, kBlock, 0, h->stream>>>(h->z4, h->tgt4, seg_len, h->params, h->rowpart, h->Mcap);
    else
        k_rowpass<2><<<grid, kBlock, 0, h->stream>>>(h->z4, h->tgt4, seg_len, h->params, h->rowpart, h->Mcap);
}

}  // namespace prg

// =============================================================================================
// Culled sweeps.  Clouds are Morton-sorted at upload, so the 128 points a wave owns
// and every group of 32 streamed points are spatially compact.  Per group one axis-aligned bounding box
// (plus max b_n for the row pass) is fetched through SGPRs; if EVERY pair of the (wave, group) block is
// provably an exact zero - exp2 argument below -127 - the whole group is skipped.  v_exp_f32 (the sweeps call it
// raw) flushes every result below 2^-126 to 0 (measured: tools/exp_denormal_probe.py, exp2(-126) already returns
// 0), so the dense kernels produce exactly 0 for these pairs too; the extra unit absorbs the fp32 rounding of the
// bound itself.  Skipped terms are exactly 0 (row pass) or below 2^-127 of a sum that is >= 1 (column pass), so
// the results are those of the dense sweeps; what changes is that late EM iterations touch ~1 % of the pairs.
// =============================================================================================
namespace {

constexpr float kCullLog2 = -prg::kCullExp;  // (cpd_sweeps.h)

#ifdef PRG_WAVE_TRACE
// Instrumented build (tools/wave_trace.py): every wave of the culled sweeps records its start / end shader clock,
// the hardware id it ran on and how many groups it evaluated: trace[(kernel * max_waves + wave) * 4 + {0..3}].
__device__ unsigned long long* g_wave_trace = nullptr;
__device__ unsigned long long g_wave_trace_cap = 0;
#define PRG_TRACE_BEGIN() const unsigned long long trace_t0 = __builtin_readcyclecounter(); unsigned trace_groups = 0
#define PRG_TRACE_GROUP() ++trace_groups
#define PRG_TRACE_END(kernel_id)                                                                                  \
    if (g_wave_trace && (threadIdx.x & 63) == 0) {                                                                \
        const unsigned long long w = ((unsigned long long)blockIdx.y * gridDim.x + blockIdx.x) * 4 + (threadIdx.x >> 6); \
        if (w < g_wave_trace_cap) {                                                                               \
            unsigned long long* o = g_wave_trace + ((unsigned long long)(kernel_id) * g_wave_trace_cap + w) * 4;  \
            o[0] = trace_t0;                                                                                      \
            o[1] = __builtin_readcyclecounter();                                                                  \
            o[2] = __builtin_amdgcn_s_getreg((4 << 11) | (0 << 6) | 4 /* HW_REG_HW_ID, 32 bits */);              \
            o[3] = trace_groups;                                                                                  \
        }                                                                                                         \
    }
#else
#define PRG_TRACE_BEGIN()
#define PRG_TRACE_GROUP()
#define PRG_TRACE_END(kernel_id)
#endif

struct alignas(32) GroupMeta { float lo[3]; float hi[3]; float aux; float pad; };

// squared distance between two axis-aligned boxes (0 if they overlap)
__device__ __forceinline__ float box_dist2(const float (&alo)[3], const float (&ahi)[3], const GroupMeta& g) {
    float d2 = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float gap = fmaxf(fmaxf(alo[k] - g.hi[k], g.lo[k] - ahi[k]), 0.f);
        d2 = fmaf(gap, gap, d2);
    }
    return d2;
}

// Bounding box of the 128 points a wave owns = union of the boxes of its four 32-point groups, which the metadata
// arrays already hold: scalar loads instead of two vector loads and 36 dependent cross-lane reduction steps - a
// wave that finds nothing to do (most waves, in late EM iterations) is gone after one round trip.
__device__ __forceinline__ void wave_box(const GroupMeta* __restrict__ own, float (&lo)[3], float (&hi)[3]) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        lo[k] = fminf(fminf(own[0].lo[k], own[1].lo[k]), fminf(own[2].lo[k], own[3].lo[k]));
        hi[k] = fmaxf(fmaxf(own[0].hi[k], own[1].hi[k]), fmaxf(own[2].hi[k], own[3].hi[k]));
    }
}

// Work mapping of the culled sweeps.  A workgroup owns 128 lane points (2 per lane) and FOUR consecutive segments
// of the stream, one per wave; the four partial results are merged through LDS, so the partial planes in HBM
// number S/4 while a wave's chain of dependent scalar loads is one short segment (512 points at C1).  That chain
// is what bounds a sparse E-step: per-wave timelines (tools/wave_trace.py) show busy waves at ~5400 cycles per
// needed group (scalar loads return out of order, every wait is a wait for everything, and there is nobody else
// on the SIMD to hide it) next to tens of thousands of waves that exit immediately - the sweep lasts as long as
// its longest wave.
//
// Which groups does a wave need?  Every lane tests ONE group of the segment against the wave's box (a 32-byte
// vector load of the group's metadata, a few VALU ops) and a ballot turns the 64 verdicts into a bit mask - one
// memory round trip for up to 2048 streamed points.  The mask also tells the wave which group comes NEXT, so the
// first quad of the next needed group is fetched while the current one is still being evaluated.

// Column pass with culling.  Lane owns the 2 adjacent columns n0 + 2*lane, +1.  `colmin_g` (may be null) holds,
// per group of 32 columns, the largest min_m d^2 of the previous E-step and `motion` the largest displacement any
// source point made since: (sqrt(colmin) + motion)^2 bounds this iteration's minimum from above (triangle
// inequality), which is what makes a far group's contribution provably < 2^-127 of the final column sum.
//
// RESID: the ONE sweep of a rigid EM iteration on the vector pipe.  Next to the online (min d^2, A = sum K)
// pair the lane keeps the RESIDUAL sums of its column, U = sum_m K (x_n - z_m) and R = sum_m K |x_n - z_m|^2, under the same
// rescaling (K = exp2(kk d^2 + off)): 4 more fma per pair, and everything the rigid M-step (cpd.py:160-192) consumes follows
// from (A, U, R) per column in fp64 (k_colfinal_resid) - no row pass, no per-source-point merge.  A chunk is 4 streamed
// points (the differences stay in registers); planes of 6 floats per column, [plane][6][ncap], and one byte per
// (128-column block, plane): touched or not - untouched partials are neither written nor read.
template <bool RESID>
__global__ __launch_bounds__(kBlock) void k_colpass_cull(const float4* __restrict__ tgt4, const float4* __restrict__ z4,
                                                         const GroupMeta* __restrict__ zmeta,
                                                         const GroupMeta* __restrict__ tmeta, int seg_len, int nseg,
                                                         const double* __restrict__ params,
                                                         const float* __restrict__ colmin_g,
                                                         const unsigned* __restrict__ motion,
                                                         float2* __restrict__ colpart, int64_t ncap,
                                                         unsigned* __restrict__ wgcount,
                                                         const EngineDecision* __restrict__ guard,
                                                         unsigned char* __restrict__ colflag) {
    PRG_TRACE_BEGIN();
    // launched ahead of the E-step's engine decision (cpd.hip, estep_impl; dense regime only - null afterwards): run
    // only if the decision names this engine
    if (guard && guard->col != 0) return;
    __shared__ float4 part[RESID ? 1 : 4][64];
    __shared__ float2 partr[RESID ? 4 : 1][6][64];
    __shared__ int arrived, wave_groups[4];
    int ngrp = 0;  // (wave, group) blocks this wave evaluates: 128 x 32 pairs each (measurement hook, wave-uniform)
    if (threadIdx.x == 0) arrived = 0;
    __syncthreads();  // (at launch, before any wave waits for memory: costs nothing; there is no barrier at the end)
    const float kk = (float)(-kLog2e / (2.0 * params[13]));
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t n0 = (int64_t)blockIdx.x * 128 + 2 * lane;
    const int seg = blockIdx.y * 4 + wv;
    f2 run = splat(INFINITY), off = splat(INFINITY), s = splat(0.f);
    f2 ux = splat(0.f), uy = splat(0.f), uz = splat(0.f), rr = splat(0.f);  // RESID: sum K (x - z), sum K |x - z|^2
    if (seg < nseg) {
        const int64_t gw = (int64_t)blockIdx.x * 4;  // the workgroup's four 32-column groups
        float lo[3], hi[3];
        wave_box(tmeta + gw, lo, hi);
        float thr = INFINITY;  // skip a group when its box is farther than thr (squared) from the wave's box
        if (colmin_g) {
            const float cmax = fmaxf(fmaxf(colmin_g[gw], colmin_g[gw + 1]), fmaxf(colmin_g[gw + 2], colmin_g[gw + 3]));
            const float r = sqrtf(cmax) + __uint_as_float(*motion);
            thr = r * r * 1.00001f + kCullLog2 / kk;  // kk < 0: kk * (d2 - seed) < -127  <=>  d2 > seed + 127 / |kk|
        }
        const int64_t base = (int64_t)seg * seg_len;
        const Quad* __restrict__ zp = reinterpret_cast<const Quad*>(z4 + base);
        const GroupMeta* __restrict__ gp = zmeta + base / prg::kGroup;
        const int ngroups = seg_len / prg::kGroup;
        // every load of the prologue is issued here, before anything is consumed: a wave that finds nothing to do -
        // most waves of a sparse E-step - then lives for one memory round trip, not one per dependent load
        GroupMeta gm = gp[min(lane, ngroups - 1)];
        // the lane's own points are fetched only once a group is needed: 150k waves x 2 KB of them is what a sparse
        // sweep otherwise spends its time on (tools/dispatch_floor.hip: the floor of a grid this size is its loads)
        f2 x = splat(0.f), y = splat(0.f), z = splat(0.f);
        bool have_points = false;
        for (int g0 = 0; g0 < ngroups; g0 += 64) {
            if (g0 > 0) gm = gp[min(g0 + lane, ngroups - 1)];
            const bool need = (g0 + lane < ngroups) && !(box_dist2(lo, hi, gm) > thr);
            unsigned long long mask = __ballot(need);
            if (mask == 0) continue;
            ngrp += __builtin_popcountll(mask);
            if (!have_points) {
                const float4 a = tgt4[n0], b = tgt4[n0 + 1];
                x = (f2){a.x, b.x};
                y = (f2){a.y, b.y};
                z = (f2){a.z, b.z};
                have_points = true;
            }
            int g = g0 + __builtin_ctzll(mask);
            mask &= mask - 1;
            Quad qa = zp[(int64_t)g * 8];
            for (;;) {
                const int gnext = mask ? g0 + __builtin_ctzll(mask) : -1;
                mask &= mask - 1;  // (0 stays 0)
                PRG_TRACE_GROUP();
                const Quad* __restrict__ q = zp + (int64_t)g * 8;
                const Quad* __restrict__ qn = zp + (int64_t)(gnext >= 0 ? gnext : g) * 8;
                if constexpr (RESID) {
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        const Quad nq = (t < 7) ? q[t + 1] : qn[0];  // prefetch: next quad, or the next needed group's first
                        f2 dx[4], dy[4], dz[4], d2[4];
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            dx[c] = x - splat(qa.q[c].x);
                            dy[c] = y - splat(qa.q[c].y);
                            dz[c] = z - splat(qa.q[c].z);
                            d2[c] = fmav(dz[c], dz[c], fmav(dy[c], dy[c], fmav(dx[c], dx[c], splat(qa.q[c].w))));
                        }
                        const f2 cm = minv(minv(d2[0], d2[1]), minv(d2[2], d2[3]));
                        if ((cm.x < run.x) | (cm.y < run.y)) {  // rare after the first trips: all five sums move to the new minimum
                            const f2 nm = minv(run, cm);
                            const f2 noff = col_offset2(kk, nm);
                            const f2 f = exp2v(noff - off);  // first use: off == +inf -> 0, and the sums are 0 anyway
                            s *= f; ux *= f; uy *= f; uz *= f; rr *= f;
                            run = nm;
                            off = noff;
                        }
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const f2 pr = exp2v(fmav(d2[c], splat(kk), off));
                            s += pr;
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
                    qa = (t < 3) ? q[2 * t + 2] : qn[0];  // last trip: first quad of the next needed group
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
                        s *= exp2v(noff - off);
                        run = nm;
                        off = noff;
                    }
#pragma unroll
                    for (int c = 0; c < 8; ++c) s += exp2v(fmav(d2[c], splat(kk), off));
                }
                }
                if (gnext < 0) break;
                g = gnext;
            }
        }
    }
    // Merge the four segments' (min, sum) pairs (sums are relative to the offset of their own minimum) without a
    // barrier: every wave leaves its pair in LDS and counts itself in; whoever arrives last does the merge.  An idle
    // wave is gone after its one round trip instead of holding a wave slot until its busiest sibling has finished.
    if constexpr (RESID) {
        if (ngrp) {  // an untouched wave contributes nothing and leaves nothing to read
            partr[wv][0][lane] = make_float2(run.x, run.y);
            partr[wv][1][lane] = make_float2(s.x, s.y);
            partr[wv][2][lane] = make_float2(ux.x, ux.y);
            partr[wv][3][lane] = make_float2(uy.x, uy.y);
            partr[wv][4][lane] = make_float2(uz.x, uz.y);
            partr[wv][5][lane] = make_float2(rr.x, rr.y);
        }
    } else {
        part[wv][lane] = make_float4(run.x, s.x, run.y, s.y);
    }
    int last = 0;
    if (lane == 0) {
        wave_groups[wv] = ngrp;
        last = atomicAdd(&arrived, 1) == 3;  // LDS ops of a wave execute in order: the pair is visible
    }
    if (__builtin_amdgcn_readfirstlane(last)) {
        const int tk[4] = {wave_groups[0], wave_groups[1], wave_groups[2], wave_groups[3]};
        if (lane == 0)  // one plain store per workgroup; summed on the host when the bench asks (prg_cpd_pair_counts)
            wgcount[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = (unsigned)(tk[0] + tk[1] + tk[2] + tk[3]);
        run = splat(INFINITY);
        off = splat(INFINITY);
        s = splat(0.f);
        if constexpr (RESID) {
            const bool any = (tk[0] | tk[1] | tk[2] | tk[3]) != 0;
            if (lane == 0) colflag[(int64_t)blockIdx.x * gridDim.y + blockIdx.y] = any ? 1 : 0;
            if (any) {
                ux = uy = uz = rr = splat(0.f);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (!tk[k]) continue;
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
                float* __restrict__ o = reinterpret_cast<float*>(colpart) + (int64_t)blockIdx.y * 6 * ncap + n0;
                *reinterpret_cast<float2*>(o) = make_float2(run.x, run.y);
                *reinterpret_cast<float2*>(o + ncap) = make_float2(s.x, s.y);
                *reinterpret_cast<float2*>(o + 2 * ncap) = make_float2(ux.x, ux.y);
                *reinterpret_cast<float2*>(o + 3 * ncap) = make_float2(uy.x, uy.y);
                *reinterpret_cast<float2*>(o + 4 * ncap) = make_float2(uz.x, uz.y);
                *reinterpret_cast<float2*>(o + 5 * ncap) = make_float2(rr.x, rr.y);
            }
        } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float4 o = part[k][lane];
            const f2 orun = {o.x, o.z}, os = {o.y, o.w};
            const f2 nm = minv(run, orun);
            const f2 noff = col_offset2(kk, nm);
            // (an empty side has sum 0 and offset FLT_MAX or +inf: its factor is exp2(-huge) = 0, never inf * 0)
            s = s * exp2v(noff - off) + os * exp2v(noff - col_offset2(kk, orun));
            run = nm;
            off = noff;
        }
        float4* out = reinterpret_cast<float4*>(colpart + (int64_t)blockIdx.y * ncap + n0);
        *out = make_float4(run.x, s.x, run.y, s.y);
        }
    }
    PRG_TRACE_END(0);
}

// Row pass with culling.  Lane owns the 2 adjacent rows m0 + 2*lane, +1; a group is skipped when
// kk * dist2(boxes) + max_n b_n < -127, i.e. every P of the block comes out of v_exp_f32 as exactly 0.
__global__ __launch_bounds__(kBlock) void k_rowpass_cull(const float4* __restrict__ z4, const float4* __restrict__ tgt4,
                                                         const GroupMeta* __restrict__ tmeta,
                                                         const GroupMeta* __restrict__ zmeta, int seg_len, int nseg,
                                                         const double* __restrict__ params,
                                                         float* __restrict__ rowpart, int64_t mcap,
                                                         unsigned char* __restrict__ rowflag,
                                                         unsigned* __restrict__ wgcount) {
    PRG_TRACE_BEGIN();
    __shared__ float2 part[4][5][64];
    __shared__ int arrived, wave_touched[4];
    int ngrp = 0;  // evaluated (wave, group) blocks, as in k_colpass_cull
    if (threadIdx.x == 0) arrived = 0;
    __syncthreads();  // (at launch; the merge at the end is barrier-free, see k_colpass_cull)
    const float kk = (float)(-kLog2e / (2.0 * params[13]));
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t m0 = (int64_t)blockIdx.x * 128 + 2 * lane;
    const int seg = blockIdx.y * 4 + wv;
    bool touched = false;  // wave-uniform: did this wave evaluate any pair of its (128 rows x segment) block?
    f2 p1 = splat(0.f), ux = splat(0.f), uy = splat(0.f), uz = splat(0.f), e = splat(0.f);
    if (seg < nseg) {
        float lo[3], hi[3];
        wave_box(zmeta + (int64_t)blockIdx.x * 4, lo, hi);
        const int64_t base = (int64_t)seg * seg_len;
        const Quad* __restrict__ tp = reinterpret_cast<const Quad*>(tgt4 + base);
        const GroupMeta* __restrict__ gp = tmeta + base / prg::kGroup;
        const int ngroups = seg_len / prg::kGroup;
        GroupMeta gm = gp[min(lane, ngroups - 1)];  // gm.aux = max b_n over the group's 32 points
        f2 zx = splat(0.f), zy = splat(0.f), zz = splat(0.f), zq = splat(0.f);  // fetched when first needed
        for (int g0 = 0; g0 < ngroups; g0 += 64) {
            if (g0 > 0) gm = gp[min(g0 + lane, ngroups - 1)];
            const bool need = (g0 + lane < ngroups) && !(fmaf(box_dist2(lo, hi, gm), kk, gm.aux) < kCullLog2);
            unsigned long long mask = __ballot(need);
            if (mask == 0) continue;
            ngrp += __builtin_popcountll(mask);
            if (!touched) {
                const float4 a = z4[m0], b = z4[m0 + 1];
                zx = (f2){a.x, b.x};
                zy = (f2){a.y, b.y};
                zz = (f2){a.z, b.z};
                zq = (f2){a.w, b.w};  // weight term, 0 for plain CPD
            }
            touched = true;
            int g = g0 + __builtin_ctzll(mask);
            mask &= mask - 1;
            Quad cq = tp[(int64_t)g * 8];
            for (;;) {
                const int gnext = mask ? g0 + __builtin_ctzll(mask) : -1;
                mask &= mask - 1;
                PRG_TRACE_GROUP();
                const Quad* __restrict__ q = tp + (int64_t)g * 8;
                const Quad* __restrict__ qn = tp + (int64_t)(gnext >= 0 ? gnext : g) * 8;
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const Quad nq = (t < 7) ? q[t + 1] : qn[0];  // prefetch: next quad, or the next needed group's first
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
                if (gnext < 0) break;
                g = gnext;
            }
        }
    }
    if (touched) {  // an untouched wave contributes nothing and leaves nothing to read
        part[wv][0][lane] = make_float2(p1.x, p1.y);
        part[wv][1][lane] = make_float2(ux.x, ux.y);
        part[wv][2][lane] = make_float2(uy.x, uy.y);
        part[wv][3][lane] = make_float2(uz.x, uz.y);
        part[wv][4][lane] = make_float2(e.x, e.y);
    }
    int last = 0;
    if (lane == 0) {
        wave_touched[wv] = touched ? ngrp : 0;  // (count of evaluated groups: non-zero iff touched)
        last = atomicAdd(&arrived, 1) == 3;
    }
    if (__builtin_amdgcn_readfirstlane(last)) {
        // k_row_moments skips the partials of untouched (128-row block, plane) pairs: neither written nor read
        const int t0 = wave_touched[0], t1 = wave_touched[1], t2 = wave_touched[2], t3 = wave_touched[3];
        const bool any = (t0 | t1 | t2 | t3) != 0;
        if (lane == 0) {
            rowflag[(int64_t)blockIdx.x * 64 + blockIdx.y] = any ? 1 : 0;
            wgcount[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = (unsigned)(t0 + t1 + t2 + t3);
        }
        if (any) {
            const int tk[4] = {t0, t1, t2, t3};
            p1 = ux = uy = uz = e = splat(0.f);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (!tk[k]) continue;
                const float2 q0 = part[k][0][lane], q1 = part[k][1][lane], q2 = part[k][2][lane], q3 = part[k][3][lane],
                             q4 = part[k][4][lane];
                p1 += (f2){q0.x, q0.y};
                ux += (f2){q1.x, q1.y};
                uy += (f2){q2.x, q2.y};
                uz += (f2){q3.x, q3.y};
                e += (f2){q4.x, q4.y};
            }
            const float4 za = z4[m0], zb = z4[m0 + 1];  // the merging wave may be one that never loaded its rows
            float* __restrict__ o = rowpart + (int64_t)blockIdx.y * 5 * mcap + m0;
            *reinterpret_cast<float2*>(o) = make_float2(p1.x, p1.y);
            *reinterpret_cast<float2*>(o + mcap) = make_float2(-ux.x, -ux.y);
            *reinterpret_cast<float2*>(o + 2 * mcap) = make_float2(-uy.x, -uy.y);
            *reinterpret_cast<float2*>(o + 3 * mcap) = make_float2(-uz.x, -uz.y);
            *reinterpret_cast<float2*>(o + 4 * mcap) = make_float2(fmaf(-za.w, p1.x, e.x), fmaf(-zb.w, p1.y, e.y));
        }
    }
    PRG_TRACE_END(1);
}

}  // namespace

namespace prg {

#ifdef PRG_WAVE_TRACE
extern "C" int prg_debug_set_wave_trace(unsigned long long* dev_buffer, unsigned long long max_waves) {
    PRG_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_wave_trace), &dev_buffer, sizeof(dev_buffer)));
    PRG_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_wave_trace_cap), &max_waves, sizeof(max_waves)));
    return PRG_OK;
}
#endif

// S segments of seg_len streamed points, four per workgroup: S/4 (rounded up) partial planes
void launch_colpass_cull(prg_cpd* h, int S, int seg_len, bool use_seed, const EngineDecision* guard, bool resid) {
    dim3 grid((unsigned)ceil_div(h->N, 128), (unsigned)ceil_div(S, 4));
    if (resid)  // planes of 6 floats per column, then one touched-flag byte per (128-column block, plane)
        k_colpass_cull<true><<<grid, kBlock, 0, h->stream>>>(h->tgt4, h->z4, reinterpret_cast<const GroupMeta*>(h->zmeta),
                                                             reinterpret_cast<const GroupMeta*>(h->tmeta), seg_len, S, h->params,
                                                             use_seed ? h->colmin + h->Ncap : nullptr,
                                                             h->motion + ((h->estep_count - 1) & 1), h->colpart, h->Ncap,
                                                             h->wgcount, guard, resid_flags(h, (int)grid.y));
    else
    k_colpass_cull<false><<<grid, kBlock, 0, h->stream>>>(h->tgt4, h->z4, reinterpret_cast<const GroupMeta*>(h->zmeta),
                                                   reinterpret_cast<const GroupMeta*>(h->tmeta), seg_len, S, h->params,
                                                   use_seed ? h->colmin + h->Ncap : nullptr,
                                                   h->motion + ((h->estep_count - 1) & 1), h->colpart, h->Ncap,
                                                   h->wgcount, guard, nullptr);
    h->wg_col = (int64_t)grid.x * grid.y;
    h->wg_col_pairs = 128.0 * kGroup;  // (a mispredicted matrix-core launch ahead of this one has left its own unit here)
    h->dense_pairs_col = 0.0;
}

void launch_rowpass_cull(prg_cpd* h, int S, int seg_len) {
    const int planes = (int)ceil_div(S, 4);
    dim3 grid((unsigned)ceil_div(h->M, 128), (unsigned)planes);
    k_rowpass_cull<<<grid, kBlock, 0, h->stream>>>(h->z4, h->tgt4, reinterpret_cast<const GroupMeta*>(h->tmeta),
                                                   reinterpret_cast<const GroupMeta*>(h->zmeta), seg_len, S, h->params,
                                                   h->rowpart, h->Mcap,
                                                   reinterpret_cast<unsigned char*>(h->rowpart + (int64_t)planes * 5 * h->Mcap),
                                                   h->wgcount + h->wg_cap);
    h->wg_row = (int64_t)grid.x * grid.y;
    h->wg_row_pairs = 128.0 * kGroup;
    h->dense_pairs_row = 0.0;
}

}  // namespace prg
