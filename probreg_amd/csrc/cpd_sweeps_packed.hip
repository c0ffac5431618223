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
    if (R == 2)
        k_colpass<1><<<grid, kBlock, 0, h->stream>>>(h->tgt4, h->z4, seg_len, h->params, h->colpart, h->Ncap);
    else
        k_colpass<2><<<grid, kBlock, 0, h->stream>>>(h->tgt4, h->z4, seg_len, h->params, h->colpart, h->Ncap);
}

void launch_rowpass_packed(prg_cpd* h, int R, int S, int seg_len) {
    dim3 grid((unsigned)ceil_div(h->M, kBlock * R), (unsigned)S);
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

constexpr float kCullLog2 = -127.0f;

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

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

// Which groups does a wave need?  Every lane tests ONE group of the segment against the wave's box (a 32-byte
// vector load of the group's metadata, a few VALU ops) and a ballot turns the 64 verdicts into a bit mask - one
// memory round trip for up to 2048 streamed points, where a scalar walk over super-group and group boxes paid a
// dependent scalar load per box (late EM iterations were bound by exactly that latency chain).  The mask also
// tells the wave which group comes NEXT, so the first quad of the next needed group is fetched while the current
// one is still being evaluated.

// Column pass with culling.  Lane owns the 2 adjacent columns n0 + 2*tid, +1.  `colmin_prev` (may be null) holds
// min_m d^2 of every column from the previous E-step and `motion` the largest displacement any source point made
// since: (sqrt(colmin) + motion)^2 bounds this iteration's minimum from above (triangle inequality), which is
// what makes a far group's contribution provably < 2^-127 of the final column sum.
__global__ __launch_bounds__(kBlock) void k_colpass_cull(const float4* __restrict__ tgt4, const float4* __restrict__ z4,
                                                         const GroupMeta* __restrict__ zmeta, int seg_len,
                                                         const double* __restrict__ params,
                                                         const float* __restrict__ colmin_prev,
                                                         const unsigned* __restrict__ motion,
                                                         float2* __restrict__ colpart, int64_t ncap) {
    const float kk = (float)(-kLog2e / (2.0 * params[13]));
    const int64_t n0 = (int64_t)blockIdx.x * (kBlock * 2) + 2 * threadIdx.x;
    const float4 a = tgt4[n0], b = tgt4[n0 + 1];
    const f2 x = {a.x, b.x}, y = {a.y, b.y}, z = {a.z, b.z};
    f2 run = splat(INFINITY), off = splat(INFINITY), s = splat(0.f);
    float lo[3], hi[3];
    lo[0] = wave_min(fminf(a.x, b.x)); hi[0] = wave_max(fmaxf(a.x, b.x));
    lo[1] = wave_min(fminf(a.y, b.y)); hi[1] = wave_max(fmaxf(a.y, b.y));
    lo[2] = wave_min(fminf(a.z, b.z)); hi[2] = wave_max(fmaxf(a.z, b.z));
    float thr = INFINITY;  // skip a group when its box is farther than thr (squared) from the wave's box
    if (colmin_prev) {
        const float delta = __uint_as_float(*motion);
        const float r0 = sqrtf(colmin_prev[n0]) + delta, r1 = sqrtf(colmin_prev[n0 + 1]) + delta;
        const float seed = wave_max(fmaxf(r0 * r0, r1 * r1)) * 1.00001f;
        thr = seed + kCullLog2 / kk;  // kk < 0: kk * (d2 - seed) < -127  <=>  d2 > seed + 127 / |kk|
    }
    const int lane = threadIdx.x & 63;
    const int64_t base = (int64_t)blockIdx.y * seg_len;
    const Quad* __restrict__ zp = reinterpret_cast<const Quad*>(z4 + base);
    const GroupMeta* __restrict__ gp = zmeta + base / prg::kGroup;
    const int ngroups = seg_len / prg::kGroup;
    for (int g0 = 0; g0 < ngroups; g0 += 64) {
        bool need = false;
        if (g0 + lane < ngroups) need = !(box_dist2(lo, hi, gp[g0 + lane]) > thr);
        unsigned long long mask = __ballot(need);
        if (mask == 0) continue;
        int g = g0 + __builtin_ctzll(mask);
        mask &= mask - 1;
        Quad qa = zp[(int64_t)g * 8];
        for (;;) {
            const int gnext = mask ? g0 + __builtin_ctzll(mask) : -1;
            mask &= mask - 1;  // (0 stays 0)
            const Quad* __restrict__ q = zp + (int64_t)g * 8;
            const Quad* __restrict__ qn = zp + (int64_t)(gnext >= 0 ? gnext : g) * 8;
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
            if (gnext < 0) break;
            g = gnext;
        }
    }
    float4* out = reinterpret_cast<float4*>(colpart + (int64_t)blockIdx.y * ncap + n0);
    *out = make_float4(run.x, s.x, run.y, s.y);
}

// Row pass with culling.  Lane owns the 2 adjacent rows m0 + 2*tid, +1; a group is skipped when
// kk * dist2(boxes) + max_n b_n < -127, i.e. every P of the block comes out of v_exp_f32 as exactly 0.
__global__ __launch_bounds__(kBlock) void k_rowpass_cull(const float4* __restrict__ z4, const float4* __restrict__ tgt4,
                                                         const GroupMeta* __restrict__ tmeta, int seg_len,
                                                         const double* __restrict__ params,
                                                         float* __restrict__ rowpart, int64_t mcap,
                                                         unsigned char* __restrict__ rowflag) {
    const float kk = (float)(-kLog2e / (2.0 * params[13]));
    const int64_t m0 = (int64_t)blockIdx.x * (kBlock * 2) + 2 * threadIdx.x;
    const float4 a = z4[m0], b = z4[m0 + 1];
    bool touched = false;  // wave-uniform: did this wave evaluate any pair of its (128 rows x segment) block?
    const f2 zx = {a.x, b.x}, zy = {a.y, b.y}, zz = {a.z, b.z}, zq = {a.w, b.w};  // zq: weight term, 0 for plain CPD
    f2 p1 = splat(0.f), ux = splat(0.f), uy = splat(0.f), uz = splat(0.f), e = splat(0.f);
    float lo[3], hi[3];
    lo[0] = wave_min(fminf(a.x, b.x)); hi[0] = wave_max(fmaxf(a.x, b.x));
    lo[1] = wave_min(fminf(a.y, b.y)); hi[1] = wave_max(fmaxf(a.y, b.y));
    lo[2] = wave_min(fminf(a.z, b.z)); hi[2] = wave_max(fmaxf(a.z, b.z));
    const int lane = threadIdx.x & 63;
    const int64_t base = (int64_t)blockIdx.y * seg_len;
    const Quad* __restrict__ tp = reinterpret_cast<const Quad*>(tgt4 + base);
    const GroupMeta* __restrict__ gp = tmeta + base / prg::kGroup;
    const int ngroups = seg_len / prg::kGroup;
    for (int g0 = 0; g0 < ngroups; g0 += 64) {
        bool need = false;
        if (g0 + lane < ngroups) {
            const GroupMeta gm = gp[g0 + lane];  // gm.aux = max b_n over the 32 points
            need = !(fmaf(box_dist2(lo, hi, gm), kk, gm.aux) < kCullLog2);
        }
        unsigned long long mask = __ballot(need);
        if (mask == 0) continue;
        touched = true;
        int g = g0 + __builtin_ctzll(mask);
        mask &= mask - 1;
        Quad cq = tp[(int64_t)g * 8];
        for (;;) {
            const int gnext = mask ? g0 + __builtin_ctzll(mask) : -1;
            mask &= mask - 1;
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
    // k_row_moments skips the partials of untouched (wave, segment) blocks: they are neither written nor read
    if (lane == 0)
        rowflag[((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64 + blockIdx.y] = touched ? 1 : 0;  // [wave block][segment]
    if (!touched) return;
    float* __restrict__ o = rowpart + (int64_t)blockIdx.y * 5 * mcap + m0;
    *reinterpret_cast<float2*>(o) = make_float2(p1.x, p1.y);
    *reinterpret_cast<float2*>(o + mcap) = make_float2(-ux.x, -ux.y);
    *reinterpret_cast<float2*>(o + 2 * mcap) = make_float2(-uy.x, -uy.y);
    *reinterpret_cast<float2*>(o + 3 * mcap) = make_float2(-uz.x, -uz.y);
    *reinterpret_cast<float2*>(o + 4 * mcap) = make_float2(fmaf(-zq.x, p1.x, e.x), fmaf(-zq.y, p1.y, e.y));
}

}  // namespace

namespace prg {

void launch_colpass_cull(prg_cpd* h, int S, int seg_len, bool use_seed) {
    dim3 grid((unsigned)ceil_div(h->N, kBlock * 2), (unsigned)S);
    k_colpass_cull<<<grid, kBlock, 0, h->stream>>>(h->tgt4, h->z4, reinterpret_cast<const GroupMeta*>(h->zmeta), seg_len,
                                                   h->params,
                                                   use_seed ? h->colmin : nullptr, h->motion + ((h->estep_count - 1) & 1), h->colpart, h->Ncap);
}

void launch_rowpass_cull(prg_cpd* h, int S, int seg_len) {
    dim3 grid((unsigned)ceil_div(h->M, kBlock * 2), (unsigned)S);
    k_rowpass_cull<<<grid, kBlock, 0, h->stream>>>(h->z4, h->tgt4, reinterpret_cast<const GroupMeta*>(h->tmeta), seg_len,
                                                   h->params,
                                                   h->rowpart, h->Mcap,
                                                   reinterpret_cast<unsigned char*>(h->rowpart + (int64_t)S * 5 * h->Mcap));
}

}  // namespace prg
