// E-step pair sweeps with the squared distances on the matrix cores, MI355X gfx950 - the DENSE regime of CPD's E-step
// (cpd.py:71-88).
//
// While sigma2 is large every source-target pair contributes and the VALU sweeps of cpd_sweeps_packed.hip are bound by
// their ~13 vector instructions per pair, 7 of which form the exponent.  Here the exponent comes from the matrix pipe:
//
//   kk |x - z|^2 + b  =  x' . (-2kk z')  +  kk |z'|^2  +  (kk |x'|^2 + b)
//
// with every f32 factor split into three bf16 pieces (24 significant bits; the six piece products per coordinate that
// matter, plus the owned point's constant in three pieces, fill 21 of the 32 K slots of ONE v_mfma_f32_16x16x32_bf16 per
// 16 x 16 block of pairs; every piece product is exact in the f32 accumulator and the streamed point's constant rides
// in the C operand).  The vector pipe is left with the exponential, the sums and the row pass' contraction.
// (f32-input MFMA was tried first and is not used: on gfx950 it executes on the vector ALUs - tools/mfma_overlap.hip,
// profiles/r2_mfma_valu_overlap_microbench.log - so it saves instructions but no time; the bf16 pipe is 2x faster
// per block and separate.)
//
// Work mapping.  A workgroup (4 waves) owns 4 x OWN tiles of 16 consecutive points of one cloud (512 points: one
// spatially compact patch, the clouds are sorted along a space-filling curve) and streams a segment of the other
// cloud through LDS in chunks of 256 points: every thread loads one float4 point, shifts it by the workgroup's
// origin, splits it into bf16 pieces and writes the A-operand slots plus f32 planes c x' y' z' |x'|^2; the waves read
// their MFMA operands straight out of LDS (ds_read_b128, conflict-free).  The next chunk's global loads are in flight
// while the current one is multiplied; one barrier per chunk.
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

#include <algorithm>
#include <map>
#include <mutex>
#include <vector>

#include "cpd_sweeps.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr double kLog2e = 1.4426950408889634;
constexpr int kBlock = prg::kSweepBlock;
constexpr int kOwn = prg::kMfmaOwn;          // tiles of 16 points a wave owns
constexpr int kWgPoints = 4 * 16 * kOwn;     // points a workgroup owns (512)
constexpr int kChunk = 256;                  // streamed points staged in LDS at a time (1 per thread; 2 x 21 KB of LDS)
constexpr int kChunkTiles = kChunk / 16;
// staged tile: 4 groups x 16 points x 8 bf16 (the A operand, 1 KB) followed by f32 planes c[16] x'[16] y'[16] z'[16] sq[16]
constexpr int kTileBytes = 1024 + 5 * 64;
constexpr int kTileFloats = kTileBytes / 4;

__device__ __forceinline__ float exp2r(float a) { return __builtin_amdgcn_exp2f(a); }
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

// bounding boxes: one per group of 32 points (zmeta / tmeta of the culled vector sweeps) and one per chunk of 256 points
struct alignas(32) BoxMeta { float lo[3]; float hi[3]; float aux; float pad; };

__device__ __forceinline__ float box_gap2(const float (&alo)[3], const float (&ahi)[3], const BoxMeta& g) {
    float d2 = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float gap = fmaxf(fmaxf(alo[k] - g.hi[k], g.lo[k] - ahi[k]), 0.f);
        d2 = fmaf(gap, gap, d2);
    }
    return d2;
}

// Box of the 512 points a workgroup owns = union of its 16 group boxes (groups that hold pads are left out: pads sit
// 1e18 away), and its centre as the workgroup's origin.  Every wave computes the same numbers.
__device__ __forceinline__ bool wg_box(const BoxMeta* __restrict__ gmeta, int64_t first_point, int64_t n_points, int lane,
                                       float (&lo)[3], float (&hi)[3]) {
    const int64_t g = first_point / 32 + (lane & 15);
    const bool valid = (g + 1) * 32 <= n_points;
    const BoxMeta m = gmeta[g];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        lo[k] = valid ? m.lo[k] : INFINITY;
        hi[k] = valid ? m.hi[k] : -INFINITY;
#pragma unroll
        for (int sh = 1; sh < 16; sh <<= 1) {
            lo[k] = fminf(lo[k], __shfl_xor(lo[k], sh, 64));
            hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], sh, 64));
        }
    }
    return hi[0] >= lo[0];
}

// Box of the 128 points a wave owns (its 4 groups), wave-uniform in SGPRs.  false: one of the groups holds pads.
struct WaveBox { float lo[3]; float hi[3]; };
__device__ __forceinline__ bool wave_box(const BoxMeta* __restrict__ gmeta, int64_t first_point, int64_t n_points, int lane,
                                         WaveBox& w) {
    const BoxMeta m = gmeta[first_point / 32 + (lane & 3)];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float l = m.lo[k], h = m.hi[k];
        l = fminf(l, __shfl_xor(l, 1, 64));
        l = fminf(l, __shfl_xor(l, 2, 64));
        h = fmaxf(h, __shfl_xor(h, 1, 64));
        h = fmaxf(h, __shfl_xor(h, 2, 64));
        w.lo[k] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(l)));
        w.hi[k] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(h)));
    }
    return first_point + 16 * kOwn <= n_points;
}

// 8 group bits (one per group of 32 streamed points of a chunk) -> 16 tile bits (two 16-point tiles per group)
__device__ __forceinline__ unsigned tiles_of_groups(unsigned g) {
    g = (g | (g << 4)) & 0x0F0Fu;
    g = (g | (g << 2)) & 0x3333u;
    g = (g | (g << 1)) & 0x5555u;
    return g | (g << 1);
}

// a = h + m + l with three bf16 pieces: 24 significant bits, every piece product exact in the f32 accumulator
struct Split3 { __bf16 h, m, l; };
__device__ __forceinline__ Split3 split3(float a) {
    Split3 s;
    s.h = (__bf16)a;
    const float r1 = a - (float)s.h;
    s.m = (__bf16)r1;
    s.l = (__bf16)(r1 - (float)s.m);
    return s;
}
// K slots of one coordinate: streamed side [h h m h l m 0 0] . owned side [h m h l h m 0 0] = every product of pieces
// down to 2^-24 of |a||b| (hh, hm, mh, hl, lh, mm)
__device__ __forceinline__ bf16x8 slots_streamed(const Split3 s) {
    const __bf16 z = (__bf16)0.f;
    return (bf16x8){s.h, s.h, s.m, s.h, s.l, s.m, z, z};
}
__device__ __forceinline__ bf16x8 slots_owned(const Split3 s) {
    const __bf16 z = (__bf16)0.f;
    return (bf16x8){s.h, s.m, s.h, s.l, s.h, s.m, z, z};
}

// one streamed point -> its A-operand slots (three coordinate groups + the group of ones that picks up the owned
// point's constant) and its f32 planes.  Pads (1e18 away) give c = -huge or -inf: every exponential of theirs is 0.
__device__ __forceinline__ void stage_point(float* __restrict__ buf, int p, const float4 v, const float4 o, float kk,
                                            bool lean = false) {
    const float dx = v.x - o.x, dy = v.y - o.y, dz = v.z - o.z;
    const float sq = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
    float* tile = buf + (p >> 4) * kTileFloats;
    bf16x8* a = reinterpret_cast<bf16x8*>(tile);
    const int j = p & 15;
    const __bf16 one = (__bf16)1.f, z = (__bf16)0.f;
    a[0 * 16 + j] = slots_streamed(split3(dx));
    a[1 * 16 + j] = slots_streamed(split3(dy));
    a[2 * 16 + j] = slots_streamed(split3(dz));
    a[3 * 16 + j] = (bf16x8){one, one, one, z, z, z, z, z};
    // f32 planes behind the operands: c[16], then (dx, dy)[16] and (dz, |d|^2)[16] as PAIRS - the row pass multiplies both
    // halves of a pair by the same P in one packed fma (lean row pass: (dz, 1) - sum P dz | sum P)
    float* f = tile + 256;
    f[j] = fmaf(kk, sq, v.w);
    *reinterpret_cast<float2*>(f + 16 + 2 * j) = make_float2(dx, dy);
    *reinterpret_cast<float2*>(f + 48 + 2 * j) = make_float2(dz, lean ? 1.f : sq);
}

// B operand of an owned point (lane l: slots of K group l / 16 for point l % 16): -2kk (x - o) per coordinate, and the
// point's constant `cst` split in three in the last group
__device__ __forceinline__ bf16x8 owned_operand(int k, float kk, float dx, float dy, float dz, float cst) {
    const float m2 = -2.f * kk;
    if (k == 0) return slots_owned(split3(m2 * dx));
    if (k == 1) return slots_owned(split3(m2 * dy));
    if (k == 2) return slots_owned(split3(m2 * dz));
    const Split3 c = split3(cst);
    const __bf16 z = (__bf16)0.f;
    return (bf16x8){c.h, c.m, c.l, z, z, z, z, z};
}

// ---- how a launch is cut into workgroups -------------------------------------------------------------------------
// grid mode (sk_g == 0): grid = (blocks of 512 owned points, segments of `chunks_per_seg` streamed chunks), one work item per
// workgroup, plane = segment.  More workgroups than the chip holds at a time (768: 3 per CU) are handed out as slots free up -
// what a culled sweep wants, whose items differ in length.  In the DENSE regime every (block, chunk) unit costs the same and
// the grid's granularity is pure loss: rounds x chunks_per_seg is 105 units per slot for the 99.8 there are at C1 (95 %), 14 for
// 12.5 on a 1/8 shard's row pass (89 %).  stream mode (sk_g > 0): exactly sk_g workgroups, all resident at once, each takes
// the contiguous run [w U / G, (w + 1) U / G) of the U = blocks x chunks units in (block, chunk) order - at most one unit more
// than its neighbour (99.7 % / 96 %) - as one work item per block the run touches.  An item of block b writes plane
// w - (the workgroup that holds b's first unit); the item that ends a block fills the block's unused planes with neutral values.
struct WorkItem { int blk; int c0; int nc; int plane; bool ends_block; };
// runs of units: the first `rem` workgroups take base + 1 units, the others base (U = G base + rem; host arithmetic, 32 bit)
struct StreamCut { int g; int pmax; unsigned base; unsigned rem; };
__device__ __forceinline__ unsigned sk_start(unsigned w, const StreamCut sc) { return w * sc.base + (w < sc.rem ? w : sc.rem); }
__device__ __forceinline__ WorkItem sk_item(unsigned u, unsigned u_end, unsigned w, unsigned nchunks, const StreamCut sc) {
    WorkItem it;
    const unsigned blk = u / nchunks, c0 = u - blk * nchunks;
    const unsigned left = nchunks - c0, want = u_end - u;
    it.blk = (int)blk;
    it.c0 = (int)c0;
    it.nc = (int)(want < left ? want : left);
    it.ends_block = c0 + (unsigned)it.nc == nchunks;
    const unsigned x = blk * nchunks, big = sc.rem * (sc.base + 1u);  // the block's first unit: which workgroup holds it?
    const unsigned wf = x < big ? x / (sc.base + 1u) : sc.rem + (x - big) / sc.base;
    it.plane = (int)(w - wf);
    return it;
}

// ---- sweep 1 on the matrix cores: den_n of cpd.py:80 --------------------------------------------------------------
// grid = (ceil(N / 512), S); plane blockIdx.y receives (min d^2 over the segment, sum of exp2(kk d^2 + L_n)) with
// L_n = prg::col_seed_offset - the same for every segment of a column.
//
// FUSED [r4]: the ONE sweep of a rigid EM iteration in the dense regime.  Everything the rigid M-step (cpd.py:160-192)
// consumes is a sum over the columns n of per-column sums over the sources m:
//   A_n = sum_m K_mn,   B_n = sum_m K_mn (z_m - o),   E_n = sum_m K_mn |z_m - o|^2        (K_mn = exp2(kk d^2 + L_n), o = the
//   block's origin) -  den_n = A_n 2^-L_n, pt1_n = den_n / (den_n + c), and with q_n = pt1_n / A_n:
//   S0 = sum pt1,  Sx = sum pt1 x,  sum_m p1_m z_m = sum_n q_n B_n + pt1_n o,  sum_m px_m z_m^T = sum_n x_n (...)^T,
//   sum_m p1_m |z_m|^2 = sum_n q_n (E_n + 2 o.B_n) + pt1_n |o|^2,  sum_n pt1_n |x_n|^2
// (k_colfinal_fused; z -> y through the transformation the E-step ran with) - no row pass, ONE exponential per pair and
// iteration instead of two.  The kernel is the column pass with the full row pass' contraction on the streamed (source) side:
// 5 channels (1, dx, dy, dz, |d|^2) per pair, the accumulators of k_rowpass_mfma<false>.  Like the lean row pass it carries no
// residual sums against the NEW transformation, so it is used only while sigma2's amplification mean |x|^2 / (sigma2 D) is
// within the fused factor (256; the decision kernel's `fused`) and the matrix-core column pass would run.  Output plane layout:
// fpart[plane][6][ncap] = (min d^2, A, Bx, By, Bz, E); grid mode only.
template <bool FUSED>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(3, FUSED ? 3 : 10))) void k_colpass_mfma(
                                                         const float4* __restrict__ tgt4, const float4* __restrict__ z4,
                                                         const BoxMeta* __restrict__ tmeta,
                                                         const BoxMeta* __restrict__ zmeta,
                                                         const BoxMeta* __restrict__ zchunk,
                                                         const float* __restrict__ colmin_prev,
                                                         const float* __restrict__ colmin_g,
                                                         const unsigned* __restrict__ motion, int chunks_per_seg,
                                                         int64_t m_total, int64_t n_total,
                                                         const double* __restrict__ params,
                                                         float2* __restrict__ colpart, int64_t ncap,
                                                         unsigned* __restrict__ wgcount, unsigned long long* __restrict__ work,
                                                         int first, int fine,
                                                         const EngineDecision* __restrict__ guard, const StreamCut sk,
                                                         float4* __restrict__ corig) {
    __shared__ __attribute__((aligned(16))) float stage[2][kChunkTiles * kTileFloats];
    __shared__ unsigned wave_tiles[kBlock / 64];
    if (guard) {  // launched ahead of the engine decision (cpd.hip, estep_impl): run only if it came out this way
        const bool fused_wanted = guard->col == 1 && guard->fused == 1;
        if (FUSED ? !fused_wanted : (guard->col != 1 || fused_wanted)) return;
        fine = guard->fine;
    }
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const float kk = (float)(-kLog2e / (2.0 * params[13]));
    const float mo = __uint_as_float(*motion);
    const int nchunks = (int)((m_total + kChunk - 1) / kChunk), nblocks = (int)((n_total + kWgPoints - 1) / kWgPoints);
    const int sk_g = FUSED ? 0 : sk.g, sk_pmax = sk.pmax;
    unsigned unit = sk_g ? sk_start(blockIdx.x, sk) : 0u;
    const unsigned unit_end = sk_g ? sk_start(blockIdx.x + 1u, sk) : 0u;
    for (;;) {  // work items of this workgroup (grid mode: exactly one)
    WorkItem item;
    if (sk_g) {
        item = sk_item(unit, unit_end, blockIdx.x, (unsigned)nchunks, sk);
    } else {
        item.blk = (int)blockIdx.x;
        item.c0 = (int)blockIdx.y * chunks_per_seg;
        item.nc = (item.c0 + chunks_per_seg < nchunks ? item.c0 + chunks_per_seg : nchunks) - item.c0;
        item.plane = blockIdx.y;
        item.ends_block = false;
    }
    const int64_t n0wg = (int64_t)item.blk * kWgPoints, n0 = n0wg + wv * (16 * kOwn);
    // the workgroup's patch: box (for the chunk test) and origin (its centre).  A workgroup that holds pads has no usable
    // box: it evaluates everything and takes its first point as origin.
    float lo[3], hi[3];
    float4 o;
    float thr = INFINITY;    // skip a chunk when its box is farther than thr (squared) from the patch
    float thr_w = INFINITY;  // ... and a group of 32 streamed points when it is farther than thr_w from this wave's 128
    WaveBox wb;
    bool wave_cull = wave_box(tmeta, n0, n_total, lane, wb) && !first && fine;
    if (n0wg + kWgPoints <= n_total && wg_box(tmeta, n0wg, n_total, lane, lo, hi)) {
        o = make_float4(0.5f * (lo[0] + hi[0]), 0.5f * (lo[1] + hi[1]), 0.5f * (lo[2] + hi[2]), 0.f);
        if (!first) {  // (first E-step of a registration: no minima to bound anything with)
            // as in k_colpass_cull: (sqrt(largest column minimum of the previous E-step) + source motion)^2 bounds this
            // E-step's minima from above; a chunk beyond that by 127 / |kk| adds < 2^-127 of any column's largest term
            float cm = colmin_g[n0wg / 32 + (lane & 15)];
            float cw = __shfl(cm, 4 * wv + (lane & 3), 64);  // the wave's own four groups
            cw = fmaxf(cw, __shfl_xor(cw, 1, 64));
            cw = fmaxf(cw, __shfl_xor(cw, 2, 64));
#pragma unroll
            for (int sh = 1; sh < 16; sh <<= 1) cm = fmaxf(cm, __shfl_xor(cm, sh, 64));
            const float r = sqrtf(cm) + mo, rw = sqrtf(cw) + mo;
            thr = r * r * 1.00001f + (-prg::kCullExp) / kk;
            thr_w = rw * rw * 1.00001f + (-prg::kCullExp) / kk;
        }
    } else {
        o = tgt4[n0wg];
        o.w = 0.f;
        wave_cull = false;
#pragma unroll
        for (int q = 0; q < 3; ++q) { lo[q] = -INFINITY; hi[q] = INFINITY; }
    }
    const int k = lane >> 4, j = lane & 15;
    bf16x8 bx[kOwn];
    float s[kOwn], tm[kOwn], off[FUSED ? 1 : kOwn];
    f32x2 uxy[FUSED ? kOwn : 1], uze[FUSED ? kOwn : 1];  // FUSED: (sum K dx, sum K dy), (sum K dz, sum K |d|^2); s = sum K
#pragma unroll
    for (int t = 0; t < kOwn; ++t) {
        const float4 x = tgt4[n0 + 16 * t + j];
        const float xx = x.x - o.x, xy = x.y - o.y, xz = x.z - o.z;
        const float xsq = fmaf(xz, xz, fmaf(xy, xy, xx * xx));
        const float offt = first ? 0.f : prg::col_seed_offset(kk, colmin_prev[n0 + 16 * t + j], mo);
        if (!FUSED) off[t] = offt;  // (FUSED: recomputed in the epilogue - eight registers the contraction needs)
        bx[t] = owned_operand(k, kk, xx, xy, xz, fmaf(kk, xsq, offt));
        s[t] = 0.f;
        tm[t] = -INFINITY;
        if (FUSED) uxy[t] = uze[t] = (f32x2){0.f, 0.f};
    }
    unsigned tiles_done = 0;
    const int64_t c0 = item.c0;
    const int nc = item.nc;  // <= 64: one ballot's worth of chunks (grid mode: chunks_per_seg; stream mode: the run length)
    // which chunks of the segment are needed: one box test per lane, one ballot (identical in all four waves)
    const BoxMeta cm = zchunk[c0 + (lane < nc ? lane : nc - 1)];
    unsigned long long mask = __ballot(lane < nc && !(box_gap2(lo, hi, cm) > thr));
    if (mask) {
        int cur = __builtin_ctzll(mask);
        mask &= mask - 1;
        float4 ra = z4[(c0 + cur) * kChunk + threadIdx.x];
        ra.w = 0.f;  // (weighted sources do not take this path)
        BoxMeta gm = zmeta[wave_cull ? (c0 + cur) * 8 + (lane & 7) : 0];  // the chunk's eight groups, for this wave's finer test
        stage_point(stage[0], threadIdx.x, ra, o, kk);
        __syncthreads();
        for (int bsel = 0;; bsel ^= 1) {
            const float* __restrict__ buf = stage[bsel];
            // groups of this chunk the wave's own 128 points can see -> tiles to evaluate (wave-uniform)
            unsigned tmask = 0xFFFFu;
            if (wave_cull)
                tmask = tiles_of_groups((unsigned)__ballot(lane < 8 && !(box_gap2(wb.lo, wb.hi, gm) > thr_w)) & 0xFFu);
            const int nxt = mask ? __builtin_ctzll(mask) : -1;
            mask &= mask - 1;
            if (nxt >= 0) {
                ra = z4[(c0 + nxt) * kChunk + threadIdx.x];
                ra.w = 0.f;
                if (wave_cull) gm = zmeta[(c0 + nxt) * 8 + (lane & 7)];
            }
            tiles_done += __popc(tmask);
            // software pipeline: the MFMA of the NEXT (tile, column tile) pair is issued before the current pair's
            // accumulator is exponentiated.  tile_step works on accumulator d of tile (a1, cz) and leaves the first
            // accumulator of tile (a1n, czn) in d.
            auto tile_step = [&](f32x4& d, const float* __restrict__ tb, const bf16x8 a1, const f32x4 cz, const bf16x8 a1n,
                                 const f32x4 czn) {
                // FUSED: this lane's four streamed sources 4k .. 4k+3 of the tile: (dx, dy) and (dz, |d|^2) pairs (stage_point)
                f32x2 xy0, xy1, xy2, xy3, zs0, zs1, zs2, zs3;
                if (FUSED) {
                    const f32x4 a01 = *reinterpret_cast<const f32x4*>(tb + 272 + 8 * k), a23 = *reinterpret_cast<const f32x4*>(tb + 276 + 8 * k),
                                b01 = *reinterpret_cast<const f32x4*>(tb + 304 + 8 * k), b23 = *reinterpret_cast<const f32x4*>(tb + 308 + 8 * k);
                    xy0 = (f32x2){a01[0], a01[1]}; xy1 = (f32x2){a01[2], a01[3]}; xy2 = (f32x2){a23[0], a23[1]}; xy3 = (f32x2){a23[2], a23[3]};
                    zs0 = (f32x2){b01[0], b01[1]}; zs1 = (f32x2){b01[2], b01[3]}; zs2 = (f32x2){b23[0], b23[1]}; zs3 = (f32x2){b23[2], b23[3]};
                }
#pragma unroll
                for (int u = 0; u < kOwn; ++u) {
                    const f32x4 dn = u + 1 < kOwn ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, bx[u + 1], cz, 0, 0, 0)
                                                  : __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1n, bx[0], czn, 0, 0, 0);
                    tm[u] = fmaxf(fmaxf(tm[u], d[0]), d[1]);
                    tm[u] = fmaxf(fmaxf(tm[u], d[2]), d[3]);
                    const float q0 = exp2r(d[0]), q1 = exp2r(d[1]), q2 = exp2r(d[2]), q3 = exp2r(d[3]);
                    s[u] += (q0 + q1) + (q2 + q3);
                    if (FUSED) {
                        const f32x2 s0 = {q0, q0}, s1 = {q1, q1}, s2 = {q2, q2}, s3 = {q3, q3};
                        uxy[u] = __builtin_elementwise_fma(s3, xy3, __builtin_elementwise_fma(s2, xy2, __builtin_elementwise_fma(s1, xy1, __builtin_elementwise_fma(s0, xy0, uxy[u]))));
                        uze[u] = __builtin_elementwise_fma(s3, zs3, __builtin_elementwise_fma(s2, zs2, __builtin_elementwise_fma(s1, zs1, __builtin_elementwise_fma(s0, zs0, uze[u]))));
                    }
                    d = dn;
                }
            };
            if (tmask == 0xFFFFu) {  // every tile (the dense regime): fixed trip count
                bf16x8 a1 = reinterpret_cast<const bf16x8*>(buf)[lane];
                f32x4 cz = *reinterpret_cast<const f32x4*>(buf + 256 + 4 * k);
                f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, bx[0], cz, 0, 0, 0);
                for (int t = 0; t < kChunkTiles; ++t) {
                    const float* __restrict__ tn = buf + (t + 1 < kChunkTiles ? t + 1 : t) * kTileFloats;
                    const bf16x8 a1n = reinterpret_cast<const bf16x8*>(tn)[lane];
                    const f32x4 czn = *reinterpret_cast<const f32x4*>(tn + 256 + 4 * k);
                    tile_step(d, buf + t * kTileFloats, a1, cz, a1n, czn);
                    a1 = a1n;
                    cz = czn;
                }
            } else if (tmask) {  // the tiles this wave can see
                int t = __builtin_ctz(tmask);
                tmask &= tmask - 1;
                bf16x8 a1 = reinterpret_cast<const bf16x8*>(buf + t * kTileFloats)[lane];
                f32x4 cz = *reinterpret_cast<const f32x4*>(buf + t * kTileFloats + 256 + 4 * k);
                f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, bx[0], cz, 0, 0, 0);
                for (;;) {
                    const bool more = tmask != 0;
                    const int t2 = more ? __builtin_ctz(tmask) : t;
                    tmask &= tmask - 1;
                    const float* __restrict__ tn = buf + t2 * kTileFloats;
                    const bf16x8 a1n = reinterpret_cast<const bf16x8*>(tn)[lane];
                    const f32x4 czn = *reinterpret_cast<const f32x4*>(tn + 256 + 4 * k);
                    tile_step(d, buf + t * kTileFloats, a1, cz, a1n, czn);
                    if (!more) break;
                    t = t2;
                    a1 = a1n;
                    cz = czn;
                }
            }
            if (nxt >= 0) stage_point(stage[bsel ^ 1], threadIdx.x, ra, o, kk);
            __syncthreads();
            if (nxt < 0) break;
        }
    }
    if (lane == 0) wave_tiles[wv] = tiles_done;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned tiles = wave_tiles[0] + wave_tiles[1] + wave_tiles[2] + wave_tiles[3];
        wgcount[(int64_t)item.plane * nblocks + item.blk] = tiles;
        if (tiles) atomicAdd(work, (unsigned long long)tiles);  // what the next E-step's engine decision goes by
    }
    if (FUSED) {
        float* __restrict__ fout = reinterpret_cast<float*>(colpart) + (int64_t)item.plane * 6 * ncap + n0;
#pragma unroll
        for (int u2 = 0; u2 < kOwn; ++u2) {
            const float st = xor_sum(s[u2]), tt = xor_max(tm[u2]);
            const float bxs = xor_sum(uxy[u2][0]), bys = xor_sum(uxy[u2][1]), bzs = xor_sum(uze[u2][0]), es = xor_sum(uze[u2][1]);
            if (lane < 16) {
                const float offt = first ? 0.f : prg::col_seed_offset(kk, colmin_prev[n0 + 16 * u2 + lane], mo);
                float dmin = fmaxf((tt - offt) / kk, 0.f);
                dmin = fmaf(dmin, 2.0e-4f, dmin) + 1.0e-12f;
                fout[16 * u2 + lane] = tt == -INFINITY ? INFINITY : dmin;
                fout[ncap + 16 * u2 + lane] = st;
                fout[2 * ncap + 16 * u2 + lane] = bxs;
                fout[3 * ncap + 16 * u2 + lane] = bys;
                fout[4 * ncap + 16 * u2 + lane] = bzs;
                fout[5 * ncap + 16 * u2 + lane] = es;
            }
        }
        if (item.plane == 0 && threadIdx.x == 0) corig[item.blk] = o;
        break;
    }
    float2* __restrict__ out = colpart + (int64_t)item.plane * ncap + n0;
#pragma unroll
    for (int u2 = 0; u2 < kOwn; ++u2) {
        const float st = xor_sum(s[u2]);
        const float tt = xor_max(tm[u2]);
        if (lane < 16) {
            // t = kk d^2 + L  ->  d^2 = (t - L) / kk; kept slightly high: the next E-step's cull bound wants an upper bound
            float dmin = fmaxf((tt - off[u2]) / kk, 0.f);
            dmin = fmaf(dmin, 2.0e-4f, dmin) + 1.0e-12f;
            out[16 * u2 + lane] = make_float2(tt == -INFINITY ? INFINITY : dmin, st);
        }
    }
    if (!sk_g) break;
    if (item.ends_block) {  // the planes this block did not need: neutral partials, no evaluated tiles
        for (int q = item.plane + 1; q < sk_pmax; ++q) {
            colpart[(int64_t)q * ncap + n0wg + threadIdx.x] = make_float2(INFINITY, 0.f);
            colpart[(int64_t)q * ncap + n0wg + kBlock + threadIdx.x] = make_float2(INFINITY, 0.f);
            if (threadIdx.x == 0) wgcount[(int64_t)q * nblocks + item.blk] = 0u;
        }
    }
    unit += item.nc;
    if (unit >= unit_end) break;
    __syncthreads();  // (wave_tiles and the staging buffers are reused by the next item)
    }  // work items
}

// ---- sweep 2 on the matrix cores: p1, px and the sigma2 residual of cpd.py:84-87 ----------------------------------
// Per (row tile, target tile): D = mfma(streamed slots, owned slots, kk|x'|^2 + b); lane l, reg r holds the pair
// (n = 4 (l/16) + r, m = l % 16); P = exp2(D); the lane adds P, P x'_n, P |x'_n|^2 into its row's partial sums (the
// four lanes of a row are added up at the end).  Output plane blockIdx.y, relative to the workgroup's origin o (stored
// in rorig[row block of 512]): p1, u' = sum P (x - o), e' = sum P |x - o|^2 - k_row_moments' residual form with o as
// the reference point.
// STREAM: the launch is cut in stream mode (see WorkItem).  A template parameter, not a run-time one: with the item loop in
// the code the register allocator needs 170 VGPRs where the single-item kernel takes 152 - one allocation granule above three
// waves per SIMD - so the stream instantiation is held to three waves per SIMD explicitly (two values computed before the item
// loop live in scratch, outside the chunk loop) and the grid-mode instantiations compile exactly as before.  Only the LEAN row
// pass has a stream instantiation: the dense regime, where stream mode applies, is where the row pass runs lean.
template <bool LEAN, bool STREAM>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(3, STREAM ? 3 : 10))) void k_rowpass_mfma(const float4* __restrict__ z4, const float4* __restrict__ tgt4,
                                                         const BoxMeta* __restrict__ zmeta,
                                                         const BoxMeta* __restrict__ tmeta,
                                                         const BoxMeta* __restrict__ tchunk, int chunks_per_seg,
                                                         int64_t n_total, int64_t m_total,
                                                         const double* __restrict__ params, float* __restrict__ rowpart,
                                                         int64_t mcap, float4* __restrict__ rorig,
                                                         unsigned char* __restrict__ rowflag,
                                                         unsigned* __restrict__ wgcount, unsigned long long* __restrict__ work,
                                                         int fine, const StreamCut sk) {
    __shared__ __attribute__((aligned(16))) float stage[2][kChunkTiles * kTileFloats];
    __shared__ unsigned wave_tiles[kBlock / 64];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const float kk = (float)(-kLog2e / (2.0 * params[13]));
    const int nchunks = (int)((n_total + kChunk - 1) / kChunk), nblocks = (int)((m_total + kWgPoints - 1) / kWgPoints);
    const int sk_pmax = sk.pmax;
    unsigned unit = STREAM ? sk_start(blockIdx.x, sk) : 0u;
    const unsigned unit_end = STREAM ? sk_start(blockIdx.x + 1u, sk) : 0u;
    for (;;) {  // work items of this workgroup (grid mode: exactly one) - see WorkItem
    WorkItem item;
    if (STREAM) {
        item = sk_item(unit, unit_end, blockIdx.x, (unsigned)nchunks, sk);
    } else {
        item.blk = (int)blockIdx.x;
        item.c0 = (int)blockIdx.y * chunks_per_seg;
        item.nc = (item.c0 + chunks_per_seg < nchunks ? item.c0 + chunks_per_seg : nchunks) - item.c0;
        item.plane = blockIdx.y;
        item.ends_block = false;
    }
    const int64_t m0wg = (int64_t)item.blk * kWgPoints, m0 = m0wg + wv * (16 * kOwn);
    float lo[3], hi[3];
    float4 o;
    WaveBox wb;
    bool wave_cull = wave_box(zmeta, m0, m_total, lane, wb) && fine;
    if (m0wg + kWgPoints <= m_total && wg_box(zmeta, m0wg, m_total, lane, lo, hi)) {
        o = make_float4(0.5f * (lo[0] + hi[0]), 0.5f * (lo[1] + hi[1]), 0.5f * (lo[2] + hi[2]), 0.f);
    } else {  // a workgroup that holds pads: no usable box, nothing is skipped
        o = z4[m0wg];
        o.w = 0.f;
        wave_cull = false;
#pragma unroll
        for (int q = 0; q < 3; ++q) { lo[q] = -INFINITY; hi[q] = INFINITY; }
    }
    const int k = lane >> 4, j = lane & 15;
    bf16x8 bz[kOwn];
    // per owned tile, over this lane's targets, in packed pairs: (sum P dx, sum P dy), (sum P dz, sum P |d|^2) and the
    // halves of sum P.  LEAN: (sum P dx, sum P dy), (sum P dz, sum P) - the residual sums sum P |d|^2 are not carried: all
    // the M-step wants from them is sum_n pt1_n |x_n|^2, which the column side has (k_colfinal -> k_xpx_columns), and while
    // sigma2 is large that sum does not have to match the row sums to the last bit (the decision kernel says when).
    f32x2 uxy[kOwn], uze[kOwn], pp[LEAN ? 1 : kOwn];
#pragma unroll
    for (int t = 0; t < kOwn; ++t) {
        const float4 z = z4[m0 + 16 * t + j];
        const float zx = z.x - o.x, zy = z.y - o.y, zz = z.z - o.z;
        bz[t] = owned_operand(k, kk, zx, zy, zz, kk * fmaf(zz, zz, fmaf(zy, zy, zx * zx)));
        uxy[t] = uze[t] = (f32x2){0.f, 0.f};
        if (!LEAN) pp[t] = (f32x2){0.f, 0.f};
    }
    unsigned tiles_done = 0;
    const int64_t c0 = item.c0;
    const int nc = item.nc;  // <= 64: one ballot's worth of chunks (grid mode: chunks_per_seg; stream mode: the run length)
    // a chunk is skipped when every P of the (patch, chunk) block is below the cull bound: kk dist^2(boxes) + max b_n < -kCullExp
    const BoxMeta cm = tchunk[c0 + (lane < nc ? lane : nc - 1)];
    unsigned long long mask = __ballot(lane < nc && !(fmaf(box_gap2(lo, hi, cm), kk, cm.aux) < -prg::kCullExp));
    if (mask) {
        int cur = __builtin_ctzll(mask);
        mask &= mask - 1;
        float4 ra = tgt4[(c0 + cur) * kChunk + threadIdx.x];
        BoxMeta gm = tmeta[wave_cull ? (c0 + cur) * 8 + (lane & 7) : 0];  // the chunk's eight groups (box, largest b_n)
        stage_point(stage[0], threadIdx.x, ra, o, kk, LEAN);
        __syncthreads();
        for (int bsel = 0;; bsel ^= 1) {
            const float* __restrict__ buf = stage[bsel];
            // the same test for this wave's own 128 rows against each group of 32 targets -> tiles to evaluate
            unsigned tmask = 0xFFFFu;
            if (wave_cull)
                tmask = tiles_of_groups(
                    (unsigned)__ballot(lane < 8 && !(fmaf(box_gap2(wb.lo, wb.hi, gm), kk, gm.aux) < -prg::kCullExp)) & 0xFFu);
            const int nxt = mask ? __builtin_ctzll(mask) : -1;
            mask &= mask - 1;
            if (nxt >= 0) {
                ra = tgt4[(c0 + nxt) * kChunk + threadIdx.x];
                if (wave_cull) gm = tmeta[(c0 + nxt) * 8 + (lane & 7)];
            }
            tiles_done += __popc(tmask);
            // tile_step: accumulator d of tile tb (operands a1, cx) -> sums; leaves the first accumulator of the next tile
            auto tile_step = [&](f32x4& d, const float* __restrict__ tb, const bf16x8 a1, const f32x4 cx, const bf16x8 a1n,
                                 const f32x4 cxn) {
                // this lane's four targets 4k .. 4k+3: (dx, dy) and (dz, 1) pairs
                const f32x4 a01 = *reinterpret_cast<const f32x4*>(tb + 272 + 8 * k), a23 = *reinterpret_cast<const f32x4*>(tb + 276 + 8 * k),
                            b01 = *reinterpret_cast<const f32x4*>(tb + 304 + 8 * k), b23 = *reinterpret_cast<const f32x4*>(tb + 308 + 8 * k);
                const f32x2 xy0 = {a01[0], a01[1]}, xy1 = {a01[2], a01[3]}, xy2 = {a23[0], a23[1]}, xy3 = {a23[2], a23[3]};
                const f32x2 zs0 = {b01[0], b01[1]}, zs1 = {b01[2], b01[3]}, zs2 = {b23[0], b23[1]}, zs3 = {b23[2], b23[3]};
#pragma unroll
                for (int u = 0; u < kOwn; ++u) {
                    const f32x4 dn = u + 1 < kOwn ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, bz[u + 1], cx, 0, 0, 0)
                                                  : __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1n, bz[0], cxn, 0, 0, 0);
                    const float q0 = exp2r(d[0]), q1 = exp2r(d[1]), q2 = exp2r(d[2]), q3 = exp2r(d[3]);
                    // the contraction in packed fp32: 8 v_pk_fma_f32 (+ 2 v_pk_add_f32) where 16 fma + 4 add stood
                    const f32x2 s0 = {q0, q0}, s1 = {q1, q1}, s2 = {q2, q2}, s3 = {q3, q3};
                    if (!LEAN) {
                        pp[u] += (f32x2){q0, q1};
                        pp[u] += (f32x2){q2, q3};
                    }
                    uxy[u] = __builtin_elementwise_fma(s3, xy3, __builtin_elementwise_fma(s2, xy2, __builtin_elementwise_fma(s1, xy1, __builtin_elementwise_fma(s0, xy0, uxy[u]))));
                    uze[u] = __builtin_elementwise_fma(s3, zs3, __builtin_elementwise_fma(s2, zs2, __builtin_elementwise_fma(s1, zs1, __builtin_elementwise_fma(s0, zs0, uze[u]))));
                    d = dn;
                }
            };
            if (tmask == 0xFFFFu) {  // every tile (the dense regime): fixed trip count
                bf16x8 a1 = reinterpret_cast<const bf16x8*>(buf)[lane];
                f32x4 cx = *reinterpret_cast<const f32x4*>(buf + 256 + 4 * k);
                f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, bz[0], cx, 0, 0, 0);
                for (int t = 0; t < kChunkTiles; ++t) {
                    const float* __restrict__ tn = buf + (t + 1 < kChunkTiles ? t + 1 : t) * kTileFloats;
                    const bf16x8 a1n = reinterpret_cast<const bf16x8*>(tn)[lane];
                    const f32x4 cxn = *reinterpret_cast<const f32x4*>(tn + 256 + 4 * k);
                    tile_step(d, buf + t * kTileFloats, a1, cx, a1n, cxn);
                    a1 = a1n;
                    cx = cxn;
                }
            } else if (tmask) {  // the tiles this wave can see
                int t = __builtin_ctz(tmask);
                tmask &= tmask - 1;
                bf16x8 a1 = reinterpret_cast<const bf16x8*>(buf + t * kTileFloats)[lane];
                f32x4 cx = *reinterpret_cast<const f32x4*>(buf + t * kTileFloats + 256 + 4 * k);
                f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, bz[0], cx, 0, 0, 0);
                for (;;) {
                    const bool more = tmask != 0;
                    const int t2 = more ? __builtin_ctz(tmask) : t;
                    tmask &= tmask - 1;
                    const float* __restrict__ tn = buf + t2 * kTileFloats;
                    const bf16x8 a1n = reinterpret_cast<const bf16x8*>(tn)[lane];
                    const f32x4 cxn = *reinterpret_cast<const f32x4*>(tn + 256 + 4 * k);
                    tile_step(d, buf + t * kTileFloats, a1, cx, a1n, cxn);
                    if (!more) break;
                    t = t2;
                    a1 = a1n;
                    cx = cxn;
                }
            }
            if (nxt >= 0) stage_point(stage[bsel ^ 1], threadIdx.x, ra, o, kk, LEAN);
            __syncthreads();
            if (nxt < 0) break;
        }
    }
    if (lane == 0) wave_tiles[wv] = tiles_done;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned tiles = wave_tiles[0] + wave_tiles[1] + wave_tiles[2] + wave_tiles[3];
        wgcount[(int64_t)item.plane * nblocks + item.blk] = tiles;
        if (tiles) atomicAdd(work, (unsigned long long)tiles);  // what the next E-step's engine decision goes by
    }
    // k_row_moments skips (128-row block, plane) partials that were never touched: neither written nor read
    const bool touched = tiles_done != 0;  // (wave-uniform: this wave's 128 rows)
    if (lane == 0) rowflag[((int64_t)item.blk * 4 + wv) * 64 + item.plane] = touched ? 1 : 0;
    if (touched) {
        float* __restrict__ out = rowpart + (int64_t)item.plane * 5 * mcap + m0;
#pragma unroll
        for (int u = 0; u < kOwn; ++u) {
            const float a0 = xor_sum(LEAN ? uze[u][1] : pp[LEAN ? 0 : u][0] + pp[LEAN ? 0 : u][1]), a1s = xor_sum(uxy[u][0]),
                        a2 = xor_sum(uxy[u][1]), a3 = xor_sum(uze[u][0]);
            if (!LEAN) {  // (LEAN: plane 4, the residual sums, stays unwritten and unread)
                const float a4 = xor_sum(uze[u][1]);
                if (lane < 16) out[4 * mcap + 16 * u + lane] = a4;
            }
            if (lane < 16) {
                out[16 * u + lane] = a0;
                out[mcap + 16 * u + lane] = a1s;
                out[2 * mcap + 16 * u + lane] = a2;
                out[3 * mcap + 16 * u + lane] = a3;
            }
        }
    }
    if (item.c0 == 0 && threadIdx.x == 0) rorig[item.blk] = o;
    if (!STREAM) break;
    if (item.ends_block) {  // the planes this block did not need: untouched, no evaluated tiles
        for (int q = item.plane + 1 + lane; q < sk_pmax; q += 64) rowflag[((int64_t)item.blk * 4 + wv) * 64 + q] = 0;
        if (threadIdx.x == 0)
            for (int q = item.plane + 1; q < sk_pmax; ++q) wgcount[(int64_t)q * nblocks + item.blk] = 0u;
    }
    unit += item.nc;
    if (unit >= unit_end) break;
    __syncthreads();  // (wave_tiles and the staging buffers are reused by the next item)
    }  // work items
}

// one box per chunk of 256 points = union of its 8 group boxes (+ the largest aux of the groups)
__global__ __launch_bounds__(kBlock) void k_chunk_meta(const BoxMeta* __restrict__ gmeta, int64_t nchunk,
                                                       BoxMeta* __restrict__ cmeta) {
    const int64_t c = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (c >= nchunk) return;
    BoxMeta o = gmeta[c * 8];
#pragma unroll
    for (int g = 1; g < 8; ++g) {
        const BoxMeta m = gmeta[c * 8 + g];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            o.lo[k] = fminf(o.lo[k], m.lo[k]);
            o.hi[k] = fmaxf(o.hi[k], m.hi[k]);
        }
        o.aux = fmaxf(o.aux, m.aux);
    }
    cmeta[c] = o;
}

// the same for the transformed source, plus the bounding box of its m real points -> stat[8..13] (lo.xyz, hi.xyz as float
// bits): one workgroup.  With `eng.dev` set, thread 0 then takes the E-step's engine decision from what
// the device holds at this point - sigma2, the motion of this E-step's transform (stat[slot]), the previous E-step's largest
// column minimum (stat[4 + (slot ^ 1)]), the box just computed - and publishes it in device memory (guard of the column
// pass launched ahead) and in the host's mailbox.
__global__ __launch_bounds__(kBlock) void k_chunk_meta_bbox(const BoxMeta* __restrict__ gmeta, int64_t nchunk,
                                                            BoxMeta* __restrict__ cmeta, const float4* __restrict__ pts,
                                                            int64_t m, unsigned* __restrict__ stat,
                                                            const double* __restrict__ params, const EngineArgs eng) {
    __shared__ float sh[kBlock / 64][6];
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int64_t c = threadIdx.x; c < nchunk; c += kBlock) {
        BoxMeta o = gmeta[c * 8];
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const BoxMeta q = gmeta[c * 8 + g];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                o.lo[k] = fminf(o.lo[k], q.lo[k]);
                o.hi[k] = fmaxf(o.hi[k], q.hi[k]);
            }
            o.aux = fmaxf(o.aux, q.aux);
            if ((c * 8 + g + 1) * 32 <= m) {  // a group of real points only
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    lo[k] = fminf(lo[k], q.lo[k]);
                    hi[k] = fmaxf(hi[k], q.hi[k]);
                }
            }
        }
        cmeta[c] = o;
    }
    const int64_t tail = (m / 32) * 32 + threadIdx.x;  // the real points of the last, partly padded group
    if (threadIdx.x < 32 && tail < m) {
        const float4 p = pts[tail];
        lo[0] = fminf(lo[0], p.x); hi[0] = fmaxf(hi[0], p.x);
        lo[1] = fminf(lo[1], p.y); hi[1] = fmaxf(hi[1], p.y);
        lo[2] = fminf(lo[2], p.z); hi[2] = fmaxf(hi[2], p.z);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int sft = 1; sft <= 32; sft <<= 1) {
            lo[k] = fminf(lo[k], __shfl_xor(lo[k], sft, 64));
            hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], sft, 64));
        }
    }
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            sh[threadIdx.x >> 6][k] = lo[k];
            sh[threadIdx.x >> 6][3 + k] = hi[k];
        }
    __syncthreads();
    if (threadIdx.x != 0) return;
    float box[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        float v = sh[0][q];
        for (int w = 1; w < kBlock / 64; ++w) v = q < 3 ? fminf(v, sh[w][q]) : fmaxf(v, sh[w][q]);
        box[q] = v;
        stat[8 + q] = __float_as_uint(v);
    }
    if (!eng.dev) return;
    // ---- the engine decision (the host's former read-back-and-decide, verbatim) ----
    const double sigma2 = params[13], nk = kLog2e / (2.0 * sigma2);
    const double mo = __uint_as_float(stat[eng.slot]), cmax = __uint_as_float(stat[4 + (eng.slot ^ 1)]), r = sqrt(cmax);
    const double width = r >= mo ? 4.0 * r * mo : (r + mo) * (r + mo);  // of the bracket of a column minimum
    const bool ok = sigma2 > 0.0 && isfinite(sigma2);
    // Which engine?  MEASURED, not predicted from sigma2: the matrix-core sweeps of the previous E-step counted the
    // (128 x 16) tiles they evaluated; per owned point that is how much of the other cloud still lies within reach.  The
    // host turns what is known about the two engines into a bound on that number (estep_impl); here the count meets it.
    // sigma2 only shrinks along a registration, so leaving is for good: the row pass first, the column pass - and with
    // it the dense regime - later.
    const EngineDecision prev = *eng.dev;
    float r_col = eng.reset ? (float)eng.streamed_col : prev.r_col, r_row = eng.reset ? (float)eng.streamed_row : prev.r_row;
    int row_off = eng.reset ? 0 : prev.row_off;
    const unsigned long long tc = eng.work[0], tr = eng.work[1];
    eng.work[0] = 0ull;
    eng.work[1] = 0ull;
    if (tc && !eng.reset) r_col = (float)((double)tc * 2048.0 / eng.owned_col);
    if (tr && !eng.reset) r_row = (float)((double)tr * 2048.0 / eng.owned_row);

    // (the row pass' bound depends on which row pass it would be: lean - cheaper per chunk, competitive for longer - or full)
    const bool lean_ok = eng.tsum && eng.tsum[3] > 0.0 && sigma2 * (double)eng.dim * eng.lean_factor * eng.owned_col >= eng.tsum[3];
    // after a fused sweep there is no row pass to count: the pairs per source point follow from the column side's count
    if (prev.fused && tc && !eng.reset) r_row = (float)((double)r_col * eng.owned_col / eng.owned_row);
    // the fused sweep (one exponential per pair instead of two) stays ahead of any pairing of the two-sweep engines for as long
    // as it may run at all, so the row pass' bound is not asked while it can
    // (the fused sweep's moments all come from the same column sums - consistent where the lean pass has to reconcile row sums
    // with column sums - and stay within 3e-6 of the oracle's sigma2 up to an amplification of 1300, tools/fused_error.py: it
    // has a factor of its own, far above the lean pass')
    const bool fused_ok = eng.tsum && eng.tsum[3] > 0.0 && sigma2 * (double)eng.dim * eng.fused_factor * eng.owned_col >= eng.tsum[3];
    const bool fused_try = eng.fused_allowed && fused_ok && !row_off;
    bool dense = ok && (eng.forced || r_col >= (fused_try ? eng.r_col_bound_fused : eng.r_col_bound));
    // [r6] A caller that wants moments only, past the fused sweep's amplification limit (sigma2 small against the cloud's extent:
    // two clusters far apart, a 10:1:1 box): what competes is TWO matrix-core sweeps against ONE residual-form sweep on the vector
    // pipe, which is exact at any amplification.  Measured from identical states (tools/single_sweep_ab.py, 100k points): clusters
    // 1.59 against 1.42 ms with 44 % of the pairs needed, the box 0.95 against 0.61 ms with 18 %; fully dense it is 2.6 against
    // 3.1 ms - the matrix cores keep such an E-step only while at least 60 % of the pairs are needed.
    if (dense && !eng.forced && eng.fused_allowed && eng.resid_allowed && !fused_try && !(r_col >= 0.6 * eng.streamed_col)) dense = false;
    if (!fused_try && !(r_row >= (lean_ok ? eng.r_row_bound : eng.r_row_bound_full))) row_off = 1;
    // the column pass needs exponent offsets before it sees the data: from the previous E-step's column minima, or - first
    // E-step, no minima yet - none at all when the farthest target / source pair is still above the flush threshold
    // (farthest corners of the two bounding boxes: every term of every column is >= 2^-110)
    double far2 = 0.0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double a = fabs((double)eng.tbox[3 + k] - (double)box[k]), b = fabs((double)box[3 + k] - (double)eng.tbox[k]);
        far2 += fmax(a, b) * fmax(a, b);
    }
    const bool first = dense && !eng.have_colmin && isfinite(far2) && nk * far2 < 110.0;
    const bool col = first || (dense && eng.have_colmin && isfinite(cmax) && nk * width < 150.0);
    // the culled vector-pipe row pass overtakes the matrix-core one earlier than the column pass does
    const bool row = dense && (eng.forced || !row_off);
    // The matrix-core row pass may drop its residual sums (k_rowpass_mfma<LEAN>) while the M-step's
    // sigma2 = (sum pt1 |x|^2 - ...) / (Np D) does not cancel much: the column-side sum differs from what the row sums imply
    // by ~1e-6 relative (fp32 tails), amplified by mean |x|^2 / (sigma2 D) - at most eng.lean_factor here.
    const bool lean = row && lean_ok;
    EngineDecision d;
    d.seq = eng.seq;
    d.col = col ? 1 : 0;
    d.first = first ? 1 : 0;
    d.row = row ? 1 : 0;
    d.fine = nk * eng.ext2 > 200.0 * (prg::kCullExp / 127.0) ? 1 : 0;  // below, every group of every chunk is needed (C1: sigma2 > 3e-2): the test is overhead
    d.dense = dense ? 1 : 0;
    d.sigma2 = (float)sigma2; d.motion = (float)mo; d.cmax = (float)cmax;
    d.nk_ext2 = (float)(nk * eng.ext2); d.nk_width = (float)(nk * width); d.nk_far2 = (float)(nk * far2);
    d.r_col = r_col; d.r_row = r_row; d.row_off = row_off; d.lean = lean ? 1 : 0;
    d.fused = eng.fused_allowed && col && row && fused_ok ? 1 : 0;
    *eng.dev = d;
    // mailbox: payload first, sequence number last, both at system scope
    EngineDecision* hm = eng.host;
    hm->col = d.col; hm->first = d.first; hm->row = d.row; hm->fine = d.fine; hm->dense = d.dense;
    hm->sigma2 = d.sigma2; hm->motion = d.motion; hm->cmax = d.cmax;
    hm->nk_ext2 = d.nk_ext2; hm->nk_width = d.nk_width; hm->nk_far2 = d.nk_far2;
    hm->r_col = d.r_col; hm->r_row = d.r_row; hm->row_off = d.row_off; hm->lean = d.lean; hm->fused = d.fused;
    __threadfence_system();
    __hip_atomic_store(&hm->seq, d.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace

namespace prg {

// Segments of whole chunks.  The chip holds 768 of these workgroups at a time (3 per CU: 43 KB of LDS, 131 / 160 VGPRs) and
// all of them take the same time in the dense regime, so the grid runs in ceil(blocks * S / 768) rounds: S is chosen in
// [4, 32] to waste the least of the last round (C1: 196 blocks x 19 segments = 4.85 rounds; the first version's 196 x 5 =
// 1.28 rounds lost 36 %); a segment holds at most 64 chunks (one ballot of box tests).
// [r6] How long a dense launch of `blocks` x ceil(chunks / cps) workgroups lasts, in chunk units: the chip takes workgroups in grid
// order (block index fastest, so every block's LAST segment - the short one when cps does not divide the chunks - goes out last)
// as its 768 slots come free.  Simulated, because the remainder matters: a 1/8 shard of C1 (25 blocks, 391 chunks) cut into 28
// segments of 14 runs 14 units; cut into 30 of 13 + one of 1 it runs 13 - the 25 short workgroups slip into the 18 free slots
// and behind each other - where round 4's "rounds x chunks per workgroup" priced it at 26.  `over`: a workgroup's own cost (loading and
// splitting its 512 points, writing its partial plane) in chunk units.
static double mfma_dense_makespan(int64_t blocks, int64_t chunks, int64_t cps, double over) {
    const int64_t segs = ceil_div(chunks, cps), rem = chunks - (segs - 1) * cps;
    std::vector<double> slot(768, 0.0);  // min-heap of the times at which the chip's workgroup slots come free
    auto later = [](double a, double b) { return a > b; };
    double end = 0.0;
    for (int64_t y = 0; y < segs; ++y) {
        const double d = (double)(y + 1 < segs ? cps : rem) + over;
        for (int64_t x = 0; x < blocks; ++x) {
            std::pop_heap(slot.begin(), slot.end(), later);
            const double t = slot.back() + d;
            slot.back() = t;
            std::push_heap(slot.begin(), slot.end(), later);
            end = std::max(end, t);
        }
    }
    return end;
}

int mfma_chunks_per_seg(int64_t owned_points, int64_t streamed_points, int S) {
    const int64_t chunks = ceil_div(streamed_points, kChunk);
    if (S <= 0) {
        const int64_t blocks = ceil_div(owned_points, kWgPoints);
        // (asked several times per E-step with the plan's two or three shapes: remembered)
        static std::mutex mu;
        static std::map<std::pair<int64_t, int64_t>, int> memo;
        static const bool old_model = getenv("PRG_MFMA_SEG_MODEL") && atoi(getenv("PRG_MFMA_SEG_MODEL")) == 4;  // round 4's count of rounds
        std::lock_guard<std::mutex> lock(mu);
        const auto key = std::make_pair(blocks, chunks);
        const auto hit = memo.find(key);
        if (hit != memo.end()) {
            S = hit->second;
        } else {
            double best = 1e30;
            S = 1;
            for (int64_t cand = 4; cand <= 32; ++cand) {
                const int64_t cps = ceil_div(chunks, std::min<int64_t>(cand, chunks));
                if (cps > 64) continue;  // (one ballot of box tests covers 64 chunks)
                const int64_t segs = ceil_div(chunks, cps);
                const double cost = old_model ? (double)ceil_div(blocks * segs, (int64_t)768) * (double)cps
                                              : mfma_dense_makespan(blocks, chunks, cps, 0.15);
                if (cost < best * 0.999) {
                    best = cost;
                    S = (int)cand;
                }
            }
            memo[key] = S;
        }
    }
    const int64_t cps = ceil_div(chunks, std::min<int64_t>(S, chunks));
    return (int)std::min<int64_t>(cps, 64);
}

// Round 4's segment rule (whole rounds of the chip x chunks per workgroup): what the engine switch's cost model was fitted with
// (cpd.hip: engine_leave_below) - its constants describe launches cut this way, and its bounds are held to measured crossovers
// (tests/test_host_logic.py), so the model keeps this rule while the launches themselves follow the simulated schedule above.
int mfma_chunks_per_seg_model(int64_t owned_points, int64_t streamed_points) {
    const int64_t chunks = ceil_div(streamed_points, kChunk), blocks = ceil_div(owned_points, kWgPoints);
    double best = 1e30;
    int64_t best_cps = chunks;
    for (int64_t cand = 4; cand <= 32; ++cand) {
        const int64_t cps = ceil_div(chunks, std::min<int64_t>(cand, chunks));
        const int64_t segs = ceil_div(chunks, cps);
        const double cost = (double)ceil_div(blocks * segs, (int64_t)768) * (double)cps;
        if (cost < best * 0.999) {
            best = cost;
            best_cps = cps;
        }
    }
    return (int)std::min<int64_t>(best_cps, 64);
}

int mfma_planes(int64_t owned_points, int64_t streamed_points, int S) {
    const int cps = mfma_chunks_per_seg(owned_points, streamed_points, S);
    return (int)ceil_div(ceil_div(streamed_points, kChunk), cps);
}

void launch_chunk_meta(prg_cpd* h, const float* gmeta, int64_t cap, float* cmeta) {
    const int64_t nchunk = cap / kChunk;
    k_chunk_meta<<<(unsigned)ceil_div(nchunk, kBlock), kBlock, 0, h->stream>>>(reinterpret_cast<const BoxMeta*>(gmeta), nchunk,
                                                                             reinterpret_cast<BoxMeta*>(cmeta));
}

void launch_chunk_meta_bbox(prg_cpd* h, const EngineArgs* eng) {
    EngineArgs none;
    none.dev = nullptr;
    none.host = nullptr;
    k_chunk_meta_bbox<<<1, kBlock, 0, h->stream>>>(reinterpret_cast<const BoxMeta*>(h->zmeta), h->Mcap / kChunk,
                                                  reinterpret_cast<BoxMeta*>(h->zchunk), h->z4, h->M, h->motion, h->params,
                                                  eng ? *eng : none);
}

// Stream mode (see WorkItem): planes a block's items can occupy, 0 when the mode does not apply - fewer than four units per
// workgroup (nothing to balance) or so few blocks that a block would spread over more planes than the merge kernels take.
constexpr int kStreamSlots = 768;  // what the chip holds of these kernels at a time: 3 per CU (43 KB of LDS each)
// workgroups of a stream-mode launch: whole rounds of the chip's slots (equal runs: r rounds take r run lengths), as many as
// keep a run within the 64 chunks one ballot covers
static int64_t stream_workgroups(int64_t blocks, int64_t chunks) {
    const int64_t units = blocks * chunks;
    return kStreamSlots * ceil_div(units, (int64_t)kStreamSlots * 64);
}
int mfma_stream_planes(int64_t owned_points, int64_t streamed_points) {
    static const bool off = getenv("PRG_MFMA_STREAM") && atoi(getenv("PRG_MFMA_STREAM")) == 0;
    const int64_t blocks = ceil_div(owned_points, kWgPoints), chunks = ceil_div(streamed_points, kChunk);
    if (off || blocks * chunks < 4 * (int64_t)kStreamSlots || blocks * chunks >= (int64_t)1 << 31) return 0;
    // most planes a block's items occupy = most workgroups whose runs touch one block (exact: walked once per shape)
    static thread_local int64_t last_blocks = -1, last_chunks = -1;
    static thread_local int last_planes = 0;
    if (blocks != last_blocks || chunks != last_chunks) {
        const int64_t units = blocks * chunks, g = stream_workgroups(blocks, chunks), base = units / g, rem = units % g;
        auto holder = [&](int64_t x) { return x < rem * (base + 1) ? x / (base + 1) : rem + (x - rem * (base + 1)) / base; };
        int64_t most = 0;
        for (int64_t b = 0; b < blocks; ++b) most = std::max(most, holder((b + 1) * chunks - 1) - holder(b * chunks) + 1);
        last_blocks = blocks;
        last_chunks = chunks;
        last_planes = (int)most;
    }
    return last_planes <= 48 ? last_planes : 0;
}

static StreamCut stream_cut(int planes, int64_t blocks, int64_t chunks) {
    StreamCut c = {0, 0, 0u, 0u};
    if (planes > 0) {
        const int64_t units = blocks * chunks, g = stream_workgroups(blocks, chunks);
        c.g = (int)g;
        c.pmax = planes;
        c.base = (unsigned)(units / g);
        c.rem = (unsigned)(units % g);
    }
    return c;
}

void launch_colpass_mfma(prg_cpd* h, int S, bool first, bool fine, const EngineDecision* guard, bool stream) {
    const int cps = mfma_chunks_per_seg(h->N, h->M, S);
    const int sp = stream ? mfma_stream_planes(h->N, h->M) : 0;
    const int64_t nblocks = ceil_div(h->N, kWgPoints);
    dim3 grid((unsigned)nblocks, (unsigned)ceil_div(ceil_div(h->M, kChunk), cps));
    h->mfma_col_planes = sp ? sp : (int)grid.y;
    const StreamCut cut = stream_cut(sp, nblocks, ceil_div(h->M, kChunk));
    if (sp) grid = dim3((unsigned)cut.g, 1);
    // (zchunk: boxes of this E-step's transformed source, written by launch_chunk_meta_bbox before the engine decision)
    k_colpass_mfma<false><<<grid, kBlock, 0, h->stream>>>(h->tgt4, h->z4, reinterpret_cast<const BoxMeta*>(h->tmeta),
                                                   reinterpret_cast<const BoxMeta*>(h->zmeta),
                                                   reinterpret_cast<const BoxMeta*>(h->zchunk), h->colmin, h->colmin + h->Ncap,
                                                   h->motion + ((h->estep_count - 1) & 1), cps, h->M, h->N, h->params,
                                                   h->colpart, h->Ncap, h->wgcount, h->eng_work, first ? 1 : 0, fine ? 1 : 0, guard,
                                                   cut, nullptr);
    h->wg_col = nblocks * h->mfma_col_planes;  // (evaluated-tile counters: one per (plane, block))
    h->wg_col_pairs = 128.0 * 16.0;  // counted unit: one wave's 128 points x one 16-point tile
    h->dense_pairs_col = 0.0;
}

// the single sweep of a rigid EM iteration (k_colpass_mfma<true>): grid mode, planes of 6 floats per column in h->colpart,
// block origins in h->corig
void launch_fused_mfma(prg_cpd* h, int S, bool first, bool fine, const EngineDecision* guard) {
    const int cps = mfma_chunks_per_seg(h->N, h->M, S);
    const int64_t nblocks = ceil_div(h->N, kWgPoints);
    dim3 grid((unsigned)nblocks, (unsigned)ceil_div(ceil_div(h->M, kChunk), cps));
    h->mfma_col_planes = (int)grid.y;
    const StreamCut none = {0, 0, 0u, 0u};
    k_colpass_mfma<true><<<grid, kBlock, 0, h->stream>>>(h->tgt4, h->z4, reinterpret_cast<const BoxMeta*>(h->tmeta),
                                                         reinterpret_cast<const BoxMeta*>(h->zmeta),
                                                         reinterpret_cast<const BoxMeta*>(h->zchunk), h->colmin, h->colmin + h->Ncap,
                                                         h->motion + ((h->estep_count - 1) & 1), cps, h->M, h->N, h->params,
                                                         h->colpart, h->Ncap, h->wgcount, h->eng_work, first ? 1 : 0, fine ? 1 : 0, guard,
                                                         none, h->corig);
    h->wg_col = nblocks * h->mfma_col_planes;
    h->wg_col_pairs = 128.0 * 16.0;
    h->dense_pairs_col = 0.0;
}

void launch_rowpass_mfma(prg_cpd* h, int S, bool fine, bool lean, bool stream) {
    const int cps = mfma_chunks_per_seg(h->M, h->N, S);
    const int sp = stream && lean ? mfma_stream_planes(h->M, h->N) : 0;  // (stream mode: the lean instantiation only)
    const int64_t nblocks = ceil_div(h->M, kWgPoints);
    dim3 grid((unsigned)nblocks, (unsigned)ceil_div(ceil_div(h->N, kChunk), cps));
    h->mfma_row_planes = sp ? sp : (int)grid.y;
    const StreamCut cut = stream_cut(sp, nblocks, ceil_div(h->N, kChunk));
    if (sp) grid = dim3((unsigned)cut.g, 1);
    // touched flags: 64 bytes per 128-row block, behind the planes this launch writes (k_row_moments is told the same count)
    unsigned char* rowflag = reinterpret_cast<unsigned char*>(h->rowpart + (int64_t)h->mfma_row_planes * 5 * h->Mcap);
    launch_chunk_meta(h, h->tmeta, h->Ncap, h->tchunk);  // target boxes with this E-step's b_n ranges (after k_colfinal)
    auto kernel = sp ? k_rowpass_mfma<true, true> : lean ? k_rowpass_mfma<true, false> : k_rowpass_mfma<false, false>;
    kernel<<<grid, kBlock, 0, h->stream>>>(h->z4, h->tgt4, reinterpret_cast<const BoxMeta*>(h->zmeta),
                                           reinterpret_cast<const BoxMeta*>(h->tmeta), reinterpret_cast<const BoxMeta*>(h->tchunk), cps,
                                           h->N, h->M, h->params, h->rowpart, h->Mcap, h->rorig, rowflag, h->wgcount + h->wg_cap,
                                           h->eng_work + 1, fine ? 1 : 0, cut);
    h->wg_row = nblocks * h->mfma_row_planes;
    h->wg_row_pairs = 128.0 * 16.0;
    h->dense_pairs_row = 0.0;
}

}  // namespace prg
