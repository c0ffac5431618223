// FilterReg rigid point-to-point EM iteration on MI355X (gfx950): GPU permutohedral lattice
// (parallel hash build, atomic splat, blur, slice) + weighted Kabsch reduction.
//
// Reference behaviour (neka-nat/probreg v0.3.7):
//   lattice   third_party/permutohedral/permutohedral.cpp:140-325 (init, SSE build) and :482-616 (compute)
//             behind probreg/gaussian_filtering.py:8-17 / probreg/cc/permutohedral_lattice_py.cc:13-21
//   E-step    probreg/filterreg.py:78-108        M-step  probreg/filterreg.py:158-196 (pt2pt)
//   Kabsch    probreg/cc/kabsch.cc:6-109
//
// The embedding arithmetic (elevate, round-half-even, rank, barycentric) is evaluated in float32 with
// explicitly un-fused operations so that every point lands in the same simplex with the same weights as
// in the reference's SSE build; vertex ids are arbitrary labels (hash order), which no output depends on.
// The splat is a float atomic add, i.e. the summation ORDER differs from the reference's sequential loop
// (float32 round-off only).  Everything here is HBM-latency / atomic bound integer and scatter work.
#include <math.h>

#include <algorithm>
#include <atomic>
#include <cstring>
#include <type_traits>

#include <stdio.h>
#include <stdlib.h>

#include "morton.h"
#include "prg_common.h"
#include "small_linalg.h"

namespace prg {
int sort_pairs_u32(void* tmp, size_t* tmp_bytes, const unsigned* keys_in, unsigned* keys_out, const int* vals_in,
                   int* vals_out, unsigned n, unsigned bits, hipStream_t stream);  // lattice_sort.hip
}

namespace {

constexpr int kBlock = 256;
constexpr unsigned long long kEmpty = 0xFFFFFFFFFFFFFFFFull;
constexpr int kMaxD = 3;

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

__device__ __forceinline__ unsigned long long pack_key(const short* k, int d) {
    unsigned long long r = 0;
    for (int i = 0; i < d; ++i) r |= (unsigned long long)(unsigned short)k[i] << (16 * i);
    return r;
}
__device__ __forceinline__ unsigned long long mix64(unsigned long long x) {
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdull;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ull;
    x ^= x >> 33;
    return x;
}

// Feature producer of the FilterReg plan (filterreg.py:84-85 fused into the embedding): point i < m is the transformed
// source z = R y + t (kept as fp64 for the M-step), point i >= m a target point; both are divided by sigma in fp64
// before the float32 cast, exactly the reference's `t_source / sigma`, `target / sigma` followed by pybind's cast.
struct FrFeat {
    const double* src;
    const double* tgt;
    const double* state;  // [0..8] rot, [9..11] t, [12] sigma2
    double* ts;           // [m][3] transformed source (written by whoever embeds a source point)
    int64_t m;
    int dim;
};

// what the resolve kernel publishes to the host (see k_resolve)
struct LatticeMail { int size, overflow, side_size, side_overflow; unsigned seq; unsigned pad[3]; };

struct Lattice {
    int device = 0;
    hipStream_t stream = nullptr;
    int64_t n = 0;   // embedded points
    int d = 0;
    int with_blur = 1;
    int size = 0;    // number of lattice vertices (host copy)
    const float* pend_vals = nullptr;   // lat_filter(defer_slice): the final value plane, waiting to be sliced
    float pend_alpha = 0.f;
    // device
    float* feat = nullptr;              // [n][d]
    unsigned long long* tkeys = nullptr;  // hash table [cap]
    int* slot_id = nullptr;             // [cap] dense id of an occupied slot
    int64_t cap = 0;
    int* pslot = nullptr;               // [n][d+1] slot, later overwritten by dense id (= offset_)
    float* bary = nullptr;              // [n][d+1]
    unsigned long long* dkeys = nullptr;  // [n*(d+1)] dense keys
    int* nb = nullptr;                  // [2][d+1][size] blur neighbours (dense id or -1)
    int* count = nullptr;               // device counters: [0] vertices, [1] table overflow flag
    int64_t cap_used = 0;               // slots of the table in use for the current build (power of two <= cap)
    // d <= 3: a table entry is (generation << 48) | packed key; an entry of another generation counts as empty, so a
    // build starts by taking the next generation instead of clearing the table (gen 0 = freshly zeroed memory)
    unsigned gen = 0, gen2 = 0;
    int prev_size[2] = {0, 0};          // last lattice size without / with blur: sizes the next hash table
    double* pinned = nullptr;           // 64 doubles of pinned host memory: small device->host read-backs (a copy
                                        // into pageable memory costs ~100 us of staging, this one a few us)
    bool built = false;                 // false after a decision-only build that stopped early (lat_build)
    float* vals = nullptr;              // [2][(size+1)][C] ping-pong value buffers
    int64_t vals_elems = 0;
    int64_t n_alloc = 0, nb_alloc = 0;
    float* io = nullptr;                // staging for values / outputs
    size_t io_bytes = 0;
    const FrFeat* prod = nullptr;       // non-null: features come from the FilterReg plan's clouds, not from `feat`
    // side table of the speculative with_blur decision (fr_build): a 1/16 subset of the points, hashed with the
    // blur scaling while the non-blur lattice is built in the main table; side[0] vertices, side[1] overflow
    unsigned long long* tkeys2 = nullptr;
    int64_t cap2 = 0;
    int* count2 = nullptr;
    bool side_pending = false;
    bool side_fuse = false;            // the side stage is prepared and rides in the next build's first embedding launch
    int side_size = 0, side_overflow = 0;
    // feature lattices (d > 3): keys are d shorts, the table holds a 64-bit hash of them (checked by a second hash)
    short* rem0s = nullptr;             // [n][d+1] rounded remainders of every point (keys are rebuilt from these)
    unsigned char* rank8 = nullptr;     // [n][d+1]
    // order-preserving splat (lat_segments): the (point, remainder) incidences of the splatted points sorted by vertex,
    // within a vertex in the REFERENCE's point order (permutohedral.cpp:491-500 walks the points in order)
    unsigned* skeys = nullptr;          // [2][cap_inc] vertex id of every incidence, before / after the sort
    int* svals = nullptr;               // [2][cap_inc] incidence index (point * (d+1) + remainder), before / after
    int* seg = nullptr;                 // [2][size] first / one-past-last sorted position of every vertex
    int64_t seg_inc_cap = 0, seg_size_cap = 0;
    void* sort_tmp = nullptr;
    size_t sort_tmp_bytes = 0;
    float* terms = nullptr;             // [ch][n_inc] the splat's terms w * in[i] in sorted order (per filter call)
    int64_t terms_elems = 0;
    int* long_list = nullptr;           // vertices whose chains are longer than kLongSeg, then their count
    int64_t long_cap = 0;
    LatticeMail* mail = nullptr;        // mapped, coherent host memory the resolve kernel publishes the counters in
    LatticeMail* mail_dev = nullptr;    // ... as the device addresses it
    unsigned mail_seq = 0;
    bool count_clean = false, count2_clean = false;  // the device counters are zero (cleared by the last resolve)
    long long* fx = nullptr;            // [(size + 1)][ch] fixed-point accumulators of the order-independent splat
                                        // (all zero between filter calls: k_fix_to_float clears what it has read)
    int64_t fx_elems = 0;
    double* fx_scale = nullptr;         // [32] 2^S_k, [32] 2^-S_k, then 32 unsigned: largest |value| per channel (float bits)
    const float* fx_scale_key = nullptr;  // the value array the scales were computed for ...
    int fx_scale_ch = 0;
    bool fx_scale_static = false;       // ... which the owner promises not to change (FilterReg plan: target moments)
    bool seg_valid = false;             // the arrays describe the current lattice for points >= seg_first
    int64_t seg_first = -1;
    const int* ref_pos = nullptr;       // device, may be null (identity): j-th splatted point of the reference's order ->
                                        // its position among the splatted points as the kernels store them
    short* kfull = nullptr;             // [size][d] full key of every lattice vertex
    unsigned long long* gcheck = nullptr;  // [size] second hash of the vertex key | 1 (0 = not yet written)
    float* scale_dev = nullptr;         // [kMaxDG] scale factors of the embedding
    int64_t g_alloc_n = 0, g_alloc_size = 0;
    int g_alloc_d = 0;
};

// ---- embedding (permutohedral.cpp:186-276, SSE build) -------------------------------------------------
// One hash table of a lattice build: entries are (generation << 48) | packed key (see Lattice::gen).
struct EmbedTable {
    unsigned long long* tkeys;
    unsigned long long mask;
    unsigned gen;
    int* count;               // [0] vertices created so far, [1] overflow flag
    int* slot_id;             // may be null (count only)
    unsigned long long* dkeys;
};

#ifndef PRG_PLAIN_PROBE
#define PRG_PLAIN_PROBE 1
#endif
constexpr bool kPlainProbe = PRG_PLAIN_PROBE != 0;
#ifndef PRG_EMBED_VECTOR_STORES
#define PRG_EMBED_VECTOR_STORES 1
#endif
constexpr bool kVectorStores = PRG_EMBED_VECTOR_STORES != 0;

// Embed one point (features f, scale factors s) and insert its D + 1 vertices into table T.  pslot_i / bary_i: where the
// point's slots and barycentric weights go, or null (the side table of the speculative with_blur decision only counts).
template <int D>
__device__ __forceinline__ void embed_insert(const float (&f)[D], float s0, float s1, float s2, const EmbedTable& T,
                                             int lane, int* __restrict__ pslot_i, float* __restrict__ bary_i) {
    constexpr int D1 = D + 1;
    unsigned long long* __restrict__ tkeys = T.tkeys;
    const unsigned long long mask = T.mask;
    const unsigned gen = T.gen;
    int* __restrict__ count = T.count;
    int* __restrict__ slot_id = T.slot_id;
    unsigned long long* __restrict__ dkeys = T.dkeys;
    const unsigned long long gbits = (unsigned long long)gen << 48;
    const float scale[3] = {s0, s1, s2};
    float elevated[D1], rem0[D1], rank[D1], bar[D1 + 1];
    float sm = 0.f;
#pragma unroll
    for (int j = D; j > 0; --j) {
        const float cf = __fmul_rn(f[j - 1], scale[j - 1]);
        elevated[j] = __fsub_rn(sm, __fmul_rn((float)j, cf));
        sm = __fadd_rn(sm, cf);
    }
    elevated[0] = sm;
    const float invd1 = 1.0f / (float)D1, fd1 = (float)D1;
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < D1; ++k) {
        float v = __fmul_rn(invd1, elevated[k]);
        v = rintf(v);  // round half to even (_mm_cvtps_epi32 under the default MXCSR, :214-218)
        rem0[k] = __fmul_rn(v, fd1);
        sum = __fadd_rn(sum, v);
    }
#pragma unroll
    for (int k = 0; k < D1; ++k) rank[k] = 0.f;
#pragma unroll
    for (int a = 0; a < D; ++a) {
        const float da = __fsub_rn(elevated[a], rem0[a]);
#pragma unroll
        for (int b = a + 1; b < D1; ++b) {
            const float db = __fsub_rn(elevated[b], rem0[b]);
            if (da < db) rank[a] += 1.f; else rank[b] += 1.f;
        }
    }
#pragma unroll
    for (int k = 0; k < D1; ++k) {
        rank[k] += sum;
        if (rank[k] < 0.f) { rank[k] += fd1; rem0[k] += fd1; }
        else if (rank[k] >= fd1) { rank[k] -= fd1; rem0[k] -= fd1; }
    }
#pragma unroll
    for (int k = 0; k <= D1; ++k) bar[k] = 0.f;
#pragma unroll
    for (int k = 0; k < D1; ++k) {
        const float v = __fmul_rn(__fsub_rn(elevated[k], rem0[k]), invd1);
        const int p = D - (int)rank[k];
#pragma unroll
        for (int q = 0; q <= D1; ++q) {  // static indexing keeps bar[] in registers
            if (q == p) bar[q] = __fadd_rn(bar[q], v);
            if (q == p + 1) bar[q] = __fsub_rn(bar[q], v);
        }
    }
    bar[0] = __fadd_rn(bar[0], __fadd_rn(1.0f, bar[D1]));
    unsigned nclaim_mask = 0;  // rounds in which this lane created a vertex, with the slot and key of each
    int fslot[D1];             // the slot every round ended on
    int cslot[D1];
    unsigned long long ckey[D1];
#pragma unroll
    for (int r = 0; r < D1; ++r) {
        cslot[r] = 0;
        ckey[r] = 0;
    }
#pragma unroll
    for (int r = 0; r < D1; ++r) {
        short key[D];
#pragma unroll
        for (int k = 0; k < D; ++k) {
            // canonical[r][rank] = r if rank <= D - r else r - (D+1)   (:166-171)
            const int rk = (int)rank[k];
            const int can = (rk <= D - r) ? r : r - D1;
            key[k] = (short)(rem0[k] + (float)can);
        }
        const unsigned long long pk = pack_key(key, D), mine = pk | gbits;
        unsigned long long slot = mix64(pk) & mask;
        bool claimed = false;
        for (int probes = 0;; ++probes) {
            if (probes > 4096) {  // table (sized from the previous lattice) is too small: the host rebuilds
                count[1] = 1;
                slot = 0;
                break;
            }
            // An ordinary (cacheable) read first: it may be stale - each XCD has its own L2, coherent with the others only
            // at kernel boundaries - but an entry of THIS generation never changes once written, so a hit or a slot taken
            // by another key of this build is final, and only a slot that LOOKS free is asked again at device scope
            // (that read goes past the L2s to the memory side and costs several times as much).  Once a vertex exists,
            // the ~N/L points sharing it never issue an atomic - a CAS storm on a few hundred hot keys costs milliseconds
            // when sigma is large.
            unsigned long long cur = kPlainProbe ? tkeys[slot] : 0ull;
            if (kPlainProbe && cur == mine) break;
            if (!kPlainProbe || (unsigned)(cur >> 48) != gen)
                cur = __hip_atomic_load(&tkeys[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((unsigned)(cur >> 48) != gen) {  // empty, or left over from an earlier build
                const unsigned long long old = atomicCAS(&tkeys[slot], cur, mine);
                if (old == cur) {
                    claimed = true;
                    break;
                }
                cur = old;  // somebody of this build got there first
            }
            if (cur == mine) break;
            slot = (slot + 1) & mask;
        }
        if (claimed) {
            nclaim_mask |= 1u << r;
            cslot[r] = (int)slot;
            ckey[r] = pk;
        }
        fslot[r] = (int)slot;
    }
    if (pslot_i) {
        // [r5] D = 3: one 16-byte store per point and array (a wave writes 1 KB contiguous) instead of four 4-byte stores at a
        // stride of 16 bytes each
        if constexpr (D1 == 4 && kVectorStores) {
            *reinterpret_cast<int4*>(pslot_i) = make_int4(fslot[0], fslot[1], fslot[2], fslot[3]);
            *reinterpret_cast<float4*>(bary_i) = make_float4(bar[0], bar[1], bar[2], bar[3]);
        } else {
#pragma unroll
            for (int r = 0; r < D1; ++r) {
                pslot_i[r] = fslot[r];
                bary_i[r] = bar[r];
            }
        }
    }
    // whoever created a vertex numbers it (no scan of the table afterwards): ONE counter update per wave - the lanes'
    // claims of all D + 1 rounds are ranked with ballots, the first claiming lane fetches the base
    unsigned long long bal[D1];
    int total = 0;
#pragma unroll
    for (int r = 0; r < D1; ++r) {
        bal[r] = __ballot((nclaim_mask >> r) & 1u);
        total += __popcll(bal[r]);
    }
    if (total) {
        unsigned long long any = 0;
#pragma unroll
        for (int r = 0; r < D1; ++r) any |= bal[r];
        const int leader = __ffsll((long long)any) - 1;
        int base = 0;
        if (lane == leader) base = atomicAdd(count, total);
        base = __shfl(base, leader, 64);
        if (slot_id) {
            int before = 0;
#pragma unroll
            for (int r = 0; r < D1; ++r) {
                if ((nclaim_mask >> r) & 1u) {
                    const int id = base + before + __popcll(bal[r] & ((1ull << lane) - 1ull));
                    slot_id[cslot[r]] = id;
                    dkeys[id] = ckey[r];
                }
                before += __popcll(bal[r]);
            }
        }
    }
}

// side.tkeys != null: the same points are ALSO embedded with the scale factors (t0, t1, t2) into the side table (count
// only) - the speculative with_blur decision of prg_fr_estep rides in the first launch of the build instead of its own.
template <int D, bool FR>
__global__ __launch_bounds__(kBlock) void k_embed(const float* __restrict__ feat, const FrFeat fr, int64_t first,
                                                  int64_t n, float s0, float s1, float s2, const EmbedTable table,
                                                  int* __restrict__ pslot, float* __restrict__ bary,
                                                  const EmbedTable side, float t0, float t1, float t2, int sample) {
    // sample 0: points [first, n); 1: every 16th point of [0, n) (a sample spread over the whole cloud: it creates most
    // vertices with few lanes of a wave after the same one); 2: all the others
    const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int period = sample >> 2;  // stage 1 takes every period-th point (sample & 3 == 1), stage 2 the others (== 2)
    sample &= 3;
    const int64_t i = sample == 0 ? first + t : (sample == 1 ? (int64_t)period * t : t + t / (period - 1) + 1);
    if (i >= n) return;
    constexpr int D1 = D + 1;
    const int lane = threadIdx.x & 63;
    float f[D];
    if (FR) {
        const double sigma = sqrt(fr.state[12]);
        if (i < fr.m) {
            // [r5] source / target / transformed source are stored 4 doubles per point (x, y, z, 0): two 16-byte accesses per
            // point that a wave issues over contiguous memory, instead of three 8-byte ones at a stride of 24 bytes
            const double2 ya = reinterpret_cast<const double2*>(fr.src)[2 * i], yb = reinterpret_cast<const double2*>(fr.src)[2 * i + 1];
            const double y[3] = {ya.x, ya.y, yb.x};
            double zt[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                double acc = 0.0;
#pragma unroll
                for (int k = 0; k < D; ++k) acc += y[k] * fr.state[3 * r + k];  // dot(points, rot.T), transformation.py:49-50
                const double z = r < D ? acc + fr.state[9 + r] : 0.0;
                zt[r] = z;
                if (r < D) f[r < D ? r : 0] = (float)(z / sigma);
            }
            reinterpret_cast<double2*>(fr.ts)[2 * i] = make_double2(zt[0], zt[1]);
            reinterpret_cast<double2*>(fr.ts)[2 * i + 1] = make_double2(zt[2], 0.0);
        } else {
            const double2 xa = reinterpret_cast<const double2*>(fr.tgt)[2 * (i - fr.m)], xb = reinterpret_cast<const double2*>(fr.tgt)[2 * (i - fr.m) + 1];
            const double xt[3] = {xa.x, xa.y, xb.x};
#pragma unroll
            for (int k = 0; k < D; ++k) f[k] = (float)(xt[k] / sigma);
        }
    } else {
#pragma unroll
        for (int k = 0; k < D; ++k) f[k] = feat[i * D + k];
    }
    embed_insert<D>(f, s0, s1, s2, table, lane, pslot + i * D1, bary + i * D1);
    if (side.tkeys) embed_insert<D>(f, t0, t1, t2, side, lane, nullptr, nullptr);
}

// ---- feature-space lattices, d > 3 (filterreg.py:121, 125-133 with feature_fn = FPFH: d = 33) ---------------------
// The same embedding with run-time d (arrays of d + 1 floats per thread, in scratch).  A key is d shorts and does not fit
// a machine word, so the table stores a 64-bit hash of it; a SECOND, independent 64-bit hash is recorded per vertex and
// checked by every point that lands on the vertex and by every neighbour look-up: two different keys that share a table
// hash are detected (the host then rebuilds with other seeds) instead of silently merged.
constexpr int kMaxDG = 64;
__device__ __forceinline__ unsigned long long hash_shorts(const short* key, int d, unsigned long long seed) {
    unsigned long long h = seed;
    for (int i = 0; i < d; ++i) {
        h ^= (unsigned long long)(unsigned short)key[i];
        h *= 0x100000001b3ull;
        h ^= h >> 29;
    }
    h = mix64(h);
    return h == kEmpty ? h - 1 : h;
}
__device__ __forceinline__ short canonical_key(float rem0, int rk, int r, int d) {
    // canonical[r][rank] = r if rank <= d - r else r - (d + 1)   (permutohedral.cpp:166-171)
    return (short)(rem0 + (float)(rk <= d - r ? r : r - (d + 1)));
}

__global__ __launch_bounds__(kBlock) void k_embed_g(const float* __restrict__ feat, int64_t n, int d,
                                                    const float* __restrict__ scale,
                                                    unsigned long long* __restrict__ tkeys, unsigned long long mask,
                                                    unsigned long long seed, int* __restrict__ pslot,
                                                    float* __restrict__ bary, short* __restrict__ rem0s,
                                                    unsigned char* __restrict__ rank8, int* __restrict__ overflow) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int d1 = d + 1;
    float elevated[kMaxDG + 1], rem0[kMaxDG + 1], rank[kMaxDG + 1], bar[kMaxDG + 2];
    float sm = 0.f;
    for (int j = d; j > 0; --j) {
        const float cf = __fmul_rn(feat[i * d + j - 1], scale[j - 1]);
        elevated[j] = __fsub_rn(sm, __fmul_rn((float)j, cf));
        sm = __fadd_rn(sm, cf);
    }
    elevated[0] = sm;
    const float invd1 = 1.0f / (float)d1, fd1 = (float)d1;
    float sum = 0.f;
    for (int k = 0; k < d1; ++k) {
        float v = rintf(__fmul_rn(invd1, elevated[k]));  // round half to even, as in k_embed
        rem0[k] = __fmul_rn(v, fd1);
        sum = __fadd_rn(sum, v);
        rank[k] = 0.f;
    }
    for (int a = 0; a < d; ++a) {
        const float da = __fsub_rn(elevated[a], rem0[a]);
        for (int b = a + 1; b < d1; ++b) {
            const float db = __fsub_rn(elevated[b], rem0[b]);
            if (da < db) rank[a] += 1.f; else rank[b] += 1.f;
        }
    }
    for (int k = 0; k < d1; ++k) {
        rank[k] += sum;
        if (rank[k] < 0.f) { rank[k] += fd1; rem0[k] += fd1; }
        else if (rank[k] >= fd1) { rank[k] -= fd1; rem0[k] -= fd1; }
    }
    for (int k = 0; k <= d1; ++k) bar[k] = 0.f;
    for (int k = 0; k < d1; ++k) {
        const float v = __fmul_rn(__fsub_rn(elevated[k], rem0[k]), invd1);
        const int p = d - (int)rank[k];
        bar[p] = __fadd_rn(bar[p], v);
        bar[p + 1] = __fsub_rn(bar[p + 1], v);
    }
    bar[0] = __fadd_rn(bar[0], __fadd_rn(1.0f, bar[d1]));
    for (int k = 0; k < d1; ++k) {
        rem0s[i * d1 + k] = (short)rem0[k];
        rank8[i * d1 + k] = (unsigned char)(int)rank[k];
    }
    for (int r = 0; r < d1; ++r) {
        short key[kMaxDG];
        for (int k = 0; k < d; ++k) key[k] = canonical_key(rem0[k], (int)rank[k], r, d);
        const unsigned long long pk = hash_shorts(key, d, seed);
        unsigned long long slot = mix64(pk) & mask;
        for (int probes = 0;; ++probes) {
            if (probes > 4096) {
                *overflow = 1;
                slot = 0;
                break;
            }
            unsigned long long cur = __hip_atomic_load(&tkeys[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (cur == kEmpty) cur = atomicCAS(&tkeys[slot], kEmpty, pk);
            if (cur == kEmpty || cur == pk) break;
            slot = (slot + 1) & mask;
        }
        pslot[i * d1 + r] = (int)slot;
        bary[i * d1 + r] = bar[r];
    }
}

// After compaction / resolve: every (point, remainder) writes its key into its vertex' row of kfull (all writers of a
// vertex write the same d shorts) and checks the vertex' second hash.  flag[0] = 1 on a table-hash collision.
__global__ __launch_bounds__(kBlock) void k_store_keys_g(const int* __restrict__ offset, int64_t n, int d,
                                                         const short* __restrict__ rem0s,
                                                         const unsigned char* __restrict__ rank8,
                                                         unsigned long long seed2, short* __restrict__ kfull,
                                                         unsigned long long* __restrict__ gcheck, int* __restrict__ flag) {
    const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int d1 = d + 1;
    if (t >= n * d1) return;
    const int64_t i = t / d1;
    const int r = (int)(t % d1);
    const int id = offset[t];
    short key[kMaxDG];
    for (int k = 0; k < d; ++k) key[k] = canonical_key((float)rem0s[i * d1 + k], (int)rank8[i * d1 + k], r, d);
    const unsigned long long g = hash_shorts(key, d, seed2) | 1ull;
    const unsigned long long old = atomicCAS(&gcheck[id], 0ull, g);
    if (old == 0ull) {
        for (int k = 0; k < d; ++k) kfull[(int64_t)id * d + k] = key[k];
    } else if (old != g) {
        *flag = 1;
    }
}

__device__ __forceinline__ int lookup_g(const unsigned long long* __restrict__ tkeys, unsigned long long mask,
                                        const int* __restrict__ slot_id, const unsigned long long* __restrict__ gcheck,
                                        const short* key, int d, unsigned long long seed, unsigned long long seed2,
                                        int* __restrict__ flag) {
    const unsigned long long pk = hash_shorts(key, d, seed);
    unsigned long long slot = mix64(pk) & mask;
    for (;;) {
        const unsigned long long k = tkeys[slot];
        if (k == pk) {
            const int id = slot_id[slot];
            if (gcheck[id] != (hash_shorts(key, d, seed2) | 1ull)) *flag = 1;  // same table hash, different key
            return id;
        }
        if (k == kEmpty) return -1;
        slot = (slot + 1) & mask;
    }
}

// blur neighbours (permutohedral.cpp:300-324) from the full keys
__global__ __launch_bounds__(kBlock) void k_neighbours_g(const short* __restrict__ kfull, int size, int d,
                                                         const unsigned long long* __restrict__ tkeys,
                                                         unsigned long long mask, const int* __restrict__ slot_id,
                                                         const unsigned long long* __restrict__ gcheck,
                                                         unsigned long long seed, unsigned long long seed2,
                                                         int* __restrict__ nb1, int* __restrict__ nb2,
                                                         int* __restrict__ flag) {
    const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (t >= (int64_t)size * (d + 1)) return;
    const int j = (int)(t / size), v = (int)(t % size);
    short n1[kMaxDG], n2[kMaxDG];
    for (int k = 0; k < d; ++k) {
        const short key = kfull[(int64_t)v * d + k];
        n1[k] = (short)(k == j ? key + d : key - 1);
        n2[k] = (short)(k == j ? key - d : key + 1);
    }
    nb1[(int64_t)j * size + v] = lookup_g(tkeys, mask, slot_id, gcheck, n1, d, seed, seed2, flag);
    nb2[(int64_t)j * size + v] = lookup_g(tkeys, mask, slot_id, gcheck, n2, d, seed, seed2, flag);
}

// Dense vertex ids for the occupied slots.  One atomic per workgroup (wave ballot + LDS prefix): a lattice of a few
// hundred thousand vertices otherwise serialises that many same-address atomics (~5 ns each).
__global__ __launch_bounds__(kBlock) void k_compact(const unsigned long long* __restrict__ tkeys, int64_t cap,
                                                    int* __restrict__ slot_id, unsigned long long* __restrict__ dkeys,
                                                    int* __restrict__ count) {
    __shared__ int wave_cnt[kBlock / 64];
    __shared__ int block_base;
    const int64_t s = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const unsigned long long k = s < cap ? tkeys[s] : kEmpty;
    const bool occ = k != kEmpty;
    const unsigned long long bal = __ballot(occ);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wave_cnt[wv] = __popcll(bal);
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0;
        for (int w = 0; w < kBlock / 64; ++w) {
            const int c = wave_cnt[w];
            wave_cnt[w] = tot;
            tot += c;
        }
        block_base = tot ? atomicAdd(count, tot) : 0;
    }
    __syncthreads();
    if (occ) {
        const int id = block_base + wave_cnt[wv] + before;
        slot_id[s] = id;
        dkeys[id] = k;
    }
}

// Every (point, remainder) slot index -> the dense vertex id of the slot.  The launch needs nothing the host does not know
// before the embedding has run, so it is issued right behind it; its first thread PUBLISHES the embedding's counters (vertex
// count, overflow flag, and the side table's pair) in the host's mapped mailbox - the host learns the lattice size while this
// kernel runs and enqueues the size-dependent launches behind it, without draining the queue - and clears them for the next
// build (`mail` null: plain resolve).
__global__ __launch_bounds__(kBlock) void k_resolve(int* __restrict__ pslot, int64_t total,
                                                    const int* __restrict__ slot_id, int* __restrict__ count,
                                                    int* __restrict__ count2, LatticeMail* __restrict__ mail,
                                                    unsigned seq) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;  // (one int4 = four entries per thread)
    if (i == 0 && mail) {
        mail->size = count[0];
        mail->overflow = count[1];
        mail->side_size = count2 ? count2[0] : 0;
        mail->side_overflow = count2 ? count2[1] : 0;
        __threadfence_system();
        __hip_atomic_store(&mail->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        count[0] = count[1] = 0;
        if (count2) count2[0] = count2[1] = 0;
    }
    if (4 * i + 3 < total) {
        int4 v = reinterpret_cast<int4*>(pslot)[i];
        v.x = slot_id[v.x];
        v.y = slot_id[v.y];
        v.z = slot_id[v.z];
        v.w = slot_id[v.w];
        reinterpret_cast<int4*>(pslot)[i] = v;
    } else {
        for (int64_t t = 4 * i; t < total; ++t) pslot[t] = slot_id[pslot[t]];
    }
}

__device__ __forceinline__ int lookup(const unsigned long long* __restrict__ tkeys, unsigned long long mask,
                                      const int* __restrict__ slot_id, unsigned long long pk, unsigned gen) {
    unsigned long long slot = mix64(pk) & mask;
    const unsigned long long want = pk | ((unsigned long long)gen << 48);
    for (;;) {
        const unsigned long long k = tkeys[slot];
        if (k == want) return slot_id[slot];
        if ((unsigned)(k >> 48) != gen) return -1;
        slot = (slot + 1) & mask;
    }
}

// blur neighbours (permutohedral.cpp:300-324): along axis j, n1 = key - 1 (all coords) with n1[j] = key[j] + d,
// n2 = key + 1 with n2[j] = key[j] - d; axis j == d only touches the implicit last coordinate.
template <int D>
__global__ __launch_bounds__(kBlock) void k_neighbours(const unsigned long long* __restrict__ dkeys, int size,
                                                       const unsigned long long* __restrict__ tkeys,
                                                       unsigned long long mask, const int* __restrict__ slot_id,
                                                       int* __restrict__ nb1, int* __restrict__ nb2, unsigned gen) {
    const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (t >= (int64_t)size * (D + 1)) return;
    const int j = (int)(t / size), v = (int)(t % size);
    const unsigned long long pk = dkeys[v];
    short key[D], n1[D], n2[D];
#pragma unroll
    for (int k = 0; k < D; ++k) {
        key[k] = (short)(unsigned short)(pk >> (16 * k));
        n1[k] = (short)(key[k] - 1);
        n2[k] = (short)(key[k] + 1);
    }
#pragma unroll
    for (int k = 0; k < D; ++k)
        if (k == j) { n1[k] = (short)(key[k] + D); n2[k] = (short)(key[k] - D); }
    nb1[(int64_t)j * size + v] = lookup(tkeys, mask, slot_id, pack_key(n1, D), gen);
    nb2[(int64_t)j * size + v] = lookup(tkeys, mask, slot_id, pack_key(n2, D), gen);
}

// ---- filtering (permutohedral.cpp:482-616) --------------------------------------------------------------
// vals layout: [(size + 1)][C], row 0 is the all-zero row that the missing-neighbour id -1 maps to.
// FX: the terms are accumulated as 64-bit fixed-point integers (term * 2^S_k, S_k per channel from the largest |value| of the
// channel: k_chan_scale) - integer addition is associative, so the atomics' arrival order no longer matters: the same bits
// in every run, and each vertex' value is the correctly rounded EXACT sum of its terms.  !FX: float atomics (arrival order).
template <bool FX>
__device__ __forceinline__ void splat_add_global(float* __restrict__ vals, long long* __restrict__ fx, int64_t idx, float term,
                                                 double mul) {
    if (FX)
        atomicAdd(reinterpret_cast<unsigned long long*>(fx + idx), (unsigned long long)__double2ll_rn((double)term * mul));
    else
        unsafeAtomicAdd(&vals[idx], term);
}

template <bool FX>
__global__ __launch_bounds__(kBlock) void k_splat(const int* __restrict__ offset, const float* __restrict__ bary,
                                                  const float* __restrict__ in, int64_t first, int64_t n, int d1,
                                                  int ch, float* __restrict__ vals, long long* __restrict__ fx,
                                                  const double* __restrict__ scale) {
    const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (t >= (n - first) * d1) return;
    const int64_t i = first + t / d1;
    const int r = (int)(t % d1);
    const int o = offset[i * d1 + r] + 1;
    const float w = bary[i * d1 + r];
    for (int k = 0; k < ch; ++k) {
        const float p = __fmul_rn(w, in[i * ch + k]);
        if (p != 0.f) splat_add_global<FX>(vals, fx, (int64_t)o * ch + k, p, FX ? scale[k] : 0.0);
    }
}

// Block-private splat: every workgroup accumulates its chunk of points into an LDS hash table keyed by the
// vertex id (ds atomics), then flushes each occupied entry with ONE global atomic per channel.  While the
// lattice is small (sigma large: a few hundred vertices shared by 2M point-vertex incidences) this removes
// the same-address global atomic storm; when a chunk touches more distinct vertices than the table holds,
// the overflow goes straight to global memory, where contention is low by then.
// (256 points / 256 slots per workgroup: 9 KB of LDS (17 KB in fixed point), 8 workgroups per CU.  The first version used 2048 / 2048 = 72 KB:
// 245 workgroups of one wave per SIMD each, every lane walking 32 incidences through dependent loads and returning LDS
// atomics with nobody to hide the latency - 52 % of the wave cycles were waits, profiles/r2_filterreg_500k_pmc.txt)
// [r5] The table's value rows are as wide as the filter has channels (STRIDE 5 for FilterReg's point-to-point pass, 8 otherwise), and
// the five-channel table has 512 slots in the LDS the 256 x 8 one takes (22 KB in fixed point): late EM iterations, where the 1024
// incidences of a workgroup's 256 points spread over more distinct vertices than 192 slots hold, overflow to global atomics less.
constexpr int kSplatMaxCh = 8;
constexpr int kSplatPts = 256;                // points per workgroup
template <bool FX, int kSplatBits, int STRIDE>
__global__ __launch_bounds__(kBlock) void k_splat_lds(const int* __restrict__ offset, const float* __restrict__ bary,
                                                      const float* __restrict__ in, int64_t first, int64_t n, int d1,
                                                      int ch, float* __restrict__ vals, long long* __restrict__ fx,
                                                      const double* __restrict__ scale) {
    typedef typename std::conditional<FX, unsigned long long, float>::type acc_t;
    constexpr int kSplatSlots = 1 << kSplatBits;  // LDS table entries (key + STRIDE channels)
    __shared__ int skey[kSplatSlots];
    __shared__ acc_t sval[kSplatSlots * STRIDE];
    __shared__ int sfill;
    for (int t = threadIdx.x; t < kSplatSlots; t += kBlock) skey[t] = -1;
    for (int t = threadIdx.x; t < kSplatSlots * STRIDE; t += kBlock) sval[t] = (acc_t)0;
    if (threadIdx.x == 0) sfill = 0;
    __syncthreads();
    double mul[kSplatMaxCh];
#pragma unroll
    for (int k = 0; k < kSplatMaxCh; ++k) mul[k] = (FX && k < ch) ? scale[k] : 0.0;
    const int64_t p0 = first + (int64_t)blockIdx.x * kSplatPts;
    const int64_t p1 = (p0 + kSplatPts < n) ? p0 + kSplatPts : n;
    for (int64_t t = (p0 - first) * d1 + threadIdx.x; t < (p1 - first) * d1; t += kBlock) {
        const int64_t i = first + t / d1;
        const int r = (int)(t % d1);
        const float w = bary[i * d1 + r];
        const int o = offset[i * d1 + r] + 1;
        bool any = false;
        float p[kSplatMaxCh];
#pragma unroll
        for (int k = 0; k < kSplatMaxCh; ++k) {
            p[k] = k < ch ? __fmul_rn(w, in[i * ch + k]) : 0.f;
            any |= p[k] != 0.f;
        }
        if (!any) continue;
        unsigned h = ((unsigned)o * 2654435761u) >> (32 - kSplatBits);
        int slot = -1;
        for (int probe = 0; probe < 16; ++probe) {
            int cur = skey[h];
            if (cur == -1 && sfill < kSplatSlots * 3 / 4) {
                cur = atomicCAS(&skey[h], -1, o);
                if (cur == -1) { atomicAdd(&sfill, 1); cur = o; }
            }
            if (cur == o) { slot = (int)h; break; }
            h = (h + 1) & (kSplatSlots - 1);
        }
#pragma unroll
        for (int k = 0; k < kSplatMaxCh; ++k) {
            if (k >= ch || p[k] == 0.f) continue;
            if (slot >= 0) {
                if (FX)
                    atomicAdd(reinterpret_cast<unsigned long long*>(&sval[slot * STRIDE + k]),
                              (unsigned long long)__double2ll_rn((double)p[k] * mul[k]));
                else
                    atomicAdd(reinterpret_cast<float*>(&sval[slot * STRIDE + k]), p[k]);
            } else {
                splat_add_global<FX>(vals, fx, (int64_t)o * ch + k, p[k], mul[k]);
            }
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < kSplatSlots; t += kBlock) {
        const int o = skey[t];
        if (o < 0) continue;
        for (int k = 0; k < ch; ++k) {
            const acc_t v = sval[t * STRIDE + k];
            if (v == (acc_t)0) continue;
            if (FX)
                atomicAdd(reinterpret_cast<unsigned long long*>(fx + (int64_t)o * ch + k), (unsigned long long)v);
            else
                unsafeAtomicAdd(&vals[(int64_t)o * ch + k], (float)v);
        }
    }
}

// largest |in[i][k]| over the splatted rows, per channel -> maxabs[k] as float bits (non-negative floats order like their
// bits); one atomic per workgroup and channel
__global__ __launch_bounds__(kBlock) void k_chan_maxabs(const float* __restrict__ in, int64_t first, int64_t n, int ch,
                                                        unsigned* __restrict__ maxabs) {
    __shared__ unsigned smax[32];
    if (threadIdx.x < 32) smax[threadIdx.x] = 0u;
    __syncthreads();
    const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t total = (n - first) * ch;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    const int64_t step = stride - stride % ch;  // a thread stays on one channel
    float v = 0.f;
    for (int64_t z = t; z < total; z += step) v = fmaxf(v, fabsf(in[first * ch + z]));
    if (t < total && v > 0.f && isfinite(v)) atomicMax(&smax[t % ch], __float_as_uint(v));
    __syncthreads();
    if (threadIdx.x < ch && smax[threadIdx.x]) atomicMax(maxabs + threadIdx.x, smax[threadIdx.x]);
}

// scale[k] = 2^S_k with n * maxabs_k * 2^S_k < 2^61 (no overflow whatever the vertex), scale[ch + k] = 2^-S_k
__global__ void k_chan_scale(const unsigned* __restrict__ maxabs, int ch, double n_points, double* __restrict__ scale) {
    const int k = threadIdx.x;
    if (k >= ch) return;
    const double m = (double)__uint_as_float(maxabs[k]);
    int e = 0;
    if (m > 0.0) {
        (void)frexp(m * n_points, &e);  // m n < 2^e
        e = 61 - e;
        e = e > 120 ? 120 : (e < -120 ? -120 : e);
    }
    scale[k] = ldexp(1.0, e);
    scale[ch + k] = ldexp(1.0, -e);
}

// fixed-point sums -> the float value plane a (row 0 of both planes: the all-zero row)
__global__ __launch_bounds__(kBlock) void k_fix_to_float(long long* __restrict__ fx, int64_t elems, int ch,
                                                         const double* __restrict__ scale, float* __restrict__ vals_a,
                                                         float* __restrict__ vals_b) {
    const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (t >= elems) return;
    const int k = (int)(t % ch);
    vals_a[t] = (float)((double)fx[t] * scale[ch + k]);
    fx[t] = 0;  // ready for the next filter call: the accumulators are never cleared by a fill
    if (t < ch) vals_b[t] = 0.f;
}

// ---- order-preserving splat ------------------------------------------------------------------------------------------
// The reference splats sequentially (permutohedral.cpp:491-500 / :548-556): points in order, per point its d + 1
// vertices, `values[o] += w * in[i]` in float32 - a vertex' value is ONE chain of float additions in point order, and a
// different order gives different float32 bits.  The atomic splats above accumulate in arrival order: run-to-run
// noise of ~1e-7 relative that the lattice amplifies from EM iteration to EM iteration (cell assignment is discontinuous
// in sigma).  Here every vertex' chain is evaluated in the reference's order: the incidences of the splatted points are
// sorted by vertex id with a STABLE sort from an input written in the reference's point order (lat_segments); per filter
// call the terms are gathered into that order (k_seg_gather) and every chain is added up term by term by one thread
// (k_segchain_thread) or, when it is long, by one wave that keeps several hundred terms in flight (k_segchain_wave) - bit
// for bit the reference's float32 values, the same in every run.
__global__ __launch_bounds__(kBlock) void k_seg_keys(const int* __restrict__ offset, const int* __restrict__ ref_pos,
                                                     int64_t first, int64_t n_inc, int d1, unsigned* __restrict__ keys,
                                                     int* __restrict__ vals, int* __restrict__ seg, int64_t seg_elems) {
    const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    for (int64_t z = t; z < seg_elems; z += (int64_t)gridDim.x * kBlock) seg[z] = 0;  // vertices nobody splats into: empty
    if (t >= n_inc) return;
    const int64_t j = t / d1;
    const int r = (int)(t % d1);
    const int64_t p = first + (ref_pos ? (int64_t)ref_pos[j] : j);
    const int64_t inc = p * d1 + r;
    keys[t] = (unsigned)offset[inc];
    vals[t] = (int)inc;
}

__global__ __launch_bounds__(kBlock) void k_seg_bounds(const unsigned* __restrict__ keys, int64_t n_inc, int size,
                                                       int* __restrict__ seg) {
    const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (t >= n_inc) return;
    const unsigned k = keys[t];
    if (t == 0 || keys[t - 1] != k) seg[k] = (int)t;
    if (t + 1 == n_inc || keys[t + 1] != k) seg[size + k] = (int)(t + 1);
}

// Pass 1 of a filter call: the terms `w * in[i]` (permutohedral.cpp:497 / :555, one float multiplication) of every
// incidence in SORTED order, one plane per channel: prod[k * stride + t].  Fully parallel - this is where the scattered
// reads happen; the chains below then stream contiguous memory.
__global__ __launch_bounds__(kBlock) void k_seg_gather(const int* __restrict__ sinc, const float* __restrict__ bary,
                                                       const float* __restrict__ in, int64_t n_inc, int d1, int ch,
                                                       float* __restrict__ prod, int64_t stride,
                                                       int* __restrict__ long_count) {
    const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (t == 0) *long_count = 0;  // (list of the long chains, filled by k_segchain_thread of this filter call)
    if (t >= n_inc) return;
    const int inc = sinc[t];
    const float w = bary[inc];
    const float* __restrict__ row = in + (int64_t)(inc / d1) * ch;
    for (int k = 0; k < ch; ++k) prod[k * stride + t] = __fmul_rn(w, row[k]);
}

// Pass 2a: one thread per vertex adds its chain up, term by term, in order.  Chains longer than kLongSeg are left to
// k_segchain_wave (their vertices are appended to `long_list`).  vals_a / vals_b: the two value planes [(size + 1)][ch];
// row 0 (the "no neighbour" row of the blur) is zeroed in both, every other row of plane a is WRITTEN by whoever owns the
// vertex - nothing is cleared beforehand.
constexpr int kLongSeg = 64;
__global__ __launch_bounds__(kBlock) void k_segchain_thread(const int* __restrict__ seg, const float* __restrict__ prod,
                                                            int64_t stride, int ch, int size, float* __restrict__ vals_a,
                                                            float* __restrict__ vals_b, int* __restrict__ long_list,
                                                            int* __restrict__ long_count) {
    const int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (blockIdx.x == 0 && threadIdx.x < ch) {
        vals_a[threadIdx.x] = 0.f;
        vals_b[threadIdx.x] = 0.f;
    }
    if (v >= size) return;
    const int start = seg[v], end = seg[size + v];
    if (end - start > kLongSeg) {
        long_list[atomicAdd(long_count, 1)] = (int)v;  // (the order of the list is irrelevant: one wave per entry)
        return;
    }
    float acc[kSplatMaxCh];
#pragma unroll
    for (int k = 0; k < kSplatMaxCh; ++k) acc[k] = 0.f;
    // eight terms per channel are fetched before any of them is added: the loads do not depend on the sums, only the
    // additions form a chain
    for (int t0 = start; t0 < end; t0 += 8) {
        float q[8][kSplatMaxCh];
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int k = 0; k < kSplatMaxCh; ++k) q[j][k] = (k < ch && t0 + j < end) ? prod[k * stride + t0 + j] : 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (t0 + j < end)
#pragma unroll
                for (int k = 0; k < kSplatMaxCh; ++k) acc[k] = __fadd_rn(acc[k], q[j][k]);
    }
#pragma unroll
    for (int k = 0; k < kSplatMaxCh; ++k)
        if (k < ch) vals_a[(v + 1) * ch + k] = acc[k];
}

// Pass 2b: one wave per long chain (while the lattice has a few hundred vertices each collects ~10^4 terms).  All 64 lanes
// fetch - 64 consecutive terms per channel and round, kRing rounds in flight -, lane k < ch adds channel k's terms up in
// order out of LDS.  The chain itself is what bounds it: one dependent float addition per term.
constexpr int kRing = 8;
__global__ __launch_bounds__(kBlock) void k_segchain_wave(const int* __restrict__ seg, const float* __restrict__ prod,
                                                          int64_t stride, int ch, int size, float* __restrict__ vals_a,
                                                          const int* __restrict__ long_list,
                                                          const int* __restrict__ long_count) {
    __shared__ __attribute__((aligned(16))) float stage[kBlock / 64][kSplatMaxCh][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t w = (int64_t)blockIdx.x * (kBlock / 64) + wv;
    if (w >= *long_count) return;
    const int v = long_list[w];
    const int start = seg[v], end = seg[size + v];
    float pr[kRing][kSplatMaxCh];
    auto issue = [&](int slot, int base) {
#pragma unroll
        for (int k = 0; k < kSplatMaxCh; ++k) pr[slot][k] = (k < ch && base + lane < end) ? prod[k * stride + base + lane] : 0.f;
    };
#pragma unroll
    for (int s = 0; s < kRing; ++s) issue(s, start + 64 * s);
    float acc = 0.f;
    float (*lds)[64] = stage[wv];
    for (int base = start; base < end; base += 64 * kRing) {
#pragma unroll
        for (int s = 0; s < kRing; ++s) {
            const int b = base + 64 * s;
            if (b < end) {  // wave-uniform
#pragma unroll
                for (int k = 0; k < kSplatMaxCh; ++k)
                    if (k < ch) lds[k][lane] = pr[s][k];
                issue(s, b + 64 * kRing);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // LDS is in order within a wave: compiler ordering only
                const int cnt = end - b < 64 ? end - b : 64;
                if (lane < ch) {
                    const float* __restrict__ c = lds[lane];
                    if (cnt == 64) {
#pragma unroll
                        for (int j = 0; j < 64; j += 4) {
                            const float4 q = *reinterpret_cast<const float4*>(c + j);
                            acc = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(acc, q.x), q.y), q.z), q.w);
                        }
                    } else {
                        for (int j = 0; j < cnt; ++j) acc = __fadd_rn(acc, c[j]);
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
    }
    if (lane < ch) vals_a[(int64_t)(v + 1) * ch + lane] = acc;
}

// seq_mask bit k set: channel k follows seqCompute (0.5*(n1+n2) evaluated in double, :510), else sseCompute.
__global__ __launch_bounds__(kBlock) void k_blur(const float* __restrict__ old, float* __restrict__ nw,
                                                 const int* __restrict__ nb1, const int* __restrict__ nb2, int size,
                                                 int ch, unsigned seq_mask) {
    const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (t >= (int64_t)size * ch) return;
    const int v = (int)(t / ch), k = (int)(t % ch);
    const float a = old[(int64_t)(nb1[v] + 1) * ch + k], b = old[(int64_t)(nb2[v] + 1) * ch + k];
    const float o = old[(int64_t)(v + 1) * ch + k];
    const float s = __fadd_rn(a, b);
    float r;
    if ((seq_mask >> k) & 1u)
        r = (float)((double)o + 0.5 * (double)s);
    else
        r = __fadd_rn(o, __fmul_rn(0.5f, s));
    nw[(int64_t)(v + 1) * ch + k] = r;
}

__global__ __launch_bounds__(kBlock) void k_slice(const int* __restrict__ offset, const float* __restrict__ bary,
                                                  const float* __restrict__ vals, int64_t n_out, int d1, int ch,
                                                  float alpha, unsigned seq_mask, float* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (t >= n_out * ch) return;
    const int64_t i = t / ch;
    const int k = (int)(t % ch);
    float acc = 0.f;
    for (int r = 0; r < d1; ++r) {
        const int o = offset[i * d1 + r] + 1;
        const float w = bary[i * d1 + r];
        const float v = vals[(int64_t)o * ch + k];
        if ((seq_mask >> k) & 1u)
            acc = __fadd_rn(acc, __fmul_rn(__fmul_rn(w, v), alpha));   // (:526) w * value * alpha
        else
            acc = __fadd_rn(acc, __fmul_rn(__fmul_rn(w, alpha), v));   // (:588-590) (w*alpha) * value
    }
    out[t] = acc;
}

int lat_free(Lattice* L) {
    void* ptrs[] = {L->feat, L->tkeys, L->slot_id, L->pslot, L->bary, L->dkeys, L->nb, L->count, L->vals, L->io,
                    L->rem0s, L->rank8, L->kfull, L->gcheck, L->scale_dev, L->tkeys2, L->count2,
                    L->skeys, L->svals, L->seg, L->sort_tmp, L->terms, L->long_list, L->fx, L->fx_scale};
    L->skeys = nullptr; L->svals = nullptr; L->seg = nullptr; L->sort_tmp = nullptr; L->terms = nullptr; L->long_list = nullptr; L->fx = nullptr; L->fx_scale = nullptr;
    L->fx_elems = 0; L->fx_scale_key = nullptr;
    L->seg_inc_cap = L->seg_size_cap = L->terms_elems = L->long_cap = 0;
    L->sort_tmp_bytes = 0;
    L->seg_valid = false;
    L->tkeys2 = nullptr; L->count2 = nullptr; L->cap2 = 0;
    L->rem0s = nullptr; L->rank8 = nullptr; L->kfull = nullptr; L->gcheck = nullptr; L->scale_dev = nullptr;
    L->g_alloc_n = L->g_alloc_size = 0;
    L->g_alloc_d = 0;
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    L->feat = nullptr; L->tkeys = nullptr; L->slot_id = nullptr; L->pslot = nullptr; L->bary = nullptr;
    L->dkeys = nullptr; L->nb = nullptr; L->count = nullptr; L->vals = nullptr; L->io = nullptr;
    L->n_alloc = L->nb_alloc = L->vals_elems = 0;
    L->io_bytes = 0;
    L->cap = 0;
    if (L->pinned) (void)hipHostFree(L->pinned);
    L->pinned = nullptr;
    if (L->mail) (void)hipHostFree(L->mail);
    L->mail = nullptr;
    L->mail_dev = nullptr;
    L->count_clean = L->count2_clean = false;
    return PRG_OK;
}

// capacity for a buffer that has to hold `need` elements now and at most `worst` ever: four times the need, at least 4M
// elements, never more than the worst case - the lattice of an EM registration grows every iteration, and every
// reallocation (hipFree synchronises the device) is a bubble in the stream
int64_t lat_grow(int64_t need, int64_t worst) {
    return std::max<int64_t>(need, std::min<int64_t>(worst, std::max<int64_t>(4 * need, (int64_t)1 << 22)));
}

int lat_ensure_io(Lattice* L, size_t bytes) {
    if (L->io && L->io_bytes >= bytes) return PRG_OK;
    if (L->io) (void)hipFree(L->io);
    L->io = nullptr;
    L->io_bytes = 0;
    PRG_HIP(hipMalloc((void**)&L->io, bytes));
    L->io_bytes = bytes;
    return PRG_OK;
}

// Build the lattice over L->feat (device, n x d float32).  Synchronises (the vertex count is needed on the host,
// as in the reference where get_lattice_size() drives the with_blur decision, filterreg.py:90-91).
// decide_above >= 0: the caller only wants this lattice if it has at most `decide_above` vertices.  The points are
// then embedded in two stages (1/16 of them first): the vertices of a subset are a subset of the vertices, so as
// soon as the count exceeds the threshold the answer is known and the build stops (L->built = false, L->size = a
// lower bound > decide_above) - no compaction, no neighbour tables, 15/16 of the hashing saved.
int lat_build_generic(Lattice* L, int64_t n, int d, int with_blur);

// scale_factor[i] = float(1/sqrt((i+2)(i+1)) * inv_std_dev), inv_std_dev a float (permutohedral.cpp:180-183)
void lat_scale(int d, int with_blur, float (&sc)[3]) {
    const int d1 = d + 1;
    const float inv_std = with_blur ? (float)(sqrt(2.0 / 3.0) * d1) : (float)(sqrt(1.0 / 6.0) * d1);
    for (int i = 0; i < 3; ++i) sc[i] = i < d ? (float)(1.0 / sqrt((double)((i + 2) * (i + 1))) * (double)inv_std) : 0.f;
}

// every how-many-th point creates the vertices in stage 1 of a build (the others look them up in stage 2); PRG_EMBED_PERIOD for A/B runs
static int embed_period() {
    static const int p = getenv("PRG_EMBED_PERIOD") ? std::max(2, std::min(64, atoi(getenv("PRG_EMBED_PERIOD")))) : 16;
    return p;
}

// Embedding launch over points [first, last) into `table`; side (tkeys may be null): the same points also go, with the
// scale factors ssc, into the side table of the speculative with_blur decision.
void launch_embed(Lattice* L, int d, int64_t first, int64_t last, const float (&sc)[3], const EmbedTable& table,
                  const EmbedTable& side, const float (&ssc)[3], int sample = 0) {
    // sample 1 / 2: first must be 0; the every-16th sample has ceil(last / 16) points, the complement the rest
    const int period = embed_period();
    const int64_t count = sample == 0 ? last - first : (sample == 1 ? prg::ceil_div(last, period) : last - prg::ceil_div(last, period));
    if (sample) sample |= period << 2;
    const unsigned nb = (unsigned)prg::ceil_div(count, kBlock);
    if (nb == 0) return;
    hipStream_t st = L->stream;
    const FrFeat none = {nullptr, nullptr, nullptr, nullptr, 0, 0};
#define PRG_EMBED(DD)                                                                                              \
    if (L->prod)                                                                                                    \
        k_embed<DD, true><<<nb, kBlock, 0, st>>>(nullptr, *L->prod, first, last, sc[0], sc[1], sc[2], table,        \
                                                 L->pslot, L->bary, side, ssc[0], ssc[1], ssc[2], sample);          \
    else                                                                                                            \
        k_embed<DD, false><<<nb, kBlock, 0, st>>>(L->feat, none, first, last, sc[0], sc[1], sc[2], table,           \
                                                  L->pslot, L->bary, side, ssc[0], ssc[1], ssc[2], sample)
    if (d == 1) { PRG_EMBED(1); }
    else if (d == 2) { PRG_EMBED(2); }
    else { PRG_EMBED(3); }
#undef PRG_EMBED
}

// next generation of a table (see Lattice::gen): wraps by zeroing the table once every 65535 builds
static int next_generation(unsigned long long* table, int64_t cap, unsigned* gen, hipStream_t st) {
    if (*gen >= 0xFFFFu) {
        PRG_HIP(hipMemsetAsync(table, 0, cap * sizeof(unsigned long long), st));
        *gen = 0;
    }
    ++*gen;
    return PRG_OK;
}

int lat_build(Lattice* L, int64_t n, int d, int with_blur, int64_t decide_above = -1) {
    PRG_REQUIRE(d >= 1 && d <= kMaxDG, PRG_ERR_INVALID, "permutohedral lattice: feature dimension %d not in [1, %d]", d,
                kMaxDG);
    if (d > kMaxD) return lat_build_generic(L, n, d, with_blur);
    const int d1 = d + 1;
    hipStream_t st = L->stream;
    if (n > L->n_alloc || d != L->d) {
        const int64_t na = n;
        for (void* p : {(void*)L->tkeys, (void*)L->slot_id, (void*)L->pslot, (void*)L->bary, (void*)L->dkeys})
            if (p) (void)hipFree(p);
        int64_t cap = 1;
        while (cap < 2 * na * d1) cap <<= 1;
        L->cap = cap;
        PRG_HIP(hipMalloc((void**)&L->tkeys, cap * sizeof(unsigned long long)));
        PRG_HIP(hipMalloc((void**)&L->slot_id, cap * sizeof(int)));
        PRG_HIP(hipMalloc((void**)&L->pslot, na * d1 * sizeof(int)));
        PRG_HIP(hipMalloc((void**)&L->bary, na * d1 * sizeof(float)));
        PRG_HIP(hipMalloc((void**)&L->dkeys, na * d1 * sizeof(unsigned long long)));
        if (!L->count) PRG_HIP(hipMalloc((void**)&L->count, 2 * sizeof(int)));
        PRG_HIP(hipMemsetAsync(L->tkeys, 0, cap * sizeof(unsigned long long), st));
        L->gen = 0;
        L->n_alloc = na;
    }
    L->n = n;
    L->d = d;
    L->with_blur = with_blur;
    float sc[3];
    lat_scale(d, with_blur, sc);
    // The table is sized from the previous lattice of the same kind (x8..16 head room: the lattice at most doubles
    // per EM iteration) so that the probes stay inside a few cache lines' worth of slots; an overflow falls back to the
    // worst-case size.  Nothing is cleared: the build takes the next generation of the table.
    const int mode = with_blur ? 1 : 0;
    int64_t capu = L->cap;
    if (L->prev_size[mode] > 0) {
        capu = 65536;
        while (capu < 8 * (int64_t)L->prev_size[mode]) capu <<= 1;
        if (capu > L->cap) capu = L->cap;
    }
    L->built = false;
    L->seg_valid = false;
    for (int attempt = 0; attempt < 2; ++attempt) {
        L->cap_used = capu;
        PRG_TRY(next_generation(L->tkeys, L->cap, &L->gen, st));
        if (!L->count_clean) PRG_HIP(hipMemsetAsync(L->count, 0, 2 * sizeof(int), st));  // (normally left clean by k_resolve)
        L->count_clean = false;
        const unsigned long long mask = (unsigned long long)capu - 1;
        const EmbedTable main_table = {L->tkeys, mask, L->gen, L->count, L->slot_id, L->dkeys};
        const EmbedTable no_side = {nullptr, 0, 0, nullptr, nullptr, nullptr};
        const float no_sc[3] = {0.f, 0.f, 0.f};
        // stage 1 = every 16th point, stage 2 = the others (PRG_EMBED_CONTIGUOUS=1: the first sixteenth / the rest, as in round 2)
        static const bool contiguous = getenv("PRG_EMBED_CONTIGUOUS") != nullptr;
        auto embed = [&](int64_t first, int64_t last) { launch_embed(L, d, first, last, sc, main_table, no_side, no_sc); };
        auto embed_stage = [&](int stage, const EmbedTable& side_table, const float (&side_sc)[3]) {
            if (contiguous) {
                if (stage == 1) launch_embed(L, d, 0, n / 16, sc, main_table, side_table, side_sc);
                else launch_embed(L, d, n / 16, n, sc, main_table, side_table, side_sc);
            } else {
                launch_embed(L, d, 0, n, sc, main_table, side_table, side_sc, stage);
            }
        };
        if (!L->pinned) PRG_HIP(hipHostMalloc((void**)&L->pinned, 64 * sizeof(double), hipHostMallocDefault));
        if (!L->mail) {
            PRG_HIP(hipHostMalloc((void**)&L->mail, sizeof(LatticeMail), hipHostMallocMapped | hipHostMallocCoherent));
            memset(L->mail, 0, sizeof(LatticeMail));
            PRG_HIP(hipHostGetDevicePointer((void**)&L->mail_dev, L->mail, 0));
        }
        int host[2] = {0, 0};
        int64_t done = 0;
        if (decide_above >= 0 && n >= 4096) {  // stage 1: a sixteenth of the points; the vertex counter tells
            done = n / 16;
            embed_stage(1, no_side, no_sc);
            PRG_HIP(hipGetLastError());
            PRG_HIP(hipMemcpyAsync(L->pinned, L->count, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
            PRG_HIP(hipStreamSynchronize(st));
            host[0] = reinterpret_cast<volatile int*>(L->pinned)[0];
            host[1] = reinterpret_cast<volatile int*>(L->pinned)[1];
            if (host[1] == 0 && host[0] > decide_above) {
                L->size = host[0];
                if (getenv("PRG_DEBUG_LATTICE"))
                    fprintf(stderr, "[lattice] decision after %lld of %lld points: >= %d vertices > %lld (blur %d)\n",
                            (long long)done, (long long)n, host[0], (long long)decide_above, with_blur);
                return PRG_OK;  // prev_size[mode] keeps the last full count
            }
        } else if (n >= 4096) {
            // no decision to take, but the table is still filled in two launches: the first sixteenth of the points
            // creates most vertices almost uncontended, the rest then find them with plain reads - one launch over
            // all points has every wave compare-and-swap the same few hundred empty slots at once (3x slower while
            // the lattice is small; [r4] once it has tens of thousands of vertices a single launch is neither faster nor
            // slower - measured at C4 with the switch at 8k / 32k / 128k vertices: 4855 / 4864 / 4871 / 4885 it/s - so the
            // two stages stay unconditional)
            done = n / 16;
            if (L->side_fuse) {  // the prepared side stage (lat_side_stage) covers exactly these points: one launch for both
                float bsc[3];
                lat_scale(d, 1, bsc);
                const EmbedTable side = {L->tkeys2, (unsigned long long)L->cap2 - 1, L->gen2, L->count2, nullptr, nullptr};
                embed_stage(1, side, bsc);
                L->side_fuse = false;
                L->side_pending = true;
            } else {
                embed_stage(1, no_side, no_sc);
            }
        }
        if (host[1] == 0) {
            if (done > 0) embed_stage(2, no_side, no_sc);
            else embed(0, n);
            // the resolve pass goes out right behind the embedding and tells the host the counters while it runs: no
            // device-to-host copy, no stream synchronisation, the queue does not drain (on an overflow - rare - it has
            // resolved garbage, which the retry overwrites)
            const unsigned seq = ++L->mail_seq;
            k_resolve<<<(unsigned)prg::ceil_div(prg::ceil_div(n * d1, 4), kBlock), kBlock, 0, st>>>(L->pslot, n * d1, L->slot_id, L->count,
                                                                                 L->side_pending ? L->count2 : nullptr,
                                                                                 L->mail_dev, seq);
            PRG_HIP(hipGetLastError());
            volatile LatticeMail* mb = L->mail;
            {
                hipError_t werr;
                const bool got = prg::wait_mailbox(&mb->seq, seq, st, &werr);
                PRG_HIP(werr);
                PRG_REQUIRE(got, PRG_ERR_HIP, "permutohedral lattice: the vertex count never reached the host");
            }
            host[0] = mb->size;
            host[1] = mb->overflow;
            L->count_clean = true;
            if (L->side_pending) {
                L->side_size = mb->side_size;
                L->side_overflow = mb->side_overflow;
                L->side_pending = false;
                L->count2_clean = true;
            }
        }
        L->size = host[0];
        if (getenv("PRG_DEBUG_LATTICE"))
            fprintf(stderr, "[lattice] attempt %d capu %lld prev %d size %d overflow %d blur %d\n", attempt, (long long)capu,
                    L->prev_size[mode], L->size, host[1], with_blur);
        if (host[1] == 0 && (int64_t)L->size * 2 <= capu) break;
        PRG_REQUIRE(capu < L->cap, PRG_ERR_STATE, "permutohedral lattice: hash table overflow at full capacity");
        capu = L->cap;
    }
    L->built = true;
    L->prev_size[mode] = L->size;
    const unsigned long long mask = (unsigned long long)L->cap_used - 1;  // (k_resolve is already in the queue)
    if (with_blur) {
        const int64_t need = 2 * (int64_t)d1 * L->size;
        if (need > L->nb_alloc) {
            // (generous: the lattice grows from iteration to iteration and hipFree / hipMalloc drain the device)
            const int64_t want = lat_grow(need, 2 * (int64_t)d1 * n * d1);
            if (L->nb) (void)hipFree(L->nb);
            L->nb = nullptr;
            PRG_HIP(hipMalloc((void**)&L->nb, want * sizeof(int)));
            L->nb_alloc = want;
        }
        int* nb1 = L->nb;
        int* nb2 = L->nb + (int64_t)d1 * L->size;
        const unsigned g = (unsigned)prg::ceil_div((int64_t)L->size * d1, kBlock);
        if (d == 1) k_neighbours<1><<<g, kBlock, 0, st>>>(L->dkeys, L->size, L->tkeys, mask, L->slot_id, nb1, nb2, L->gen);
        else if (d == 2) k_neighbours<2><<<g, kBlock, 0, st>>>(L->dkeys, L->size, L->tkeys, mask, L->slot_id, nb1, nb2, L->gen);
        else k_neighbours<3><<<g, kBlock, 0, st>>>(L->dkeys, L->size, L->tkeys, mask, L->slot_id, nb1, nb2, L->gen);
        PRG_HIP(hipGetLastError());
    }
    return PRG_OK;
}

// Decision stage of the blurred lattice into the side table, WITHOUT synchronising and without a launch of its own: 1/16
// of the points are hashed with the blur scaling (by the next lat_build's first embedding launch, side_fuse) and counted
// as they create vertices; the count is read back by that lat_build (side_pending).  A subset's vertices are a subset of the vertices, so side_size > threshold proves that the blurred
// lattice is too large.
int lat_side_stage(Lattice* L, int64_t n, int d) {
    const int d1 = d + 1;
    hipStream_t st = L->stream;
    const int64_t n16 = prg::ceil_div(n, embed_period());
    int64_t want = 1;
    while (want < 4 * n16 * d1) want <<= 1;
    if (want > L->cap2) {
        if (L->tkeys2) (void)hipFree(L->tkeys2);
        L->tkeys2 = nullptr;
        PRG_HIP(hipMalloc((void**)&L->tkeys2, want * sizeof(unsigned long long)));
        PRG_HIP(hipMemsetAsync(L->tkeys2, 0, want * sizeof(unsigned long long), st));
        L->gen2 = 0;
        L->cap2 = want;
    }
    if (!L->count2) PRG_HIP(hipMalloc((void**)&L->count2, 2 * sizeof(int)));
    if (!L->pinned) PRG_HIP(hipHostMalloc((void**)&L->pinned, 64 * sizeof(double), hipHostMallocDefault));
    float sc[3];
    lat_scale(d, 1, sc);
    PRG_TRY(next_generation(L->tkeys2, L->cap2, &L->gen2, st));
    if (!L->count2_clean) PRG_HIP(hipMemsetAsync(L->count2, 0, 2 * sizeof(int), st));
    L->count2_clean = false;
    (void)sc;
    L->side_fuse = true;  // launched by the next lat_build on this lattice, together with its first sixteenth
    return PRG_OK;
}

// Feature lattices (3 < d <= 64): the structure of lat_build with the hashed-key kernels above.  Synchronises.
int lat_build_generic(Lattice* L, int64_t n, int d, int with_blur) {
    const int d1 = d + 1;
    hipStream_t st = L->stream;
    if (n > L->n_alloc || d != L->d) {
        for (void* p : {(void*)L->tkeys, (void*)L->slot_id, (void*)L->pslot, (void*)L->bary, (void*)L->dkeys})
            if (p) (void)hipFree(p);
        L->tkeys = nullptr; L->slot_id = nullptr; L->pslot = nullptr; L->bary = nullptr; L->dkeys = nullptr;
        int64_t cap = 1;
        while (cap < 2 * n * d1) cap <<= 1;
        L->cap = cap;
        PRG_HIP(hipMalloc((void**)&L->tkeys, cap * sizeof(unsigned long long)));
        PRG_HIP(hipMalloc((void**)&L->slot_id, cap * sizeof(int)));
        PRG_HIP(hipMalloc((void**)&L->pslot, n * d1 * sizeof(int)));
        PRG_HIP(hipMalloc((void**)&L->bary, n * d1 * sizeof(float)));
        PRG_HIP(hipMalloc((void**)&L->dkeys, n * d1 * sizeof(unsigned long long)));
        if (!L->count) PRG_HIP(hipMalloc((void**)&L->count, 2 * sizeof(int)));
        L->n_alloc = n;
        L->prev_size[0] = L->prev_size[1] = 0;
    }
    if (n > L->g_alloc_n || d != L->g_alloc_d) {
        if (L->rem0s) (void)hipFree(L->rem0s);
        if (L->rank8) (void)hipFree(L->rank8);
        L->rem0s = nullptr; L->rank8 = nullptr;
        PRG_HIP(hipMalloc((void**)&L->rem0s, n * d1 * sizeof(short)));
        PRG_HIP(hipMalloc((void**)&L->rank8, n * d1));
        L->g_alloc_n = n;
        L->g_alloc_d = d;
    }
    if (!L->scale_dev) PRG_HIP(hipMalloc((void**)&L->scale_dev, kMaxDG * sizeof(float)));
    if (!L->pinned) PRG_HIP(hipHostMalloc((void**)&L->pinned, 64 * sizeof(double), hipHostMallocDefault));
    L->n = n;
    L->d = d;
    L->with_blur = with_blur;
    // scale_factor[i] = float(1/sqrt((i+2)(i+1)) * inv_std_dev), inv_std_dev a float (permutohedral.cpp:180-183)
    const float inv_std = with_blur ? (float)(sqrt(2.0 / 3.0) * d1) : (float)(sqrt(1.0 / 6.0) * d1);
    float sc[kMaxDG];
    for (int i = 0; i < kMaxDG; ++i) sc[i] = i < d ? (float)(1.0 / sqrt((double)((i + 2) * (i + 1))) * (double)inv_std) : 0.f;
    PRG_HIP(hipMemcpyAsync(L->scale_dev, sc, sizeof(sc), hipMemcpyHostToDevice, st));
    PRG_HIP(hipStreamSynchronize(st));  // (sc lives on this stack frame)
    volatile int* host = reinterpret_cast<volatile int*>(L->pinned);
    L->built = false;
    L->seg_valid = false;
    for (int attempt = 0; attempt < 4; ++attempt) {
        const unsigned long long seed = 0x9e3779b97f4a7c15ull * (2 * attempt + 1), seed2 = 0xc2b2ae3d27d4eb4full * (2 * attempt + 3);
        const int64_t capu = L->cap;
        L->cap_used = capu;
        const unsigned long long mask = (unsigned long long)capu - 1;
        PRG_HIP(hipMemsetAsync(L->tkeys, 0xFF, capu * sizeof(unsigned long long), st));
        PRG_HIP(hipMemsetAsync(L->count, 0, 2 * sizeof(int), st));
        L->count_clean = false;
        k_embed_g<<<(unsigned)prg::ceil_div(n, kBlock), kBlock, 0, st>>>(L->feat, n, d, L->scale_dev, L->tkeys, mask, seed,
                                                                         L->pslot, L->bary, L->rem0s, L->rank8, L->count + 1);
        k_compact<<<(unsigned)prg::ceil_div(capu, kBlock), kBlock, 0, st>>>(L->tkeys, capu, L->slot_id, L->dkeys, L->count);
        PRG_HIP(hipGetLastError());
        PRG_HIP(hipMemcpyAsync(L->pinned, L->count, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
        PRG_HIP(hipStreamSynchronize(st));
        PRG_REQUIRE(host[1] == 0, PRG_ERR_STATE, "permutohedral lattice: hash table overflow at full capacity");
        L->size = host[0];
        if ((int64_t)L->size > L->g_alloc_size) {
            if (L->kfull) (void)hipFree(L->kfull);
            if (L->gcheck) (void)hipFree(L->gcheck);
            L->kfull = nullptr; L->gcheck = nullptr;
            const int64_t want = (int64_t)L->size + L->size / 4 + 1024;
            PRG_HIP(hipMalloc((void**)&L->kfull, want * d * sizeof(short)));
            PRG_HIP(hipMalloc((void**)&L->gcheck, want * sizeof(unsigned long long)));
            L->g_alloc_size = want;
        }
        k_resolve<<<(unsigned)prg::ceil_div(prg::ceil_div(n * d1, 4), kBlock), kBlock, 0, st>>>(L->pslot, n * d1, L->slot_id, nullptr, nullptr, nullptr, 0u);
        PRG_HIP(hipMemsetAsync(L->gcheck, 0, (size_t)L->size * sizeof(unsigned long long), st));
        PRG_HIP(hipMemsetAsync(L->count + 1, 0, sizeof(int), st));  // now the collision flag
        k_store_keys_g<<<(unsigned)prg::ceil_div(n * d1, kBlock), kBlock, 0, st>>>(L->pslot, n, d, L->rem0s, L->rank8, seed2,
                                                                                  L->kfull, L->gcheck, L->count + 1);
        if (with_blur) {
            const int64_t need = 2 * (int64_t)d1 * L->size;
            if (need > L->nb_alloc) {
                if (L->nb) (void)hipFree(L->nb);
                L->nb = nullptr;
                PRG_HIP(hipMalloc((void**)&L->nb, need * sizeof(int)));
                L->nb_alloc = need;
            }
            k_neighbours_g<<<(unsigned)prg::ceil_div((int64_t)L->size * d1, kBlock), kBlock, 0, st>>>(
                L->kfull, L->size, d, L->tkeys, mask, L->slot_id, L->gcheck, seed, seed2, L->nb,
                L->nb + (int64_t)d1 * L->size, L->count + 1);
        }
        PRG_HIP(hipGetLastError());
        PRG_HIP(hipMemcpyAsync(L->pinned, L->count, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
        PRG_HIP(hipStreamSynchronize(st));
        if (host[1] == 0) {
            L->built = true;
            return PRG_OK;
        }
        // two different keys shared a 64-bit table hash (probability ~1e-7 per build): other seeds, again
    }
    prg::set_error("permutohedral lattice: key hash collisions persisted over 4 seeds");
    return PRG_ERR_STATE;
}

// How the splat accumulates (prg_lattice_set_splat_mode; process-wide):
//   0  float atomics in arrival order (round-off level run-to-run noise; measurement baseline)
//   1  fixed-point atomics (default): order-independent - the same bits in every run, each vertex the correctly rounded exact sum
//   2  the reference's own order: every vertex one sequential float32 chain in point order - the reference's bits
//      (permutohedral.cpp:491-500); costs a sort of the incidences per lattice and a strictly sequential chain per vertex
static int g_splat_mode = []() {
    const char* e = getenv("PRG_SPLAT_MODE");
    const int m = e ? atoi(e) : 1;
    return m < 0 || m > 2 ? 1 : m;
}();

// Sorted incidence lists of the current lattice for the points >= first (once per lattice build; every filter call on the
// lattice reuses them).  No synchronisation.
int lat_segments(Lattice* L, int64_t first) {
    if (L->seg_valid && L->seg_first == first) return PRG_OK;
    const int d1 = L->d + 1;
    hipStream_t st = L->stream;
    const int64_t n_inc = (L->n - first) * d1;
    PRG_REQUIRE(n_inc > 0 && n_inc < (int64_t)1 << 31 && L->n * d1 < (int64_t)1 << 31, PRG_ERR_INVALID,
                "permutohedral lattice: too many point-vertex incidences for the ordered splat");
    if (n_inc > L->seg_inc_cap) {
        if (L->skeys) (void)hipFree(L->skeys);
        if (L->svals) (void)hipFree(L->svals);
        L->skeys = nullptr; L->svals = nullptr;
        PRG_HIP(hipMalloc((void**)&L->skeys, 2 * n_inc * sizeof(unsigned)));
        PRG_HIP(hipMalloc((void**)&L->svals, 2 * n_inc * sizeof(int)));
        L->seg_inc_cap = n_inc;
    }
    if ((int64_t)L->size > L->seg_size_cap) {
        if (L->seg) (void)hipFree(L->seg);
        L->seg = nullptr;
        const int64_t want = (int64_t)L->size + L->size / 4 + 1024;
        PRG_HIP(hipMalloc((void**)&L->seg, 2 * want * sizeof(int)));
        L->seg_size_cap = want;
    }
    unsigned bits = 1;
    while (((int64_t)1 << bits) < (int64_t)L->size) ++bits;
    size_t need = 0;
    PRG_TRY(prg::sort_pairs_u32(nullptr, &need, L->skeys, L->skeys + L->seg_inc_cap, L->svals, L->svals + L->seg_inc_cap,
                                (unsigned)n_inc, bits, st));
    if (need > L->sort_tmp_bytes) {
        if (L->sort_tmp) (void)hipFree(L->sort_tmp);
        L->sort_tmp = nullptr;
        L->sort_tmp_bytes = 0;
        PRG_HIP(hipMalloc(&L->sort_tmp, need + (need >> 2) + 256));
        L->sort_tmp_bytes = need + (need >> 2) + 256;
    }
    const unsigned g = (unsigned)prg::ceil_div(n_inc, kBlock);
    k_seg_keys<<<g, kBlock, 0, st>>>(L->pslot, L->ref_pos, first, n_inc, d1, L->skeys, L->svals, L->seg, 2 * (int64_t)L->size);
    size_t bytes = L->sort_tmp_bytes;
    PRG_TRY(prg::sort_pairs_u32(L->sort_tmp, &bytes, L->skeys, L->skeys + L->seg_inc_cap, L->svals,
                                L->svals + L->seg_inc_cap, (unsigned)n_inc, bits, st));
    k_seg_bounds<<<g, kBlock, 0, st>>>(L->skeys + L->seg_inc_cap, n_inc, L->size, L->seg);
    PRG_HIP(hipGetLastError());
    L->seg_valid = true;
    L->seg_first = first;
    return PRG_OK;
}

// Filter `ch` channels: in [n][ch] (device) -> out [n_out][ch] (device); only points >= first are splatted
// (callers pass first > 0 only when the skipped rows are known to be zero).
// defer_slice: stop before the slice step and leave (final value plane, alpha) in L->pend_vals / L->pend_alpha - FilterReg's
// point-to-point M-step slices inside its own terms kernel (k_fr_terms<true>); whoever else needs `out` runs k_slice then.
int lat_slice(Lattice* L, const float* vals, float alpha, int ch, int64_t n_out, unsigned seq_mask, float* out) {
    k_slice<<<(unsigned)prg::ceil_div(n_out * ch, kBlock), kBlock, 0, L->stream>>>(L->pslot, L->bary, vals, n_out, L->d + 1, ch,
                                                                                  alpha, seq_mask, out);
    PRG_HIP(hipGetLastError());
    return PRG_OK;
}

int lat_filter(Lattice* L, const float* in, int ch, int64_t first, int64_t n_out, unsigned seq_mask, float* out,
               bool defer_slice = false) {
    const int d1 = L->d + 1;
    hipStream_t st = L->stream;
    const int64_t plane = (int64_t)(L->size + 1) * ch;
    if (2 * plane > L->vals_elems) {
        const int64_t want = lat_grow(2 * plane, 2 * (L->n * d1 + 1) * ch);
        if (L->vals) (void)hipFree(L->vals);
        L->vals = nullptr;
        PRG_HIP(hipMalloc((void**)&L->vals, want * sizeof(float)));
        L->vals_elems = want;
    }
    float* a = L->vals;
    float* b = L->vals + plane;
    if (g_splat_mode == 2 && ch <= kSplatMaxCh) {
        PRG_TRY(lat_segments(L, first));
        const int* sinc = L->svals + L->seg_inc_cap;
        const int64_t n_inc = (L->n - first) * d1;
        if (n_inc * ch > L->terms_elems) {
            if (L->terms) (void)hipFree(L->terms);
            L->terms = nullptr;
            PRG_HIP(hipMalloc((void**)&L->terms, (size_t)n_inc * ch * sizeof(float)));
            L->terms_elems = n_inc * ch;
        }
        const int64_t max_long = n_inc / (kLongSeg + 1) + 1;
        if (max_long + 1 > L->long_cap) {
            if (L->long_list) (void)hipFree(L->long_list);
            L->long_list = nullptr;
            PRG_HIP(hipMalloc((void**)&L->long_list, (size_t)(max_long + 1) * sizeof(int)));
            L->long_cap = max_long + 1;
        }
        int* long_count = L->long_list + max_long;
        k_seg_gather<<<(unsigned)prg::ceil_div(n_inc, kBlock), kBlock, 0, st>>>(sinc, L->bary, in, n_inc, d1, ch, L->terms, n_inc,
                                                                               long_count);
        k_segchain_thread<<<(unsigned)prg::ceil_div(L->size, kBlock), kBlock, 0, st>>>(L->seg, L->terms, n_inc, ch, L->size, a, b,
                                                                                      L->long_list, long_count);
        k_segchain_wave<<<(unsigned)prg::ceil_div(max_long, kBlock / 64), kBlock, 0, st>>>(L->seg, L->terms, n_inc, ch, L->size, a,
                                                                                         L->long_list, long_count);
    } else if (g_splat_mode >= 1) {
        // fixed point: per-channel scale from the largest |value| (cached while the caller says the values have not changed)
        if (!L->fx_scale) {
            PRG_HIP(hipMalloc((void**)&L->fx_scale, (2 * 32 + 32) * sizeof(double)));
            L->fx_scale_key = nullptr;
        }
        unsigned* maxabs = reinterpret_cast<unsigned*>(L->fx_scale + 64);
        if (L->fx_scale_key != in || L->fx_scale_ch != ch || !L->fx_scale_static) {
            PRG_HIP(hipMemsetAsync(maxabs, 0, 32 * sizeof(unsigned), st));
            const int64_t total = (L->n - first) * ch;
            const unsigned gm = (unsigned)std::min<int64_t>(prg::ceil_div(total, kBlock), 2048);
            k_chan_maxabs<<<gm, kBlock, 0, st>>>(in, first, L->n, ch, maxabs);
            k_chan_scale<<<1, 32, 0, st>>>(maxabs, ch, (double)(L->n - first), L->fx_scale);
            L->fx_scale_key = in;
            L->fx_scale_ch = ch;
        }
        if (plane > L->fx_elems) {
            if (L->fx) (void)hipFree(L->fx);
            L->fx = nullptr;
            const int64_t want = lat_grow(plane, (L->n * d1 + 1) * ch);
            PRG_HIP(hipMalloc((void**)&L->fx, (size_t)want * sizeof(long long)));
            PRG_HIP(hipMemsetAsync(L->fx, 0, (size_t)want * sizeof(long long), st));  // from here on k_fix_to_float keeps it zero
            L->fx_elems = want;
        }
        static const bool wide_table = !(getenv("PRG_SPLAT_TABLE") && atoi(getenv("PRG_SPLAT_TABLE")) == 0);
        if (ch <= 5 && wide_table)
            k_splat_lds<true, 9, 5><<<(unsigned)prg::ceil_div(L->n - first, kSplatPts), kBlock, 0, st>>>(
                L->pslot, L->bary, in, first, L->n, d1, ch, a, L->fx, L->fx_scale);
        else if (ch <= kSplatMaxCh)
            k_splat_lds<true, 8, 8><<<(unsigned)prg::ceil_div(L->n - first, kSplatPts), kBlock, 0, st>>>(
                L->pslot, L->bary, in, first, L->n, d1, ch, a, L->fx, L->fx_scale);
        else
            k_splat<true><<<(unsigned)prg::ceil_div((L->n - first) * d1, kBlock), kBlock, 0, st>>>(
                L->pslot, L->bary, in, first, L->n, d1, ch, a, L->fx, L->fx_scale);
        k_fix_to_float<<<(unsigned)prg::ceil_div(plane, kBlock), kBlock, 0, st>>>(L->fx, plane, ch, L->fx_scale, a, b);
    } else {
        PRG_HIP(hipMemsetAsync(a, 0, 2 * plane * sizeof(float), st));
        if (ch <= kSplatMaxCh)
            k_splat_lds<false, 8, 8><<<(unsigned)prg::ceil_div(L->n - first, kSplatPts), kBlock, 0, st>>>(
                L->pslot, L->bary, in, first, L->n, d1, ch, a, nullptr, nullptr);
        else
            k_splat<false><<<(unsigned)prg::ceil_div((L->n - first) * d1, kBlock), kBlock, 0, st>>>(
                L->pslot, L->bary, in, first, L->n, d1, ch, a, nullptr, nullptr);
    }
    if (L->with_blur) {
        const int* nb1 = L->nb;
        const int* nb2 = L->nb + (int64_t)d1 * L->size;
        for (int j = 0; j < d1; ++j) {
            k_blur<<<(unsigned)prg::ceil_div((int64_t)L->size * ch, kBlock), kBlock, 0, st>>>(
                a, b, nb1 + (int64_t)j * L->size, nb2 + (int64_t)j * L->size, L->size, ch, seq_mask);
            float* t = a; a = b; b = t;
        }
    }
    const float alpha = 1.0f / (1.0f + powf(2.0f, (float)-L->d));
    PRG_HIP(hipGetLastError());
    if (defer_slice) {
        L->pend_vals = a;
        L->pend_alpha = alpha;
        return PRG_OK;
    }
    return lat_slice(L, a, alpha, ch, n_out, seq_mask, out);
}

}  // namespace

struct prg_ph {
    Lattice L;
};

// =============================================================================================
// FilterReg plan
// =============================================================================================
struct prg_filterreg {
    Lattice L;
    int64_t M = 0, N = 0;
    int D = 0;
    double* src = nullptr;   // [M][D] fp64 source
    double* tgt = nullptr;   // [N][D] fp64 target
    double* nrm = nullptr;   // [N][3] fp64 target normals (point-to-plane objective), optional
    int ch = 5;              // value channels: 1 | y(3) | |y|^2  (+ normal(3) when normals are set)
    double* ts = nullptr;    // [M][3] fp64 transformed source
    float* vin = nullptr;    // [M+N][ch] values (source rows zero)
    float* vout = nullptr;   // [M][ch] filtered m0, m1(3), m2 (, nx(3))
    double* state = nullptr; // [64]: 0..8 rot, 9..11 t, 12 sigma2, 13 q, 14 nonzero count, 15 sigma2_new
    double* part = nullptr;  // block partials
    int64_t part_blocks = 0;
    bool slice_pending = false;  // the last E-step stopped before its slice step (lat_filter defer_slice); see fr_flush_slice
    std::vector<int> tgt_order;  // Morton order of the target (kernel position -> caller's index); see prg_fr_set_target
    int* ref_pos = nullptr;      // [N] device: caller's index -> kernel position (the ordered splat walks the caller's order)
    int* src_perm = nullptr;     // [M] device: kernel position -> caller's index of the (Morton-sorted) source; null: caller's order
    bool have_src = false, have_tgt = false, have_estep = false;
    FrFeat prod;             // feature producer handed to the embedding kernels
    int last_blur = 1;       // with_blur of the previous E-step: which lattice the next one tries first
};

namespace {

constexpr int kFrComp = 32;  // 0 sw,1-3 sw*m,4-6 sw*t,7 sw2,8-10 sw2*m,11-13 sw2*t,14-22 sw2*m*t^T,23 q,24 s2num,25 m0m0,26 cnt

// values [M+N][ch]: source rows 0; target rows (1, y, |y|^2 [, normal])   (filterreg.py:92-105)
__global__ __launch_bounds__(kBlock) void k_fr_values(const double* __restrict__ tgt, const double* __restrict__ nrm,
                                                      int64_t m, int64_t n, int dim, int ch,
                                                      float* __restrict__ vin) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= m + n) return;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (i >= m) {
        const double* y = tgt + (i - m) * 4;  // (4 doubles per point)
        double s = 0.0;
        v[0] = 1.0f;
        for (int k = 0; k < dim; ++k) {
            v[1 + k] = (float)y[k];
            s += y[k] * y[k];
        }
        v[4] = (float)s;
        if (ch == 8)
            for (int k = 0; k < 3; ++k) v[5 + k] = (float)nrm[(i - m) * 3 + k];
    }
    for (int k = 0; k < ch; ++k) vin[i * ch + k] = v[k];
}

// per-point M-step terms (filterreg.py:163-182, 190-195) -> block partials [nblk][kFrComp]
// c = w/(1-w) * n/m * (2 sigma2 pi)^(dim/2)   (filterreg.py:164), evaluated on the device: no host round trip
__device__ __forceinline__ double fr_uniform_c(double wfac, int dim, double sigma2) {
    return wfac * pow(2.0 * sigma2 * M_PI, dim * 0.5);
}

__device__ void fr_finish_body(const double* __restrict__ part, int nblk, int dim, int update_sigma2, double min_sigma2,
                               double* __restrict__ state);

// SLICE: the lattice's slice step (permutohedral.cpp:521-528 / :586-592: barycentric interpolation of the D + 1 enclosing
// vertices, the reference's two arithmetic flavours per channel) is done HERE, per source point, instead of in a k_slice launch
// of its own: the five filtered values of a point go to `vout` (prg_fr_get_estep) and straight into the point's M-step terms -
// one launch, one dependent-launch gap and one 22 MB re-read less per EM iteration.  The Kabsch finish is a launch of its own
// again (k_fr_finish): folded into the last workgroup of this kernel (round 3) it doubled the kernel's registers (200 VGPRs:
// 2 waves per SIMD for the 512 workgroups that never run it) and cost 57 us where the two launches take 16 + 12.
template <bool SLICE>
__global__ __launch_bounds__(kBlock) void k_fr_terms(const float* __restrict__ vout_in, float* __restrict__ vout_w, int ch,
                                                     const int* __restrict__ offset, const float* __restrict__ bary,
                                                     const float* __restrict__ vals, float alpha,
                                                     const double* __restrict__ ts, int64_t m, int dim, double wfac,
                                                     const double* __restrict__ state, double* __restrict__ part) {
    __shared__ double sh[4][kFrComp];
    double a[kFrComp];
#pragma unroll
    for (int k = 0; k < kFrComp; ++k) a[k] = 0.0;
    const double sigma2 = state[12];
    const double c = fr_uniform_c(wfac, dim, sigma2);
    // grid-stride: a few hundred workgroups, each thread sums several points before the (32-component) reduction
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < m; i += (int64_t)gridDim.x * kBlock) {
        float f[5];
        if (SLICE) {  // (ch == 5, seq_mask 0x11: channels 0 and 4 follow seqCompute, 1..3 sseCompute - see k_slice)
            const int d1 = dim + 1;
#pragma unroll
            for (int k = 0; k < 5; ++k) f[k] = 0.f;
            for (int r = 0; r < d1; ++r) {
                const int o = offset[i * d1 + r] + 1;
                const float w = bary[i * d1 + r];
                const float* __restrict__ v = vals + (int64_t)o * 5;
                const float wa = __fmul_rn(w, alpha);
                f[0] = __fadd_rn(f[0], __fmul_rn(__fmul_rn(w, v[0]), alpha));
                f[1] = __fadd_rn(f[1], __fmul_rn(wa, v[1]));
                f[2] = __fadd_rn(f[2], __fmul_rn(wa, v[2]));
                f[3] = __fadd_rn(f[3], __fmul_rn(wa, v[3]));
                f[4] = __fadd_rn(f[4], __fmul_rn(__fmul_rn(w, v[4]), alpha));
            }
#pragma unroll
            for (int k = 0; k < 5; ++k) vout_w[i * 5 + k] = f[k];
        } else {
#pragma unroll
            for (int k = 0; k < 5; ++k) f[k] = vout_in[i * ch + k];
        }
        const float m0 = f[0];
        if (m0 != 0.f) {
            const double2 za = reinterpret_cast<const double2*>(ts)[2 * i], zb = reinterpret_cast<const double2*>(ts)[2 * i + 1];
            const double z[3] = {za.x, za.y, zb.x};  // (4 doubles per point)
            const float m1[3] = {f[1], f[2], f[3]};
            const float m2 = f[4];
            float tg[3];  // m1m0 = m1 / m0 in float32 (:172)
            for (int k = 0; k < 3; ++k) tg[k] = k < dim ? __fdiv_rn(m1[k], m0) : 0.f;
            const double m0m0 = (double)m0 / ((double)m0 + c);       // :173
            const double dr = sqrt(m0m0 / sigma2);                    // :174
            const double w = (double)(float)dr;                        // the Kabsch binding casts to float32
            const double w2 = w * w;
            double mod[3];
            for (int k = 0; k < 3; ++k) mod[k] = k < dim ? (double)(float)z[k] : 0.0;
            a[0] += w;
            a[7] += w2;
            double r2 = 0.0, zz = 0.0, zm1 = 0.0;
            for (int k = 0; k < 3; ++k) {
                a[1 + k] += w * mod[k];
                a[4 + k] += w * (double)tg[k];
                a[8 + k] += w2 * mod[k];
                a[11 + k] += w2 * (double)tg[k];
                for (int j = 0; j < 3; ++j) a[14 + 3 * k + j] += w2 * mod[k] * (double)tg[j];
                if (k < dim) {
                    const double rx = dr * (z[k] - (double)tg[k]);
                    r2 += rx * rx;
                    zz += z[k] * z[k];
                    zm1 += z[k] * (double)m1[k];
                }
            }
            a[23] += sqrt(r2);                                                          // q term, :181-182
            a[24] += ((double)m0 * zz - 2.0 * zm1 + (double)m2) / ((double)m0 + c);    // :192-194
            a[25] += m0m0;
            a[26] += 1.0;
        }
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < kFrComp; ++k) {
        const double s = wave_sum(a[k]);
        if (lane == 0) sh[wv][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < kFrComp)
        part[(int64_t)blockIdx.x * kFrComp + threadIdx.x] =
            sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
}

// point-to-plane M-step terms (filterreg.py:183-186 -> cc/point_to_plane.cc:6-32): per point with m0 != 0
//   v = t_source (float32), t = m1/m0, n = nx/m0, w = sqrt(m0m0/sigma2) (float32),
//   residual = n.(t - v), jac = [v x n, n]:  ata += w jac jac^T (21 upper entries), atb += w residual jac,
//   r_sum += w^2 residual^2.  comps: [0..20] ata, [21..26] atb, [27] r_sum, [28] sigma2 numerator, [29] m0m0, [30] count
__global__ __launch_bounds__(kBlock) void k_fr_terms_pt2pl(const float* __restrict__ vout, const double* __restrict__ ts,
                                                           int64_t m, double wfac, const double* __restrict__ state,
                                                           double* __restrict__ part) {
    __shared__ double sh[4][kFrComp];
    double a[kFrComp];
#pragma unroll
    for (int k = 0; k < kFrComp; ++k) a[k] = 0.0;
    const double sigma2 = state[12];
    const double c = fr_uniform_c(wfac, 3, sigma2);
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < m; i += (int64_t)gridDim.x * kBlock) {
        const float m0 = vout[i * 8];
        if (m0 != 0.f) {
            const double z[3] = {ts[i * 4], ts[i * 4 + 1], ts[i * 4 + 2]};  // (4 doubles per point)
            double v[3], t[3], n[3];
            double zz = 0.0, zm1 = 0.0;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float m1 = vout[i * 8 + 1 + k];
                v[k] = (double)(float)z[k];
                t[k] = (double)__fdiv_rn(m1, m0);
                n[k] = (double)__fdiv_rn(vout[i * 8 + 5 + k], m0);
                zz += z[k] * z[k];
                zm1 += z[k] * (double)m1;
            }
            const double m0m0 = (double)m0 / ((double)m0 + c);
            const double w = (double)(float)sqrt(m0m0 / sigma2);
            const double residual = n[0] * (t[0] - v[0]) + n[1] * (t[1] - v[1]) + n[2] * (t[2] - v[2]);
            const double jac[6] = {v[1] * n[2] - v[2] * n[1], v[2] * n[0] - v[0] * n[2], v[0] * n[1] - v[1] * n[0],
                                   n[0], n[1], n[2]};
            int idx = 0;
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int q = r; q < 6; ++q) a[idx++] += w * jac[r] * jac[q];
#pragma unroll
            for (int r = 0; r < 6; ++r) a[21 + r] += w * residual * jac[r];
            a[27] += w * w * residual * residual;
            a[28] += ((double)m0 * zz - 2.0 * zm1 + (double)vout[i * 8 + 4]) / ((double)m0 + c);
            a[29] += m0m0;
            a[30] += 1.0;
        }
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < kFrComp; ++k) {
        const double s = wave_sum(a[k]);
        if (lane == 0) sh[wv][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < kFrComp)
        part[(int64_t)blockIdx.x * kFrComp + threadIdx.x] =
            sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
}

// sum over the block partials of component c for this thread's slice (8 slices of 32 components): four independent
// accumulators keep four loads in flight - a single dependent chain of ~250 loads costs ~60 us on its own
__device__ __forceinline__ double sum_partials(const double* __restrict__ part, int nblk, int slice, int c) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int b = slice;
    for (; b + 24 < nblk; b += 32) {
        s0 += part[(int64_t)b * kFrComp + c];
        s1 += part[(int64_t)(b + 8) * kFrComp + c];
        s2 += part[(int64_t)(b + 16) * kFrComp + c];
        s3 += part[(int64_t)(b + 24) * kFrComp + c];
    }
    for (; b < nblk; b += 8) s0 += part[(int64_t)b * kFrComp + c];
    return (s0 + s1) + (s2 + s3);
}

// 6 x 6 SPD solve (the reference uses Eigen's LDLT on the upper triangle), twist -> Rodrigues rotation
// (se3_op.py:21-56), composition with the previous transform, optional sigma2 update.  One workgroup.
__global__ __launch_bounds__(kBlock) void k_fr_finish_pt2pl(const double* __restrict__ part, int nblk,
                                                            int update_sigma2, double min_sigma2,
                                                            double* __restrict__ state) {
    __shared__ double sh[8][32];
    __shared__ double mom[32];
    const int c = threadIdx.x & 31, slice = threadIdx.x >> 5;
    sh[slice][c] = sum_partials(part, nblk, slice, c);
    __syncthreads();
    if (threadIdx.x < 32) {
        double t = 0.0;
        for (int k = 0; k < 8; ++k) t += sh[k][threadIdx.x];
        mom[threadIdx.x] = t;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    state[14] = mom[30];
    if (mom[30] == 0.0) {
        state[13] = nan("");
        state[16] = 0.0;
        return;
    }
    state[16] = 1.0;
    // Cholesky solve of ata tw = atb (static indices)
    double A[6][6], bvec[6], tw[6];
    {
        int idx = 0;
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int q = r; q < 6; ++q) { A[r][q] = mom[idx]; A[q][r] = mom[idx]; ++idx; }
#pragma unroll
        for (int r = 0; r < 6; ++r) bvec[r] = mom[21 + r];
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double d = A[j][j];
#pragma unroll
        for (int k = 0; k < 6; ++k) d -= (k < j) ? A[j][k] * A[j][k] : 0.0;
        d = sqrt(fmax(d, 1e-300));
        A[j][j] = d;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            if (i <= j) continue;
            double t = A[i][j];
#pragma unroll
            for (int k = 0; k < 6; ++k) t -= (k < j) ? A[i][k] * A[j][k] : 0.0;
            A[i][j] = t / d;
        }
    }
    double yv[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double t = bvec[i];
#pragma unroll
        for (int k = 0; k < 6; ++k) t -= (k < i) ? A[i][k] * yv[k] : 0.0;
        yv[i] = t / A[i][i];
    }
#pragma unroll
    for (int ii = 0; ii < 6; ++ii) {
        const int i = 5 - ii;
        double t = yv[i];
#pragma unroll
        for (int k = 0; k < 6; ++k) t -= (k > i) ? A[k][i] * tw[k] : 0.0;
        tw[i] = t / A[i][i];
    }
    // twist -> (rotation, translation), se3_op.py:21-41
    double tr[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    const double twd = sqrt(tw[0] * tw[0] + tw[1] * tw[1] + tw[2] * tw[2]);
    if (twd != 0.0) {
        const double n0 = tw[0] / twd, n1 = tw[1] / twd, n2 = tw[2] / twd;
        const double cc = cos(twd), ss = sin(twd), oc = 1.0 - cc;
        tr[0][0] = cc + oc * n0 * n0;      tr[0][1] = oc * n0 * n1 - ss * n2; tr[0][2] = oc * n0 * n2 + ss * n1;
        tr[1][0] = oc * n1 * n0 + ss * n2; tr[1][1] = cc + oc * n1 * n1;      tr[1][2] = oc * n1 * n2 - ss * n0;
        tr[2][0] = oc * n2 * n0 - ss * n1; tr[2][1] = oc * n2 * n1 + ss * n0; tr[2][2] = cc + oc * n2 * n2;
    }
    // rot = tr @ rot_p ; t = t_p @ tr^T + tw[3:]   (se3_op.py:44-56)
    double rp[3][3], tp[3], rn[3][3], tn[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        tp[i] = state[9 + i];
#pragma unroll
        for (int j = 0; j < 3; ++j) rp[i][j] = state[3 * i + j];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        double tt = 0.0;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            double r = 0.0;
#pragma unroll
            for (int k = 0; k < 3; ++k) r += tr[i][k] * rp[k][j];
            rn[i][j] = r;
            tt += tr[i][j] * tp[j];
        }
        tn[i] = tt + tw[3 + i];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        state[9 + i] = tn[i];
#pragma unroll
        for (int j = 0; j < 3; ++j) state[3 * i + j] = rn[i][j];
    }
    state[13] = mom[27];  // q = r_sum
    state[18] = mom[27];  // ... of the last iteration that had anything to fit (a later all-zero one overwrites [13] with NaN)
    state[19] += 1.0;
    state[17] = state[12];
    state[15] = update_sigma2 ? mom[28] / (3.0 * mom[29]) : state[12];
    if (min_sigma2 >= 0.0) state[12] = fmax(state[15], min_sigma2);  // negative: do not advance (see k_fr_finish)
}

// weighted Kabsch from moments (cc/kabsch.cc:6-109): mom[0] sw, [1..3] sw*model, [4..6] sw*target, [7] sw2,
// [8..10] sw2*model, [11..13] sw2*target, [14..22] sw2*model*target^T.  Centroids use w, the covariance w^2.
__device__ void kabsch_from_moments(const double* mom, int dim, double (&dr)[3][3], double (&dt)[3]) {
    for (int i = 0; i < 3; ++i) {
        dt[i] = 0.0;
        for (int j = 0; j < 3; ++j) dr[i][j] = (i == j) ? 1.0 : 0.0;
    }
    const double sw = mom[0];
    if (sw != 0.0) {
        double mc[3], tc[3], H[3][3];
        for (int k = 0; k < 3; ++k) { mc[k] = mom[1 + k] / sw; tc[k] = mom[4 + k] / sw; }
        const double sw2 = mom[7];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j)
                H[i][j] = (mom[14 + 3 * i + j] - mc[i] * mom[11 + j] - mom[8 + i] * tc[j] + sw2 * mc[i] * tc[j]) / sw2;
        if (dim == 3) {
            double U[3][3], V[3][3], sv[3];
            prg::jacobi_svd(H, 3, U, V, sv);
            const double dd = prg::det3(U, 3) * prg::det3(V, 3);  // det(U V), kabsch.cc:48
            double c[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {  // the correction goes on the smallest singular value (static indices only)
                bool is_min = true;
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    if (j != k && (sv[j] < sv[k] || (sv[j] == sv[k] && j < k))) is_min = false;
                c[k] = is_min ? dd : 1.0;
            }
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    double r = 0.0;
#pragma unroll
                    for (int k = 0; k < 3; ++k) r += c[k] * V[i][k] * U[j][k];  // V diag U^T
                    dr[i][j] = r;
                }
        } else {
            const double ang = atan2(H[0][1] - H[1][0], H[0][0] + H[1][1]);  // kabsch.cc:98
            dr[0][0] = dr[1][1] = cos(ang);
            dr[0][1] = -sin(ang);
            dr[1][0] = sin(ang);
        }
        for (int i = 0; i < 3; ++i) {
            double r = 0.0;
            for (int k = 0; k < 3; ++k) r += dr[i][k] * mc[k];
            dt[i] = tc[i] - r;
        }
    }
}

// weighted Kabsch from the moments (cc/kabsch.cc:6-109) + composition (filterreg.py:180) - one workgroup
__global__ __launch_bounds__(kBlock) void k_fr_finish(const double* __restrict__ part, int nblk, int dim,
                                                      int update_sigma2, double min_sigma2,
                                                      double* __restrict__ state) {
    fr_finish_body(part, nblk, dim, update_sigma2, min_sigma2, state);
}

__device__ void fr_finish_body(const double* __restrict__ part, int nblk, int dim, int update_sigma2, double min_sigma2,
                               double* __restrict__ state) {
    __shared__ double sh[8][32];
    __shared__ double mom[32];
    const int c = threadIdx.x & 31, slice = threadIdx.x >> 5;
    sh[slice][c] = sum_partials(part, nblk, slice, c);
    __syncthreads();
    if (threadIdx.x < 32) {
        double t = 0.0;
        for (int k = 0; k < 8; ++k) t += sh[k][threadIdx.x];
        mom[threadIdx.x] = t;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    state[14] = mom[26];
    if (mom[26] == 0.0) {  // every m0 == 0: keep the previous transform, q = None (:167-168)
        state[13] = nan("");
        state[16] = 0.0;
        return;
    }
    state[16] = 1.0;
    double dr[3][3], dt[3];
    kabsch_from_moments(mom, dim, dr, dt);
    // rot = dr @ rot_p ; t = t_p @ dr^T + dt   (filterreg.py:180)
    double rp[3][3], tp[3], rn[3][3], tn[3];
    for (int i = 0; i < 3; ++i) {
        tp[i] = state[9 + i];
        for (int j = 0; j < 3; ++j) rp[i][j] = state[3 * i + j];
    }
    for (int i = 0; i < 3; ++i) {
        double tt = 0.0;
        for (int j = 0; j < 3; ++j) {
            double r = 0.0;
            for (int k = 0; k < 3; ++k) r += dr[i][k] * rp[k][j];
            rn[i][j] = r;
            tt += dr[i][j] * tp[j];
        }
        tn[i] = tt + dt[i];
    }
    for (int i = 0; i < 3; ++i) {
        state[9 + i] = tn[i];
        for (int j = 0; j < 3; ++j) state[3 * i + j] = rn[i][j];
    }
    state[13] = mom[23];
    state[18] = mom[23];  // q of the last iteration that had anything to fit (a later all-zero one overwrites [13] with NaN)
    state[19] += 1.0;     // ... and how many of those there were since prg_fr_set_state
    state[17] = state[12];                                               // sigma2 this step was computed with
    state[15] = update_sigma2 ? mom[24] / (3.0 * mom[25]) : state[12];  // :192-195 (3.0 hard-coded there)
    // self._sigma2 = max(res.sigma2, min_sigma2), :140 - for every legal min_sigma2 (0 included); a NEGATIVE value
    // is the explicit "leave the device sigma2 alone" request of the stand-alone M-step entry points
    if (min_sigma2 >= 0.0) state[12] = fmax(state[15], min_sigma2);
}

// caller-supplied E-step arrays -> the [m][ch] value layout and the [m][3] fp64 transformed source the M-step
// kernels read (prg_fr_mstep_from_arrays)
__global__ __launch_bounds__(kBlock) void k_fr_pack_estep(const double* __restrict__ tsrc, const float* __restrict__ m0,
                                                          const float* __restrict__ m1, const float* __restrict__ m2,
                                                          const float* __restrict__ nx, int64_t m, int dim, int ch,
                                                          double* __restrict__ ts, float* __restrict__ vout) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= m) return;
    for (int k = 0; k < 4; ++k) ts[i * 4 + k] = k < dim ? tsrc[i * dim + k] : 0.0;  // (4 doubles per point)
    float* o = vout + i * ch;
    o[0] = m0[i];
    for (int k = 0; k < 3; ++k) o[1 + k] = k < dim ? m1[i * dim + k] : 0.f;
    o[4] = m2 ? m2[i] : 0.f;
    if (ch == 8)
        for (int k = 0; k < 3; ++k) o[5 + k] = nx[i * 3 + k];
}

// out[perm[i]][0 .. ncols) = vout[i][col0 .. col0 + ncols): the plan's per-source-point values back in the caller's order
__global__ __launch_bounds__(kBlock) void k_fr_columns(const float* __restrict__ vout, int ch, int col0, int ncols,
                                                       const int* __restrict__ perm, int64_t m, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= m) return;
    const int64_t j = perm ? perm[i] : i;
    for (int k = 0; k < ncols; ++k) out[j * ncols + k] = vout[i * ch + col0 + k];
}

// stand-alone Kabsch: moments of (model, target, weight) float32 clouds -> partials [nblk][kFrComp]
__global__ __launch_bounds__(kBlock) void k_kabsch_terms(const float* __restrict__ model,
                                                         const float* __restrict__ target,
                                                         const float* __restrict__ weight, int64_t n, int dim,
                                                         double* __restrict__ part) {
    __shared__ double sh[4][kFrComp];
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    double a[kFrComp];
#pragma unroll
    for (int k = 0; k < kFrComp; ++k) a[k] = 0.0;
    if (i < n) {
        const double w = weight[i], w2 = w * w;
        double mod[3] = {0, 0, 0}, tg[3] = {0, 0, 0};
        for (int k = 0; k < dim; ++k) { mod[k] = model[i * dim + k]; tg[k] = target[i * dim + k]; }
        a[0] = w;
        a[7] = w2;
        for (int k = 0; k < 3; ++k) {
            a[1 + k] = w * mod[k];
            a[4 + k] = w * tg[k];
            a[8 + k] = w2 * mod[k];
            a[11 + k] = w2 * tg[k];
            for (int j = 0; j < 3; ++j) a[14 + 3 * k + j] = w2 * mod[k] * tg[j];
        }
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < kFrComp; ++k) {
        const double s = wave_sum(a[k]);
        if (lane == 0) sh[wv][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < kFrComp)
        part[(int64_t)blockIdx.x * kFrComp + threadIdx.x] =
            sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
}

__global__ __launch_bounds__(kBlock) void k_kabsch_finish(const double* __restrict__ part, int nblk, int dim,
                                                          double* __restrict__ out) {
    __shared__ double sh[8][32];
    __shared__ double mom[32];
    const int c = threadIdx.x & 31, slice = threadIdx.x >> 5;
    sh[slice][c] = sum_partials(part, nblk, slice, c);
    __syncthreads();
    if (threadIdx.x < 32) {
        double t = 0.0;
        for (int k = 0; k < 8; ++k) t += sh[k][threadIdx.x];
        mom[threadIdx.x] = t;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    double dr[3][3], dt[3];
    kabsch_from_moments(mom, dim, dr, dt);
    for (int i = 0; i < 3; ++i) {
        out[9 + i] = dt[i];
        for (int j = 0; j < 3; ++j) out[3 * i + j] = dr[i][j];
    }
}

}  // namespace

extern "C" {

int prg_lattice_set_splat_mode(int mode) {
    PRG_REQUIRE(mode >= 0 && mode <= 2, PRG_ERR_INVALID,
                "prg_lattice_set_splat_mode: mode must be 0 (float atomics), 1 (fixed-point atomics) or 2 (reference order)");
    g_splat_mode = mode;
    return PRG_OK;
}

// ---------------------------------------------------------------------------------------------
// stand-alone lattice (gaussian_filtering.Permutohedral)
// ---------------------------------------------------------------------------------------------
int prg_ph_create(prg_ph** out, int device, void* hip_stream) {
    PRG_REQUIRE(out != nullptr, PRG_ERR_INVALID, "prg_ph_create: out is NULL");
    int count = 0;
    PRG_HIP(hipGetDeviceCount(&count));
    PRG_REQUIRE(device >= 0 && device < count, PRG_ERR_INVALID, "prg_ph_create: device %d out of range", device);
    prg_ph* h = new (std::nothrow) prg_ph();
    PRG_REQUIRE(h != nullptr, PRG_ERR_NOMEM, "prg_ph_create: out of host memory");
    h->L.device = device;
    h->L.stream = (hipStream_t)hip_stream;
    *out = h;
    return PRG_OK;
}

int prg_ph_destroy(prg_ph* h) {
    if (!h) return PRG_OK;
    prg::DeviceGuard g(h->L.device);
    (void)hipStreamSynchronize(h->L.stream);
    lat_free(&h->L);
    delete h;
    return PRG_OK;
}

int prg_ph_init(prg_ph* h, const float* points_hd, int64_t n, int dim, int with_blur) {
    PRG_REQUIRE(h && points_hd, PRG_ERR_INVALID, "prg_ph_init: NULL argument");
    PRG_REQUIRE(n > 0 && dim >= 1 && dim <= kMaxDG, PRG_ERR_INVALID,
                "prg_ph_init: need n > 0 and feature dimension in [1, %d] (got n=%lld d=%d)", kMaxDG, (long long)n, dim);
    prg::DeviceGuard g(h->L.device);
    Lattice* L = &h->L;
    if (L->feat) (void)hipFree(L->feat);
    L->feat = nullptr;
    PRG_HIP(hipMalloc((void**)&L->feat, (size_t)n * dim * sizeof(float)));
    PRG_HIP(hipMemcpyAsync(L->feat, points_hd, (size_t)n * dim * sizeof(float), hipMemcpyDefault, L->stream));
    return lat_build(L, n, dim, with_blur ? 1 : 0);
}

int prg_ph_lattice_size(prg_ph* h, int* size) {
    PRG_REQUIRE(h && size, PRG_ERR_INVALID, "prg_ph_lattice_size: NULL argument");
    PRG_REQUIRE(h->L.n > 0, PRG_ERR_STATE, "prg_ph_lattice_size: lattice not initialised");
    *size = h->L.size;
    return PRG_OK;
}

int prg_ph_filter(prg_ph* h, const float* values_hd, int channels, float* out_hd) {
    PRG_REQUIRE(h && values_hd && out_hd, PRG_ERR_INVALID, "prg_ph_filter: NULL argument");
    PRG_REQUIRE(h->L.n > 0, PRG_ERR_STATE, "prg_ph_filter: lattice not initialised");
    PRG_REQUIRE(channels >= 1 && channels <= 32, PRG_ERR_INVALID, "prg_ph_filter: channels must be in [1, 32]");
    prg::DeviceGuard g(h->L.device);
    Lattice* L = &h->L;
    const size_t nb = (size_t)L->n * channels * sizeof(float);
    PRG_TRY(lat_ensure_io(L, 2 * nb));
    float* din = L->io;
    float* dout = L->io + (size_t)L->n * channels;
    PRG_HIP(hipMemcpyAsync(din, values_hd, nb, hipMemcpyDefault, L->stream));
    // <= 2 channels take the reference's seqCompute arithmetic, more take sseCompute (permutohedral.cpp:612-615)
    const unsigned seq_mask = channels <= 2 ? 0xFFFFFFFFu : 0u;
    PRG_TRY(lat_filter(L, din, channels, 0, L->n, seq_mask, dout));
    PRG_HIP(hipMemcpyAsync(out_hd, dout, nb, hipMemcpyDefault, L->stream));
    PRG_HIP(hipStreamSynchronize(L->stream));
    return PRG_OK;
}

// ---------------------------------------------------------------------------------------------
// FilterReg plan
// ---------------------------------------------------------------------------------------------
int prg_fr_create(prg_filterreg** out, int device, void* hip_stream) {
    PRG_REQUIRE(out != nullptr, PRG_ERR_INVALID, "prg_fr_create: out is NULL");
    int count = 0;
    PRG_HIP(hipGetDeviceCount(&count));
    PRG_REQUIRE(device >= 0 && device < count, PRG_ERR_INVALID, "prg_fr_create: device %d out of range", device);
    prg::DeviceGuard g(device);
    prg_filterreg* h = new (std::nothrow) prg_filterreg();
    PRG_REQUIRE(h != nullptr, PRG_ERR_NOMEM, "prg_fr_create: out of host memory");
    h->L.device = device;
    h->L.stream = (hipStream_t)hip_stream;
    hipError_t e = hipMalloc((void**)&h->state, 64 * sizeof(double));
    if (e != hipSuccess) {
        delete h;
        prg::set_error("prg_fr_create: hipMalloc failed: %s", hipGetErrorString(e));
        return PRG_ERR_HIP;
    }
    (void)hipMemsetAsync(h->state, 0, 64 * sizeof(double), h->L.stream);
    h->L.fx_scale_static = true;  // the plan's value array (target moments) only changes in fr_alloc
    *out = h;
    return PRG_OK;
}

int prg_fr_destroy(prg_filterreg* h) {
    if (!h) return PRG_OK;
    prg::DeviceGuard g(h->L.device);
    (void)hipStreamSynchronize(h->L.stream);
    lat_free(&h->L);
    for (void* p : {(void*)h->src, (void*)h->tgt, (void*)h->ts, (void*)h->vin, (void*)h->vout, (void*)h->state,
                    (void*)h->part, (void*)h->nrm, (void*)h->ref_pos, (void*)h->src_perm})
        if (p) (void)hipFree(p);
    delete h;
    return PRG_OK;
}

static int fr_alloc(prg_filterreg* h) {
    if (!(h->have_src && h->have_tgt)) return PRG_OK;
    const int64_t tot = h->M + h->N;
    for (void* p : {(void*)h->ts, (void*)h->vin, (void*)h->vout, (void*)h->part})
        if (p) (void)hipFree(p);
    h->ts = nullptr; h->vin = nullptr; h->vout = nullptr; h->part = nullptr;
    h->L.fx_scale_key = nullptr;  // new values: the fixed-point scales are recomputed by the next filter call
    PRG_HIP(hipMalloc((void**)&h->ts, (size_t)h->M * 4 * sizeof(double)));  // (x, y, z, 0) per point
    PRG_HIP(hipMalloc((void**)&h->vin, (size_t)tot * 8 * sizeof(float)));
    PRG_HIP(hipMalloc((void**)&h->vout, (size_t)h->M * 8 * sizeof(float)));
    h->part_blocks = std::min<int64_t>(prg::ceil_div(h->M, kBlock), 512);  // grid-stride M-step term kernels
    PRG_HIP(hipMalloc((void**)&h->part, (size_t)h->part_blocks * kFrComp * sizeof(double)));
    k_fr_values<<<(unsigned)prg::ceil_div(tot, kBlock), kBlock, 0, h->L.stream>>>(h->tgt, h->nrm, h->M, h->N, h->D,
                                                                                  h->ch, h->vin);
    PRG_HIP(hipGetLastError());
    return PRG_OK;
}

int prg_fr_set_source(prg_filterreg* h, const double* source_hd, int64_t m, int dim) {
    PRG_REQUIRE(h && source_hd, PRG_ERR_INVALID, "prg_fr_set_source: NULL argument");
    PRG_REQUIRE(m > 0 && (dim == 2 || dim == 3), PRG_ERR_INVALID, "prg_fr_set_source: need m > 0, dim in {2,3}");
    PRG_REQUIRE(!h->have_tgt || h->D == dim, PRG_ERR_INVALID, "prg_fr_set_source: dim mismatch with target");
    prg::DeviceGuard g(h->L.device);
    PRG_HIP(hipStreamSynchronize(h->L.stream));
    if (h->src) (void)hipFree(h->src);
    h->src = nullptr;
    PRG_HIP(hipMalloc((void**)&h->src, (size_t)m * 4 * sizeof(double)));  // (x, y, z, 0) per point: 16-byte accesses in k_embed
    if (h->src_perm) (void)hipFree(h->src_perm);
    h->src_perm = nullptr;
    static const bool sort_source = getenv("PRG_FR_SOURCE_ORDER") == nullptr;  // (set: keep the caller's order, as in round 2)
    {
        // The source is stored in Morton order like the target: with the lattice's vertices created by a sample spread over
        // both clouds (every 16th point, lat_build), the other points only LOOK vertices up, and neighbouring lanes of a
        // sorted cloud look up the same few table lines.  Every per-point output crosses the ABI through src_perm.
        std::vector<double> host((size_t)m * dim), padded((size_t)m * 4, 0.0);
        PRG_HIP(hipMemcpy(host.data(), source_hd, host.size() * sizeof(double), hipMemcpyDefault));
        std::vector<int> order;
        if (sort_source && m >= 4096) order = prg::morton_order(host.data(), m, dim);
        for (int64_t i = 0; i < m; ++i) {
            const int64_t j = order.empty() ? i : order[(size_t)i];
            for (int k = 0; k < dim; ++k) padded[(size_t)i * 4 + k] = host[(size_t)j * dim + k];
        }
        if (!order.empty()) {
            PRG_HIP(hipMalloc((void**)&h->src_perm, (size_t)m * sizeof(int)));
            PRG_HIP(hipMemcpyAsync(h->src_perm, order.data(), (size_t)m * sizeof(int), hipMemcpyHostToDevice, h->L.stream));
        }
        PRG_HIP(hipMemcpyAsync(h->src, padded.data(), padded.size() * sizeof(double), hipMemcpyHostToDevice, h->L.stream));
        PRG_HIP(hipStreamSynchronize(h->L.stream));
    }
    h->M = m;
    h->D = dim;
    h->have_src = true;
    h->have_estep = false;
    return fr_alloc(h);
}

int prg_fr_set_target(prg_filterreg* h, const double* target_hd, int64_t n, int dim) {
    PRG_REQUIRE(h && target_hd, PRG_ERR_INVALID, "prg_fr_set_target: NULL argument");
    PRG_REQUIRE(n > 0 && (dim == 2 || dim == 3), PRG_ERR_INVALID, "prg_fr_set_target: need n > 0, dim in {2,3}");
    PRG_REQUIRE(!h->have_src || h->D == dim, PRG_ERR_INVALID, "prg_fr_set_target: dim mismatch with source");
    prg::DeviceGuard g(h->L.device);
    PRG_HIP(hipStreamSynchronize(h->L.stream));
    if (h->tgt) (void)hipFree(h->tgt);
    h->tgt = nullptr;
    PRG_HIP(hipMalloc((void**)&h->tgt, (size_t)n * 4 * sizeof(double)));  // (x, y, z, 0) per point
    // The target is stored in Morton order: the splat works on 2048 consecutive target points per workgroup, and
    // spatially close points share lattice vertices, so the workgroup-private LDS table absorbs most updates and
    // the flush touches few global vertices.  Every E-step output is per SOURCE point, so no order leaks out.
    std::vector<double> host((size_t)n * dim), sorted((size_t)n * 4, 0.0);
    PRG_HIP(hipMemcpy(host.data(), target_hd, host.size() * sizeof(double), hipMemcpyDefault));
    h->tgt_order = prg::morton_order(host.data(), n, dim);
    for (int64_t i = 0; i < n; ++i)
        for (int k = 0; k < dim; ++k) sorted[(size_t)i * 4 + k] = host[(size_t)h->tgt_order[i] * dim + k];
    PRG_HIP(hipMemcpyAsync(h->tgt, sorted.data(), sorted.size() * sizeof(double), hipMemcpyHostToDevice, h->L.stream));
    // ... and the splat still adds every vertex' terms up in the CALLER's point order (the reference's): caller index -> kernel position
    std::vector<int> inv((size_t)n);
    for (int64_t i = 0; i < n; ++i) inv[(size_t)h->tgt_order[i]] = (int)i;
    if (h->ref_pos) (void)hipFree(h->ref_pos);
    h->ref_pos = nullptr;
    PRG_HIP(hipMalloc((void**)&h->ref_pos, (size_t)n * sizeof(int)));
    PRG_HIP(hipMemcpyAsync(h->ref_pos, inv.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, h->L.stream));
    PRG_HIP(hipStreamSynchronize(h->L.stream));
    h->N = n;
    h->D = dim;
    h->have_tgt = true;
    h->have_estep = false;
    if (h->nrm) (void)hipFree(h->nrm);  // normals belong to the previous target
    h->nrm = nullptr;
    h->ch = 5;
    return fr_alloc(h);
}

int prg_fr_set_state(prg_filterreg* h, const double* rot9, const double* t3, double sigma2) {
    PRG_REQUIRE(h && rot9 && t3, PRG_ERR_INVALID, "prg_fr_set_state: NULL argument");
    PRG_REQUIRE(sigma2 > 0.0, PRG_ERR_INVALID, "prg_fr_set_state: sigma2 must be > 0 (got %g)", sigma2);
    prg::DeviceGuard g(h->L.device);
    double buf[20];
    for (int i = 0; i < 9; ++i) buf[i] = rot9[i];
    for (int i = 0; i < 3; ++i) buf[9 + i] = t3[i];
    buf[12] = sigma2;
    for (int i = 13; i < 20; ++i) buf[i] = 0.0;
    buf[15] = sigma2;
    PRG_HIP(hipMemcpyAsync(h->state, buf, sizeof(buf), hipMemcpyHostToDevice, h->L.stream));
    PRG_HIP(hipStreamSynchronize(h->L.stream));
    h->last_blur = 1;  // a (re)started registration begins with a large sigma2: try the blurred lattice first
    return PRG_OK;
}

int prg_fr_estep(prg_filterreg* h, double alpha, int* lattice_size, int* with_blur) {
    PRG_REQUIRE(h && h->have_src && h->have_tgt, PRG_ERR_STATE, "prg_fr_estep: clouds not set");
    prg::DeviceGuard g(h->L.device);
    const int64_t tot = h->M + h->N;
    // features are produced inside the embedding kernels (transform, division by sigma, float32 cast)
    h->prod = FrFeat{h->src, h->tgt, h->state, h->ts, h->M, h->D};
    h->L.prod = &h->prod;
    h->L.ref_pos = h->ref_pos;
    int blur = 1;
    // filterreg.py:90-91: the blurred lattice is used only if it has at most N * alpha vertices.  The answer is exact
    // every time; what changes with the previous E-step's answer is which lattice is built FIRST:
    //   previous blurred     -> the blurred lattice, whole (one synchronisation); if it came out too large, the other one;
    //   previous not blurred -> the non-blurred lattice, whole, while a 1/16 subset hashed with the blur scaling proves
    //                           that the blurred one is still too large (same synchronisation); if the proof fails, the
    //                           staged decision of round 1 (rare: sigma2 would have to grow again).
    const double thr = (double)h->N * alpha;
    const int64_t decide = thr >= 0.0 && thr < 2.0e9 ? (int64_t)floor(thr) : -1;
    bool done = false;
    if (h->last_blur == 0 && decide >= 0 && tot >= 4096) {
        PRG_TRY(lat_side_stage(&h->L, tot, h->D));
        PRG_TRY(lat_build(&h->L, tot, h->D, 0));
        if (h->L.side_overflow == 0 && h->L.side_size > decide) {
            blur = 0;
            done = true;
        }
    }
    if (!done) {
        PRG_TRY(lat_build(&h->L, tot, h->D, 1, h->last_blur == 1 ? -1 : decide));
        if (!h->L.built || (double)h->L.size > thr) {
            blur = 0;
            PRG_TRY(lat_build(&h->L, tot, h->D, 0));
        }
    }
    h->last_blur = blur;
    // one fused 5-channel pass: channels 0 (m0) and 4 (m2) are single-channel filters in the reference
    // (seqCompute arithmetic), channels 1..3 (m1) its 3-channel filter (sseCompute arithmetic)
    // (point-to-point plans, ch == 5: the slice step waits for its consumer - the M-step slices inside its terms kernel)
    h->slice_pending = h->ch == 5;
    PRG_TRY(lat_filter(&h->L, h->vin, h->ch, h->M, h->M, 0x11u, h->vout, h->slice_pending));  // normals (ch 5..7): 3-channel filter
    if (lattice_size) *lattice_size = h->L.size;
    if (with_blur) *with_blur = blur;
    h->have_estep = true;
    return PRG_OK;
}

// the E-step's deferred slice, for every consumer of `vout` other than the point-to-point M-step
static int fr_flush_slice(prg_filterreg* h) {
    if (!h->slice_pending) return PRG_OK;
    h->slice_pending = false;
    return lat_slice(&h->L, h->L.pend_vals, h->L.pend_alpha, h->ch, h->M, 0x11u, h->vout);
}

// columns [col0, col0 + ncols) of the filtered values, one row per source point in the CALLER's order -> out_hd
static int fr_fetch_columns(prg_filterreg* h, int col0, int ncols, float* out_hd) {
    hipStream_t st = h->L.stream;
    PRG_TRY(fr_flush_slice(h));
    PRG_TRY(lat_ensure_io(&h->L, (size_t)h->M * ncols * sizeof(float)));
    k_fr_columns<<<(unsigned)prg::ceil_div(h->M, kBlock), kBlock, 0, st>>>(h->vout, h->ch, col0, ncols, h->src_perm, h->M, h->L.io);
    PRG_HIP(hipGetLastError());
    PRG_HIP(hipMemcpyAsync(out_hd, h->L.io, (size_t)h->M * ncols * sizeof(float), hipMemcpyDefault, st));
    PRG_HIP(hipStreamSynchronize(st));
    return PRG_OK;
}

int prg_fr_get_estep(prg_filterreg* h, float* m0_hd, float* m1_hd, float* m2_hd) {
    PRG_REQUIRE(h && h->have_estep, PRG_ERR_STATE, "prg_fr_get_estep: no E-step has been run");
    prg::DeviceGuard g(h->L.device);
    if (m0_hd) PRG_TRY(fr_fetch_columns(h, 0, 1, m0_hd));
    if (m1_hd) PRG_TRY(fr_fetch_columns(h, 1, h->D, m1_hd));
    if (m2_hd) PRG_TRY(fr_fetch_columns(h, 4, 1, m2_hd));
    return PRG_OK;
}

static int fr_read_state(prg_filterreg* h, double* out_host, int count) {
    prg::DeviceGuard g(h->L.device);
    hipStream_t st = h->L.stream;
    if (!h->L.pinned) PRG_HIP(hipHostMalloc((void**)&h->L.pinned, 64 * sizeof(double), hipHostMallocDefault));
    PRG_HIP(hipMemcpyAsync(h->L.pinned, h->state, count * sizeof(double), hipMemcpyDeviceToHost, st));
    PRG_HIP(hipStreamSynchronize(st));
    for (int i = 0; i < count; ++i) out_host[i] = h->L.pinned[i];
    return PRG_OK;
}

int prg_fr_get_state(prg_filterreg* h, double* out_host) {
    PRG_REQUIRE(h && out_host, PRG_ERR_INVALID, "prg_fr_get_state: NULL argument");
    return fr_read_state(h, out_host, 20);
}

int prg_fr_mstep(prg_filterreg* h, double w, int update_sigma2, double min_sigma2, double* out_host) {
    PRG_REQUIRE(h && h->have_estep, PRG_ERR_STATE, "prg_fr_mstep: run prg_fr_estep first");
    PRG_REQUIRE(w >= 0.0 && w < 1.0, PRG_ERR_INVALID, "prg_fr_mstep: w must be in [0, 1) (got %g)", w);
    prg::DeviceGuard g(h->L.device);
    hipStream_t st = h->L.stream;
    const double wfac = w / (1.0 - w) * (double)h->N / (double)h->M;
    const int nblk = (int)h->part_blocks;
    if (h->slice_pending) {  // slice + terms in one launch (the values still go to vout for prg_fr_get_estep)
        h->slice_pending = false;
        k_fr_terms<true><<<nblk, kBlock, 0, st>>>(nullptr, h->vout, 5, h->L.pslot, h->L.bary, h->L.pend_vals, h->L.pend_alpha, h->ts,
                                                  h->M, h->D, wfac, h->state, h->part);
    } else {
        k_fr_terms<false><<<nblk, kBlock, 0, st>>>(h->vout, nullptr, h->ch, nullptr, nullptr, nullptr, 0.f, h->ts, h->M, h->D, wfac,
                                                   h->state, h->part);
    }
    k_fr_finish<<<1, kBlock, 0, st>>>(h->part, nblk, h->D, update_sigma2, min_sigma2, h->state);
    PRG_HIP(hipGetLastError());
    return out_host ? fr_read_state(h, out_host, 18) : PRG_OK;  // NULL: nothing is read back, the stream keeps running
}

int prg_fr_set_target_normals(prg_filterreg* h, const double* normals_hd) {
    PRG_REQUIRE(h && h->have_tgt, PRG_ERR_STATE, "prg_fr_set_target_normals: target not set");
    PRG_REQUIRE(h->D == 3 || !normals_hd, PRG_ERR_INVALID, "prg_fr_set_target_normals: point-to-plane needs 3-D clouds");
    prg::DeviceGuard g(h->L.device);
    PRG_HIP(hipStreamSynchronize(h->L.stream));
    if (h->nrm) (void)hipFree(h->nrm);
    h->nrm = nullptr;
    h->ch = 5;
    if (normals_hd) {
        PRG_HIP(hipMalloc((void**)&h->nrm, (size_t)h->N * 3 * sizeof(double)));
        std::vector<double> host((size_t)h->N * 3), sorted((size_t)h->N * 3);  // same order as the stored target
        PRG_HIP(hipMemcpy(host.data(), normals_hd, host.size() * sizeof(double), hipMemcpyDefault));
        for (int64_t i = 0; i < h->N; ++i)
            for (int k = 0; k < 3; ++k) sorted[(size_t)i * 3 + k] = host[(size_t)h->tgt_order[i] * 3 + k];
        PRG_HIP(hipMemcpyAsync(h->nrm, sorted.data(), (size_t)h->N * 3 * sizeof(double), hipMemcpyHostToDevice, h->L.stream));
        PRG_HIP(hipStreamSynchronize(h->L.stream));
        h->ch = 8;
    }
    h->have_estep = false;
    return fr_alloc(h);
}

int prg_fr_get_nx(prg_filterreg* h, float* nx_hd) {
    PRG_REQUIRE(h && h->have_estep && h->ch == 8 && nx_hd, PRG_ERR_STATE,
                "prg_fr_get_nx: needs target normals and an E-step");
    prg::DeviceGuard g(h->L.device);
    return fr_fetch_columns(h, 5, 3, nx_hd);
}

int prg_fr_mstep_pt2pl(prg_filterreg* h, double w, int update_sigma2, double min_sigma2, double* out_host) {
    PRG_REQUIRE(h && h->have_estep, PRG_ERR_STATE, "prg_fr_mstep_pt2pl: run prg_fr_estep first");
    PRG_REQUIRE(h->ch == 8, PRG_ERR_STATE, "prg_fr_mstep_pt2pl: target normals have not been set");
    PRG_REQUIRE(w >= 0.0 && w < 1.0, PRG_ERR_INVALID, "prg_fr_mstep_pt2pl: w must be in [0, 1) (got %g)", w);
    prg::DeviceGuard g(h->L.device);
    hipStream_t st = h->L.stream;
    const double wfac = w / (1.0 - w) * (double)h->N / (double)h->M;
    const int nblk = (int)h->part_blocks;
    PRG_TRY(fr_flush_slice(h));
    k_fr_terms_pt2pl<<<nblk, kBlock, 0, st>>>(h->vout, h->ts, h->M, wfac, h->state, h->part);
    k_fr_finish_pt2pl<<<1, kBlock, 0, st>>>(h->part, nblk, update_sigma2, min_sigma2, h->state);
    PRG_HIP(hipGetLastError());
    return out_host ? fr_read_state(h, out_host, 18) : PRG_OK;
}

// RigidFilterReg._maximization_step on explicit arrays (filterreg.py:158-196) - same kernels as prg_fr_mstep /
// prg_fr_mstep_pt2pl, fed from the caller's buffers instead of a plan's last E-step
int prg_fr_mstep_from_arrays(int device, void* hip_stream, const double* t_source_hd, int64_t m, int dim,
                             int64_t n_target, const float* m0_hd, const float* m1_hd, const float* m2_hd,
                             const float* nx_hd, const double* rot9, const double* t3, double sigma2, double w,
                             double* out_host) {
    PRG_REQUIRE(t_source_hd && m0_hd && m1_hd && rot9 && t3 && out_host, PRG_ERR_INVALID,
                "prg_fr_mstep_from_arrays: NULL argument");
    PRG_REQUIRE(m > 0 && n_target > 0 && (dim == 2 || dim == 3), PRG_ERR_INVALID,
                "prg_fr_mstep_from_arrays: need m > 0, n_target > 0 and dim in {2,3}");  // "dim must be 2 or 3", :161
    PRG_REQUIRE(!nx_hd || dim == 3, PRG_ERR_INVALID, "prg_fr_mstep_from_arrays: point-to-plane needs 3-D clouds");
    PRG_REQUIRE(w >= 0.0 && w < 1.0, PRG_ERR_INVALID, "prg_fr_mstep_from_arrays: w must be in [0, 1) (got %g)", w);
    PRG_REQUIRE(sigma2 > 0.0, PRG_ERR_INVALID, "prg_fr_mstep_from_arrays: sigma2 must be > 0 (got %g)", sigma2);
    prg::DeviceGuard g(device);
    PRG_REQUIRE(g.ok, PRG_ERR_HIP, "prg_fr_mstep_from_arrays: hipSetDevice(%d) failed", device);
    hipStream_t st = (hipStream_t)hip_stream;
    const int ch = nx_hd ? 8 : 5;
    const int nblk = (int)std::min<int64_t>(prg::ceil_div(m, kBlock), 512), npack = (int)prg::ceil_div(m, kBlock);
    struct Tmp {
        void* p = nullptr;
        ~Tmp() { if (p) (void)hipFree(p); }
    } b_in, b_ts, b_v, b_part, b_state;
    // staging: t_source | m0 | m1 | m2 | nx
    const size_t o_ts = 0, o_m0 = o_ts + (size_t)m * dim * sizeof(double), o_m1 = o_m0 + (size_t)m * sizeof(float),
                 o_m2 = o_m1 + (size_t)m * dim * sizeof(float), o_nx = o_m2 + (size_t)m * sizeof(float),
                 total = o_nx + (size_t)m * 3 * sizeof(float);
    PRG_HIP(hipMalloc(&b_in.p, total));
    PRG_HIP(hipMalloc(&b_ts.p, (size_t)m * 4 * sizeof(double)));  // (x, y, z, 0) per point, as the terms kernels read it
    PRG_HIP(hipMalloc(&b_v.p, (size_t)m * ch * sizeof(float)));
    PRG_HIP(hipMalloc(&b_part.p, (size_t)nblk * kFrComp * sizeof(double)));
    PRG_HIP(hipMalloc(&b_state.p, 64 * sizeof(double)));
    char* in = (char*)b_in.p;
    PRG_HIP(hipMemcpyAsync(in + o_ts, t_source_hd, (size_t)m * dim * sizeof(double), hipMemcpyDefault, st));
    PRG_HIP(hipMemcpyAsync(in + o_m0, m0_hd, (size_t)m * sizeof(float), hipMemcpyDefault, st));
    PRG_HIP(hipMemcpyAsync(in + o_m1, m1_hd, (size_t)m * dim * sizeof(float), hipMemcpyDefault, st));
    if (m2_hd) PRG_HIP(hipMemcpyAsync(in + o_m2, m2_hd, (size_t)m * sizeof(float), hipMemcpyDefault, st));
    if (nx_hd) PRG_HIP(hipMemcpyAsync(in + o_nx, nx_hd, (size_t)m * 3 * sizeof(float), hipMemcpyDefault, st));
    double host_state[64] = {0.0};
    for (int i = 0; i < 9; ++i) host_state[i] = rot9[i];
    for (int i = 0; i < 3; ++i) host_state[9 + i] = t3[i];
    host_state[12] = sigma2;
    PRG_HIP(hipMemcpyAsync(b_state.p, host_state, sizeof(host_state), hipMemcpyHostToDevice, st));
    k_fr_pack_estep<<<npack, kBlock, 0, st>>>((const double*)(in + o_ts), (const float*)(in + o_m0),
                                             (const float*)(in + o_m1), m2_hd ? (const float*)(in + o_m2) : nullptr,
                                             nx_hd ? (const float*)(in + o_nx) : nullptr, m, dim, ch, (double*)b_ts.p,
                                             (float*)b_v.p);
    const double wfac = w / (1.0 - w) * (double)n_target / (double)m;
    const int update_sigma2 = m2_hd ? 1 : 0;
    if (nx_hd) {
        k_fr_terms_pt2pl<<<nblk, kBlock, 0, st>>>((const float*)b_v.p, (const double*)b_ts.p, m, wfac,
                                                  (const double*)b_state.p, (double*)b_part.p);
        k_fr_finish_pt2pl<<<1, kBlock, 0, st>>>((const double*)b_part.p, nblk, update_sigma2, -1.0, (double*)b_state.p);
    } else {
        k_fr_terms<false><<<nblk, kBlock, 0, st>>>((const float*)b_v.p, nullptr, ch, nullptr, nullptr, nullptr, 0.f,
                                                   (const double*)b_ts.p, m, dim, wfac, (const double*)b_state.p, (double*)b_part.p);
        k_fr_finish<<<1, kBlock, 0, st>>>((const double*)b_part.p, nblk, dim, update_sigma2, -1.0, (double*)b_state.p);
    }
    PRG_HIP(hipGetLastError());
    PRG_HIP(hipMemcpyAsync(host_state, b_state.p, 18 * sizeof(double), hipMemcpyDeviceToHost, st));
    PRG_HIP(hipStreamSynchronize(st));
    for (int i = 0; i < 18; ++i) out_host[i] = host_state[i];
    return PRG_OK;
}

// stand-alone weighted Kabsch (probreg/cc/kabsch.cc:6-109 behind _kabsch.kabsch / kabsch2d)
int prg_kabsch_weighted(int device, void* hip_stream, const float* model_hd, const float* target_hd,
                        const float* weight_hd, int64_t n, int dim, double* rot_host, double* t_host) {
    PRG_REQUIRE(model_hd && target_hd && weight_hd && rot_host && t_host, PRG_ERR_INVALID,
                "prg_kabsch_weighted: NULL argument");
    PRG_REQUIRE(n > 0 && (dim == 2 || dim == 3), PRG_ERR_INVALID, "prg_kabsch_weighted: need n > 0, dim in {2,3}");
    prg::DeviceGuard g(device);
    PRG_REQUIRE(g.ok, PRG_ERR_HIP, "prg_kabsch_weighted: hipSetDevice(%d) failed", device);
    hipStream_t st = (hipStream_t)hip_stream;
    const int nblk = (int)prg::ceil_div(n, kBlock);
    struct Tmp {
        void* p = nullptr;
        ~Tmp() { if (p) (void)hipFree(p); }
    } bm, bt, bw, bp, bo;
    PRG_HIP(hipMalloc(&bm.p, (size_t)n * dim * sizeof(float)));
    PRG_HIP(hipMalloc(&bt.p, (size_t)n * dim * sizeof(float)));
    PRG_HIP(hipMalloc(&bw.p, (size_t)n * sizeof(float)));
    PRG_HIP(hipMalloc(&bp.p, (size_t)nblk * kFrComp * sizeof(double)));
    PRG_HIP(hipMalloc(&bo.p, 12 * sizeof(double)));
    PRG_HIP(hipMemcpyAsync(bm.p, model_hd, (size_t)n * dim * sizeof(float), hipMemcpyDefault, st));
    PRG_HIP(hipMemcpyAsync(bt.p, target_hd, (size_t)n * dim * sizeof(float), hipMemcpyDefault, st));
    PRG_HIP(hipMemcpyAsync(bw.p, weight_hd, (size_t)n * sizeof(float), hipMemcpyDefault, st));
    k_kabsch_terms<<<nblk, kBlock, 0, st>>>((const float*)bm.p, (const float*)bt.p, (const float*)bw.p, n, dim,
                                            (double*)bp.p);
    k_kabsch_finish<<<1, kBlock, 0, st>>>((const double*)bp.p, nblk, dim, (double*)bo.p);
    PRG_HIP(hipGetLastError());
    double out[12];
    PRG_HIP(hipMemcpyAsync(out, bo.p, sizeof(out), hipMemcpyDeviceToHost, st));
    PRG_HIP(hipStreamSynchronize(st));
    for (int i = 0; i < dim; ++i) {
        t_host[i] = out[9 + i];
        for (int j = 0; j < dim; ++j) rot_host[i * dim + j] = out[3 * i + j];
    }
    return PRG_OK;
}

}  // extern "C"
