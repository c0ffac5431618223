// Launchers of the two E-step pair sweeps (see cpd.hip header comment).
//   packed : explicit float2 (v_pk_*_f32) arithmetic, R = 2 or 4 points per lane
//   scalar : one fp32 op per pair (translation unit built with -fno-slp-vectorize), for A/B
//            comparison on hardware; selected by negative R in prg_cpd_set_tuning.
#pragma once
#include "cpd_plan.h"

namespace prg {
constexpr int kSweepBlock = 256;
constexpr int kColChunk = 8;   // source points consumed per column-pass loop trip
constexpr int kRowChunk = 4;   // target points consumed per row-pass loop trip
constexpr int kOverRead = 8;   // points read past the end of the last segment (software prefetch)

// Exponent offset of a column's running sum: the float just BELOW -kk * dmin (kk < 0, dmin >= 0).  Rounded down so
// that the largest term, exp2(kk * dmin + off), has an exponent <= 0 whatever dmin is - a round-to-nearest offset
// leaves a residual of +-ulp(off)/2, which for a pad-only segment (dmin ~ 3e36) is 1e29 and overflows exp2.  An
// overflowing product lands on FLT_MAX the same way.  The column pass kernels and k_colfinal both call this, so the
// offset is reproduced bit for bit and the bookkeeping sum_true = s * 2^-off is exact.
__device__ __forceinline__ float col_offset(float kk, float dmin) {
    const float p = -(kk * dmin);
    return p > 0.f ? __uint_as_float(__float_as_uint(p) - 1u) : 0.f;
}

// ---- matrix-core sweeps (cpd_sweeps_mfma.hip) ----
constexpr int kMfmaOwn = 8;                       // tiles of 16 points a wave owns (128 rows / columns)
constexpr int kMfmaWgPoints = 4 * 16 * kMfmaOwn;  // points a workgroup owns and shifts to one origin (512)
// Exponent offset of a column's sum in the matrix-core column pass, from what is known BEFORE the sweep: cm = min d^2 of
// the previous E-step, mo = largest source displacement since.  This E-step's minimum lies in
// [max(sqrt(cm) - mo, 0)^2, (sqrt(cm) + mo)^2] = [lo, hi]; with g = |kk| (hi - lo) the offset |kk| lo + max(g - 80, 0) keeps
// the largest term's exponent inside [-80, max(g - 80, 0)]: no overflow and no loss to flushed terms while g < 180 (the
// host checks the widest bracket).  Explicitly rounded operations: the column pass and k_colfinal must agree bit for bit.
__device__ __forceinline__ float col_seed_offset(float kk, float cm, float mo) {
    const float r = __fsqrt_rn(cm);
    const float rl = fmaxf(__fsub_rn(r, mo), 0.f), rh = __fadd_rn(r, mo);
    const float lo = __fmul_rn(rl, rl), hi = __fmul_rn(rh, rh);
    const float nk = -kk;
    const float g = __fmul_rn(nk, __fsub_rn(hi, lo));
    return __fadd_rn(__fmul_rn(nk, lo), fmaxf(__fsub_rn(g, 80.f), 0.f));
}
// first: no seeds from a previous E-step (offsets 0, no culling); fine: every wave also tests its own 128 points against
// the groups of 32 streamed points of each chunk (pays once sigma2 is small enough for some of them to be skipped)
// guard (may be null): device copy of the E-step's engine decision - the launch was issued ahead of it and returns at
// once unless `col` names its engine; the matrix-core kernel also takes `fine` from there
// stream: cut the launch into equal runs of (block, chunk) units over exactly the workgroups the chip holds (dense regime; see
// WorkItem in cpd_sweeps_mfma.hip) when mfma_stream_planes allows it; the launchers leave the plane count the merge kernels
// must read in h->mfma_col_planes / h->mfma_row_planes
void launch_colpass_mfma(prg_cpd* h, int S, bool first, bool fine, const EngineDecision* guard, bool stream);  // S segments of the streamed cloud (0 = fill the chip once)
int mfma_stream_planes(int64_t owned_points, int64_t streamed_points);  // planes per block in stream mode, 0: not applicable
// zchunk + bounding box of z4 -> motion[8..13]; eng != null: the last thread also takes the engine decision (EngineArgs)
void launch_chunk_meta_bbox(prg_cpd* h, const EngineArgs* eng);
void launch_fused_mfma(prg_cpd* h, int S, bool first, bool fine, const EngineDecision* guard);  // the single sweep (k_colpass_mfma<FUSED>)
void launch_rowpass_mfma(prg_cpd* h, int S, bool fine, bool lean, bool stream);  // lean: without the residual sums (plane 4)
int mfma_planes(int64_t owned_points, int64_t streamed_points, int S);  // partial planes those segments occupy
int mfma_chunks_per_seg(int64_t owned_points, int64_t streamed_points, int S);  // 256-point chunks one workgroup walks
int mfma_chunks_per_seg_model(int64_t owned_points, int64_t streamed_points);  // ... under round 4's rule (the engine switch's cost model)

// ---- the single sweep of a rigid iteration on the vector pipe, found and run by the column block's owner (cpd_sweeps_owner.hip) ----
constexpr int kOwnerWaves = 8;       // waves of the workgroup that owns 128 columns
constexpr int kOwnerMaxPlanes = 32;  // parts the stream is dealt out over per column block at most (partial planes of the merge)
constexpr int kOwnerColsPerLane = 1; // default of owner_cols_per_lane() (PRG_OWNER_CPL): measured below
int owner_cols_per_lane();           // 1 or 2: a wave of the owner sweep owns 64 or 128 columns
int owner_planes(int64_t owned_points, int64_t streamed_points);
// (min, A, Ux, Uy, Uz, R) -> colpart[plane][6][Ncap] + touched flags (resid_flags): what k_colfinal_resid<false> reads
void launch_colpass_owner(prg_cpd* h, bool use_seed, int planes);

// ---- sparse-regime work queue (cpd_sweeps_queue.hip) ----
constexpr int kQueueChunkGroups = 512;   // streamed groups (of 32 points) one wave of the build pass tests: 8 mask words
constexpr int kQueueTargetUnits = 16384; // the units grow (8 -> 16 -> 32 groups) when a sweep had more than twice as many (~2 per wave slot)
constexpr int kQueueMaxUnits = 98304;    // fine units the queue holds (250 MB of row-pass partials); beyond, a chunk is one coarse unit
constexpr int kQueueWorkgroups = 2048;   // persistent grid: 8 workgroups of 4 waves per CU
// q_init: groups per unit to start from (0: adapt from the previous sweep over the queue; the caller passes 32 when the previous
// E-step's sweep ran on another engine, i.e. was dense)
// resid: the residual-form single sweep of a rigid iteration (k_colpass_queue<true>): (min, A, Ux, Uy, Uz, R) -> colpart[unit][6][128]
int launch_colpass_queue(prg_cpd* h, bool use_seed, int q_init, bool resid = false);  // partial (min, sum) pairs -> colpart[unit][128]
int launch_rowpass_queue(prg_cpd* h, int q_init);                 // partial sums -> rowpart[unit][5][128]
int64_t queue_max_units(int64_t owned_points, int64_t streamed_points);
int prepare_queues(prg_cpd* h);  // allocations of both queues for the plan's current clouds

// Cull bound of all sweeps: a block of P is skipped when every entry is provably below 2^-kCullExp - of 1 (row pass: P itself,
// a column of P sums to <= 1) or of its column's largest term (column pass).  127 is "exact zero in fp32" (raw v_exp_f32 flushes
// below 2^-126; rounds 1-3).  48 leaves out at most M x 2^-48 = 3.6e-10 (M = 1e5; 3.6e-7 at M = 1e8) of any column sum or
// moment - below the fp32 accumulation noise of the sums themselves (~1e-7 ... 1e-6 against the fp64 oracle) and four orders
// below the sigma2 tolerance - while the cutoff radius^2 = kCullExp x 2 sigma2 ln 2 shrinks by 2.6: in the mid regime (cutoff
// disc much larger than a 128-point box) the sweeps evaluate less than half the pairs.  Measured at C1 (window of EM iterations
// 0..19, profiles/r4_cull_exponent_*.log): 127 -> 542 it/s, 64 -> 601, 48 -> 643; parity tests unchanged.
#ifndef PRG_CULL_EXP
#define PRG_CULL_EXP 48
#endif
constexpr float kCullExp = (float)PRG_CULL_EXP;
constexpr int kGroup = 32;     // streamed points per cull group (8 scalar quad loads)
constexpr int kSuper = 256;    // quantum of a culled segment's length (8 groups)
// culled variants (packed arithmetic, 2 adjacent points per lane); seg_len must be a multiple of kGroup
// resid: the residual-form single sweep of a rigid iteration (k_colpass_cull<true>): planes of 6 floats per column
// [plane][6][Ncap] in h->colpart, followed by one touched-flag byte per (128-column block, plane) - resid_flags()
void launch_colpass_cull(prg_cpd* h, int S, int seg_len, bool use_seed, const EngineDecision* guard = nullptr, bool resid = false);
inline unsigned char* resid_flags(const prg_cpd* h, int planes) {
    return reinterpret_cast<unsigned char*>(reinterpret_cast<float*>(h->colpart) + (int64_t)planes * 6 * h->Ncap);
}
void launch_rowpass_cull(prg_cpd* h, int S, int seg_len);
void launch_colpass_packed(prg_cpd* h, int R, int S, int seg_len);
void launch_rowpass_packed(prg_cpd* h, int R, int S, int seg_len);
void launch_colpass_scalar(prg_cpd* h, int R, int S, int seg_len);
void launch_rowpass_scalar(prg_cpd* h, int R, int S, int seg_len);
}  // namespace prg
