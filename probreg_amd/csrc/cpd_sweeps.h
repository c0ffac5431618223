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

constexpr int kGroup = 32;     // streamed points per cull group (8 scalar quad loads)
constexpr int kSuper = 256;    // quantum of a culled segment's length (8 groups)
// culled variants (packed arithmetic, 2 adjacent points per lane); seg_len must be a multiple of kGroup
void launch_colpass_cull(prg_cpd* h, int S, int seg_len, bool use_seed);
void launch_rowpass_cull(prg_cpd* h, int S, int seg_len);
void launch_colpass_packed(prg_cpd* h, int R, int S, int seg_len);
void launch_rowpass_packed(prg_cpd* h, int R, int S, int seg_len);
void launch_colpass_scalar(prg_cpd* h, int R, int S, int seg_len);
void launch_rowpass_scalar(prg_cpd* h, int R, int S, int seg_len);
}  // namespace prg
