// Internal definition of the CPD plan (opaque `prg_cpd` of include/probreg_hip.h).
#pragma once
#include <vector>

#include "prg_common.h"

// Which engine an E-step's sweeps run on while a registration is in the dense regime.  The decision needs
// sigma2, the source motion of this E-step's transform and the previous E-step's largest column minimum - all of them on
// the device - so the device takes it (last thread of k_chunk_meta_bbox) and publishes it twice: in device memory, where
// the column-pass launch the host issued AHEAD of the answer checks that it is the wanted one, and in a mapped host
// mailbox the host polls while that launch runs - no stream synchronisation, the queue never drains.
struct EngineDecision {
    unsigned seq;  // E-step counter the decision belongs to (host mailbox: written last)
    int col;       // 1: matrix-core column pass, 0: culled vector-pipe column pass
    int first;     // ... without seeds (first E-step of a registration)
    int row;       // 1: matrix-core row pass
    int fine;      // per-wave tile masks on (some groups of a chunk can be skipped by now)
    int dense;     // 0: the registration has left the dense regime for good (the host stops asking)
    float sigma2, motion, cmax, nk_ext2, nk_width, nk_far2;  // what it was decided from (PRG_DEBUG_ENGINE)
    // the switch's memory (device copy only carries it from E-step to E-step): pairs the matrix-core sweeps last evaluated
    // per owned point - targets of the column pass, source points of the row pass - and "the row pass has left them"
    float r_col, r_row;
    int row_off;
    int lean;      // 1: the matrix-core row pass runs without its residual sums (sigma2 large enough, see k_chunk_meta_bbox)
    int fused;     // 1: ONE sweep for this E-step (k_colpass_mfma<FUSED>: rigid M-step moments from the column side, no row pass)
    int pad_;      // (64-bit counters and doubles sit right behind the device copy of this struct)
};
static_assert(sizeof(EngineDecision) % 8 == 0, "the engine state behind EngineDecision needs 8-byte alignment");
struct EngineArgs {  // host -> k_chunk_meta_bbox, by value
    double ext2;
    // matrix-core sweeps while they evaluate at least this many pairs per owned point (see estep_impl)
    double r_col_bound, r_row_bound;  // (r_row_bound: for the lean row pass)
    double r_col_bound_fused;         // ... the dense regime's lower end while the fused single sweep may run (it competes with TWO vector-pipe sweeps)
    double r_row_bound_full;          // ... for the row pass with its residual sums (amplification above the lean factor)
    double owned_col, owned_row;  // N_local, M
    double streamed_col, streamed_row;  // M, N_local: what the count is when nothing is culled (the switch's initial memory)
    const double* tsum;           // (sum x, sum y, sum z, sum |x|^2) of the local target
    double lean_factor;           // lean row pass while mean |x|^2 / (sigma2 D) <= this
    double fused_factor;          // ... and the fused single sweep while it is <= this
    int dim;
    unsigned long long* work;     // [2] (128 x 16) tiles the matrix-core column / row pass of the PREVIOUS E-step evaluated
    float tbox[6];
    int slot, have_colmin, forced, reset;
    int fused_allowed;            // this E-step feeds a rigid M-step and nothing else (prg_cpd_iterate / prg_cpd_set_moments_only)
    int resid_allowed;            // ... and below the matrix cores it would run as ONE residual-form sweep on the vector pipe (exact at any amplification)
    unsigned seq;
    EngineDecision* dev;
    EngineDecision* host;
};

// Work queue of one sparse-regime sweep (cpd_sweeps_queue.hip): 16-bit need-masks per (owned block of 128 points, segment of 16
// streamed groups), the number of non-empty segments per block, and the units the persistent waves pop.
struct SweepQueue {
    unsigned long long* masks = nullptr;  // [nblocks][nchunk][8] one bit per streamed group of 32: needed or not
    int2* chunk = nullptr;            // [nblocks][nchunk] (first slot, units) of every chunk of 512 groups
    int2* units = nullptr;            // [<= max units] (block, first group | count << 22), in append order
    int* ctrl = nullptr;              // counters, see k_queue_build
    unsigned* ucount = nullptr;       // [units] (128 x 32) blocks each unit evaluated (prg_cpd_pair_counts)
    int64_t cap_blocks = 0, cap_units = 0;
    int cap_chunk = 0;
    int64_t nblocks = 0;
    int nchunk = 0, cap_soft = 0;
};

// Communicator of the per-iteration all-reduce (comm.hip): an RCCL communicator the library created (prg_comm_create) or
// one the caller handed over (prg_comm_adopt).
struct prg_comm {
    void* nccl = nullptr;  // ncclComm_t
    int rank = 0, nranks = 1, device = 0;
    bool owned = true;     // prg_comm_destroy calls ncclCommDestroy
    int64_t calls = 0;     // all-reduces issued (tests / measurements)
};

// Row accumulator block: 4 fp64 planes of Mcap (p1, px0, px1, px2), written by the moment kernel.
struct prg_cpd {
    int device = 0;
    hipStream_t stream = nullptr;

    int64_t M = 0, N = 0, Nglobal = 0;
    int D = 0;
    int64_t Mcap = 0, Ncap = 0;  // padded capacities (multiples of 1024, + slack for segmenting)

    // clouds in their kernel layout (float4 per point: x, y, z, aux)
    float4* src4 = nullptr;   // original source y_m            (aux = 0)
    float4* z4 = nullptr;     // transformed source z_m         (aux = 0), rewritten every E-step
    float4* tgt4 = nullptr;   // local target x_n               (aux = b_n = -log2(den_n + c))
    float* pt1 = nullptr;     // [Ncap] column sums of P

    // column pass partials [SA][Ncap] (dmin, sum)
    float2* colpart = nullptr;
    int64_t colpart_elems = 0;
    // row pass partials [SB][5][Mcap] (p1, ux, uy, uz, e)
    float* rowpart = nullptr;
    int64_t rowpart_elems = 0;
    // fp64 per-row block [4][Mcap]
    double* rowacc = nullptr;
    // moment block partials [nblk][24]
    double* mompart = nullptr;
    int64_t mompart_elems = 0;

    double* state = nullptr;    // owned: PRG_NMOMENTS + PRG_NPARAMS doubles
    double* moments = nullptr;  // -> state or caller-bound memory
    double* params = nullptr;   // -> state + PRG_NMOMENTS

    // tuning (0 = auto)
    int r_col = 0, seg_col = 0, r_row = 0, seg_row = 0;

    // spatial sorting + exact culling
    bool opt_sort_src = true, opt_sort_tgt = true, opt_cull = true;
    int* perm_src = nullptr;    // [M] sorted position -> original index (nullptr = identity order)
    int* perm_tgt = nullptr;    // [N]
    float* zmeta = nullptr;     // [Mcap/32][8] per group of 32 transformed source points: lo.xyz, hi.xyz, -, -
    float* tmeta = nullptr;     // [Ncap/32][8] per group of 32 target points: lo.xyz, hi.xyz, max b_n, min b_n
    float* colmin = nullptr;    // [Ncap] min_m d^2 per column from the previous E-step (seed of the cull bound),
                                // followed by [Ncap/32] per-group maxima of it
    unsigned* motion = nullptr; // [16] float bits, slot = parity of the E-step: [0,1] max_m |z_new - z_old| of the transform,
                                // [4,5] largest column minimum of the E-step (what the host needs to decide whether the
                                // matrix-core column pass may run); [8..13] bounding box of the transformed
                                // source (lo.xyz, hi.xyz) of the current E-step
    bool have_colmin = false;
    // matrix-core (dense regime) sweeps
    float4* rorig = nullptr;    // [Mcap/512] origin of each 512-row block of the last matrix-core row pass
    float4* corig = nullptr;    // [Ncap/512] origin of each 512-column block of the last fused sweep
    bool fused_in_iterate = true;  // prg_cpd_iterate(RIGID) switches moments_only on for its own E-steps (prg_cpd_set_moments_only(2): not)
    bool moments_only = false;  // E-steps of this plan feed a RIGID M-step and nothing else: the dense regime may run the fused
                                // single sweep, which leaves no per-point p1 / px (prg_cpd_set_moments_only, prg_cpd_iterate)
    bool init_rot_orthonormal = true;  // ... only from a rotation: the column-side sums are mapped back through s R
    bool resid_sweep = true;    // ... and, where the column pass runs on the vector pipe, the residual-form single sweep (DESIGN.md 3.1f;
                                // prg_cpd_set_resid_sweep(0): two sweeps there)
    int pred_fused = 0;         // the previous E-step ran the fused sweep (what the host launches ahead of the decision)
    bool last_estep_fused = false;
    bool rowacc_valid = false;  // the per-point block (p1, px) holds the last E-step's result (not after a fused sweep)
    float* zchunk = nullptr;    // [Mcap/256][8] box of every 256-point chunk of the transformed source (per E-step)
    float* tchunk = nullptr;    // [Ncap/256][8] box + largest b_n of every 256-point chunk of the target (per E-step)
    int dense_engine = 1;       // 0: VALU sweeps only, 1: matrix-core sweeps in the dense regime,
                                // 2: both sweeps on the matrix cores, always (tests)
    double dense_bound = 0.0;    // > 0: matrix-core column pass while it evaluates at least this many source points per target (0: estep_impl's model)
    double* tsum_local = nullptr;            // (sum x, sum y, sum z, sum |x|^2) of the local target, beside the decision
    unsigned long long* eng_work = nullptr;  // [2] tiles evaluated by the matrix-core column / row pass (read + cleared by the decision)
    int q_first_col = 32, q_first_row = 32;  // groups per unit of the first queue sweep after a matrix-core one
    bool eng_reset = true;       // the switch's memory is void (new registration, engine mode changed)
    bool mfma_off = false;      // this registration has left the dense regime: no more engine decisions
    EngineDecision* eng_dev = nullptr;   // device copy of the current E-step's decision (guard of the column-pass launches)
    EngineDecision* eng_host = nullptr;  // mapped, coherent host memory: the mailbox the host polls
    EngineDecision* eng_host_dev = nullptr;  // ... as the device addresses it
    int pred_col = 1;           // the column-pass engine the host launches ahead of the decision (= the previous decision)
    int pred_fine = 0;          // ... and whether that decision had the per-wave tile masks on (0: dense regime -> stream mode)
    bool mfma_grid_fine = false; // the previous matrix-core column pass skipped >= 10 % of its pairs: launches are cut into >= 3 rounds of shorter segments
    bool mfma_stream = true;    // dense-regime launches of the matrix-core sweeps are cut in stream mode (prg_cpd_set_stream_mode)
    int mfma_col_planes = 0, mfma_row_planes = 0;  // partial planes the last matrix-core column / row pass wrote (grid or stream mode)
    bool last_estep_row_lean = false;  // ... matrix-core row pass without its residual sums
    double fused_factor = 256.0;       // fused single sweep while mean |x|^2 / (sigma2 D) <= this (prg_cpd_set_fused_factor)
    double lean_factor = -1.0;         // lean row pass while mean |x|^2 / (sigma2 D) <= this (< 0: the default, 64; prg_cpd_set_lean_factor)
    bool last_estep_mfma = false, last_estep_row_mfma = false;  // engines of the last E-step's column / row pass
    double text2 = 0.0, sext2 = 0.0;  // squared bounding-box diagonals of the local target and of the source
    float tbox[6] = {0, 0, 0, 0, 0, 0};  // bounding box of the local target (lo.xyz, hi.xyz)
    // sparse regime: device-built work queues of the two sweeps; sparse_engine 1 = use them for the
    // vector-pipe sweeps of large clouds AND the owner sweep (cpd_sweeps_owner.hip) for single-sweep rigid iterations (default),
    // 0 = the grid-per-(block, segment) culled sweeps of round 1, 2 = queue always, 3 = round 5's default (1 without the owner sweep)
    SweepQueue qcol, qrow;
    int sparse_engine = 1;
    bool qcol_live = false, qrow_live = false;  // the previous E-step's column / row pass ran over the queue (its unit size adapts from there)
    // measurement hook: evaluated (wave, group) blocks per workgroup of the last culled column / row pass
    // ([0, wg_cap) column pass, [wg_cap, 2 wg_cap) row pass); wg_col / wg_row = workgroups of the last launches,
    // dense_pairs_* = pairs covered by the last NON-culled launches (0 when the culled kernels ran)
    unsigned* wgcount = nullptr;
    int64_t wg_cap = 0, wg_col = 0, wg_row = 0;
    double wg_col_pairs = 128.0 * 32.0, wg_row_pairs = 128.0 * 32.0;  // pairs one counted block stands for
    double dense_pairs_col = 0.0, dense_pairs_row = 0.0;
    uint64_t estep_count = 0;   // parity selects the motion slot of the current E-step

    // staging for uploads / moments_from_estep
    void* stage = nullptr;
    size_t stage_bytes = 0;

    // non-rigid state
    float* G = nullptr;        // [M][M] float32 (row-major)
    double* W = nullptr;       // [M][3] float64 (row-major, 3 columns always)
    // ... or its pivoted-Cholesky factor G = F F^T (non-rigid CPD): F[k * f_ld + i], k < f_rank
    double* F = nullptr;
    int64_t f_ld = 0;          // round_up(M, 256) + 32
    int f_rank = 0, f_cap = 0;
    int nr_solver = 1;         // 1: low-rank factor when the rank allows (default), 0: dense G + M x M Cholesky
    int nr_max_rank = 0;       // 0: min(2048, M / 2)
    double nr_tol = 1.0e-11;   // the factor stops when the largest residual diagonal entry of G - F F^T is below this (the
                               // reference's own float32 G is 6e-8 from the exact kernel; 1e-11 is rank 119 at C3 - one 128-block
                               // of the reduced system - where 1e-14 is rank 176)
    double beta = 0.0;
    bool nonrigid = false;
    double* nr_work = nullptr;  // [16 M] doubles: G.W product and scratch
    bool gw_valid = false;      // nr_work[0, 3M) holds G W of the current W (left by the M-step, consumed by the transform)
    size_t nr_work_bytes = 0;
    double* nr_solve = nullptr;  // M-step workspace: S (fp64 M x M), block inverses, vectors
    size_t nr_solve_bytes = 0;
    int* nr_info = nullptr;      // device: first non-positive pivot + 1 of a low-rank M-step since the last report (0: none)
    double* nr_prior = nullptr;            // [4][M]: p1_tilde, px_tilde (3 planes) of ConstrainedNonRigidCPD
    double nr_alpha = 0.0;                 // 0 = no correspondence priors
    hipStream_t nr_stream2 = nullptr;      // side stream of the look-ahead Cholesky
    std::vector<hipEvent_t> nr_events;

    // per-source log-weights (BCPD E-step, bcpd.py:53-72): srcw[m] = ln a_m <= 0 in kernel (sorted) order; the
    // transform kernel turns it into the additive squared distance q_m = -2 sigma2 ln a_m carried in z4.w
    float* srcw = nullptr;
    double uniform_ratio = 0.0;  // > 0: replaces M / N in the outlier constant of cpd.py:78-79
    bool bcpd = false;           // G is the inverse multiquadric kernel, W holds the displacement v_hat

    double* pinned = nullptr;    // 64 doubles of pinned host memory for the per-iteration parameter read-back

    bool have_source = false, have_target = false, have_estep = false;
    double last_w = 0.0;

    // multi-GPU: when set, prg_cpd_estep / prg_cpd_init_sums end with the SUM all-reduce of the moment block (and, for a
    // non-rigid plan, of the per-point block) on the plan's stream; not owned by the plan
    prg_comm* comm = nullptr;
};

namespace prg {
struct UniqueId { char bytes[PRG_COMM_ID_BYTES]; };  // ncclUniqueId
int comm_all_reduce_f64(prg_comm* c, double* buf_dev, int64_t count, hipStream_t st);
int ensure_stage(prg_cpd* h, size_t bytes);
// kd-tree order of a cloud built on the device (spatial_order.hip): pts_dev [n][dim] floats in the caller's order -> perm_dev [n]
int device_kd_order(const float* pts_dev, int64_t n, int dim, int* perm_dev, hipStream_t st, int leaf = 32);
// non-rigid (cpd_nonrigid.hip)
int nonrigid_displacement(prg_cpd* h, const double** gw_out);  // G W of the current W (device, [M][3])
int nonrigid_gw(prg_cpd* h, const double* w3, double* out3);  // out3[m][3] = G * w3[m][3] (fp64)
int nonrigid_free(prg_cpd* h);
int build_kernel_matrix(prg_cpd* h, int kind, double param);  // kind 0: rbf(beta), 1: inverse multiquadric(c)
// low-rank factor (cpd_nonrigid.hip): out[k][0..2] = sum_i F[k][i] x3[i][0..2]  /  out3[i][0..2] = sum_k F[k][i] v[k][0..2]
int lowrank_ft3(prg_cpd* h, const double* x3, double* out);
int lowrank_apply(prg_cpd* h, const double* v, double* out3);
}  // namespace prg
