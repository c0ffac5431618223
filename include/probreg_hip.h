/*
 * probreg_hip.h - C ABI of libprobreg_hip.so, the MI355X (gfx950) engine behind
 * probreg's CPD / FilterReg EM hot path.
 *
 * Every entry point names the reference interface it replaces (paths relative
 * to the neka-nat/probreg tree, v0.3.7).  Conventions:
 *   - all functions return 0 on success, a negative prg_status on failure;
 *     prg_last_error() returns a thread-local message for the last failure;
 *   - "hd" pointers may be host OR device pointers (copied with
 *     hipMemcpyDefault); "dev" pointers must be device pointers;
 *   - point clouds are row-major (count x D) float32, D = 2 or 3;
 *   - one handle = one device = one HIP stream; calls on a handle are
 *     asynchronous on that stream unless the name ends in a host read-back
 *     (get_*, *_host) which synchronises the stream;
 *   - no torch / numpy types appear anywhere in this file.
 */
#ifndef PROBREG_HIP_H
#define PROBREG_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum prg_status {
    PRG_OK = 0,
    PRG_ERR_INVALID = -1,   /* bad argument (ValueError / AssertionError in the reference) */
    PRG_ERR_HIP = -2,       /* HIP runtime error; message holds hipGetErrorString */
    PRG_ERR_STATE = -3,     /* call order violated (e.g. estep before set_target) */
    PRG_ERR_NOMEM = -4
} prg_status;

typedef enum prg_tf_kind {
    PRG_TF_RIGID = 0,       /* probreg.cpd.RigidCPD    (cpd.py:123-192) */
    PRG_TF_AFFINE = 1,      /* probreg.cpd.AffineCPD   (cpd.py:195-244) */
    PRG_TF_NONRIGID = 2     /* probreg.cpd.NonRigidCPD (cpd.py:247-303) */
} prg_tf_kind;

const char* prg_last_error(void);
int prg_version(void);
int prg_device_count(int* count);

/* ------------------------------------------------------------------------------------------
 * Layout of the two small fp64 device blocks every CPD plan owns.
 *
 * MOMENTS (PRG_NMOMENTS doubles) - the per-iteration all-reduce payload:
 *   [0]      S0   = sum_m p1_m                        (n_p,            cpd.py:88)
 *   [1..3]   Sx   = sum_m px_m                        (xp.sum(px,0),   cpd.py:169)
 *   [4..6]   Sy   = sum_m p1_m y_m                    (source.T @ p1,  cpd.py:170)
 *   [7..15]  Sxy  = sum_m px_m y_m^T   row-major 3x3  (px.T @ source,  cpd.py:173)
 *   [16..21] Syy  = sum_m p1_m y_m y_m^T  (xx,xy,xz,yy,yz,zz)          (cpd.py:179/234)
 *   [22]     Sxx  = sum_n pt1_n |x_n|^2               (cpd.py:183/237)
 *   [23]     reserved
 *   [24..26] sum_n x_n   [27] sum_n |x_n|^2   (target sums for the sigma2 initialiser,
 *                                              math_utils.py:28-29; local shard, all-reduced)
 *   [28..31] reserved (zero)
 * PARAMS (PRG_NPARAMS doubles):
 *   [0..8]   linear part, row-major 3x3: rot (rigid) or b (affine)   transformation.py:43-78
 *   [9..11]  t
 *   [12]     scale (1 for affine)
 *   [13]     sigma2        [14] q        [15] n_p of the last E-step
 *   [16]     iteration counter (as double)
 *   [17..31] reserved
 * ---------------------------------------------------------------------------------------- */
#define PRG_NMOMENTS 32
#define PRG_NPARAMS 32

typedef struct prg_cpd prg_cpd;

/* Plan life cycle.  `hip_stream` is a hipStream_t (NULL = the device's null stream).
 * Replaces: CoherentPointDrift.__init__ backend selection, cpd.py:42-59. */
int prg_cpd_create(prg_cpd** out, int device, void* hip_stream);
int prg_cpd_destroy(prg_cpd* h);

/* [r6] The order a CPD plan stores a cloud in (sorted position -> original index): the in-order walk of a left-aligned kd-tree
 * with 32-point leaves - every aligned run of 2^k leaves (the 32-point groups, 128-point blocks, 256-point chunks and 512-point
 * blocks the sweeps cull by) is one axis-aligned cell of the cloud.  on_device != 0: built on the GPU, level by
 * level (csrc/spatial_order.hip, what prg_cpd_set_source / prg_cpd_set_target run); 0: the host build (csrc/morton.h).  Both give
 * the same cells.  No reference counterpart (probreg keeps the caller's order); tests and tools. */
int prg_spatial_order(const float* points_hd, int64_t n, int dim, int on_device, int* perm_host);
/* Engine options, before the clouds are uploaded (all default to 1): Morton-sort the source / the target at
 * upload (every output keeps the caller's point order) and skip (wave, 32-point group) blocks whose every pair
 * is an exact zero in fp32.  The non-rigid path keeps the source unsorted. */
int prg_cpd_set_options(prg_cpd* h, int sort_source, int sort_target, int cull);

/* Engine of the E-step's DENSE regime (sigma2 large: every pair contributes).  mode 1 (default; clouds of >= 8192 points
 * each): the pair sweeps take their exponents from the matrix cores (bf16x3-split MFMA distance blocks)
 * while that is the faster engine - decided per E-step on the device from the number of pairs the matrix-core sweeps of
 * the previous E-step evaluated (they skip exact zeros in blocks of 512 x 256 and, per wave, 128 x 16) against a cost
 * model of the two engines; from there on the culled vector-pipe sweeps, which skip 128 x 32 blocks and share the work out
 * evenly, are faster and the registration stays on them (the row pass leaves first).  mode 0: vector-pipe sweeps only;
 * mode 2: both sweeps on the matrix cores always (tests, measurements).  bound > 0 replaces the model's bound of the
 * column pass: matrix cores while they evaluate at least `bound` source points per target; 0 keeps the current setting.
 * prg_cpd_last_estep_engine reports which engine the last E-step's column pass used (1 = matrix cores). */
int prg_cpd_set_dense_engine(prg_cpd* h, int mode, double bound);
int prg_cpd_last_estep_engine(prg_cpd* h, int* engine);
/* The bounds of that decision for a source of m points and a local target of n_local (host arithmetic, no device needed):
 * the matrix-core column pass is left below *col_bound evaluated source points per target, the row pass below *row_bound
 * evaluated targets per source point. */
int prg_cpd_engine_bounds(int64_t m, int64_t n_local, double* col_bound, double* row_bound);
/* Sparse regime (sigma2 small: most 128 x 32 blocks of P are exact zeros): 1 (default) - with M and the local N both >= 32768
 * the vector-pipe sweeps run over a device-built work queue (need-masks -> units -> persistent waves,
 * csrc/cpd_sweeps_queue.hip), smaller problems on the grid; 2 - the queue always; 0 - always one wave per (128-point block,
 * 512-point segment) whether it finds work or not (the culled sweeps of csrc/cpd_sweeps_packed.hip).  Same pairs, same
 * arithmetic, exact either way; tests / measurements.  [r6] Under mode 1 the SINGLE sweep of a rigid iteration
 * (prg_cpd_set_moments_only) is the owner sweep of csrc/cpd_sweeps_owner.hip wherever it runs on the vector pipe - the column
 * block's workgroup finds its cells through the chunk / group hierarchy of the kd-ordered clouds, no queue, no build pass
 * (DESIGN.md 3.1g); 3 - round 5's default: mode 1 without the owner sweep (the residual-form sweep over the queue / grid). */
int prg_cpd_set_sparse_engine(prg_cpd* h, int mode);
/* ... and for both sweeps of the last E-step: 1 = matrix cores, 0 = vector pipe (column pass, row pass).  An E-step that ran as a
 * single sweep (prg_cpd_last_estep_fused) has no row pass: row_engine and prg_cpd_last_estep_lean report 0 for it. */
int prg_cpd_last_estep_engines(prg_cpd* h, int* col_engine, int* row_engine);
/* 1 if the last E-step's matrix-core row pass ran WITHOUT its residual sums sum_n P_mn |x_n - o|^2 (19 instead of 21 flop
 * per pair): while sigma2 * D * 64 >= mean |x|^2 of the local target the M-step's sum_n pt1_n |x_n|^2 is taken from the column
 * side instead. */
int prg_cpd_last_estep_lean(prg_cpd* h, int* lean);
/* The amplification up to which the row pass may run lean: while mean |x|^2 / (sigma2 D) of the local target <= factor
 * (default 64; 0 = never; a huge value = wherever the matrix-core row pass runs - tests and tools/lean_error.py measure the
 * sigma2 error of the lean pass far beyond where the default allows it); negative restores the default. */
int prg_cpd_set_lean_factor(prg_cpd* h, double factor);
/* How a DENSE-regime launch of the matrix-core sweeps (nothing to skip yet) is cut into workgroups: 1 (default) - equal runs
 * of (512-point block, 256-point chunk) units over whole rounds of the workgroups the chip holds, at most one unit of
 * difference between any two; 0 - the grid of (block, segment) workgroups every other regime uses.  Same pairs, same
 * arithmetic; the partial sums are merged in a different order (tests, measurements). */
int prg_cpd_set_stream_mode(prg_cpd* h, int on);

/* Upload the (already centred) source cloud, replicated on every device.
 * Replaces: CoherentPointDrift.set_source, cpd.py:61-62. */
int prg_cpd_set_source(prg_cpd* h, const float* source_hd, int64_t m, int dim);

/* Upload this device's contiguous shard of the target cloud; n_global is the number of
 * target points over all shards (it enters `c`, cpd.py:78-79, and q0, cpd.py:148).
 * Replaces: the `target` argument of CoherentPointDrift.registration, cpd.py:106. */
int prg_cpd_set_target(prg_cpd* h, const float* target_hd, int64_t n_local, int dim, int64_t n_global);

/* Optional: make the plan accumulate its moments into caller-owned device memory
 * (PRG_NMOMENTS doubles) so the caller can all-reduce them in place (RCCL). */
int prg_cpd_bind_moments(prg_cpd* h, double* moments_dev);
/* Device addresses of the blocks (for in-place collectives / zero-copy views). */
int prg_cpd_moments_ptr(prg_cpd* h, double** moments_dev);
int prg_cpd_params_ptr(prg_cpd* h, double** params_dev);

/* ---- multi-GPU: the one exchange step of the path ---------------------------------------------
 * The reference is single-process (its EM loop, cpd.py:106-120, is what gets sharded): every rank keeps the whole
 * source, owns a run of target rows, and the partial MOMENTS of its E-step (cpd.py:84-88 restricted to its columns)
 * are summed over the ranks before the M-step (cpd.py:160-192 / 219-244), which every rank then runs identically.
 * A prg_comm wraps an RCCL communicator (librccl is bound with dlopen on first use - a single-GPU caller never
 * loads it): rank 0 calls prg_comm_unique_id and hands the PRG_COMM_ID_BYTES bytes to the other ranks by any means
 * (the Python layer broadcasts them through torch.distributed), every rank calls prg_comm_create (ncclCommInitRank;
 * collective).  prg_comm_adopt wraps an ncclComm_t the caller already has (not destroyed by prg_comm_destroy).
 * With prg_cpd_set_comm(plan, comm) the plan's prg_cpd_init_sums and prg_cpd_estep END with the SUM all-reduce of
 * MOMENTS (init_sums: all 32 doubles; an E-step: the 24 it wrote; a non-rigid plan: the per-point block of prg_cpd_rowacc_ptr as well) on the plan's own stream: an EM
 * iteration is enqueue-only, nothing between the E-step's last kernel and the M-step leaves the library.
 * comm == NULL detaches (the caller all-reduces MOMENTS itself, e.g. through prg_cpd_bind_moments). */
#define PRG_COMM_ID_BYTES 128
typedef struct prg_comm prg_comm;
int prg_comm_available(int* rccl_version);            /* PRG_OK when librccl could be bound (version: ncclGetVersion) */
int prg_comm_unique_id(unsigned char* id_out);        /* [PRG_COMM_ID_BYTES], host */
int prg_comm_create(prg_comm** out, const unsigned char* id, int rank, int nranks, int device);
int prg_comm_adopt(prg_comm** out, void* nccl_comm, int device);
int prg_comm_info(prg_comm* c, int* rank, int* nranks, int64_t* all_reduces_issued);
int prg_comm_destroy(prg_comm* c);
/* In-place SUM all-reduce of `count` doubles at a device address, on `hip_stream` (what the plan issues itself). */
int prg_comm_all_reduce_f64(prg_comm* c, double* buf_dev, int64_t count, void* hip_stream);
int prg_cpd_set_comm(prg_cpd* h, prg_comm* comm);
/* `n_iter` EM iterations of the rigid / affine registration enqueued back to back: transform + E-step [+ all-reduce]
 * + M-step, the loop body of CoherentPointDrift.registration (cpd.py:110-113) without its host-side convergence test
 * (tol < 0, no callbacks).  Nothing is read back; prg_cpd_get_params afterwards synchronises. */
int prg_cpd_iterate(prg_cpd* h, int kind, int update_scale, double w, int n_iter);
/* What a RIGID EM iteration needs of its E-step (cpd.py:160-192) are 23 sums, not the per-point arrays p1 / px.  mode 1: every
 * prg_cpd_estep of this plan may therefore run, while sigma2 is large (dense regime, sigma2's amplification within the fused
 * factor), as ONE sweep over the pairs - the column pass with the row pass' contraction on the source side, the moments taken
 * from per-column sums - instead of two: MOMENTS, pt1 and the parameter block come out as always,
 * prg_cpd_get_estep has no p1 / px to return after such an E-step (it says so), and the caller must follow it with
 * prg_cpd_mstep(PRG_TF_RIGID).  mode 0 (default): only prg_cpd_iterate(PRG_TF_RIGID) does this, for its own E-steps.
 * mode 2: nobody does (two sweeps always: tests, measurements).  prg_cpd_last_estep_fused reports what the last E-step did. */
int prg_cpd_set_moments_only(prg_cpd* h, int mode);
int prg_cpd_last_estep_fused(prg_cpd* h, int* fused);
/* The fused sweep runs while the matrix-core column pass would and mean |x|^2 / (sigma2 D) of the local target <= factor
 * (default 256: sigma2 within 1.4e-6 of the fp64 oracle's there, within 3e-6 at 1300 - profiles/r4_fused_error_100k.log);
 * tests force it further with a huge value, 0 switches it off. */
int prg_cpd_set_fused_factor(prg_cpd* h, double factor);
/* Below the dense regime (column pass on the vector pipe) the same E-steps - those that feed nothing but a rigid M-step,
 * cpd.py:160-192 - run as ONE sweep as well: the culled / queued column pass carries the residual sums of its columns,
 * U_n = sum_m K (x_n - z_m) and R_n = sum_m K |x_n - z_m|^2, and the moments follow per column in fp64 (no
 * amplification limit: the sums are residuals against the current transformation).  on = 1 (default) / 0: two sweeps there
 * (A/B measurements, tests).  prg_cpd_last_estep_fused reports 1 for either single sweep, prg_cpd_last_estep_engine which pipe
 * its column pass ran on. */
int prg_cpd_set_resid_sweep(prg_cpd* h, int on);

/* sigma2 initialiser, step 1: local target sums -> MOMENTS[24..27] (others zeroed).
 * Replaces: mu.squared_kernel_sum, math_utils.py:28-29 -> cc/math_utils.cc:5-15
 * (closed form, never materialises M x N). All-reduce MOMENTS between step 1 and 2. */
int prg_cpd_init_sums(prg_cpd* h);
/* step 2: sigma2_0 and q0 = 1 + N*D/2*log(sigma2_0) into PARAMS.  `init_params_host` (NULL = identity,
 * zero, 1, zero) holds 16 doubles: [0..8] linear part, [9..11] t, [12] scale, [13..15] delta =
 * origin_target - origin_source when the caller subtracted different origins from the two uploaded
 * clouds (sigma2_0 is a mean squared source-target distance and is not invariant to that).
 * Replaces: RigidCPD._initialize cpd.py:145-153, AffineCPD._initialize :209-217,
 * NonRigidCPD._initialize :277-282. */
int prg_cpd_init_params(prg_cpd* h, const double* init_params_host);

/* One E-step on the local shard with the transformation currently in PARAMS:
 * transform (transformation.py:49-50 / 77-78 / 101-102), column pass (den, cpd.py:74-82),
 * row pass (P1, PX in residual form, cpd.py:84-87) and the fp64 moment reduction.
 * Result: MOMENTS[0..22] (local partial sums).  `w` is the uniform-noise weight.
 * Replaces: CoherentPointDrift.expectation_step, cpd.py:71-88. */
int prg_cpd_estep(prg_cpd* h, double w);

/* Same E-step with HIP events recorded on the plan's stream between its kernels.
 * ms_out[0..4] = transform, column pass, column finalise, row pass, moment reduction; ms_out[5] = total.
 * (measurement hook for bench.py's `roofline` object; synchronises the stream) */
int prg_cpd_estep_timed(prg_cpd* h, double w, float* ms_out);

/* Measurement hook: source-target pairs the column / row pass of the LAST E-step actually evaluated.  The culled
 * sweeps skip (wave, 32-point group) blocks whose every pair is an exact zero; each workgroup
 * leaves its count of evaluated blocks (128 x 32 pairs each) in a device array that is summed here.  Dense
 * launches report the pairs their grid covers (pads included).  Synchronises the stream. */
int prg_cpd_pair_counts(prg_cpd* h, double* col_pairs, double* row_pairs);

/* M-step from (all-reduced) MOMENTS into PARAMS; identical on every rank.
 * Replaces: RigidCPD._maximization_step cpd.py:160-192 (update_scale as there) and
 * AffineCPD._maximization_step cpd.py:219-244. */
int prg_cpd_mstep(prg_cpd* h, int kind, int update_scale);

/* Host read-back / overwrite of PARAMS (synchronises the stream). */
int prg_cpd_get_params(prg_cpd* h, double* params_host);
int prg_cpd_set_params(prg_cpd* h, const double* params_host);
int prg_cpd_get_moments(prg_cpd* h, double* moments_host);

/* Materialise the EstepResult of the last prg_cpd_estep in the reference's terms
 * (cpd.py:17, :84-88): pt1[n_local], p1[m], px[m*dim] as float64; any pointer may be NULL.
 * p1 / px are this shard's partial sums over its local target points. */
int prg_cpd_get_estep(prg_cpd* h, double* pt1_hd, double* p1_hd, double* px_hd);
/* Transformed source (m x dim, float32) used by the last E-step. */
int prg_cpd_get_tsource(prg_cpd* h, float* tsource_hd);

/* M-step from explicitly supplied EstepResult arrays (the reference's public
 * maximization_step(target, estep_res) signature, cpd.py:90-93): uploads pt1/p1/px
 * (float64, host or device, caller's point order), rebuilds MOMENTS[0..22] on the device and the
 * per-point (p1, px) block the non-rigid solve reads (NonRigidCPD.maximization_step, cpd.py:272-303:
 * follow with prg_cpd_set_params(sigma2_p) and prg_cpd_mstep_nonrigid); no M-step yet. */
int prg_cpd_moments_from_estep(prg_cpd* h, const double* pt1_hd, const double* p1_hd, const double* px_hd);

/* Tuning knobs (0 keeps the automatic value): points per lane (2 or 4 select the non-culled packed sweeps, -2 / -4
 * their scalar form; 0 = automatic, which is also the only setting that uses the culled sweeps) and the number of
 * segments (<= 256) the streamed axis is split into, for the column and the row pass.  The culled sweeps evaluate
 * four consecutive segments per workgroup, so S segments occupy ceil(S / 4) partial planes. */
int prg_cpd_set_tuning(prg_cpd* h, int r_col, int seg_col, int r_row, int seg_row);

/* ---- non-rigid CPD ------------------------------------------------------------------- */
/* Build G_ij = exp(-|y_i-y_j|^2/(2*beta)) once per source.
 * Replaces: NonRigidTransformation.__init__ transformation.py:91-99 -> mu.rbf_kernel
 * math_utils.py:36-37 -> cc/math_utils.cc:17-19.
 * The plan does not store the M x M matrix when it does not have to: a Gaussian kernel matrix of a point cloud is
 * numerically low rank, and the plan keeps its pivoted-Cholesky factor G = F F^T (fp64, M x r, columns evaluated on the
 * fly; the factorisation stops when every entry of G - F F^T is below `tol`, default 1e-11 - the reference's float32 G is
 * 6e-8 from the exact kernel).  Every later product with
 * G and the M-step's solve then cost O(M r) / O(M r^2).  When the rank would exceed max_rank the plan
 * falls back to the dense float32 matrix and the M x M fp64 Cholesky. */
int prg_cpd_nonrigid_build_g(prg_cpd* h, double beta);
/* Solver of the next prg_cpd_nonrigid_build_g.  mode 1 (default): low-rank factor when the rank allows; mode 0: always
 * the dense matrix.  max_rank: 0 = min(2048, M / 2); tol: 0 keeps the current value. */
int prg_cpd_nonrigid_set_solver(prg_cpd* h, int mode, int max_rank, double tol);
/* Rank of the factor the plan holds (0: it holds the dense matrix). */
int prg_cpd_nonrigid_rank(prg_cpd* h, int* rank);
/* Copy G (m*m float32, row-major) out - parity tests / `tf.g` attribute (evaluated on the spot when the plan holds only
 * the factor). */
int prg_cpd_nonrigid_get_g(prg_cpd* h, float* g_hd);
/* Set / get W (m x dim float64).  W = 0 after build_g (cpd.py:281). */
int prg_cpd_nonrigid_set_w(prg_cpd* h, const double* w_hd);
int prg_cpd_nonrigid_get_w(prg_cpd* h, double* w_hd);
/* T = Y + G W for the current W (m x dim float64): NonRigidTransformation._transform on the control
 * points, transformation.py:101-102. */
int prg_cpd_nonrigid_apply(prg_cpd* h, double* t_hd);
/* Correspondence priors of ConstrainedNonRigidCPD (cpd.py:306-404): p1_tilde [m] and px_tilde [m x dim]
 * (float64) are the row sums / target-weighted row sums of the 0-1 prior matrix, `alpha` its reliability;
 * the next prg_cpd_mstep_nonrigid then solves cpd.py:391-396.  NULL pointers clear the priors. */
int prg_cpd_nonrigid_set_priors(prg_cpd* h, const double* p1_tilde_hd, const double* px_tilde_hd, double alpha);
/* Device address of the per-point E-step block (4*m doubles: p1[m], px0[m], px1[m], px2[m])
 * followed by MOMENTS-style scalars; this is the non-rigid all-reduce payload. */
int prg_cpd_rowacc_ptr(prg_cpd* h, double** rowacc_dev, int64_t* count);
/* Non-rigid M-step: W = solve(diag(p1) G + lmd*sigma2_prev*I, px - diag(p1) Y), T = Y + G W,
 * sigma2 = (tr(X^T diag(pt1) X) - 2 tr(px^T T) + tr(T^T diag(p1) T)) / (n_p D), q := sigma2.
 * Replaces: NonRigidCPD._maximization_step, cpd.py:284-303. */
int prg_cpd_mstep_nonrigid(prg_cpd* h, double lmd);

/* ---- Bayesian CPD -------------------------------------------------- */
/* Per-source weights of the E-step: P_mn = a_m e_mn / (c + sum_m a_m e_mn), e_mn = exp(-|x_n - z_m|^2 / 2 sigma2),
 * with a_m = exp(log_weights[m]), log_weights <= 0 ([m] float64, host or device, caller's point order); NULL
 * clears them.  `uniform_ratio` > 0 replaces M/N in the outlier constant c = (2 pi sigma2)^(D/2) w/(1-w) * ratio.
 * BCPD's E-step (bcpd.py:53-72) is this with a_m = alpha_m exp(-s^2 D Sigma_mm / 2 sigma2) and ratio = 1/N
 * (both rescaled by max a_m on the host); nu' = pt1, nu = p1, px come from prg_cpd_get_estep as for CPD.
 * Replaces: BayesianCoherentPointDrift.expectation_step, bcpd.py:53-72 (the M x N `pmat`, its `np.kron`
 * products :69-70 and the Python list comprehension :57). */
int prg_cpd_set_source_weights(prg_cpd* h, const double* log_weights_hd, double uniform_ratio);
/* G = inverse multiquadric kernel 1/sqrt(|y_i - y_j|^2 + c) of the source in float32 (cc/math_utils.cc:32-34,
 * bcpd.py:107) and BCPD mode: the transform becomes z = s R (y + v_hat) + t (CombinedTransformation) with the
 * v_hat of the last prg_cpd_bcpd_solve (0 before the first). */
int prg_cpd_bcpd_build_g(prg_cpd* h, double c);
/* Core of CombinedBCPD._maximization_step (bcpd.py:123-133) for nu = nu_hd ([m] float64, caller's order) or, when
 * nu_hd is NULL, the p1 of the last E-step:
 *   Sigma = (lmd G^-1 + cfac diag(nu))^-1,  v_hat = cfac * Sigma * diag(nu) * resid   (resid = T^-1(x_hat) - y),
 * returned as v_hat [m x dim] and diag(Sigma) [m] (float64, caller's point order); v_hat also stays on the
 * device for the next transform.  Woodbury form on S = (lmd/cfac) I + D^1/2 G D^1/2 - no G^-1, no M x M inverse:
 * fp64 Cholesky + a triangular solve with M right-hand sides on the matrix cores (4/3 M^3 flop). */
int prg_cpd_bcpd_solve(prg_cpd* h, double lmd, double cfac, const double* nu_hd, const double* resid_hd,
                       double* vhat_hd, double* sigma_diag_hd);

/* ---- direct Gauss transform ------------------------------------------------------------ */
/* out[c*t + i] = sum_j weights[c*s + j] * exp(-|target_i - source_j|^2 / h^2), float64 out.
 * Unlike the CPD clouds these are FLOAT64 (row-major count x dim): the reference's direct path works on the arrays as
 * given (float64), and a narrow kernel is sensitive to the rounding of the coordinates - the differences are formed in
 * fp64, distance and exponential in fp32, weights and sums in fp64; up to four weight rows share one sweep.
 * Replaces: gauss_transform._gauss_transform_direct / Direct.compute / GaussTransform.compute
 * (gauss_transform.py:10-25, 46-60). */
int prg_gauss_transform_direct(int device, void* hip_stream, const double* source_hd, int64_t s,
                               const double* target_hd, int64_t t, int dim, const double* weights_hd,
                               int n_weight_rows, double h, double* out_hd);

/* sum_{m,n} |x_m - y_n|^2 / (M*D*N), closed form in fp64 on the device.
 * Replaces: mu.squared_kernel_sum, math_utils.py:28-29. */
int prg_squared_kernel_sum(int device, void* hip_stream, const float* x_hd, int64_t m,
                           const float* y_hd, int64_t n, int dim, double* out_host);
/* Dense K = exp(-|x_i-y_j|^2/(2*beta)) float32 (rows = x). Replaces mu.rbf_kernel, math_utils.py:36-37. */
int prg_rbf_kernel(int device, void* hip_stream, const float* x_hd, int64_t m, const float* y_hd,
                   int64_t n, int dim, double beta, float* out_hd);
/* Dense K = 1/sqrt(|x_i-y_j|^2 + c) float32 (rows = x).  Replaces mu.inverse_multiquadric_kernel,
 * math_utils.py:50-51 -> cc/math_utils.cc:32-34 (BCPD's coherence kernel, bcpd.py:107). */
int prg_inverse_multiquadric_kernel(int device, void* hip_stream, const float* x_hd, int64_t m, const float* y_hd,
                                    int64_t n, int dim, double c, float* out_hd);
/* mean_i min_j |a_i - b_j| by brute force on the GPU (float32 distances, float64 mean).
 * Replaces mu.compute_rmse, math_utils.py:32-33 (cKDTree query; BCPD's convergence criterion bcpd.py:93). */
int prg_nn_mean_distance(int device, void* hip_stream, const float* a_hd, int64_t m, const float* b_hd, int64_t n,
                         int dim, double* out_host);

/* ---- permutohedral lattice (Gaussian filtering) --------------------------------------------- */
/* Replaces the pybind class probreg._permutohedral_lattice.Permutohedral
 * (cc/permutohedral_lattice_py.cc:13-21 over third_party/permutohedral/permutohedral.cpp) behind
 * probreg.gaussian_filtering.Permutohedral (gaussian_filtering.py:8-17). */
typedef struct prg_ph prg_ph;
/* How the splat (permutohedral.cpp:491-500 / :548-556, `values[o] += w * in[i]` over the points in order) accumulates,
 * process-wide, for prg_ph_filter and prg_fr_estep:
 *   1 (default)  64-bit fixed-point atomics: order independent - identical bits from run to run, every vertex the correctly
 *                rounded exact sum of its terms (the reference's float32 chain carries its own round-off; the two agree to it);
 *   2            the reference's own order: one sequential float32 chain per vertex in point order (stable sort of the
 *                point-vertex incidences by vertex, then the chains) - the reference's BITS, at the price of the sort and of
 *                chains as long as the busiest vertex has points;
 *   0            float atomics in arrival order (round-off level run-to-run noise; the measurement baseline). */
int prg_lattice_set_splat_mode(int mode);
int prg_ph_create(prg_ph** out, int device, void* hip_stream);
int prg_ph_destroy(prg_ph* h);
/* init(features, with_blur): points is n x d row-major float32 (the reference passes the transpose). */
int prg_ph_init(prg_ph* h, const float* points_hd, int64_t n, int dim, int with_blur);
/* get_lattice_size() */
int prg_ph_lattice_size(prg_ph* h, int* size);
/* filter(v, start): values n x channels row-major float32 -> out, same shape.  `start` does not exist
 * here because the reference drops it (permutohedral.cpp:608-616). */
int prg_ph_filter(prg_ph* h, const float* values_hd, int channels, float* out_hd);

/* ---- FilterReg rigid point-to-point ------------------------------------------------------------ */
typedef struct prg_filterreg prg_filterreg;
int prg_fr_create(prg_filterreg** out, int device, void* hip_stream);
int prg_fr_destroy(prg_filterreg* h);
/* Clouds as float64 (the reference transforms / scales in float64 before the float32 lattice cast).
 * Replaces: RigidFilterReg(source=...) filterreg.py:150-156 and the `target` of registration, :120. */
int prg_fr_set_source(prg_filterreg* h, const double* source_hd, int64_t m, int dim);
int prg_fr_set_target(prg_filterreg* h, const double* target_hd, int64_t n, int dim);
/* Current rigid transform (rot: row-major 3x3 with the dim x dim block filled, t: 3) and sigma2. */
int prg_fr_set_state(prg_filterreg* h, const double* rot9, const double* t3, double sigma2);
/* E-step: transform the source, build the lattice over [t_source; target] / sigma (rebuilt without blur
 * when it has more than n*alpha vertices), filter (1 | target | |target|^2) and keep the first m rows.
 * Replaces: FilterReg.expectation_step, filterreg.py:78-108 (pt2pt). */
int prg_fr_estep(prg_filterreg* h, double alpha, int* lattice_size, int* with_blur);
/* m0 [m], m1 [m x dim], m2 [m] of the last E-step (float32; any pointer may be NULL). */
int prg_fr_get_estep(prg_filterreg* h, float* m0_hd, float* m1_hd, float* m2_hd);
/* M-step (weighted Kabsch + composition + optional sigma2 update).  out_host[18]: [0..8] rot, [9..11] t,
 * [12] sigma2 of the NEXT iteration, [13] q, [14] number of points with m0 != 0, [15] new (un-clamped) sigma2,
 * [16] 1 if a transform was estimated (0 = every m0 was zero: the reference returns q = None,
 * filterreg.py:167-168), [17] sigma2 this step used.  min_sigma2 >= 0 advances the device state like the driver
 * (`self._sigma2 = max(res.sigma2, min_sigma2)`, filterreg.py:140) so the loop needs no upload per iteration;
 * a negative min_sigma2 leaves the device sigma2 untouched (M-step only).  out_host may be NULL: the M-step is only
 * enqueued, nothing is read back and the host does not wait (a driver with a fixed iteration count and no callbacks
 * reads the state once at the end, prg_fr_get_state).
 * Replaces: RigidFilterReg._maximization_step filterreg.py:158-196 + cc/kabsch.cc:6-109. */
int prg_fr_mstep(prg_filterreg* h, double w, int update_sigma2, double min_sigma2, double* out_host);
/* The device state (synchronises): out_host[20] = the 18 entries of prg_fr_mstep, then [18] q of the last M-step that had
 * anything to fit (an all-zero one sets [13] to NaN and [16] to 0), [19] the number of such M-steps since prg_fr_set_state. */
int prg_fr_get_state(prg_filterreg* h, double* out_host);

/* Point-to-plane objective (filterreg.py:101-105, 183-186): target normals (n x 3 float64, NULL clears) add a
 * 3-channel filter `nx` to the E-step; prg_fr_mstep_pt2pl solves the 6 x 6 twist system
 * (cc/point_to_plane.cc:6-32), applies se3_op.twist_mul (se3_op.py:44-56); out_host as prg_fr_mstep with
 * [13] q = sum w^2 residual^2. */
int prg_fr_set_target_normals(prg_filterreg* h, const double* normals_hd);
int prg_fr_get_nx(prg_filterreg* h, float* nx_hd);
int prg_fr_mstep_pt2pl(prg_filterreg* h, double w, int update_sigma2, double min_sigma2, double* out_host);

/* M-step from caller-supplied E-step arrays - the reference's public FilterReg.maximization_step(t_source, target,
 * estep_res, w, objective_type) -> RigidFilterReg._maximization_step (filterreg.py:110-113, 158-196).  No handle:
 * t_source [m x dim] float64; m0 [m], m1 [m x dim], m2 [m] (NULL = sigma2 is not re-estimated, filterreg.py:190),
 * nx [m x 3] (non-NULL selects the point-to-plane objective, filterreg.py:183-186) float32; n_target enters the
 * outlier constant (filterreg.py:164); rot9 / t3 / sigma2 = trans_p and the current variance.  out_host[18] as
 * prg_fr_mstep ([12] repeats the input sigma2, [15] the new one). */
int prg_fr_mstep_from_arrays(int device, void* hip_stream, const double* t_source_hd, int64_t m, int dim,
                             int64_t n_target, const float* m0_hd, const float* m1_hd, const float* m2_hd,
                             const float* nx_hd, const double* rot9, const double* t3, double sigma2, double w,
                             double* out_host);

/* Weighted Kabsch on float32 clouds: centroids weighted by w, covariance by w^2; rot_host dim x dim
 * row-major, t_host dim.  Replaces: _kabsch.kabsch / kabsch2d (cc/kabsch_py.cc, cc/kabsch.cc:6-109). */
int prg_kabsch_weighted(int device, void* hip_stream, const float* model_hd, const float* target_hd,
                        const float* weight_hd, int64_t n, int dim, double* rot_host, double* t_host);

#ifdef __cplusplus
}
#endif
#endif /* PROBREG_HIP_H */
