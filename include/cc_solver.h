/*
 * cc_solver.h -- C ABI of the MI355X-native reprojection-error LM solver.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/torch types.  It replaces, for
 * the hot path only, what the reference builds out of ceres::Problem / ceres::Solve:
 *   - cc_intrinsics_*  <->  Calibrator::Optimize            (/root/reference/src/calibrator.cpp:221-336)
 *   - cc_rig_*         <->  ExtrinsicsCalibrator::Optimize  (/root/reference/src/extrinsics_calibrator.cpp:86-257)
 *   - cc_distort / cc_undistort <-> Calibrator::Distort / Undistort (calibrator.cpp:118-166)
 * The C++ classes in camera_calibrator_amd/csrc/ (same names and signatures as the reference's
 * calibrator.hh / extrinsics_calibrator.hh) call these entry points; INTEGRATION.md shows the
 * binding a maintainer of the reference would add.
 *
 * All entry points return 0 on success or a negative cc_status; cc_last_error() gives text.
 * The library is HIP-only: there is no CPU fallback, every compute entry point fails with
 * CC_ERR_NO_DEVICE when no gfx950 device is usable.
 *
 * Conventions (identical to the reference): intrinsics order fx fy px py k1 k2 p1 p2 k3
 * (calibrator.cpp:168-179); quaternions w x y z (calibrator.cpp:277-278); observations and 3-D
 * points are float32 as the reference API stores them (types.hh:10-15); parameters are fp64.
 */
#ifndef CC_SOLVER_H
#define CC_SOLVER_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef enum cc_status {
  CC_OK = 0,
  CC_ERR_BAD_ARGUMENT = -1,
  CC_ERR_NO_DEVICE = -2,
  CC_ERR_HIP = -3,
  CC_ERR_COMM = -4,
  CC_ERR_STATE = -5
} cc_status;

/* Termination reasons (mirror Ceres' TerminationType + message for this path). */
enum {
  CC_NO_CONVERGENCE = 0,       /* max_iterations reached */
  CC_CONVERGENCE_GRADIENT = 1,
  CC_CONVERGENCE_PARAMETER = 2,
  CC_CONVERGENCE_FUNCTION = 3,
  CC_FAILURE_INVALID_STEPS = 4,
  CC_MIN_RADIUS = 5,
  CC_FAILURE_EXCHANGE = 6      /* multi-GPU mailbox exchange timed out (solve returns CC_ERR_COMM) */
};

/* Solver options.  cc_options_init() sets the Ceres 2.x defaults overlaid with what the
 * reference sets at calibrator.cpp:314-321 (non-monotonic steps, 100 iterations). Tolerances
 * < 0 disable the corresponding test. */
typedef struct cc_options {
  int32_t max_iterations;                     /* 100 (calibrator.cpp:319); rig default 1000 (extrinsics_calibrator.cpp:211) */
  int32_t use_nonmonotonic_steps;             /* 1 (calibrator.cpp:315) */
  int32_t max_consecutive_nonmonotonic_steps; /* 5 */
  int32_t jacobi_scaling;                     /* 1 */
  int32_t max_consecutive_invalid_steps;      /* 5 */
  int32_t check_interval;                     /* LM iterations enqueued per host poll of the device 'done' flag (4) */
  double function_tolerance;                  /* 1e-6 */
  double gradient_tolerance;                  /* 1e-10 */
  double parameter_tolerance;                 /* 1e-8 */
  double initial_radius;                      /* 1e4 */
  double max_radius;                          /* 1e16 */
  double min_radius;                          /* 1e-32 */
  double min_relative_decrease;               /* 1e-3 */
  double min_lm_diagonal;                     /* 1e-6 */
  double max_lm_diagonal;                     /* 1e32 */
  int32_t use_graph;                          /* 1: replay a captured hipGraph of check_interval iterations */
  int32_t profile_kernels;                    /* 1: bracket every kernel launch with hipEvents (implies no graph) */
} cc_options;

typedef struct cc_iteration {
  double cost;
  double cost_change;
  double model_cost_change;
  double relative_decrease;
  double gradient_max_norm;   /* Ceres': ||x - Plus(x, -g)||_inf (pose blocks with |g_rot| >= 1/4: tangent max-norm) */
  double step_norm;
  double radius;
  int32_t accepted;
  int32_t valid;
} cc_iteration;

enum { CC_K_SWEEP = 0, CC_K_DECIDE = 1, CC_K_ELIM = 2, CC_K_SOLVE = 3, CC_K_ALLREDUCE = 4, CC_K_UPDATE = 5, CC_K_REDUCE = 6, CC_K_COUNT = 8 };

typedef struct cc_summary {
  int32_t iterations;        /* LM iterations executed (accepted + rejected + invalid) */
  int32_t successful_steps;
  int32_t termination;
  int32_t log_len;
  double initial_cost;
  double final_cost;
  double seconds;            /* host wall time of the solve call, device work included */
  cc_iteration* log;         /* caller-provided buffer (may be NULL) */
  int32_t log_capacity;
  int32_t sweeps;            /* Jacobian sweeps launched (1 initial + 1 per valid LM iteration) */
  /* profile_kernels=1 only: per-kernel device time from hipEvents on the solver's stream */
  double kernel_ms[CC_K_COUNT];      /* total ms per kernel kind, over the launches that DID WORK (see kernel_idle_*) */
  int32_t kernel_launches[CC_K_COUNT];
  /* the launches of the last chunk that came after the terminating decision and returned at once (rounds are enqueued
   * check_interval at a time): kept apart so that kernel_ms / kernel_launches is the time of a launch that does work */
  double kernel_idle_ms[CC_K_COUNT];
  int32_t kernel_idle_launches[CC_K_COUNT];
} cc_summary;

void cc_options_init(cc_options* o);
const char* cc_last_error(void);
/* Pinned host memory, cached between calls, for a caller that packs its inputs itself (the C++ classes do: one memcpy per frame
 * of the reference's vector<Points2D> / vector<Points3D> arguments, calibrator.cpp:261-292). Optional: every entry point takes
 * any host pointer. Returns NULL when the allocation fails. */
void* cc_host_staging_acquire(size_t bytes);
void cc_host_staging_release(void* p);
/* Wall milliseconds of the phases of this thread's last cc_intrinsics_estimate / cc_intrinsics_optimize:
 * [0] handle + device arena, [1] upload, [2] Zhang initialisation (estimate only), [3] solve, [4] read-back + teardown. */
void cc_last_call_timing(double out_ms[5]);
const char* cc_version(void);
/* The library keeps a few things between calls so that a caller that re-estimates as images arrive (the reference's workflow
 * builds a fresh solver per call: cam_calibration.py:290-322) does not pay for allocation every time. What is retained, at most:
 * per device four arena pieces (the device memory of a solver handle, each <= 1.25 GB) and pooled device blocks up to 8 GB, one
 * 64 MB scratch block; per process four pinned host staging pieces (each <= 1.25 GB), a few 512-byte pinned blocks, streams, the
 * host worker threads (cc_parallel_for) and the regrouping arrays of the last rig call. This gives the idle ones back (safe at
 * any time from any thread; pieces in use by a running call stay; the next call allocates / starts them again). */
void cc_release_caches(void);
/* Solver form and reruns of this thread's LAST one-shot call (cc_intrinsics_estimate / _optimize / _views / _multi,
 * cc_rig_optimize / _frames / _multi): these calls destroy their handles before returning, so cc_*_solver_status cannot be asked.
 * form: 0 several kernels per iteration, 1 / 2 / 4 the persistent per-solve kernel; reruns > 0: a persistent solve gave up
 * (workgroups not co-resident: another tenant, a tool that serialises kernels, a host thread inside a device-wide runtime call)
 * and the solve was run again in the several-kernel form -- late by 10 ms (intrinsics) / 42 ms (rig) when the first round never
 * came together, 1.3 s in a later round; equal to rounding; note says what the kernel reported. form is what the solve RAN in.
 * A give-up is remembered per device and process, not per handle (these calls make a new handle every time): the next 8 solves
 * AND 2 s on that device do not try the persistent form at all (form 0, reruns 0), then ONE solve probes it; a probe that gives
 * up doubles the window (at most 1024 solves / 10 min), one that completes ends it. The C++ classes expose this as
 * LastSolverReruns() / LastSolverNote() / LastSolverForm(). */
int cc_last_call_solver_status(int32_t* form, int32_t* reruns, char* note, int32_t note_capacity);
/* The library's host worker pool, for a caller's own parallel phases (the C++ classes flatten / fill their arrays with it,
 * extrinsics_calibrator.cpp): fn(ctx, part) for part = 0 .. parts - 1, part 0 on the calling thread, returns when all are done.
 * The threads are created at first use (at most 15), kept for the life of the process, joined by cc_release_caches.
 * One job at a time (a second host thread waits its turn). Re-entry is allowed: a cc_parallel_for -- or any library call that
 * uses the pool (cc_rig_optimize*, cc_rig_get_state) -- made FROM a part runs its parts inline on that thread, one after the
 * other, and cc_release_caches called from a part leaves the pool's threads alone.
 * cc_parallel_parts: how many parts the library itself would use for n items of which a part should hold min_per_part. */
void cc_parallel_for(int32_t parts, void (*fn)(void* ctx, int32_t part), void* ctx);
int32_t cc_parallel_parts(int64_t n, int64_t min_per_part);
int32_t cc_host_pool_threads(void);
/* Number of usable HIP devices (0 if none); never touches the oracle or a CPU path. */
int cc_device_count(void);

/* ---------------------------------------------------------------------------------------------
 * Single-camera intrinsics problem: 9 shared intrinsics + one (q,t) pose per frame.
 * Frames are ragged: frame f owns observations [frame_offsets[f], frame_offsets[f+1]).
 * ------------------------------------------------------------------------------------------- */
typedef struct cc_intrinsics cc_intrinsics;

/* Uploads the observations to HBM (device `device`, a stream of its own). uv: 2N floats (pixels),
 * xyz: 3N floats. */
int cc_intrinsics_create(int32_t device, int64_t n_frames, const int64_t* frame_offsets,
                         const float* uv, const float* xyz, cc_intrinsics** out);
void cc_intrinsics_destroy(cc_intrinsics* h);

/* Sets the current parameters (and remembers them as the initial state for cc_intrinsics_reset).
 * const_mask bit i freezes intrinsic i (ForceDistortionToConstant(d) -> bit d+4, calibrator.cpp:338-340). */
int cc_intrinsics_set_state(cc_intrinsics* h, const double* intr9, uint32_t const_mask,
                            const double* q_wxyz, const double* t_xyz);
int cc_intrinsics_reset(cc_intrinsics* h); /* device-side copy back to the last set_state */
int cc_intrinsics_get_state(cc_intrinsics* h, double* intr9, double* q_wxyz, double* t_xyz);

/* One Jacobian sweep at the current state: per-frame 16x16 Gram blocks
 * G_f = sum_rows v v^T, v = [J_intr(9) J_pose(6) r] (row-major, blocks may be NULL) and the
 * total cost 1/2 sum r^2. */
int cc_intrinsics_eval(cc_intrinsics* h, double* blocks, double* cost);

/* Runs the LM loop on the device from the current state (replaces the ceres::Solve call, calibrator.cpp:323-324).
 * Two forms, same arithmetic, chosen when the handle is created (cc_intrinsics_solver_form):
 *   persistent -- ONE kernel launch per solve: every frame keeps a team of a resident workgroup for the whole solve
 *     (pose, Gram blocks and back-substitution matrix in LDS), a control workgroup takes the trust-region decisions;
 *     used whenever every frame fits a resident team (<= 4 frames per compute unit: 1020 frames on an MI355X) and the
 *     device is the solve's own; CC_INTR_PERSIST=0 disables it;
 *   two kernels per LM iteration (sweep, decide + eliminate + solve), replayed from a captured graph -- any size. */
int cc_intrinsics_solve(cc_intrinsics* h, const cc_options* opt, cc_summary* summary);
/* 0: two kernels per iteration; 1, 2, 4: the persistent kernel with that many frames per workgroup. With an exchange
 * attached the answer is what the ranks agreed on (cc_intrinsics_exchange_attach). */
int cc_intrinsics_solver_form(cc_intrinsics* h);
/* The same, with the history of the handle: *reruns = persistent solves that gave up (a wait inside the kernel timed out:
 * not every workgroup resident) and were run again in the two-kernel form, which the handle then keeps; note = what the
 * last one reported (NUL-terminated, truncated to note_capacity). Any output may be NULL. */
int cc_intrinsics_solver_status(cc_intrinsics* h, int32_t* form, int32_t* reruns, char* note, int32_t note_capacity);

/* Measurement aid: launches `n` steady-state Jacobian sweeps (candidate step + sweep, exactly the
 * kernel an LM iteration runs) back to back on the solver's stream between two hipEvents and
 * returns the average ms per launch. Needs a completed cc_intrinsics_solve with >= 1 iteration.
 * The accepted point is left untouched. */
int cc_intrinsics_profile_sweep(cc_intrinsics* h, int32_t n, double* avg_ms);
/* Measurement aid for the persistent form (one launch = one complete solve): runs `n` solves from the state of the
 * last set_state, each launch bracketed by two hipEvents on the solver's stream; returns the average ms per launch
 * and the evaluations (initial + one per LM iteration) a launch made. CC_ERR_STATE when the handle does not use the
 * persistent form or has an exchange attached. */
int cc_intrinsics_profile_solve(cc_intrinsics* h, const cc_options* opt, int32_t n, double* avg_launch_ms, int32_t* sweeps_per_launch);

/* One-shot convenience: create + set_state + solve + get_state + destroy. This is the call
 * Calibrator::Optimize makes in place of calibrator.cpp:236-324. */
int cc_intrinsics_optimize(const cc_options* opt, int32_t device, int64_t n_frames,
                           const int64_t* frame_offsets, const float* uv, const float* xyz,
                           double* intr9, uint32_t const_mask, double* q_wxyz, double* t_xyz,
                           cc_summary* summary);

/* Calibrator::Estimate in one call (calibrator.cpp:47-68: Zhang initialisation, then Optimize): the observations are
 * uploaded once, cc_zhang_init's kernels run on the handle's device copies, and the solve starts from their K and poses
 * rounded to float -- exactly what the two calls cc_zhang_init + cc_intrinsics_optimize exchange through the class
 * members, so the results are identical. distortion5 (k1 k2 p1 p2 k3) = the object's current distortion (may be NULL:
 * zeros); K_init9 (may be NULL) receives Zhang's K; intr9 / q_wxyz [4F] / t_xyz [3F] receive the optimised state.
 * Needs >= 3 frames with >= 4 points each. */
int cc_intrinsics_estimate(const cc_options* opt, int32_t device, int64_t n_frames, const int64_t* frame_offsets,
                           const float* uv, const float* xyz, const double* distortion5, uint32_t const_mask,
                           float* K_init9, double* intr9, double* q_wxyz, double* t_xyz, cc_summary* summary);
/* The same for views given as separate arrays, the shape of the reference's arguments (vector<Points2D> / vector<Points3D>,
 * calibrator.cpp:47-68): view i has counts[i] points, uv_views[i] = 2 floats per point, xyz_views[i] = 3. The library packs
 * the views into cached pinned memory in pieces and uploads each piece while it packs the next. */
int cc_intrinsics_estimate_views(const cc_options* opt, int32_t device, int64_t n_frames, const float* const* uv_views,
                                 const float* const* xyz_views, const int64_t* counts, const double* distortion5,
                                 uint32_t const_mask, float* K_init9, double* intr9, double* q_wxyz, double* t_xyz,
                                 cc_summary* summary);
/* cc_intrinsics_optimize for views given as separate arrays (Calibrator::Optimize's arguments, calibrator.cpp:70-74). */
int cc_intrinsics_optimize_views(const cc_options* opt, int32_t device, int64_t n_frames, const float* const* uv_views,
                                 const float* const* xyz_views, const int64_t* counts, double* intr9, uint32_t const_mask,
                                 double* q_wxyz, double* t_xyz, cc_summary* summary);

/* Multi-GPU inside ONE process, ONE host thread (SURVEY.md 8(b) thread model): the one-shot call over several
 * devices. Frames are sharded contiguously by observation count (cc_partition_frames), one handle + stream per
 * device, the per-iteration exchange (112 + 16 doubles) goes through mailboxes in peer HBM wired inside the process
 * (hipDeviceEnablePeerAccess, no hipIpc), every device's launches are enqueued before any device is waited for.
 * devices[i] may repeat (several shards on one GPU). n_devices 1..8; with 1 it is cc_intrinsics_optimize.
 * Calibrator::SetDevices selects it behind the class surface. */
int cc_intrinsics_optimize_multi(const cc_options* opt, int32_t n_devices, const int32_t* devices,
                                 int64_t n_frames, const int64_t* frame_offsets, const float* uv,
                                 const float* xyz, double* intr9, uint32_t const_mask, double* q_wxyz,
                                 double* t_xyz, cc_summary* summary);

/* Multi-GPU: one process per GPU, each owning a contiguous shard of frames. Rank 0 calls
 * cc_comm_get_unique_id and broadcasts the 128 bytes (e.g. over torch.distributed); every rank
 * then attaches its handle. Per LM iteration the handles all-reduce (RCCL, sum, fp64) only the
 * reduced shared-parameter blocks (64 doubles after elimination, 64 after the sweep). */
int cc_comm_get_unique_id(uint8_t id[128]);
int cc_intrinsics_comm_init(cc_intrinsics* h, const uint8_t id[128], int32_t rank, int32_t nranks);

/* Multi-GPU within one node, without a collective library in the loop (preferred for <= 8 ranks):
 * the two reductions of an iteration are <= 1 KB and purely latency-bound, so every rank STORES its
 * partial sums straight into a mailbox in each peer's HBM over xGMI and the consumer kernels wait
 * on flags in their own memory (cc_device.hpp). The iteration stays three/four kernels inside one
 * captured graph. Protocol: every rank calls _export (allocates the mailbox, returns its 64-byte
 * hipIpcMemHandle), the caller all-gathers the handles over its control plane (rank order), every
 * rank calls _attach with all nranks*64 bytes (COLLECTIVE for nranks > 1: the ranks agree, through the mailboxes,
 * on the form of the solver they all run; a peer that does not attach within 10 s -> CC_ERR_COMM). All ranks must then call cc_intrinsics_solve /
 * cc_intrinsics_reset the same number of times with the same options. A peer that does not show
 * up within 10 s makes the solve return CC_ERR_COMM (termination CC_FAILURE_EXCHANGE) instead of
 * hanging. Keep the handle alive until every rank has finished its last solve. */
int cc_intrinsics_exchange_export(cc_intrinsics* h, uint8_t handle[64]);
int cc_intrinsics_exchange_attach(cc_intrinsics* h, int32_t rank, int32_t nranks, const uint8_t* handles);

/* Contiguous frame partition balanced by observation count (host logic, no GPU needed).
 * first_frame has nranks+1 entries; rank r owns frames [first_frame[r], first_frame[r+1]). */
int cc_partition_frames(int64_t n_frames, const int64_t* frame_offsets, int32_t nranks,
                        int64_t* first_frame);

/* ---------------------------------------------------------------------------------------------
 * Rig problem (ExtrinsicsCalibrator::Optimize, extrinsics_calibrator.cpp:86-257): one pose per
 * camera (camera_T_rig, shared) and one per observation frame (rig_T_world); world points are
 * constant; residuals in normalised image coordinates with ceres::HuberLoss(huber_a).
 * Observations are grouped by frame: frame f owns [obs_frame_offsets[f], obs_frame_offsets[f+1]).
 * obs_world indexes world_xyz (3 floats per point). cam_frozen[c] != 0 keeps camera c constant;
 * cameras / frames without observations are left untouched (they never enter the problem).
 * Any number of cameras: only cameras that are observed and not frozen own columns of the reduced system
 * (6 each, at most 255 in all, i.e. 42 optimised cameras; at most 64 observed cameras). Up to 127 coordinates the
 * tuned kernels run; beyond, plainer ones with a blocked factorisation of the reduced system (sixteen-column panels, trailing
 * updates on the matrix pipe); both take either exchange (since round 4 the large ones too: column sums posted / all-reduced,
 * one solving block).
 * ------------------------------------------------------------------------------------------- */
typedef struct cc_rig cc_rig;

int cc_rig_create(int32_t device, int64_t n_cams, int64_t n_frames, int64_t n_world,
                  const int64_t* obs_frame_offsets, const uint32_t* obs_cam,
                  const uint64_t* obs_world, const float* obs_uv, const float* world_xyz,
                  const uint8_t* cam_frozen, double huber_a, cc_rig** out);
void cc_rig_destroy(cc_rig* h);
int cc_rig_set_state(cc_rig* h, const double* cam_q, const double* cam_t, const double* frame_q,
                     const double* frame_t);
int cc_rig_reset(cc_rig* h);
int cc_rig_solve(cc_rig* h, const cc_options* opt, cc_summary* summary);
/* Which form cc_rig_solve runs (without profiling):
 *   2 -- the whole solve as ONE launch of the lean persistent kernel (+ its control workgroup's launch): poses only, at most 4
 *        observed cameras, 24 shared coordinates, ~1020 frames, the device to itself; the default where it fits. If its
 *        workgroups cannot all be resident the solve is run again in form 0 (same result, 1.3 s late, once per handle:
 *        the handle stays on form 0 -- cc_rig_solver_status says so and why);
 *   0 -- three kernels per LM iteration (sweep, decision + elimination, reduce + solve step + pose update): any size, any
 *        exchange; CC_RIG_PERSIST=0 forces it.
 * (Round 3's form 1, a glued persistent kernel behind CC_RIG_PERSIST=1, was slower than form 0 everywhere and is gone.) */
int cc_rig_solver_form(cc_rig* h);
/* The same, with the history of the handle: *reruns = lean persistent solves that gave up (not every workgroup resident)
 * and were run again in form 0; note (NUL-terminated, truncated to note_capacity) = the reason of the last one -- round
 * reached, workers started, whether and where the control workgroup ran. Any output may be NULL. */
int cc_rig_solver_status(cc_rig* h, int32_t* form, int32_t* reruns, char* note, int32_t note_capacity);
/* Any output may be NULL. obs_cost[k] = 1/2 rho(|r_k|^2) at the current point, in the caller's
 * observation order (extrinsics_calibrator.cpp:219-225). */
int cc_rig_get_state(cc_rig* h, double* cam_q, double* cam_t, double* frame_q, double* frame_t,
                     double* obs_cost);
/* Total robustified cost at the current point (one sweep; this rank's observations only). */
int cc_rig_eval(cc_rig* h, double* cost);
/* Multi-GPU: each rank creates its handle with a contiguous shard of frames (cc_partition_frames on
 * obs_frame_offsets) and ALL cameras / world points; then attaches as for the intrinsics problem.
 * Per LM iteration: one all-reduce of the reduced (6C)x(6C) system (PC + 32 doubles) and one of
 * 4 + 6C statistics. */
int cc_rig_comm_init(cc_rig* h, const uint8_t id[128], int32_t rank, int32_t nranks);
/* Mailbox exchange for the rig path (same protocol as cc_intrinsics_exchange_*; <= 8 ranks of one node).
 * _attach is collective: every rank must call it (the per-rank "camera seen" flags are summed through
 * the mailboxes, like cc_rig_comm_init does with an all-reduce). */
int cc_rig_exchange_export(cc_rig* h, uint8_t handle[64]);
int cc_rig_exchange_attach(cc_rig* h, int32_t rank, int32_t nranks, const uint8_t* handles);

/* EXTENSION (BASELINE.json configs[3]-[4] name it, nothing in the reference does it):
 * the rig problem with 9 intrinsics (fx fy px py k1 k2 p1 p2 k3, calibrator.cpp:168-179) shared by all
 * cameras and co-optimised with the poses. Observations are PIXELS: the residual is the composition of
 * ReprojectionErrorExtrinsics (extrinsics_calibrator.cpp:51-84) and DistortNormalized/DistortPixels
 * (calibrator.cpp:70-95). huber_a is in pixels, <= 0 switches the loss off. 6 columns per optimised camera + 9
 * must stay <= 255. Multi-GPU through cc_rig_exchange_* / cc_rig_comm_init like the plain rig problem (intrinsics
 * replicated), at every supported size.
 * The handle is a cc_rig: set_state / reset / solve / get_state / eval / destroy are the cc_rig_* calls;
 * cc_rigk_set_intrinsics must be called once before the first solve (const_mask bit i freezes intrinsic i). */
int cc_rigk_create(int32_t device, int64_t n_cams, int64_t n_frames, int64_t n_world,
                   const int64_t* obs_frame_offsets, const uint32_t* obs_cam, const uint64_t* obs_world,
                   const float* obs_uv_pixels, const float* world_xyz, const uint8_t* cam_frozen,
                   double huber_a, cc_rig** out);
int cc_rigk_set_intrinsics(cc_rig* h, const double* intr9, uint32_t const_mask);
int cc_rigk_get_intrinsics(cc_rig* h, double* intr9);
/* Same extension with one set of 9 intrinsics PER CAMERA (BASELINE.json configs[4]: "full intrinsics+extrinsics
 * co-optimisation"): shared block = 6 per optimised camera + 9 per observed camera (<= 255: seventeen cameras; <= 127,
 * eight cameras, for the tuned kernels and for several GPUs).
 * cc_rigk_set_intrinsics sets every camera's set at once, cc_rigk_set_camera_intrinsics one camera's (with its own
 * constant mask, cf. Calibrator::ForceDistortionToConstant); cc_rigk_get_camera_intrinsics reads one set. */
int cc_rigk_create_per_camera(int32_t device, int64_t n_cams, int64_t n_frames, int64_t n_world,
                              const int64_t* obs_frame_offsets, const uint32_t* obs_cam, const uint64_t* obs_world,
                              const float* obs_uv_pixels, const float* world_xyz, const uint8_t* cam_frozen,
                              double huber_a, cc_rig** out);
int cc_rigk_set_camera_intrinsics(cc_rig* h, int64_t camera, const double* intr9, uint32_t const_mask);
int cc_rigk_get_camera_intrinsics(cc_rig* h, int64_t camera, double* intr9);

/* One-shot: the call ExtrinsicsCalibrator::Optimize makes in place of
 * extrinsics_calibrator.cpp:92-225. opt == NULL -> cc_options_init with max_iterations = 1000. */
int cc_rig_optimize(const cc_options* opt, int32_t device, int64_t n_cams, int64_t n_frames,
                    int64_t n_world, const int64_t* obs_frame_offsets, const uint32_t* obs_cam,
                    const uint64_t* obs_world, const float* obs_uv, const float* world_xyz,
                    double* cam_q, double* cam_t, const uint8_t* cam_frozen, double* frame_q,
                    double* frame_t, double huber_a, double* obs_cost, cc_summary* summary);

/* The same for observations the caller keeps FRAME BY FRAME as arrays of records (ExtrinsicsCalibrator's per-frame lists of
 * sightings -- camera id, point id, normalised image point, cost -- extrinsics_calibrator.cpp:33-49 fills them): frame f has
 * counts[f] records starting at frame_records[f], `stride` bytes apart; in a record the camera id and the global world point
 * id are 64-bit unsigned integers at camera_offset / world_offset, the normalised image point two floats at uv_offset, and
 * the robustified cost 0.5 rho(||r||^2) is WRITTEN as a double at cost_offset (negative: not written). The library reads the
 * records in place (no flat copies on the caller's side) and writes the costs back into them. */
typedef struct cc_obs_layout {
  int64_t stride, camera_offset, world_offset, uv_offset, cost_offset;
} cc_obs_layout;
int cc_rig_optimize_frames(const cc_options* opt, int32_t device, int64_t n_cams, int64_t n_frames, int64_t n_world,
                           void* const* frame_records, const int64_t* counts, const cc_obs_layout* layout,
                           const float* world_xyz, double* cam_q, double* cam_t, const uint8_t* cam_frozen,
                           double* frame_q, double* frame_t, double huber_a, cc_summary* summary);

/* The same for observations kept frame by frame as COLUMNS (round 5: what ExtrinsicsCalibrator holds now -- per frame one
 * array of camera ids, one of global world point ids, one of normalised image points, one of costs): column[f] is the first
 * entry of frame f, entries `*_stride` bytes apart (arrays: the stride is the entry's size; records: the stride of the record
 * and column pointers into it -- cc_rig_optimize_frames is exactly that). Ids are unsigned, 4 or 8 bytes wide (`*_width`);
 * the image point two floats; the cost a double, WRITTEN (cost == NULL: not written). A frame without observations may have
 * NULL entries. Why columns: at 8 M observations the solve streams 20 bytes per observation instead of a 40-byte record and
 * the costs come back as sequential stores instead of one partial store per record. */
typedef struct cc_obs_columns {
  const void* const* camera; int64_t camera_stride; int32_t camera_width;
  const void* const* world;  int64_t world_stride;  int32_t world_width;
  const void* const* uv;     int64_t uv_stride;
  void* const* cost;         int64_t cost_stride;
} cc_obs_columns;
int cc_rig_optimize_columns(const cc_options* opt, int32_t device, int64_t n_cams, int64_t n_frames, int64_t n_world,
                            const cc_obs_columns* columns, const int64_t* counts,
                            const float* world_xyz, double* cam_q, double* cam_t, const uint8_t* cam_frozen,
                            double* frame_q, double* frame_t, double huber_a, cc_summary* summary);

/* The same over several devices from one host thread (cf. cc_intrinsics_optimize_multi): frames sharded, cameras and
 * world points replicated; ExtrinsicsCalibrator::SetDevices selects it behind the class surface. */
int cc_rig_optimize_multi(const cc_options* opt, int32_t n_devices, const int32_t* devices, int64_t n_cams,
                          int64_t n_frames, int64_t n_world, const int64_t* obs_frame_offsets,
                          const uint32_t* obs_cam, const uint64_t* obs_world, const float* obs_uv,
                          const float* world_xyz, double* cam_q, double* cam_t, const uint8_t* cam_frozen,
                          double* frame_q, double* frame_t, double huber_a, double* obs_cost,
                          cc_summary* summary);

/* ---------------------------------------------------------------------------------------------
 * Zhang's closed-form initialisation on the device: what Calibrator::Estimate does before calling
 * Optimize (calibrator.cpp:47-66): per-frame DLT homography from (world x,y) -> image
 * (geometry.cpp:70-105), K from the homographies (geometry.cpp:123-177), per-frame pose
 * (geometry.cpp:179-203) as a float quaternion w x y z and translation. Needs >= 3 frames with
 * >= 4 points each. Outputs K9 (row-major 3x3); q_wxyz [4F], t_xyz [3F], homographies [9F] may be
 * NULL. The DLT null vector's sign is fixed so that the board lies in front of the camera.
 * ------------------------------------------------------------------------------------------- */
int cc_zhang_init(int32_t device, int64_t n_frames, const int64_t* frame_offsets, const float* uv,
                  const float* xyz, float* K9, float* q_wxyz, float* t_xyz, float* homographies);

/* ---------------------------------------------------------------------------------------------
 * Point kernels of the Calibrator surface.
 * ------------------------------------------------------------------------------------------- */
/* Calibrator::Distort (calibrator.cpp:157-166): normalised -> pixel coordinates, float arithmetic
 * with the reference's promotions. K: row-major 3x3, dist: k1 k2 p1 p2 k3. */
int cc_distort(int32_t device, const float* K9, const float* dist5, int64_t n,
               const float* xy_normalized, float* uv_out);
/* Calibrator::Undistort (calibrator.cpp:118-155): pixel -> undistorted normalised coordinates. */
int cc_undistort(int32_t device, const float* K9, const float* dist5, int64_t n, const float* uv,
                 float* xy_out);

#ifdef __cplusplus
}
#endif
#endif
