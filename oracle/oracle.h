/*
 * oracle.h -- CPU restatement (fp64, dependency-free C++17 behind a C ABI) of the
 * reprojection-error trust-region LM path of buq2/camera_calibrator.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under camera_calibrator_amd/ may include,
 * link or call this.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, as the checker / timed CPU baseline.
 *
 * Parity status (see DESIGN.md "Oracle"):
 *   - Calibrator path: WEAKLY PINNED.  The reference cannot be built here (Ceres,
 *     Eigen, OpenCV absent) and ships no golden vectors; its only pin for this path
 *     is src/test_calibrator.cpp:45-72 (K within 1 %, one-sided), which
 *     tests/test_oracle_reference_pins.py re-expresses on this restatement.
 *   - ExtrinsicsCalibrator path: PARITY UNPINNED (src/test_extrinsics_calibrator.cpp
 *     asserts nothing).
 *   The arithmetic of the path lives in Ceres Solver (conanfile.txt:3,
 *   "ceres-solver/[>=2.1]", unpinned, not vendored).  This file restates Ceres'
 *   published trust-region Levenberg-Marquardt algorithm (TrustRegionMinimizer,
 *   LevenbergMarquardtStrategy, TrustRegionStepEvaluator, QuaternionManifold,
 *   HuberLoss + Corrector) with an exact block-Schur linear solve in place of
 *   ITERATIVE_SCHUR/CG and without inner iterations, anchored on the reference's
 *   call sites (src/calibrator.cpp:221-336, src/extrinsics_calibrator.cpp:86-257).
 *   Jacobians are analytic and cross-checked against sympy / mpmath / finite
 *   differences / scipy.optimize in tests/.
 */
#ifndef CC_ORACLE_H
#define CC_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Solver options. Defaults (oc_options_init) are the Ceres 2.x defaults overlaid with
 * what the reference sets at src/calibrator.cpp:314-321. */
typedef struct oc_options {
  int32_t max_iterations;                     /* 100 (calibrator.cpp:319); rig: 1000 */
  int32_t use_nonmonotonic_steps;             /* 1   (calibrator.cpp:315) */
  int32_t max_consecutive_nonmonotonic_steps; /* 5   (Ceres default) */
  int32_t jacobi_scaling;                     /* 1   (Ceres default) */
  int32_t max_consecutive_invalid_steps;      /* 5   (Ceres default) */
  int32_t num_threads;                        /* 1   (Ceres default; reference sets none) */
  double function_tolerance;                  /* 1e-6 */
  double gradient_tolerance;                  /* 1e-10 */
  double parameter_tolerance;                 /* 1e-8 */
  double initial_radius;                      /* 1e4 */
  double max_radius;                          /* 1e16 */
  double min_radius;                          /* 1e-32 */
  double min_relative_decrease;               /* 1e-3 */
  double min_lm_diagonal;                     /* 1e-6 */
  double max_lm_diagonal;                     /* 1e32 */
} oc_options;

typedef struct oc_iteration {
  double cost;              /* cost of the accepted point after this iteration */
  double cost_change;       /* x_cost - candidate_cost */
  double model_cost_change;
  double relative_decrease; /* step quality */
  double gradient_max_norm;
  double step_norm;
  double radius;            /* radius after the update */
  int32_t accepted;
  int32_t valid;
} oc_iteration;

enum {
  OC_NO_CONVERGENCE = 0,
  OC_CONVERGENCE_GRADIENT = 1,
  OC_CONVERGENCE_PARAMETER = 2,
  OC_CONVERGENCE_FUNCTION = 3,
  OC_FAILURE_INVALID_STEPS = 4,
  OC_MIN_RADIUS = 5
};

typedef struct oc_summary {
  int32_t iterations;       /* LM iterations executed (accepted + rejected + invalid) */
  int32_t successful_steps;
  int32_t termination;
  int32_t log_len;
  double initial_cost;
  double final_cost;
  double seconds;           /* wall time of the solve loop */
  oc_iteration* log;        /* caller-provided, may be NULL */
  int32_t log_capacity;
  int32_t pad_;
} oc_summary;

void oc_options_init(oc_options* o);

/* ---- single-camera intrinsics model (src/calibrator.cpp:70-95,168-219) ---- */
/* intr = fx fy px py k1 k2 p1 p2 k3 (calibrator.cpp:168-179); q = w x y z; J is 2x15 row-major:
 * 9 intrinsics, 3 rotation tangent (QuaternionManifold), 3 translation. J may be NULL. */
void oc_intrinsics_residual(const double* intr, const double* q, const double* t,
                            const double* X, const double* uv, double* res, double* J);

/* Per-frame 16x16 Gram blocks G_f = sum_rows v v^T, v = [J_intr(9) J_pose(6) r], row-major
 * blocks[F][256] (may be NULL), and total cost = 1/2 sum r^2 (returned). */
double oc_intrinsics_blocks(int64_t n_frames, const int64_t* frame_offsets, const float* uv,
                            const float* xyz, const double* intr, uint32_t const_mask,
                            const double* q, const double* t, double* blocks, int32_t num_threads);

/* Full LM solve (restates Calibrator::Optimize, calibrator.cpp:221-336). intr/q/t are in-out. */
int oc_intrinsics_solve(const oc_options* opt, int64_t n_frames, const int64_t* frame_offsets,
                        const float* uv, const float* xyz, double* intr, uint32_t const_mask,
                        double* q, double* t, oc_summary* summary);

/* Sharded variant used by the world_size>1 CPU (gloo) tests: the rank owns frames
 * [0,n_frames) of its shard; allreduce(ctx, buf, n, op) must reduce buf over ranks in place. */
typedef void (*oc_allreduce_fn)(void* ctx, double* buf, int32_t n, int32_t op /*0 sum, 1 max*/);
int oc_intrinsics_solve_sharded(const oc_options* opt, int64_t n_frames,
                                const int64_t* frame_offsets, const float* uv, const float* xyz,
                                double* intr, uint32_t const_mask, double* q, double* t,
                                oc_summary* summary, oc_allreduce_fn allreduce, void* ctx);

/* Distort (calibrator.cpp:157-166, float arithmetic) / Undistort (calibrator.cpp:118-155). */
void oc_distort(const float* K9, const float* dist5, int64_t n, const float* xy_norm, float* uv_out);
void oc_undistort(const float* K9, const float* dist5, int64_t n, const float* uv, float* xy_out);

/* ---- rig model (src/extrinsics_calibrator.cpp:51-84) ---- */
/* J is 2x12 row-major: camera rot(3) t(3), frame rot(3) t(3) (unscaled by the loss). */
void oc_rig_residual(const double* q_rw, const double* t_rw, const double* q_cr, const double* t_cr,
                     const double* X, const double* uv, double* res, double* J);

int oc_rig_solve(const oc_options* opt, int64_t n_cams, int64_t n_frames, int64_t n_world,
                 const int64_t* obs_frame_offsets, const uint32_t* obs_cam,
                 const uint64_t* obs_world, const float* obs_uv, const float* world_xyz,
                 double* cam_q, double* cam_t, const uint8_t* cam_frozen, double* frame_q,
                 double* frame_t, double huber_a, double* obs_cost, oc_summary* summary);

/* Sharded variant (world_size>1 CPU tests): the rank owns frames [0,n_frames) of its shard with their
 * observations; cameras and world points are replicated. */
int oc_rig_solve_sharded(const oc_options* opt, int64_t n_cams, int64_t n_frames, int64_t n_world,
                         const int64_t* obs_frame_offsets, const uint32_t* obs_cam,
                         const uint64_t* obs_world, const float* obs_uv, const float* world_xyz,
                         double* cam_q, double* cam_t, const uint8_t* cam_frozen, const uint8_t* cam_seen_global,
                         double* frame_q, double* frame_t, double huber_a, double* obs_cost,
                         oc_summary* summary, oc_allreduce_fn allreduce, void* ctx);

/* ---- EXTENSION (nothing in the reference does this): rig poses + 9 intrinsics
 * shared by all cameras, pixel observations = the two functors composed (extrinsics_calibrator.cpp:51-84
 * then calibrator.cpp:70-95). J is 2x21 row-major: camera rot(3) t(3), frame rot(3) t(3), k(9).
 * huber_a <= 0 switches the loss off; kmask bit i freezes intrinsic i (fx fy px py k1 k2 p1 p2 k3). ---- */
void oc_rigk_residual(const double* intr9, const double* q_rw, const double* t_rw, const double* q_cr,
                      const double* t_cr, const double* X, const double* uv, double* res, double* J);
int oc_rigk_solve(const oc_options* opt, int64_t n_cams, int64_t n_frames, int64_t n_world,
                  const int64_t* obs_frame_offsets, const uint32_t* obs_cam, const uint64_t* obs_world,
                  const float* obs_uv_pixels, const float* world_xyz, double* intr9, uint32_t kmask,
                  double* cam_q, double* cam_t, const uint8_t* cam_frozen, double* frame_q, double* frame_t,
                  double huber_a, double* obs_cost, oc_summary* summary);
/* per_camera == 0: as oc_rigk_solve (intr[9], kmask[1]); != 0: one set of intrinsics per camera (intr[n_cams][9],
 * kmask[n_cams]); a set no observation uses is not part of the problem. */
int oc_rigk_solve_sets(const oc_options* opt, int64_t n_cams, int64_t n_frames, int64_t n_world,
                       const int64_t* obs_frame_offsets, const uint32_t* obs_cam, const uint64_t* obs_world,
                       const float* obs_uv_pixels, const float* world_xyz, int32_t per_camera, double* intr,
                       const uint32_t* kmask, double* cam_q, double* cam_t, const uint8_t* cam_frozen,
                       double* frame_q, double* frame_t, double huber_a, double* obs_cost, oc_summary* summary);

/* ---- Zhang initialisation (src/geometry.cpp:70-203, src/calibrator.cpp:47-68) ---- */
void oc_estimate_homography(int64_t n, const float* p1, int32_t stride1, const float* p2,
                            int32_t stride2, float* H9);
void oc_estimate_k_from_homographies(int64_t n, const float* Hs, float* K9);
void oc_recover_extrinsics(const float* Kinv9, const float* H9, float* R9, float* t3);
void oc_fix_rotation_matrix(const float* R9, float* out9);
/* Calibrator::Estimate up to (not including) Optimize: K (float 3x3 row-major), q (w x y z), t */
void oc_zhang_init(int64_t n_frames, const int64_t* frame_offsets, const float* uv,
                   const float* xyz, float* K9, float* q, float* t);

/* ---- synthetic data (src/data_generator.cpp, OpenCV-free) ---- */
typedef struct oc_generator oc_generator;
oc_generator* oc_generator_create(int32_t width, int32_t height);
void oc_generator_destroy(oc_generator* g);
void oc_generator_set_k(oc_generator* g, const float* K9);
void oc_generator_set_distortion(oc_generator* g, const float* dist5);
void oc_generator_set_noise(oc_generator* g, float noise);
/* data_generator.cpp:77-124; returns number of rejected candidates */
int64_t oc_generator_planar(oc_generator* g, int32_t num_p, float* uv, float* xyz);
/* data_generator.cpp:148-184 */
int64_t oc_generator_points(oc_generator* g, int32_t num_p, float* uv, float* xyz);

/* Rig scenario of src/test_extrinsics_calibrator.cpp:48-134 with sizes as arguments.
 * Outputs: cam_T (C x 16, initial/distorted, column-major 4x4), cam_T_true, frame_T (F x 16),
 * world_xyz (F*M x 3), obs (F*M*C): cam, world id, uv. Every camera sees every point. */
void oc_rig_scenario(int32_t n_cams, int32_t n_frames, int32_t pts_per_frame, uint32_t seed,
                     float* cam_T, float* cam_T_true, float* frame_T, float* world_xyz,
                     uint32_t* obs_cam, uint64_t* obs_world, float* obs_uv);

/* Affine3f (column-major 4x4 float) -> double quaternion (w x y z) + translation, as
 * extrinsics_calibrator.cpp:116-130; and back, as :228-256. */
void oc_affine_to_qt(const float* T16, double* q, double* t);
void oc_qt_to_affine(const double* q, const double* t, float* T16);

#ifdef __cplusplus
}
#endif
#endif
