// cc_rig.hip -- multi-camera rig pose bundle adjustment on MI355X (gfx950).
//
// Replaces the ceres::Problem/ceres::Solve block of ExtrinsicsCalibrator::Optimize
// (/root/reference/src/extrinsics_calibrator.cpp:92-225): shared block = one 6-dof pose per camera
// (camera_T_rig), one 6-dof pose per observation frame (rig_T_world), constant world points,
// residuals in normalised coordinates, ceres::HuberLoss(3/500).
//
// Observations are regrouped on the host into (frame, camera) groups. Inside a group both poses are constants and a row's
// frame columns are one 6 x 6 matrix applied to its camera columns (J_frame = J_cam M), so a group's sweep accumulates the Gram
// of [J_cam(6) r] only -- 28 sums on plain FMAs -- and the frame blocks follow from the group's adjoint M. The sweeps:
//   k_rig_sweep_frame  one workgroup per FRAME, a wave per group, compact records [G7 | G_cc M], per-frame sums (the default)
//   k_rig_sweep_adj    one workgroup per group, 16 x 16 tile per group (large rigs: the plain kernels read tiles; tests)
//   k_rig_persist_w    the whole solve in one launch for rigs of at most four observed cameras (lean persistent form)
//   k_rig_sweep_k2     extension (intrinsics): 16-column Gram on plain FMAs, two waves per group, compact records
//   k_rig_sweep_adjk   extension, tiles (large rigs; CC_RIG_K_COMPACT=0)
// (The first formulation -- every row's 13 / 22 columns through the matrix pipe, k_rig_sweep -- lost to the adjoint forms in
// round 2 and was deleted in round 5; git history has it.) Only cameras that are
// observed AND not frozen own columns of the reduced system (any number of frozen / unobserved cameras
// costs nothing, cf. the Parse-keeps-cameras quirk of extrinsics_calibrator.cpp:348-351). Per LM iteration:
//   sweep  : per group, residuals + Jacobian rows + Huber scaling -> Gram sums / records, cost, model term
//   init   : (first evaluation only) Jacobi scaling of the shared block, trust-region state
//   elim   : trust-region decision; one wave per frame: 6x6 Cholesky of the frame block,
//            Z = L^-1 [H_fs | g_f] staged in LDS, Y = L^-T Z for the back-substitution; the Schur
//            products Z^T Z of four frames at a time are accumulated on the matrix cores
//            (v_mfma_f64_16x16x4_f64 over 16-column tiles of the shared block); per-camera sums of the
//            shared-block entries ("direct" sums); one partial row per block
//   reduce+solve : column sums of the partial rows; the LAST block to finish assembles the reduced
//            system in LDS, dense Cholesky, substitutions, gradient / radius tests, camera (and
//            intrinsics) candidates, control block
//   update : per frame, back-substitute the pose step, candidate pose (QuaternionManifold::Plus)
//
// EXTENSION (cc_rigk_*, SURVEY 8f rank 4, no counterpart in the reference): the same kernels, templated
// where it matters, with 9 intrinsics appended to the shared block -- one set shared by all cameras or one
// set per camera -- and pixel observations (k_rig_sweep_k2; k_rig_sweep_adjk for the large-rig kernels).
#include <algorithm>
#include <chrono>
#include <cstring>
#include <numeric>
#include <tuple>
#include <type_traits>
#include <vector>

#include "cc_common.hpp"
#include <mutex>
#include "cc_device.hpp"
#include "cc_persist_dev.hpp"

namespace cc {

constexpr int kRigMaxS = 127;   // shared tangent coordinates: S + 1 (right-hand side) <= 128 = 8 column tiles of 16
constexpr int kRigK = 9;              // intrinsics per set (extension)
constexpr int kRigMaxElimBlocks = 256;
constexpr int kRigDirectPerLane = 24; // direct-sum accumulators per lane in the elim kernel (ND <= 1536)
constexpr int kRigTilesPerWave = 9;   // 36 upper tiles of an 8 x 8 tile grid over 4 waves
constexpr int kDE0 = 27, kDEK = 135;  // direct entries per observed camera (poses only / with intrinsics)

enum { RIG_K_NONE = 0, RIG_K_SHARED = 1, RIG_K_PER_CAMERA = 2 };

struct RigDev {
  int64_t F, N, NG;
  int32_t C;            // cameras as the caller numbers them
  int32_t CO;           // observed cameras (own groups and direct sums)
  int32_t CK;           // intrinsics sets: 0, 1 (shared) or C (per camera)
  int32_t S, SW;        // shared tangent size; SW = S + 1 (right-hand-side column)
  int32_t T, nT, ZS;    // 16-column tiles over SW, upper tile pairs T(T+1)/2, staged Z row stride (doubles)
  int32_t DE, ND;       // direct entries per observed camera, CO * DE
  int32_t PC;           // partial-row length: nT * 256 + ND + 2
  int32_t pc_dir, pc_fail, pc_gmax;
  int32_t nblk;
  const float* uv;        // [N] float2, (frame, camera)-sorted
  const int32_t* widx;    // [N] world point index
  const float* wxyz;      // [3P]
  const float* oxyz;      // [3N] world point of every observation (wxyz gathered once at creation: the sweep's prefetch is one
                          //      independent load per observation instead of an index -> point chain)
  const int64_t* goff;    // [NG+1] observation range of each group
  const int32_t* gframe;  // [NG]
  const int32_t* gcam;    // [NG]
  const int64_t* fgoff;   // [F+1] group range of each frame
  const int4* fwave;      // [F][8] (frames of at most eight groups: k_rig_sweep_frame<.., true>) group j of frame f in ONE
                          // 16-byte record {first observation (lo, hi), observations, camera}: the wave that sweeps it reads this
                          // and nothing else before its observations (fgoff -> goff / gcam -> observations was three loads deep)
  const int4* gk2;        // [NG][2] (with intrinsics) everything k_rig_sweep_k2 needs to know about a group in ONE 32-byte record:
                          // {first observation lo, hi, observations, frame} {camera, intrinsics set, camera held constant, 0} -- the
                          // chain group -> camera -> intrinsics set was three dependent scalar loads deep before the first record load
  const int32_t* cam_goff;   // [C+1]
  const int32_t* cam_glist;  // [NG] groups of each camera
  const uint8_t* cam_fixed;  // [C] pose held constant (frozen, or unobserved by every rank)
  const int32_t* pcol;       // [C] first shared column of the camera's pose, -1: constant
  const int32_t* kcol;       // [C] first shared column of the camera's intrinsics, -1: none
  const int32_t* kset;       // [C] intrinsics set of the camera (0 when shared)
  const int32_t* kscol;      // [max(CK,1)] first shared column of intrinsics set s, -1: not in the problem
  const int32_t* obs_cam;    // [CO] camera id of observed camera j
  const int32_t* fslot;      // [F][CO] group of (frame, observed camera j), -1: the camera does not see the frame
  const int32_t* colinfo;    // [SW] (observed camera j << 8) | (kind << 4) | component; kind 0 pose, 1 own intrinsics,
                             //      2 intrinsics shared by all cameras, 3 right-hand side
  const int16_t* dmap;       // [DE] offset of direct entry e inside a group block
  const int32_t* dent;       // [ND] direct entry e -> (observed camera << 16) | offset inside its group block
  const uint8_t* tile_i;     // [nT] tile pairs (ti <= tj), row-major upper triangle
  const uint8_t* tile_j;
  // where the solve step puts reduced value e (host-built, rig_layout): >= 0 element of the LDS matrix (row * (S + 1)
  // + col), -1 nowhere, <= -2 entry -(d + 2) of the gradient (direct sums) / of the right-hand side (tiles)
  const int32_t* dir_dst;    // [ND]
  const int32_t* dir_next;   // [ND] next direct entry that adds into the same place (an intrinsics set shared by cameras), -1: none
  const int16_t* dir_sa;     // [ND] the two shared columns whose Jacobi scales multiply the entry
  const int16_t* dir_sb;
  const int32_t* tile_dst;   // [nT * 256]
  const int32_t* colpin;     // [S] -1: free column; (intrinsics set << 4) | component: constant when that mask bit is set
  double* cam;      // [2][C][8] q(4) t(3)
  double* pose;     // [2][F][8]
  double* camrec;   // [C][32] R(9) t(3) unscaled step(6)
  double* frec;     // [F][32] R(9) t(3) unscaled step(6)
  double* gblocks;  // [2][NG][gstride]: 256 (poses only) or 3 x 256 (AA | AB | BB tiles, with intrinsics)
  double* gstats;   // [NG][2] cost, model term
  double* fstats;   // [F][2] step^2, |x|^2
  double* ghd0;     // [NG][8] diag of H_cc at the initial point
  double* gcomp;    // compact record per group and buffer for the next sweep's model-cost term: [2][NG][64] 7-column Gram (28),
                    // M (36) (k_rig_sweep_adj); with intrinsics [2][NG][320] 16-column Gram (256), M (36) (k_rig_sweep_adjk).
                    // FRAME form (fmode, k_rig_sweep_frame): [2][NG][64] 7-column Gram G7 (28) and T = G_cc M (36) -- ALL the
                    // elimination reads of a group (direct sums: G7; coupling columns: T); no 16 x 16 tile is written
  double* fsum;     // FRAME form: [2][F][32] per frame H_ff (21, packed lower triangle) and g_f (6), summed over its groups
  int32_t fmode;    // 1: the sweep is k_rig_sweep_frame (one workgroup per FRAME); gstats then holds one row per frame
  int32_t kcm;      // 1: (with intrinsics) the sweep is k_rig_sweep_k2: compact records of kRigRecK doubles in gcomp, no tiles
  double* sp;       // [F][8]
  double* ss;       // [128]
  double* ds;       // [128] scaled shared step
  double* Y;        // [F][6*SW]
  double* partial;  // [nblk][PC]
  double* vec;      // [PC + 32] column sums of the partial rows + one max-gradient slot per rank
  double* vec_stats;  // [4 + S] globally reduced sweep statistics (multi-GPU only)
  int32_t comm, rank, nranks;  // comm != 0: statistics / partial sums pass through an exchange
  P2pDev x;                    // mailbox exchange (cc_device.hpp); x.on != 0 replaces the RCCL all-reduces
  double* shared_stats;  // [4] step^2 and |x|^2 of the shared block (candidate)
  LmCtl* ctl;
  LmCtl* ctl_next;
  LmOpts* opts;
  cc_iteration* log;
  int32_t log_cap;
  unsigned* arrive;      // [16] words right behind *ctl_next (the host reads both in one copy): [0] blocks of k_rig_reduce that
                         // have stored their sums (last-block-done), [1] flag word of the fused pose update, [2] slices of
                         // k_rig_init that have arrived, [3] FAILURE word: an in-kernel wait of k_rig_reduce timed out
  unsigned long long* pub_seq;   // [1] device: chunks published so far
  unsigned long long* host_pub;  // pinned host memory: [0] sequence word the host spins on, [2..19] control block, [20] failure word
  double huber_a;
  double huber_b, huber_2a, huber_ha;   // a^2, 2 a, a / 2 as KERNEL ARGUMENTS: scalar registers from the start (computed in the kernel they
                                        // are vector results the compiler keeps -- and, in k_rig_sweep_frame, spills -- across the passes)
  // EXTENSION: intrinsics in the shared block, pixel observations. kmode 0: off.
  int32_t kmode, gstride;
  int32_t init_slices;   // blocks of k_rig_init that share the diagonal sums of a set of intrinsics common to all cameras
  const uint32_t* kmask;  // [max(CK,1)] bit i: intrinsic i of the set is held constant
  double* intr;       // [2][max(CK,1)][16]
  double* krec;       // [max(CK,1)][32]: candidate intrinsics [0..8], unscaled step [16..24]
  double* ghdk;       // [NG][16] diag of the intrinsics block of each group at the initial point
};

// Progress words of a solve (zeroed with the control block, read by the host only after a wait gave up): launches of each
// kind that have STARTED -- with the mailbox epochs this names the link of a stalled chain (rig_describe_stall).
// NOT in the sweep kernels: one scalar branch and an atomic at the top of k_rig_sweep_adj<1> moved its register allocation
// from 126 VGPRs / 87 SGPRs without spills to 128 / 66 with ten spilled registers -- 56.8 -> 82.2 us per launch at BASELINE
// configs[4] (gpurun_out/r4g, same box A/B against the round-3 library). The statistics / elimination counters bracket the sweep.
enum { RIG_PROG_SWEEP = 0, RIG_PROG_STATS, RIG_PROG_INIT, RIG_PROG_ELIM, RIG_PROG_REDUCE, RIG_PROG_SOLVE, RIG_PROG_UPDATE, RIG_PROG_COUNT };
__device__ __forceinline__ void rig_progress(const RigDev& P, int kind) {
  if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_fetch_add(P.arrive + 4 + kind, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (no return value: nothing waits)
}

__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  return v;
}

// (rsqrt_pos -- 1 / sqrt(d) for a Cholesky pivot or a Huber weight: cc_common.hpp; the caller tests d > 0 && isfinite(d))

// ceres::HuberLoss(a) + Corrector (rho'' <= 0): residual and Jacobian scaled by sqrt(rho').
// Outlier rows: |r| = s y and sqrt(a / |r|) = q rsqrt(q), q = a y, from two refined reciprocal square roots y = rsqrt(s)
// (each within an ulp or two: 17 instructions). The correctly rounded sqrt, divide, sqrt sequence is 34 with its scaling and
// special-case selects, and the WHOLE wave executes it whenever one of its 64 rows is an outlier -- with the reference's
// threshold of three pixels that is most waves of a noisy scenario: 56.3 -> 50.5 us per launch of the frame sweep at
// 8 x 2000 x 500. -DCC_RIG_EXACT_HUBER restores the library calls (A/B and parity forensics).
__device__ __forceinline__ void huber_outlier(double a, double s, double& r, double& sr) {
#ifdef CC_RIG_EXACT_HUBER
  r = sqrt(s);
  sr = sqrt(fmax(2.2250738585072014e-308, a / r));
#else
  const double y = rsqrt_pos(s);
  r = s * y;
  const double q = fmax(2.2250738585072014e-308, a * y);   // (a = 0: the same tiny weight the guarded divide gives)
  sr = q * rsqrt_pos(q);
#endif
}
__device__ __forceinline__ void huber(double a, double s, double& rho, double& sr) {
  const double b = a * a;
  if (s > b) {
    double r;
    huber_outlier(a, s, r, sr);
    rho = 2.0 * a * r - b;
  } else {
    rho = s;
    sr = 1.0;
  }
}

struct RigObs {  // per-observation quantities shared by both rows
  double b0, b1, b2, a0, a1, a2, x, y, iz, ru, rv;
};

// ReprojectionErrorExtrinsics::operator() (extrinsics_calibrator.cpp:51-84)
__device__ __forceinline__ void rig_common(const double* Rf, const double* tf, const double* Rc, const double* tc,
                                           double X0, double X1, double X2, double u, double v, RigObs& o) {
  o.b0 = Rf[0] * X0 + Rf[1] * X1 + Rf[2] * X2;
  o.b1 = Rf[3] * X0 + Rf[4] * X1 + Rf[5] * X2;
  o.b2 = Rf[6] * X0 + Rf[7] * X1 + Rf[8] * X2;
  const double r0 = o.b0 + tf[0], r1 = o.b1 + tf[1], r2 = o.b2 + tf[2];
  o.a0 = Rc[0] * r0 + Rc[1] * r1 + Rc[2] * r2;
  o.a1 = Rc[3] * r0 + Rc[4] * r1 + Rc[5] * r2;
  o.a2 = Rc[6] * r0 + Rc[7] * r1 + Rc[8] * r2;
  const double xc = o.a0 + tc[0], yc = o.a1 + tc[1], zc = o.a2 + tc[2];
  o.iz = 1.0 / zc;
  o.x = xc * o.iz;
  o.y = yc * o.iz;
  o.ru = o.x - u;
  o.rv = o.y - v;
}

// EXTENSION: pixel model behind the rig chain. Given the normalised point (o.x, o.y, o.iz) it returns
// the pixel residuals, B = d residual / d x_cam (what rig_row chains through both poses) and the two
// rows of d residual / d k (DistortNormalized / DistortPixels, calibrator.cpp:70-95).
struct RigKObs {
  double ru, rv, Bu0, Bu1, Bu2, Bv0, Bv1, Bv2;
  double ju[9], jv[9];
};
__device__ __forceinline__ void rigk_obs(const double* k, const RigObs& o, double u, double v, RigKObs& r) {
  const double x = o.x, y = o.y, iz = o.iz;
  const double fx = k[0], fy = k[1];
  const double r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
  const double m = 1.0 + k[4] * r2 + k[5] * r4 + k[8] * r6;
  const double xd = x * m + 2.0 * k[6] * x * y + k[7] * (r2 + 2.0 * x * x);
  const double yd = y * m + 2.0 * k[7] * x * y + k[6] * (r2 + 2.0 * y * y);
  r.ru = fx * xd + k[2] - u;
  r.rv = fy * yd + k[3] - v;
  r.ju[0] = xd; r.ju[1] = 0.0; r.ju[2] = 1.0; r.ju[3] = 0.0;
  r.ju[4] = fx * x * r2; r.ju[5] = fx * x * r4; r.ju[6] = fx * 2.0 * x * y; r.ju[7] = fx * (r2 + 2.0 * x * x); r.ju[8] = fx * x * r6;
  r.jv[0] = 0.0; r.jv[1] = yd; r.jv[2] = 0.0; r.jv[3] = 1.0;
  r.jv[4] = fy * y * r2; r.jv[5] = fy * y * r4; r.jv[6] = fy * (r2 + 2.0 * y * y); r.jv[7] = fy * 2.0 * x * y; r.jv[8] = fy * y * r6;
  const double mp = k[4] + 2.0 * k[5] * r2 + 3.0 * k[8] * r4;
  const double dxx = m + 2.0 * mp * x * x + 2.0 * k[6] * y + 6.0 * k[7] * x;
  const double dxy = 2.0 * mp * x * y + 2.0 * k[6] * x + 2.0 * k[7] * y;
  const double dyy = m + 2.0 * mp * y * y + 2.0 * k[7] * x + 6.0 * k[6] * y;
  r.Bu0 = fx * dxx * iz; r.Bu1 = fx * dxy * iz; r.Bu2 = -(fx * dxx * x + fx * dxy * y) * iz;
  r.Bv0 = fy * dxy * iz; r.Bv1 = fy * dyy * iz; r.Bv2 = -(fy * dxy * x + fy * dyy * y) * iz;
}

// (timing-only builds: the middle workgroup leaves wall-clock marks in shared_stats[32..], scripts/time_rig_reduce.py)
#ifdef CC_RIG_TIMING
#define RSW_MARK(i) do { if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) P.shared_stats[32 + (i)] = (double)wall_clock64(); } while (0)
#else
#define RSW_MARK(i) do { } while (0)
#endif

// ---------------------------------------------------------------------------------------------
// sweep of the reference's problem (poses only) WITHOUT the matrix pipe. Inside a (frame, camera) group both poses are
// constants, and every row's frame columns are one 6 x 6 matrix applied to its camera columns (rig_row):
//     J_frame = J_cam * M,   M = | Rc            0  |   rows: camera (rotation, translation), columns: frame,
//                                | 2 [Rc_i x tf] Rc |   Rc_i = i-th row of the camera rotation, tf = frame translation
// (m = Rc^T B and b x m = Rc^T (a x B) - tf x m, a = Rc (b + tf)). So a group needs the Gram of SEVEN columns
// [J_cam(6) r], 28 unique numbers accumulated by each lane on its own observations with plain FMAs (21 per row: the
// normalised-image rows have one structural zero each) -- against 2 x 16 matrix instructions of 64 cycles per 64
// observations for the 16 x 16 product, half of which is the redundant triangle and a quarter padding. The block
// the other kernels read (same 16 x 16 layout [cam frame r]) is assembled once per group: CF = CC M, FF = M^T CC M,
// g_f = M^T g_c. A fixed camera zeroes its own blocks AFTER the frame blocks were derived from them.
// Rounding differs from the 13-column product by O(eps (|a| / |b|)^2) in the frame-rotation block (a: point relative to
// the camera pose's origin, b: rotated world point).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void untri(int idx, int& i, int& j) {  // packed lower index -> (i >= j)
  i = 0;
  while (tri(i + 1, 0) <= idx) ++i;
  j = idx - tri(i, 0);
}
// 1 / z for the depth of a point in front of the camera (z far from the ends of the exponent range): the hardware
// estimate and two Newton steps, five instructions instead of the twelve of the IEEE division sequence (scaling, fix-up).
// Not correctly rounded: within an ulp or two of 1 / z, so the adjoint sweeps (the default) are not bit-identical to
// the division form that k_rig_obs_cost and the oracle keep (parity is to the stated tolerances under either;
// CC_RIG_EXACT_DIV keeps the division for A/B). Degenerate depths: z = 0 (and z = +-inf) give NaN here (0 * inf inside
// the first fma) where the division gives +-inf / 0. Both are "not finite" to everything downstream -- the candidate
// cost fails isfinite() in lm_trial and counts as DBL_MAX, a Gram block holding either fails the Cholesky's
// `d > 0 && isfinite(d)` test -> invalid step -> the radius shrinks -- so a point that lands on the camera plane is
// rejected the same way in both forms; a select on the result would cost three instructions per observation of ~165.
// A point BEHIND the camera (z < 0) is an ordinary finite value in both.
__device__ __forceinline__ double recip_depth(double z) {
#ifdef CC_RIG_EXACT_DIV
  return 1.0 / z;
#else
  double r = __builtin_amdgcn_rcp(z);
  r = fma(fma(-z, r, 1.0), r, r);
  r = fma(fma(-z, r, 1.0), r, r);
  return r;
#endif
}

template <int SKIP>
__device__ __forceinline__ void adj_accumulate(const double* w, double* acc) {
#pragma unroll
  for (int i = 0; i < 7; ++i) {
#pragma unroll
    for (int j = 0; j <= i; ++j) {
      if (i != SKIP && j != SKIP) acc[i * (i + 1) / 2 + j] = fma(w[i], w[j], acc[i * (i + 1) / 2 + j]);
    }
  }
}

// 32 per-lane values -> their 64-lane sums, value e left in lanes 2e and 2e + 1. Each exchange halves the values a lane
// carries (31 exchanges and adds instead of 32 x 6), and none of them goes through the LDS crossbar: the two widest
// are the lane-swap instructions of gfx950 (v_permlane32_swap: lanes 32..63 of the first register <-> lanes 0..31 of the
// second; v_permlane16_swap: odd 16-lane rows of the first <-> even rows of the second -- after either, first + second
// is the pairwise sum of the first register's values in the lower lanes / even rows and of the second's in the others),
// the rest DPP moves inside a row. Partner masks 32, 16, 8, 7 (half-row mirror), 2, 1 are independent, so every value
// collects all 64 lanes; the lane bit that picks the half kept in each step (5, 4, 3, 2, 1) differs between partners and
// all earlier ones agree.
template <int N>
__device__ __forceinline__ void reduce_swap32(double* p) {
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const auto l = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(p[i]), (unsigned)__double2loint(p[i + N]), false, false);
    const auto h = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(p[i]), (unsigned)__double2hiint(p[i + N]), false, false);
    p[i] = __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
  }
}
template <int N>
__device__ __forceinline__ void reduce_swap16(double* p) {
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const auto l = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(p[i]), (unsigned)__double2loint(p[i + N]), false, false);
    const auto h = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(p[i]), (unsigned)__double2hiint(p[i + N]), false, false);
    p[i] = __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
  }
}
template <int N, int CTRL, int BIT>
__device__ __forceinline__ void reduce_dpp(double* p, int lane) {
  const bool up = (lane & BIT) != 0;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const double lo = p[i], hi = p[i + N];
    const double send = up ? lo : hi, keep = up ? hi : lo;
    p[i] = keep + dpp_f64<CTRL>(send);
  }
}
__device__ __forceinline__ void reduce_scatter32(double* p, int lane) {
  reduce_swap32<16>(p);
  reduce_swap16<8>(p);
  reduce_dpp<4, 0x128, 8>(p, lane);   // row_ror:8
  reduce_dpp<2, 0x141, 4>(p, lane);   // row_half_mirror
  reduce_dpp<1, 0x4E, 2>(p, lane);    // quad_perm:[2,3,0,1]
  p[0] += dpp_f64<0xB1>(p[0]);        // quad_perm:[1,0,3,2]
}

#ifndef CC_RIG_ADJ_WAVES
#define CC_RIG_ADJ_WAVES 4   // waves per SIMD this sweep is compiled for (A/B knob)
#endif
// The sweep of one group as a function: k_rig_sweep_adj (one workgroup per group) and the persistent per-solve kernel
// (k_rig_persist: WL -- "wave-local": the caller is ONE wave of a larger workgroup sweeping the groups of its frame one
// after the other, so there is no workgroup barrier in here, the scratch `lds` is the wave's own, and the camera records
// come from `camrec` -- there: the copy the control workgroup broadcast, in LDS).
constexpr int kRigSweepAdjLds(int NW) { return 64 + 64 + 8 + NW * 32 + 32 + 36; }   // doubles of scratch
// where a group's sweep reads and leaves things: global memory (the stand-alone kernels, k_rig_persist) or the LDS of a
// workgroup that keeps its frames resident (k_rig_persist_w)
struct RigSweepIO {
  const double* camrec;     // [C][32] camera records of the point to evaluate
  const double* frec;       // [32]    record of the group's frame
  const double* comp_old;   // [64]    compact record of the group at the accepted point
  double* block_out;        // [256]   the group's 16 x 16 block at the evaluated point
  double* comp_out;         // [64]    its compact record
  double* stats_out;        // [2]     cost, model-cost term
  double* hd0_out;          // [8]     diagonal of H_cc (first evaluation)
};
__device__ __forceinline__ RigSweepIO rig_sweep_io_global(const RigDev& P, int64_t g, int cur, int dst) {
  return RigSweepIO{P.camrec, P.frec + (size_t)P.gframe[g] * 32, P.gcomp + ((size_t)cur * P.NG + g) * 64,
                    P.gblocks + ((size_t)dst * P.NG + g) * (size_t)P.gstride, P.gcomp + ((size_t)dst * P.NG + g) * 64, P.gstats + g * 2, P.ghd0 + g * 8};
}
template <int NW, bool WL>
__device__ __forceinline__ void rig_sweep_adj_body(const RigDev& P, const int64_t g, const int phase, const int cur, double* lds, const RigSweepIO io) {
  constexpr int NT = NW * 64;      // threads
  double* sm = lds;                    // [64] camera record [0..31], frame record [32..63]
  double* s_old = sm + 64;             // [64] the accepted point's compact record of this group (gcomp)
  double* s_e = s_old + 64;            // [8]  step of the seven columns: e = dc + M_old df, 1
  double* s_red = s_e + 8;             // [NW * 32] per wave: 28 Gram sums, cost, model-cost term
  double* s_g = s_red + NW * 32;       // [32] their totals
  double* s_m = s_g + 32;              // [36] M
  auto sync = [] { if (WL) wave_lds_fence(); else __syncthreads(); };
  int tid_ = WL ? (int)(threadIdx.x & 63) : (int)threadIdx.x;
  if (WL) asm volatile("" : "+v"(tid_));   // (a fresh copy per call: nothing derived from it is hoisted out of the persistent kernel's round loop)
  const int tid = tid_, lane = tid & 63, wave = tid >> 6;
  RSW_MARK(0);
  const int f = P.gframe[g], c = P.gcam[g];
  const int64_t s0 = P.goff[g], s1 = P.goff[g + 1];
  const int dst = phase == 0 ? cur : (cur ^ 1);
  const bool fixed = P.cam_fixed[c] != 0;
  // chunks dealt to the waves starting at wave (g mod NW), observations fetched one pass ahead by unconditional loads:
  // (idle slots re-read the group's first observation)
  const int otid = (((tid >> 6) - (int)(g & (NW - 1))) & (NW - 1)) * 64 + lane;
  const int n = (int)(s1 - s0);                                   // observations of the group (never empty)
  const int wrem = n - (otid >> 6) * 64;
  const int npass = wrem > 0 ? (wrem + NT - 1) / NT : 0;
  // Observations are fetched TWO passes ahead (two register sets, the loop unrolled by two): one pass of the other
  // waves on the SIMD does not cover the memory latency once the whole chip streams. Unconditional loads (idle slots
  // re-read the group's first observation), 32-bit offsets from the group's uniform base.
  const float2* uvg = reinterpret_cast<const float2*>(P.uv) + s0;
  struct F3 { float x, y, z; };
  const F3* xg = reinterpret_cast<const F3*>(P.oxyz) + s0;
  struct ObsRaw { float2 m; F3 X; };
  auto fetch = [&](int k, ObsRaw& r) {
    const int kc = k < n ? k : 0;
    r.m = uvg[kc];
    r.X = xg[kc];
  };
  ObsRaw oa, ob;
  fetch(otid, oa);
  fetch(otid + NT, ob);
  if (tid < 64) {
    // (both buffers hold valid memory: the record of `cur` is read whatever the phase, used only behind phase != 0)
    const double rec = tid < 32 ? io.camrec[c * 32 + tid] : io.frec[tid - 32];
    const double old = io.comp_old[tid];
    sm[tid] = rec;
    s_old[tid] = old;
  }
  sync();
  RSW_MARK(1);
  double acc[32];
#pragma unroll
  for (int e = 0; e < 32; ++e) acc[e] = 0.0;
  // the chain of both poses as one: a = Rc (Rf X + tf) = Rca X + tca (the frame columns are not formed row by row, so
  // the rotated world point is not needed on its own). Uniform addresses: scalar loads, the values live in SGPRs.
  double Rca[9], tca[3], tcs[3];
  {
    const double* cr = io.camrec + (size_t)c * 32;
    const double* fr = io.frec;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
      for (int j = 0; j < 3; ++j) Rca[3 * i + j] = rfl(cr[3 * i] * fr[j] + cr[3 * i + 1] * fr[3 + j] + cr[3 * i + 2] * fr[6 + j]);
      tca[i] = rfl(cr[3 * i] * fr[9] + cr[3 * i + 1] * fr[10] + cr[3 * i + 2] * fr[11]);
      tcs[i] = cr[9 + i];
    }
  }
  if (tid < 48) {   // M[a][b], a = tid >> 3, b = tid & 7 < 6
    const int a = tid >> 3, b = tid & 7;
    const int a3 = a < 3 ? a : a - 3, b3 = b < 3 ? b : b - 3;
    const int b1 = b3 == 2 ? 0 : b3 + 1, b2 = b3 == 0 ? 2 : b3 - 1;
    const double diag = sm[3 * a3 + (b3 < 3 ? b3 : 0)];
    const double cross = 2.0 * (sm[3 * a3 + b1] * sm[32 + 9 + b2] - sm[3 * a3 + b2] * sm[32 + 9 + b1]);   // 2 (Rc_i x tf)_b
    const double v = (a < 3) == (b < 3) ? diag : (a >= 3 ? cross : 0.0);
    if (b < 6) s_m[a * 6 + b] = v;
  }
  // model-cost term of the group at the accepted point, q = d^T g + 1/2 d^T H d with d = [dc df]: every row's J d is
  // J_cam e, e = dc + M_old df (dc = 0 for a fixed camera), so q = 1/2 (e' G7 e' - G7[6][6]) with e' = [e 1]
  if (phase != 0 && tid < 8) {
    double e = tid < 6 ? (fixed ? 0.0 : sm[12 + tid]) : (tid == 6 ? 1.0 : 0.0);
    const int row = tid < 6 ? tid : 0;
#pragma unroll
    for (int b = 0; b < 6; ++b) e = fma(tid < 6 ? s_old[28 + row * 6 + b] : 0.0, sm[32 + 12 + b], e);
    s_e[tid] = e;
  }
  const double ha = P.huber_a;
  RSW_MARK(2);
  struct ObsD { double u, v, X0, X1, X2; };
  auto widen = [](const ObsRaw& r, ObsD& d) { d.u = (double)r.m.x; d.v = (double)r.m.y; d.X0 = (double)r.X.x; d.X1 = (double)r.X.y; d.X2 = (double)r.X.z; };
  auto pass = [&](int k, const ObsD& r) {
    const bool valid = k < n;
    const double X0 = r.X0, X1 = r.X1, X2 = r.X2;
    RigObs o;
    o.a0 = Rca[0] * X0 + Rca[1] * X1 + Rca[2] * X2 + tca[0];
    o.a1 = Rca[3] * X0 + Rca[4] * X1 + Rca[5] * X2 + tca[1];
    o.a2 = Rca[6] * X0 + Rca[7] * X1 + Rca[8] * X2 + tca[2];
    o.iz = recip_depth(o.a2 + tcs[2]);
    o.x = (o.a0 + tcs[0]) * o.iz;
    o.y = (o.a1 + tcs[1]) * o.iz;
    o.ru = o.x - r.u;
    o.rv = o.y - r.v;
    double rho, sr;
    huber(ha, o.ru * o.ru + o.rv * o.rv, rho, sr);
    if (valid) acc[28] += 0.5 * rho;
    if (!valid) sr = 0.0;
    // rows as rig_row forms them, camera columns and residual only, with B_u = (iz, 0, -x iz), B_v = (0, iz, -y iz):
    // u: sr [2 Bu2 a1, 2 (Bu0 a2 - Bu2 a0), -2 Bu0 a1, Bu0, 0, Bu2, ru], v: sr [2 (Bv2 a1 - Bv1 a2), -2 Bv2 a0, 2 Bv1 a0, 0, Bv1, Bv2, rv]
    const double pz = sr * o.iz, qu = -(pz * o.x), qv = -(pz * o.y);     // sr Bu0 = sr Bv1, sr Bu2, sr Bv2
    const double pz2 = pz + pz, qu2 = qu + qu, qv2 = qv + qv;
    double w[7];
    w[0] = qu2 * o.a1; w[1] = pz2 * o.a2 - qu2 * o.a0; w[2] = -(pz2 * o.a1);
    w[3] = pz; w[4] = 0.0; w[5] = qu; w[6] = sr * o.ru;
    adj_accumulate<4>(w, acc);
    w[0] = qv2 * o.a1 - pz2 * o.a2; w[1] = -(qv2 * o.a0); w[2] = pz2 * o.a0;
    w[3] = 0.0; w[4] = pz; w[5] = qv; w[6] = sr * o.rv;
    adj_accumulate<3>(w, acc);
  };
  // (each register set is widened to doubles BEFORE it is refilled: the loaded registers are then dead and the refill
  // reuses them -- a set kept alive across its own refill would be rotated by copies that wait for every load in flight)
  int p = 0;
  for (; p + 1 < npass; p += 2) {    // pairs of passes, no branch inside (a conditional second half brings the copies back)
    const int k = p * NT + otid;
    ObsD d;
    widen(oa, d);
    fetch(k + 2 * NT, oa);
    pass(k, d);
    if (p == 0) RSW_MARK(3);
    widen(ob, d);
    fetch(k + 3 * NT, ob);
    pass(k + NT, d);
  }
  if (p < npass) {
    ObsD d;
    widen(oa, d);
    pass(p * NT + otid, d);
  }
  RSW_MARK(4);
  if (phase != 0 && tid < 27) {
    int i, j;
    untri(tid, i, j);
    acc[29] = (i == j ? 0.5 : 1.0) * s_e[i] * s_e[j] * s_old[tid];
  }
  reduce_scatter32(acc, lane);   // value e in lanes 2e, 2e + 1
  const int ve = lane >> 1;
  if (NW > 1) {
    if ((lane & 1) == 0) s_red[wave * 32 + ve] = acc[0];
    sync();
    if (tid < 32) {
      double t = s_red[tid];
#pragma unroll
      for (int w2 = 1; w2 < NW; ++w2) t += s_red[w2 * 32 + tid];
      s_g[tid] = t;
    }
  } else if ((lane & 1) == 0) {
    s_g[ve] = acc[0];
  }
  sync();
  if (wave != 0) return;
  // The block the other kernels read, [cam frame r]^2 in a 16 x 16 tile, is N^T G7 N with N (7 x 13) = [I6 M 0; 0 0 1]
  // (the identity zeroed for a fixed camera): two matrix products, T = G7 N and N^T T. The first product's result rows
  // k and k + 4 sit in the very lanes that feed them to the second as its B operand.
  const int k0 = lane >> 4, j = lane & 15;
  const double mv0 = s_m[k0 * 6 + (j >= 6 && j < 12 ? j - 6 : 0)];
  const double mv1 = s_m[(k0 < 2 ? k0 + 4 : 0) * 6 + (j >= 6 && j < 12 ? j - 6 : 0)];
  const double id = fixed ? 0.0 : 1.0;
  const double n0 = j < 6 ? (j == k0 ? id : 0.0) : (j < 12 ? mv0 : 0.0);                       // N[k0][j], k0 = 0..3
  const double n1 = k0 < 2 ? (j < 6 ? (j == k0 + 4 ? id : 0.0) : (j < 12 ? mv1 : 0.0))       // N[k0 + 4][j]: rows 4, 5,
                           : (k0 == 2 && j == 12 ? 1.0 : 0.0);                                //   the residual row 6, nothing
  const int gi = j < 7 ? j : 0;
  const int h0 = gi > k0 ? gi : k0, l0 = gi > k0 ? k0 : gi;
  const int k1 = k0 + 4 < 7 ? k0 + 4 : 0;
  const int h1 = gi > k1 ? gi : k1, l1 = gi > k1 ? k1 : gi;
  const double gv0 = s_g[h0 * (h0 + 1) / 2 + l0], gv1 = s_g[h1 * (h1 + 1) / 2 + l1];
  const double a0 = j < 7 ? gv0 : 0.0;                       // G7[j][k0]
  const double a1 = (j < 7 && k0 + 4 < 7) ? gv1 : 0.0;       // G7[j][k0 + 4]
  d4 T = {0.0, 0.0, 0.0, 0.0};
  T = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, n0, T, 0, 0, 0);
  T = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, n1, T, 0, 0, 0);
  d4 B = {0.0, 0.0, 0.0, 0.0};
  B = __builtin_amdgcn_mfma_f64_16x16x4f64(n0, T[0], B, 0, 0, 0);
  B = __builtin_amdgcn_mfma_f64_16x16x4f64(n1, T[1], B, 0, 0, 0);
  double* out = io.block_out;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = k0 + 4 * r;
    out[row * 16 + j] = B[r];
    if (phase == 0 && row < 6 && row == j) io.hd0_out[row] = B[r];  // diag of H_cc
  }
  // compact record of this point for the next sweep's model-cost term: G7 (28) and M (36)
  io.comp_out[lane] = lane < 28 ? s_g[lane < 28 ? lane : 0] : s_m[lane >= 28 ? lane - 28 : 0];
  if (lane == 0) {
    io.stats_out[0] = s_g[28];
    io.stats_out[1] = s_g[29];
  }
  RSW_MARK(5);
}

template <int NW>
__global__ __launch_bounds__(NW * 64, NW == 1 ? CC_RIG_ADJ_WAVES : 3) void k_rig_sweep_adj(RigDev P) {   // (NW > 1: few, large groups -- registers rather than residency)
  __shared__ double s_lds[kRigSweepAdjLds(NW)];
  const LmCtl* ctl = P.ctl;
  const int done = ctl->done, phase = ctl->phase, step_valid = ctl->step_valid, cur = ctl->cur;
  if (done) return;
  if (phase != 0 && !step_valid) return;
  rig_sweep_adj_body<NW, false>(P, blockIdx.x, phase, cur, s_lds, rig_sweep_io_global(P, blockIdx.x, cur, phase == 0 ? cur : (cur ^ 1)));
}

// ---------------------------------------------------------------------------------------------
// FRAME form of the poses-only sweep (round 4; the default of the three-kernel path): one workgroup per FRAME, its NWF waves
// deal the frame's (frame, camera) groups among themselves and sweep them one after the other with the main loop of
// k_rig_sweep_adj (7-column Gram per group, plain FMAs). What changes is everything AROUND that loop:
//   * the frame record is read once per wave, not per group; a wave requests its NEXT group's first observations before it
//     reduces the current one;
//   * nothing but the 28 numbers of G7 leaves the group: the 16 x 16 tile N^T G7 N of k_rig_sweep_adj (four matrix
//     instructions behind ~150 instructions of operand set-up per group, 2 KB written per group and read back by the
//     elimination) is never formed. Wave 0 ends the frame with ONE assembly for all its groups, eight lanes per group:
//     T = G_cc M (the 6 x 6 coupling block the elimination's camera columns are made of), the group's share M^T T of the frame
//     block and M^T g_c of its gradient, added over the groups by lane exchanges. A group's record is [G7 (28) | T (36)],
//     the frame's [H_ff (21) | g_f (6)]: 64 + 32/CO doubles per group where the tile form wrote 256 + 64;
//   * the model-cost term of the step needs no adjoint of the accepted point any more: with the OLD records
//     q = sum_g (1/2 dc' G_cc dc + dc' g_c + dc' T df) + 1/2 df' H_ff df + df' g_f   (dc = 0 for a camera held constant);
//   * cost and model-cost term are ONE row per frame (gstats[f]): the elimination's statistics pass reads F rows instead
//     of NG (BASELINE configs[4]: 2000 instead of 16000 in each of its 256 blocks).
// Arithmetic of a row and of G7: k_rig_sweep_adj's, instruction for instruction (same sums in the same order per lane, same
// butterfly). LDS (dynamic): per group slot 32 doubles (G7, cost), staging 64 per group of an assembly pass, small scratch.
// ---------------------------------------------------------------------------------------------
constexpr int kRigFrameLdsDoubles(int CO) { return CO * 32 + 8 * 64 + 64 + CO * 16; }
// ONE: every wave sweeps at most ONE group (a frame has no more groups than the workgroup has waves: rigs of up to eight
// observed cameras) -- no loop over groups, and the kernel fits the 128 registers of four waves per SIMD like k_rig_sweep_adj<1>
// does; with the loop (more groups than waves) the passes spill 12 - 19 registers at 128, so that variant is compiled for
// three waves per SIMD (141 registers).
template <int NWF, bool ONE>
__global__ __launch_bounds__(NWF * 64, ONE ? 4 : (NWF <= 4 ? 3 : 2)) void k_rig_sweep_frame(RigDev P) {
  extern __shared__ __attribute__((aligned(16))) double sf_lds[];
  double* s_G = sf_lds;                     // [CO][32]  G7 (28), cost (28) of every group of the frame
  double* s_rec = s_G + (size_t)P.CO * 32;  // [8][64]   records of an assembly pass, staged for one coalesced store
  double* s_fr = s_rec + 8 * 64;            // [64]      frame record of the evaluated point (32), then scratch
  double* s_cam = s_fr + 64;                // [CO][16]  per group: its camera's rotation (9), unscaled step (6), held-constant flag -- left here by the
                                            //           wave that sweeps the group, at its START, for the assembly at the workgroup's end (round 5: the
                                            //           assembly fetched them itself, group -> camera -> record, two dependent round trips on the tail)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t f = blockIdx.x;
  const double* fr = P.frec + (size_t)f * 32;
  const double ha = P.huber_a;
  const double hb = P.huber_b, h2a = P.huber_2a, hha = P.huber_ha;
  struct F3 { float x, y, z; };
  struct ObsRaw { float2 m; F3 X; };
  struct ObsD { double u, v, X0, X1, X2; };
  auto widen = [](const ObsRaw& r, ObsD& d) { d.u = (double)r.m.x; d.v = (double)r.m.y; d.X0 = (double)r.X.x; d.X1 = (double)r.X.y; d.X2 = (double)r.X.z; };
  // ---- the wave's groups, one after the other. The first two passes' observations are requested BEFORE the control block
  // is looked at: a launch that returns at once wastes two loads per lane, every other one starts its longest chain
  // (slot record -> observations) with the kernel.
  ObsRaw oa, ob;
  int64_t g0, s0_one = 0;
  int ng, n_one = 0, c_one = 0;
  constexpr bool FW = ONE;
  if (FW) {
    const int4 sl = P.fwave[f * 8 + wave];
    s0_one = (int64_t)(((unsigned long long)(unsigned)sl.y << 32) | (unsigned)sl.x);
    n_one = __builtin_amdgcn_readfirstlane(sl.z);
    c_one = __builtin_amdgcn_readfirstlane(sl.w);
    const float2* uvg = reinterpret_cast<const float2*>(P.uv) + s0_one;
    const F3* xg = reinterpret_cast<const F3*>(P.oxyz) + s0_one;
    const int k0 = lane < n_one ? lane : 0, k1 = lane + 64 < n_one ? lane + 64 : 0;
    oa.m = uvg[k0]; oa.X = xg[k0];
    ob.m = uvg[k1]; ob.X = xg[k1];
    g0 = P.fgoff[f];
    ng = (int)(P.fgoff[f + 1] - g0);            // groups of this frame (0: no observation)
  } else {
    g0 = P.fgoff[f];
    ng = (int)(P.fgoff[f + 1] - g0);
    const int64_t gq = g0 + (wave < ng ? wave : 0);
    const int64_t s0 = P.goff[gq < P.NG ? gq : 0], s1 = P.goff[(gq < P.NG ? gq : 0) + 1];
    const int n = (int)(s1 - s0);
    const float2* uvg = reinterpret_cast<const float2*>(P.uv) + s0;
    const F3* xg = reinterpret_cast<const F3*>(P.oxyz) + s0;
    const int k0 = lane < n ? lane : 0, k1 = lane + 64 < n ? lane + 64 : 0;
    oa.m = uvg[k0]; oa.X = xg[k0];
    ob.m = uvg[k1]; ob.X = xg[k1];
  }
  const LmCtl* ctl = P.ctl;
  const int done = ctl->done, phase = ctl->phase, step_valid = ctl->step_valid, cur = ctl->cur;
  if (done) return;
  if (phase != 0 && !step_valid) return;
  const int dst = phase == 0 ? cur : (cur ^ 1);
  auto sweep_group = [&](const int j) {
    const int64_t g = g0 + j;
    // (the camera index is uniform, and the compiler must know it: the camera record then comes by scalar loads)
    const int c = FW ? c_one : __builtin_amdgcn_readfirstlane(P.gcam[g]);
    const int64_t s0 = FW ? s0_one : P.goff[g];
    const int n = FW ? n_one : (int)(P.goff[g + 1] - s0);
    const int npass = (n + 63) >> 6;
    const float2* uvg = reinterpret_cast<const float2*>(P.uv) + s0;
    const F3* xg = reinterpret_cast<const F3*>(P.oxyz) + s0;
    auto fetch = [&](int k, ObsRaw& r) {
      const int kc = k < n ? k : 0;
      r.m = uvg[kc];
      r.X = xg[kc];
    };
    if (lane < 16) {   // the group's camera for the assembly (one coalesced load, requested with the first observations)
      const double v = lane < 15 ? P.camrec[(size_t)c * 32 + (lane < 9 ? lane : lane + 3)] : (double)P.cam_fixed[c];
      s_cam[j * 16 + lane] = v;
    }
    // the chain of both poses as one: a = Rc (Rf X + tf) = Rca X + tca. Uniform addresses: scalar loads, values in SGPRs.
    double Rca[9], tca[3], tcs[3];
    {
      const double* cr = P.camrec + (size_t)c * 32;
      // (the frame record is re-read -- scalar loads, twelve values -- for every group: read once above the loop, the copies
      // the products need in vector registers stay alive across the whole loop and are spilled: 28 registers of scratch)
      const double* frl = fr;
      asm volatile("" : "+s"(frl));
#pragma unroll
      for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int k = 0; k < 3; ++k) Rca[3 * i + k] = rfl(cr[3 * i] * frl[k] + cr[3 * i + 1] * frl[3 + k] + cr[3 * i + 2] * frl[6 + k]);
        tca[i] = rfl(cr[3 * i] * frl[9] + cr[3 * i + 1] * frl[10] + cr[3 * i + 2] * frl[11]);
        tcs[i] = rfl(cr[9 + i]);
      }
    }
    double acc[32];
#pragma unroll
    for (int e = 0; e < 32; ++e) acc[e] = 0.0;
    auto pass = [&](int k, const ObsD& r) {
      const bool valid = k < n;
      const double X0 = r.X0, X1 = r.X1, X2 = r.X2;
      RigObs o;
      o.a0 = Rca[0] * X0 + Rca[1] * X1 + Rca[2] * X2 + tca[0];
      o.a1 = Rca[3] * X0 + Rca[4] * X1 + Rca[5] * X2 + tca[1];
      o.a2 = Rca[6] * X0 + Rca[7] * X1 + Rca[8] * X2 + tca[2];
      o.iz = recip_depth(o.a2 + tcs[2]);
      o.x = (o.a0 + tcs[0]) * o.iz;
      o.y = (o.a1 + tcs[1]) * o.iz;
      o.ru = o.x - r.u;
      o.rv = o.y - r.v;
      double rho, sr;
      {
        const double ss = o.ru * o.ru + o.rv * o.rv;
        if (ss > hb) {
          double rr;
          huber_outlier(ha, ss, rr, sr);
          rho = h2a * (rr - hha);
        } else { rho = ss; sr = 1.0; }
      }
      if (valid) acc[28] += 0.5 * rho;
      if (!valid) sr = 0.0;
      const double pz = sr * o.iz, qu = -(pz * o.x), qv = -(pz * o.y);
      const double pz2 = pz + pz, qu2 = qu + qu, qv2 = qv + qv;
      double w[7];
      w[0] = qu2 * o.a1; w[1] = pz2 * o.a2 - qu2 * o.a0; w[2] = -(pz2 * o.a1);
      w[3] = pz; w[4] = 0.0; w[5] = qu; w[6] = sr * o.ru;
      adj_accumulate<4>(w, acc);
      w[0] = qv2 * o.a1 - pz2 * o.a2; w[1] = -(qv2 * o.a0); w[2] = pz2 * o.a0;
      w[3] = 0.0; w[4] = pz; w[5] = qv; w[6] = sr * o.rv;
      adj_accumulate<3>(w, acc);
    };
    int p = 0;
    for (; p + 1 < npass; p += 2) {
      const int k = p * 64 + lane;
      ObsD d;
      widen(oa, d);
      fetch(k + 128, oa);
      pass(k, d);
      widen(ob, d);
      fetch(k + 192, ob);
      pass(k + 64, d);
    }
    if (p < npass) {
      ObsD d;
      widen(oa, d);
      pass(p * 64 + lane, d);
    }
    reduce_scatter32(acc, lane);   // value e in lanes 2e, 2e + 1
    if ((lane & 1) == 0 && (lane >> 1) < 29) s_G[j * 32 + (lane >> 1)] = acc[0];
    // (loop form only) the NEXT group's first two passes (requested behind the reduction: held across it, the ten registers of the two sets
    // push the butterfly over the kernel's 128 and spill)
    if (!ONE) {
      const int jn = j + NWF < ng ? j + NWF : j;
      const int64_t gn = g0 + jn;
      const int64_t t0 = P.goff[gn], t1 = P.goff[gn + 1];
      const int nn = (int)(t1 - t0);
      const float2* uvn = reinterpret_cast<const float2*>(P.uv) + t0;
      const F3* xn = reinterpret_cast<const F3*>(P.oxyz) + t0;
      const int k0 = lane < nn ? lane : 0, k1 = lane + 64 < nn ? lane + 64 : 0;
      oa.m = uvn[k0]; oa.X = xn[k0];
      ob.m = uvn[k1]; ob.X = xn[k1];
    }
  };
  // (ONE: straight-line code -- as a loop, even one that runs once, the compiler hoists the Huber constants and lane
  // predicates out of it and keeps them in registers across the passes: sixteen spilled at 128)
  if (ONE) { if (wave < ng) sweep_group(wave); }
  else for (int j = wave; j < ng; j += NWF) sweep_group(j);
  if (tid < 32) s_fr[tid] = fr[tid];
  if (NWF > 1) __syncthreads(); else wave_lds_fence();
  // model-cost term of the step at the accepted point: per group from its OLD record (lanes l < 6: row a = l), 1/2 dc_a (G_cc dc)_a
  // + dc_a g_c,a + dc_a (T df)_a, and the frame's own block from the old frame record. It needs nothing of THIS sweep's sums, so
  // in a workgroup of several waves WAVE 1 forms it while wave 0 assembles the frame (round 5: the old records' round trip was
  // on wave 0's chain, behind the barrier).
  const double* comp_old = P.gcomp + (size_t)cur * P.NG * 64;
  auto model_cost_group = [&](int j, int gi_, int l_) -> double {
    if (phase == 0) return 0.0;
    const bool live = j < ng;
    const int64_t g = g0 + (live ? j : 0);
    const double* crl = s_cam + (size_t)(live ? j : 0) * 16;
    const bool fixed = crl[15] != 0.0;
    const int a = l_ < 6 ? l_ : 0;
    const double* old = comp_old + (size_t)g * 64;
    double gd = 0.0, td = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const int hi = a > k ? a : k, lo = a > k ? k : a;
      gd = fma(old[hi * (hi + 1) / 2 + lo], fixed ? 0.0 : crl[9 + k], gd);
      td = fma(old[28 + a * 6 + k], s_fr[12 + k], td);
    }
    const double dca = fixed ? 0.0 : crl[9 + a];
    (void)gi_;
    return (live && l_ < 6) ? dca * (0.5 * gd + old[21 + a] + td) : 0.0;
  };
  auto model_cost_frame = [&](int gi_, int l_) -> double {
    if (!(phase != 0 && gi_ == 0 && l_ < 6 && ng > 0)) return 0.0;
    const double* fo = P.fsum + ((size_t)cur * P.F + f) * 32;
    double hd = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const int hi = l_ > k ? l_ : k, lo = l_ > k ? k : l_;
      hd = fma(fo[hi * (hi + 1) / 2 + lo], s_fr[12 + k], hd);
    }
    return s_fr[12 + l_] * (0.5 * hd + fo[21 + l_]);
  };
  if (NWF > 1 && wave == 1) {
    int lane_q = threadIdx.x & 63;
    asm volatile("" : "+v"(lane_q));
    const int gq = lane_q >> 3, lq = lane_q & 7;
    double q = model_cost_frame(gq, lq);
    for (int jb = 0; jb < ng; jb += 8) q += model_cost_group(jb + gq, gq, lq);
    q = wave_sum(q);
    if (lane_q == 0) P.gstats[f * 2 + 1] = q;
    return;
  }
  if (wave != 0) return;
  // ---- the frame's assembly: eight lanes per group, eight groups per pass. Lane (gi, l): l < 6 owns column l of T and of the
  // group's share of H_ff; l == 6 the gradient column (g_c -> M^T g_c); l == 7 idles.
  // (every per-lane index below comes from a LAUNDERED copy of the lane id: derived from the original they are hoisted above
  // the group loop and kept alive -- spilled -- across its passes)
  int lane_a = threadIdx.x & 63;
  asm volatile("" : "+v"(lane_a));
  const int gi = lane_a >> 3, l = lane_a & 7;
  const double tf0 = s_fr[9], tf1 = s_fr[10], tf2 = s_fr[11];
  double hsum[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};   // column l of H_ff (l < 6) / g_f (l == 6), over this lane's groups
  double cost = 0.0, qm = 0.0;
  double* comp_dst = P.gcomp + (size_t)dst * P.NG * 64;
  for (int jb = 0; jb < ng; jb += 8) {
    const int j = jb + gi;
    const bool live = j < ng;
    const int64_t g = g0 + (live ? j : 0);
    const double* crl = s_cam + (size_t)(live ? j : 0) * 16;   // [0..8] rotation, [9..14] step, [15] held constant
    const bool fixed = crl[15] != 0.0;
    if (NWF == 1) qm += model_cost_group(j, gi, l);   // (workgroups of several waves: wave 1's, below)
    double Rc[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) Rc[i] = crl[i];
    const double* G = s_G + (size_t)(live ? j : 0) * 32;
    // K[i][b] = 2 (Rc_i x tf)_b: the rotation block of the adjoint's lower left
    double K[9];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      K[3 * i + 0] = 2.0 * (Rc[3 * i + 1] * tf2 - Rc[3 * i + 2] * tf1);
      K[3 * i + 1] = 2.0 * (Rc[3 * i + 2] * tf0 - Rc[3 * i + 0] * tf2);
      K[3 * i + 2] = 2.0 * (Rc[3 * i + 0] * tf1 - Rc[3 * i + 1] * tf0);
    }
    // column l of M: M = [Rc 0; K Rc]
    double mc[6];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int b3 = l < 3 ? l : (l < 6 ? l - 3 : 0);
      const double rv = b3 == 0 ? Rc[3 * k] : (b3 == 1 ? Rc[3 * k + 1] : Rc[3 * k + 2]);
      const double kv = b3 == 0 ? K[3 * k] : (b3 == 1 ? K[3 * k + 1] : K[3 * k + 2]);
      mc[k] = l < 3 ? rv : 0.0;
      mc[3 + k] = l < 3 ? kv : rv;
    }
    // tcol = column l of T = G_cc M (l < 6), or g_c (l == 6)
    double tcol[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      double t = 0.0;
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const int hi = r > k ? r : k, lo = r > k ? k : r;
        t = fma(G[hi * (hi + 1) / 2 + lo], mc[k], t);
      }
      tcol[r] = l < 6 ? t : G[21 + r];
    }
    // u = M^T tcol: u[a'] = sum_k M[k][a'] tcol[k]
    double u[6];
#pragma unroll
    for (int ap = 0; ap < 3; ++ap) {
      u[ap] = Rc[ap] * tcol[0] + Rc[3 + ap] * tcol[1] + Rc[6 + ap] * tcol[2] + K[ap] * tcol[3] + K[3 + ap] * tcol[4] + K[6 + ap] * tcol[5];
      u[3 + ap] = Rc[ap] * tcol[3] + Rc[3 + ap] * tcol[4] + Rc[6 + ap] * tcol[5];
    }
    if (live && l < 7) {
#pragma unroll
      for (int r = 0; r < 6; ++r) hsum[r] += u[r];
    }
    if (live && l == 7) cost += G[28];
    // stage the record [G7 | T] of the pass's groups, then one coalesced store per group
    if (l < 6) {
#pragma unroll
      for (int r = 0; r < 6; ++r) s_rec[gi * 64 + 28 + r * 6 + l] = tcol[r];
    }
    for (int e = l; e < 28; e += 8) s_rec[gi * 64 + e] = G[e];
    wave_lds_fence();
    {
      const int nb = ng - jb < 8 ? ng - jb : 8;
      double* out = comp_dst + (size_t)(g0 + jb) * 64;
      for (int e = lane_a; e < nb * 64; e += 64) out[e] = s_rec[e];
      if (phase == 0 && live && l < 6) P.ghd0[g * 8 + l] = fixed ? 0.0 : G[l * (l + 1) / 2 + l];
    }
    wave_lds_fence();
  }
  // ---- sums over the lanes that share l (the groups of the frame): lane bits 3, 4, 5
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    hsum[r] += __shfl_xor(hsum[r], 8, 64);
    hsum[r] += __shfl_xor(hsum[r], 16, 64);
    hsum[r] += __shfl_xor(hsum[r], 32, 64);
  }
  // frame record of the evaluated point: H_ff (packed lower triangle: entry (r, l), r >= l, from column l) and g_f
  double* fs = P.fsum + ((size_t)dst * P.F + f) * 32;
  if (gi == 0 && l < 6) {
#pragma unroll
    for (int r = 0; r < 6; ++r)
      if (r >= l) fs[r * (r + 1) / 2 + l] = hsum[r];
  }
  if (gi == 0 && l == 6) {
#pragma unroll
    for (int r = 0; r < 6; ++r) fs[21 + r] = hsum[r];
  }
  if (NWF == 1) qm += model_cost_frame(gi, l);
  cost = wave_sum(cost);
  if (NWF == 1) qm = wave_sum(qm);
  if (lane_a == 0) {
    P.gstats[f * 2] = cost;
    if (NWF == 1) P.gstats[f * 2 + 1] = qm;
  }
}

// ---------------------------------------------------------------------------------------------
// EXTENSION (pixel observations through the camera's intrinsics): the same idea with the matrix pipe. A row's 22 columns
// [J_cam(6) J_frame(6) r | J_k(9)] carry only SIXTEEN independent ones, X = [J_cam(6) r J_k(9)]: one 16 x 16 product
// per row set instead of the two of the first formulation (all 22 columns of a row through the matrix pipe: retired in round 5), one staged tile instead of two, no frame columns to
// form. The three tiles the other kernels read are assembled per group from G = X^T X and N (7 x 13) = [I6 M 0; 0 0 1]:
// AA = N^T G[0:7, 0:7] N, AB = N^T G[0:7, 7:16], BB = G[7:16, 7:16]. The compact record kept for the next sweep's
// model-cost term is G itself and M: q = 1/2 (e'^T G e' - G[6][6]), e' = [dc + M_old df, 1, dk].
// ---------------------------------------------------------------------------------------------
#ifndef CC_RIG_ADJK_WAVES
#define CC_RIG_ADJK_WAVES 3   // waves per SIMD the sweep with intrinsics is compiled for (A/B knob)
#endif
constexpr int kRigCompK = 320;   // doubles per group and buffer of the compact record with intrinsics: G (256), M (36)
// NW = waves per workgroup: one when the groups alone fill the chip (every wave then amortises the prologue, the
// cross-lane epilogue and the assembly over all passes of its group and there is no cross-wave reduction), four otherwise.
template <int NW>
__global__ __launch_bounds__(NW * 64, CC_RIG_ADJK_WAVES) void k_rig_sweep_adjk(RigDev P) {
  constexpr int NT = NW * 64, EPT = 256 / NT;
  __shared__ __attribute__((aligned(16))) double s_stage[NW * kStageDoublesPerWave];   // per wave 64 x 16; then the partial products
  __shared__ double sm[96];        // camera record, frame record, intrinsics record (candidate [0..8], step [16..24])
  __shared__ double s_G[256];      // G
  __shared__ double s_mold[36];    // M of the accepted point
  __shared__ double s_e[16];       // e'
  __shared__ double s_m[36];       // M
  __shared__ double s_w[8];        // per wave: model-cost term, cost
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t g = blockIdx.x;
  const int f = P.gframe[g], c = P.gcam[g];
  const int64_t s0 = P.goff[g], s1 = P.goff[g + 1];
  const LmCtl* ctl = P.ctl;
  const int done = ctl->done, phase = ctl->phase, step_valid = ctl->step_valid, cur = ctl->cur;
  if (done) return;
  if (phase != 0 && !step_valid) return;
  const int dst = phase == 0 ? cur : (cur ^ 1);
  const bool fixed = P.cam_fixed[c] != 0;
  const int ks = P.kset[c];
  const int otid = (((tid >> 6) - (int)(g & (NW - 1))) & (NW - 1)) * 64 + lane;
  const int n = (int)(s1 - s0);
  const int wrem = n - (otid >> 6) * 64;
  const int npass = wrem > 0 ? (wrem + NT - 1) / NT : 0;
  const float2* uvg = reinterpret_cast<const float2*>(P.uv) + s0;
  struct F3 { float x, y, z; };
  const F3* xg = reinterpret_cast<const F3*>(P.oxyz) + s0;
  struct ObsRaw { float2 m; F3 X; };
  auto fetch = [&](int k, ObsRaw& r) {
    const int kc = k < n ? k : 0;
    r.m = uvg[kc];
    r.X = xg[kc];
  };
  ObsRaw oa, ob;
  fetch(otid, oa);
  fetch(otid + NT, ob);
  const double* comp_old = P.gcomp + ((size_t)cur * P.NG + g) * kRigCompK;
  {
    // records: camera [0..31], frame [32..63], intrinsics [64..95]; M of the accepted point
    const double r0 = tid < 32 ? P.camrec[c * 32 + tid] : (tid < 64 ? P.frec[(size_t)f * 32 + (tid - 32)] : P.krec[ks * 32 + ((tid - 64) & 31)]);
    const double r1 = P.krec[ks * 32 + (tid & 31)];
    const double mo = comp_old[256 + (tid < 36 ? tid : 0)];
    if (tid < 96) sm[tid] = r0;
    if (NW == 1 && tid < 32) sm[64 + tid] = r1;
    if (tid < 36) s_mold[tid] = mo;
  }
  double g_old[EPT];
#pragma unroll
  for (int e = 0; e < EPT; ++e) g_old[e] = comp_old[tid + e * NT];
  __syncthreads();
  double Rca[9], tca[3], tcs[3], kk[9];
  {
    const double* cr = P.camrec + (size_t)c * 32;
    const double* fr = P.frec + (size_t)f * 32;
    const double* kr = P.krec + (size_t)ks * 32;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
      for (int j = 0; j < 3; ++j) Rca[3 * i + j] = rfl(cr[3 * i] * fr[j] + cr[3 * i + 1] * fr[3 + j] + cr[3 * i + 2] * fr[6 + j]);
      tca[i] = rfl(cr[3 * i] * fr[9] + cr[3 * i + 1] * fr[10] + cr[3 * i + 2] * fr[11]);
      tcs[i] = cr[9 + i];
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) kk[i] = kr[i];
  }
  if (tid < 48) {   // M[a][b], a = tid >> 3, b = tid & 7 < 6 (see k_rig_sweep_adj)
    const int a = tid >> 3, b = tid & 7;
    const int a3 = a < 3 ? a : a - 3, b3 = b < 3 ? b : b - 3;
    const int b1 = b3 == 2 ? 0 : b3 + 1, b2 = b3 == 0 ? 2 : b3 - 1;
    const double diag = sm[3 * a3 + (b3 < 3 ? b3 : 0)];
    const double cross = 2.0 * (sm[3 * a3 + b1] * sm[32 + 9 + b2] - sm[3 * a3 + b2] * sm[32 + 9 + b1]);
    const double v = (a < 3) == (b < 3) ? diag : (a >= 3 ? cross : 0.0);
    if (b < 6) s_m[a * 6 + b] = v;
  } else if (tid < 64) {   // e'
    const int t = tid - 48;
    double e = t < 6 ? (fixed ? 0.0 : sm[12 + t]) : (t == 6 ? 1.0 : sm[64 + 16 + (t - 7)]);
    const int row = t < 6 ? t : 0;
#pragma unroll
    for (int b = 0; b < 6; ++b) e = fma(t < 6 ? s_mold[row * 6 + b] : 0.0, sm[32 + 12 + b], e);
    s_e[t] = e;
  }
  __syncthreads();
  double qterm = 0.0;
  if (phase != 0) {
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int t = tid + e * NT;
      if (t != 6 * 16 + 6) qterm += 0.5 * s_e[t >> 4] * s_e[t & 15] * g_old[e];
    }
  }
  const double ha = P.huber_a;
  const uint32_t kmask = P.kmask[ks];
  double* stage = s_stage + wave * kStageDoublesPerWave;
  d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
  double cost = 0.0;
  struct ObsD { double u, v, X0, X1, X2; };
  auto widen = [](const ObsRaw& r, ObsD& d) { d.u = (double)r.m.x; d.v = (double)r.m.y; d.X0 = (double)r.X.x; d.X1 = (double)r.X.y; d.X2 = (double)r.X.z; };
  auto pass = [&](int k, const ObsD& r) {
    const bool valid = k < n;
    RigObs o;
    o.a0 = Rca[0] * r.X0 + Rca[1] * r.X1 + Rca[2] * r.X2 + tca[0];
    o.a1 = Rca[3] * r.X0 + Rca[4] * r.X1 + Rca[5] * r.X2 + tca[1];
    o.a2 = Rca[6] * r.X0 + Rca[7] * r.X1 + Rca[8] * r.X2 + tca[2];
    o.iz = recip_depth(o.a2 + tcs[2]);
    o.x = (o.a0 + tcs[0]) * o.iz;
    o.y = (o.a1 + tcs[1]) * o.iz;
    RigKObs ko;
    rigk_obs(kk, o, r.u, r.v, ko);
    double rho, sr;
    huber(ha, ko.ru * ko.ru + ko.rv * ko.rv, rho, sr);
    if (valid) cost += 0.5 * rho;
    if (!valid) sr = 0.0;
    double w[16];
    w[0] = sr * (2.0 * (ko.Bu2 * o.a1 - ko.Bu1 * o.a2)); w[1] = sr * (2.0 * (ko.Bu0 * o.a2 - ko.Bu2 * o.a0)); w[2] = sr * (2.0 * (ko.Bu1 * o.a0 - ko.Bu0 * o.a1));
    w[3] = sr * ko.Bu0; w[4] = sr * ko.Bu1; w[5] = sr * ko.Bu2; w[6] = sr * ko.ru;
#pragma unroll
    for (int q = 0; q < 9; ++q) w[7 + q] = (kmask & (1u << q)) ? 0.0 : sr * ko.ju[q];
    stage_row(stage, lane, w);
    wave_lds_fence();
    gram_rows(stage, lane, acc0, acc1);
    wave_lds_fence();
    w[0] = sr * (2.0 * (ko.Bv2 * o.a1 - ko.Bv1 * o.a2)); w[1] = sr * (2.0 * (ko.Bv0 * o.a2 - ko.Bv2 * o.a0)); w[2] = sr * (2.0 * (ko.Bv1 * o.a0 - ko.Bv0 * o.a1));
    w[3] = sr * ko.Bv0; w[4] = sr * ko.Bv1; w[5] = sr * ko.Bv2; w[6] = sr * ko.rv;
#pragma unroll
    for (int q = 0; q < 9; ++q) w[7 + q] = (kmask & (1u << q)) ? 0.0 : sr * ko.jv[q];
    stage_row(stage, lane, w);
    wave_lds_fence();
    gram_rows(stage, lane, acc0, acc1);
    wave_lds_fence();
  };
  int p = 0;
  for (; p + 1 < npass; p += 2) {
    const int k = p * NT + otid;
    ObsD d;
    widen(oa, d);
    fetch(k + 2 * NT, oa);
    pass(k, d);
    widen(ob, d);
    fetch(k + 3 * NT, ob);
    pass(k + NT, d);
  }
  if (p < npass) {
    ObsD d;
    widen(oa, d);
    pass(p * NT + otid, d);
  }
  if (NW > 1) __syncthreads();   // (every wave done with its staging tile before the partial products overwrite them)
  {
    const int slot = (lane >> 4) * 16 + (lane & 15);
    double* dstp = NW > 1 ? s_stage + wave * 256 : s_G;
#pragma unroll
    for (int r = 0; r < 4; ++r) dstp[slot + 64 * r] = acc0[r] + acc1[r];
  }
  const double qw = wave_sum(qterm), cw = wave_sum(cost);
  if (lane == 0) { s_w[wave] = qw; s_w[4 + wave] = cw; }
  __syncthreads();
  if (NW > 1) {
    s_G[tid] = (s_stage[tid] + s_stage[256 + tid]) + (s_stage[512 + tid] + s_stage[768 + tid]);
    __syncthreads();
  }
  double* out = P.gblocks + ((size_t)dst * P.NG + g) * (size_t)P.gstride;
  const int k0 = lane >> 4, j = lane & 15;
  auto role = [&](int what) {
    if (what < 2) {
      const double mv0 = s_m[k0 * 6 + (j >= 6 && j < 12 ? j - 6 : 0)];
      const double mv1 = s_m[(k0 < 2 ? k0 + 4 : 0) * 6 + (j >= 6 && j < 12 ? j - 6 : 0)];
      const double id = fixed ? 0.0 : 1.0;
      const double n0 = j < 6 ? (j == k0 ? id : 0.0) : (j < 12 ? mv0 : 0.0);                       // N[k0][j]
      const double n1 = k0 < 2 ? (j < 6 ? (j == k0 + 4 ? id : 0.0) : (j < 12 ? mv1 : 0.0))       // N[k0 + 4][j]
                               : (k0 == 2 && j == 12 ? 1.0 : 0.0);
      const int k1 = k0 < 3 ? k0 + 4 : 0;
      d4 B = {0.0, 0.0, 0.0, 0.0};
      if (what == 0) {          // AA = N^T (G7 N)
        const int gi = j < 7 ? j : 0;
        const double gv0 = s_G[gi * 16 + k0], gv1 = s_G[gi * 16 + k1];
        const double a0 = j < 7 ? gv0 : 0.0, a1 = (j < 7 && k0 < 3) ? gv1 : 0.0;
        d4 T = {0.0, 0.0, 0.0, 0.0};
        T = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, n0, T, 0, 0, 0);
        T = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, n1, T, 0, 0, 0);
        B = __builtin_amdgcn_mfma_f64_16x16x4f64(n0, T[0], B, 0, 0, 0);
        B = __builtin_amdgcn_mfma_f64_16x16x4f64(n1, T[1], B, 0, 0, 0);
      } else {                  // AB = N^T G[0:7, 7:16]
        const int hj = j < 9 ? 7 + j : 7;
        const double hv0 = s_G[k0 * 16 + hj], hv1 = s_G[k1 * 16 + hj];
        const double h0 = j < 9 ? hv0 : 0.0, h1 = (j < 9 && k0 < 3) ? hv1 : 0.0;
        B = __builtin_amdgcn_mfma_f64_16x16x4f64(n0, h0, B, 0, 0, 0);
        B = __builtin_amdgcn_mfma_f64_16x16x4f64(n1, h1, B, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = k0 + 4 * r;
        out[what * 256 + row * 16 + j] = B[r];
        if (what == 0 && phase == 0 && row < 6 && row == j) P.ghd0[g * 8 + row] = B[r];   // diag of H_cc
      }
    } else if (what == 2) {     // BB = G[7:16, 7:16]
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = k0 + 4 * r;
        const bool in = row < 9 && j < 9;
        const double gv = s_G[(in ? 7 + row : 0) * 16 + (in ? 7 + j : 0)];
        const double val = in ? gv : 0.0;
        out[512 + row * 16 + j] = val;
        if (phase == 0 && row < 9 && row == j) P.ghdk[g * 16 + row] = val;   // diag of H_kk
      }
    } else {
      if (lane == 0) {
        P.gstats[g * 2] = NW > 1 ? (s_w[4] + s_w[5]) + (s_w[6] + s_w[7]) : s_w[4];
        P.gstats[g * 2 + 1] = NW > 1 ? (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]) : s_w[0];
      }
      if (lane < 36) P.gcomp[((size_t)dst * P.NG + g) * kRigCompK + 256 + lane] = s_m[lane];
    }
  };
  if (NW > 1) {
    role(wave);
  } else {
    role(0); role(1); role(2); role(3);
  }
#pragma unroll
  for (int e = 0; e < EPT; ++e) P.gcomp[((size_t)dst * P.NG + g) * kRigCompK + tid + e * NT] = s_G[tid + e * NT];
}

// ---------------------------------------------------------------------------------------------
// EXTENSION, round 5: the sweep with intrinsics WITHOUT the matrix pipe, and with compact records (k_rig_sweep_k2).
// On gfx950 v_mfma_f64_16x16x4_f64 runs at the rate of v_fma_f64 (32 flop per clock and SIMD) and the two share one datapath
// (profiles/r01/microbench_f64.txt), so the 16 x 16 product of k_rig_sweep_adjk pays 2 x 256 multiply-adds per observation
// for 2 x 136 useful ones, plus a staging round trip through LDS per row set. Here the 16-column Gram of
// X = [J_cam(6) r J_k(9)] is accumulated by plain FMAs on its lower triangle, skipping the structural zeros of the pixel
// model (d u / d (fy, py) = d v / d (fx, px) = 0: 105 products per row instead of 256): 210 FMAs per observation.
// 132 accumulators do not fit one lane's 256 registers next to the projection, so a group is swept by TWO waves that split
// the pairs (kK2 below, 66 each) and show each other their rows through LDS: per pass each wave evaluates 64 observations,
// leaves the 28 non-zero row entries of each in LDS, and accumulates ITS pairs over both waves' 128 observations.
// What leaves the group is ONE record of 256 doubles (P.gcomp) -- everything the elimination reads of a group:
//   [0..134]   the direct sums in dmap order: G_cc (21) g_c (6) H_ck (54) H_kk (45) g_k (9)      [135] r^2
//   [136..171] T = G_cc M (camera columns of the frame's coupling)   [172..225] H_fk = M^T H_ck (its intrinsics columns)
//   [226..246] the group's share M^T G_cc M of the frame block       [247..252] its share M^T g_c of the frame gradient
// instead of three 16 x 16 tiles and a 320-double compact record (8.5 KB per group and buffer -> 2 KB). The model-cost term of
// a step is a weighted sum of the OLD record's entries (weights: products of the step's components, k2_qcoef).
// ---------------------------------------------------------------------------------------------
constexpr int kRigRecK = 256;
constexpr int kRkR2 = 135, kRkT = 136, kRkFK = 172, kRkHff = 226, kRkGf = 247, kRkEnd = 253;
constexpr bool k2_in_u(int c) { return c != 8 && c != 10; }   // columns with a non-zero entry in the u row / the v row
constexpr bool k2_in_v(int c) { return c != 7 && c != 9; }
struct K2Split {
  signed char owner[136];   // wave that accumulates pair p = tri(i, j); -1: structurally zero
  unsigned char slot[136];  // its accumulator: 0..63 summed by the butterfly (the sum ends in lane `slot`), 64.. by wave_sum
  unsigned char inv[2][64]; // pair of butterfly slot s of wave w (255: none)
  unsigned char dir[136];   // direct entry e (dmap order, [135] = r^2) -> pair
  unsigned char extra[2][4];   // pair of accumulator 64 + x of wave w (255: none)
  int n[2];
};
constexpr K2Split k2_make_split() {
  K2Split s{};
  int cost[2] = {0, 0};
  s.n[0] = s.n[1] = 0;
  for (int w = 0; w < 2; ++w) for (int k = 0; k < 64; ++k) s.inv[w][k] = 255;
  for (int w = 0; w < 2; ++w) for (int k = 0; k < 4; ++k) s.extra[w][k] = 255;
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j <= i; ++j) {
      const int p = i * (i + 1) / 2 + j;
      const int c = ((k2_in_u(i) && k2_in_u(j)) ? 1 : 0) + ((k2_in_v(i) && k2_in_v(j)) ? 1 : 0);
      if (c == 0) { s.owner[p] = -1; s.slot[p] = 0; continue; }
      const int w = cost[0] <= cost[1] ? 0 : 1;
      s.owner[p] = (signed char)w;
      s.slot[p] = (unsigned char)s.n[w];
      if (s.n[w] < 64) s.inv[w][s.n[w]] = (unsigned char)p; else s.extra[w][s.n[w] - 64] = (unsigned char)p;
      s.n[w]++;
      cost[w] += c;
    }
  // direct entries: columns 0..5 camera, 6 residual, 7..15 intrinsics
  int e = 0;
  for (int i = 0; i < 6; ++i) for (int j = 0; j <= i; ++j) s.dir[e++] = (unsigned char)(i * (i + 1) / 2 + j);
  for (int i = 0; i < 6; ++i) s.dir[e++] = (unsigned char)(6 * 7 / 2 + i);
  for (int i = 0; i < 6; ++i) for (int j = 0; j < 9; ++j) s.dir[e++] = (unsigned char)((7 + j) * (8 + j) / 2 + i);
  for (int i = 0; i < 9; ++i) for (int j = 0; j <= i; ++j) s.dir[e++] = (unsigned char)((7 + i) * (8 + i) / 2 + 7 + j);
  for (int j = 0; j < 9; ++j) s.dir[e++] = (unsigned char)((7 + j) * (8 + j) / 2 + 6);
  s.dir[e++] = (unsigned char)(6 * 7 / 2 + 6);
  return s;
}
constexpr K2Split kK2 = k2_make_split();
static_assert(kK2.n[0] <= 68 && kK2.n[1] <= 68 && kK2.n[0] + kK2.n[1] == 132, "pairs per wave");
constexpr int kK2Acc = 68;

// the products of ONE row (ROW 0: u, 1: v) that wave W accumulates
template <int W, int ROW>
__device__ __forceinline__ void k2_accumulate(const double* w, double* acc) {
#pragma unroll
  for (int i = 0; i < 16; ++i) {
#pragma unroll
    for (int j = 0; j <= i; ++j) {
      const int p = i * (i + 1) / 2 + j;
      if (kK2.owner[p] == W) {
        const int s = kK2.slot[p];
        if (ROW == 0 ? (k2_in_u(i) && k2_in_u(j)) : (k2_in_v(i) && k2_in_v(j))) acc[s] = fma(w[i], w[j], acc[s]);
      }
    }
  }
}
// 64 per-lane values -> their 64-lane sums, value e in lane e (reduce_scatter32 with one more halving in front and the last
// step a halving too)
__device__ __forceinline__ void reduce_scatter64(double* p, int lane) {
  reduce_swap32<32>(p);
  reduce_swap16<16>(p);
  reduce_dpp<8, 0x128, 8>(p, lane);   // row_ror:8
  reduce_dpp<4, 0x141, 4>(p, lane);   // row_half_mirror
  reduce_dpp<2, 0x4E, 2>(p, lane);    // quad_perm:[2,3,0,1]
  reduce_dpp<1, 0xB1, 1>(p, lane);    // quad_perm:[1,0,3,2]
}
// Weight of entry e of a group's OLD record in the model-cost term q = g^T d + 1/2 d^T H d of the step d = (dc, df, dk): a product
// of (at most) two step components and 1/2 or 1. The steps lie where the records lie in LDS (sm: camera [12..17], frame
// [44..49], intrinsics [80..88]); WHICH two, per entry, is a compile-time table (one word per entry: index of the first factor
// | index of the second << 8 | flags << 16; index 255 = the constant one; flag 1 / 2: the first / second factor is a camera
// component and vanishes for a camera held constant; flag 4: weight 1/2; flag 8: weight 0). Computed by index arithmetic per
// thread and group (division by 6 and 9, two triangular-index searches) it was 2.5 k cycles of a group's 27 k.
constexpr unsigned k2_qdesc(int e) {
  int ka = 0, ia = 0, kb = 0, ib = 0, half = 0, zero = 0;
  auto tri_i = [](int idx) { int i = 0; while ((i + 1) * (i + 2) / 2 <= idx) ++i; return i; };
  if (e < 21) { const int i = tri_i(e), j = e - i * (i + 1) / 2; ka = 1; ia = i; kb = 1; ib = j; half = i == j; }
  else if (e < 27) { ka = 1; ia = e - 21; }
  else if (e < 81) { const int i = (e - 27) / 9, j = (e - 27) - 9 * i; ka = 1; ia = i; kb = 3; ib = j; }
  else if (e < 126) { const int i = tri_i(e - 81), j = (e - 81) - i * (i + 1) / 2; ka = 3; ia = i; kb = 3; ib = j; half = i == j; }
  else if (e < 135) { ka = 3; ia = e - 126; }
  else if (e < kRkT) { zero = 1; }
  else if (e < kRkFK) { const int i = (e - kRkT) / 6, j = (e - kRkT) - 6 * i; ka = 1; ia = i; kb = 2; ib = j; }
  else if (e < kRkHff) { const int i = (e - kRkFK) / 9, j = (e - kRkFK) - 9 * i; ka = 2; ia = i; kb = 3; ib = j; }
  else if (e < kRkGf) { const int i = tri_i(e - kRkHff), j = (e - kRkHff) - i * (i + 1) / 2; ka = 2; ia = i; kb = 2; ib = j; half = i == j; }
  else if (e < kRkEnd) { ka = 2; ia = e - kRkGf; }
  else zero = 1;
  const unsigned xa = ka == 0 ? 255u : (unsigned)((ka == 1 ? 12 : (ka == 2 ? 44 : 80)) + ia);
  const unsigned xb = kb == 0 ? 255u : (unsigned)((kb == 1 ? 12 : (kb == 2 ? 44 : 80)) + ib);
  return xa | (xb << 8) | ((unsigned)((ka == 1 ? 1 : 0) | (kb == 1 ? 2 : 0) | (half ? 4 : 0) | (zero ? 8 : 0)) << 16);
}
// The tables a LANE indexes at run time, as one array of words that a workgroup copies into LDS once, under its first round
// trip (read from device memory where they are needed -- behind the lane sums, in the record assembly -- each was a memory
// round trip on the tail's chain): [0..255] k2_qdesc, [256..287] inv (bytes), [288..321] dir (bytes), [322..323] extra (bytes).
constexpr int kK2TabWords = 324;
struct K2Tab { unsigned w[kK2TabWords]; };
constexpr K2Tab k2_make_tab() {
  K2Tab t{};
  for (int e = 0; e < 256; ++e) t.w[e] = k2_qdesc(e);
  const K2Split s = k2_make_split();
  for (int i = 0; i < 128; ++i) t.w[256 + i / 4] |= (unsigned)s.inv[i / 64][i % 64] << (8 * (i % 4));
  for (int i = 0; i < 136; ++i) t.w[288 + i / 4] |= (unsigned)s.dir[i] << (8 * (i % 4));
  for (int i = 0; i < 8; ++i) t.w[322 + i / 4] |= (unsigned)s.extra[i / 4][i % 4] << (8 * (i % 4));
  return t;
}
__device__ const K2Tab kK2Tab = k2_make_tab();
__device__ __forceinline__ int k2_tab_byte(const unsigned* tab, int word0, int i) { return (int)((tab[word0 + (i >> 2)] >> (8 * (i & 3))) & 255u); }
__device__ __forceinline__ double k2_qcoef(const unsigned* tab, int e, const double* sm, double cam_on) {
  const unsigned d = tab[e];
  const int xa = (int)(d & 255u), xb = (int)((d >> 8) & 255u);
  const unsigned fl = d >> 16;
  double fa = sm[xa == 255 ? 0 : xa], fb = sm[xb == 255 ? 0 : xb];
  fa = xa == 255 ? 1.0 : ((fl & 1u) ? fa * cam_on : fa);
  fb = xb == 255 ? 1.0 : ((fl & 2u) ? fb * cam_on : fb);
  const double w = (fl & 8u) ? 0.0 : ((fl & 4u) ? 0.5 : 1.0);
  return w * fa * fb;
}

#ifndef CC_RIG_K2_WAVES
#define CC_RIG_K2_WAVES 2   // waves per SIMD the kernel is compiled for (256 registers)
#endif
// A workgroup (two waves) sweeps groups blockIdx.x, blockIdx.x + gridDim.x, ... one after the other (the grid is four workgroups
// per compute unit: what fits next to the 248 registers). What a group needs before its first pass -- its indices, then its
// records, old record and first observations: two dependent round trips of ~2 us each under load, a third of a workgroup's life
// at 500 observations per group when every group was a workgroup of its own (profiles/r05/k2_stage_marks.jsonl) -- is requested
// during the PREVIOUS group: the indices at its start, the rest right behind its main loop, under its lane sums and assembly.
struct K2Group {   // what is known about a group before its sweep starts
  int f, c, ks, n, fixed;
  int64_t s0;
  uint32_t kmask;
};
__global__ __launch_bounds__(128, CC_RIG_K2_WAVES) void k_rig_sweep_k2(RigDev P) {
  __shared__ __attribute__((aligned(16))) d2 s_rows[2 * 14 * 64];   // [wave][q][lane]: u row entries (q < 7), v row entries
  __shared__ double sm[96];        // camera record, frame record, intrinsics record (candidate [0..8], step [16..24])
  __shared__ double s_G[144];      // lower triangle of G by pair, [136] cost of wave 0, [137] of wave 1, [138..139] model-cost sums
  __shared__ double s_m[36];       // M
  __shared__ double s_T[128];      // T (36) | H_fk (54) | g_f share (6) | H_ff share (21)
  __shared__ long long s_next;     // persistent grid: the group this workgroup sweeps next
  __shared__ unsigned s_tab[kK2TabWords];   // the lane-indexed tables (kK2Tab), staged once
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const LmCtl* ctl = P.ctl;
  const int done = ctl->done, phase = ctl->phase, step_valid = ctl->step_valid, cur = ctl->cur;
  if (done) return;
  if (phase != 0 && !step_valid) return;
  const int dst = phase == 0 ? cur : (cur ^ 1);
  {   // (the first group's loop-top barrier is in front of every read of the tables)
    const unsigned t0 = kK2Tab.w[tid], t1 = kK2Tab.w[tid + 128], t2 = kK2Tab.w[tid + 256 < kK2TabWords ? tid + 256 : 0];
    s_tab[tid] = t0; s_tab[tid + 128] = t1;
    if (tid + 256 < kK2TabWords) s_tab[tid + 256] = t2;
  }
  const int64_t NG = P.NG, stride = gridDim.x;
  const bool dynamic = stride < NG;   // (uniform) arrive[14] work counter, arrive[15] workgroups that have left: zeroed at the start of a solve
                                      // and by the last workgroup of every launch to leave (every fetch of the launch is over by then)
#ifdef CC_RIG_K2_TIMING   // (timing-only build: shader-clock cycles per phase, summed over the groups and passes of wave 0 of the middle workgroup -> shared_stats[40..])
  long long k2t[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  long long k2last = clock64();
  const long long k2wall0 = wall_clock64();
  int k2groups = 0;
#define K2_T(i) do { const long long now_ = clock64(); k2t[i] += now_ - k2last; k2last = now_; } while (0)
#else
#define K2_T(i) do { } while (0)
#endif
  struct F3 { float x, y, z; };
  struct ObsRaw { float2 m; F3 X; };
  struct ObsD { double u, v, X0, X1, X2; };
  auto widen = [](const ObsRaw& r, ObsD& d) { d.u = (double)r.m.x; d.v = (double)r.m.y; d.X0 = (double)r.X.x; d.X1 = (double)r.X.y; d.X2 = (double)r.X.z; };
  auto group_indices = [&](int64_t g, K2Group& q) {   // ONE round trip: the group's record (then the mask of its intrinsics set)
    const int64_t gc = g < NG ? g : 0;
    const int4 a = P.gk2[2 * gc], b = P.gk2[2 * gc + 1];
    q.s0 = (int64_t)(((unsigned long long)(unsigned)a.y << 32) | (unsigned)a.x);
    q.n = a.z; q.f = a.w;
    q.c = b.x; q.ks = b.y; q.fixed = b.z;
    q.kmask = P.kmask[q.ks];
  };
  // vector loads of a group: the lane's first observation, one value of the three records, two of the group's old record
  auto group_loads = [&](int64_t g, const K2Group& q, ObsRaw& o0, double& recv, double& old0, double& old1) {
    const float2* uvg = reinterpret_cast<const float2*>(P.uv) + q.s0;
    const F3* xg = reinterpret_cast<const F3*>(P.oxyz) + q.s0;
    const int kc = tid < q.n ? tid : 0;
    o0.m = uvg[kc];
    o0.X = xg[kc];
    recv = tid < 32 ? P.camrec[q.c * 32 + tid] : (tid < 64 ? P.frec[(size_t)q.f * 32 + (tid - 32)] : P.krec[q.ks * 32 + ((tid - 64) & 31)]);
    const double* rec_old = P.gcomp + ((size_t)cur * NG + (g < NG ? g : 0)) * kRigRecK;
    old0 = rec_old[tid];
    old1 = rec_old[tid + 128];
  };
  const double ha = P.huber_a;
  // ONE branch on the wave for the whole loop over groups (each arm its own register allocation: with a branch per row the arms
  // met four times a pass, ~90 register moves each to reconcile them)
  auto sweep = [&](auto wtag) {
  constexpr int W = decltype(wtag)::value;
  int64_t g = blockIdx.x;
  K2Group q;
  group_indices(g, q);
  ObsRaw oa;
  double recv, old0, old1;
  group_loads(g, q, oa, recv, old0, old1);
  d2* mine = s_rows + (size_t)W * 14 * 64 + lane;
  const d2* theirs = s_rows + (size_t)(W ^ 1) * 14 * 64 + lane;
  while (g < NG) {
    // ---- the group's records into LDS; indices of the NEXT group requested
    // (every per-lane index of the group's head and tail comes from a LAUNDERED copy of the thread id: derived from the original
    // they are hoisted out of the loop over groups and kept alive -- spilled -- across its passes)
    int tid_h = threadIdx.x;
    asm volatile("" : "+v"(tid_h));
    if (tid_h < 96) sm[tid_h] = recv;
    // persistent grid (fewer workgroups than groups): the next group comes from a counter, so that a workgroup that was handed
    // cheap groups takes more of them (static strides ended 8 us behind one workgroup per group)
    if (dynamic && tid_h == 0) s_next = (long long)stride + (long long)__hip_atomic_fetch_add(P.arrive + 14, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    lds_barrier();   // (LDS only: __syncthreads() would also wait for the previous group's record STORES, a memory round trip per group in a persistent grid)
    K2_T(8);
    int64_t gn = g + stride;
    if (dynamic) {
      const long long v = s_next;
      gn = (int64_t)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)v));
    }
    K2Group qn = q;
    const bool more = gn < NG;   // (uniform)
    if (more) group_indices(gn, qn);
    const int n = q.n, npass = (n + 127) >> 7;      // (the same for both waves: they meet at two barriers per pass)
    const bool fixed = q.fixed != 0;
    const uint32_t kmask = q.kmask;
    const float2* uvg = reinterpret_cast<const float2*>(P.uv) + q.s0;
    const F3* xg = reinterpret_cast<const F3*>(P.oxyz) + q.s0;
    auto fetch = [&](int k, ObsRaw& r) {
      const int kc = k < n ? k : 0;
      r.m = uvg[kc];
      r.X = xg[kc];
    };
    // the chain of both poses as one (k_rig_sweep_adj), from the records in LDS (uniform addresses), values in scalar registers
    double Rca[9], tca[3], tcs[3], kk[9];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
      for (int j = 0; j < 3; ++j) Rca[3 * i + j] = rfl(sm[3 * i] * sm[32 + j] + sm[3 * i + 1] * sm[32 + 3 + j] + sm[3 * i + 2] * sm[32 + 6 + j]);
      tca[i] = rfl(sm[3 * i] * sm[32 + 9] + sm[3 * i + 1] * sm[32 + 10] + sm[3 * i + 2] * sm[32 + 11]);
      tcs[i] = rfl(sm[9 + i]);
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) kk[i] = rfl(sm[64 + i]);
    K2_T(9);
    if (tid_h < 48) {   // M[a][b], a = tid >> 3, b = tid & 7 < 6 (see k_rig_sweep_adj)
      const int a = tid_h >> 3, b = tid_h & 7;
      const int a3 = a < 3 ? a : a - 3, b3 = b < 3 ? b : b - 3;
      const int b1 = b3 == 2 ? 0 : b3 + 1, b2 = b3 == 0 ? 2 : b3 - 1;
      const double diag = sm[3 * a3 + (b3 < 3 ? b3 : 0)];
      const double cross = 2.0 * (sm[3 * a3 + b1] * sm[32 + 9 + b2] - sm[3 * a3 + b2] * sm[32 + 9 + b1]);
      const double v = (a < 3) == (b < 3) ? diag : (a >= 3 ? cross : 0.0);
      if (b < 6) s_m[a * 6 + b] = v;
    }
    K2_T(10);
    // model-cost term of the step at the accepted point, from the group's old record (two entries per thread)
    double qterm = 0.0;
    if (phase != 0) qterm = k2_qcoef(s_tab, tid_h, sm, fixed ? 0.0 : 1.0) * old0 + k2_qcoef(s_tab, tid_h + 128, sm, fixed ? 0.0 : 1.0) * old1;
    K2_T(0);
    double acc[kK2Acc];
#pragma unroll
    for (int e = 0; e < kK2Acc; ++e) acc[e] = 0.0;
    double cost = 0.0;
    auto pass = [&](int k, const ObsD& r) {
      const bool valid = k < n;
#ifdef CC_RIG_K2_TIMING
      { double t_ = r.u + r.X0; asm volatile("" : "+v"(t_)); K2_T(5); }   // (the wait for this pass's observations, on its own)
#endif
      RigObs o;
      o.a0 = Rca[0] * r.X0 + Rca[1] * r.X1 + Rca[2] * r.X2 + tca[0];
      o.a1 = Rca[3] * r.X0 + Rca[4] * r.X1 + Rca[5] * r.X2 + tca[1];
      o.a2 = Rca[6] * r.X0 + Rca[7] * r.X1 + Rca[8] * r.X2 + tca[2];
      o.iz = recip_depth(o.a2 + tcs[2]);
      o.x = (o.a0 + tcs[0]) * o.iz;
      o.y = (o.a1 + tcs[1]) * o.iz;
      // pixel model (rigk_obs, DistortNormalized / DistortPixels of calibrator.cpp:70-95) with the Huber weight folded into
      // the factors every row entry carries anyway (sr fx, sr fy, sr fx / z, sr fy / z): an entry costs one instruction
      const double x = o.x, y = o.y;
      const double x2 = x * x, y2 = y * y, xy = x * y;
      const double r2 = x2 + y2, r4 = r2 * r2, r6 = r4 * r2;
      const double m = 1.0 + kk[4] * r2 + kk[5] * r4 + kk[8] * r6;
      const double ax = r2 + 2.0 * x2, ay = r2 + 2.0 * y2;
      const double xd = x * m + 2.0 * kk[6] * xy + kk[7] * ax;
      const double yd = y * m + 2.0 * kk[7] * xy + kk[6] * ay;
      const double ru = kk[0] * xd + kk[2] - r.u, rv = kk[1] * yd + kk[3] - r.v;
      const double mp = kk[4] + 2.0 * kk[5] * r2 + 3.0 * kk[8] * r4;
      const double dxx = m + 2.0 * mp * x2 + 2.0 * kk[6] * y + 6.0 * kk[7] * x;
      const double dxy = 2.0 * mp * xy + 2.0 * kk[6] * x + 2.0 * kk[7] * y;
      const double dyy = m + 2.0 * mp * y2 + 2.0 * kk[7] * x + 6.0 * kk[6] * y;
      double rho, sr;
      huber(ha, ru * ru + rv * rv, rho, sr);
      if (valid) cost += 0.5 * rho;
      if (!valid) sr = 0.0;
      const double sfx = sr * kk[0], sfy = sr * kk[1];
      const double a0d = o.a0 + o.a0, a1d = o.a1 + o.a1, a2d = o.a2 + o.a2;   // (2 a: the rotation columns are 2 a x B)
      // Row by row, each used up at once: formed, left in LDS for the other wave (its 14 non-zero entries: u skips columns 8 and
      // 10, v columns 7 and 9), accumulated -- so that only one row of 14 is alive next to the accumulators at any time.
      {
        double w[16];
        const double gz = sfx * o.iz;
        const double B0 = gz * dxx, B1 = gz * dxy, B2 = -(B0 * x + B1 * y);
        w[0] = B2 * a1d - B1 * a2d; w[1] = B0 * a2d - B2 * a0d; w[2] = B1 * a0d - B0 * a1d;
        w[3] = B0; w[4] = B1; w[5] = B2; w[6] = sr * ru;
        const double tx = sfx * x;
        w[7] = sr * xd; w[8] = 0.0; w[9] = sr; w[10] = 0.0;
        w[11] = tx * r2; w[12] = tx * r4; w[13] = tx * (y + y); w[14] = sfx * ax; w[15] = tx * r6;
        if (kmask != 0u) {   // (uniform: intrinsics held constant have no column)
#pragma unroll
          for (int qq = 0; qq < 9; ++qq) w[7 + qq] = (kmask & (1u << qq)) ? 0.0 : w[7 + qq];
        }
        mine[0 * 64] = d2{w[0], w[1]}; mine[1 * 64] = d2{w[2], w[3]}; mine[2 * 64] = d2{w[4], w[5]}; mine[3 * 64] = d2{w[6], w[7]};
        mine[4 * 64] = d2{w[9], w[11]}; mine[5 * 64] = d2{w[12], w[13]}; mine[6 * 64] = d2{w[14], w[15]};
        k2_accumulate<W, 0>(w, acc);
      }
      {
        double w[16];
        const double gz = sfy * o.iz;
        const double B0 = gz * dxy, B1 = gz * dyy, B2 = -(B0 * x + B1 * y);
        w[0] = B2 * a1d - B1 * a2d; w[1] = B0 * a2d - B2 * a0d; w[2] = B1 * a0d - B0 * a1d;
        w[3] = B0; w[4] = B1; w[5] = B2; w[6] = sr * rv;
        const double ty = sfy * y;
        w[7] = 0.0; w[8] = sr * yd; w[9] = 0.0; w[10] = sr;
        w[11] = ty * r2; w[12] = ty * r4; w[13] = sfy * ay; w[14] = ty * (x + x); w[15] = ty * r6;
        if (kmask != 0u) {
#pragma unroll
          for (int qq = 0; qq < 9; ++qq) w[7 + qq] = (kmask & (1u << qq)) ? 0.0 : w[7 + qq];
        }
        mine[7 * 64] = d2{w[0], w[1]}; mine[8 * 64] = d2{w[2], w[3]}; mine[9 * 64] = d2{w[4], w[5]}; mine[10 * 64] = d2{w[6], w[8]};
        mine[11 * 64] = d2{w[10], w[11]}; mine[12 * 64] = d2{w[12], w[13]}; mine[13 * 64] = d2{w[14], w[15]};
        k2_accumulate<W, 1>(w, acc);
      }
      K2_T(1);
      lds_barrier();
      K2_T(2);
      {
        double w[16];
        d2 t;
        t = theirs[0 * 64]; w[0] = t.x; w[1] = t.y;  t = theirs[1 * 64]; w[2] = t.x; w[3] = t.y;
        t = theirs[2 * 64]; w[4] = t.x; w[5] = t.y;  t = theirs[3 * 64]; w[6] = t.x; w[7] = t.y;
        t = theirs[4 * 64]; w[9] = t.x; w[11] = t.y; t = theirs[5 * 64]; w[12] = t.x; w[13] = t.y;
        t = theirs[6 * 64]; w[14] = t.x; w[15] = t.y;
        w[8] = 0.0; w[10] = 0.0;
        k2_accumulate<W, 0>(w, acc);
      }
      {
        double w[16];
        d2 t;
        t = theirs[7 * 64]; w[0] = t.x; w[1] = t.y;  t = theirs[8 * 64]; w[2] = t.x; w[3] = t.y;
        t = theirs[9 * 64]; w[4] = t.x; w[5] = t.y;  t = theirs[10 * 64]; w[6] = t.x; w[8] = t.y;
        t = theirs[11 * 64]; w[10] = t.x; w[11] = t.y; t = theirs[12 * 64]; w[12] = t.x; w[13] = t.y;
        t = theirs[13 * 64]; w[14] = t.x; w[15] = t.y;
        w[7] = 0.0; w[9] = 0.0;
        k2_accumulate<W, 1>(w, acc);
      }
      K2_T(3);
      lds_barrier();   // (both waves have read: the rows may be overwritten by the next pass)
      K2_T(4);
    };
    // one register set, observations one pass ahead (two sets -- the pair of passes unrolled -- spill: 607 us against 209 at 8 x 2000 x 500)
#ifdef CC_RIG_K2_TWO_AHEAD   // (A/B: two register sets, observations two passes ahead, the pair of passes unrolled)
    ObsRaw ob;
    fetch(tid + 128, ob);
    int p = 0;
    for (; p + 1 < npass; p += 2) {
      const int k = p * 128 + tid;
      ObsD d;
      widen(oa, d);
      fetch(k + 256, oa);
      pass(k, d);
      widen(ob, d);
      fetch(k + 384, ob);
      pass(k + 128, d);
    }
    if (p < npass) {
      ObsD d;
      widen(oa, d);
      pass(p * 128 + tid, d);
    }
#else
    // (measured and dropped: one word of the observations two passes ahead, loaded and thrown away so that the real prefetch finds
    // its lines in L2 -- 210.7 us against 207.9 at 8 x 2000 x 500: the ~0.7 k cycles a pass waits for its observations are not
    // cache misses of the prefetch)
    for (int p = 0; p < npass; ++p) {
      const int k = p * 128 + tid;
      ObsD d;
      widen(oa, d);
      fetch(k + 128, oa);
      pass(k, d);
    }
#endif
    // ---- the NEXT group's loads go out now: they travel under this group's lane sums and assembly
    if (more) group_loads(gn, qn, oa, recv, old0, old1);
    // ---- sums over the lanes: 64 accumulators by the butterfly (value s ends in lane s), the others and the scalars by wave_sum
    int tid_t = threadIdx.x;
    asm volatile("" : "+v"(tid_t));
    const int lane_t = tid_t & 63;
    reduce_scatter64(acc, lane_t);
    {
      const int pr = k2_tab_byte(s_tab, 256, W * 64 + lane_t);
      if (pr != 255) s_G[pr] = acc[0];
    }
    {
      // the (at most four) accumulators beyond the butterfly, the cost and the model-cost term: eight values through three halving
      // steps and three plain ones (value e in lanes with bits 5, 4, 3 = e), instead of six full wave sums
      double v8[8] = {acc[64], acc[65], acc[66], acc[67], cost, qterm, 0.0, 0.0};
      reduce_swap32<4>(v8);
      reduce_swap16<2>(v8);
      reduce_dpp<1, 0x128, 8>(v8, lane_t);   // row_ror:8
      double t = v8[0];
      t += dpp_f64<0x141>(t);                // row_half_mirror (partner l ^ 7: stays inside the eight lanes that share bits 5, 4, 3)
      t += dpp_f64<0x4E>(t);                 // quad_perm:[2,3,0,1]
      t += dpp_f64<0xB1>(t);                 // quad_perm:[1,0,3,2]
      const int e8 = ((lane_t >> 5) & 1) * 4 + ((lane_t >> 4) & 1) * 2 + ((lane_t >> 3) & 1);
      if ((lane_t & 7) == 0) {
        if (e8 < 4) { const int pr = k2_tab_byte(s_tab, 322, W * 4 + e8); if (pr != 255) s_G[pr] = t; }
        else if (e8 == 4) s_G[136 + W] = t;
        else if (e8 == 5) s_G[138 + W] = t;
      }
    }
    K2_T(6);
    if (tid_t < 4) s_G[tid_t == 0 ? 43 : (tid_t == 1 ? 62 : (tid_t == 2 ? 53 : 64))] = 0.0;   // the structurally zero pairs (fy, fx) (py, fx) (px, fy) (py, px)
    lds_barrier();
    auto G = [&](int i, int j) { const int hi = i > j ? i : j, lo = i > j ? j : i; return s_G[hi * (hi + 1) / 2 + lo]; };
    // ---- the frame's couplings through the group's adjoint: T = G_cc M, H_fk = M^T H_ck, g_f = M^T g_c
    if (tid_t < 96) {
      double t = 0.0;
      if (tid_t < 36) {
        const int r = tid_t / 6, l = tid_t - 6 * r;
#pragma unroll
        for (int k = 0; k < 6; ++k) t = fma(G(r, k), s_m[k * 6 + l], t);
      } else if (tid_t < 90) {
        const int a = (tid_t - 36) / 9, j = (tid_t - 36) - 9 * a;
#pragma unroll
        for (int k = 0; k < 6; ++k) t = fma(s_m[k * 6 + a], G(k, 7 + j), t);
      } else {
        const int a = tid_t - 90;
#pragma unroll
        for (int k = 0; k < 6; ++k) t = fma(s_m[k * 6 + a], G(k, 6), t);
      }
      s_T[tid_t] = t;
    } else if (tid_t < 117) {   // share of the frame block, (M^T G_cc M)[i][j], i >= j -- in the same phase, straight from G (no T: no second barrier)
      const int e = tid_t - 96;
      int i = 0;
      while ((i + 1) * (i + 2) / 2 <= e) ++i;
      const int j = e - i * (i + 1) / 2;
      double t = 0.0;
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        double u = 0.0;
#pragma unroll
        for (int l = 0; l < 6; ++l) u = fma(G(k, l), s_m[l * 6 + j], u);
        t = fma(s_m[k * 6 + i], u, t);
      }
      s_T[tid_t] = t;
    }
    lds_barrier();
    double* rec = P.gcomp + ((size_t)dst * NG + g) * kRigRecK;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int e = tid_t + 128 * h;
      double v = 0.0;
      if (e <= kRkR2) v = s_G[k2_tab_byte(s_tab, 288, e)];
      else if (e < kRkHff) v = s_T[e - kRkT];
      else if (e < kRkGf) v = s_T[96 + (e - kRkHff)];
      else if (e < kRkEnd) v = s_T[90 + (e - kRkGf)];
      rec[e] = v;
    }
    if (phase == 0) {
      if (tid_t < 6) P.ghd0[g * 8 + tid_t] = fixed ? 0.0 : G(tid_t, tid_t);
      else if (tid_t >= 64 && tid_t < 73) P.ghdk[g * 16 + (tid_t - 64)] = G(7 + (tid_t - 64), 7 + (tid_t - 64));
    }
    if (tid_t == 0) {
      P.gstats[g * 2] = s_G[136] + s_G[137];
      P.gstats[g * 2 + 1] = s_G[138] + s_G[139];
    }
    K2_T(7);
#ifdef CC_RIG_K2_TIMING
    ++k2groups;
#endif
    g = gn;
    q = qn;
    // (the next group's first barrier -- behind its store of the records into sm -- separates this group's last reads of s_G, s_T
    // and s_m from the writes that follow)
  }
  if (dynamic && tid == 0) {
    const unsigned left = __hip_atomic_fetch_add(P.arrive + 15, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (left == (unsigned)stride - 1u) {
      __hip_atomic_store(P.arrive + 14, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(P.arrive + 15, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
#ifdef CC_RIG_K2_TIMING
  if (blockIdx.x == gridDim.x / 2 && tid == 0) {
    for (int qq = 0; qq < 8; ++qq) P.shared_stats[40 + qq] = (double)k2t[qq];
    P.shared_stats[48] = (double)(wall_clock64() - k2wall0);
    P.shared_stats[49] = (double)k2groups;
    for (int qq = 8; qq < 12; ++qq) P.shared_stats[42 + qq] = (double)k2t[qq];   // [50..53]: head of a group in pieces
  }
#endif
  };
  if (wave == 0) sweep(std::integral_constant<int, 0>{}); else sweep(std::integral_constant<int, 1>{});
}

// ---------------------------------------------------------------------------------------------
// update: per frame, back-substitute the pose step and form the candidate pose. 16 lanes/frame, 16 frames per
// 256-thread block (`fblk` = which sixteen). SC1: the shared step `ds` was written by another workgroup of the
// SAME launch (fused into k_rig_reduce): read it with sc1 loads.
// ---------------------------------------------------------------------------------------------
// What the update of a frame needs besides the shared step: fetched by the blocks of k_rig_reduce WHILE they wait for
// the solving block's flag (SW <= 32: two Y columns per lane), so that only the step itself is read behind the flag.
// Everything unconditional (all sixteen lanes of a frame fetch the frame's scalars: same addresses, one transaction).
struct RigUpdPre {   // (y: columns l, l + 16, l + 32, l + 48 of the frame's six rows of Y -- up to 64 shared columns)
  double y[4][6], p0[7], p1[7], sp[6];
  int g0, g1;
};
__device__ __forceinline__ void rig_update_prefetch(const RigDev& P, int64_t f, RigUpdPre& x) {   // f: frame of this thread's sixteen lanes
  const int tid = threadIdx.x, l = tid & 15;
  const int64_t fc = f < P.F ? f : 0;
  const double* Yf = P.Y + (size_t)fc * 6 * P.SW;
#pragma unroll
  for (int h = 0; h < 4; ++h) {
    const int k = l + 16 * h, kc = k < P.SW ? k : 0;
    if (16 * h < P.SW) {   // (uniform)
#pragma unroll
      for (int i = 0; i < 6; ++i) x.y[h][i] = Yf[i * P.SW + kc];
    } else {
#pragma unroll
      for (int i = 0; i < 6; ++i) x.y[h][i] = 0.0;
    }
  }
#pragma unroll
  for (int i = 0; i < 7; ++i) { x.p0[i] = P.pose[(size_t)fc * 8 + i]; x.p1[i] = P.pose[((size_t)P.F + fc) * 8 + i]; }
#pragma unroll
  for (int i = 0; i < 6; ++i) x.sp[i] = P.sp[fc * 8 + i];
  x.g0 = P.fgoff[fc]; x.g1 = P.fgoff[fc + 1];
}

template <bool SC1, bool PRE = false>
// f: frame of this thread's sixteen lanes; ds_lds: the shared step in LDS (persistent kernel), else read from P.ds
__device__ __forceinline__ void rig_update_body(const RigDev& P, int phase, int cur, int64_t f, const RigUpdPre& pre, const double* ds_lds = nullptr) {
  const int dst = phase == 0 ? cur : (cur ^ 1);
  int tid_ = threadIdx.x;
  if (ds_lds) asm volatile("" : "+v"(tid_));
  const int tid = tid_, l = tid & 15;
  const bool valid = f < P.F;
  const int64_t fc = valid ? f : 0;
  double u[6] = {0, 0, 0, 0, 0, 0};
  if (phase != 0 && PRE) {
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const int k = l + 16 * h;
      if (16 * h >= P.SW) continue;   // (uniform)
      double d = 1.0;   // (column S is the right-hand side)
      const double dk = __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(P.ds) + (k < P.S ? k : 0),
                                                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      if (k < P.S) d = dk;
      if (k < P.SW) {
#pragma unroll
        for (int i = 0; i < 6; ++i) u[i] += pre.y[h][i] * d;
      }
    }
  } else if (phase != 0) {
    const double* Yf = P.Y + (size_t)fc * 6 * P.SW;
    for (int k = l; k < P.SW; k += 16) {
      double d = 1.0;
      if (k < P.S)
        d = ds_lds ? ds_lds[k]
          : SC1 ? __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(P.ds) + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                : P.ds[k];
#pragma unroll
      for (int i = 0; i < 6; ++i) u[i] += Yf[i * P.SW + k] * d;
    }
  }
  if (phase != 0) {   // the sixteen lanes of a frame add up their columns
#pragma unroll
    for (int i = 0; i < 6; ++i) u[i] = row16_sum(u[i]);
  }
  if (!valid || l != 0) return;
  bool active;
  double q[4], t[3], spf[6];
  if (PRE) {
    active = pre.g1 > pre.g0;
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = cur ? pre.p1[i] : pre.p0[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) t[i] = cur ? pre.p1[4 + i] : pre.p0[4 + i];
#pragma unroll
    for (int i = 0; i < 6; ++i) spf[i] = pre.sp[i];
  } else {
    active = P.fgoff[f + 1] > P.fgoff[f];
    const double* pc = P.pose + ((size_t)cur * P.F + f) * 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = pc[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) t[i] = pc[4 + i];
#pragma unroll
    for (int i = 0; i < 6; ++i) spf[i] = phase != 0 ? P.sp[f * 8 + i] : 0.0;
  }
  double dp[6] = {0, 0, 0, 0, 0, 0};
  double step2 = 0.0;
  if (phase != 0) {
    if (active) {
#pragma unroll
      for (int i = 0; i < 6; ++i) dp[i] = -u[i] * spf[i];
      double qn[4];
      quat_plus(q, dp, qn);
#pragma unroll
      for (int i = 0; i < 4; ++i) { const double d = qn[i] - q[i]; step2 += d * d; q[i] = qn[i]; }
#pragma unroll
      for (int i = 0; i < 3; ++i) { const double tn = t[i] + dp[3 + i]; const double d = tn - t[i]; step2 += d * d; t[i] = tn; }
    }
    double* pd = P.pose + ((size_t)dst * P.F + f) * 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) pd[i] = q[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) pd[4 + i] = t[i];
  }
  double R[9];
  quat_to_R(q, R);
  double* rec = P.frec + (size_t)f * 32;
#pragma unroll
  for (int i = 0; i < 9; ++i) rec[i] = R[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) rec[9 + i] = t[i];
#pragma unroll
  for (int i = 0; i < 6; ++i) rec[12 + i] = dp[i];
  P.fstats[f * 2] = step2;
  P.fstats[f * 2 + 1] = active ? q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3] + t[0] * t[0] + t[1] * t[1] + t[2] * t[2] : 0.0;
}

__global__ __launch_bounds__(256) void k_rig_update(RigDev P) {
  rig_progress(P, RIG_PROG_UPDATE);
  const LmCtl* ctl = P.ctl;
  if (ctl->done) return;
  const int phase = ctl->phase;
  if (phase != 0 && !ctl->step_valid) return;
  RigUpdPre none;   // (unused: PRE = false)
  rig_update_body<false, false>(P, phase, ctl->cur, (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4), none);
}

// deterministic block-wide sum of one value per thread (256 threads); result valid for thread 0
__device__ __forceinline__ double block_sum256(double v, double* s4) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s4[threadIdx.x >> 6] = v;
  __syncthreads();
  return (s4[0] + s4[1]) + (s4[2] + s4[3]);
}

// column sums of gstats[NG][2] and fstats[F][2] -> out[0..3] = cost, q, step2, xnorm2 (all threads after return)
// pre_g / pre_f (optional): the rows i = u * 256 + tid, u < 8, of gstats / fstats requested by the caller at kernel start
// (at most 2048 rows each: the frame form's one row per frame) -- same sums in the same order, one round trip earlier
__device__ __forceinline__ void rig_reduce_stats(const RigDev& P, bool want, double* s16, double* out, const d2* pre_g = nullptr,
                                                 const d2* pre_f = nullptr) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double a[4] = {0, 0, 0, 0};
  if (want && pre_g) {
#pragma unroll
    for (int u = 0; u < 8; ++u) { a[0] += pre_g[u].x; a[1] += pre_g[u].y; }
#pragma unroll
    for (int u = 0; u < 8; ++u) { a[2] += pre_f[u].x; a[3] += pre_f[u].y; }
  } else if (want) {
    const d2* gs2 = reinterpret_cast<const d2*>(P.gstats);
    const d2* fs2 = reinterpret_cast<const d2*>(P.fstats);
    // up to sixteen loads in flight per thread: one round trip per 4096 groups instead of one per 256 (the plain loop waited
    // for every load: 20 us for the 16000 groups of BASELINE configs[4], in every block of the elimination)
    const int64_t nrows = P.fmode ? P.F : P.NG;   // (frame form: one row of cost / model-cost term per FRAME)
    for (int64_t i0 = 0; i0 < nrows; i0 += 16 * 256) {
      d2 v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) { const int64_t i = i0 + u * 256 + tid; v[u] = i < nrows ? gs2[i] : d2{0.0, 0.0}; }
#pragma unroll
      for (int u = 0; u < 16; ++u) { a[0] += v[u].x; a[1] += v[u].y; }
    }
    for (int64_t i0 = 0; i0 < P.F; i0 += 8 * 256) {
      d2 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int64_t i = i0 + u * 256 + tid; v[u] = i < P.F ? fs2[i] : d2{0.0, 0.0}; }
#pragma unroll
      for (int u = 0; u < 8; ++u) { a[2] += v[u].x; a[3] += v[u].y; }
    }
    // (measured: clamped unconditional loads + selects are 1-2 us slower here than these selects on the loaded value)
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) a[k] = wave_sum(a[k]);
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) s16[wave * 4 + k] = a[k];
  }
  __syncthreads();
  if (tid < 4) out[tid] = (s16[tid] + s16[4 + tid]) + (s16[8 + tid] + s16[12 + tid]);
  __syncthreads();
}

// Sum over this rank's groups of the diagonal entry of shared column k at the initial point (Jacobi scaling of
// the shared block). All 256 threads call; the result is valid for thread 0.
// Diagonal of the shared block at the initial point, all S columns -> out[0..S) (LDS), for the Jacobi scale. The columns
// of one camera (6 pose coordinates, kind 0) or of one intrinsics set (9, kinds 1 and 2) are consecutive and sum over the
// same groups, so a run is reduced together: the group indices of four steps are fetched first, then their values (two
// round trips per 1024 groups and run, fixed summation order). The first version walked one column at a time with a
// dependent index -> value load pair per step: 138 us for BASELINE configs[4], once per solve.
__device__ __forceinline__ void rig_diag_sums(const RigDev& P, double* s4, double* out, int only_run = -1, int slice = 0, int nslices = 1) {
  const int tid = threadIdx.x;
  int run = 0;
  for (int k = 0; k < P.S; ++run) {
    const int info = P.colinfo[k], kind = (info >> 4) & 15, co = info >> 8;
    const int n = kind == 0 ? 6 : kRigK;
    if (only_run >= 0 && run != only_run) { k += n; continue; }   // (k_rig_init: one run per block)
    const double* src = kind == 0 ? P.ghd0 : P.ghdk;
    const int stride = kind == 0 ? 8 : 16;
    int64_t lo = 0, hi = P.NG;
    if (kind != 2) { const int c = P.obs_cam[co]; lo = P.cam_goff[c]; hi = P.cam_goff[c + 1]; }
    else if (nslices > 1) {   // (k_rig_init: a set shared by all cameras is summed by several blocks, each over a slice of the groups)
      const int64_t len = (P.NG + nslices - 1) / nslices;
      lo = (int64_t)slice * len < P.NG ? (int64_t)slice * len : P.NG;
      hi = lo + len < P.NG ? lo + len : P.NG;
    }
    double h[kRigK];
#pragma unroll
    for (int c = 0; c < kRigK; ++c) h[c] = 0.0;
    for (int64_t i0 = lo; i0 < hi; i0 += 4 * 256) {
      int64_t g[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t i = i0 + u * 256 + tid;
        const int64_t ic = i < hi ? i : lo;   // (unconditional loads; idle slots are selected away below)
        g[u] = kind == 2 ? ic : (int64_t)P.cam_glist[ic];
      }
      double v[4][kRigK];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int c = 0; c < kRigK; ++c) v[u][c] = c < n ? src[(size_t)g[u] * stride + c] : 0.0;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const bool live = i0 + u * 256 + tid < hi;
#pragma unroll
        for (int c = 0; c < kRigK; ++c) h[c] += live ? v[u][c] : 0.0;
      }
    }
#pragma unroll
    for (int c = 0; c < kRigK; ++c) {
      if (c < n) {   // (uniform)
        const double sum = block_sum256(h[c], s4);
        if (tid == 0) out[k + c] = sum;
      }
    }
    k += n;
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// stats (multi-GPU only, one block): local sums of the sweep statistics -> vec_stats, which is then
// exchanged. [0..3] cost, model term, step^2, |x|^2; in phase 0 also [4 + k] = diagonal sum of shared
// column k over this rank's groups (Jacobi scaling of the shared block).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rig_stats(RigDev P) {
  rig_progress(P, RIG_PROG_STATS);
  __shared__ double s4[4];
  __shared__ double s16[16];
  __shared__ double s_out[4];
  const LmCtl* ctl = P.ctl;
  if (ctl->done) return;
  const int tid = threadIdx.x, phase = ctl->phase;
  const bool need = phase == 0 || (ctl->cand_pending && ctl->step_valid);
  rig_reduce_stats(P, need, s16, s_out);
  if (tid < 4) P.vec_stats[tid] = need ? s_out[tid] : 0.0;
  {
    __shared__ double s_diag[256];
    if (phase == 0) rig_diag_sums(P, s4, s_diag);
    for (int k = tid; k < P.S; k += 256) P.vec_stats[4 + k] = phase == 0 ? s_diag[k] : 0.0;
  }
  if (P.x.on) {
    // mailbox exchange (kind 1): post the local statistics, wait for every rank's, write the sums back
    // (k_rig_init / k_rig_elim read vec_stats as they do after an all-reduce)
    __shared__ double s_post[4 + 256];
    __shared__ int s_ok;
    __syncthreads();
    const int n = 4 + P.S;   // (up to 259: large rigs)
    for (int k = tid; k < n; k += 256) s_post[k] = P.vec_stats[k];
    __syncthreads();
    const unsigned long long epoch = P.x.seq[1] + 1ull;
    p2p_post(P.x, 1, epoch, P.rank, P.nranks, s_post, n);
    p2p_collect_to(P.x, 1, epoch, P.rank, P.nranks, n, P.vec_stats, &s_ok);
    if (tid == 0) {
      P.x.seq[1] = epoch;
      if (s_ok == 0) {
        LmCtl c = *ctl;
        c.done = 1; c.term = CC_FAILURE_EXCHANGE;
        *P.ctl = c; *P.ctl_next = c;
      }
    }
  }
}

// one-off (attach time): sum of the per-rank "camera seen" flags through the mailboxes (kind 1)
__global__ __launch_bounds__(128) void k_rig_flag_exchange(RigDev P, const double* in, double* out, int n, int* ok) {
  __shared__ double s_post[128];
  __shared__ int s_ok;
  const int tid = threadIdx.x;
  if (tid < n) s_post[tid] = in[tid];
  __syncthreads();
  const unsigned long long epoch = P.x.seq[1] + 1ull;
  p2p_post(P.x, 1, epoch, P.rank, P.nranks, s_post, n);
  const double a = p2p_collect(P.x, 1, epoch, P.rank, P.nranks, n, &s_ok);
  if (tid < n) out[tid] = a;
  if (tid == 0) { P.x.seq[1] = epoch; *ok = s_ok; }
}

// ---------------------------------------------------------------------------------------------
// init (one block, first evaluation only): Jacobi scale of the shared block, trust-region state
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rig_init(RigDev P) {
  rig_progress(P, RIG_PROG_INIT);
  __shared__ double s4[4];
  __shared__ double s16[16];
  __shared__ double s_out[4];
  __shared__ double s_ss[256];
  const LmCtl* ctl = P.ctl;
  // Only block 0 looks at the control block: it ends by storing lm_init's result (phase = 1) into it, so a block of this
  // launch that is dispatched late (busy or partitioned GPU) would see the flipped phase, return, and leave its run's
  // Jacobi scales unset and -- a sliced run -- the arrival counter short. The kernel is launched in the first round of a
  // solve only (rig_enqueue_round, `initial`), where the phase IS 0 unless an exchange has already failed; the scales a
  // failed solve computes for nothing are harmless.
  if (blockIdx.x == 0 && (ctl->done || ctl->phase != 0)) return;
  const int tid = threadIdx.x;
  const bool jac = P.opts->jacobi_scaling != 0;
  // Single GPU: 1 + (runs of columns) blocks. Block r > 0 sums the diagonal of run r - 1 (one camera's poses or one
  // intrinsics set) and writes its Jacobi scales; block 0 does the rest. Nothing is exchanged between the blocks. (One
  // block doing all runs took 130 us at BASELINE configs[4]: tens of thousands of scattered 8-byte loads through one CU.)
  if (blockIdx.x > 0) {
    if (P.comm) return;
    // which run, and which slice of it (a set of intrinsics shared by all cameras sums over EVERY group: 16000 at
    // BASELINE configs[4], 61 us in one block; init_slices blocks take a slice each and the last one to arrive adds the
    // partial sums up in slice order)
    int b = (int)blockIdx.x - 1, run = 0, slice = 0, ns = 1, k0 = 0;
    for (int k = 0; k < P.S; ++run) {
      const int kind = (P.colinfo[k] >> 4) & 15;
      const int cnt = kind == 2 ? P.init_slices : 1;
      if (b < cnt) { slice = b; ns = cnt; k0 = k; break; }
      b -= cnt;
      k += kind == 0 ? 6 : kRigK;
    }
    for (int k = tid; k < 256; k += 256) s_ss[k] = -1.0;
    __syncthreads();
    rig_diag_sums(P, s4, s_ss, run, slice, ns);
    if (ns == 1) {
      for (int k = tid; k < P.S; k += 256)
        if (s_ss[k] >= 0.0) P.ss[k] = jac ? 1.0 / (1.0 + sqrt(s_ss[k])) : 1.0;
      return;
    }
    __shared__ int s_last_slice;
    unsigned long long* part = reinterpret_cast<unsigned long long*>(P.partial);   // (free until the first elimination)
    if (tid < kRigK) __hip_atomic_store(part + slice * 16 + tid, (unsigned long long)__double_as_longlong(s_ss[k0 + tid]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) s_last_slice = __hip_atomic_fetch_add(P.arrive + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == (unsigned)ns;
    __syncthreads();
    if (!s_last_slice) return;
    if (tid == 0) __hip_atomic_store(P.arrive + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // next solve
    if (tid < kRigK) {
      double t = 0.0;
      for (int q = 0; q < ns; ++q) t += __longlong_as_double((long long)__hip_atomic_load(part + q * 16 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      P.ss[k0 + tid] = jac ? 1.0 / (1.0 + sqrt(t)) : 1.0;
    }
    return;
  }
  if (P.comm) {
    if (tid < 4) s_out[tid] = P.vec_stats[tid];
    if (tid < P.S) P.ss[tid] = jac ? 1.0 / (1.0 + sqrt(P.vec_stats[4 + tid])) : 1.0;
  } else {
    rig_reduce_stats(P, true, s16, s_out);
  }
  __syncthreads();
  // |x|^2 of the shared block: one value per thread and step (thread 0 walking the cameras alone waited for 65
  // dependent loads at BASELINE configs[4])
  double x2 = 0.0;
  {
    const int cur0 = ctl->cur;
    for (int i = tid; i < P.C * 7; i += 256) {
      const int cc2 = i / 7;
      const double v = P.cam[((size_t)cur0 * P.C + cc2) * 8 + (i - cc2 * 7)];
      x2 += P.cam_fixed[cc2] ? 0.0 : v * v;
    }
    for (int i = tid; i < P.CK * kRigK; i += 256) {   // every intrinsic of a set that is in the problem counts in |x|
      const int ks = i / kRigK;
      const double v = P.intr[((size_t)cur0 * P.CK + ks) * 16 + (i - ks * kRigK)];
      x2 += P.kscol[ks] >= 0 ? v * v : 0.0;
    }
  }
  const double x2_shared = block_sum256(x2, s4);
  if (tid == 0) {
    LmCtl c = *ctl;
    const LmOpts o = *P.opts;
    const double xn2 = s_out[3] + x2_shared;
    lm_init(c, o, s_out[0], sqrt(xn2));
    *P.ctl = c;
    *P.ctl_next = c;
  }
}

// ---------------------------------------------------------------------------------------------
// elim: trust-region decision (every block, same answer; block 0 publishes it), then the elimination of the
// frame poses. One WAVE per frame, four frames per block iteration:
//   lanes 0..26 sum the frame block A = sum_groups H_ff (21 entries) and g_f (6) over the frame's groups;
//   every lane factors the damped 6x6 block in registers; lane k owns shared column k (and k + 64):
//   w = column of [H_fs | g_f] (scaled), z = L^-1 w -> staged in LDS, y = L^-T z -> Y (back-substitution);
//   the wave also adds its frame's entries of the shared diagonal blocks into per-lane accumulators;
//   then the four waves contract the 24 staged rows of Z on the matrix cores: tile pair (ti <= tj) of
//   the (SW x SW) product Z^T Z goes to wave (index mod 4), 6 k-steps of v_mfma_f64_16x16x4_f64.
// Partial row of a block: [nT tiles x 256 | ND direct sums | Cholesky failures | max |g_frame|].
// ---------------------------------------------------------------------------------------------
// (timing-only builds: block 0 leaves wall-clock marks in shared_stats[20..], scripts/time_rig_reduce.py)
#ifdef CC_RIG_TIMING
#define ELIM_MARK(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) P.shared_stats[20 + (i)] = (double)wall_clock64(); } while (0)
#else
#define ELIM_MARK(i) do { } while (0)
#endif

// NR = direct-sum accumulators per lane: 8 covers ND <= 512 (the usual rigs: <= 18 observed cameras with poses only, 3 with
// intrinsics), 24 the full range; the small variant exists because the kernel sits at the register limit.
// The elimination as a function: k_rig_elim (a launch of its own: trust-region decision, then the elimination) and the
// persistent per-solve kernel (PS: the decision is the control workgroup's -- which buffer holds the point to eliminate,
// the radius, whether this is the first elimination, and the Jacobi scales of the shared columns come as arguments).
// FM: the sweep was k_rig_sweep_frame -- a group's record is [G7 (28) | T (36)] (P.gcomp), the frame block comes summed (P.fsum).
// KC: the sweep was k_rig_sweep_k2 (intrinsics, compact records of kRigRecK doubles in P.gcomp: offsets kRk*).
template <bool HK, int NR, bool PS, bool FM = false, bool KC = false>
__device__ __forceinline__ void rig_elim_body(const RigDev& P, char* smem_raw, const int ps_cur, const double ps_radius, const bool ps_first,
                                              const double* ps_ss) {
  static_assert(!(HK && FM), "the frame form is the poses-only sweep's");
  static_assert(!KC || HK, "compact K records belong to the sweep with intrinsics");
  double* s_Z = reinterpret_cast<double*>(smem_raw);         // [24][ZS] staged Z rows of the four frames
  double* s_A = s_Z + 24 * P.ZS;                             // [4][32] frame block broadcast, per wave
  double* s_red = s_A + 4 * 32;                              // [4][1024] cross-wave reduction scratch
  // large variant (NR > 8): the direct-sum accumulators of a wave live in LDS, one slot per lane and register index --
  // as registers they pushed the kernel over the 512-VGPR limit (48 spilled VGPRs, scratch traffic in the frame loop)
  constexpr bool kLdsAcc = NR > 8;
  double* s_dacc = s_red + 4 * 1024;                         // [4][NR * 64] (large variant only)
  __shared__ double s_ss[kRigMaxS + 1];
  __shared__ double s16[16];
  __shared__ double s_tot[4];
  __shared__ double s_fg[8];
  __shared__ LmCtl s_ctl;
  int tid_ = threadIdx.x;
  if (PS) asm volatile("" : "+v"(tid_));   // (a fresh copy per call: the lane tables below are rebuilt every round of the persistent kernel instead of
                                              //  being hoisted out of its round loop and kept -- spilled -- across the sweep)
  const int tid = tid_, lane = tid & 63, wave = tid >> 6;
  const LmCtl* ctl = P.ctl;
#ifdef CC_RIG_TIMING
  const long long tm0 = wall_clock64();
#endif
  const int ctl_done = PS ? 0 : ctl->done, ctl_phase = PS ? 1 : ctl->phase;
  // what thread 0 needs for the trust-region decision, fetched now instead of behind the statistics barrier
  LmCtl c_in;
  LmOpts o_in;
  double sh0 = 0.0, sh1 = 0.0;
  if constexpr (!PS) { c_in = *ctl; o_in = *P.opts; sh0 = P.shared_stats[0]; sh1 = P.shared_stats[1]; }
  // ... and, frame form, the statistics rows themselves (one per frame: eight per thread up to 2048 frames): requested next
  // to the control block instead of behind it
  constexpr bool kPre = FM && !PS;
  d2 pre_g[kPre ? 8 : 1], pre_f[kPre ? 8 : 1];
  const bool pre = kPre && P.fmode && P.F <= 2048 && !P.comm;
  if constexpr (kPre) {
    if (pre) {
      const d2* gs2 = reinterpret_cast<const d2*>(P.gstats);
      const d2* fs2 = reinterpret_cast<const d2*>(P.fstats);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int64_t i = u * 256 + tid;
        pre_g[u] = i < P.F ? gs2[i] : d2{0.0, 0.0};
        pre_f[u] = i < P.F ? fs2[i] : d2{0.0, 0.0};
      }
    }
  }
  // ---- loads that do not depend on the trust-region decision go out first, under the statistics round trip: the
  // lane's static tables and the group slots of the block's first four frames
  const int CO = P.CO;
  const int64_t f_first = (int64_t)blockIdx.x * 4 + wave;
  const int gj_first = (f_first < P.F && lane < CO) ? P.fslot[f_first * CO + lane] : -1;
  // static (frame-independent) description of what this lane owns
  // frame-block entry of lane e < 27 (offset inside a group's AA tile); lanes holding a diagonal entry also
  // store the frame's Jacobi scale in the first elimination
  int a_off = 0, sp_i = -1;
  if (lane < 21) {
    int i = 0;
    while (tri(i + 1, 0) <= lane) ++i;
    const int j = lane - tri(i, 0);
    a_off = KC ? kRkHff + lane : (6 + i) * 16 + 6 + j;
    if (i == j) sp_i = i;
  } else if (lane < 27) {
    a_off = KC ? kRkHff + lane : (6 + (lane - 21)) * 16 + 12;
  }
  // shared columns of this lane: k = lane and lane + 64
  int c_kind[2], c_co[2], c_comp[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int k = lane + 64 * h;
    c_kind[h] = -1; c_co[h] = 0; c_comp[h] = 0;
    if (k < P.SW) {
      const int info = P.colinfo[k];
      c_kind[h] = (info >> 4) & 15; c_co[h] = info >> 8; c_comp[h] = info & 15;
    }
  }
  // direct-sum entries of this lane: e = lane + 64 r -> (observed camera, offset inside the group block)
  int d_ent[NR];   // (observed camera << 16) | offset, -1: nothing (packed: registers are scarce here)
  double dacc[kLdsAcc ? 1 : NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int e = lane + 64 * r;
    if (kLdsAcc) s_dacc[(wave * NR + r) * 64 + lane] = 0.0; else dacc[r] = 0.0;
    const int t = P.dent[e < P.ND ? e : 0];   // (unconditional load + select: a conditional load waits on its own)
    d_ent[r] = e < P.ND ? t : -1;
  }
  // tile pairs of this wave's accumulators
  int t_ij[kRigTilesPerWave];   // ti | tj << 8, wave-uniform (scalar registers)
#pragma unroll
  for (int u = 0; u < kRigTilesPerWave; ++u) {
    const int idx = 4 * u + wave, ic = idx < P.nT ? idx : 0;
    t_ij[u] = __builtin_amdgcn_readfirstlane((int)P.tile_i[ic] | ((int)P.tile_j[ic] << 8));
  }
  if (ctl_done || ctl_phase == 0) return;
#ifdef CC_RIG_TIMING
  const long long tm1 = wall_clock64();
#endif
  bool pending = false;
  if constexpr (!PS) {
  pending = ctl->cand_pending != 0;
  if (P.comm) {
    if (tid < 4) s_tot[tid] = P.vec_stats[tid];
    __syncthreads();
  } else {
    if (kPre && pre) rig_reduce_stats(P, pending && ctl->step_valid, s16, s_tot, pre_g, pre_f);
    else rig_reduce_stats(P, pending && ctl->step_valid, s16, s_tot);
  }
  if (tid == 0) {
    LmCtl c = c_in;
    const LmOpts& o = o_in;
    if (pending) {
      double step2 = s_tot[2], xn2 = s_tot[3];
      if (c.step_valid) { step2 += sh0; xn2 += sh1; }
      cc_iteration rec;
      const int len0 = c.log_len;
      lm_decide(c, o, &rec, s_tot[0], s_tot[1], step2, xn2);
      if (blockIdx.x == 0 && c.log_len != len0 && c.log_len <= P.log_cap) P.log[c.log_len - 1] = rec;
    }
    s_ctl = c;
    if (blockIdx.x == 0) *P.ctl_next = c;
  }
  }   // (!PS)
  if (tid < P.S) s_ss[tid] = (PS ? ps_ss : P.ss)[tid];
  for (int i = tid; i < 24 * P.ZS; i += 256) s_Z[i] = 0.0;   // padding columns stay zero
  __syncthreads();
#ifdef CC_RIG_TIMING
  const long long tm2 = wall_clock64();
#endif
  if (!PS && s_ctl.done) return;
  const int cur = PS ? ps_cur : s_ctl.cur;
  const double inv_radius = 1.0 / (PS ? ps_radius : s_ctl.radius);
  const double mn = P.opts->min_lm_diagonal, mx = P.opts->max_lm_diagonal;
  const bool first_elim = PS ? ps_first : (ctl->phase == 1 && s_ctl.iter == 0 && !pending);   // Jacobi scale of the frame blocks
  const bool jac = P.opts->jacobi_scaling != 0;
  const int SW = P.SW, S = P.S, ZS = P.ZS;
  const size_t gs = FM ? (size_t)64 : (KC ? (size_t)kRigRecK : (size_t)P.gstride);
  const double* blocks = (FM || KC) ? P.gcomp + (size_t)cur * P.NG * gs : P.gblocks + (size_t)cur * P.NG * gs;

  double c_ss[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) { const int k = lane + 64 * h; c_ss[h] = k < S ? s_ss[k] : (k < SW ? 1.0 : 0.0); }
  d4 acc[kRigTilesPerWave];
#pragma unroll
  for (int u = 0; u < kRigTilesPerWave; ++u) acc[u] = d4{0.0, 0.0, 0.0, 0.0};
  double gmax = 0.0, nfail = 0.0;
  double* As = s_A + wave * 32;
#ifdef CC_RIG_TIMING
  if (blockIdx.x == 0 && tid == 0) { P.shared_stats[20] = (double)tm0; P.shared_stats[21] = (double)tm1; P.shared_stats[22] = (double)tm2; }
#endif
  ELIM_MARK(3);

  int gj_next = gj_first;
  const int nr = (P.ND + 63) >> 6;   // direct-sum registers in use (uniform)
  for (int64_t fb = (int64_t)blockIdx.x * 4; fb < P.F; fb += (int64_t)gridDim.x * 4) {
    const int64_t f = fb + wave;
    // group of (frame, observed camera j) on lane j: fetched one pass ahead (-1 beyond the last frame)
    const int gj = gj_next;
    {
      const int64_t fn = f + (int64_t)gridDim.x * 4;
      gj_next = (fn < P.F && lane < CO) ? P.fslot[fn * CO + lane] : -1;
    }
    const bool live = __any(gj >= 0);   // the frame has observations (wave-uniform)
    if (live) {
      // the frame's Jacobi scale (overwritten below in the first elimination, which computes it)
      double sf[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) sf[i] = P.sp[f * 8 + i];
      // (the frame's quaternion at the point being eliminated: for the gradient norm, same round trip)
      const double* fqp = P.pose + ((size_t)cur * P.F + f) * 8;
      const double fq0 = fqp[0], fq1 = fqp[1], fq2 = fqp[2], fq3 = fqp[3];
      // ---- loads: frame block entries (lanes < 27, summed over the groups), column data, direct entries
      double a_e = 0.0;
      if (FM) {
        a_e = P.fsum[((size_t)cur * P.F + f) * 32 + (lane < 27 ? lane : 0)];   // (the sweep summed the frame's groups)
        if (lane >= 27) a_e = 0.0;
      } else {
        for (int j0 = 0; j0 < CO; j0 += 8) {
          double v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int g = j0 + u < CO ? __builtin_amdgcn_readlane(gj, j0 + u) : -1;
            v[u] = (g >= 0 && lane < 27) ? blocks[(size_t)g * gs + a_off] : 0.0;
          }
          a_e += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        }
      }
      double w[2][6];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int i = 0; i < 6; ++i) w[h][i] = 0.0;
        const int gsel = __shfl(gj, c_co[h], 64);
        if ((c_kind[h] == 0 || c_kind[h] == 1) && gsel >= 0) {
          const double* G = blocks + (size_t)gsel * gs;
#pragma unroll
          for (int i = 0; i < 6; ++i)
            w[h][i] = FM ? G[28 + c_comp[h] * 6 + i]
                    : KC ? (c_kind[h] == 0 ? G[kRkT + c_comp[h] * 6 + i] : G[kRkFK + i * 9 + c_comp[h]])
                         : (c_kind[h] == 0 ? G[c_comp[h] * 16 + 6 + i] : G[256 + (6 + i) * 16 + c_comp[h]]);
        }
      }
      if (HK && P.kmode == RIG_K_SHARED) {
        // columns of the intrinsics shared by all cameras: sum of the frame's groups. Four groups per round trip, unconditional
        // loads (group 0 stands in for a camera that does not see the frame) and selects: written as a loop over the groups with
        // a `continue`, every group's six loads waited for on their own -- CO dependent round trips per frame (round 5: the
        // elimination at 8 x 2000 x 500 with shared intrinsics 43 -> ... us)
        for (int j0 = 0; j0 < CO; j0 += 4) {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            if (c_kind[h] != 2) continue;   // (nine lanes own such a column; the others skip the batch)
            double t[4][6];
            bool ok[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int g = j0 + u < CO ? __builtin_amdgcn_readlane(gj, j0 + u < CO ? j0 + u : 0) : -1;
              ok[u] = g >= 0;
              const double* G = blocks + (size_t)(ok[u] ? g : 0) * gs + (KC ? kRkFK : 256);
#pragma unroll
              for (int i = 0; i < 6; ++i) t[u][i] = KC ? G[i * 9 + c_comp[h]] : G[(6 + i) * 16 + c_comp[h]];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
              for (int i = 0; i < 6; ++i) w[h][i] += ok[u] ? t[u][i] : 0.0;
          }
        }
      }
      // direct entries, eight registers per round trip: unconditional loads (group 0 stands in where a lane has
      // nothing to fetch) followed by selects -- a load inside a lane-dependent `if` gets a wait of its own
      constexpr int DB = NR <= 8 ? 8 : 4;   // loads per round trip (the large variant has no registers to spare)
#pragma unroll
      for (int r0 = 0; r0 < NR; r0 += DB) {
        if (r0 < nr) {   // (uniform)
          double dx[DB];
          bool dk[DB];
#pragma unroll
          for (int u = 0; u < DB; ++u) {
            const int t = d_ent[r0 + u];
            const int g = __shfl(gj, t < 0 ? 0 : (t >> 16), 64);
            dk[u] = t >= 0 && g >= 0;
            dx[u] = blocks[(size_t)(dk[u] ? g : 0) * gs + (t & 0xffff)];
          }
#pragma unroll
          for (int u = 0; u < DB; ++u) {
            if (kLdsAcc) s_dacc[(wave * NR + r0 + u) * 64 + lane] += dk[u] ? dx[u] : 0.0;   // (own slot: no conflict)
            else dacc[r0 + u] += dk[u] ? dx[u] : 0.0;
          }
        }
      }
      // ---- broadcast the frame block
      if (lane < 27) As[lane] = a_e;
      wave_lds_fence();
      double A[27];
#pragma unroll
      for (int i = 0; i < 27; ++i) A[i] = As[i];
      wave_lds_fence();
      if (fb == (int64_t)blockIdx.x * 4) ELIM_MARK(4);
      if (first_elim) {
#pragma unroll
        for (int i = 0; i < 6; ++i) sf[i] = jac ? 1.0 / (1.0 + sqrt(A[tri(i, i)])) : 1.0;
        if (sp_i >= 0) P.sp[f * 8 + sp_i] = jac ? 1.0 / (1.0 + sqrt(a_e)) : 1.0;
      }
      double L[21], Li[6];
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) L[tri(i, j)] = sf[i] * A[tri(i, j)] * sf[j];
#pragma unroll
      for (int i = 0; i < 6; ++i) L[tri(i, i)] += clampd(L[tri(i, i)], mn, mx) * inv_radius;
      bool ok = true;
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        double d = L[tri(j, j)];
#pragma unroll
        for (int k = 0; k < j; ++k) d -= L[tri(j, k)] * L[tri(j, k)];
        ok = ok && (d > 0.0) && isfinite(d);
        const double inv = rsqrt_pos(d);
        L[tri(j, j)] = d * inv;
        Li[j] = inv;
#pragma unroll
        for (int i = j + 1; i < 6; ++i) {
          double a = L[tri(i, j)];
#pragma unroll
          for (int k = 0; k < j; ++k) a -= L[tri(i, k)] * L[tri(j, k)];
          L[tri(i, j)] = a * inv;
        }
      }
      if (!ok) nfail += 1.0;
      if (fb == (int64_t)blockIdx.x * 4) ELIM_MARK(5);
      {   // the frame's share of Ceres' gradient_max_norm, ||x - Plus(x, -g)||_inf (pose_grad_proj_max, cc_common.hpp)
        const double q4[4] = {fq0, fq1, fq2, fq3};
        gmax = fmax(gmax, pose_grad_proj_max(q4, &A[21]));
      }
      // ---- columns: z = L^-1 w, y = L^-T z
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int k = lane + 64 * h;
        if (k < SW) {
          double z[6], y[6];
#pragma unroll
          for (int i = 0; i < 6; ++i) {
            double a = c_kind[h] == 3 ? sf[i] * A[21 + i] : sf[i] * w[h][i] * c_ss[h];
#pragma unroll
            for (int kk = 0; kk < i; ++kk) a -= L[tri(i, kk)] * z[kk];
            z[i] = a * Li[i];
          }
#pragma unroll
          for (int i = 5; i >= 0; --i) {
            double a = z[i];
#pragma unroll
            for (int kk = i + 1; kk < 6; ++kk) a -= L[tri(kk, i)] * y[kk];
            y[i] = a * Li[i];
          }
#pragma unroll
          for (int i = 0; i < 6; ++i) {
            s_Z[(wave * 6 + i) * ZS + k] = z[i];
            P.Y[((size_t)f * 6 + i) * SW + k] = y[i];
          }
        }
      }
    } else {
      for (int k = lane; k < SW; k += 64)
#pragma unroll
        for (int i = 0; i < 6; ++i) s_Z[(wave * 6 + i) * ZS + k] = 0.0;
    }
    __syncthreads();
    if (fb == (int64_t)blockIdx.x * 4) ELIM_MARK(6);
    // ---- Schur products of the four staged frames on the matrix cores
#pragma unroll
    for (int u = 0; u < kRigTilesPerWave; ++u) {
      const int idx = 4 * u + wave;
      if (idx < P.nT) {
        const int ti = t_ij[u] & 255, tj = t_ij[u] >> 8;
        const int col = lane & 15, sub = lane >> 4;
#pragma unroll
        for (int ksx = 0; ksx < 6; ++ksx) {
          const double* row = s_Z + (4 * ksx + sub) * ZS;
          const double a = row[16 * ti + col], b = row[16 * tj + col];
          acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[u], 0, 0, 0);
        }
      }
    }
    __syncthreads();
    if (fb == (int64_t)blockIdx.x * 4) ELIM_MARK(7);
  }
  ELIM_MARK(8);

  // ---- one partial row per block
  double* prow = P.partial + (size_t)blockIdx.x * P.PC;
  // tiles: each belongs to exactly one wave. C/D layout: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
  for (int u = 0; u < kRigTilesPerWave; ++u) {
    const int idx = 4 * u + wave;
    if (idx < P.nT) {
#pragma unroll
      for (int r = 0; r < 4; ++r) prow[(size_t)idx * 256 + ((lane >> 4) + 4 * r) * 16 + (lane & 15)] = acc[u][r];
    }
  }
  // direct sums: four waves each hold partial sums of the same entries
  if (kLdsAcc) {
    __syncthreads();
    for (int e = tid; e < P.ND; e += 256)
      prow[P.pc_dir + e] = (s_dacc[e] + s_dacc[NR * 64 + e]) + (s_dacc[2 * NR * 64 + e] + s_dacc[3 * NR * 64 + e]);
  }
#pragma unroll
  for (int r0 = 0; r0 < (kLdsAcc ? 0 : NR); r0 += 16) {
    if (r0 * 64 >= P.ND) continue;   // (uniform) nothing left in this chunk
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (r0 + r < NR) s_red[wave * 1024 + r * 64 + lane] = dacc[r0 + r];
    __syncthreads();
    for (int i = tid; i < 1024; i += 256) {
      const int e = r0 * 64 + i;
      if (e < P.ND) prow[P.pc_dir + e] = (s_red[i] + s_red[1024 + i]) + (s_red[2048 + i] + s_red[3072 + i]);
    }
  }
  gmax = wave_max(gmax);
  if (lane == 0) { s_fg[wave] = gmax; s_fg[4 + wave] = nfail; }
  __syncthreads();
  if (tid == 0) {
    prow[P.pc_fail] = (s_fg[4] + s_fg[5]) + (s_fg[6] + s_fg[7]);
    prow[P.pc_gmax] = fmax(fmax(s_fg[0], s_fg[1]), fmax(s_fg[2], s_fg[3]));
  }
  ELIM_MARK(9);
}

template <bool HK, int NR, bool FM = false, bool KC = false>
__global__ __launch_bounds__(256) void k_rig_elim(RigDev P) {
  rig_progress(P, RIG_PROG_ELIM);
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  rig_elim_body<HK, NR, false, FM, KC>(P, smem_raw, 0, 1.0, false, nullptr);
}

// ---------------------------------------------------------------------------------------------
// The solve step, run by ONE block of 256 threads on the reduced sums (vec: [nT tiles | direct | fail | 0],
// then one max-gradient slot per rank): assembles the damped reduced system in LDS (lower triangle), dense
// Cholesky (all four waves, one barrier per column), then wave 0 alone: substitutions with lane i owning
// b[i] and b[i + 64] (cross-lane values through v_readlane, no barrier), gradient / radius tests
// (lm_finalize order), camera and intrinsics candidates, control block.
// SRC: where a reduced value comes from. 0: plain loads of P.vec (after an all-reduce, RCCL route);
// 1: write-through stores of the reduce blocks, read with sc1 loads (last-block-done, single GPU);
// 2: the ranks' mailbox slots, polled and added in rank order (last-block-done, mailbox exchange).
// Tried and dropped in round 1 (S = 24): a single-wave factorisation through LDS (33 us vs 20) and a
// register-tiled one with only the pivot column crossing threads through LDS (23 us).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double readlane_d(double x, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(x), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(x), l);
  return __hiloint2double(hi, lo);
}



// shared step: write-through, so that workgroups of the same launch can read it behind a flag (sc1 loads)
__device__ __forceinline__ void store_ds(double* p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct RigVal {   // reader of reduced value e
  const RigDev& P; unsigned long long epoch; long long t0; int* s_ok; const double* lds_vec;
  template <int SRC>
  __device__ __forceinline__ double get(int e) const {
    if (SRC == 3) return lds_vec[e];   // (persistent kernels: the control workgroup keeps the reduced row in LDS)
    if (SRC == 0) return P.vec[e];
    if (SRC == 1)
      return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(P.vec) + e,
                                                              __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    return p2p_poll_sum(P.x, 0, epoch, P.rank, P.nranks, e, t0, s_ok);
  }
};

// writes the camera / intrinsics records the sweep reads, from the current parameters plus (step_ok) the step
// x (LDS, scaled shared step with the sign of b: the step is -x * ss). Returns this thread's share of
// (step^2, |x_cand|^2) of the shared block. All 256 threads call.
__device__ __forceinline__ void rig_candidates(const RigDev& P, const double* x, const double* ss, bool have_step, int cur, int dst,
                                               double& step2, double& xn2) {
  const int tid = threadIdx.x;
  step2 = 0.0; xn2 = 0.0;
  for (int c = tid; c < P.C; c += 256) {
    const double* pc = P.cam + ((size_t)cur * P.C + c) * 8;
    double q[4] = {pc[0], pc[1], pc[2], pc[3]}, t[3] = {pc[4], pc[5], pc[6]};
    double dc[6] = {0, 0, 0, 0, 0, 0};
    const int p0 = P.pcol[c];
    if (have_step) {
      if (p0 >= 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) dc[k] = -x[p0 + k] * ss[p0 + k];
        double qn[4];
        quat_plus(q, dc, qn);
#pragma unroll
        for (int k = 0; k < 4; ++k) { const double d = qn[k] - q[k]; step2 += d * d; q[k] = qn[k]; }
#pragma unroll
        for (int k = 0; k < 3; ++k) { const double tn = t[k] + dc[3 + k]; const double d = tn - t[k]; step2 += d * d; t[k] = tn; }
        xn2 += q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3] + t[0] * t[0] + t[1] * t[1] + t[2] * t[2];
      }
      double* pd = P.cam + ((size_t)dst * P.C + c) * 8;
#pragma unroll
      for (int k = 0; k < 4; ++k) pd[k] = q[k];
#pragma unroll
      for (int k = 0; k < 3; ++k) pd[4 + k] = t[k];
    }
    double R[9];
    quat_to_R(q, R);
    double* rec = P.camrec + c * 32;
#pragma unroll
    for (int k = 0; k < 9; ++k) rec[k] = R[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) rec[9 + k] = t[k];
#pragma unroll
    for (int k = 0; k < 6; ++k) rec[12 + k] = dc[k];
  }
  // extension: candidate intrinsics, thread (set, j). Every intrinsic of a set that is in the problem counts in
  // |x| (cf. IntrinsicsProblem), frozen ones do not move.
  for (int i = tid; i < P.CK * kRigK; i += 256) {
    const int s = i / kRigK, j = i - s * kRigK;
    const int k0 = P.kscol[s];
    const double kc = P.intr[((size_t)cur * P.CK + s) * 16 + j];
    double dk = 0.0;
    if (have_step && k0 >= 0 && !((P.kmask[s] >> j) & 1u)) dk = -x[k0 + j] * ss[k0 + j];
    const double kn = kc + dk;
    if (have_step) {
      P.intr[((size_t)dst * P.CK + s) * 16 + j] = kn;
      if (k0 >= 0) { step2 += dk * dk; xn2 += kn * kn; }
    }
    P.krec[s * 32 + j] = kn;
    P.krec[s * 32 + 16 + j] = dk;
  }
}

// One panel (columns j0 .. j0 + nc - 1, nc <= 8) of the Cholesky factorisation of the S x S system in LDS, on ONE
// wave: lane i keeps the panel's entries of row i (and, TWO, of row i + 64) in registers. A column step takes the
// pivot with v_readlane, scales the column, puts it into the LDS vector `colbuf` and reads the multipliers of the
// panel's remaining columns back as uniform-address LDS reads (LDS operations of one wave execute in order: no
// barrier). The forward substitution of the right-hand side (b0 / b1: rows i / i + 64) rides along; v0 / v1 collect
// 1 / L_ii. The loop over panels is rolled (rig_solve_block), so the code stays a few hundred instructions:
// the fully unrolled whole-matrix-in-registers form this replaces ran 20 KB of straight-line code once per launch and
// was bound by instruction fetch (profiles/r02/rig_reduce_breakdown.txt).
template <bool TWO>
__device__ __forceinline__ void chol_panel(double* A, int S, int LD, int j0, int nc, double* colbuf, double& b0, double& b1,
                                           double& v0, double& v1, bool& okw) {
  const int lane = threadIdx.x & 63, i0 = lane, i1 = lane + 64;
  double p0[8], p1[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int col = j0 + c;
    // (unconditional loads from a clamped row, then a select: conditional loads become one branch and one wait each)
    const double x0 = A[(size_t)(i0 < S ? i0 : S - 1) * LD + col];
    const double x1 = TWO ? A[(size_t)(i1 < S ? i1 : S - 1) * LD + col] : 0.0;
    p0[c] = (c < nc && i0 < S && col <= i0) ? x0 : 0.0;
    p1[c] = (TWO && c < nc && i1 < S && col <= i1) ? x1 : 0.0;
  }
  // The pivot of the NEXT column is taken ahead of the column's own update (two lane reads and one FMA, the very
  // operation the update performs on that entry, so the value is the same bit for bit): its reciprocal square root is
  // then computed while the LDS round trip of the multipliers is in flight instead of behind it.
  double d = (!TWO || j0 < 64) ? readlane_d(p0[0], j0 & 63) : readlane_d(p1[0], j0 & 63);
  double inv = rsqrt_pos(d);
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    if (c < nc) {   // (uniform)
      const int col = j0 + c;
      okw = okw && (d > 0.0) && isfinite(d);
      const double l0 = i0 == col ? d * inv : (i0 > col ? p0[c] * inv : 0.0);
      const double l1 = TWO ? (i1 == col ? d * inv : (i1 > col ? p1[c] * inv : 0.0)) : 0.0;
      const double inv_c = inv;
      p0[c] = l0;
      if (i0 == col) v0 = inv;
      colbuf[i0] = l0;
      if (TWO) { p1[c] = l1; if (i1 == col) v1 = inv; colbuf[i1] = l1; }
      wave_lds_fence();
      if (c + 1 < 8 && c + 1 < nc) {
        const int cn = col + 1;
        const double ln = (!TWO || cn < 64) ? readlane_d(l0, cn & 63) : readlane_d(l1, cn & 63);          // L[cn][col]
        const double pn = (!TWO || cn < 64) ? readlane_d(p0[c + 1], cn & 63) : readlane_d(p1[c + 1], cn & 63);
        d = fma(-ln, ln, pn);
        inv = rsqrt_pos(d);
      }
      // (no test against nc here: columns beyond the panel's end are computed on whatever colbuf holds and never
      // stored -- a uniform branch per column would put every LDS read behind its own wait)
      double m[8];
#pragma unroll
      for (int c2 = c + 1; c2 < 8; ++c2) m[c2] = colbuf[j0 + c2];   // L[j0 + c2][col], same address in every lane
#pragma unroll
      for (int c2 = c + 1; c2 < 8; ++c2) {
        p0[c2] = fma(-l0, m[c2], p0[c2]);
        if (TWO) p1[c2] = fma(-l1, m[c2], p1[c2]);
      }
      // forward substitution: y_col = b_col / L_col,col, b_i -= L_i,col y_col (i > col)
      const double yj = ((!TWO || col < 64) ? readlane_d(b0, col & 63) : readlane_d(b1, col & 63)) * inv_c;
      b0 = i0 == col ? yj : (i0 > col ? b0 - l0 * yj : b0);
      if (TWO) b1 = i1 == col ? yj : (i1 > col ? b1 - l1 * yj : b1);
      wave_lds_fence();
    }
  }
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int col = j0 + c;
    if (c < nc && i0 < S && col <= i0) A[(size_t)i0 * LD + col] = p0[c];
    if (TWO && c < nc && i1 < S && col <= i1) A[(size_t)i1 * LD + col] = p1[c];
  }
}

// Backward substitution L^T x = y on the same wave (lane i: rows i and, TWO, i + 64; v = 1 / L_ii). The factor entries
// a lane needs do not depend on the running solution: they are fetched eight steps ahead and pre-multiplied by
// 1 / L_jj, so that a step is one lane read and one FMA on the dependent chain; x_i = b_i / L_ii is formed at the end.
template <bool TWO>
__device__ __forceinline__ void chol_backward(const double* A, int S, int LD, double& b0, double& b1, double v0, double v1) {
  const int lane = threadIdx.x & 63, i0 = lane, i1 = lane + 64;
  for (int j0 = S - 1; j0 >= 0; j0 -= 8) {
    double a0[8], a1[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int j = j0 - u, jr = j >= 0 ? j : 0;
      // (unconditional loads from row jr, then a select: a conditional load is a branch and a wait of its own)
      const double x0 = A[(size_t)jr * LD + i0];
      const double x1 = TWO ? A[(size_t)jr * LD + (i1 < LD ? i1 : 0)] : 0.0;
      const double vj = (!TWO || jr < 64) ? readlane_d(v0, jr & 63) : readlane_d(v1, jr & 63);
      a0[u] = (j >= 0 && i0 < j) ? x0 * vj : 0.0;
      a1[u] = (TWO && j >= 0 && i1 < j) ? x1 * vj : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int jr = j0 - u >= 0 ? j0 - u : 0;   // (steps below row 0 multiply by the zeros selected above)
      const double bj = (!TWO || jr < 64) ? readlane_d(b0, jr & 63) : readlane_d(b1, jr & 63);   // final: rows > j are done
      b0 -= a0[u] * bj;
      if (TWO) b1 -= a1[u] * bj;
    }
  }
  b0 *= v0;
  if (TWO) b1 *= v1;
}

// Trailing update A[t0.., t0..] -= P P^T (P = the panel's nc <= 4 KS columns from j0, rows t0..S-1) ON THE MATRIX PIPE: the
// lower 16 x 16 tiles of the trailing triangle are dealt to the four waves, KS v_mfma_f64_16x16x4_f64 per tile; per element 3
// LDS operations instead of the 18 of the element-wise form (S = 114: the trailing updates were a third of the solve
// step). Two tiles per round: every LDS read of both (operands and the elements to update) is issued before the first
// matrix instruction -- one LDS round trip and one matrix-pipe latency per pair instead of per tile. All 256 threads call.
// RE: one past the last ROW updated -- S, or S + 1 when the right-hand side rides along as row S of the matrix (chol_block4).
template <int KS>
__device__ __forceinline__ void chol_trail_mfma(double* A, int S, int LD, int j0, int nc, int t0, int RE) {
  const int tid = threadIdx.x;
  const int nt = RE - t0, n16 = (nt + 15) >> 4, ntile = n16 * (n16 + 1) / 2;
  const int wv = tid >> 6, ln = tid & 63, kq = ln >> 4, c16 = ln & 15;
  for (int tb = wv; tb < ntile; tb += 8) {
    double am[2][KS], bm[2][KS], old[2][4];
    int at[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int t = tb + 4 * u;
      const bool live = t < ntile;
      const int tc = live ? t : 0;
      int ti = (int)((sqrtf(8.0f * (float)tc + 1.0f) - 1.0f) * 0.5f);
      ti = ti * (ti + 1) / 2 > tc ? ti - 1 : ti;
      ti = (ti + 1) * (ti + 2) / 2 <= tc ? ti + 1 : ti;
      const int tj = tc - ti * (ti + 1) / 2;
      const int R = t0 + 16 * ti, Cc = t0 + 16 * tj;
      const bool ina = live && R + c16 < RE, inb = live && Cc + c16 < S;
      const int ra = ina ? R + c16 : S - 1, rb = inb ? Cc + c16 : S - 1;      // (unconditional loads, then selects)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int kc = 4 * ks + kq;                                            // panel column of this lane in k-step ks
        const int cc = j0 + kc < S ? j0 + kc : S - 1;                          // (stays inside the LDS block; selected away)
        const double xa = A[(size_t)ra * LD + cc], xb = A[(size_t)rb * LD + cc];
        am[u][ks] = (ina && kc < nc) ? xa : 0.0;
        bm[u][ks] = (inb && kc < nc) ? xb : 0.0;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = R + kq + 4 * r, col = Cc + c16;
        at[u][r] = (live && row < RE && col < S && col <= row) ? row * LD + col : -1;
        old[u][r] = A[at[u][r] >= 0 ? at[u][r] : 0];
      }
    }
    d4 T0 = {0.0, 0.0, 0.0, 0.0}, T1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      T0 = __builtin_amdgcn_mfma_f64_16x16x4f64(am[0][ks], bm[0][ks], T0, 0, 0, 0);
      T1 = __builtin_amdgcn_mfma_f64_16x16x4f64(am[1][ks], bm[1][ks], T1, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (at[0][r] >= 0) A[at[0][r]] = old[0][r] - T0[r];
      if (at[1][r] >= 0) A[at[1][r]] = old[1][r] - T1[r];
    }
  }
}

// Cholesky of the damped reduced system FOUR columns at a time with LOOK-AHEAD (round 4; medium systems, 24 < S <= 63:
// BASELINE configs[4] is S = 42), the right-hand side riding along as row S of the matrix. One barrier per block of four:
//   wave 0 owns the serial chain. Lane l holds the four entries of row j0 + l in the block's columns, fully updated; a
//     column is pivot (lane read) -> rsqrt -> scale -> up to three updates of (lane read + FMA) -- no LDS round trip and
//     no barrier on the chain. It stores the panel, and behind the barrier applies THIS panel's rank-4 update to the NEXT
//     block's four columns itself (sixteen FMAs per row, multipliers by uniform LDS reads) and goes straight on factoring;
//   waves 1..3 meanwhile give the REST of the trailing matrix (columns beyond the next block, the right-hand side's row
//     included) the same rank-4 update on the matrix pipe: one v_mfma_f64_16x16x4_f64 per 16 x 16 tile, operands straight
//     from LDS, tiles fixed for the whole factorisation (addresses and validity computed once per lane).
// What was measured on the way (scripts/time_chol.py, one workgroup, hot, S = 42, shader cycles at 2.41 GHz): round 3's
// eight-column panels on wave 0 + trailing updates 32.2 k (13.4 us; in the solving block 9.0 + 4.8 us); sixteen-column
// register-row panels with lane reads 13.0 + 3.0 us in the solving block (360 dependent lane-read / FMA triples per panel
// on one wave); four-column blocks with the 4 x 4 diagonal block in closed form on every thread, two barriers and the
// trailing update on all four waves 38.1 k, of which the trailing update 20 k (tiles re-anchored per step) / 15 k (fixed
// tiles) -- a dependent fp64 instruction costs ~20 cycles when a SIMD has one wave to run, so what counts is the LENGTH of
// the dependent chain (~12 instructions per column: 42 x 240 cycles = 4.2 us is the floor), and everything that can
// leave the chain's wave must. s_inv[j] receives 1 / L_jj (backward substitution). All 256 threads call; returns whether
// every pivot was positive and finite (valid in every thread).
#ifdef CC_RIG_TIMING
#define B4_MARK(k) do { if (marks) { const long long t_ = wall_clock64(); b4t[k] += t_ - b4last; b4last = t_; } } while (0)
#else
#define B4_MARK(k) do { } while (0)
#endif
__device__ __forceinline__ bool chol_block4(double* A, int S, int LD, double* s_inv, double* marks = nullptr, int ablate = 0) {
  const int tid = threadIdx.x, wv = tid >> 6, ln = tid & 63, kq = ln >> 4, c16 = ln & 15;
  __shared__ int s_okb;
  if (tid == 0) s_okb = 1;
#ifdef CC_RIG_TIMING
  long long b4t[6] = {0, 0, 0, 0, 0, 0}, b4last = wall_clock64();
#endif
  // ---- waves 1..3: the tiles of the trailing update, dealt round robin; fixed rows / columns 16 ti.. / 16 tj.. (ti >= tj)
  const int n16 = (S + 1 + 15) >> 4, ntile = n16 * (n16 + 1) / 2;
  constexpr int kMaxT = 4;   // tiles per wave: ntile <= 10 over three waves (S <= 63)
  int ra[kMaxT], rb[kMaxT], rowmin[kMaxT], colmin[kMaxT], e0[kMaxT];
  bool ina[kMaxT], inb[kMaxT], live[kMaxT];
#pragma unroll
  for (int u = 0; u < kMaxT; ++u) {
    const int t = (wv - 1) + 3 * u;
    live[u] = wv > 0 && t < ntile;
    const int tc = live[u] ? t : 0;
    int ti = (int)((sqrtf(8.0f * (float)tc + 1.0f) - 1.0f) * 0.5f);
    ti = ti * (ti + 1) / 2 > tc ? ti - 1 : ti;
    ti = (ti + 1) * (ti + 2) / 2 <= tc ? ti + 1 : ti;
    const int tj = tc - ti * (ti + 1) / 2;
    const int R = 16 * ti, Cc = 16 * tj;
    ina[u] = live[u] && R + c16 <= S;
    inb[u] = live[u] && Cc + c16 < S;
    ra[u] = (ina[u] ? R + c16 : S) * LD;
    rb[u] = (inb[u] ? Cc + c16 : S - 1) * LD;
    rowmin[u] = R;                        // tile rows R + kq + 4 r, column Cc + c16
    colmin[u] = Cc + c16;
    e0[u] = (R + kq) * LD + Cc + c16;     // element r of this lane: e0 + 4 r LD
  }
  // ---- wave 0: lane l is ROW l of the matrix for the whole factorisation (row S: the right-hand side); x = its entries in the
  // current block's columns, fully updated
  double x[4] = {0.0, 0.0, 0.0, 0.0};
  bool ok = true;
  const double* Row = A + (size_t)(ln <= S ? ln : S) * LD;
  if (wv == 0) {
    const int nb0 = S < 4 ? S : 4;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const double v = Row[c < nb0 ? c : 0];
      x[c] = (ln <= S && c < nb0 && c <= ln) ? v : 0.0;
    }
  }
  for (int j0 = 0; j0 < S; j0 += 4) {
    const int nb = S - j0 < 4 ? S - j0 : 4, t0 = j0 + nb;
    const int nbn = S - t0 < 4 ? S - t0 : 4;   // width of the next block (<= 0: there is none)
    if (wv == 0) {
      // ---- the block's columns, one after the other
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (c < nb) {   // (uniform)
          const double d = readlane_d(x[c], j0 + c);
          ok = ok && (d > 0.0) && isfinite(d);
          const double inv = rsqrt_pos(d);
          const double y = x[c] * inv;            // lane j0 + c: d * inv = L_cc; lanes above it: not part of the column
          x[c] = y;
          if (ln == j0 + c) s_inv[j0 + c] = inv;
#pragma unroll
          for (int c2 = c + 1; c2 < 4; ++c2) x[c2] = fma(-y, readlane_d(y, (j0 + c2) & 63), x[c2]);
        }
      }
      B4_MARK(0);
      if (ln <= S) {
        double* W = A + (size_t)ln * LD + j0;
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (c < nb && j0 + c <= ln) W[c] = x[c];
      }
    }
    __syncthreads();   // panel j0 is in LDS; waves 1..3 have finished the previous block's trailing update
    B4_MARK(1);
    if (t0 >= S) break;
    if (wv == 0) {
      // ---- look-ahead: this panel's update of the NEXT block's columns. The row's entries there (final but for this
      // panel: the barrier) and the sixteen multipliers L[t0 + c][j0 + k] (uniform addresses) come in ONE LDS round trip; the
      // row's own panel entries are the registers x[] (lane = row for the whole factorisation). (Multipliers by lane reads
      // instead, with the products under the round trip of the four entries: 21.4 k cycles against 19.9 k -- a lane read
      // into a scalar register followed by its use costs more than a broadcast LDS read.)
      double xn[4], m[4][4];
#pragma unroll
      for (int c = 0; c < 4; ++c) xn[c] = Row[(c < nbn ? t0 + c : 0)];
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int k = 0; k < 4; ++k) m[c][k] = A[(size_t)(t0 + (c < nbn ? c : 0)) * LD + j0 + (k < nb ? k : 0)];   // L[t0 + c][j0 + k]: uniform address, one round trip for all twenty
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        double a = xn[c];
#pragma unroll
        for (int k = 0; k < 4; ++k) a = fma(-(k < nb ? x[k] : 0.0), m[c][k], a);
        xn[c] = a;
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) x[c] = (ln <= S && c < nbn && t0 + c <= ln) ? xn[c] : 0.0;
      B4_MARK(2);
    } else if (t0 + 4 < S && !(ablate & 1)) {
      // ---- waves 1..3: rank-nb update of the rest, columns >= t0 + 4 (the next block's are wave 0's), rows up to S
      const int kc = j0 + (kq < nb ? kq : 0);   // the lane's panel column (one k-step: column kq)
#pragma unroll
      for (int u0 = 0; u0 < kMaxT; u0 += 2) {
        if ((live[u0] && rowmin[u0] + 15 >= t0 + 4) || (u0 + 1 < kMaxT && live[u0 + 1] && rowmin[u0 + 1] + 15 >= t0 + 4)) {   // (uniform; tiles wholly above the corner are finished)
          double am[2], bm[2], old[2][4];
#pragma unroll
          for (int v = 0; v < 2; ++v) {
            const int u = u0 + v;
            const double xa = A[ra[u] + kc], xb = A[rb[u] + kc];
            am[v] = (ina[u] && kq < nb) ? xa : 0.0;
            bm[v] = (inb[u] && kq < nb) ? xb : 0.0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int e = e0[u] + 4 * r * LD;
              old[v][r] = A[(live[u] && e < (S + 1) * LD) ? e : 0];
            }
          }
          d4 T0 = {0.0, 0.0, 0.0, 0.0}, T1 = {0.0, 0.0, 0.0, 0.0};
          T0 = __builtin_amdgcn_mfma_f64_16x16x4f64(am[0], bm[0], T0, 0, 0, 0);
          T1 = __builtin_amdgcn_mfma_f64_16x16x4f64(am[1], bm[1], T1, 0, 0, 0);
#pragma unroll
          for (int v = 0; v < 2; ++v) {
            const int u = u0 + v;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int row = rowmin[u] + kq + 4 * r, col = colmin[u];
              if (live[u] && col >= t0 + 4 && col < S && row <= S && col <= row) A[e0[u] + 4 * r * LD] = old[v][r] - (v == 0 ? T0[r] : T1[r]);
            }
          }
        }
      }
      B4_MARK(3);
    }
  }
  if (wv == 0 && ln == 0 && !ok) s_okb = 0;
  __syncthreads();
#ifdef CC_RIG_TIMING
  if (marks && tid == 0) { for (int k = 0; k < 5; ++k) marks[k] = (double)b4t[k]; }
#endif
  return s_okb != 0;
}

// Timing-only builds (-DCC_RIG_TIMING, scripts/time_rig_reduce.py): the solving block leaves wall-clock marks
// (100 MHz) in shared_stats[8..]; the product build compiles them away.
#ifdef CC_RIG_TIMING
#define RIG_MARK(i) do { if (threadIdx.x == 0) P.shared_stats[8 + (i)] = (double)wall_clock64(); } while (0)
#else
#define RIG_MARK(i) do { } while (0)
#endif

template <int SRC>
__device__ void rig_solve_block(const RigDev& P, double* smem, const LmCtl* cn_in = nullptr, const double* vec_lds = nullptr, unsigned flag_epoch = 0u) {   // cn_in: the persistent kernels' control block (LDS); vec_lds: their reduced row (SRC 3)
  const int S = P.S, LD = (S + 1) | 1;   // odd row stride: a column walks all LDS banks
  double* A = smem;                       // [S][LD] lower triangle of the reduced system
  double* s_b = A + (size_t)S * LD;       // [128] right-hand side, then the solution x
  double* s_gs = s_b + 128;               // [128] unscaled shared gradient
  double* s_hd = s_gs + 128;              // [128] diagonal of the scaled H_ss
  double* s_inv = s_hd + 128;             // [128] 1 / L_jj
  double* s_ss = s_inv + 128;             // [128] Jacobi scale of the shared block
  __shared__ int s_ok, s_cholok, s_stepok, s_go;
  __shared__ double s4[4];
  __shared__ double s8[8];
  __shared__ LmCtl s_c;
  const int tid = threadIdx.x, lane = tid & 63;
  const LmCtl* cn = cn_in ? cn_in : P.ctl_next;
  const int cur = cn->cur, dst = cur ^ 1;
  const double radius = cn->radius;
  const LmOpts o = *P.opts;
  RigVal val{P, SRC == 2 ? P.x.seq[0] + 1ull : 0ull, wall_clock64(), &s_ok, vec_lds};
  if (tid == 0) { s_ok = 1; s_cholok = 1; s_stepok = 0; s_go = 0; s_c = *cn; }
  for (int i = tid; i < S * LD; i += 256) A[i] = 0.0;
  if (tid < 128) { s_b[tid] = 0.0; s_gs[tid] = 0.0; s_hd[tid] = 0.0; s_inv[tid] = 0.0; s_ss[tid] = tid < S ? P.ss[tid] : 0.0; }
  const int pin = tid < S ? P.colpin[tid] : -1;
  const bool pinned = pin >= 0 && ((P.kmask[pin >> 4] >> (pin & 15)) & 1u) != 0;
  __syncthreads();
  // ---- 1. one pass over the reduced values (loads batched eight deep: one round trip per batch, not per value;
  // where a value goes comes from host-built tables, fetched in the same round trip): per-camera sums of the
  // shared-block entries -> scaled H_ss (lower triangle) and unscaled gradient, minus the Schur products Z^T Z.
  // An element of A gets at most two contributions (one of each kind), added with LDS atomics onto zero: x + y is
  // commutative, so the result does not depend on who comes first. An intrinsics set shared by several cameras
  // is summed along its chain (dir_next), in camera order, by the first camera's thread.
  constexpr int NB = SRC == 2 ? 1 : 8;   // (a mailbox read is a polling loop of its own: no batching there)
  const double fail = val.get<SRC>(P.pc_fail);
  const double gm_r = (tid < P.nranks && tid < 32) ? val.get<SRC>(P.PC + tid) : 0.0;
  for (int e0 = 0; e0 < P.ND; e0 += NB * 256) {
    double v[NB];
    int d[NB], sa[NB], sb[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const int e = e0 + u * 256 + tid;
      v[u] = 0.0; d[u] = -1; sa[u] = 0; sb[u] = 0;
      if (e >= P.ND) continue;
      // (the value is requested together with its table entries, not behind them: one round trip instead of two; the
      // mailbox reader polls and stays conditional)
      double acc = SRC == 2 ? 0.0 : val.get<SRC>(P.pc_dir + e);
      d[u] = P.dir_dst[e];
      sa[u] = P.dir_sa[e]; sb[u] = P.dir_sb[e];
      const int nx = P.dir_next[e];
      if (d[u] == -1) continue;
      if (SRC == 2) acc = val.get<SRC>(P.pc_dir + e);
      for (int n = nx; n >= 0; n = P.dir_next[n]) acc += val.get<SRC>(P.pc_dir + n);
      v[u] = acc;
    }
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      if (d[u] >= 0) {
        const double x = s_ss[sa[u]] * v[u] * s_ss[sb[u]];
        __hip_atomic_fetch_add(&A[d[u]], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (sa[u] == sb[u]) s_hd[sa[u]] = x;
      } else if (d[u] <= -2) {
        s_gs[-2 - d[u]] = v[u];
      }
    }
  }
  for (int i0 = 0; i0 < P.nT * 256; i0 += NB * 256) {
    double v[NB];
    int d[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const int i = i0 + u * 256 + tid;
      v[u] = 0.0; d[u] = -1;
      if (i >= P.nT * 256) continue;
      if (SRC != 2) v[u] = val.get<SRC>(i);   // (with the table entry, not behind it)
      d[u] = P.tile_dst[i];
      if (SRC == 2 && d[u] != -1) v[u] = val.get<SRC>(i);
    }
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      if (d[u] >= 0) __hip_atomic_fetch_add(&A[d[u]], -v[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else if (d[u] <= -2) s_b[-2 - d[u]] = -v[u];
    }
  }
  __syncthreads();
  RIG_MARK(3);
  // ---- 2. right-hand side, LM diagonal, constant coordinates (held intrinsics) become identity rows
  if (tid < S) s_b[tid] = pinned ? 0.0 : s_b[tid] + s_ss[tid] * s_gs[tid];
  __syncthreads();
  if (tid < S) {
    if (pinned) {
      for (int k = 0; k < tid; ++k) A[(size_t)tid * LD + k] = 0.0;
      for (int k = tid + 1; k < S; ++k) A[(size_t)k * LD + tid] = 0.0;
      A[(size_t)tid * LD + tid] = 1.0;
    } else {
      A[(size_t)tid * LD + tid] += clampd(s_hd[tid], o.min_lm_diagonal, o.max_lm_diagonal) / radius;
    }
  }
  // gradient of the accepted point: max-norm over the tangent coordinates (frames: per-rank slots)
  {
    double g = gm_r;
    if (tid < S && !pinned) {
      // Ceres' gradient_max_norm, ||x - Plus(x, -g)||_inf: a camera's pose block through Plus (its first coordinate's thread;
      // pose_grad_proj_max, cc_common.hpp), every other coordinate as it is
      const int info = P.colinfo[tid], kind = (info >> 4) & 15, comp = info & 15;
      if (kind != 0) {
        g = fmax(g, fabs(s_gs[tid]));
      } else if (comp == 0) {
        const double* qc = P.cam + ((size_t)cur * P.C + P.obs_cam[info >> 8]) * 8;
        const double q4[4] = {qc[0], qc[1], qc[2], qc[3]};
        const double g6[6] = {s_gs[tid], s_gs[tid + 1], s_gs[tid + 2], s_gs[tid + 3], s_gs[tid + 4], s_gs[tid + 5]};
        g = fmax(g, pose_grad_proj_max(q4, g6));
      }
    }
    g = wave_max(g);
    if (lane == 0) s4[tid >> 6] = g;
  }
  __syncthreads();
  if (tid == 0) {
    LmCtl c = s_c;
    const double gmax = fmax(fmax(s4[0], s4[1]), fmax(s4[2], s4[3]));
    // (stored whether or not the step was accepted: after a rejected step the accepted point, hence its gradient and this
    // very value, is unchanged -- testing the record's `accepted` flag first was a global load on the solving block's path)
    if (c.log_len > 0 && c.log_len <= P.log_cap) P.log[c.log_len - 1].gradient_max_norm = gmax;
    if (s_ok == 0) { c.done = 1; c.term = CC_FAILURE_EXCHANGE; }
    else if (lm_finalize(c, o, gmax)) s_go = 1;
    if (fail > 0.0) s_cholok = 0;
    s_c = c;
  }
  __syncthreads();
  RIG_MARK(4);
  // Small reduced systems (poses of one to four optimised cameras: S = 6, 12, 18, 24 -- BASELINE configs[3] is S = 18): the
  // whole solve on wave 0 with the matrix distributed by rows over the lanes (chol_solve_rows, cc_device.hpp): pivots and
  // multipliers travel through v_readlane, no LDS vector, no panel loop, no barrier. S = 18: 7.5 -> ~3 us for the
  // factorisation and both substitutions (profiles/r03/rig_stage_marks.jsonl). CC_RIG_PANEL_ONLY=1 (build flag) keeps the
  // panel form for A/B.
#ifndef CC_RIG_PANEL_ONLY
  const bool small_rows = S == 6 || S == 12 || S == 18 || S == 24;
#else
  const bool small_rows = false;
#endif
  if (s_go && small_rows) {
    if (tid < 64) {
      bool okw = true;
      double xs = 0.0;
      auto solve_rows = [&](auto tag) {
        constexpr int SS = decltype(tag)::value;
        const int i = lane < SS ? lane : SS - 1;   // (lanes beyond the system repeat its last row: finite, never read)
        double a[SS], x[SS];
#pragma unroll
        for (int k = 0; k < SS; ++k) a[k] = A[(size_t)i * LD + (k <= i ? k : i)];
        okw = chol_solve_rows<SS>(a, s_b[i], x);
#pragma unroll
        for (int k = 0; k < SS; ++k) xs = lane == k ? x[k] : xs;
      };
      if (S == 6) solve_rows(std::integral_constant<int, 6>{});
      else if (S == 12) solve_rows(std::integral_constant<int, 12>{});
      else if (S == 18) solve_rows(std::integral_constant<int, 18>{});
      else solve_rows(std::integral_constant<int, 24>{});
      const bool fin = lane >= S || isfinite(xs);
      const bool step_ok = s_cholok != 0 && okw && __all(fin);
      if (lane < S) { s_b[lane] = xs; if (SRC != 3) store_ds(P.ds + lane, -xs); }
      if (SRC != 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the update blocks of this launch read ds behind a flag (SRC 3: the step travels in a broadcast)
      if (lane == 0) s_stepok = step_ok ? 1 : 0;
    }
    __syncthreads();
#ifndef CC_RIG_PANEL8
  } else if (s_go && S <= 63) {   // (chol_block4: the right-hand side is row S on lane S of wave 0)
    // ---- medium systems: four columns at a time on all four waves, right-hand side as row S (chol_block4)
#ifdef CC_RIG_TIMING
    const long long tf0 = wall_clock64();
    const long long cy0 = clock64();
#endif
    const bool okb = chol_block4(A, S, LD, s_inv, P.shared_stats + 48);
#ifdef CC_RIG_TIMING
    if (tid == 0) { P.shared_stats[16] = (double)(wall_clock64() - tf0); P.shared_stats[17] = 0.0; P.shared_stats[18] = (double)wall_clock64();
                    P.shared_stats[54] = (double)(clock64() - cy0); P.shared_stats[55] = (double)(wall_clock64() - tf0); }   // shader cycles / 100 MHz ticks: the clock the solving block runs at
#endif
    if (tid < 64) {
      const int i0 = lane;
      double b0 = i0 < S ? s_b[i0] : 0.0, b1 = 0.0;          // y = L^-1 b (row S of the matrix)
      const double v0 = i0 < S ? s_inv[i0] : 0.0;
      chol_backward<false>(A, S, LD, b0, b1, v0, 0.0);
      const bool fin = i0 >= S || isfinite(b0);
      const bool step_ok = s_cholok != 0 && okb && __all(fin);
      if (i0 < S) { s_b[i0] = b0; if (SRC != 3) store_ds(P.ds + i0, -b0); }
      if (SRC != 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the update blocks of this launch read ds behind a flag
      if (lane == 0) s_stepok = step_ok ? 1 : 0;
    }
    __syncthreads();
#ifdef CC_RIG_TIMING
    if (tid == 0) P.shared_stats[19] = (double)wall_clock64();
#endif
#endif
  } else if (s_go) {
    // ---- Cholesky of the damped reduced system in LDS, eight columns at a time (S <= 127):
    //   panel:    wave 0 (chol_panel), forward substitution included;
    //   trailing: all 256 threads, A[i][k] -= sum_c L[i][c] L[k][c] over the panel's columns; two barriers per panel.
    const int i0 = lane, i1 = lane + 64;
    double b0 = 0.0, b1 = 0.0, v0 = 0.0, v1 = 0.0;
    bool okw = true;
    if (tid < 64) { b0 = i0 < S ? s_b[i0] : 0.0; b1 = i1 < S ? s_b[i1] : 0.0; }
#ifdef CC_RIG_TIMING
    long long tw = 0, tt = 0;
#endif
    for (int j0 = 0; j0 < S; j0 += 8) {
      const int nc = S - j0 < 8 ? S - j0 : 8;
#ifdef CC_RIG_TIMING
      const long long ta = wall_clock64();
#endif
      if (tid < 64) {
        if (S <= 64) chol_panel<false>(A, S, LD, j0, nc, s_inv, b0, b1, v0, v1, okw);
        else chol_panel<true>(A, S, LD, j0, nc, s_inv, b0, b1, v0, v1, okw);
      }
      __syncthreads();
#ifdef CC_RIG_TIMING
      const long long tb = wall_clock64();
      tw += tb - ta;
#endif
      const int t0 = j0 + nc;
#ifndef CC_CHOL_TRAIL_VALU
      if (t0 < S && S > 64) {
        // (large systems only -- cameras with their own intrinsics; for S <= 64 the element-wise form below is as fast
        // or faster: S = 18, 48.7 vs 49.6 us per iteration)
        chol_trail_mfma<2>(A, S, LD, j0, nc, t0, S);
      } else if (t0 < S) {
#else
      if (t0 < S) {
#endif
        // trailing triangle rows t0..S-1, columns t0..row, as a flat list of elements dealt to the threads three at a
        // time: all LDS reads of a batch are issued before its first write (the elements are distinct and none lies
        // in the panel's columns, which the compiler cannot know), so a batch costs one LDS round trip, not three
        const int nt = S - t0, ne = nt * (nt + 1) / 2;
        for (int e0 = 0; e0 < ne; e0 += 3 * 256) {
          double acc[3], li[3][8], lk[3][8];
          int at[3];
#pragma unroll
          for (int u = 0; u < 3; ++u) {
            const int e = e0 + u * 256 + tid;
            at[u] = -1;
            acc[u] = 0.0;
#pragma unroll
            for (int c = 0; c < 8; ++c) { li[u][c] = 0.0; lk[u][c] = 0.0; }
            if (e < ne) {
              int n = (int)((sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
              while (n * (n + 1) / 2 > e) --n;
              while ((n + 1) * (n + 2) / 2 <= e) ++n;
              const int i = t0 + n, k = t0 + (e - n * (n + 1) / 2);
              at[u] = i * LD + k;
              acc[u] = A[at[u]];
#pragma unroll
              for (int c = 0; c < 8; ++c) {   // (loads past the panel's end stay inside the LDS block; selected away)
                const double x = A[(size_t)i * LD + j0 + c], y = A[(size_t)k * LD + j0 + c];
                li[u][c] = c < nc ? x : 0.0;
                lk[u][c] = c < nc ? y : 0.0;
              }
            }
          }
#pragma unroll
          for (int u = 0; u < 3; ++u) {
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[u] -= li[u][c] * lk[u][c];
          }
#pragma unroll
          for (int u = 0; u < 3; ++u)
            if (at[u] >= 0) A[at[u]] = acc[u];
        }
      }
      __syncthreads();
#ifdef CC_RIG_TIMING
      tt += wall_clock64() - tb;
#endif
    }
#ifdef CC_RIG_TIMING
    if (tid == 0) { P.shared_stats[16] = (double)tw; P.shared_stats[17] = (double)tt; P.shared_stats[18] = (double)wall_clock64(); }
#endif
    if (tid < 64) {
      if (S <= 64) chol_backward<false>(A, S, LD, b0, b1, v0, v1);
      else chol_backward<true>(A, S, LD, b0, b1, v0, v1);
      const bool fin = (i0 >= S || isfinite(b0)) && (i1 >= S || isfinite(b1));
      const bool step_ok = s_cholok != 0 && __all(okw) && __all(fin);
      if (i0 < S) { s_b[i0] = b0; if (SRC != 3) store_ds(P.ds + i0, -b0); }
      if (i1 < S) { s_b[i1] = b1; if (SRC != 3) store_ds(P.ds + i1, -b1); }
      if (SRC != 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the update blocks of this launch read ds behind a flag
      if (lane == 0) s_stepok = step_ok ? 1 : 0;
    }
    __syncthreads();
#ifdef CC_RIG_TIMING
    if (tid == 0) P.shared_stats[19] = (double)wall_clock64();
#endif
  }
  RIG_MARK(5);
  const bool have_step = s_go != 0 && s_stepok != 0;
  // Fused launch (k_rig_reduce<0>, flag_epoch != 0): the blocks waiting to update their frames need the shared step -- stored
  // and drained above -- and three bits of the control block that are final by now (done and cur do not change below,
  // step_valid is have_step): the flag goes up HERE, and the frame updates run under the camera candidates, block sums and
  // control block below instead of behind them (2.8 us at BASELINE configs[4] size). The next kernel reads the control block;
  // this one is not over before it is written.
  if (flag_epoch != 0u && tid == 0) {
    const unsigned fl = (flag_epoch << 3) | (s_c.done ? 4u : 0u) | ((s_go ? have_step : (s_c.step_valid != 0)) ? 2u : 0u) | (unsigned)(s_c.cur & 1);
    __hip_atomic_store(P.arrive + 1, fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // ---- camera / intrinsics candidates and records (nothing moves unless a valid step exists)
  double st2 = 0.0, xs2 = 0.0;
  if (have_step) rig_candidates(P, s_b, s_ss, true, cur, dst, st2, xs2);
  {   // both block sums behind one pair of barriers
    const double a = wave_sum(st2), b2 = wave_sum(xs2);
    __syncthreads();
    if (lane == 0) { s8[tid >> 6] = a; s8[4 + (tid >> 6)] = b2; }
    __syncthreads();
  }
  const double st = (s8[0] + s8[1]) + (s8[2] + s8[3]);
  const double xs = (s8[4] + s8[5]) + (s8[6] + s8[7]);
  if (tid == 0) {
    LmCtl c = s_c;
    if (s_go) {
      c.step_valid = have_step ? 1 : 0;
      c.cand_pending = 1;
      P.shared_stats[0] = st;
      P.shared_stats[1] = xs;
    }
    if (SRC == 2) P.x.seq[0] = val.epoch;
    *P.ctl = c;
    *P.ctl_next = c;
  }
  RIG_MARK(6);
}

// Hands the control block (and the failure word of the in-kernel waits) to the host without a copy engine in the way:
// payload first, then the sequence word the host spins on (system-scope stores into pinned host memory; one thread).
// Last reduce launch of a host chunk only (cf. publish_to_host, cc_intrinsics_dev.hpp).
__device__ __forceinline__ void rig_publish(const RigDev& P, const LmCtl& c) {
  if (!P.host_pub) return;
  const unsigned long long* w = reinterpret_cast<const unsigned long long*>(&c);
#pragma unroll
  for (int i = 0; i < (int)(sizeof(LmCtl) / 8); ++i)
    __hip_atomic_store(P.host_pub + 2 + i, w[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  const unsigned failed = __hip_atomic_load(P.arrive + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(P.host_pub + 2 + sizeof(LmCtl) / 8, (unsigned long long)failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  const unsigned long long seq = *P.pub_seq + 1ull;
  *P.pub_seq = seq;
  __hip_atomic_store(P.host_pub, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// The solve step as a kernel of its own.
//   SRC 0 (RCCL route): after the all-reduce of P.vec.
//   SRC 2 (mailbox exchange on a device this rank SHARES with other shards or processes: rig_enqueue_round): this ONE
//          block collects every rank's posts (k_rig_reduce<4> made ours) in rank order inside rig_solve_block, and -- last
//          launch of a host chunk but for the pose update -- publishes the control block to the host. No block of any
//          launch of this form waits for another block: the only waits are this block's polls of its own mailbox.
template <int SRC>
__global__ __launch_bounds__(256) void k_rig_solve(RigDev P, int publish) {
  rig_progress(P, RIG_PROG_SOLVE);
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const LmCtl* cn = P.ctl_next;
  if (cn->done) {
    if (threadIdx.x == 0) { *P.ctl = *cn; if (SRC == 2 && publish) rig_publish(P, *cn); }
    return;
  }
  if (cn->phase == 0) return;
  rig_solve_block<SRC>(P, reinterpret_cast<double*>(smem_raw));
  if (SRC == 2 && publish) {
    __syncthreads();
    if (threadIdx.x == 0) rig_publish(P, *P.ctl);   // (written by this very thread a moment ago)
  }
}

// Large rigs on the mailbox exchange: k_rig_reduce<4> posted this rank's column sums into every mailbox; ONE block collects
// all ranks' posts in rank order into P.vec, which k_rig_solve_big then reads as it does after an all-reduce.
__global__ __launch_bounds__(256) void k_rig_collect(RigDev P) {
  __shared__ int s_ok;
  const LmCtl* cn = P.ctl_next;
  if (cn->done || cn->phase == 0) return;
  const unsigned long long epoch = P.x.seq[0] + 1ull;
  p2p_collect_to(P.x, 0, epoch, P.rank, P.nranks, P.PC + 32, P.vec, &s_ok);
  if (threadIdx.x == 0) {
    P.x.seq[0] = epoch;
    if (s_ok == 0) {
      LmCtl c = *cn;
      c.done = 1; c.term = CC_FAILURE_EXCHANGE;
      *P.ctl = c; *P.ctl_next = c;
    }
  }
}

// first launch of a solve: camera / intrinsics records of the starting point (what the first sweep reads)
__global__ __launch_bounds__(256) void k_rig_records(RigDev P) {
  const LmCtl* ctl = P.ctl;
  if (ctl->done || ctl->phase != 0) return;
  double a, b;
  rig_candidates(P, nullptr, nullptr, false, ctl->cur, ctl->cur, a, b);
}

// ---------------------------------------------------------------------------------------------
// reduce (+ solve + update): column sums (max for the last column) of the elimination partial rows, 16 columns
// per block and step, 16 row groups per column, 16 loads in flight per thread. Deterministic.
// MODE 0 (single GPU): the sums are stored write-through, the block arrives on a counter and the LAST block
// to arrive runs the solve step on them (sc1 loads, no fence: MI355X guide, valid hand-off forms).
// MODE 3 (mailbox exchange): every block posts its sums straight into all ranks' mailboxes; the last block
// to arrive collects them in rank order inside the solve step. MODE 2 (RCCL): sums -> P.vec, nothing else.
// MODE 4 (mailbox exchange on a SHARED device): sums -> every rank's mailbox, nothing else -- the solve step and the pose
// update are launches of their own (k_rig_solve<2>, k_rig_update), so no block waits for another one.
// MODES 0 and 3 then run the POSE UPDATE in the same launch: the grid is at most one block per CU (all of them
// resident), the blocks that are not last wait for a flag word the solver stores (epoch | done | step_valid | cur,
// sc1, behind its drained sc1 stores of the shared step) and every block updates its share of the frames. The wait is
// bounded (10 s of the wall clock) like the mailbox polls.
// ---------------------------------------------------------------------------------------------

template <int MODE>
__global__ __launch_bounds__(256) void k_rig_reduce(RigDev P, int publish) {
  rig_progress(P, RIG_PROG_REDUCE);
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  __shared__ double s_r[16][16];
  __shared__ double s_post[48];
  __shared__ double s_tail;
  __shared__ int s_last;
  __shared__ unsigned s_flag;
  constexpr bool FUSED = MODE == 0 || MODE == 3;   // solve step + pose update in this launch (its blocks wait for each other)
  const LmCtl* cn = P.ctl_next;
  if (cn->done) {
    if (FUSED && blockIdx.x == 0 && threadIdx.x == 0) { *P.ctl = *cn; if (publish) rig_publish(P, *cn); }
    return;
  }
  if (cn->phase == 0) return;
  // an earlier launch of this solve gave up waiting (below): the state is half updated, the host will report it
  // (rig_wait); do not wait another ten seconds per remaining round of the chunk
  if (FUSED && __hip_atomic_load(P.arrive + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
    if (publish && blockIdx.x == 0 && threadIdx.x == 0) rig_publish(P, *cn);
    return;
  }
  const int tid = threadIdx.x, c = tid & 15, grp = tid >> 4;  // 16 columns x 16 row groups per step
#ifdef CC_RIG_TIMING
  const long long t_entry = wall_clock64();
#endif
  unsigned epoch0 = 0;
  if (FUSED) epoch0 = __hip_atomic_load(P.arrive + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 3;   // before we arrive
  for (int first = blockIdx.x * 16; first < P.PC; first += gridDim.x * 16) {
    const int o = first + c;
    const bool is_max = o == P.pc_gmax;
    double a = 0.0;
    if (o < P.PC) {
      for (int r0 = grp; r0 < P.nblk; r0 += 256) {
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = r0 + 16 * u < P.nblk ? P.partial[(size_t)(r0 + 16 * u) * P.PC + o] : 0.0;
        if (is_max) {
#pragma unroll
          for (int u = 0; u < 16; ++u) a = fmax(a, v[u]);
        } else {
          a += (((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]))) +
               (((v[8] + v[9]) + (v[10] + v[11])) + ((v[12] + v[13]) + (v[14] + v[15])));
        }
      }
    }
    __syncthreads();   // (readers of the previous step)
    s_r[grp][c] = a;
    __syncthreads();
    if (tid < 16 && o < P.PC) {
      double r = 0.0;
      if (is_max) { for (int g2 = 0; g2 < 16; ++g2) r = fmax(r, s_r[g2][c]); }
      else {
        r = (((s_r[0][c] + s_r[1][c]) + (s_r[2][c] + s_r[3][c])) + ((s_r[4][c] + s_r[5][c]) + (s_r[6][c] + s_r[7][c]))) +
            (((s_r[8][c] + s_r[9][c]) + (s_r[10][c] + s_r[11][c])) + ((s_r[12][c] + s_r[13][c]) + (s_r[14][c] + s_r[15][c])));
      }
      unsigned long long* vw = reinterpret_cast<unsigned long long*>(P.vec);
      if (is_max) {   // the per-rank slot carries the max (a sum exchange then keeps it); the column itself is 0
        if (MODE == 0) __hip_atomic_store(vw + P.PC + P.rank, (unsigned long long)__double_as_longlong(r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else P.vec[P.PC + P.rank] = r;
        s_tail = r;
        r = 0.0;
      }
      if (MODE == 0) __hip_atomic_store(vw + o, (unsigned long long)__double_as_longlong(r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else P.vec[o] = r;
      s_post[c] = r;
    }
    if (MODE == 3 || MODE == 4) {
      // mailbox exchange (kind 0): the block that owns the max column also posts the 32 per-rank max slots
      // (ours set, the others zero). The epoch is stable here: only the solve step advances it.
      __syncthreads();
      const unsigned long long epoch = P.x.seq[0] + 1ull;
      const int ncol = P.PC - first < 16 ? P.PC - first : 16;
      p2p_post(P.x, 0, epoch, P.rank, P.nranks, s_post, ncol, first);
      if (first <= P.pc_gmax && P.pc_gmax < first + 16) {
        if (tid < 32) s_post[16 + tid] = tid == P.rank ? s_tail : 0.0;
        __syncthreads();
        p2p_post(P.x, 0, epoch, P.rank, P.nranks, s_post + 16, 32, P.PC);
      }
    }
  }
  if (!FUSED) return;
#ifdef CC_RIG_TIMING
  const long long t_sums = wall_clock64();
#endif
  // ---- last-block-done: every storing wave drains its stores, the block arrives, the last one solves
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    const unsigned prev = __hip_atomic_fetch_add(P.arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = prev + 1u == gridDim.x;
  }
  __syncthreads();
  // every block -- the solving one included -- requests what the update of its first sixteen frames needs NOW, before it
  // waits (or solves): behind the flag only the shared step is still to be read
  RigUpdPre pre;
  const bool use_pre = P.SW <= 64 && (int64_t)blockIdx.x * 16 < P.F;
  rig_update_prefetch(P, (int64_t)blockIdx.x * 16 + (tid >> 4), pre);   // (unconditional: loads inside an `if` would be waited for at its end)
  if (s_last) {
    if (tid == 0) __hip_atomic_store(P.arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // next launch
#ifdef CC_RIG_TIMING
    if (tid == 0) { P.shared_stats[8] = (double)t_entry; P.shared_stats[9] = (double)t_sums; }
#endif
    RIG_MARK(2);
    rig_solve_block<MODE == 0 ? 1 : 2>(P, reinterpret_cast<double*>(smem_raw), nullptr, nullptr, MODE == 0 ? epoch0 + 1u : 0u);   // (MODE 0: raises the flag itself, early)
    // the shared step (sc1 stores of wave 0) has been drained inside; hand the outcome to the waiting blocks
    __syncthreads();
    if (tid == 0) {
      const LmCtl* c = P.ctl;   // written by this very thread a moment ago
      s_flag = ((epoch0 + 1u) << 3) | (c->done ? 4u : 0u) | (c->step_valid ? 2u : 0u) | (unsigned)(c->cur & 1);
      __hip_atomic_store(P.arrive + 1, s_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // (the host only decides on it whether another chunk follows; whoever reads poses synchronises the stream first)
      if (publish) rig_publish(P, *c);
    }
  }
  if (!s_last && tid == 0) {
    const long long t0 = wall_clock64();
    unsigned f;
    for (;;) {
      f = __hip_atomic_load(P.arrive + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((f >> 3) == epoch0 + 1u) break;
      if (wall_clock64() - t0 > kP2pTimeoutTicks) {
        // The solving block did not publish within 10 s: the blocks of this launch were not all resident (the grid is sized
        // for that at launch, rig_reduce_blocks) or the solve step waits for a peer rank. Leave without updating and SAY SO:
        // the failure word makes every later launch a no-op and the host return CC_ERR_COMM (rig_wait).
        __hip_atomic_store(P.arrive + 3, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        f = 4u;
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
    s_flag = f;
  }
  __syncthreads();
  const unsigned flag = s_flag;
  if ((flag & 4u) || !(flag & 2u)) return;   // done, or no valid step: the poses stay
  const int cur = (int)(flag & 1u);
  if (use_pre) rig_update_body<true, true>(P, 1, cur, (int64_t)blockIdx.x * 16 + (tid >> 4), pre);
  for (int64_t fblk = use_pre ? (int64_t)blockIdx.x + gridDim.x : (int64_t)blockIdx.x; fblk * 16 < P.F; fblk += gridDim.x)
    rig_update_body<true, false>(P, 1, cur, fblk * 16 + (tid >> 4), pre);
#ifdef CC_RIG_TIMING
  __syncthreads();
  if (threadIdx.x == 0 && s_last) P.shared_stats[15] = (double)wall_clock64();   // mark 7 (written this way: RIG_MARK(7) inside `if (s_last)` trips a register-class bug of the compiler)
#endif
}

// =============================================================================================
// LARGE reduced systems (128 <= S <= 255: more than 21 optimised cameras, or more than 8 with intrinsics of their own --
// or more direct sums than k_rig_elim keeps, 12+ observed cameras with intrinsics; the reference takes any number of
// cameras, extrinsics_calibrator.cpp:9-17). The kernels above are built around
// S + 1 <= 128 (two shared columns per lane, nine tile accumulators per wave, the reduced system in LDS with a row stride);
// rather than bend them, such problems run the same arithmetic in a plainer form -- correctness first, no tuning:
//   k_rig_elim_big : one block per frame at a time, thread k owns shared column k (S + 1 <= 256); the 6 x 6 factor is
//                    computed by every thread; Schur products Z^T Z accumulated per 16 x 16 tile with plain FMAs, entry
//                    `tid` of every tile in a register (<= 136 tiles); the direct sums in LDS. Same partial-row layout.
//   k_rig_reduce<2>: the column sums (unchanged) -> P.vec
//   k_rig_solve_big: one block; the reduced system as a lower triangle packed by rows in LDS (S <= 193) or column-major
//                    in global memory, the right-hand side as row S; left-looking Cholesky, thread i owns row i, sixteen
//                    columns of both rows per round trip, two barriers per column; backward substitution with one
//                    barrier per step; the tests, candidates and control block of rig_solve_block.
//   k_rig_update   : unchanged.
// Sweep, init, records, statistics: unchanged (their shared-column arrays hold 256 entries).
// =============================================================================================
constexpr int kRigBigMaxS = 255;
constexpr int kRigBigTiles = 136;   // upper tile pairs of a 16 x 16 tile grid
__host__ __device__ constexpr int big_tile(int a, int b) { return a * 16 - a * (a - 1) / 2 + (b - a); }   // (a <= b < 16)

template <bool HK>
__global__ __launch_bounds__(256) void k_rig_elim_big(RigDev P) {
  rig_progress(P, RIG_PROG_ELIM);
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* s_Z = reinterpret_cast<double*>(smem_raw);   // [6][256] staged Z rows of the block's current frame
  double* s_d = s_Z + 6 * 256;                          // [ND] direct sums of the block
  __shared__ double s_A[32];
  __shared__ double s_ss[kRigBigMaxS + 1];
  __shared__ double s16[16];
  __shared__ double s_tot[4];
  __shared__ double s_fg[8];
  __shared__ int s_g[64];
  __shared__ LmCtl s_ctl;
  const int tid = threadIdx.x;
  const LmCtl* ctl = P.ctl;
  if (ctl->done || ctl->phase == 0) return;
  // ---- trust-region decision: every block, same answer; block 0 publishes it (as in k_rig_elim)
  const bool pending = ctl->cand_pending != 0;
  if (P.comm) {   // (sharded: k_rig_stats exchanged them)
    if (tid < 4) s_tot[tid] = P.vec_stats[tid];
    __syncthreads();
  } else {
    rig_reduce_stats(P, pending && ctl->step_valid, s16, s_tot);
  }
  if (tid == 0) {
    LmCtl c = *ctl;
    const LmOpts o = *P.opts;
    if (pending) {
      double step2 = s_tot[2], xn2 = s_tot[3];
      if (c.step_valid) { step2 += P.shared_stats[0]; xn2 += P.shared_stats[1]; }
      cc_iteration rec;
      const int len0 = c.log_len;
      lm_decide(c, o, &rec, s_tot[0], s_tot[1], step2, xn2);
      if (blockIdx.x == 0 && c.log_len != len0 && c.log_len <= P.log_cap) P.log[c.log_len - 1] = rec;
    }
    s_ctl = c;
    if (blockIdx.x == 0) *P.ctl_next = c;
  }
  if (tid < P.S) s_ss[tid] = P.ss[tid];
  for (int i = tid; i < 6 * 256; i += 256) s_Z[i] = 0.0;
  for (int i = tid; i < P.ND; i += 256) s_d[i] = 0.0;
  __syncthreads();
  if (s_ctl.done) return;
  const int cur = s_ctl.cur;
  const double inv_radius = 1.0 / s_ctl.radius;
  const double mn = P.opts->min_lm_diagonal, mx = P.opts->max_lm_diagonal;
  const bool first_elim = ctl->phase == 1 && s_ctl.iter == 0 && !pending;
  const bool jac = P.opts->jacobi_scaling != 0;
  const int SW = P.SW, S = P.S, CO = P.CO, T = P.T;
  const size_t gs = (size_t)P.gstride;
  const double* blocks = P.gblocks + (size_t)cur * P.NG * gs;
  // this thread's shared column and frame-block entry
  int c_kind = -1, c_co = 0, c_comp = 0;
  if (tid < SW) { const int info = P.colinfo[tid]; c_kind = (info >> 4) & 15; c_co = info >> 8; c_comp = info & 15; }
  const double c_ss = tid < S ? s_ss[tid] : (tid < SW ? 1.0 : 0.0);
  int a_off = 0, sp_i = -1;
  if (tid < 21) {
    int i = 0;
    while (tri(i + 1, 0) <= tid) ++i;
    const int j = tid - tri(i, 0);
    a_off = (6 + i) * 16 + 6 + j;
    if (i == j) sp_i = i;
  } else if (tid < 27) {
    a_off = (6 + (tid - 21)) * 16 + 12;
  }
  double acc[kRigBigTiles];
#pragma unroll
  for (int t = 0; t < kRigBigTiles; ++t) acc[t] = 0.0;
  // (failure count and gradient maximum of the block live in LDS, s_fg[0] / s_fg[1]: thread 0 alone touches them)
  if (tid == 0) { s_fg[0] = 0.0; s_fg[1] = 0.0; }
  const int tr = tid >> 4, tc = tid & 15;
  for (int64_t f = blockIdx.x; f < P.F; f += gridDim.x) {
    if (tid < 64) s_g[tid] = tid < CO ? P.fslot[f * CO + tid] : -1;
    __syncthreads();
    bool live = false;
    for (int j = 0; j < CO; ++j) live = live || s_g[j] >= 0;
    if (live) {
      if (tid < 27) {
        // (eight loads per round trip, unconditional from a clamped group, then selects: one load per wait took 20 us of a
        // frame's 32 with 40 observed cameras; the sum keeps its order)
        double a_e = 0.0;
        for (int j0 = 0; j0 < CO; j0 += 8) {
          double v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int g = j0 + u < CO ? s_g[j0 + u] : -1;
            const double x = blocks[(size_t)(g >= 0 ? g : 0) * gs + a_off];
            v[u] = g >= 0 ? x : 0.0;
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) a_e += v[u];
        }
        s_A[tid] = a_e;
        if (first_elim && sp_i >= 0) P.sp[f * 8 + sp_i] = jac ? 1.0 / (1.0 + sqrt(a_e)) : 1.0;
      }
      __syncthreads();
      double A[27], sf[6];
#pragma unroll
      for (int i = 0; i < 27; ++i) A[i] = s_A[i];
#pragma unroll
      for (int i = 0; i < 6; ++i) sf[i] = first_elim ? (jac ? 1.0 / (1.0 + sqrt(A[tri(i, i)])) : 1.0) : P.sp[f * 8 + i];
      double L[21], Li[6];
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) L[tri(i, j)] = sf[i] * A[tri(i, j)] * sf[j];
#pragma unroll
      for (int i = 0; i < 6; ++i) L[tri(i, i)] += clampd(L[tri(i, i)], mn, mx) * inv_radius;
      bool ok = true;
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        double d = L[tri(j, j)];
#pragma unroll
        for (int k = 0; k < j; ++k) d -= L[tri(j, k)] * L[tri(j, k)];
        ok = ok && (d > 0.0) && isfinite(d);
        const double inv = rsqrt_pos(d);
        L[tri(j, j)] = d * inv;
        Li[j] = inv;
#pragma unroll
        for (int i = j + 1; i < 6; ++i) {
          double a = L[tri(i, j)];
#pragma unroll
          for (int k = 0; k < j; ++k) a -= L[tri(i, k)] * L[tri(j, k)];
          L[tri(i, j)] = a * inv;
        }
      }
      if (tid == 0) {
        if (!ok) s_fg[0] += 1.0;
        const double* fqp = P.pose + ((size_t)cur * P.F + f) * 8;
        const double q4[4] = {fqp[0], fqp[1], fqp[2], fqp[3]};
        s_fg[1] = fmax(s_fg[1], pose_grad_proj_max(q4, &A[21]));   // Ceres' gradient_max_norm (cc_common.hpp)
      }
      if (tid < SW) {
        double w[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        if (c_kind == 0 || c_kind == 1) {
          const int g = s_g[c_co];
          if (g >= 0) {
            const double* G = blocks + (size_t)g * gs;
#pragma unroll
            for (int i = 0; i < 6; ++i) w[i] = c_kind == 0 ? G[c_comp * 16 + 6 + i] : G[256 + (6 + i) * 16 + c_comp];
          }
        } else if (HK && c_kind == 2) {
          for (int j = 0; j < CO; ++j) {
            const int g = s_g[j];
            if (g < 0) continue;
            const double* G = blocks + (size_t)g * gs + 256;
#pragma unroll
            for (int i = 0; i < 6; ++i) w[i] += G[(6 + i) * 16 + c_comp];
          }
        }
        double z[6], y[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          double a = c_kind == 3 ? sf[i] * A[21 + i] : sf[i] * w[i] * c_ss;
#pragma unroll
          for (int kk = 0; kk < i; ++kk) a -= L[tri(i, kk)] * z[kk];
          z[i] = a * Li[i];
        }
#pragma unroll
        for (int i = 5; i >= 0; --i) {
          double a = z[i];
#pragma unroll
          for (int kk = i + 1; kk < 6; ++kk) a -= L[tri(kk, i)] * y[kk];
          y[i] = a * Li[i];
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          s_Z[i * 256 + tid] = z[i];
          P.Y[((size_t)f * 6 + i) * SW + tid] = y[i];
        }
      }
      for (int e0 = tid; e0 < P.ND; e0 += 4 * 256) {   // (four entries per round trip)
        double v[4];
        bool k[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int e = e0 + 256 * u;
          const int t = P.dent[e < P.ND ? e : 0];
          const int g = s_g[t >> 16];
          k[u] = e < P.ND && g >= 0;
          v[u] = blocks[(size_t)(g >= 0 ? g : 0) * gs + (t & 0xffff)];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (k[u]) s_d[e0 + 256 * u] += v[u];
      }
      __syncthreads();
      // Schur products of this frame: entry (tr, tc) of every upper tile pair (a, b), a <= b < T. The loops run over the
      // largest tile grid with compile-time accumulator indices (no tables: 136 pairs of table entries in scalar registers
      // spilled hundreds of them); which pairs exist is a uniform test.
#pragma unroll
      for (int a = 0; a < 16; ++a) {
        if (a < T) {
          double za[6];
#pragma unroll
          for (int i = 0; i < 6; ++i) za[i] = s_Z[i * 256 + 16 * a + tr];
#pragma unroll
          for (int b = a; b < 16; ++b) {
            if (b < T) {
              double x = acc[big_tile(a, b)];
#pragma unroll
              for (int i = 0; i < 6; ++i) x = fma(za[i], s_Z[i * 256 + 16 * b + tc], x);
              acc[big_tile(a, b)] = x;
            }
          }
        }
      }
    }
    __syncthreads();
  }
  double* prow = P.partial + (size_t)blockIdx.x * P.PC;
  // (the partial row numbers the pairs of the T x T grid in the same order: a, then b)
#pragma unroll
  for (int a = 0; a < 16; ++a)
#pragma unroll
    for (int b = a; b < 16; ++b)
      if (b < T) prow[(size_t)(a * T - a * (a - 1) / 2 + (b - a)) * 256 + tid] = acc[big_tile(a, b)];
  for (int e = tid; e < P.ND; e += 256) prow[P.pc_dir + e] = s_d[e];
  if (tid == 0) { prow[P.pc_fail] = s_fg[0]; prow[P.pc_gmax] = s_fg[1]; }
}

// accessor of the reduced system's lower triangle (rows 0..S, row S = right-hand side; S columns). In LDS: packed by
// rows -- thread i owns row i, a batch of its entries is one base address plus immediates. In global memory (L2-resident:
// 0.5 MB at S = 255): ROW-major with the stride the host's destination tables use -- a thread's sixteen panel entries are
// 128 contiguous bytes, the sixteen columns of a trailing tile's row one transaction (round 3 kept it column-major for its
// left-looking factorisation, one row per thread).
constexpr int kRigBigPanelDoubles = 256 * 17 + 64;
template <bool PACKED>
struct BigA {
  double* p; int LD;
  __device__ __forceinline__ double& at(int i, int k) const {   // k <= i <= S, k < S
    return PACKED ? p[i * (i + 1) / 2 + k] : p[(size_t)i * LD + k];
  }
  // host-built destinations are row * LD + col (rig_layout)
  __device__ __forceinline__ double& at_dst(int dst) const { const int i = dst / LD, k = dst - i * LD; return at(i, k); }
};

template <bool PACKED>
__global__ __launch_bounds__(256) void k_rig_solve_big(RigDev P, double* Aglobal) {
  rig_progress(P, RIG_PROG_SOLVE);
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const LmCtl* cn = P.ctl_next;
  if (cn->done) {
    if (threadIdx.x == 0) *P.ctl = *cn;
    return;
  }
  if (cn->phase == 0) return;
  const int S = P.S, LD = (S + 1) | 1;
  double* lds = reinterpret_cast<double*>(smem_raw);
  double* s_b = lds;                  // [256] right-hand side -> y -> x
  double* s_gs = s_b + 256;           // [256] unscaled shared gradient
  double* s_hd = s_gs + 256;          // [256] diagonal of the scaled H_ss
  double* s_inv = s_hd + 256;         // [256] 1 / L_jj
  double* s_ss = s_inv + 256;         // [256]
  double* s_pan = s_ss + 256;         // [256][17] the factorisation's panel, then [64] micro-block words
  BigA<PACKED> A{PACKED ? s_pan + kRigBigPanelDoubles : Aglobal, LD};
  __shared__ int s_cholok, s_stepok, s_go;
  __shared__ double s4[4];
  __shared__ double s8[8];
  __shared__ double s_r;
  __shared__ LmCtl s_c;
  const int tid = threadIdx.x, lane = tid & 63;
  const int cur = cn->cur, dst = cur ^ 1;
  const double radius = cn->radius;
  const LmOpts o = *P.opts;
  if (tid == 0) { s_cholok = 1; s_stepok = 0; s_go = 0; s_c = *cn; }
  for (int i = tid; i <= S; i += 256)   // (row S: the right-hand side)
    for (int k = 0; k <= i && k < S; ++k) A.at(i, k) = 0.0;
  s_b[tid] = 0.0; s_gs[tid] = 0.0; s_hd[tid] = 0.0; s_inv[tid] = 0.0; s_ss[tid] = tid < S ? P.ss[tid] : 0.0;
  const int pin = tid < S ? P.colpin[tid] : -1;
  const bool pinned = pin >= 0 && ((P.kmask[pin >> 4] >> (pin & 15)) & 1u) != 0;
  __syncthreads();
  // ---- assembly from the column sums (P.vec): direct sums, then minus the Schur products. An element gets at most one
  // contribution of each kind; the two loops are separated by a barrier, so plain read-modify-write is safe. Loads are
  // batched eight deep (table entry and value together, then the eight elements): 30720 tile entries at S = 234 were 120
  // dependent round trips per thread one at a time -- the largest piece of the launch once the factorisation was blocked.
  for (int e0 = 0; e0 < P.ND; e0 += 8 * 256) {
    int d[8], sa[8], sb[8];
    double acc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + u * 256 + tid, ec = e < P.ND ? e : 0;
      d[u] = P.dir_dst[ec]; sa[u] = P.dir_sa[ec]; sb[u] = P.dir_sb[ec];
      acc[u] = P.vec[P.pc_dir + ec];
      if (e >= P.ND) d[u] = -1;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (d[u] == -1) continue;
      const int e = e0 + u * 256 + tid;
      for (int n = P.dir_next[e]; n >= 0; n = P.dir_next[n]) acc[u] += P.vec[P.pc_dir + n];
      if (d[u] >= 0) {
        const double x = s_ss[sa[u]] * acc[u] * s_ss[sb[u]];
        A.at_dst(d[u]) = x;          // (zeroed above, one direct contribution at most: a plain store)
        if (sa[u] == sb[u]) s_hd[sa[u]] = x;
      } else {
        s_gs[-2 - d[u]] = acc[u];
      }
    }
  }
  __syncthreads();
  for (int i0 = 0; i0 < P.nT * 256; i0 += 8 * 256) {
    int d[8];
    double v[8], old[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u * 256 + tid, ic = i < P.nT * 256 ? i : 0;
      d[u] = P.tile_dst[ic];
      v[u] = P.vec[ic];
      if (i >= P.nT * 256) d[u] = -1;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) old[u] = A.at_dst(d[u] >= 0 ? d[u] : 0);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (d[u] >= 0) A.at_dst(d[u]) = old[u] - v[u];
      else if (d[u] <= -2) s_b[-2 - d[u]] = -v[u];
    }
  }
  const double fail = P.vec[P.pc_fail];
  const double gm_r = (tid < P.nranks && tid < 32) ? P.vec[P.PC + tid] : 0.0;
  __syncthreads();
  if (tid < S) s_b[tid] = pinned ? 0.0 : s_b[tid] + s_ss[tid] * s_gs[tid];
  __syncthreads();
  if (tid < S) {
    if (pinned) {
      for (int k = 0; k < tid; ++k) A.at(tid, k) = 0.0;
      for (int k = tid + 1; k < S; ++k) A.at(k, tid) = 0.0;
      A.at(tid, tid) = 1.0;
    } else {
      A.at(tid, tid) += clampd(s_hd[tid], o.min_lm_diagonal, o.max_lm_diagonal) / radius;
    }
  }
  {
    double g = gm_r;
    if (tid < S && !pinned) {
      // Ceres' gradient_max_norm, ||x - Plus(x, -g)||_inf: a camera's pose block through Plus (its first coordinate's thread;
      // pose_grad_proj_max, cc_common.hpp), every other coordinate as it is
      const int info = P.colinfo[tid], kind = (info >> 4) & 15, comp = info & 15;
      if (kind != 0) {
        g = fmax(g, fabs(s_gs[tid]));
      } else if (comp == 0) {
        const double* qc = P.cam + ((size_t)cur * P.C + P.obs_cam[info >> 8]) * 8;
        const double q4[4] = {qc[0], qc[1], qc[2], qc[3]};
        const double g6[6] = {s_gs[tid], s_gs[tid + 1], s_gs[tid + 2], s_gs[tid + 3], s_gs[tid + 4], s_gs[tid + 5]};
        g = fmax(g, pose_grad_proj_max(q4, g6));
      }
    }
    g = wave_max(g);
    if (lane == 0) s4[tid >> 6] = g;
  }
  __syncthreads();
  if (tid == 0) {
    LmCtl c = s_c;
    const double gmax = fmax(fmax(s4[0], s4[1]), fmax(s4[2], s4[3]));
    if (c.log_len > 0 && c.log_len <= P.log_cap) P.log[c.log_len - 1].gradient_max_norm = gmax;
    if (lm_finalize(c, o, gmax)) s_go = 1;
    if (fail > 0.0) s_cholok = 0;
    s_c = c;
  }
  __syncthreads();
  if (s_go) {
    // ---- blocked right-looking Cholesky (round 4), sixteen columns per panel, all 256 threads; the right-hand side is ROW S
    // of the matrix, so its forward substitution is what every other row undergoes. Per panel:
    //   thread i = row i holds the panel's sixteen entries of its row in registers; the panel is factored four columns at a
    //   time: the 4 x 4 diagonal block is published (sixteen LDS words), factored in closed form by every thread, every row
    //   below solves its four entries against it and takes the rank-4 update of its remaining panel entries with the
    //   multipliers the block's rows publish -- two barriers per FOUR columns, no dot product over finished columns;
    //   the panel goes back to the matrix and into an LDS tile [rows][17], and the trailing matrix takes its rank-16 update
    //   on the matrix pipe: 16 x 16 tiles dealt to the four waves, four at a time (every load of the four -- operands from
    //   the LDS panel, the elements themselves from LDS / L2 -- is issued before the first product).
    // Round 3's left-looking form, one row per thread and a dot product over all finished columns per entry, two barriers
    // per COLUMN: S = 234, 971 us per launch (profiles/r03/rig_big.jsonl).
    const int i = tid;
    if (tid < S) A.at(S, tid) = s_b[tid];
    __syncthreads();
    double* Pn = s_pan;        // [256][17] the panel, rows by matrix row
    double* s_d = s_pan + 256 * 17;   // [16] diagonal block of a micro-block, [16 + 12 * 4] multipliers of the panel's later rows
    bool ok = true;
    for (int j0 = 0; j0 < S; j0 += 16) {
      const int nc = S - j0 < 16 ? S - j0 : 16;
      const bool row_in = i >= j0 && i <= S;
      double pv[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const int col = j0 + (c < nc ? c : 0);
        const double x = A.at(row_in ? i : S, col <= (row_in ? i : S) ? col : 0);
        pv[c] = (row_in && c < nc && (j0 + c <= i || i == S)) ? x : 0.0;
      }
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) {
        const int c0 = 4 * mb;
        if (c0 < nc) {   // (uniform)
          const int nb = nc - c0 < 4 ? nc - c0 : 4;
          const int rb = i - (j0 + c0);   // row inside the micro-block: 0 .. nb - 1
          if (rb >= 0 && rb < nb) {
#pragma unroll
            for (int c = 0; c < 4; ++c) s_d[rb * 4 + c] = pv[c0 + c];
          }
          __syncthreads();
          double a00 = s_d[0], a10 = s_d[4], a11 = s_d[5], a20 = s_d[8], a21 = s_d[9], a22 = s_d[10], a30 = s_d[12], a31 = s_d[13], a32 = s_d[14], a33 = s_d[15];
          if (nb < 2) { a10 = 0.0; a11 = 1.0; }
          if (nb < 3) { a20 = 0.0; a21 = 0.0; a22 = 1.0; }
          if (nb < 4) { a30 = 0.0; a31 = 0.0; a32 = 0.0; a33 = 1.0; }
          const double i0 = rsqrt_pos(a00);
          ok = ok && (a00 > 0.0) && isfinite(a00);
          const double l10 = a10 * i0, l20 = a20 * i0, l30 = a30 * i0;
          const double d1 = fma(-l10, l10, a11);
          const double i1 = rsqrt_pos(d1);
          ok = ok && (d1 > 0.0) && isfinite(d1);
          const double l21 = fma(-l20, l10, a21) * i1, l31 = fma(-l30, l10, a31) * i1;
          const double d2 = fma(-l21, l21, fma(-l20, l20, a22));
          const double i2 = rsqrt_pos(d2);
          ok = ok && (d2 > 0.0) && isfinite(d2);
          const double l32 = fma(-l31, l21, fma(-l30, l20, a32)) * i2;
          const double d3 = fma(-l32, l32, fma(-l31, l31, fma(-l30, l30, a33)));
          const double i3 = rsqrt_pos(d3);
          ok = ok && (d3 > 0.0) && isfinite(d3);
          if (i == 0) { s_inv[j0 + c0] = i0; if (nb > 1) s_inv[j0 + c0 + 1] = i1; if (nb > 2) s_inv[j0 + c0 + 2] = i2; if (nb > 3) s_inv[j0 + c0 + 3] = i3; }
          // the row's four entries against the block (rows inside the block get their own row of L: the same recurrence
          // stopped at the diagonal)
          {
            const double x0 = pv[c0], x1 = nb > 1 ? pv[c0 + 1] : 0.0, x2 = nb > 2 ? pv[c0 + 2] : 0.0, x3 = nb > 3 ? pv[c0 + 3] : 0.0;
            const double y0 = x0 * i0;
            const double y1 = fma(-y0, l10, x1) * i1;
            const double y2 = fma(-y1, l21, fma(-y0, l20, x2)) * i2;
            const double y3 = fma(-y2, l32, fma(-y1, l31, fma(-y0, l30, x3))) * i3;
            const bool below = rb >= nb || (i == S);   // (row S: the right-hand side, below everything)
            // inside the block: row rb of L = entries up to the diagonal (y_c for c < rb is L[rb][c]; the diagonal is d * inv)
            pv[c0] = below ? y0 : (rb == 0 ? a00 * i0 : (rb > 0 ? y0 : pv[c0]));
            if (nb > 1) pv[c0 + 1] = below ? y1 : (rb == 1 ? d1 * i1 : (rb > 1 ? y1 : pv[c0 + 1]));
            if (nb > 2) pv[c0 + 2] = below ? y2 : (rb == 2 ? d2 * i2 : (rb > 2 ? y2 : pv[c0 + 2]));
            if (nb > 3) pv[c0 + 3] = below ? y3 : (rb == 3 ? d3 * i3 : pv[c0 + 3]);
          }
          // multipliers of the panel's later columns: rows j0 + c2 (c2 >= c0 + 4) publish their four new entries
          const int rl = i - j0;   // row inside the panel
          if (rl >= c0 + 4 && rl < 16 && rl < nc) {
#pragma unroll
            for (int c = 0; c < 4; ++c) s_d[16 + (rl - 4) * 4 + c] = pv[c0 + c];
          }
          __syncthreads();
#pragma unroll
          for (int c2 = c0 + 4; c2 < 16; ++c2) {
            if (c2 < nc) {   // (uniform)
              double acc = pv[c2];
#pragma unroll
              for (int c = 0; c < 4; ++c) acc = fma(-(c < nb ? pv[c0 + c] : 0.0), s_d[16 + (c2 - 4) * 4 + c], acc);
              pv[c2] = (row_in && (j0 + c2 <= i || i == S)) ? acc : 0.0;
            }
          }
        }
      }
      // the panel: back to the matrix, and into its LDS tile for the trailing update
      if (row_in) {
#pragma unroll
        for (int c = 0; c < 16; ++c)
          if (c < nc && (j0 + c <= i || i == S)) A.at(i, j0 + c) = pv[c];
      }
#pragma unroll
      for (int c = 0; c < 16; ++c) Pn[i * 17 + c] = (row_in && c < nc && i >= j0 + nc) ? pv[c] : 0.0;   // (rows below the panel: the update's operands)
      __syncthreads();
      // ---- trailing update: rows t0..S (right-hand side included), columns t0..S-1, 16 x 16 tiles at multiples of 16
      const int t0 = j0 + nc;
      if (t0 < S) {
        const int wv = tid >> 6, ln = tid & 63, kq = ln >> 4, c16 = ln & 15;
        const int tlo = t0 >> 4, n16 = (S + 1 + 15) >> 4;
        // tiles (ti, tj), tlo <= tj <= ti < n16, numbered row by row; wave w takes numbers w, w + 4, ...: four per round
        const int nrow = n16 - tlo, ntile = nrow * (nrow + 1) / 2;
        for (int tb = wv; tb < ntile; tb += 16) {
          double am[4][4], bm[4][4], old[4][4];
          int at_i[4][4], at_k[4][4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int t = tb + 4 * u;
            const bool live = t < ntile;
            const int tc = live ? t : 0;
            int tr = (int)((sqrtf(8.0f * (float)tc + 1.0f) - 1.0f) * 0.5f);
            tr = tr * (tr + 1) / 2 > tc ? tr - 1 : tr;
            tr = (tr + 1) * (tr + 2) / 2 <= tc ? tr + 1 : tr;
            const int tq = tc - tr * (tr + 1) / 2;
            const int R = 16 * (tlo + tr), Cc = 16 * (tlo + tq);
            const int ra = R + c16 <= S ? R + c16 : S, rb2 = Cc + c16 < S ? Cc + c16 : S - 1;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
              const double xa = Pn[ra * 17 + 4 * ks + kq], xb = Pn[rb2 * 17 + 4 * ks + kq];
              am[u][ks] = (live && R + c16 <= S) ? xa : 0.0;
              bm[u][ks] = (live && Cc + c16 < S) ? xb : 0.0;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int row = R + kq + 4 * r, col = Cc + c16;
              const bool v = live && row >= t0 && row <= S && col >= t0 && col < S && (col <= row);
              at_i[u][r] = v ? row : -1;
              at_k[u][r] = v ? col : 0;
              old[u][r] = A.at(v ? row : S, v ? col : 0);
            }
          }
          d4 T[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) T[u] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int u = 0; u < 4; ++u) T[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(am[u][ks], bm[u][ks], T[u], 0, 0, 0);
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (at_i[u][r] >= 0) A.at(at_i[u][r], at_k[u][r]) = old[u][r] - T[u][r];
        }
      }
      __syncthreads();
    }
    if (!ok && tid == 0) s_cholok = 0;
    if (tid < S) s_b[tid] = A.at(S, tid);
    __syncthreads();
    // ---- backward substitution L^T x = y in blocks of sixteen unknowns, from the last: wave 0 solves the block's triangle
    // (lane j holds y_j and column j of the block; sixteen steps of lane read + FMA, no barrier), publishes x, and every
    // row above the block subtracts its sixteen products at once -- two barriers per SIXTEEN unknowns (round 3: one per
    // unknown, each behind a dependent load)
    {
      double* s_x = s_pan;   // [16] the block's solution
      for (int kb = ((S - 1) >> 4) << 4; kb >= 0; kb -= 16) {
        const int nbk = S - kb < 16 ? S - kb : 16;
        if (tid < 64) {
          const int j = lane < nbk ? lane : 0;
          double bj = s_b[kb + j];
          double lcol[16];
#pragma unroll
          for (int k = 0; k < 16; ++k) {
            const int kr = kb + (k < nbk ? k : 0);
            const double x = A.at(kr > kb + j ? kr : kb + j, kb + j);   // L[kb + k][kb + j] for k > j
            lcol[k] = (k < nbk && k > j) ? x : 0.0;
          }
#pragma unroll
          for (int k = 15; k >= 0; --k) {
            if (k < nbk) {   // (uniform)
              const double xk = readlane_d(bj, k) * s_inv[kb + k];
              bj = lane == k ? xk : fma(-lcol[k], xk, bj);
            }
          }
          if (lane < nbk) { s_b[kb + lane] = bj; s_x[lane] = bj; }
        }
        __syncthreads();
        if (i < kb) {
          double acc = s_b[i];
          double l[16];
#pragma unroll
          for (int k = 0; k < 16; ++k) l[k] = A.at(kb + (k < nbk ? k : 0), i);
#pragma unroll
          for (int k = 0; k < 16; ++k) acc = fma(-(k < nbk ? l[k] : 0.0), s_x[k], acc);
          s_b[i] = acc;
        }
        __syncthreads();
      }
    }
    if (tid < 64) {
      bool fin = true;
      for (int k = lane; k < S; k += 64) {
        fin = fin && isfinite(s_b[k]);
        P.ds[k] = -s_b[k];
      }
      const bool step_ok = s_cholok != 0 && __all(fin);
      if (lane == 0) s_stepok = step_ok ? 1 : 0;
    }
    __syncthreads();
  }
  double st2 = 0.0, xs2 = 0.0;
  const bool have_step = s_go != 0 && s_stepok != 0;
  if (have_step) rig_candidates(P, s_b, s_ss, true, cur, dst, st2, xs2);
  {
    const double a = wave_sum(st2), b2 = wave_sum(xs2);
    __syncthreads();
    if (lane == 0) { s8[tid >> 6] = a; s8[4 + (tid >> 6)] = b2; }
    __syncthreads();
  }
  if (tid == 0) {
    LmCtl c = s_c;
    if (s_go) {
      c.step_valid = have_step ? 1 : 0;
      c.cand_pending = 1;
      P.shared_stats[0] = (s8[0] + s8[1]) + (s8[2] + s8[3]);
      P.shared_stats[1] = (s8[4] + s8[5]) + (s8[6] + s8[7]);
    }
    *P.ctl = c;
    *P.ctl_next = c;
  }
}

// =============================================================================================
// THE RIG SOLVE AS ONE PERSISTENT KERNEL (round 3; poses only, single GPU, at most four frames per compute unit).
// TWO FORMS. This one, k_rig_persist, GLUES the three kernels' bodies together: AN EXPERIMENT, OFF BY DEFAULT (CC_RIG_PERSIST=1,
// used where the lean form below does not fit): correct -- the rig test suite passes on it -- and SLOWER than the three
// kernels it replaces: 81 against 47 us per iteration at BASELINE configs[3] size (profiles/r03/rig_persist_marks.jsonl).
// The bodies need up to 444 registers per thread, so a compute unit holds ONE wave per SIMD: the sweep of a frame's groups
// runs one after the other with every memory round trip exposed (25.6 us where the stand-alone sweep, sixteen waves deep,
// takes 9), and pose update, elimination and solve step each run 1.3 - 2 x slower for the same reason.
// The LEAN form further down (k_rig_persist_w + k_rig_persist_ctl: small rigs, the per-frame state in LDS, one wave per
// group, the workers under 128 / 256 registers) is what runs BY DEFAULT where it fits: 39 us at configs[3] size. Seams,
// control workgroup and host side are shared.
// Three launches per LM iteration cost this path ~19 of its ~47 us at BASELINE configs[3] (ramp of a launch, dependent
// read of the control block, the gap; profiles/r03/rig_c4_kernel_stats.csv): here ONE launch runs the whole solve, built
// from the very functions the three kernels run (rig_update_body, rig_sweep_adj_body, rig_elim_body, rig_solve_block,
// rig_candidates), with the seams of cc_intrinsics_persist.hip between them (cc_persist_dev.hpp: self-validating words,
// no atomics, no flags, bounded waits).
//   grid    : G = ceil(F / 4) worker workgroups + 1 control workgroup, 256 threads each, all resident (host: occupancy).
//   worker b: frames 4b .. 4b + 3, one wave each, for the whole solve. Round: [broadcast B: step + camera records] ->
//             pose update of its frames -> sweep of their groups (one wave: the groups of its frame one after the other)
//             -> statistics row -> [broadcast A: decision] -> elimination of its four frames -> partial row, compacted
//             to the K entries the reduced system uses -> posts it; then adds up ITS share of the K columns over all G
//             rows (column c belongs to worker c mod G: every worker reads G x K / G words -- the column sums of
//             k_rig_reduce, spread over the workers) and posts the sums.
//   control : owns the trust-region state. Gathers the statistics rows -> decision (first round: Jacobi scales of the
//             shared columns, |x|, lm_init -- what k_rig_init does) -> broadcast A; gathers the K column sums ->
//             reduced solve, candidates, records (rig_solve_block, unchanged) -> broadcast B.
// What a workgroup writes to global memory for its own later use (poses, frame records, group blocks, Y, partial row)
// it reads back itself: plain stores and loads on one compute unit. Sums over rows run in a fixed order.
// A wait that gives up sets the failure word (arrive[3]): nothing further happens, the host returns CC_ERR_COMM and the
// handle goes back to the three-kernel form.
// =============================================================================================
struct RigPersistDev {
  u64* sbox;            // [G][KS][2]  statistics rows: cost, model term, step^2, |x|^2, S diagonal sums (first round)
  u64* abox;            // [2 + S][2]  broadcast A: flags (1 done | cur << 3), radius, Jacobi scales of the shared columns (first round)
  u64* rbox;            // [G][K][2]   elimination rows (compacted)
  u64* cbox;            // [K][2]      column sums
  u64* pbox, *pcbox;    // the same two for the elimination on the ASSUMED decision (k_rig_persist_w; null: nobody assumes)
  u64* ybox;            // [NB][2]     broadcast B: flags (1 done | 2 step valid | cur << 3), radius, step[S], camera records [C][32]
  const int32_t* comp;  // [K] entry of the partial-row layout behind compact index k (the last two: failures, gradient maximum)
  const int32_t* slots; // [K] the same entries for k_rig_persist_w: 1 << 30 | p << 8 | q: sum_i Z[i][p] Z[i][q]; c << 16 | offset: entry of observed
                        //     camera c's block; -1: failed factorisations; -2: gradient maximum
  int32_t G, K, KS, NB;
  unsigned epoch0;      // tags: epoch0 + round + 1 (boxes are zeroed when they would wrap)
  unsigned* claim;      // [1] the control candidate that exchanges epoch0 + 1 in first is the control workgroup (k_rig_persist_ctl)
  unsigned long long* gate;   // pinned host word: the worker that finds all G workers started stores epoch0 + 1 into it and the HOST then
                              //   launches the control (rig_launch); null: no gate (the candidates run when they run)
  int32_t max_rounds, timeout_shift, first_shift;   // (first_shift: the workers' wait for the control's FIRST broadcast)
};

constexpr int kRigPersistMaxS = 48;     // shared coordinates (8 optimised cameras)
constexpr int kRigPersistMaxC = 9;      // cameras (records travel in broadcast B)
constexpr int kRigPersistMaxNB = 2 + kRigPersistMaxS + 32 * kRigPersistMaxC;   // (the control workgroup's own copies are sized for 48 coordinates; the workers take kRpwMaxS)

// One wave waits until the n doubles of a broadcast box carry `tag` and leaves them in dst[0..n) (LDS). false: gave up.
__device__ __forceinline__ bool rig_bcast_wait(const u64* box, unsigned tag, int n, double* dst, unsigned* fail, int tshift) {
  int lane = threadIdx.x & 63;
  asm volatile("" : "+v"(lane));   // (a fresh copy: the eleven word addresses of a lane are not worth keeping across a round)
  constexpr int W = (2 * kRigPersistMaxNB + 63) / 64;
  const long long t0 = wall_clock64();
  u64 v[W];
  for (unsigned spins = 0;; ++spins) {
    bool ok = true;
#pragma unroll
    for (int j = 0; j < W; ++j) {
      const int w = lane + 64 * j;
      if (64 * j < 2 * n) {   // (uniform)
        v[j] = ag_ld(box + (w < 2 * n ? w : 0));
        ok = ok && (w >= 2 * n || (unsigned)(v[j] >> 32) == tag);
      }
    }
    if (__all(ok)) break;
    if ((spins & 63u) == 63u && (timed_out(t0, tshift) || ag_ld32(fail) != 0u)) {
      if (lane == 0) __hip_atomic_store(fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return false;
    }
    __builtin_amdgcn_s_sleep(1);
  }
#pragma unroll
  for (int j = 0; j < W; ++j) {
    const int w = lane + 64 * j;
    if (w < 2 * n) reinterpret_cast<unsigned*>(dst)[w] = (unsigned)v[j];   // word 2i = low half of double i
  }
  return true;
}

// All 256 threads: thread t (< G) waits for entry `col` of row t (NC columns per row, up to NB columns in one round trip),
// then the block adds the G values in a fixed order (maximum for is_max). out[j] valid for every thread after return.
template <int NB>
__device__ __forceinline__ bool rig_gather_cols(const u64* box, int G, int rowlen, const int* cols, int ncols, int maxcol, unsigned tag, double* s4, double* out,
                                                unsigned* fail, int tshift) {
  const int tid = threadIdx.x;
  __shared__ int s_good;
  if (tid == 0) s_good = 1;
  __syncthreads();
  u64 lo[NB], hi[NB];
  bool good = true;
  if (tid < G) {
    const long long t0 = wall_clock64();
    for (unsigned spins = 0;; ++spins) {
      bool ok = true;
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int c = cols[j < ncols ? j : 0];
        const u64* p = box + ((size_t)tid * rowlen + c) * 2;
        lo[j] = ag_ld(p);
        hi[j] = ag_ld(p + 1);
        ok = ok && (j >= ncols || ((unsigned)(lo[j] >> 32) == tag && (unsigned)(hi[j] >> 32) == tag));
      }
      if (ok) break;
      if ((spins & 63u) == 63u && (timed_out(t0, tshift) || ag_ld32(fail) != 0u)) { good = false; break; }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  if (!good) s_good = 0;
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    if (j < ncols) {   // (uniform)
      const double v = (tid < G && good) ? ungranule(lo[j], hi[j]) : 0.0;
      double r;
      if (cols[j] == maxcol) {
        const double m = wave_max(v);
        __syncthreads();
        if ((tid & 63) == 0) s4[tid >> 6] = m;
        __syncthreads();
        r = fmax(fmax(s4[0], s4[1]), fmax(s4[2], s4[3]));
      } else {
        r = block_sum256(v, s4);
      }
      out[j] = r;
    }
  }
  __syncthreads();
  return s_good != 0;
}

#ifdef CC_RIG_PTIMING
#define RPW_MARK(i) do { if (round == 3 && blockIdx.x == 0 && threadIdx.x == 0) P.vec_stats[8 + (i)] = (double)wall_clock64(); } while (0)
#define RPC_MARK(i) do { if (round == 3 && threadIdx.x == 0) P.vec_stats[40 + (i)] = (double)wall_clock64(); } while (0)
#else
#define RPW_MARK(i) do { } while (0)
#define RPC_MARK(i) do { } while (0)
#endif
// The control workgroup of the persistent rig kernels (256 threads; dynamic LDS: the solve step's, rig_solve_block): block G of
// k_rig_persist, or a launch of its own next to the lean workers of k_rig_persist_w (k_rig_persist_ctl).
__device__ __forceinline__ void rig_persist_control(const RigDev& P, const RigPersistDev& Q, char* smem_raw) {
  __shared__ double s_bc[kRigPersistMaxNB];     // broadcast B of this round: flags, radius, step, camera records
  __shared__ double s_a[2 + kRigPersistMaxS];   // broadcast A
  __shared__ double s_ss[kRigPersistMaxS + 1];
  __shared__ double s4[4];
  __shared__ int s_cols[16];
  __shared__ int s_flag;
  const int tid = threadIdx.x;
  const int S = P.S, C = P.C, G = Q.G, K = Q.K, KS = Q.KS, NB = Q.NB;
  unsigned* fail = P.arrive + 3;
    // =========================================================================== control workgroup
    __shared__ LmCtl s_ctl;
    __shared__ cc_iteration s_rec;
    __shared__ double s_tot[4 + kRigPersistMaxS];
    __shared__ int s_has_rec, s_hit;
    double* smem = reinterpret_cast<double*>(smem_raw);
    double* vl = smem + (size_t)S * ((S + 1) | 1) + 5 * 128;   // [PC + 32] the reduced row in the layout rig_solve_block reads, behind its own LDS
    if (tid == 0) s_ctl = *P.ctl;   // (zeros: rig_begin)
    for (int i = tid; i < P.PC + 32; i += 256) vl[i] = 0.0;   // entries the compact rows never touch stay zero
    // the solve step's destination tables, copied to LDS once: rig_solve_block walks them every round, and here nothing
    // but this workgroup's latency is on the critical path (flat loads of LDS addresses through the same RigDev fields)
    RigDev Pc = P;
    {
      int32_t* t_tile = reinterpret_cast<int32_t*>(vl + P.PC + 32);
      int32_t* t_dd = t_tile + P.nT * 256;
      int32_t* t_dn = t_dd + P.ND;
      int16_t* t_sa = reinterpret_cast<int16_t*>(t_dn + P.ND);
      int16_t* t_sb = t_sa + P.ND;
      for (int i = tid; i < P.nT * 256; i += 256) t_tile[i] = P.tile_dst[i];
      for (int i = tid; i < P.ND; i += 256) { t_dd[i] = P.dir_dst[i]; t_dn[i] = P.dir_next[i]; t_sa[i] = P.dir_sa[i]; t_sb[i] = P.dir_sb[i]; }
      Pc.tile_dst = t_tile; Pc.dir_dst = t_dd; Pc.dir_next = t_dn; Pc.dir_sa = t_sa; Pc.dir_sb = t_sb;
    }
    __syncthreads();
    {   // records of the starting point (k_rig_records) -> broadcast B of round 0
      double a, b;
      rig_candidates(P, nullptr, nullptr, false, s_ctl.cur, s_ctl.cur, a, b);
    }
    __syncthreads();
    bool failed = false;
    for (int round = 0; round < Q.max_rounds; ++round) {
      const unsigned e = Q.epoch0 + (unsigned)round + 1u;
      const bool phase0 = round == 0;
      // ---- broadcast B(e): what this round's sweep evaluates
      RPC_MARK(0);
      if (tid == 0) {
        s_bc[0] = (double)((s_ctl.done ? 1 : 0) | ((phase0 || s_ctl.step_valid) ? 2 : 0) | ((s_ctl.cur & 1) << 3));
        s_bc[1] = s_ctl.radius;
      }
      if (!phase0 && tid < S) s_bc[2 + tid] = -smem[(size_t)S * ((S + 1) | 1) + tid];   // the shared step: -x of rig_solve_block (s_b)
      if (phase0 && tid < S) s_bc[2 + tid] = 0.0;
      for (int i = tid; i < 32 * C; i += 256) s_bc[2 + S + i] = P.camrec[i];
      __syncthreads();
      for (int w = tid; w < 2 * NB; w += 256) ag_st(Q.ybox + w, granule(e, s_bc[w >> 1], w & 1));
      if (s_ctl.done) break;
      // ---- statistics rows -> decision
      RPC_MARK(1);
      const bool swept = phase0 || s_ctl.step_valid;
      {
        const int nst = phase0 ? KS : 4;
        for (int c0 = 0; c0 < nst; c0 += 8) {
          if (tid < 8) s_cols[tid] = c0 + tid;
          __syncthreads();
          double out8[8];
          const int nc = nst - c0 < 8 ? nst - c0 : 8;
          if (!rig_gather_cols<8>(Q.sbox, G, KS, s_cols, nc, -1, e, s4, out8, fail, Q.timeout_shift)) failed = true;
          if (tid == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
              if (j < nc) s_tot[c0 + j] = out8[j];
          }
          __syncthreads();
        }
      }
      if (failed) break;
      // |x|^2 of the shared block at the starting point (k_rig_init)
      RPC_MARK(2);
      double x2_shared = 0.0;
      if (phase0) {
        double x2 = 0.0;
        const int cur0 = s_ctl.cur;
        for (int i = tid; i < C * 7; i += 256) {
          const int cc2 = i / 7;
          const double v = P.cam[((size_t)cur0 * C + cc2) * 8 + (i - cc2 * 7)];
          x2 += P.cam_fixed[cc2] ? 0.0 : v * v;
        }
        x2_shared = block_sum256(x2, s4);
      }
      if (tid == 0) {
        LmCtl c = s_ctl;
        const LmOpts o = *P.opts;
        const int prev_cur = c.cur & 1, was_valid = c.step_valid;
        const double prev_radius = c.radius;
        s_has_rec = 0;
        if (phase0) {
          for (int k = 0; k < S; ++k) s_ss[k] = o.jacobi_scaling ? 1.0 / (1.0 + sqrt(s_tot[4 + k])) : 1.0;
          lm_init(c, o, s_tot[0], sqrt(s_tot[3] + x2_shared));
        } else if (c.cand_pending) {
          double step2 = swept ? s_tot[2] : 0.0, xn2 = swept ? s_tot[3] : 0.0;
          if (c.step_valid) { step2 += P.shared_stats[0]; xn2 += P.shared_stats[1]; }
          const int len0 = c.log_len;
          lm_decide(c, o, &s_rec, swept ? s_tot[0] : 0.0, swept ? s_tot[1] : 0.0, step2, xn2);
          s_has_rec = (c.log_len != len0 && c.log_len <= P.log_cap) ? 1 : 0;
          if (s_has_rec) P.log[c.log_len - 1] = s_rec;
        }
        if (!c.done && round + 1 >= Q.max_rounds) { c.done = 1; c.term = CC_NO_CONVERGENCE; }
        s_ctl = c;
        // did the workers' assumption hold? (the expression they evaluate: persist_spec_radius)
        s_hit = Q.pbox != nullptr && !phase0 && swept && was_valid && !c.done && (c.cur & 1) == (prev_cur ^ 1) &&
                c.radius == persist_spec_radius(prev_radius, o.max_radius);
        s_a[0] = (double)((c.done ? 1 : 0) | (s_hit ? 4 : 0) | ((c.cur & 1) << 3));
        s_a[1] = c.radius;
      }
      __syncthreads();
      if (tid < S) { s_a[2 + tid] = phase0 ? s_ss[tid] : 0.0; if (phase0) P.ss[tid] = s_ss[tid]; }
      __syncthreads();
      for (int w = tid; w < 2 * (2 + S); w += 256) ag_st(Q.abox + w, granule(e, s_a[w >> 1], w & 1));
      RPC_MARK(3);
      if (s_ctl.done) {
        if (tid == 0) { *P.ctl = s_ctl; *P.ctl_next = s_ctl; }
        break;
      }
      // ---- the K column sums -> the layout rig_solve_block reads (P.vec), then the solve step
      if (tid == 0) s_flag = 0;
      __syncthreads();
      {
        const long long t0 = wall_clock64();
        bool good = true;
        for (int k0 = tid; k0 < K && good; k0 += 256 * 4) {
          u64 lo[4], hi[4];
          for (unsigned spins = 0;; ++spins) {
            bool ok = true;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int k = k0 + 256 * u;
              const u64* p = (s_hit ? Q.pcbox : Q.cbox) + (size_t)(k < K ? k : k0) * 2;
              lo[u] = ag_ld(p);
              hi[u] = ag_ld(p + 1);
              ok = ok && (unsigned)(lo[u] >> 32) == e && (unsigned)(hi[u] >> 32) == e;
            }
            if (ok) break;
            if ((spins & 63u) == 63u && (timed_out(t0, Q.timeout_shift) || ag_ld32(fail) != 0u)) { good = false; break; }
            __builtin_amdgcn_s_sleep(1);
          }
          if (!good) break;
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int k = k0 + 256 * u;
            if (k < K) {
              const double v = ungranule(lo[u], hi[u]);
              if (k == K - 1) vl[P.PC + P.rank] = v;   // the gradient maximum rides in the rank's slot (k_rig_reduce)
              else vl[Q.comp[k]] = v;
            }
          }
        }
        if (!good) s_flag = 1;
      }
      __syncthreads();
      if (s_flag == 1) { failed = true; break; }
      RPC_MARK(4);
      if (tid == 0) { *P.ctl = s_ctl; *P.ctl_next = s_ctl; }   // (rig_solve_block finishes the record of this round in P.log)
      __syncthreads();
      rig_solve_block<3>(Pc, smem, &s_ctl, vl);
      __syncthreads();
      RPC_MARK(5);
      if (tid == 0) s_ctl = *P.ctl;   // as the solve step left it (this workgroup wrote it)
      __syncthreads();
    }
    // ---- the solve is over
    __syncthreads();
    if (tid == 0) {
      LmCtl c = s_ctl;
      if (failed || ag_ld32(fail) != 0u) {
        __hip_atomic_store(fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        c.done = 1; c.term = CC_FAILURE_EXCHANGE;
      }
      if (!c.done) { c.done = 1; c.term = CC_NO_CONVERGENCE; }
      *P.ctl = c;
      *P.ctl_next = c;
      rig_publish(P, c);
    }
}

// ---------------------------------------------------------------------------------------------
// The LEAN workers of the persistent rig solve (k_rig_persist_w; small rigs: at most 4 observed cameras, 24 shared
// coordinates) with the control workgroup as a launch of its own beside them (k_rig_persist_ctl: rig_persist_control, on a
// second stream -- it needs 230 registers a thread, the workers must stay under 128 to put SIXTEEN waves on a compute
// unit). Same seams, same rows, same control as k_rig_persist; what differs is where a worker keeps its four frames:
// in LDS -- poses, frame records, the 16 x 16 blocks and compact records of their groups (both buffers), Y, the Jacobi
// scales -- and how it works on them: one WAVE PER GROUP in the sweep (sixteen at once; k_rig_persist: four, one after
// the other), sixteen lanes per frame in the pose update, one wave per frame in the elimination, which builds the
// compact row straight from a slot table (cc_intrinsics_persist.hip's way) instead of going through the partial-row layout.
// ---------------------------------------------------------------------------------------------
constexpr int kRpwMaxS = 24;        // shared coordinates: four optimised cameras (none of them frozen)
constexpr int kRpwMaxCO = 4;        // observed cameras = groups of a frame = sweep waves of a team
constexpr int kRpwYS = 28;          // row stride of Y / Z (S + 1 <= 25 columns)
constexpr int kRpwMaxK = 448;       // compact row entries (S = 24: 325 Schur + 108 direct + 2)
constexpr int kRpwInfoG = 32;       // s_info: [0..31] colinfo of the shared columns, [32..47] group of (team, slot)
// Per-team scratch (doubles). Every region is placed from the SIZE of the one before it, and the sizes from the capacities
// above: a capacity cannot be raised without the layout following (round 3: a slot sized by hand overflowed beyond four
// cameras in all and was found late -- c533722).
enum { RPW_Y = 0,                                  // [6][kRpwYS]   Y
       RPW_Z = RPW_Y + 6 * kRpwYS,                 // [6][kRpwYS]   Z
       RPW_POSE = RPW_Z + 6 * kRpwYS,              // [2][8]        the frame's pose, both buffers
       RPW_FREC = RPW_POSE + 2 * 8,                // [32]          frame record: R(9) t(3) step(6)
       RPW_SP = RPW_FREC + 32,                     // [8]           Jacobi scale of the pose block (6)
       RPW_A = RPW_SP + 8,                         // [32]          damped frame block (21) + its gradient (6)
       RPW_GST = RPW_A + 32,                       // [kRpwMaxCO][2] group statistics
       RPW_FST = RPW_GST + 2 * kRpwMaxCO,          // [2]           frame statistics
       RPW_HD0 = RPW_FST + 2,                      // [kRpwMaxCO][8] diagonals of the camera blocks (first round)
       RPW_ROW = ((RPW_HD0 + 8 * kRpwMaxCO + 7) / 8) * 8,   // [kRpwMaxK] the frame's compact row
       RPW_TEAM = RPW_ROW + kRpwMaxK };
// Workgroup scratch (doubles): broadcast B (flags, radius, step[S], records of ALL C cameras), broadcast A (flags, radius, S
// scales), Jacobi scales (S + 1), statistics row (4 + S), slot table int[kRpwMaxK], sums, s_info int[kRpwInfoG + 4 teams x
// kRpwMaxCO], column list int[8] + good flag.
constexpr int kRpwBcDoubles = ((2 + kRpwMaxS + 32 * kRigPersistMaxC + 7) / 8) * 8 + 24;   // (+ 24: round 3's slot was 344 for nine cameras; kept)
enum { RPW_BC = 0,
       RPW_AB = RPW_BC + kRpwBcDoubles,
       RPW_SS = RPW_AB + 32,
       RPW_SROW = RPW_SS + 32,
       RPW_SLOT = RPW_SROW + 32,                   // int[kRpwMaxK]
       RPW_S16 = RPW_SLOT + kRpwMaxK / 2,
       RPW_INFO = RPW_S16 + 16,                    // int[kRpwInfoG + 16]
       RPW_COLS = RPW_INFO + (kRpwInfoG + 4 * kRpwMaxCO) / 2,   // int[8], int good
       RPW_WG = RPW_COLS + 8 };
static_assert(kRpwYS >= kRpwMaxS + 1, "a row of Y / Z holds the S shared columns and the right-hand side");
static_assert(kRpwInfoG >= kRpwMaxS + 1, "s_info[0..kRpwInfoG) holds colinfo of the S + 1 columns");
static_assert(kRpwMaxK % 2 == 0 && kRpwMaxK >= (kRpwMaxS + 1) * (kRpwMaxS + 2) / 2 + kRpwMaxCO * kDE0 + 2, "compact row: Schur entries of S + 1 columns, 27 direct entries per observed camera, failures, gradient maximum");
static_assert(2 + kRpwMaxS + 32 * kRigPersistMaxC <= RPW_AB - RPW_BC, "broadcast B (step + records of every camera) must fit its LDS slot");
static_assert(2 + kRpwMaxS <= RPW_SS - RPW_AB && kRpwMaxS + 1 <= RPW_SROW - RPW_SS && 4 + kRpwMaxS <= RPW_SLOT - RPW_SROW, "broadcast A, scales and statistics row must fit their LDS slots");
static_assert(27 <= RPW_GST - RPW_A && 6 <= RPW_A - RPW_SP && 18 <= RPW_SP - RPW_FREC, "frame block + gradient, pose scale and frame record must fit their LDS slots");
static_assert(RPW_AB == 344 && RPW_SS == 376 && RPW_SROW == 408 && RPW_SLOT == 440 && RPW_TEAM == ((12 * kRpwYS + 16 + 32 + 8 + 32 + 8 + 2 + 32 + 7) / 8) * 8 + kRpwMaxK,
              "layout as measured in round 3 (profiles/r03/rig_persist_marks.jsonl); a change of capacity moves it knowingly");
constexpr int rpw_lds_doubles(int teams) { return teams * (2048 + 512 + 1024 + RPW_TEAM) + RPW_WG; }

// column sums over the G rows of a box, for a workgroup of NW waves (cf. rig_gather_cols; thread t < G polls row t)
template <int NB, int NW>
__device__ __forceinline__ bool rig_gather_cols_w(const u64* box, int G, int rowlen, const int* cols, int ncols, int maxcol, unsigned tag, double* s16, int* s_good,
                                                  double* out, unsigned* fail, int tshift) {
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  if (tid == 0) *s_good = 1;
  __syncthreads();
  u64 lo[NB], hi[NB];
  bool good = true;
  if (tid < G) {
    const long long t0 = wall_clock64();
    for (unsigned spins = 0;; ++spins) {
      bool ok = true;
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int c = cols[j < ncols ? j : 0];
        const u64* p = box + ((size_t)tid * rowlen + c) * 2;
        lo[j] = ag_ld(p);
        hi[j] = ag_ld(p + 1);
        ok = ok && (j >= ncols || ((unsigned)(lo[j] >> 32) == tag && (unsigned)(hi[j] >> 32) == tag));
      }
      if (ok) break;
      if ((spins & 63u) == 63u && (timed_out(t0, tshift) || ag_ld32(fail) != 0u)) { good = false; break; }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  if (!good) *s_good = 0;
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    if (j < ncols) {   // (uniform)
      const double v = (tid < G && good) ? ungranule(lo[j], hi[j]) : 0.0;
      const bool is_max = cols[j] == maxcol;
      const double w = is_max ? wave_max(v) : wave_sum(v);
      __syncthreads();
      if ((tid & 63) == 0) s16[tid >> 6] = w;
      __syncthreads();
      double r = s16[0];
#pragma unroll
      for (int u = 1; u < NW; ++u) r = is_max ? fmax(r, s16[u]) : r + s16[u];
      out[j] = r;
    }
  }
  __syncthreads();
  return *s_good != 0;
}

// The control workgroup FINDS its compute unit (round 4). Workgroups of a launch are dealt round-robin over the eight XCDs and
// never move; with G = 250 workers two XCDs hold 32 of them -- every compute unit -- and WHICH two is not fixed (the XCD
// block 0 of a launch goes to varies: MI355X guide, workgroup dispatch), so no block index can be told in advance to land
// next to a free compute unit (round 3 launched (G mod 8) + 1 blocks and let the last one work: right only when both
// launches start their round on the same XCD). This launch has kRigCtlCandidates blocks -- two per XCD -- and the FIRST one
// that gets to run claims the solve (one exchange on a word tagged with the solve's epoch) and is the control; the others
// leave as soon as they run (those queued on a full XCD: when the workers are gone). A claim needs a free compute unit
// somewhere, which G <= 255 leaves; the XCD that gave it is recorded (arrive[12]) for the host's diagnostics.
constexpr int kRigCtlCandidates = 16;
__global__ __launch_bounds__(256) void k_rig_persist_ctl(RigDev P, RigPersistDev Q) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  __shared__ int s_mine;
  if (threadIdx.x == 0) {
    const unsigned tag = Q.epoch0 + 1u;
    const unsigned prev = __hip_atomic_exchange(Q.claim, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_mine = prev != tag;
    if (prev != tag) {
      unsigned xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
      __hip_atomic_store(P.arrive + 12, ((unsigned)blockIdx.x << 8) | (xcc + 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __syncthreads();
  if (!s_mine) return;
  rig_persist_control(P, Q, smem_raw);
}

template <int TEAMS>
__global__ __launch_bounds__(TEAMS * 256) void k_rig_persist_w(RigDev P, RigPersistDev Q) {
  extern __shared__ __attribute__((aligned(16))) double rpw_lds[];
  constexpr int NT = TEAMS * 256;           // threads
  double* s_tile = rpw_lds;                 // [TEAMS][4 slots][2][256]
  double* s_comp = s_tile + TEAMS * 2048;   // [TEAMS][4][2][64]
  double* s_sw = s_comp + TEAMS * 512;      // [TEAMS * 4 waves][256] sweep scratch
  double* s_tm = s_sw + TEAMS * 1024;       // [TEAMS][RPW_TEAM]
  double* s_wg = s_tm + TEAMS * RPW_TEAM;   // [RPW_WG]
  const int tid0 = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid0 >> 6), team = wave >> 2, twave = wave & 3, lane = tid0 & 63;
  const int S = P.S, SW = P.SW, CO = P.CO, G = Q.G, K = Q.K, KS = Q.KS, NB = Q.NB;
  unsigned* fail = P.arrive + 3;
  const int64_t f = (int64_t)blockIdx.x * TEAMS + team;
  const bool has_frame = f < P.F;
  const int g_mine = __builtin_amdgcn_readfirstlane((has_frame && twave < CO) ? P.fslot[f * CO + twave] : -1);   // the group this wave sweeps
  double* tm = s_tm + team * RPW_TEAM;
  double* s_bc = s_wg + RPW_BC;
  double* s_a = s_wg + RPW_AB;
  double* s_ss = s_wg + RPW_SS;
  double* s_row = s_wg + RPW_SROW;
  int* s_slot = reinterpret_cast<int*>(s_wg + RPW_SLOT);
  double* s16 = s_wg + RPW_S16;
  int* s_info = reinterpret_cast<int*>(s_wg + RPW_INFO);   // [0..31] colinfo, [32..47] group of (team, slot)
  int* s_cols = reinterpret_cast<int*>(s_wg + RPW_COLS);
  int* s_good = s_cols + 8;
  // ---- every worker is RESIDENT once all G have passed this point: the last one tells the host, which launches the control
  // only then -- its candidates therefore only ever run on compute units the workers left free (k_rig_persist_ctl)
  if (Q.gate && tid0 == 0) {
    const unsigned prev = __hip_atomic_fetch_add(P.arrive + 13, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (prev + 1u == (unsigned)G) __hip_atomic_store(Q.gate, (unsigned long long)(Q.epoch0 + 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  // ---- start of the solve: poses of the workgroup's frames (buffer 0 holds the starting point: rig_begin), tables
  for (int i = tid0; i < TEAMS * (2048 + 512); i += NT) s_tile[i] = 0.0;
  for (int i = tid0; i < TEAMS * RPW_TEAM; i += NT) s_tm[i] = 0.0;
  for (int i = tid0; i < K; i += NT) s_slot[i] = Q.slots[i];
  if (tid0 < SW) s_info[tid0] = P.colinfo[tid0];
  if (tid0 >= 64 && tid0 < 64 + 4 * TEAMS) {
    const int t = (tid0 - 64) >> 2, j = (tid0 - 64) & 3;
    const int64_t ff = (int64_t)blockIdx.x * TEAMS + t;
    s_info[kRpwInfoG + t * 4 + j] = (ff < P.F && j < CO) ? P.fslot[ff * CO + j] : -1;
  }
  __syncthreads();
  if (has_frame && twave == 0 && lane < 8) {
    const double v = lane < 7 ? P.pose[(size_t)f * 8 + lane] : 0.0;
    tm[RPW_POSE + lane] = v;
    tm[RPW_POSE + 8 + lane] = v;
  }
  __syncthreads();
  int cur = 0;
  double radius = 1.0;
  for (int round = 0; round < Q.max_rounds; ++round) {
    const unsigned e = Q.epoch0 + (unsigned)round + 1u;
    const bool phase0 = round == 0;
    // ---- broadcast B: step and camera records
    RPW_MARK(0);
    if (wave == 0 && !rig_bcast_wait(Q.ybox, e, NB, s_bc, fail, phase0 ? Q.first_shift : Q.timeout_shift)) s_bc[0] = 1.0;
    __syncthreads();
    const int flb = (int)s_bc[0];
    RPW_MARK(1);
    if (flb & 1) { cur = (flb >> 3) & 1; break; }
    cur = (flb >> 3) & 1;
    const bool swept = (flb & 2) != 0;
    const int dst = phase0 ? cur : (cur ^ 1);
    if (swept) {
      // ---- pose update of the frame (rig_update_body's arithmetic): sixteen lanes of the team's first wave
      int ul = tid0 & 63;
      asm volatile("" : "+v"(ul));
      if (has_frame && twave == 0 && ul < 16) {
        const int lane = ul;
        double u[6] = {0, 0, 0, 0, 0, 0};
        if (!phase0) {
          for (int k = lane; k < SW; k += 16) {
            const double d = k < S ? s_bc[2 + k] : 1.0;
#pragma unroll
            for (int i = 0; i < 6; ++i) u[i] += tm[RPW_Y + i * kRpwYS + k] * d;
          }
#pragma unroll
          for (int i = 0; i < 6; ++i) u[i] = row16_sum(u[i]);
        }
        if (lane == 0) {
          bool active = false;
          for (int j = 0; j < 4; ++j) active = active || s_info[kRpwInfoG + team * 4 + j] >= 0;
          double q[4], t[3], dp[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
          for (int i = 0; i < 4; ++i) q[i] = tm[RPW_POSE + cur * 8 + i];
#pragma unroll
          for (int i = 0; i < 3; ++i) t[i] = tm[RPW_POSE + cur * 8 + 4 + i];
          double step2 = 0.0;
          if (!phase0) {
            if (active) {
#pragma unroll
              for (int i = 0; i < 6; ++i) dp[i] = -u[i] * tm[RPW_SP + i];
              double qn[4];
              quat_plus_tab(q, dp, qn);   // (series coefficients from a table: as literals they are hoisted out of the round loop and spilled)
#pragma unroll
              for (int i = 0; i < 4; ++i) { const double d = qn[i] - q[i]; step2 += d * d; q[i] = qn[i]; }
#pragma unroll
              for (int i = 0; i < 3; ++i) { const double tn = t[i] + dp[3 + i]; const double d = tn - t[i]; step2 += d * d; t[i] = tn; }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) tm[RPW_POSE + dst * 8 + i] = q[i];
#pragma unroll
            for (int i = 0; i < 3; ++i) tm[RPW_POSE + dst * 8 + 4 + i] = t[i];
          }
          double R[9];
          quat_to_R(q, R);
#pragma unroll
          for (int i = 0; i < 9; ++i) tm[RPW_FREC + i] = R[i];
#pragma unroll
          for (int i = 0; i < 3; ++i) tm[RPW_FREC + 9 + i] = t[i];
#pragma unroll
          for (int i = 0; i < 6; ++i) tm[RPW_FREC + 12 + i] = dp[i];
          tm[RPW_FST] = step2;
          tm[RPW_FST + 1] = active ? q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3] + t[0] * t[0] + t[1] * t[1] + t[2] * t[2] : 0.0;
        }
      }
      __syncthreads();
      // ---- sweep: one wave per group
      RPW_MARK(2);
      if (g_mine >= 0) {
        const RigSweepIO io{s_bc + 2 + S, tm + RPW_FREC, s_comp + ((team * 4 + twave) * 2 + cur) * 64, s_tile + ((team * 4 + twave) * 2 + dst) * 256,
                            s_comp + ((team * 4 + twave) * 2 + dst) * 64, tm + RPW_GST + 2 * twave, tm + RPW_HD0 + 8 * twave};
        rig_sweep_adj_body<1, true>(P, g_mine, phase0 ? 0 : 1, cur, s_sw + wave * 256, io);
      }
      __syncthreads();
      RPW_MARK(3);
    }
    // ---- statistics row of the workgroup (teams and slots in order)
    {
      int tid = tid0;
      asm volatile("" : "+v"(tid));
      if (tid == 0) {
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        if (swept) {
          for (int t = 0; t < TEAMS; ++t) {
            for (int j = 0; j < 4; ++j)
              if (s_info[kRpwInfoG + t * 4 + j] >= 0) { a0 += s_tm[t * RPW_TEAM + RPW_GST + 2 * j]; a1 += s_tm[t * RPW_TEAM + RPW_GST + 2 * j + 1]; }
            if ((int64_t)blockIdx.x * TEAMS + t < P.F) { a2 += s_tm[t * RPW_TEAM + RPW_FST]; a3 += s_tm[t * RPW_TEAM + RPW_FST + 1]; }
          }
        }
        s_row[0] = a0; s_row[1] = a1; s_row[2] = a2; s_row[3] = a3;
      }
      if (phase0 && tid >= 64 && tid < 64 + S) {   // diagonal of H_cc per shared column (Jacobi scaling)
        const int k = tid - 64, info = s_info[k], j = info >> 8, comp = info & 15;
        double d = 0.0;
        for (int t = 0; t < TEAMS; ++t)
          if (s_info[kRpwInfoG + t * 4 + j] >= 0) d += s_tm[t * RPW_TEAM + RPW_HD0 + 8 * j + comp];
        s_row[4 + k] = d;
      }
      __syncthreads();
      const int nst = phase0 ? KS : 4;
      if (tid < 2 * nst) ag_st(Q.sbox + ((size_t)blockIdx.x * KS) * 2 + tid, granule(e, s_row[tid >> 1], tid & 1));
    }
    RPW_MARK(4);
    // ---- the assumed decision (cf. cc_intrinsics_persist.hip): candidate accepted, radius at its clamp -- the normal outcome of a
    // step that works. The workers eliminate the candidate NOW, next to the control's gathering and deciding; when the
    // decision is what was assumed (broadcast A says so) the rows are already where the control looks for them.
    const bool spec = !phase0 && swept;
    const double radius_spec = persist_spec_radius(radius, P.opts->max_radius);
    auto eliminate_and_post = [&](const int cur_e, const double radius_e, const bool first_e, u64* rowbox, u64* colbox, const bool is_spec) {
    // ---- elimination of the frame: the team's first wave
    if (twave == 0) {
      int ln = lane;
      asm volatile("" : "+v"(ln));
      double* rowt = tm + RPW_ROW;
      bool exists[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) exists[j] = has_frame && s_info[kRpwInfoG + team * 4 + j] >= 0;
      const bool live = exists[0] || exists[1] || exists[2] || exists[3];
      const double* T0 = s_tile + ((team * 4) * 2 + cur_e) * 256;   // slot j: T0 + j * 512
      bool ok = true;
      double gmaxp = 0.0;
      if (live) {
        // frame block A = sum over the groups: lanes 0..26 (21 entries of H_ff, 6 of g_f)
        if (ln < 27) {
          int a_off;
          if (ln < 21) { int i = 0; while (tri(i + 1, 0) <= ln) ++i; a_off = (6 + i) * 16 + 6 + (ln - tri(i, 0)); }
          else a_off = (6 + (ln - 21)) * 16 + 12;
          double a_e = 0.0;
#pragma unroll
          for (int j = 0; j < 4; ++j) a_e += exists[j] ? T0[j * 512 + a_off] : 0.0;
          tm[RPW_A + ln] = a_e;
        }
        wave_lds_fence();
        const bool jac = P.opts->jacobi_scaling != 0;
        const double mn = P.opts->min_lm_diagonal, mx = P.opts->max_lm_diagonal;
        const double inv_radius = 1.0 / radius_e;
        double sf[6], L[21], Li[6], gf[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) gf[i] = tm[RPW_A + 21 + i];
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
          for (int j = 0; j <= i; ++j) L[tri(i, j)] = tm[RPW_A + tri(i, j)];
        if (first_e) {
#pragma unroll
          for (int i = 0; i < 6; ++i) sf[i] = jac ? 1.0 / (1.0 + sqrt(L[tri(i, i)])) : 1.0;
          if (ln < 6) {
            double sl = 0.0;
#pragma unroll
            for (int i = 0; i < 6; ++i) sl = ln == i ? sf[i] : sl;
            tm[RPW_SP + ln] = sl;
          }
        } else {
#pragma unroll
          for (int i = 0; i < 6; ++i) sf[i] = tm[RPW_SP + i];
        }
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
          for (int j = 0; j <= i; ++j) L[tri(i, j)] = sf[i] * L[tri(i, j)] * sf[j];
#pragma unroll
        for (int i = 0; i < 6; ++i) L[tri(i, i)] += clampd(L[tri(i, i)], mn, mx) * inv_radius;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          double d = L[tri(j, j)];
#pragma unroll
          for (int k = 0; k < j; ++k) d -= L[tri(j, k)] * L[tri(j, k)];
          ok = ok && (d > 0.0) && isfinite(d);
          const double inv = rsqrt_pos(d);
          L[tri(j, j)] = d * inv;
          Li[j] = inv;
#pragma unroll
          for (int i = j + 1; i < 6; ++i) {
            double a = L[tri(i, j)];
#pragma unroll
            for (int k = 0; k < j; ++k) a -= L[tri(i, k)] * L[tri(j, k)];
            L[tri(i, j)] = a * inv;
          }
        }
        {   // the frame's share of Ceres' gradient_max_norm (pose_grad_proj_max, cc_common.hpp)
          double q4[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) q4[i] = tm[RPW_POSE + cur_e * 8 + i];
          gmaxp = pose_grad_proj_max_tab(q4, gf);
        }
        // the factor is the same in every lane: scalar registers from here on
#pragma unroll
        for (int i = 0; i < 21; ++i) L[i] = rfl(L[i]);
#pragma unroll
        for (int i = 0; i < 6; ++i) { Li[i] = rfl(Li[i]); sf[i] = rfl(sf[i]); }
        if (ln < SW) {   // shared column ln (ln == S: the right-hand side)
          const int info = s_info[ln], kind = (info >> 4) & 15, j = info >> 8, comp = info & 15;
          const double sc = ln < S ? s_ss[ln] : 1.0;
          const double* Tj = T0 + (kind == 0 ? j : 0) * 512;
          const bool ex = kind == 0 && ((j == 0 && exists[0]) || (j == 1 && exists[1]) || (j == 2 && exists[2]) || (j == 3 && exists[3]));
          double z[6], y[6];
#pragma unroll
          for (int i = 0; i < 6; ++i) {
            const double w = Tj[comp * 16 + 6 + i];
            double a = kind == 3 ? sf[i] * gf[i] : (ex ? sf[i] * w * sc : 0.0);
#pragma unroll
            for (int k = 0; k < i; ++k) a -= L[tri(i, k)] * z[k];
            z[i] = a * Li[i];
          }
#pragma unroll
          for (int i = 5; i >= 0; --i) {
            double a = z[i];
#pragma unroll
            for (int k = i + 1; k < 6; ++k) a -= L[tri(k, i)] * y[k];
            y[i] = a * Li[i];
          }
#pragma unroll
          for (int i = 0; i < 6; ++i) { tm[RPW_Z + i * kRpwYS + ln] = z[i]; tm[RPW_Y + i * kRpwYS + ln] = y[i]; }
        }
        wave_lds_fence();
      }
      // the frame's compact row, slot k on lane k mod 64
      for (int k = ln; k < K; k += 64) {
        const int code = s_slot[k];
        double v = 0.0;
        if (live) {
          if (code == -1) v = ok ? 0.0 : 1.0;
          else if (code == -2) v = gmaxp;
          else if (code & (1 << 30)) {
            const int pcol = (code >> 8) & 255, qcol = code & 255;
#pragma unroll
            for (int i = 0; i < 6; ++i) v += tm[RPW_Z + i * kRpwYS + pcol] * tm[RPW_Z + i * kRpwYS + qcol];
          } else {
            const int j = code >> 16;
            const bool ex = (j == 0 && exists[0]) || (j == 1 && exists[1]) || (j == 2 && exists[2]) || (j == 3 && exists[3]);
            v = ex ? T0[j * 512 + (code & 0xffff)] : 0.0;
          }
        }
        rowt[k] = v;
      }
    }
    __syncthreads();
    // ---- the workgroup's row (teams in order) -> granules
    if (!is_spec) RPW_MARK(6); else RPW_MARK(9);
    {
      int tid = tid0;
      asm volatile("" : "+v"(tid));
      for (int k = tid; k < K; k += NT) {
        double v = s_tm[RPW_ROW + k];
        for (int t = 1; t < TEAMS; ++t) {
          const double w = s_tm[t * RPW_TEAM + RPW_ROW + k];
          v = k == K - 1 ? fmax(v, w) : v + w;
        }
        u64* q = rowbox + ((size_t)blockIdx.x * K + k) * 2;
        ag_st(q, granule(e, v, 0));
        ag_st(q + 1, granule(e, v, 1));
      }
    }
    if (!is_spec) RPW_MARK(7); else RPW_MARK(10);
    // ---- this workgroup's share of the column sums: columns b, b + G, ...
    for (int c0 = (int)blockIdx.x; c0 < K; c0 += 8 * G) {
      int tidc = tid0;
      asm volatile("" : "+v"(tidc));
      if (tidc < 8) s_cols[tidc] = c0 + tidc * G < K ? c0 + tidc * G : c0;
      __syncthreads();
      int nc = 0;
      for (int j = 0; j < 8; ++j) nc += c0 + j * G < K ? 1 : 0;
      double out8[8];
      const bool okg = rig_gather_cols_w<8, TEAMS * 4>(rowbox, G, K, s_cols, nc, K - 1, e, s16, s_good, out8, fail, Q.timeout_shift);
      if (!okg && tidc == 0) __hip_atomic_store(fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (okg && tidc < 2 * nc) {
        const int j = tidc >> 1;
        double v = 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u) v = j == u ? out8[u] : v;
        ag_st(colbox + (size_t)s_cols[j] * 2 + (tidc & 1), granule(e, v, tidc & 1));
      }
      __syncthreads();
    }
    };
    if (spec) eliminate_and_post(dst, radius_spec, false, Q.pbox, Q.pcbox, true);
    // ---- broadcast A: the decision
    if (wave == 0 && !rig_bcast_wait(Q.abox, e, 2 + S, s_a, fail, Q.timeout_shift)) s_a[0] = 1.0;
    __syncthreads();
    const int fla = (int)s_a[0];
    RPW_MARK(5);
    if (fla & 1) { cur = (fla >> 3) & 1; break; }
    if (fla & 4) {   // the assumption held
      cur = dst;
      radius = radius_spec;
    } else {
      cur = (fla >> 3) & 1;
      radius = s_a[1];
      {
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        if (phase0 && tid < S) s_ss[tid] = s_a[2 + tid];
      }
      __syncthreads();
      eliminate_and_post(cur, radius, phase0, Q.rbox, Q.cbox, false);
    }
    RPW_MARK(8);
  }
  // ---- the solve is over: the frame's accepted pose goes back to global memory (cc_rig_get_state, the next solve)
  __syncthreads();
  if (ag_ld32(fail) == 0u && has_frame && twave == 0 && lane < 7) P.pose[((size_t)cur * P.F + f) * 8 + lane] = tm[RPW_POSE + cur * 8 + lane];
}

#ifdef CC_RIG_TIMING
// Timing-only: the factorisation routines alone, hot, on one workgroup (scripts/time_chol.py): `reps` factorisations of the same
// S x S matrix (+ right-hand side) from global memory; out[0] = 100 MHz ticks per factorisation, out[1] = shader cycles.
__global__ __launch_bounds__(256) void k_chol_bench(const double* Ain, int S, int reps, int which, double* out) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* A = reinterpret_cast<double*>(smem_raw);
  const int LD = (S + 1) | 1, tid = threadIdx.x, lane = tid & 63;
  double* s_b = A + (size_t)S * LD;
  double* s_inv = s_b + 128;
  long long ticks = 0, cyc = 0;
  double chk = 0.0;
  for (int r = 0; r < reps; ++r) {
    for (int i = tid; i < (S + 1) * LD; i += 256) A[i] = Ain[i];
    __syncthreads();
    const long long t0 = wall_clock64(), c0 = clock64();
    if (which == 0 || which >= 2) {
      chol_block4(A, S, LD, s_inv, nullptr, which == 2 ? 1 : 0);
    } else {
      double b0 = 0.0, b1 = 0.0, v0 = 0.0, v1 = 0.0;
      bool okw = true;
      if (tid < 64) b0 = lane < S ? s_b[lane] : 0.0;
      for (int j0 = 0; j0 < S; j0 += 8) {
        const int nc = S - j0 < 8 ? S - j0 : 8;
        if (tid < 64) chol_panel<false>(A, S, LD, j0, nc, s_inv, b0, b1, v0, v1, okw);
        __syncthreads();
        const int t0c = j0 + nc;
        if (t0c < S) chol_trail_mfma<2>(A, S, LD, j0, nc, t0c, S);
        __syncthreads();
      }
      if (tid < 64 && lane < S) s_b[lane] = b0;
    }
    __syncthreads();
    ticks += wall_clock64() - t0; cyc += clock64() - c0;
    chk += s_b[S - 1];
  }
  if (tid == 0) { out[0] = (double)ticks / reps; out[1] = (double)cyc / reps; out[2] = chk; }
}
#endif

// creation: world point of every observation
__global__ void k_rig_expand_xyz(int64_t n, const int32_t* widx, const float* wxyz, float* oxyz) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t w = widx[i];
  oxyz[i * 3] = wxyz[w * 3]; oxyz[i * 3 + 1] = wxyz[w * 3 + 1]; oxyz[i * 3 + 2] = wxyz[w * 3 + 2];
}

// per-observation robustified cost at the accepted point (extrinsics_calibrator.cpp:219-225)
__global__ void k_rig_obs_cost(RigDev P, int cur, double* out /*sorted order*/) {
  const int64_t g = blockIdx.x;
  const int f = P.gframe[g], c = P.gcam[g];
  const double* pc = P.cam + ((size_t)cur * P.C + c) * 8;
  const double* pf = P.pose + ((size_t)cur * P.F + f) * 8;
  double Rc[9], Rf[9];
  quat_to_R(pc, Rc);
  quat_to_R(pf, Rf);
  const float2* uv2 = reinterpret_cast<const float2*>(P.uv);
  const double* kk = P.kmode ? P.intr + ((size_t)cur * P.CK + P.kset[c]) * 16 : nullptr;
  for (int64_t idx = P.goff[g] + threadIdx.x; idx < P.goff[g + 1]; idx += blockDim.x) {
    const float2 m = uv2[idx];
    const int64_t w = P.widx[idx];
    RigObs o;
    rig_common(Rf, pf + 4, Rc, pc + 4, (double)P.wxyz[w * 3], (double)P.wxyz[w * 3 + 1], (double)P.wxyz[w * 3 + 2],
               (double)m.x, (double)m.y, o);
    double ru = o.ru, rv = o.rv;
    if (kk) {
      RigKObs ko;
      rigk_obs(kk, o, (double)m.x, (double)m.y, ko);
      ru = ko.ru; rv = ko.rv;
    }
    double rho, sr;
    huber(P.huber_a, ru * ru + rv * rv, rho, sr);
    out[idx] = 0.5 * rho;
  }
}

}  // namespace cc

// =============================================================================================
// host side
// =============================================================================================
namespace cc {
struct Comm;
int comm_create(const uint8_t id[128], int rank, int nranks, Comm** out);
void comm_destroy(Comm* c);
int comm_allreduce_sum(Comm* c, double* buf, int n, hipStream_t stream);
}  // namespace cc

// dynamic LDS of the persistent kernels' control workgroup: the solve step's, the reduced row, the destination tables
static size_t rig_persist_ctl_lds(size_t solve_lds, const cc::RigDev& d) {
  return solve_lds + (size_t)(d.PC + 32) * 8 + (((size_t)4 * (d.nT * 256 + 2 * d.ND) + (size_t)4 * d.ND + 7) & ~(size_t)7);
}

struct cc_rig {
  int device = 0;
  hipStream_t stream = nullptr;
  cc::RigDev d{};
  int64_t C = 0, F = 0, N = 0, NG = 0, P = 0;
  int n_runs = 0;            // runs of shared columns (one per optimised camera, one per intrinsics set): blocks of k_rig_init
  int n_shared_runs = 0;     // of which sets of intrinsics common to all cameras (summed by init_slices blocks each)
  int sweep_waves = 4;       // waves per workgroup of the poses-only sweep (2: small groups that outnumber the slots)
  bool frame_allowed = true; // poses-only, three-kernel path: the FRAME form of the sweep (k_rig_sweep_frame); CC_RIG_SWEEP_FRAME=0: one workgroup per group (k_rig_sweep_adj)
  int frame_waves = 2;       // waves per frame workgroup of the frame form
  int kmode = 0;
  std::vector<int64_t> perm;  // sorted position -> caller's observation index
  bool perm_inverse = false;   // perm[k] = regrouped position of the caller's observation k (records) instead of perm[i] = caller's index of position i
  std::vector<std::pair<void*, size_t>> allocs;   // the chunks dev_alloc carves buffers from (cc::pool_alloc: recycled between handles)
  // pinned staging of the small host-built tables while the handle is being created (rig_create_impl): dev_upload copies a table
  // in and enqueues an asynchronous copy on the handle's stream instead of one synchronous hipMemcpy per table (~25 of them)
  char* up_stage = nullptr;
  size_t up_cap = 0, up_used = 0;
  char* chunk_cur = nullptr;
  size_t chunk_left = 0;
  double* init_cam = nullptr;
  double* init_pose = nullptr;
  double* d_cost = nullptr;
  bool have_state = false;
  cc::LmCtl* h_ctl = nullptr;
  void* pinned = nullptr;       // the pinned block h_ctl and host_pub live in
  hipGraphExec_t graph[3] = {nullptr, nullptr, nullptr};   // first chunk (with the preparation) | chunk of check_interval rounds | of twice as many
  int graph_iters = 0;
  cc::Comm* comm = nullptr;
  cc::Mailbox mailbox;          // mailbox exchange (cc_rig_exchange_export / _attach)
  double* init_intr = nullptr;  // [max(CK,1)][16] (extension)
  uint32_t* d_kmask = nullptr;  // same memory as d.kmask
  bool have_intr = false;
  bool exchange = false;
  uint8_t* d_cam_fixed = nullptr;          // same memory as d.cam_fixed
  std::vector<uint8_t> frozen, seen;       // host copies (user freeze flags, locally observed cameras)
  std::vector<uint8_t> seen_any;           // cameras observed by any rank (what the column layout is built for)
  std::vector<int32_t> gframe_h, gcam_h;   // host copies of the group tables (layout rebuilds)
  std::vector<int64_t> fgoff_h;
  std::vector<int64_t> goff_h;             // [NG + 1] observation range of each group
  size_t elim_lds = 0, solve_lds = 0;
  bool big = false;             // 128 <= S <= 255: the plain kernels (k_rig_elim_big, k_rig_solve_big), no exchange
  bool persist_lean_allowed = true;   // CC_RIG_PERSIST=0: three kernels per iteration always
  double* d_cam_backup = nullptr;   // [C][8] cameras of the starting point (the lean persistent solve may be run again in the three-kernel form)
  int p_teams = 4;              // frames per workgroup of the lean form
  bool persist_w_ok = false;    // ... and so can its lean form (k_rig_persist_w + k_rig_persist_ctl: <= 4 observed cameras, <= 24 shared coordinates)
  int form_reruns = 0;             // lean persistent solves that gave up and were run again in the three-kernel form
  int lean_strikes = 0;            // ... of them in the first round, in a row (two demote the handle)
  std::string form_note;           // why (cc_rig_solver_status)
  bool gated = false;              // the last lean solve launched its control behind the workers' residency word
  hipStream_t stream2 = nullptr;   // the control workgroup's launch of the lean form
  hipEvent_t ev_begin = nullptr;
  cc::RigPersistDev pq{};
  unsigned p_epoch = 0;         // tags handed out so far
  size_t p_box_words = 0;       // seam boxes: one allocation of this many 8-byte words (re-zeroed before the tags wrap)
  bool big_packed = false;      // ... with the reduced system as a packed triangle in LDS (else in bigA, global memory)
  double* bigA = nullptr;
  volatile unsigned long long* host_pub = nullptr;   // = h_ctl's pinned block: [0] sequence word, [2..19] control block, [20] failure word
  unsigned long long pub_count = 0;                  // chunks published so far
  cc::LmCtl last_st{};          // control block as the last solve / reset left it (no read-back at the start of a solve)
  bool st_known = false;
  cc::LmOpts cached_opts{};     // what the device holds
  bool opts_valid = false;
  int co_resident = 1;         // shards / processes whose k_rig_reduce launches share this device (cc_rig_optimize_multi counts them,
                               // cc_rig_exchange_attach derives ceil(ranks / visible devices); CC_RIG_CO_RESIDENT overrides)
  int reduce_blocks = 0;       // grid of the fused reduce + solve + update launch (rig_size_reduce_grid)
  size_t reduce_key = ~(size_t)0;
  std::vector<hipEvent_t> events;
  int k2_grid = 1024;          // workgroups of k_rig_sweep_k2: four per compute unit (rig_layout)
  std::vector<int> event_kind;
  std::vector<int> event_round;   // round of the solve a probed launch belongs to (summarise_probes)
  int enq_round = 0;
};

namespace cc {

// Device buffers of a handle are carved out of a few large chunks (bump allocation, 256-byte aligned):
// a one-shot caller pays for a handful of hipMalloc / hipFree calls instead of forty.
template <class T>
static int dev_alloc(cc_rig* h, T** p, size_t n) {
  const size_t bytes = (std::max<size_t>(n, 1) * sizeof(T) + 255) & ~(size_t)255;
  if (h->chunk_left < bytes) {
    const size_t sz = std::max<size_t>(bytes, (size_t)16 << 20);
    void* c = nullptr;
    size_t got = 0;
    if (int rc = pool_alloc(h->device, sz, &c, &got)) return rc;
    h->allocs.emplace_back(c, got);
    h->chunk_cur = static_cast<char*>(c);
    h->chunk_left = got;
  }
  *p = reinterpret_cast<T*>(h->chunk_cur);
  h->chunk_cur += bytes;
  h->chunk_left -= bytes;
  return 0;
}
template <class T>
static int dev_zeroed(cc_rig* h, T** p, size_t n) {
  if (int rc = dev_alloc(h, p, n)) return rc;
  // (on the handle's stream: everything that reads the buffer runs there, and the creating call synchronises it before it returns)
  if (h->stream) CC_HIP(hipMemsetAsync(*p, 0, std::max<size_t>(n, 1) * sizeof(T), h->stream));
  else CC_HIP(hipMemset(*p, 0, std::max<size_t>(n, 1) * sizeof(T)));
  return 0;
}
template <class T>
static int dev_upload(cc_rig* h, const T** p, const std::vector<T>& v) {
  T* q = nullptr;
  if (int rc = dev_alloc(h, &q, v.size())) return rc;
  const size_t bytes = v.size() * sizeof(T), padded = (bytes + 63) & ~(size_t)63;
  if (bytes && h->up_stage && h->stream && h->up_used + padded <= h->up_cap) {
    std::memcpy(h->up_stage + h->up_used, v.data(), bytes);
    CC_HIP(hipMemcpyAsync(q, h->up_stage + h->up_used, bytes, hipMemcpyHostToDevice, h->stream));
    h->up_used += padded;
  } else if (bytes) {
    CC_HIP(hipMemcpy(q, v.data(), bytes, hipMemcpyHostToDevice));
  }
  *p = q;
  return 0;
}

static void rig_drop_graphs(cc_rig* h) {
  for (auto& g : h->graph)
    if (g) { hipGraphExecDestroy(g); g = nullptr; }
}

struct RigProbe {  // optional hipEvent bracket around one launch
  cc_rig* h; int kind; bool on; int round_shift; hipEvent_t e0 = nullptr, e1 = nullptr;
  RigProbe(cc_rig* h_, int kind_, bool on_, int round_shift_ = 0) : h(h_), kind(kind_), on(on_), round_shift(round_shift_) {
    if (on) { hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0, h->stream); }
  }
  ~RigProbe() {
    if (on) { hipEventRecord(e1, h->stream); h->events.push_back(e0); h->events.push_back(e1); h->event_kind.push_back(kind); h->event_round.push_back(h->enq_round + round_shift); }
  }
};

// hipFuncAttributeMaxDynamicSharedMemorySize, set once per (device, kernel) and size: a handle's layout asks for ~20 of them, a
// fresh handle per call (the reference's workflow) would pay ~10 us each every time
static int lds_attr(int device, const void* fn, int bytes) {
  static std::mutex mu;
  static std::vector<std::tuple<int, const void*, int>> seen;
  {
    std::lock_guard<std::mutex> lk(mu);
    for (const auto& e : seen) if (std::get<0>(e) == device && std::get<1>(e) == fn && std::get<2>(e) >= bytes) return 0;
  }
  CC_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  std::lock_guard<std::mutex> lk(mu);
  for (auto& e : seen) if (std::get<0>(e) == device && std::get<1>(e) == fn) { std::get<2>(e) = std::max(std::get<2>(e), bytes); return 0; }
  seen.emplace_back(device, fn, bytes);
  return 0;
}

// Shared-block layout for the cameras in `seen_any` (observed by at least one rank): which camera owns which
// columns, the tile grid of the Schur products, the direct-sum table -- and every buffer whose size depends on
// them. Called by create (with the locally observed cameras) and again by the multi-GPU attach calls when
// another rank observes a camera this one does not.
static int rig_layout(cc_rig* h, const std::vector<uint8_t>& seen_any) {
  if (const char* e = getenv("CC_RIG_PERSIST")) h->persist_lean_allowed = atoi(e) != 0;
  RigDev& d = h->d;
  const int64_t C = h->C, F = h->F;
  const int kmode = h->kmode;
  h->seen_any = seen_any;
  std::vector<int32_t> pcol((size_t)C, -1), kcol((size_t)C, -1), kset((size_t)C, 0), cobs((size_t)C, -1), obs_cam, colinfo;
  std::vector<uint8_t> fixed((size_t)C, 1);
  int S = 0;
  for (int64_t c = 0; c < C; ++c)
    if (seen_any[(size_t)c]) { cobs[(size_t)c] = (int32_t)obs_cam.size(); obs_cam.push_back((int32_t)c); }
  for (int64_t c = 0; c < C; ++c)
    if (seen_any[(size_t)c] && !h->frozen[(size_t)c]) {
      pcol[(size_t)c] = S;
      fixed[(size_t)c] = 0;
      for (int a = 0; a < 6; ++a) colinfo.push_back((cobs[(size_t)c] << 8) | (0 << 4) | a);
      S += 6;
    }
  const int CK = kmode == RIG_K_NONE ? 0 : (kmode == RIG_K_SHARED ? 1 : (int)C);
  std::vector<int32_t> kscol((size_t)std::max(CK, 1), -1);
  if (kmode == RIG_K_SHARED && !obs_cam.empty()) {
    kscol[0] = S;
    for (int64_t c = 0; c < C; ++c) if (seen_any[(size_t)c]) kcol[(size_t)c] = S;
    for (int j = 0; j < kRigK; ++j) colinfo.push_back((0 << 8) | (2 << 4) | j);
    S += kRigK;
  } else if (kmode == RIG_K_PER_CAMERA) {
    for (int64_t c = 0; c < C; ++c) {
      kset[(size_t)c] = (int32_t)c;
      if (!seen_any[(size_t)c]) continue;
      kcol[(size_t)c] = S;
      kscol[(size_t)c] = S;
      for (int j = 0; j < kRigK; ++j) colinfo.push_back((cobs[(size_t)c] << 8) | (1 << 4) | j);
      S += kRigK;
    }
  }
  colinfo.push_back((0 << 8) | (3 << 4) | 0);   // right-hand side
  const int CO = (int)obs_cam.size();
  if (S > kRigBigMaxS)
    return fail(CC_ERR_BAD_ARGUMENT, "cc_rig_create: %d optimised shared coordinates (6 per observed non-frozen camera%s); at most %d",
                S, kmode ? " + 9 per intrinsics set" : "", kRigBigMaxS);
  if (CO > 64) return fail(CC_ERR_BAD_ARGUMENT, "cc_rig_create: %d observed cameras; at most 64", CO);
  const int DE = kmode ? kDEK : kDE0;
  // beyond what the tuned kernels are built around (S <= 127 shared coordinates, 1536 direct sums): the plain ones
  // (k_rig_elim_big, k_rig_solve_big) -- on one GPU or sharded, over either exchange (this function runs again from the attach
  // calls when a peer observes a camera this rank does not: the layout below is then the large-rig one for the GLOBAL cameras)
  h->big = S > kRigMaxS || CO * DE > 64 * kRigDirectPerLane;
  if (const char* e = getenv("CC_RIG_FORCE_BIG")) h->big = h->big || atoi(e) != 0;   // (test knob: the plain kernels on any problem, sharded or not)
  // the frame form of the sweep feeds the tuned elimination only (the plain large-rig kernels read the 16 x 16 tiles)
  d.fmode = (!kmode && h->frame_allowed && S <= kRigMaxS && CO * DE <= 64 * kRigDirectPerLane &&
             !(getenv("CC_RIG_FORCE_BIG") && atoi(getenv("CC_RIG_FORCE_BIG")) != 0)) ? 1 : 0;
  // with intrinsics: compact records + the FMA sweep (k_rig_sweep_k2) feed the tuned elimination; the plain large-rig kernels
  // read the three tiles of k_rig_sweep_adjk. CC_RIG_K_COMPACT=0 keeps the tile form for A/B and the record-against-tile test.
  // Which one: k_rig_sweep_k2 pays ~6 us per group beyond its passes (two dependent round trips in front, lane sums of 66 + 66
  // accumulators and the assembly behind) against ~2 us per 128 observations in them, k_rig_sweep_adjk runs at 91 % of the fp64
  // pipe whatever the group size (profiles/r05/pmc_rigk_c5_*.csv). Measured at 8 M observations, sweep alone, k2 / tiles:
  // 250 per group 297 / 232 us, 500: 201-213 / 208, 1000: 165 / 197, 2000: 148 / 200, 4000: 140 / 209
  // (profiles/r05/rigk_sweep_vs_group_size.txt) -- compact records from ~450 observations per group on. CC_RIG_K_COMPACT=1 / 0 forces.
  {
    const double per_group = h->NG > 0 ? (double)h->N / (double)h->NG : 0.0;
    bool on = per_group >= 448.0;
    if (const char* e = getenv("CC_RIG_K_COMPACT")) on = atoi(e) != 0;
    d.kcm = (kmode && !h->big && on) ? 1 : 0;
  }
  if (d.kcm) {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device) != hipSuccess || cus < 1) { (void)hipGetLastError(); cus = 256; }
    // a workgroup per group by default: the persistent form (CC_RIG_K2_GRID=1024: four workgroups per compute unit, each looping over
    // groups with the next group's loads under the current one's tail) measured SLOWER, 241 against 201 us at 8 x 2000 x 500 --
    // what a group costs beyond its passes is instructions (lane sums, assembly), not latency, and other workgroups already cover the latency
    (void)cus;
    h->k2_grid = INT32_MAX;
    if (const char* e = getenv("CC_RIG_K2_GRID")) { const int v = atoi(e); if (v >= 1) h->k2_grid = v; }   // (A/B; any grid is correct)
  }
  d.C = (int32_t)C; d.CO = CO; d.CK = CK; d.S = S; d.SW = S + 1;
  d.T = (d.SW + 15) / 16; d.nT = d.T * (d.T + 1) / 2; d.ZS = 16 * d.T + ((d.T & 1) ? 0 : 16);
  d.DE = DE; d.ND = CO * DE;
  d.pc_dir = d.nT * 256; d.pc_fail = d.pc_dir + d.ND; d.pc_gmax = d.pc_fail + 1; d.PC = d.pc_gmax + 1;
  d.nblk = (int)std::max<int64_t>(1, std::min<int64_t>(kRigMaxElimBlocks, h->big ? F : (F + 3) / 4));   // (big: one frame at a time per block)
  std::vector<int16_t> dmap((size_t)DE);
  for (int i = 0; i < 6; ++i) for (int j = 0; j <= i; ++j) dmap[(size_t)(i * (i + 1) / 2 + j)] = (int16_t)(i * 16 + j);
  for (int i = 0; i < 6; ++i) dmap[(size_t)(21 + i)] = (int16_t)(i * 16 + 12);
  if (kmode) {
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 9; ++j) dmap[(size_t)(27 + i * 9 + j)] = (int16_t)(256 + i * 16 + j);
    for (int i = 0; i < 9; ++i) for (int j = 0; j <= i; ++j) dmap[(size_t)(81 + i * (i + 1) / 2 + j)] = (int16_t)(512 + i * 16 + j);
    for (int j = 0; j < 9; ++j) dmap[(size_t)(126 + j)] = (int16_t)(256 + 12 * 16 + j);
  }
  std::vector<uint8_t> ti, tj;
  for (int a = 0; a < d.T; ++a) for (int b = a; b < d.T; ++b) { ti.push_back((uint8_t)a); tj.push_back((uint8_t)b); }
  // solve-step tables: where reduced value e goes
  const int LD = (S + 1) | 1;   // row stride of the reduced system in the solve step's LDS (odd)
  std::vector<int32_t> dir_dst((size_t)std::max(CO * DE, 1), -1), dir_next((size_t)std::max(CO * DE, 1), -1);
  std::vector<int16_t> dir_sa((size_t)std::max(CO * DE, 1), 0), dir_sb((size_t)std::max(CO * DE, 1), 0);
  auto untri_h = [](int idx, int& i, int& j) { i = 0; while ((i + 1) * (i + 2) / 2 <= idx) ++i; j = idx - i * (i + 1) / 2; };
  for (int co = 0; co < CO; ++co) {
    const int c = obs_cam[(size_t)co], p0 = pcol[(size_t)c], k0 = kcol[(size_t)c];
    for (int idx = 0; idx < DE; ++idx) {
      const size_t e = (size_t)co * DE + idx;
      if (idx < 21) {
        if (p0 < 0) continue;
        int i, j; untri_h(idx, i, j);
        dir_dst[e] = (p0 + i) * LD + p0 + j; dir_sa[e] = (int16_t)(p0 + i); dir_sb[e] = (int16_t)(p0 + j);
      } else if (idx < 27) {
        if (p0 >= 0) dir_dst[e] = -2 - (p0 + idx - 21);
      } else if (idx < 81) {
        if (p0 < 0 || k0 < 0) continue;
        const int t = idx - 27, i = t / 9, j = t - i * 9;   // H_ck[i][j]; intrinsics columns follow all pose columns
        dir_dst[e] = (k0 + j) * LD + p0 + i; dir_sa[e] = (int16_t)(k0 + j); dir_sb[e] = (int16_t)(p0 + i);
      } else {
        if (k0 < 0) continue;
        // an intrinsics set shared by several cameras: the first camera's entry collects the others' along a chain
        int prev = -1, next = -1;
        for (int co2 = 0; co2 < co; ++co2) if (kcol[(size_t)obs_cam[(size_t)co2]] == k0) prev = co2;
        for (int co2 = CO - 1; co2 > co; --co2) if (kcol[(size_t)obs_cam[(size_t)co2]] == k0) next = co2;
        if (next >= 0) dir_next[e] = next * DE + idx;
        if (prev >= 0) continue;
        if (idx < 126) {
          int i, j; untri_h(idx - 81, i, j);
          dir_dst[e] = (k0 + i) * LD + k0 + j; dir_sa[e] = (int16_t)(k0 + i); dir_sb[e] = (int16_t)(k0 + j);
        } else {
          dir_dst[e] = -2 - (k0 + idx - 126);
        }
      }
    }
  }
  std::vector<int32_t> tile_dst((size_t)d.nT * 256, -1);
  for (int t = 0; t < d.nT; ++t)
    for (int r = 0; r < 16; ++r)
      for (int c2 = 0; c2 < 16; ++c2) {
        const int pp = 16 * ti[(size_t)t] + r, qq = 16 * tj[(size_t)t] + c2;
        if (pp > qq || pp >= S || qq > S) continue;
        tile_dst[(size_t)t * 256 + r * 16 + c2] = qq < S ? qq * LD + pp : -2 - pp;
      }
  std::vector<int32_t> colpin((size_t)std::max(S, 1), -1);
  for (int k = 0; k < S; ++k) {
    const int info = colinfo[(size_t)k], kind = (info >> 4) & 15, comp = info & 15;
    if (kind == 1) colpin[(size_t)k] = (kset[(size_t)obs_cam[(size_t)(info >> 8)]] << 4) | comp;
    else if (kind == 2) colpin[(size_t)k] = comp;
  }
  std::vector<int32_t> fslot((size_t)F * std::max(CO, 1), -1);
  for (int64_t g = 0; g < h->NG; ++g) fslot[(size_t)h->gframe_h[(size_t)g] * CO + cobs[(size_t)h->gcam_h[(size_t)g]]] = (int32_t)g;
  if (int rc = dev_upload(h, &d.pcol, pcol)) return rc;
  if (int rc = dev_upload(h, &d.kcol, kcol)) return rc;
  if (int rc = dev_upload(h, &d.kset, kset)) return rc;
  if (int rc = dev_upload(h, &d.kscol, kscol)) return rc;
  if (int rc = dev_upload(h, &d.obs_cam, obs_cam)) return rc;
  h->n_runs = 0;
  h->n_shared_runs = 0;
  for (int32_t info : colinfo) if ((info & 15) == 0 && ((info >> 4) & 15) < 3) { ++h->n_runs; if (((info >> 4) & 15) == 2) ++h->n_shared_runs; }
  if (int rc = dev_upload(h, &d.colinfo, colinfo)) return rc;
  if (int rc = dev_upload(h, &d.dmap, dmap)) return rc;
  {
    // (frame form of the sweep: a group's record is the packed G7, whose index of direct entry e -- H_cc (i, j) at i (i + 1) / 2
    // + j, g_c,i at 21 + i -- IS e)
    const bool fm = !kmode && h->frame_allowed && !(S > kRigMaxS || CO * DE > 64 * kRigDirectPerLane || (getenv("CC_RIG_FORCE_BIG") && atoi(getenv("CC_RIG_FORCE_BIG")) != 0));
    std::vector<int32_t> dent((size_t)CO * DE);
    for (int c = 0; c < CO; ++c) for (int e = 0; e < DE; ++e) dent[(size_t)c * DE + e] = (c << 16) | ((fm || d.kcm) ? e : (int)(uint16_t)dmap[(size_t)e]);   // (k_rig_sweep_k2's record starts with the direct entries in this order)
    if (int rc = dev_upload(h, &d.dent, dent)) return rc;
  }
  if (int rc = dev_upload(h, &d.tile_i, ti)) return rc;
  if (int rc = dev_upload(h, &d.tile_j, tj)) return rc;
  if (int rc = dev_upload(h, &d.fslot, fslot)) return rc;
  if (int rc = dev_upload(h, &d.dir_dst, dir_dst)) return rc;
  if (int rc = dev_upload(h, &d.dir_next, dir_next)) return rc;
  if (int rc = dev_upload(h, &d.dir_sa, dir_sa)) return rc;
  if (int rc = dev_upload(h, &d.dir_sb, dir_sb)) return rc;
  if (int rc = dev_upload(h, &d.tile_dst, tile_dst)) return rc;
  if (int rc = dev_upload(h, &d.colpin, colpin)) return rc;
  if (kmode) {   // group records of k_rig_sweep_k2 (a camera held constant depends on who observes it: rebuilt with every layout)
    std::vector<int4> gk((size_t)h->NG * 2);
    for (int64_t g = 0; g < h->NG; ++g) {
      const int64_t s0 = h->goff_h[(size_t)g];
      const int c = h->gcam_h[(size_t)g];
      gk[(size_t)2 * g] = int4{(int)(unsigned)((unsigned long long)s0 & 0xffffffffull), (int)(unsigned)((unsigned long long)s0 >> 32),
                               (int)(h->goff_h[(size_t)g + 1] - s0), h->gframe_h[(size_t)g]};
      gk[(size_t)2 * g + 1] = int4{c, kset[(size_t)c], fixed[(size_t)c] ? 1 : 0, 0};
    }
    if (int rc = dev_upload(h, &d.gk2, gk)) return rc;
  }
  if (!h->d_cam_fixed) { if (int rc = dev_alloc(h, &h->d_cam_fixed, (size_t)C)) return rc; d.cam_fixed = h->d_cam_fixed; }
  CC_HIP(hipMemcpy(h->d_cam_fixed, fixed.data(), fixed.size(), hipMemcpyHostToDevice));
  if (int rc = dev_zeroed(h, &d.Y, (size_t)F * 6 * d.SW)) return rc;
  if (int rc = dev_zeroed(h, &d.partial, (size_t)d.nblk * d.PC)) return rc;
  if (int rc = dev_zeroed(h, &d.vec, (size_t)d.PC + 32)) return rc;
  if (int rc = dev_zeroed(h, &d.vec_stats, (size_t)4 + 256)) return rc;
  h->elim_lds = ((size_t)24 * d.ZS + 4 * 32 + 4 * 1024 + (d.ND > 8 * 64 ? 4 * kRigDirectPerLane * 64 : 0)) * sizeof(double);
  h->solve_lds = ((size_t)S * ((S + 1) | 1) + 5 * 128) * sizeof(double);
  if (h->big) {
    h->elim_lds = ((size_t)6 * 256 + d.ND) * sizeof(double);
    const size_t fixed_lds = ((size_t)5 * 256 + kRigBigPanelDoubles) * sizeof(double);
    const size_t packed = fixed_lds + (size_t)(S + 1) * (S + 2) / 2 * sizeof(double);   // (rows 0..S: the right-hand side is row S)
    h->big_packed = packed + 2048 <= 160 * 1024;
    h->solve_lds = h->big_packed ? packed : fixed_lds;
    if (!h->big_packed)
      if (int rc = dev_zeroed(h, &h->bigA, (size_t)(S + 2) * ((S + 1) | 1))) return rc;
    if (int rc_ = lds_attr(h->device, reinterpret_cast<const void*>(k_rig_elim_big<false>), (int)h->elim_lds)) return rc_;
    if (int rc_ = lds_attr(h->device, reinterpret_cast<const void*>(k_rig_elim_big<true>), (int)h->elim_lds)) return rc_;
    if (int rc_ = lds_attr(h->device, reinterpret_cast<const void*>(k_rig_solve_big<true>), (int)h->solve_lds)) return rc_;
    if (int rc_ = lds_attr(h->device, reinterpret_cast<const void*>(k_rig_solve_big<false>), (int)h->solve_lds)) return rc_;
    rig_drop_graphs(h);
    return 0;
  }
  if (int rc_ = lds_attr(h->device, reinterpret_cast<const void*>(k_rig_elim<false, 8>), (int)h->elim_lds)) return rc_;
  if (int rc_ = lds_attr(h->device, reinterpret_cast<const void*>(k_rig_elim<false, 8, true>), (int)h->elim_lds)) return rc_;
  if (int rc_ = lds_attr(h->device, reinterpret_cast<const void*>(k_rig_elim<false, kRigDirectPerLane, true>), (int)h->elim_lds)) return rc_;
  if (int rc_ = lds_attr(h->device, reinterpret_cast<const void*>(k_rig_elim<true, 8>), (int)h->elim_lds)) return rc_;
  if (int rc_ = lds_attr(h->device, reinterpret_cast<const void*>(k_rig_elim<false, kRigDirectPerLane>), (int)h->elim_lds)) return rc_;
  if (int rc_ = lds_attr(h->device, reinterpret_cast<const void*>(k_rig_elim<true, kRigDirectPerLane>), (int)h->elim_lds)) return rc_;
  if (int rc_ = lds_attr(h->device, reinterpret_cast<const void*>(k_rig_elim<true, 8, false, true>), (int)h->elim_lds)) return rc_;
  if (int rc_ = lds_attr(h->device, reinterpret_cast<const void*>(k_rig_elim<true, kRigDirectPerLane, false, true>), (int)h->elim_lds)) return rc_;
  if (int rc_ = lds_attr(h->device, reinterpret_cast<const void*>(k_rig_reduce<0>), (int)h->solve_lds)) return rc_;
  if (int rc_ = lds_attr(h->device, reinterpret_cast<const void*>(k_rig_reduce<3>), (int)h->solve_lds)) return rc_;
  if (int rc_ = lds_attr(h->device, reinterpret_cast<const void*>(k_rig_solve<0>), (int)h->solve_lds)) return rc_;
  if (int rc_ = lds_attr(h->device, reinterpret_cast<const void*>(k_rig_solve<2>), (int)h->solve_lds)) return rc_;
  // ---- the persistent per-solve form (k_rig_persist_w + k_rig_persist_ctl): poses only, at most four frames per compute
  // unit, every workgroup resident. (Round 3 also kept a GLUED form, k_rig_persist -- the three kernels' bodies in one
  // launch, 444 registers a thread, 81 us per iteration where the three kernels take 47: retired in round 4, no
  // configuration was found where it won; profiles/r03/rig_persist_marks.jsonl has its timeline.)
  h->persist_w_ok = false;
  if (!kmode && h->persist_lean_allowed && S >= 1 && S <= kRpwMaxS && CO <= kRpwMaxCO && C <= kRigPersistMaxC && F <= 4 * 256) {
    std::vector<int32_t> comp;
    for (size_t i = 0; i < tile_dst.size(); ++i) if (tile_dst[i] != -1) comp.push_back((int32_t)i);
    for (int e = 0; e < d.ND; ++e) if (dir_dst[(size_t)e] != -1) comp.push_back(d.pc_dir + e);
    comp.push_back(d.pc_fail);
    comp.push_back(d.pc_gmax);
    RigPersistDev& q = h->pq;
    q.G = (int32_t)((F + 3) / 4); q.K = (int32_t)comp.size(); q.KS = 4 + S; q.NB = 2 + S + 32 * (int32_t)C;
    // the lean form (k_rig_persist_w) takes the fewest frames per workgroup that still leave every XCD a compute unit for the
    // control workgroup: fewer frames per compute unit = more of the chip in the sweep
    const bool lean_shape = (int)comp.size() <= kRpwMaxK;
    h->p_teams = 4;
    for (int t : {1, 2, 4}) if ((F + t - 1) / t <= 255) { h->p_teams = t; break; }   // (G = 256 would fill every XCD: no compute unit for the control)
    q.G = (int32_t)((F + h->p_teams - 1) / h->p_teams);
    if (int rc = dev_upload(h, &q.comp, comp)) return rc;
    {   // the same entries as the lean workers build them (k_rig_persist_w)
      std::vector<int32_t> slots(comp.size());
      for (size_t k = 0; k + 2 < comp.size(); ++k) {
        const int i = comp[k];
        if (i < d.pc_dir) {
          const int t = i / 256, r = (i % 256) / 16, c2 = i % 16;
          slots[k] = (1 << 30) | ((16 * ti[(size_t)t] + r) << 8) | (16 * tj[(size_t)t] + c2);
        } else {
          const int e = i - d.pc_dir;
          slots[k] = ((e / DE) << 16) | (int)(uint16_t)dmap[(size_t)(e % DE)];
        }
      }
      slots[comp.size() - 2] = -1;
      slots[comp.size() - 1] = -2;
      if (int rc = dev_upload(h, &q.slots, slots)) return rc;
    }
    const int Gmax = q.G;
    const size_t n_s = (size_t)Gmax * q.KS * 2, n_a = (size_t)(2 + S) * 2, n_r = (size_t)Gmax * q.K * 2, n_c = (size_t)q.K * 2, n_y = (size_t)q.NB * 2;
    u64* base = nullptr;
    h->p_box_words = n_s + n_a + 2 * n_r + 2 * n_c + n_y + 1;
    if (int rc = dev_zeroed(h, &base, h->p_box_words)) return rc;
    q.sbox = base; q.abox = q.sbox + n_s; q.rbox = q.abox + n_a; q.cbox = q.rbox + n_r; q.ybox = q.cbox + n_c;
    q.pbox = q.ybox + n_y; q.pcbox = q.pbox + n_r;
    q.claim = reinterpret_cast<unsigned*>(q.pcbox + n_c);
    h->p_epoch = 0;
    int cus = 0;
    hipError_t e1 = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device);
    if (e1 != hipSuccess) (void)hipGetLastError();
    if (!h->d_cam_backup) { if (int rc = dev_zeroed(h, &h->d_cam_backup, (size_t)C * 8)) return rc; }
    h->persist_w_ok = false;
    if (e1 == hipSuccess && lean_shape && q.G <= 255 && cus >= 256) {
      int pw = 0;
      const int lb = rpw_lds_doubles(h->p_teams) * 8;
      const void* kw = h->p_teams == 1 ? reinterpret_cast<const void*>(k_rig_persist_w<1>) : h->p_teams == 2 ? reinterpret_cast<const void*>(k_rig_persist_w<2>)
                                                                                           : reinterpret_cast<const void*>(k_rig_persist_w<4>);
      hipError_t e2 = hipFuncSetAttribute(kw, hipFuncAttributeMaxDynamicSharedMemorySize, lb);
      if (e2 == hipSuccess) e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(k_rig_persist_ctl), hipFuncAttributeMaxDynamicSharedMemorySize, (int)rig_persist_ctl_lds(h->solve_lds, d));
      if (e2 == hipSuccess)
        e2 = h->p_teams == 1 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&pw, k_rig_persist_w<1>, 256, (size_t)lb)
           : h->p_teams == 2 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&pw, k_rig_persist_w<2>, 512, (size_t)lb)
                             : hipOccupancyMaxActiveBlocksPerMultiprocessor(&pw, k_rig_persist_w<4>, 1024, (size_t)lb);
      if (e2 != hipSuccess) (void)hipGetLastError();
      h->persist_w_ok = e2 == hipSuccess && pw >= 1;
    }
  }
  rig_drop_graphs(h);
  return 0;
}

// upper bounds of the exchanged vector sizes (every non-frozen camera observed): mailbox slots are allocated
// before the ranks know which cameras the others see
static void rig_exchange_bounds(const cc_rig* h, int* doubles_kind0, int* doubles_kind1) {
  int S = 0, CO = (int)h->C;
  for (int64_t c = 0; c < h->C; ++c) if (!h->frozen[(size_t)c]) S += 6;
  if (h->kmode == RIG_K_SHARED) S += kRigK;
  if (h->kmode == RIG_K_PER_CAMERA) S += kRigK * (int)h->C;
  S = std::min(S, kRigBigMaxS);
  CO = std::min(CO, 64);
  const int T = (S + 1 + 15) / 16;
  *doubles_kind0 = T * (T + 1) / 2 * 256 + CO * (h->kmode ? kDEK : kDE0) + 2 + 32;
  *doubles_kind1 = std::max(4 + S, (int)std::min<int64_t>(h->C, 128));
}

// Grid of the fused reduce + solve + pose-update launch (k_rig_reduce<0/3>). Its blocks wait for each other inside the
// launch (everybody but the last arriver spins on the flag word until the solve step has run), so EVERY block must be
// resident at once -- next to the blocks of the other shards or processes that share the device, whose solve steps may in
// turn wait for OUR posts. The bound is what the occupancy query admits for this kernel's LDS footprint (110 KB at
// S = 114: one block per CU) times the CUs, divided by the launches that share the device; one block per CU is kept in
// hand where several fit (the query reads one high for kernels with 81..112 SGPRs: MI355X guide, residency). The column
// sums and the pose update loop over chunks, so any grid >= 1 is correct; 128 blocks are the most that ever paid
// (2000 frames: 124.4 us per iteration with 128, 127.2 with 64; CC_RIG_REDUCE_BLOCKS for A/B).
// Launches of this rank that share the device with other shards of the same solve (cc_rig_optimize_multi with a repeated
// device id) or with other processes' ranks (cc_rig_exchange_attach with more ranks than visible devices).
static int rig_co_resident(const cc_rig* h) {
  static const int env_co = getenv("CC_RIG_CO_RESIDENT") ? std::max(1, atoi(getenv("CC_RIG_CO_RESIDENT"))) : 0;
  return env_co ? env_co : std::max(1, h->co_resident);
}
// The mailbox exchange on a SHARED device runs UNFUSED: column sums + posts (k_rig_reduce<4>), the solve step as ONE block
// (k_rig_solve<2>), the pose update (k_rig_update). In the fused launch every block but one spins until the solving block
// has run, and the solving block meanwhile polls the PEERS' posts -- a dependency chain across processes through blocks
// that must all stay resident (rank A's spinners <- A's solving block <- B's posts <- B's reduce launch <- B's sweep and
// elimination finding room next to A's and C's spinners). Three ranks x 74 blocks of 110 KB of LDS stalled for 10 s on
// that chain inside longer sessions (round 3, tests/test_gpu_exchange.py); margins on the grid (64 -> 128 -> occupancy
// query -> 7/8 of it) only moved the point where it happened. Unfused, no block of any launch waits for another block
// and exactly one block per rank polls, which is what the intrinsics path does and what survived the same sessions.
// The fused launch stays for a device this rank has to itself (it saves two launch boundaries per iteration).
static bool rig_unfused_exchange(const cc_rig* h) { return h->exchange && !h->big && rig_co_resident(h) > 1; }

static int rig_size_reduce_grid(cc_rig* h) {
  static const int rcap = getenv("CC_RIG_REDUCE_BLOCKS") ? std::max(1, atoi(getenv("CC_RIG_REDUCE_BLOCKS"))) : 128;
  const int co = rig_co_resident(h);
  const size_t key = (h->solve_lds << 8) ^ ((size_t)co << 1) ^ (h->exchange ? 1u : 0u) ^ ((size_t)h->d.PC << 40) ^ ((size_t)h->F << 20);
  if (key == h->reduce_key && h->reduce_blocks > 0) return 0;
  if (h->big || rig_unfused_exchange(h)) {   // column sums only (k_rig_reduce<2> / <4>): nothing waits inside that launch
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(rcap, (h->d.PC + 15) / 16));
    if (blocks != h->reduce_blocks) rig_drop_graphs(h);
    h->reduce_blocks = blocks;
    h->reduce_key = key;
    return 0;
  }
  int per_cu = 0, cus = 0;
  if (h->exchange) CC_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_rig_reduce<3>, 256, h->solve_lds));
  else CC_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_rig_reduce<0>, 256, h->solve_lds));
  CC_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device));
  if (per_cu < 1 || cus < 1) return fail(CC_ERR_HIP, "k_rig_reduce does not fit a compute unit (%zu bytes of LDS)", h->solve_lds);
  if (per_cu > 1) per_cu -= 1;
  per_cu = std::min(per_cu, 8);
  // ... and an eighth of the chip stays free (another tenant's kernels; a grid that needs every last compute unit hangs on
  // the first one that is not available). co > 1 does not get here with an exchange (rig_unfused_exchange); it still
  // divides the bound for CC_RIG_CO_RESIDENT set by hand on a single-rank handle.
  const int64_t resident = std::max<int64_t>(1, (int64_t)per_cu * cus * 7 / 8 / co);
  const int64_t want = std::max<int64_t>((h->d.PC + 15) / 16, (h->F + 15) / 16);
  const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(rcap, want), resident));
  if (blocks != h->reduce_blocks) rig_drop_graphs(h);   // the grid is baked into captured launches
  h->reduce_blocks = blocks;
  h->reduce_key = key;
  return 0;
}

// One round: sweep -> [statistics exchange] -> [init, first round only] -> decision + elimination ->
// reduce + solve step -> pose update (what the next round's sweep evaluates). The first round of a solve is
// the initial evaluation.
static int rig_enqueue_round(cc_rig* h, bool initial, bool profile, bool publish = false) {
  const RigDev& d = h->d;
  struct RoundCount { cc_rig* h; ~RoundCount() { h->enq_round++; } } count_round{h};
  { RigProbe p(h, CC_K_SWEEP, profile);
    if (d.kcm) hipLaunchKernelGGL(k_rig_sweep_k2, dim3((unsigned)std::min<int64_t>(h->NG, h->k2_grid)), dim3(128), 0, h->stream, d);   // (each workgroup loops over groups)
    else if (d.kmode && h->sweep_waves == 1) hipLaunchKernelGGL(k_rig_sweep_adjk<1>, dim3((unsigned)h->NG), dim3(64), 0, h->stream, d);
    else if (d.kmode) hipLaunchKernelGGL(k_rig_sweep_adjk<4>, dim3((unsigned)h->NG), dim3(256), 0, h->stream, d);
    else if (d.fmode) {
      const size_t fl = (size_t)kRigFrameLdsDoubles(d.CO) * 8;
      const bool one = d.CO <= h->frame_waves;   // a wave per group: no loop over groups in the kernel
      if (h->frame_waves == 1 && one) hipLaunchKernelGGL((k_rig_sweep_frame<1, true>), dim3((unsigned)h->F), dim3(64), fl, h->stream, d);
      else if (h->frame_waves == 1) hipLaunchKernelGGL((k_rig_sweep_frame<1, false>), dim3((unsigned)h->F), dim3(64), fl, h->stream, d);
      else if (h->frame_waves == 2 && one) hipLaunchKernelGGL((k_rig_sweep_frame<2, true>), dim3((unsigned)h->F), dim3(128), fl, h->stream, d);
      else if (h->frame_waves == 2) hipLaunchKernelGGL((k_rig_sweep_frame<2, false>), dim3((unsigned)h->F), dim3(128), fl, h->stream, d);
      else if (h->frame_waves == 4 && one) hipLaunchKernelGGL((k_rig_sweep_frame<4, true>), dim3((unsigned)h->F), dim3(256), fl, h->stream, d);
      else if (h->frame_waves == 4) hipLaunchKernelGGL((k_rig_sweep_frame<4, false>), dim3((unsigned)h->F), dim3(256), fl, h->stream, d);
      else if (one) hipLaunchKernelGGL((k_rig_sweep_frame<8, true>), dim3((unsigned)h->F), dim3(512), fl, h->stream, d);
      else hipLaunchKernelGGL((k_rig_sweep_frame<8, false>), dim3((unsigned)h->F), dim3(512), fl, h->stream, d);
    }
    else if (h->sweep_waves == 4) hipLaunchKernelGGL((k_rig_sweep_adj<4>), dim3((unsigned)h->NG), dim3(256), 0, h->stream, d);
    else if (h->sweep_waves == 2) hipLaunchKernelGGL((k_rig_sweep_adj<2>), dim3((unsigned)h->NG), dim3(128), 0, h->stream, d);
    else hipLaunchKernelGGL((k_rig_sweep_adj<1>), dim3((unsigned)h->NG), dim3(64), 0, h->stream, d); }
  if (h->comm || h->exchange) {
    { RigProbe p(h, CC_K_DECIDE, profile); hipLaunchKernelGGL(k_rig_stats, dim3(1), dim3(256), 0, h->stream, d); }
    if (h->comm) { RigProbe p(h, CC_K_ALLREDUCE, profile); if (int rc = comm_allreduce_sum(h->comm, d.vec_stats, 4 + d.S, h->stream)) return rc; }
  }
  if (initial) { RigProbe p(h, CC_K_DECIDE, profile); hipLaunchKernelGGL(k_rig_init, dim3(1 + (unsigned)h->n_runs + (unsigned)h->n_shared_runs * (unsigned)(d.init_slices - 1)), dim3(256), 0, h->stream, d); }
  if (h->big) {
    { RigProbe p(h, CC_K_ELIM, profile);
      if (d.kmode) hipLaunchKernelGGL(k_rig_elim_big<true>, dim3(d.nblk), dim3(256), h->elim_lds, h->stream, d);
      else hipLaunchKernelGGL(k_rig_elim_big<false>, dim3(d.nblk), dim3(256), h->elim_lds, h->stream, d); }
    if (h->exchange) {   // sharded over the mailboxes: posts, then one collecting block (nothing waits inside a launch of many blocks)
      { RigProbe p(h, CC_K_REDUCE, profile); hipLaunchKernelGGL(k_rig_reduce<4>, dim3((unsigned)std::max(1, h->reduce_blocks)), dim3(256), 0, h->stream, d, 0); }
      { RigProbe p(h, CC_K_ALLREDUCE, profile); hipLaunchKernelGGL(k_rig_collect, dim3(1), dim3(256), 0, h->stream, d); }
    } else {
      { RigProbe p(h, CC_K_REDUCE, profile); hipLaunchKernelGGL(k_rig_reduce<2>, dim3((unsigned)std::max(1, h->reduce_blocks)), dim3(256), 0, h->stream, d, 0); }
      if (h->comm) { RigProbe p(h, CC_K_ALLREDUCE, profile); if (int rc = comm_allreduce_sum(h->comm, d.vec, d.PC + 32, h->stream)) return rc; }
    }
    { RigProbe p(h, CC_K_SOLVE, profile);
      if (h->big_packed) hipLaunchKernelGGL(k_rig_solve_big<true>, dim3(1), dim3(256), h->solve_lds, h->stream, d, h->bigA);
      else hipLaunchKernelGGL(k_rig_solve_big<false>, dim3(1), dim3(256), h->solve_lds, h->stream, d, h->bigA); }
    { RigProbe p(h, CC_K_UPDATE, profile); hipLaunchKernelGGL(k_rig_update, dim3((unsigned)((h->F + 15) / 16)), dim3(256), 0, h->stream, d); }
    return 0;
  }
  { RigProbe p(h, CC_K_ELIM, profile);
    const bool small = d.ND <= 8 * 64;
    if (d.kcm && small) hipLaunchKernelGGL((k_rig_elim<true, 8, false, true>), dim3(d.nblk), dim3(256), h->elim_lds, h->stream, d);
    else if (d.kcm) hipLaunchKernelGGL((k_rig_elim<true, kRigDirectPerLane, false, true>), dim3(d.nblk), dim3(256), h->elim_lds, h->stream, d);
    else if (d.kmode && small) hipLaunchKernelGGL((k_rig_elim<true, 8>), dim3(d.nblk), dim3(256), h->elim_lds, h->stream, d);
    else if (d.kmode) hipLaunchKernelGGL((k_rig_elim<true, kRigDirectPerLane>), dim3(d.nblk), dim3(256), h->elim_lds, h->stream, d);
    else if (d.fmode && small) hipLaunchKernelGGL((k_rig_elim<false, 8, true>), dim3(d.nblk), dim3(256), h->elim_lds, h->stream, d);
    else if (d.fmode) hipLaunchKernelGGL((k_rig_elim<false, kRigDirectPerLane, true>), dim3(d.nblk), dim3(256), h->elim_lds, h->stream, d);
    else if (small) hipLaunchKernelGGL((k_rig_elim<false, 8>), dim3(d.nblk), dim3(256), h->elim_lds, h->stream, d);
    else hipLaunchKernelGGL((k_rig_elim<false, kRigDirectPerLane>), dim3(d.nblk), dim3(256), h->elim_lds, h->stream, d); }
  // (every block of the fused launch must be resident at once: its grid comes from rig_size_reduce_grid, rig_begin)
  const unsigned rblocks = (unsigned)std::max(1, h->reduce_blocks);
  if (h->comm) {
    { RigProbe p(h, CC_K_REDUCE, profile); hipLaunchKernelGGL(k_rig_reduce<2>, dim3(rblocks), dim3(256), 0, h->stream, d, 0); }
    { RigProbe p(h, CC_K_ALLREDUCE, profile); if (int rc = comm_allreduce_sum(h->comm, d.vec, d.PC + 32, h->stream)) return rc; }
    { RigProbe p(h, CC_K_SOLVE, profile); hipLaunchKernelGGL(k_rig_solve<0>, dim3(1), dim3(256), h->solve_lds, h->stream, d, 0); }
    { RigProbe p(h, CC_K_UPDATE, profile); hipLaunchKernelGGL(k_rig_update, dim3((unsigned)((h->F + 15) / 16)), dim3(256), 0, h->stream, d); }
  } else if (rig_unfused_exchange(h)) {   // shared device: nobody waits for a block of its own launch (see rig_unfused_exchange)
    { RigProbe p(h, CC_K_REDUCE, profile); hipLaunchKernelGGL(k_rig_reduce<4>, dim3(rblocks), dim3(256), 0, h->stream, d, 0); }
    { RigProbe p(h, CC_K_SOLVE, profile); hipLaunchKernelGGL(k_rig_solve<2>, dim3(1), dim3(256), h->solve_lds, h->stream, d, publish ? 1 : 0); }
    { RigProbe p(h, CC_K_UPDATE, profile); hipLaunchKernelGGL(k_rig_update, dim3((unsigned)((h->F + 15) / 16)), dim3(256), 0, h->stream, d); }
  } else {   // reduce + solve step + pose update in one launch
    RigProbe p(h, CC_K_SOLVE, profile);
    if (h->exchange) hipLaunchKernelGGL(k_rig_reduce<3>, dim3(rblocks), dim3(256), h->solve_lds, h->stream, d, publish ? 1 : 0);
    else hipLaunchKernelGGL(k_rig_reduce<0>, dim3(rblocks), dim3(256), h->solve_lds, h->stream, d, publish ? 1 : 0);
  }
  return 0;
}

// head of a solve: records of the starting point for the first sweep
static void rig_enqueue_prep(cc_rig* h) {
  hipLaunchKernelGGL(k_rig_records, dim3(1), dim3(256), 0, h->stream, h->d);
  hipLaunchKernelGGL(k_rig_update, dim3((unsigned)((h->F + 15) / 16)), dim3(256), 0, h->stream, h->d);
}

static int rig_write_ctl(cc_rig* h, const LmCtl& c) {
  CC_HIP(hipMemcpyAsync(h->d.ctl, &c, sizeof(c), hipMemcpyHostToDevice, h->stream));
  CC_HIP(hipMemcpyAsync(h->d.ctl_next, &c, sizeof(c), hipMemcpyHostToDevice, h->stream));
  return 0;
}
// control block as the last kernel left it; *wait_failed (optional): the failure word of k_rig_reduce's in-kernel waits
static int rig_read_ctl(cc_rig* h, LmCtl* c, bool* wait_failed = nullptr) {
  CC_HIP(hipMemcpyAsync(h->h_ctl, h->d.ctl_next, sizeof(LmCtl) + 4 * sizeof(unsigned), hipMemcpyDeviceToHost, h->stream));
  CC_HIP(hipStreamSynchronize(h->stream));
  *c = *h->h_ctl;
  if (wait_failed) *wait_failed = reinterpret_cast<const unsigned*>(h->h_ctl + 1)[3] != 0u;
  return 0;
}

// Waits for the chunk just enqueued: spins on the sequence word its last reduce launch stores into pinned host memory
// (no copy engine, no stream synchronisation on the way), then takes the control block and the failure word from next
// to it. A stream that has gone idle without the word showing up (a kernel fault) falls back to a copy.
// `lean`: the chunk is a lean persistent solve -- its ONLY publisher is the control workgroup on h->stream2, and the workers
// on h->stream leave as soon as they have seen `done` in their boxes, a few microseconds BEFORE the control has written the
// control block and the sequence word. The fallback is therefore taken only when BOTH streams are idle (ADVICE round 3:
// with h->stream alone the host could re-synchronise its count and read a half-written control block while the control
// workgroup was still publishing, and the late publication then satisfied the NEXT solve's wait at once).
static int rig_wait_published(cc_rig* h, LmCtl* c, bool* wait_failed, bool lean = false) {
  const unsigned long long want = ++h->pub_count;
  for (unsigned spins = 0;; ++spins) {
    if (__atomic_load_n(const_cast<const unsigned long long*>(h->host_pub), __ATOMIC_ACQUIRE) == want) break;
    if ((spins & 0xfffu) == 0xfffu) {
      hipError_t q = hipStreamQuery(h->stream);
      if (q == hipSuccess && lean && h->stream2) q = hipStreamQuery(h->stream2);
      if (q == hipSuccess) {
        if (__atomic_load_n(const_cast<const unsigned long long*>(h->host_pub), __ATOMIC_ACQUIRE) == want) break;
        h->pub_count = __atomic_load_n(const_cast<const unsigned long long*>(h->host_pub), __ATOMIC_ACQUIRE);
        return rig_read_ctl(h, c, wait_failed);
      }
      if (q != hipErrorNotReady) return fail(CC_ERR_HIP, "stream failed while waiting for the rig solver: %s", hipGetErrorString(q));
    }
  }
  std::memcpy(c, const_cast<const unsigned long long*>(h->host_pub) + 2, sizeof(LmCtl));
  *wait_failed = h->host_pub[2 + sizeof(LmCtl) / 8] != 0ull;
  return 0;
}

}  // namespace cc

extern "C" void cc_rig_destroy(cc_rig* h);

namespace cc {
static std::mutex g_perm_mu;
static std::vector<int64_t> g_perm_cache;   // storage of the last destroyed handle's permutation (cc_rig_create takes it over)
void rig_release_host_caches() {
  std::lock_guard<std::mutex> lk(g_perm_mu);
  std::vector<int64_t>().swap(g_perm_cache);
}
struct HostPhases {   // CC_RIG_HOST_TIMING=1: wall milliseconds of the host-side phases of a call, to stderr
  const char* who; bool on; std::chrono::steady_clock::time_point t; std::string line;
  explicit HostPhases(const char* w) : who(w), on(getenv("CC_RIG_HOST_TIMING") != nullptr), t(std::chrono::steady_clock::now()) {}
  void mark(const char* name) {
    if (!on) return;
    const auto n = std::chrono::steady_clock::now();
    char b[96]; snprintf(b, sizeof(b), " %s %.3f", name, std::chrono::duration<double, std::milli>(n - t).count());
    line += b; t = n;
  }
  ~HostPhases() { if (on) fprintf(stderr, "[%s]%s\n", who, line.c_str()); }
};
struct RigCreateGuard {  // releases a half-built handle on every early return
  cc_rig* h;
  bool ok = false;
  ~RigCreateGuard() { if (!ok) cc_rig_destroy(h); }
};
}  // namespace cc

extern "C" {

}  // extern "C"

namespace cc {
// Where cc_rig_create reads the observations: three flat arrays (cc_rig_create), or records the caller keeps frame by frame
// (cc_rig_optimize_frames: ExtrinsicsCalibrator's per-frame lists, read in place).
struct RigObsSource {
  const uint32_t* cam = nullptr; const uint64_t* world = nullptr; const float* uv = nullptr;
  const void* const* frames = nullptr; cc_obs_layout lay{};
};
struct RigFlatFrame {
  const uint32_t* cam; const uint64_t* world; const float* uv;
  RigFlatFrame(const RigObsSource& s, int64_t, int64_t base) : cam(s.cam + base), world(s.world + base), uv(s.uv + 2 * base) {}
  uint64_t camera(int64_t i) const { return cam[i]; }
  uint64_t point(int64_t i) const { return world[i]; }
  void pixel(int64_t i, float* o) const { o[0] = uv[2 * i]; o[1] = uv[2 * i + 1]; }
};
struct RigRecordFrame {
  const unsigned char* p; int64_t stride, oc, ow, ou;
  RigRecordFrame(const RigObsSource& s, int64_t f, int64_t)
      : p(static_cast<const unsigned char*>(s.frames[f])), stride(s.lay.stride), oc(s.lay.camera_offset), ow(s.lay.world_offset), ou(s.lay.uv_offset) {}
  uint64_t camera(int64_t i) const { uint64_t v; std::memcpy(&v, p + i * stride + oc, 8); return v; }
  uint64_t point(int64_t i) const { uint64_t v; std::memcpy(&v, p + i * stride + ow, 8); return v; }
  void pixel(int64_t i, float* o) const { std::memcpy(o, p + i * stride + ou, 8); }
};
struct RigRegroupPart { std::vector<int32_t> gframe, gcam, per_frame; std::vector<int64_t> gend; int64_t bad = -1; int bad_kind = 0; };

// frames [f0, f1) of the regrouping pass (one host thread): counting sort by camera inside each frame (stable: observation
// order is kept within a group); only the cameras that occur in the frame are visited, so thousands of idle cameras cost
// nothing. Ids are checked on the way; the regrouped pixels / world indices go straight into the staging block.
template <class Frame, bool INV>
static void rig_regroup_part(const RigObsSource& src, int64_t C, int64_t n_world, const int64_t* off, int64_t f0, int64_t f1,
                             int64_t* perm, float* uv_s, int32_t* widx_s, uint8_t* seen_p, RigRegroupPart& L,
                             const std::function<bool(int64_t, int64_t)>& flush) {
  std::vector<int64_t> cnt((size_t)C, 0), start((size_t)C, 0);
  std::vector<uint32_t> present;
  // the regrouped arrays go up in (at most) four pieces per thread, each as soon as it is written: the transfers of the
  // first three run under the regrouping of what follows
  const int64_t n_part = off[f1] - off[f0];
  const int64_t piece = n_part >= ((int64_t)1 << 18) ? (n_part + 3) / 4 : n_part + 1;   // (small ranges: one piece -- a copy costs the host 7.5 us)
  int64_t f_sent = f0, next_flush = off[f0] + piece;
  for (int64_t f = f0; f < f1; ++f) {
    if (off[f] >= next_flush && f > f_sent) {
      if (!flush(f_sent, f)) { L.bad = off[f_sent]; L.bad_kind = 2; return; }
      f_sent = f;
      next_flush = off[f] + piece;
    }
    const int64_t base = off[f], n = off[f + 1] - off[f];
    if (n == 0) { L.per_frame.push_back(0); continue; }
    const Frame fr(src, f, base);
    present.clear();
    for (int64_t i = 0; i < n; ++i) {
      const uint64_t c = fr.camera(i);
      if (c >= (uint64_t)C) { L.bad = base + i; L.bad_kind = 0; return; }
      if (fr.point(i) >= (uint64_t)n_world) { L.bad = base + i; L.bad_kind = 1; return; }
      if (cnt[c]++ == 0) present.push_back((uint32_t)c);
    }
    std::sort(present.begin(), present.end());
    int64_t pos = base;
    for (uint32_t c : present) {
      start[c] = pos;
      L.gframe.push_back((int32_t)f);
      L.gcam.push_back((int32_t)c);
      __atomic_store_n(seen_p + c, (uint8_t)1, __ATOMIC_RELAXED);
      pos += cnt[c];
      L.gend.push_back(pos);
    }
    L.per_frame.push_back((int32_t)present.size());
    for (int64_t i = 0; i < n; ++i) {
      const int64_t dst = start[fr.camera(i)]++;
      if (INV) perm[base + i] = dst; else perm[dst] = base + i;   // (records: where observation i of the frame went -- the costs are written record by record)
      fr.pixel(i, uv_s + 2 * dst);
      widx_s[dst] = (int32_t)fr.point(i);
    }
    for (uint32_t c : present) cnt[c] = 0;
  }
  if (f1 > f_sent && !flush(f_sent, f1)) { L.bad = off[f_sent]; L.bad_kind = 2; }
}
}  // namespace cc

extern "C" {

static int rig_create_impl(int32_t device, int64_t C, int64_t F, int64_t n_world, const int64_t* off,
                           const cc::RigObsSource& src,
                           const float* world_xyz, const uint8_t* cam_frozen, double huber_a, int kmode, cc_rig** out) {
  using namespace cc;
  HostPhases hp("cc_rig_create");
  if (!out || !off || C < 1 || F < 1 || n_world < 0) return fail(CC_ERR_BAD_ARGUMENT, "cc_rig_create: bad arguments");
  if (off[0] != 0) return fail(CC_ERR_BAD_ARGUMENT, "obs_frame_offsets[0] must be 0");
  const int64_t N = off[F];
  for (int64_t f = 0; f < F; ++f)
    if (off[f + 1] < off[f]) return fail(CC_ERR_BAD_ARGUMENT, "obs_frame_offsets must be non-decreasing");
  // device-side indices are 32-bit (world point, group, camera) and launch grids are unsigned
  if (C > (1 << 20) || F >= INT32_MAX || n_world >= INT32_MAX || N >= ((int64_t)1 << 40))
    return fail(CC_ERR_BAD_ARGUMENT, "cc_rig_create: problem too large (cameras < 2^20, frames and world points < 2^31)");
  if (N > 0 && ((!src.frames && (!src.cam || !src.world || !src.uv)) || !world_xyz)) return fail(CC_ERR_BAD_ARGUMENT, "cc_rig_create: NULL arrays");
  if (src.frames) {
    const cc_obs_layout& l = src.lay;
    if (l.stride < 8 || l.camera_offset < 0 || l.world_offset < 0 || l.uv_offset < 0 || l.camera_offset + 8 > l.stride ||
        l.world_offset + 8 > l.stride || l.uv_offset + 8 > l.stride || (l.cost_offset >= 0 && l.cost_offset + 8 > l.stride))
      return fail(CC_ERR_BAD_ARGUMENT, "cc_rig_optimize_frames: record layout does not fit its stride");
    for (int64_t f = 0; f < F; ++f)
      if (off[f + 1] > off[f] && !src.frames[f]) return fail(CC_ERR_BAD_ARGUMENT, "cc_rig_optimize_frames: frame %lld has no records", (long long)f);
  }
  hp.mark("checks");   // (camera and world point ids: inside the regrouping pass)
  if (int rc = select_device(device)) return rc;
  cc_rig* h = new cc_rig();
  RigCreateGuard guard{h};
  h->device = device; h->C = C; h->F = F; h->N = N; h->P = n_world; h->kmode = kmode;
  // ---- regroup: within each frame, stable sort by camera -> (frame, camera) groups. ONE pass over the caller's arrays, cut
  // into contiguous frame ranges of equal observation counts for up to 16 host threads: ids checked, groups listed, the
  // permutation kept (per-observation costs go back in the caller's order) and the regrouped pixel / world-index arrays written
  // straight into the cached pinned staging block the uploads below read (8 M observations: 55 ms on one thread through three
  // freshly allocated vectors -- more than the whole solve -- before round 4's end)
  {
    std::lock_guard<std::mutex> lk(g_perm_mu);
    h->perm.swap(g_perm_cache);          // (an earlier handle's storage: already faulted in; never shrunk, h->N is the length)
  }
  if ((int64_t)h->perm.size() < N) h->perm.resize((size_t)N);
  const size_t b_uv = ((size_t)N * 2 * sizeof(float) + 255) & ~(size_t)255;
  bool st_cached = false;
  char* st = static_cast<char*>(staging_get(b_uv + (size_t)N * sizeof(int32_t) + 256, &st_cached));
  if (!st) return fail(CC_ERR_HIP, "cc_rig_create: pinned staging memory could not be allocated");
  struct StGuard {   // the uploads read the block until the handle's stream has drained
    void* p; cc_rig* h;
    ~StGuard() { if (h->stream) (void)hipStreamSynchronize(h->stream); staging_put(p); }
  } stg{st, h};
  float* uv_s = reinterpret_cast<float*>(st);
  int32_t* widx_s = reinterpret_cast<int32_t*>(st + b_uv);
  if (int rc = stream_get(device, &h->stream)) return rc;
  struct UpGuard {   // pinned staging of the table uploads (dev_upload): given back once the stream has drained
    cc_rig* h; void* p = nullptr;
    ~UpGuard() { if (p) { if (h->stream) (void)hipStreamSynchronize(h->stream); h->up_stage = nullptr; h->up_cap = h->up_used = 0; staging_put(p); } }
  } upg{h};
  {
    // tables: per group 16 B, per frame 4 CO + 128 B, per shared column / direct entry a few words -- 4 MB covers BASELINE configs[4]
    // several times over; a table that does not fit takes the synchronous copy
    const size_t cap = (size_t)4 << 20;
    bool up_cached = false;
    upg.p = staging_get(cap, &up_cached);
    if (upg.p) { h->up_stage = static_cast<char*>(upg.p); h->up_cap = cap; h->up_used = 0; }
  }
  float* duv = nullptr;
  int32_t* dw = nullptr;
  if (int rc = dev_alloc(h, &duv, (size_t)N * 2)) return rc;
  if (int rc = dev_alloc(h, &dw, (size_t)N)) return rc;
  // (called by the regrouping threads, each for the frames it has just written)
  const auto flush = [&](int64_t fa, int64_t fb) -> bool {
    const int64_t a = off[fa], n = off[fb] - off[fa];
    if (n <= 0) return true;
    if (hipSetDevice(device) != hipSuccess) return false;
    return hipMemcpyAsync(duv + 2 * a, uv_s + 2 * a, (size_t)n * 2 * sizeof(float), hipMemcpyHostToDevice, h->stream) == hipSuccess &&
           hipMemcpyAsync(dw + a, widx_s + a, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, h->stream) == hipSuccess;
  };
  std::vector<int64_t>& goff = h->goff_h;
  goff.assign(1, 0);
  h->fgoff_h.assign((size_t)F + 1, 0);
  std::vector<int32_t>& gframe = h->gframe_h;
  std::vector<int32_t>& gcam = h->gcam_h;
  std::vector<uint8_t> seen((size_t)C, 0);
  {
    // (per-thread tables of C entries: one thread when the camera ids are sparse in a huge range)
    const int parts = C <= 65536 ? parallel_parts(N, (int64_t)1 << 15) : 1;   // (parts of >= 32 k observations on the worker pool: no thread is created per call, so fine parts are cheap)
    std::vector<int64_t> pf((size_t)parts + 1, 0);
    if (int rc = cc_partition_frames(F, off, parts, pf.data())) return rc;
    using Part = RigRegroupPart;
    std::vector<Part> part((size_t)parts);
    int64_t* perm = h->perm.data();
    h->perm_inverse = src.frames != nullptr;
    uint8_t* seen_p = seen.data();
    parallel_tasks(parts, [&](int t) {
      if (src.frames) rig_regroup_part<RigRecordFrame, true>(src, C, n_world, off, pf[(size_t)t], pf[(size_t)t + 1], perm, uv_s, widx_s, seen_p, part[(size_t)t], flush);
      else rig_regroup_part<RigFlatFrame, false>(src, C, n_world, off, pf[(size_t)t], pf[(size_t)t + 1], perm, uv_s, widx_s, seen_p, part[(size_t)t], flush);
    });
    for (const Part& L : part)   // (parts are in frame order: the first one with a bad id holds the first bad observation)
      if (L.bad >= 0)
        return L.bad_kind == 2 ? fail(CC_ERR_HIP, "cc_rig_create: upload of the regrouped observations failed: %s", hipGetErrorString(hipGetLastError()))
                               : fail(CC_ERR_BAD_ARGUMENT, L.bad_kind == 0 ? "observation %lld: camera id out of range" : "observation %lld: world point id out of range", (long long)L.bad);
    size_t ng_total = 0;
    for (const Part& L : part) ng_total += L.gframe.size();
    gframe.reserve(ng_total); gcam.reserve(ng_total); goff.reserve(ng_total + 1);
    for (int t = 0; t < parts; ++t) {
      const Part& L = part[(size_t)t];
      gframe.insert(gframe.end(), L.gframe.begin(), L.gframe.end());
      gcam.insert(gcam.end(), L.gcam.begin(), L.gcam.end());
      goff.insert(goff.end(), L.gend.begin(), L.gend.end());
      for (size_t i = 0; i < L.per_frame.size(); ++i) {
        const int64_t f = pf[(size_t)t] + (int64_t)i;
        h->fgoff_h[(size_t)f + 1] = h->fgoff_h[(size_t)f] + L.per_frame[i];
      }
    }
  }
  hp.mark("regroup");
  const int64_t NG = (int64_t)gframe.size();
  h->NG = NG;
  if (NG == 0) return fail(CC_ERR_BAD_ARGUMENT, "cc_rig_create: no observations");
  if (NG >= INT32_MAX) return fail(CC_ERR_BAD_ARGUMENT, "cc_rig_create: too many (frame, camera) groups");
  std::vector<int32_t> cam_goff((size_t)C + 1, 0), cam_glist((size_t)NG);
  for (int64_t g = 0; g < NG; ++g) cam_goff[(size_t)gcam[g] + 1]++;
  for (int64_t c = 0; c < C; ++c) cam_goff[c + 1] += cam_goff[c];
  {
    std::vector<int32_t> fill(cam_goff.begin(), cam_goff.end() - 1);
    for (int64_t g = 0; g < NG; ++g) cam_glist[(size_t)fill[gcam[g]]++] = (int32_t)g;
  }
  h->frozen.assign((size_t)C, 0);
  if (cam_frozen) for (int64_t c = 0; c < C; ++c) h->frozen[(size_t)c] = cam_frozen[c] ? 1 : 0;
  h->seen = seen;
  hp.mark("gather");

  RigDev& d = h->d;
  d.F = F; d.N = N; d.NG = NG;
  d.huber_a = (kmode && !(huber_a > 0.0)) ? 1e300 : huber_a;   // extension: a <= 0 switches the loss off
  d.huber_b = d.huber_a * d.huber_a; d.huber_2a = d.huber_a + d.huber_a; d.huber_ha = 0.5 * d.huber_a;
  d.comm = 0; d.rank = 0; d.nranks = 1;
  d.kmode = kmode; d.gstride = kmode ? 768 : 256;
  d.init_slices = (int32_t)std::min<int64_t>(16, std::max<int64_t>(1, (NG + 511) / 512));
  d.uv = duv; d.widx = dw;   // (uploaded piece by piece by the regrouping threads, above)
  {
    float* w = nullptr;
    if (int rc = dev_alloc(h, &w, (size_t)n_world * 3)) return rc;
    if (n_world > 0) CC_HIP(hipMemcpy(w, world_xyz, (size_t)n_world * 3 * sizeof(float), hipMemcpyHostToDevice));
    d.wxyz = w;
    float* ox = nullptr;
    if (int rc = dev_alloc(h, &ox, (size_t)N * 3)) return rc;
    hipLaunchKernelGGL(k_rig_expand_xyz, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, h->stream, N, d.widx, d.wxyz, ox);
    CC_HIP(hipGetLastError());
    d.oxyz = ox;
  }
  hp.mark("upload_obs");
  if (int rc = dev_upload(h, &d.goff, goff)) return rc;
  if (int rc = dev_upload(h, &d.gframe, gframe)) return rc;
  if (int rc = dev_upload(h, &d.gcam, gcam)) return rc;
  if (int rc = dev_upload(h, &d.fgoff, h->fgoff_h)) return rc;
  {
    // slot records of the frame form (read only by its one-group-per-wave variant: frames of at most eight groups)
    std::vector<int4> fslot((size_t)F * 8, int4{0, 0, 0, 0});
    for (int64_t f = 0; f < F; ++f) {
      const int64_t a = h->fgoff_h[(size_t)f], b = h->fgoff_h[(size_t)f + 1];
      for (int64_t g = a; g < b && g - a < 8; ++g) {
        int4& sl = fslot[(size_t)f * 8 + (size_t)(g - a)];
        sl.x = (int)(unsigned)((unsigned long long)goff[(size_t)g] & 0xffffffffull);
        sl.y = (int)(unsigned)((unsigned long long)goff[(size_t)g] >> 32);
        sl.z = (int)(goff[(size_t)g + 1] - goff[(size_t)g]);
        sl.w = gcam[(size_t)g];
      }
    }
    if (int rc = dev_upload(h, &d.fwave, fslot)) return rc;
  }
  if (int rc = dev_upload(h, &d.cam_goff, cam_goff)) return rc;
  if (int rc = dev_upload(h, &d.cam_glist, cam_glist)) return rc;
  if (const char* e = getenv("CC_RIG_SWEEP_FRAME")) h->frame_allowed = atoi(e) != 0;
  hp.mark("upload_idx");
  if (int rc = rig_layout(h, seen)) return rc;
  hp.mark("layout");
  {
    // frame form: waves per frame workgroup -- enough workgroups x waves to fill the chip's 4096 wave slots, never more waves
    // than a frame has groups (each wave sweeps whole groups)
    int nw = 1;
    while (nw < 8 && (int64_t)F * nw < 4096) nw *= 2;
    while (nw > 1 && nw / 2 >= d.CO) nw /= 2;
    // ... and a wave PER group when the rig has at most eight observed cameras and the frames alone do not fill the chip twice
    // over: that kernel has no loop over groups and runs four waves per SIMD (three with the loop)
    if (d.CO <= 8 && (int64_t)F * nw < 2 * 4096) { nw = 1; while (nw < d.CO) nw *= 2; }
    h->frame_waves = nw;
    if (const char* e = getenv("CC_RIG_FRAME_WAVES")) { const int v = atoi(e); if (v == 1 || v == 2 || v == 4 || v == 8) h->frame_waves = v; }
  }
  if (int rc = dev_zeroed(h, &d.fsum, (size_t)2 * F * 32)) return rc;
  const size_t CKn = (size_t)std::max(d.CK, 1);
  if (int rc = dev_zeroed(h, &d.cam, (size_t)2 * C * 8)) return rc;
  if (int rc = dev_zeroed(h, &d.pose, (size_t)2 * F * 8)) return rc;
  if (int rc = dev_zeroed(h, &d.camrec, (size_t)C * 32)) return rc;
  if (int rc = dev_zeroed(h, &d.frec, (size_t)F * 32)) return rc;
  if (int rc = dev_alloc(h, &d.gblocks, (size_t)2 * NG * d.gstride)) return rc;
  {
    // sweep workgroup size (poses-only problem; measured on MI355X, profiles/r02/rig_sweep_waves.txt): a workgroup per
    // (frame, camera) group of 4, 2 or 1 waves. Fewer waves per group = more, smaller workgroups in flight at
    // different phases (better latency hiding, no cross-wave reduction) but less parallelism inside a group:
    // one wave when the groups alone oversubscribe the chip's 4096 wave slots or have a single 64-observation chunk,
    // two when they at least fill a quarter of them, four otherwise. CC_RIG_SWEEP_WG_WAVES forces one (A/B, tests).
    const double per_group = NG > 0 ? (double)N / (double)NG : 0.0;
    h->sweep_waves = kmode ? 4 : ((NG >= 4096 || per_group <= 64.0) ? 1 : (NG >= 1024 ? 2 : 4));
    // (k_rig_sweep_adj: one wave per group is the fastest at every measured shape that has a thousand groups)
    if (!kmode) h->sweep_waves = (NG >= 1024 || per_group <= 64.0) ? 1 : (NG >= 512 ? 2 : 4);
    if (kmode) h->sweep_waves = (NG >= 1024 || per_group <= 64.0) ? 1 : 4;
    if (const char* e = getenv("CC_RIG_SWEEP_WG_WAVES")) { const int v = atoi(e); if ((!kmode && (v == 1 || v == 2 || v == 4)) || (kmode && (v == 1 || v == 4))) h->sweep_waves = v; }
  }
  if (int rc = dev_zeroed(h, &d.intr, 2 * CKn * 16)) return rc;
  if (int rc = dev_zeroed(h, &d.krec, CKn * 32)) return rc;
  if (int rc = dev_zeroed(h, &d.ghdk, (size_t)(kmode ? NG * 16 : 16))) return rc;
  if (int rc = dev_zeroed(h, &h->init_intr, CKn * 16)) return rc;
  if (int rc = dev_zeroed(h, &h->d_kmask, CKn)) return rc;
  d.kmask = h->d_kmask;
  if (int rc = dev_zeroed(h, &d.gstats, (size_t)std::max<int64_t>(NG, F) * 2)) return rc;   // (frame form: one row per frame)
  if (int rc = dev_zeroed(h, &d.fstats, (size_t)F * 2)) return rc;
  if (int rc = dev_zeroed(h, &d.ghd0, (size_t)NG * 8)) return rc;
  if (int rc = dev_zeroed(h, &d.gcomp, (size_t)2 * NG * (kmode ? kRigCompK : 64))) return rc;
  if (int rc = dev_zeroed(h, &d.sp, (size_t)F * 8)) return rc;
  if (int rc = dev_zeroed(h, &d.ss, (size_t)256)) return rc;
  if (int rc = dev_zeroed(h, &d.ds, (size_t)256)) return rc;
  if (int rc = dev_zeroed(h, &d.shared_stats, (size_t)64)) return rc;   // [0..3] statistics, [8..] timing marks (CC_RIG_TIMING builds)
  // one piece: control block | its copy for the host (ctl_next) | 16 synchronisation words (RigDev::arrive) | publication
  // counter -- a solve starts by zeroing the first three with ONE fill (rig_begin)
  if (int rc = dev_zeroed(h, &d.ctl, (size_t)4)) return rc;
  static_assert(sizeof(LmCtl) % 16 == 0 && sizeof(LmCtl) >= 16 * sizeof(unsigned) + 8, "arrive words and the counter behind ctl_next");
  d.ctl_next = d.ctl + 1;
  d.arrive = reinterpret_cast<unsigned*>(d.ctl + 2);
  d.pub_seq = reinterpret_cast<unsigned long long*>(d.ctl + 3);
  if (int rc = dev_alloc(h, &d.opts, (size_t)1)) return rc;
  d.log_cap = 4096;
  if (int rc = dev_alloc(h, &d.log, (size_t)d.log_cap)) return rc;
  if (int rc = dev_alloc(h, &h->init_cam, (size_t)C * 8)) return rc;
  if (int rc = dev_alloc(h, &h->init_pose, (size_t)F * 8)) return rc;
  if (int rc = dev_alloc(h, &h->d_cost, (size_t)N)) return rc;
  {
    // one cached 512-byte pinned block: [0..191] publication (sequence word, control block, failure word) | [192..] staging
    // of rig_read_ctl
    char* pin = static_cast<char*>(pinned_block_get());
    if (!pin) return fail(CC_ERR_HIP, "hipHostMalloc failed");
    h->pinned = pin;
    h->host_pub = reinterpret_cast<volatile unsigned long long*>(pin);
    h->h_ctl = reinterpret_cast<LmCtl*>(pin + 192);
    static_assert(16 + sizeof(LmCtl) + 8 <= 192 && 192 + sizeof(LmCtl) + 16 <= 512, "pinned block layout");
    h->host_pub[0] = 0ull;   // (a recycled block may carry an old sequence number; the device counter starts at 0)
    h->host_pub[22] = 0ull;  // residency word of the lean persistent form (rig_launch)
    h->pub_count = 0;
    void* dev_view = nullptr;
    CC_HIP(hipHostGetDevicePointer(&dev_view, pin, 0));
    d.host_pub = reinterpret_cast<unsigned long long*>(dev_view);
  }
  CC_HIP(hipStreamSynchronize(h->stream));   // the observations are up (the staging block goes back to the cache)
  hp.mark("alloc_rest");
  guard.ok = true;
  *out = h;
  return CC_OK;
}

int cc_rig_create(int32_t device, int64_t C, int64_t F, int64_t n_world, const int64_t* off,
                  const uint32_t* obs_cam, const uint64_t* obs_world, const float* obs_uv,
                  const float* world_xyz, const uint8_t* cam_frozen, double huber_a, cc_rig** out) {
  cc::RigObsSource src;
  src.cam = obs_cam; src.world = obs_world; src.uv = obs_uv;
  return rig_create_impl(device, C, F, n_world, off, src, world_xyz, cam_frozen, huber_a, cc::RIG_K_NONE, out);
}

// EXTENSION: the same handle with 9 intrinsics shared by all cameras; obs_uv in pixels
int cc_rigk_create(int32_t device, int64_t C, int64_t F, int64_t n_world, const int64_t* off,
                   const uint32_t* obs_cam, const uint64_t* obs_world, const float* obs_uv_pixels,
                   const float* world_xyz, const uint8_t* cam_frozen, double huber_a, cc_rig** out) {
  cc::RigObsSource src;
  src.cam = obs_cam; src.world = obs_world; src.uv = obs_uv_pixels;
  return rig_create_impl(device, C, F, n_world, off, src, world_xyz, cam_frozen, huber_a, cc::RIG_K_SHARED, out);
}

// EXTENSION: one set of 9 intrinsics PER CAMERA (BASELINE.json configs[4]: "full intrinsics+extrinsics co-optimisation")
int cc_rigk_create_per_camera(int32_t device, int64_t C, int64_t F, int64_t n_world, const int64_t* off,
                              const uint32_t* obs_cam, const uint64_t* obs_world, const float* obs_uv_pixels,
                              const float* world_xyz, const uint8_t* cam_frozen, double huber_a, cc_rig** out) {
  cc::RigObsSource src;
  src.cam = obs_cam; src.world = obs_world; src.uv = obs_uv_pixels;
  return rig_create_impl(device, C, F, n_world, off, src, world_xyz, cam_frozen, huber_a, cc::RIG_K_PER_CAMERA, out);
}

// camera < 0: every set (the one shared set, or all per-camera sets)
static int rigk_set(cc_rig* h, int64_t camera, const double* intr9, uint32_t const_mask) {
  using namespace cc;
  if (!h || !intr9) return fail(CC_ERR_BAD_ARGUMENT, "cc_rigk_set_intrinsics: NULL argument");
  if (!h->d.kmode) return fail(CC_ERR_STATE, "cc_rigk_set_intrinsics: the handle was created without intrinsics (cc_rig_create)");
  const int CK = h->d.CK;
  if (camera >= 0 && (h->d.kmode != RIG_K_PER_CAMERA || camera >= CK))
    return fail(CC_ERR_BAD_ARGUMENT, "cc_rigk_set_camera_intrinsics: needs a per-camera handle and a camera id below %d", CK);
  CC_HIP(hipSetDevice(h->device));
  CC_HIP(hipStreamSynchronize(h->stream));
  double k16[16] = {0};
  for (int i = 0; i < 9; ++i) k16[i] = intr9[i];
  const uint32_t m = const_mask & 0x1ffu;
  for (int s = 0; s < CK; ++s) {
    if (camera >= 0 && s != camera) continue;
    CC_HIP(hipMemcpy(h->init_intr + (size_t)s * 16, k16, sizeof(k16), hipMemcpyHostToDevice));
    CC_HIP(hipMemcpy(h->d_kmask + s, &m, sizeof(m), hipMemcpyHostToDevice));
  }
  h->have_intr = true;
  return h->have_state ? cc_rig_reset(h) : CC_OK;
}

int cc_rigk_set_intrinsics(cc_rig* h, const double* intr9, uint32_t const_mask) { return rigk_set(h, -1, intr9, const_mask); }
int cc_rigk_set_camera_intrinsics(cc_rig* h, int64_t camera, const double* intr9, uint32_t const_mask) {
  if (camera < 0) return cc::fail(CC_ERR_BAD_ARGUMENT, "cc_rigk_set_camera_intrinsics: negative camera id");
  return rigk_set(h, camera, intr9, const_mask);
}

int cc_rigk_get_camera_intrinsics(cc_rig* h, int64_t camera, double* intr9) {
  using namespace cc;
  if (!h || !intr9) return fail(CC_ERR_BAD_ARGUMENT, "cc_rigk_get_intrinsics: NULL argument");
  if (!h->d.kmode) return fail(CC_ERR_STATE, "cc_rigk_get_intrinsics: the handle was created without intrinsics");
  const int64_t s = h->d.kmode == RIG_K_SHARED ? 0 : camera;
  if (s < 0 || s >= h->d.CK) return fail(CC_ERR_BAD_ARGUMENT, "cc_rigk_get_camera_intrinsics: camera id out of range");
  CC_HIP(hipSetDevice(h->device));
  LmCtl c;
  if (int rc = rig_read_ctl(h, &c)) return rc;
  double k16[16];
  CC_HIP(hipMemcpy(k16, h->d.intr + ((size_t)(c.cur & 1) * h->d.CK + (size_t)s) * 16, sizeof(k16), hipMemcpyDeviceToHost));
  for (int i = 0; i < 9; ++i) intr9[i] = k16[i];
  return CC_OK;
}
int cc_rigk_get_intrinsics(cc_rig* h, double* intr9) { return cc_rigk_get_camera_intrinsics(h, 0, intr9); }

void cc_rig_destroy(cc_rig* h) {
  if (h) cc::last_call_status_record(cc_rig_solver_form(h), h->form_reruns, h->form_note);   // (what a one-shot call's caller can still ask for)
  if (!h) return;
  cc::HostPhases hp("cc_rig_destroy");
  hipSetDevice(h->device);
  bool stream_ok = true;
  if (h->stream) stream_ok = hipStreamSynchronize(h->stream) == hipSuccess;
  if (h->stream2) {   // (back to the process's stream cache like h->stream: creating and destroying one per handle cost 1.3 ms of every one-shot call)
    if (hipStreamSynchronize(h->stream2) == hipSuccess) cc::stream_put(h->device, h->stream2);
    else hipStreamDestroy(h->stream2);
  }

  hp.mark("sync");
  if (h->ev_begin) hipEventDestroy(h->ev_begin);
  cc::rig_drop_graphs(h);
  for (auto e : h->events) hipEventDestroy(e);
  if (h->comm) cc::comm_destroy(h->comm);
  cc::mailbox_release(&h->mailbox);
  hp.mark("graphs_events");
  for (auto& a : h->allocs) cc::pool_free(h->device, a.first, a.second);
  hp.mark("free");
  {
    std::lock_guard<std::mutex> lk(cc::g_perm_mu);
    if (h->perm.size() > cc::g_perm_cache.size() && h->perm.size() <= ((size_t)1 << 28)) h->perm.swap(cc::g_perm_cache);
  }
  cc::pinned_block_put(h->pinned);
  if (stream_ok) cc::stream_put(h->device, h->stream);   // idle and reusable
  else if (h->stream) hipStreamDestroy(h->stream);      // never hand a failed stream to the next handle
  delete h;
}

int cc_rig_set_state(cc_rig* h, const double* cam_q, const double* cam_t, const double* frame_q, const double* frame_t) {
  using namespace cc;
  if (!h || !cam_q || !cam_t || !frame_q || !frame_t) return fail(CC_ERR_BAD_ARGUMENT, "cc_rig_set_state: NULL argument");
  CC_HIP(hipSetDevice(h->device));
  std::vector<double> cam((size_t)h->C * 8, 0.0), pose((size_t)h->F * 8, 0.0);
  for (int64_t c = 0; c < h->C; ++c) {
    for (int i = 0; i < 4; ++i) cam[c * 8 + i] = cam_q[c * 4 + i];
    for (int i = 0; i < 3; ++i) cam[c * 8 + 4 + i] = cam_t[c * 3 + i];
  }
  for (int64_t f = 0; f < h->F; ++f) {
    for (int i = 0; i < 4; ++i) pose[f * 8 + i] = frame_q[f * 4 + i];
    for (int i = 0; i < 3; ++i) pose[f * 8 + 4 + i] = frame_t[f * 3 + i];
  }
  CC_HIP(hipStreamSynchronize(h->stream));
  CC_HIP(hipMemcpy(h->init_cam, cam.data(), cam.size() * sizeof(double), hipMemcpyHostToDevice));
  CC_HIP(hipMemcpy(h->init_pose, pose.data(), pose.size() * sizeof(double), hipMemcpyHostToDevice));
  h->have_state = true;
  return cc_rig_reset(h);
}

int cc_rig_reset(cc_rig* h) {
  using namespace cc;
  if (!h || !h->have_state) return fail(CC_ERR_STATE, "cc_rig_reset: no state set");
  CC_HIP(hipSetDevice(h->device));
  LmCtl c{};
  if (int rc = rig_write_ctl(h, c)) return rc;
  h->last_st = c;
  h->st_known = true;
  CC_HIP(hipMemcpyAsync(h->d.cam, h->init_cam, (size_t)h->C * 8 * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  CC_HIP(hipMemcpyAsync(h->d.pose, h->init_pose, (size_t)h->F * 8 * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  if (h->d.kmode) CC_HIP(hipMemcpyAsync(h->d.intr, h->init_intr, (size_t)h->d.CK * 16 * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  return CC_OK;
}

}  // extern "C"

namespace cc {
// Captures the head of a solve (optional) and `rounds` rounds into an executable graph; on any failure the
// stream is taken out of capture mode again and nothing is kept.
static int rig_capture(cc_rig* h, bool first_chunk, int rounds, hipGraphExec_t* out) {
  hipGraph_t g = nullptr;
  CC_HIP(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
  int rc = 0;
  if (first_chunk) rig_enqueue_prep(h);
  for (int i = 0; i < rounds && !rc; ++i) rc = rig_enqueue_round(h, first_chunk && i == 0, false, i == rounds - 1);
  const hipError_t e_end = hipStreamEndCapture(h->stream, &g);
  if (rc || e_end != hipSuccess) {
    if (g) hipGraphDestroy(g);
    (void)hipGetLastError();
    return rc ? rc : fail(CC_ERR_HIP, "hipStreamEndCapture failed: %s", hipGetErrorString(e_end));
  }
  const hipError_t e_inst = hipGraphInstantiate(out, g, nullptr, nullptr, 0);
  hipGraphDestroy(g);
  if (e_inst != hipSuccess) { *out = nullptr; return fail(CC_ERR_HIP, "hipGraphInstantiate failed: %s", hipGetErrorString(e_inst)); }
  return 0;
}
}  // namespace cc

namespace cc {
// A solve in phases (cf. cc_intrinsics.hip): begin -> { launch a chunk -> wait } ... -> finish, so that one host
// thread can drive several handles (devices) in lock step.
struct RigRun {
  cc_options o;
  bool profile = false, use_graph = false;
  bool persist = false;   // this solve runs as ONE launch of k_rig_persist / k_rig_persist_w
  bool no_persist = false;   // (a rerun after that launch could not get its workgroups resident)
  bool rerun = false;        // set by rig_wait: the lean persistent launch gave up, nothing was written back
  int launched = 0;
  LmCtl st{};
  std::chrono::steady_clock::time_point t0;
};

static int rig_begin(cc_rig* h, const cc_options* opt, RigRun* r) {
  r->t0 = std::chrono::steady_clock::now();
  if (opt) r->o = *opt; else { cc_options_init(&r->o); r->o.max_iterations = 1000; }  // extrinsics_calibrator.cpp:211
  cc_options& o = r->o;
  if (o.check_interval < 1) o.check_interval = 1;
  if (o.max_iterations > h->d.log_cap - 1) o.max_iterations = h->d.log_cap - 1;
  r->profile = o.profile_kernels != 0;
  r->use_graph = o.use_graph != 0 && !h->comm && !r->profile;
  r->launched = 0;
  CC_HIP(hipSetDevice(h->device));
  // (the control block a solve starts from is the one the last solve published or the zeros of a reset: the host has it)
  LmCtl st = h->last_st;
  if (!h->st_known)
    if (int rc = rig_read_ctl(h, &st)) return rc;
  h->st_known = false;   // until this solve has ended
  if (st.cur & 1) {
    CC_HIP(hipMemcpyAsync(h->d.cam, h->d.cam + (size_t)h->C * 8, (size_t)h->C * 8 * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    CC_HIP(hipMemcpyAsync(h->d.pose, h->d.pose + (size_t)h->F * 8, (size_t)h->F * 8 * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    if (h->d.kmode) CC_HIP(hipMemcpyAsync(h->d.intr, h->d.intr + (size_t)h->d.CK * 16, (size_t)h->d.CK * 16 * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  }
  LmOpts lo;
  opts_from_public(o, &lo);
  if (!h->opts_valid || std::memcmp(&lo, &h->cached_opts, sizeof(lo)) != 0) {   // (uploaded when they change, not per solve)
    CC_HIP(hipMemcpyAsync(h->d.opts, &lo, sizeof(lo), hipMemcpyHostToDevice, h->stream));
    h->cached_opts = lo;
    h->opts_valid = true;
  }
  if (int rc = rig_size_reduce_grid(h)) return rc;
  if (h->stream2) CC_HIP(hipStreamSynchronize(h->stream2));   // (the control launch of a previous lean persistent solve: long over, but not on h->stream)
  // fresh control block (both copies) and synchronisation words -- counters, flag word, failure word: a failed solve may
  // have left any of them behind -- in ONE fill (they are one piece of memory, rig_create_impl)
  CC_HIP(hipMemsetAsync(h->d.ctl, 0, 2 * sizeof(LmCtl) + 16 * sizeof(unsigned), h->stream));
  for (auto e : h->events) hipEventDestroy(e);
  h->events.clear();
  h->event_kind.clear();
  h->event_round.clear();
  h->enq_round = 0;
  if (r->use_graph && h->graph_iters != o.check_interval) { rig_drop_graphs(h); h->graph_iters = o.check_interval; }
  return 0;
}

static int rig_launch(cc_rig* h, RigRun* r, int chunk) {
  CC_HIP(hipSetDevice(h->device));
  if (chunk == 0 && !r->no_persist && h->persist_w_ok && !r->profile && !h->comm && !h->exchange && !h->big && h->co_resident <= 1) {
    // the whole solve in one launch (k_rig_persist); the control workgroup publishes when it is over
    RigPersistDev q = h->pq;
    q.max_rounds = r->o.max_iterations + 2;
    q.timeout_shift = 27;   // 1.3 s of the 100 MHz wall clock
    // ... and 42 ms for the control workgroup to appear at all (its launch follows the workers' residency word by microseconds):
    // when another host thread sits in a call that waits for the whole device (hipFree, hipHostFree) while holding the
    // runtime's lock, THIS thread's control launch waits behind it and the device never goes idle under the spinning workers --
    // seen with three threads making one-shot calls at once: 1.3 s per collision before, the solve's rerun 42 ms late now
    q.first_shift = 22;
    if (h->p_epoch > 0x7fff0000u - (unsigned)q.max_rounds) {   // the 32-bit tags would wrap: start over on zeroed boxes
      CC_HIP(hipMemsetAsync(h->pq.sbox, 0, h->p_box_words * sizeof(unsigned long long), h->stream));   // (the claim word is its last)
      h->p_epoch = 0;
      h->host_pub[22] = 0ull;   // (no launch of this handle is in flight: rig_begin synchronised both streams' predecessors)
    }
    q.epoch0 = h->p_epoch;
    h->p_epoch += (unsigned)q.max_rounds + 2u;
    // lean workers (sixteen waves a compute unit) + the control workgroup as a launch of its own on a second stream, behind
    // everything rig_begin put on the first
    CC_HIP(hipMemcpyAsync(h->d_cam_backup, h->d.cam, (size_t)h->C * 8 * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    if (!h->stream2) { if (int rc = stream_get(h->device, &h->stream2)) return rc; }
    if (!h->ev_begin) CC_HIP(hipEventCreateWithFlags(&h->ev_begin, hipEventDisableTiming));
    CC_HIP(hipEventRecord(h->ev_begin, h->stream));
    CC_HIP(hipStreamWaitEvent(h->stream2, h->ev_begin, 0));
    static const bool drop_control = getenv("CC_RIG_PERSIST_TEST_NO_CONTROL") && atoi(getenv("CC_RIG_PERSIST_TEST_NO_CONTROL")) != 0;   // (test hook: the workers' first wait gives up)
    // Workers first; the control's candidates are launched when every worker is RESIDENT (the last worker to start stores the
    // solve's tag into a pinned word this thread spins on: ~10 us, once per solve), so that whichever candidate runs first sits
    // on a compute unit no worker needs. (First version of the round: hipStreamWaitValue32 on signal memory -- right, but a
    // command processor that finds the value not there yet goes to sleep: +154 us per solve at 250 workgroups of 1024
    // threads. Without any gate a candidate that starts BEFORE the workers can claim the last compute unit of an XCD that
    // 32 workers need: seen at once on the first box tried, CC_RIG_CTL_GATE=0.)
    static const bool use_gate = !(getenv("CC_RIG_CTL_GATE") && atoi(getenv("CC_RIG_CTL_GATE")) == 0);
    q.gate = use_gate ? const_cast<unsigned long long*>(h->host_pub) + 22 : nullptr;
    h->gated = q.gate != nullptr;
    const size_t lb = (size_t)rpw_lds_doubles(h->p_teams) * 8;
    if (h->p_teams == 1) hipLaunchKernelGGL(k_rig_persist_w<1>, dim3((unsigned)q.G), dim3(256), lb, h->stream, h->d, q);
    else if (h->p_teams == 2) hipLaunchKernelGGL(k_rig_persist_w<2>, dim3((unsigned)q.G), dim3(512), lb, h->stream, h->d, q);
    else hipLaunchKernelGGL(k_rig_persist_w<4>, dim3((unsigned)q.G), dim3(1024), lb, h->stream, h->d, q);
    CC_HIP(hipGetLastError());
    if (q.gate) {
      const auto tg = std::chrono::steady_clock::now();
      const unsigned long long want = (unsigned long long)(q.epoch0 + 1u);
      for (unsigned spins = 0; __atomic_load_n(const_cast<const unsigned long long*>(q.gate), __ATOMIC_ACQUIRE) != want; ++spins)
        if ((spins & 0x3ffu) == 0x3ffu && std::chrono::steady_clock::now() - tg > std::chrono::milliseconds(5)) break;   // (not all resident: the workers will give up and the solve is rerun)
    }
    if (!drop_control)
      hipLaunchKernelGGL(k_rig_persist_ctl, dim3((unsigned)kRigCtlCandidates), dim3(256), rig_persist_ctl_lds(h->solve_lds, h->d), h->stream2, h->d, q);
    CC_HIP(hipGetLastError());
    r->persist = true;
    r->launched += q.max_rounds;
    return 0;
  }
  // Rounds per chunk: check_interval, and twice that from the third chunk on -- a solve that has not ended after two looks
  // is a long one, and every look costs the host's round trip (publication seen + launch: 2 us per iteration at configs[4]
  // size with chunks of four, measured by varying check_interval) against the rounds that return at once after the solve
  // has ended (three kernels of ~4 us each per round, (n - 1) / 2 rounds on average).
  const int n = r->o.check_interval * (chunk >= 2 ? 2 : 1) + (chunk == 0 ? 1 : 0);
  if (r->use_graph) {
    const int which = chunk == 0 ? 0 : (chunk >= 2 ? 2 : 1);
    if (!h->graph[which])
      if (int rc = rig_capture(h, which == 0, n, &h->graph[which])) { rig_drop_graphs(h); return rc; }
    CC_HIP(hipGraphLaunch(h->graph[which], h->stream));
  } else {
    if (chunk == 0) rig_enqueue_prep(h);
    for (int i = 0; i < n; ++i)
      if (int rc = rig_enqueue_round(h, chunk == 0 && i == 0, r->profile, i == n - 1)) return rc;
    CC_HIP(hipGetLastError());
  }
  r->launched += n;
  return 0;
}

// After a wait inside the kernels gave up: this rank's position, in words -- launches of each kind started in this solve,
// the synchronisation words, and what its mailbox holds of every rank's posts. Every rank that gives up prints its own;
// together they name the stalled link (whose reduce launch never started, or whose post never arrived).
static std::string rig_describe_stall(cc_rig* h) {
  (void)hipStreamSynchronize(h->stream);   // (every later launch of the chunk returns at once: done / failure word)
  struct { LmCtl c; unsigned w[16]; } snap{};
  if (hipMemcpy(&snap, h->d.ctl_next, sizeof(snap), hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); return "state unreadable"; }
  static const char* names[RIG_PROG_COUNT] = {"sweep (not counted)", "stats", "init", "elim", "reduce", "solve", "update"};
  char buf[256];
  std::string out = "launches started this solve:";
  for (int k = 0; k < RIG_PROG_COUNT; ++k) { std::snprintf(buf, sizeof(buf), " %s %u", names[k], snap.w[4 + k]); out += buf; }
  std::snprintf(buf, sizeof(buf), "; %s form, reduce grid %d, co-resident %d; reduce blocks arrived %u, flag word epoch %u (done %u, step %u, cur %u), failure word %u; ",
                rig_unfused_exchange(h) ? "unfused (reduce / solve / update as three launches)" : "fused reduce + solve + update",
                h->reduce_blocks, rig_co_resident(h), snap.w[0], snap.w[1] >> 3, (snap.w[1] >> 2) & 1u, (snap.w[1] >> 1) & 1u, snap.w[1] & 1u, snap.w[3]);
  out += buf;
  if (h->exchange) out += mailbox_describe(&h->mailbox, h->d.rank, h->d.nranks);
  return out;
}

static int rig_wait(cc_rig* h, RigRun* r) {
  CC_HIP(hipSetDevice(h->device));
  bool wait_failed = false;
  const bool lean_run = r->persist && h->persist_w_ok && h->stream2 != nullptr;   // (what rig_launch put on two streams)
  if (int rc = ((h->comm || h->big) ? rig_read_ctl(h, &r->st, &wait_failed) : rig_wait_published(h, &r->st, &wait_failed, lean_run))) return rc;
  if (r->st.done) { h->last_st = r->st; h->st_known = true; if (r->persist && !wait_failed) h->lean_strikes = 0; }
  if (wait_failed && r->persist) {
    // The lean form keeps the starting point intact: cc_rig_solve runs the solve again, three kernels per iteration, and this
    // handle stays on that form. NOT silently (ADVICE / review of round 3): the demotion is counted and its reason kept for
    // cc_rig_solver_status -- launches started, where the control workgroup ran (or that it never did), the round reached.
    // (a give-up in the very first round -- the control never showed up within 42 ms -- demotes the handle only the second
    // time in a row: a host thread that lost its time slice between the two launches is not a property of the device)
    const bool first_round = r->st.iter == 0 && !r->st.done;
    if (!first_round || ++h->lean_strikes >= 2) h->persist_w_ok = false;
    h->form_reruns++;
    unsigned w[16] = {};
    (void)hipStreamSynchronize(h->stream);
    if (hipMemcpy(w, h->d.arrive, sizeof(w), hipMemcpyDeviceToHost) != hipSuccess) (void)hipGetLastError();
    char note[512];
    std::snprintf(note, sizeof(note), "lean persistent solve gave up in round %d after a %s wait (%d worker workgroups of %d threads + control): "
                  "%u workers had started, control workgroup %s (candidate %u, XCD %d)%s; the solve was run again with three kernels per "
                  "iteration and the handle %s (kernel-serialising tools, a CU mask, another tenant on the device or another host thread "
                  "in a call that waits for the device cause this)",
                  r->st.iter, first_round ? "42 ms" : "1.3 s", h->pq.G, h->p_teams * 256, w[13], w[12] ? "claimed a compute unit" : "NEVER RAN", w[12] >> 8,
                  (int)(w[12] & 0xffu) - 1, h->gated ? "" : ", control launched without waiting for the workers (CC_RIG_CTL_GATE=0)",
                  h->persist_w_ok ? "tries the lean form again next time" : "stays on that form");
    h->form_note = note;
    r->rerun = true;
    return 0;
  }
  if (wait_failed) {
    const std::string where = rig_describe_stall(h);
    return fail(CC_ERR_COMM, "k_rig_reduce: the solving block did not publish within 10 s (iteration %d): its launch was not fully "
                "resident (%d blocks; shards or processes sharing the device? CC_RIG_CO_RESIDENT) or a peer rank stalled. %s", r->st.iter, h->reduce_blocks, where.c_str());
  }
  if (r->st.done && r->st.term == CC_FAILURE_EXCHANGE) {
    const std::string where = rig_describe_stall(h);
    return fail(CC_ERR_COMM, "mailbox exchange timed out: a peer rank did not post within 10 s (iteration %d). %s", r->st.iter, where.c_str());
  }
  if (!r->st.done && r->launched > r->o.max_iterations + 4 * r->o.check_interval + 2)
    return fail(CC_ERR_STATE, "rig LM loop did not terminate (iter=%d)", r->st.iter);
  return 0;
}

static int rig_finish(cc_rig* h, RigRun* r, cc_summary* summary) {
  const LmCtl& st = r->st;
  CC_HIP(hipSetDevice(h->device));
  if (summary) {
    cc_iteration* user_log = summary->log;
    const int cap = summary->log_capacity;
    summary->iterations = st.iter;
    summary->successful_steps = st.n_success;
    summary->termination = st.term;
    summary->initial_cost = st.initial_cost;
    summary->final_cost = st.x_cost;
    summary->sweeps = st.sweeps;
    const int n = user_log ? std::min(std::min(st.log_len, cap), h->d.log_cap) : 0;
    summary->log_len = n;
    if (n > 0) CC_HIP(hipMemcpy(user_log, h->d.log, (size_t)n * sizeof(cc_iteration), hipMemcpyDeviceToHost));
    summarise_probes(h->events, r->profile ? h->event_kind : std::vector<int>(), h->event_round, st.iter, summary,
                     [](float* ms, hipEvent_t a, hipEvent_t b) { return hipEventElapsedTime(ms, a, b) == hipSuccess; });
    summary->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - r->t0).count();
  }
  for (auto e : h->events) hipEventDestroy(e);
  h->events.clear();
  h->event_kind.clear();
  h->event_round.clear();
  return CC_OK;
}
}  // namespace cc

extern "C" {

int cc_rig_solve(cc_rig* h, const cc_options* opt, cc_summary* summary) {
  using namespace cc;
  if (!h || !h->have_state) return fail(CC_ERR_STATE, "cc_rig_solve: no state set");
  if (h->d.kmode && !h->have_intr) return fail(CC_ERR_STATE, "cc_rig_solve: cc_rigk_set_intrinsics has not been called");
  RigRun r;
  HostPhases hp("cc_rig_solve");
  if (int rc = rig_begin(h, opt, &r)) return rc;
  hp.mark("begin");
  // (the lean form is two launches that wait for each other inside their kernels: one such solve at a time per device and
  // process -- persist_mutex, cc_common.hpp; a second host thread waits here instead of inside a kernel for 1.3 s)
  std::unique_lock<std::mutex> lean_lock(persist_mutex(h->device), std::defer_lock);
  for (int chunk = 0;; ++chunk) {
    if (chunk == 0 && !r.no_persist && h->persist_w_ok && !r.profile && !h->comm && !h->exchange && !h->big && h->co_resident <= 1)
      lean_lock.lock();
    if (int rc = rig_launch(h, &r, chunk)) return rc;
    if (chunk == 0) hp.mark("launch0");
    if (int rc = rig_wait(h, &r)) return rc;
    // (a lean solve that gave up: its control launch on the second stream may still be queued or spinning -- the lock is kept
    // until both streams have drained below, or another thread's lean solve would start next to that late control workgroup)
    if (lean_lock.owns_lock() && !r.rerun) lean_lock.unlock();
    if (chunk == 0) hp.mark("wait0");
    if (r.rerun) {
      // The lean persistent launch could not get every workgroup resident (a device shared with another process, or the
      // control launch not scheduled next to the workers): a wait inside it gave up after 1.3 s. Frame poses go back to
      // global memory only at the end of a solve that did not fail, and the cameras of the starting point were put
      // aside: restore them and run the solve again, three kernels per iteration (no co-residency needed).
      const hipError_t e_s1 = hipStreamSynchronize(h->stream);
      const hipError_t e_s2 = h->stream2 ? hipStreamSynchronize(h->stream2) : hipSuccess;
      if (lean_lock.owns_lock()) lean_lock.unlock();
      CC_HIP(e_s1);
      CC_HIP(e_s2);
      CC_HIP(hipMemcpyAsync(h->d.cam, h->d_cam_backup, (size_t)h->C * 8 * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
      h->last_st = LmCtl{};   // (buffer 0 holds the starting point)
      h->st_known = true;
      h->pub_count = __atomic_load_n(const_cast<const unsigned long long*>(h->host_pub), __ATOMIC_ACQUIRE);
      r = RigRun{};
      r.no_persist = true;
      if (int rc = rig_begin(h, opt, &r)) return rc;
      r.no_persist = true;
      chunk = -1;
      continue;
    }
    if (r.st.done) break;
  }
  hp.mark("chunks");
  const int rc_fin = rig_finish(h, &r, summary);
  hp.mark("finish");
  return rc_fin;
}

int cc_rig_solver_form(cc_rig* h) {
  using namespace cc;
  if (!h) return fail(CC_ERR_BAD_ARGUMENT, "cc_rig_solver_form: NULL handle");
  if (!(h->persist_w_ok && !h->comm && !h->exchange && !h->big && h->co_resident <= 1)) return 0;
  return 2;
}

int cc_rig_solver_status(cc_rig* h, int32_t* form, int32_t* reruns, char* note, int32_t note_capacity) {
  using namespace cc;
  if (!h) return fail(CC_ERR_BAD_ARGUMENT, "cc_rig_solver_status: NULL handle");
  if (form) *form = cc_rig_solver_form(h);
  if (reruns) *reruns = h->form_reruns;
  if (note && note_capacity > 0) std::snprintf(note, (size_t)note_capacity, "%s", h->form_note.c_str());
  return CC_OK;
}

int cc_rig_get_state(cc_rig* h, double* cam_q, double* cam_t, double* frame_q, double* frame_t, double* obs_cost) {
  using namespace cc;
  if (!h) return fail(CC_ERR_BAD_ARGUMENT, "cc_rig_get_state: NULL handle");
  CC_HIP(hipSetDevice(h->device));
  LmCtl c;
  if (int rc = rig_read_ctl(h, &c)) return rc;
  const int cur = c.cur & 1;
  if (cam_q || cam_t) {
    std::vector<double> cam((size_t)h->C * 8);
    CC_HIP(hipMemcpy(cam.data(), h->d.cam + (size_t)cur * h->C * 8, cam.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (int64_t k = 0; k < h->C; ++k) {
      if (cam_q) for (int i = 0; i < 4; ++i) cam_q[k * 4 + i] = cam[k * 8 + i];
      if (cam_t) for (int i = 0; i < 3; ++i) cam_t[k * 3 + i] = cam[k * 8 + 4 + i];
    }
  }
  if (frame_q || frame_t) {
    std::vector<double> pose((size_t)h->F * 8);
    CC_HIP(hipMemcpy(pose.data(), h->d.pose + (size_t)cur * h->F * 8, pose.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (int64_t f = 0; f < h->F; ++f) {
      if (frame_q) for (int i = 0; i < 4; ++i) frame_q[f * 4 + i] = pose[f * 8 + i];
      if (frame_t) for (int i = 0; i < 3; ++i) frame_t[f * 3 + i] = pose[f * 8 + 4 + i];
    }
  }
  if (obs_cost && h->N > 0) {
    hipLaunchKernelGGL(k_rig_obs_cost, dim3((unsigned)h->NG), dim3(256), 0, h->stream, h->d, cur, h->d_cost);
    CC_HIP(hipGetLastError());
    // through the cached pinned block (full PCIe rate, no fresh 8N-byte vector to fault in), back into the caller's order on
    // several host threads
    bool st_cached = false;
    double* sorted = static_cast<double*>(staging_get((size_t)h->N * sizeof(double), &st_cached));
    if (!sorted) return fail(CC_ERR_HIP, "cc_rig_get_state: pinned staging memory could not be allocated");
    struct StGuard { void* p; hipStream_t s; ~StGuard() { (void)hipStreamSynchronize(s); staging_put(p); } } stg{sorted, h->stream};
    CC_HIP(hipMemcpyAsync(sorted, h->d_cost, (size_t)h->N * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    CC_HIP(hipStreamSynchronize(h->stream));
    const int64_t* perm = h->perm.data();
    parallel_ranges(h->N, (int64_t)1 << 15, [&](int, int64_t a, int64_t b) {
      if (h->perm_inverse) { for (int64_t i = a; i < b; ++i) obs_cost[i] = sorted[perm[i]]; }
      else { for (int64_t i = a; i < b; ++i) obs_cost[perm[i]] = sorted[i]; }
    });
  }
  return CC_OK;
}

int cc_rig_eval(cc_rig* h, double* cost) {
  using namespace cc;
  if (!h || !h->have_state || !cost) return fail(CC_ERR_STATE, "cc_rig_eval: no state set");
  if (h->d.kmode && !h->have_intr) return fail(CC_ERR_STATE, "cc_rig_eval: cc_rigk_set_intrinsics has not been called");
  std::vector<double> oc((size_t)h->N);
  if (int rc = cc_rig_get_state(h, nullptr, nullptr, nullptr, nullptr, oc.data())) return rc;
  double c = 0.0;
  for (double v : oc) c += v;
  *cost = c;
  return CC_OK;
}

}  // extern "C"

namespace cc {
// multi-GPU attach, common part: the column layout must be the same on every rank, so it is rebuilt for the
// cameras that ANY rank observes (flags[c] > 0) when that differs from what this rank built at create time
static int rig_adopt_global_cameras(cc_rig* h, const std::vector<double>& flags) {
  std::vector<uint8_t> any((size_t)h->C);
  for (int64_t c = 0; c < h->C; ++c) any[(size_t)c] = flags[(size_t)c] > 0.0 ? 1 : 0;
  if (any == h->seen_any) return 0;
  CC_HIP(hipStreamSynchronize(h->stream));
  return rig_layout(h, any);
}
}  // namespace cc

extern "C" {

int cc_rig_comm_init(cc_rig* h, const uint8_t id[128], int32_t rank, int32_t nranks) {
  using namespace cc;
  if (!h || !id || rank < 0 || nranks < 1 || rank >= nranks || nranks > 32)
    return fail(CC_ERR_BAD_ARGUMENT, "cc_rig_comm_init: bad arguments (nranks must be 1..32)");
  CC_HIP(hipSetDevice(h->device));
  if (h->comm) { comm_destroy(h->comm); h->comm = nullptr; }
  rig_drop_graphs(h);
  if (int rc = comm_create(id, rank, nranks, &h->comm)) return rc;
  h->d.comm = 1; h->d.rank = rank; h->d.nranks = nranks;
  // a camera is part of the problem if ANY rank observes it: sum the per-rank "seen" flags
  std::vector<double> flags((size_t)h->C, 0.0);
  for (int64_t c = 0; c < h->C; ++c) flags[(size_t)c] = h->seen[(size_t)c] ? 1.0 : 0.0;
  double* d_flags = nullptr;
  CC_HIP(hipMalloc(&d_flags, flags.size() * sizeof(double)));
  hipError_t e1 = hipMemcpyAsync(d_flags, flags.data(), flags.size() * sizeof(double), hipMemcpyHostToDevice, h->stream);
  int rc = e1 == hipSuccess ? comm_allreduce_sum(h->comm, d_flags, (int)h->C, h->stream) : 0;
  hipError_t e2 = hipMemcpyAsync(flags.data(), d_flags, flags.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream);
  hipError_t e3 = hipStreamSynchronize(h->stream);
  hipFree(d_flags);
  CC_HIP(e1); CC_HIP(e2); CC_HIP(e3);
  if (rc) return rc;
  return rig_adopt_global_cameras(h, flags);
}

int cc_rig_exchange_export(cc_rig* h, uint8_t handle[64]) {
  using namespace cc;
  if (!h || !handle) return fail(CC_ERR_BAD_ARGUMENT, "cc_rig_exchange_export: NULL argument");
  CC_HIP(hipSetDevice(h->device));
  CC_HIP(hipStreamSynchronize(h->stream));
  rig_drop_graphs(h);
  mailbox_release(&h->mailbox);
  h->d.x = P2pDev{};
  h->exchange = false;
  int k0 = 0, k1 = 0;
  rig_exchange_bounds(h, &k0, &k1);
  return mailbox_export(&h->mailbox, k0, k1, handle);
}

// collective: every rank must call it (the "camera seen" flags are summed through the mailboxes)
int cc_rig_exchange_attach(cc_rig* h, int32_t rank, int32_t nranks, const uint8_t* handles) {
  using namespace cc;
  if (!h || !handles || rank < 0 || nranks < 1 || rank >= nranks || nranks > kP2pMaxRanks)
    return fail(CC_ERR_BAD_ARGUMENT, "cc_rig_exchange_attach: bad arguments (nranks must be 1..%d)", kP2pMaxRanks);
  if (!h->mailbox.local) return fail(CC_ERR_STATE, "cc_rig_exchange_attach: call cc_rig_exchange_export first");
  if (h->comm) return fail(CC_ERR_STATE, "cc_rig_exchange_attach: an RCCL communicator is already attached");
  if (h->C > 128) return fail(CC_ERR_BAD_ARGUMENT, "cc_rig_exchange_attach: at most 128 cameras with the mailbox exchange");
  CC_HIP(hipSetDevice(h->device));
  rig_drop_graphs(h);
  if (int rc = mailbox_attach(&h->mailbox, rank, nranks, handles, &h->d.x)) return rc;
  h->d.comm = 1; h->d.rank = rank; h->d.nranks = nranks;
  h->exchange = true;
  {
    // ranks that share this device (one process per GPU: 1; the one-GPU test box: all of them). The handles do not say
    // where the peers live, so the count is the pigeonhole bound over the devices this process can see.
    int ndev = 1;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) ndev = 1;
    h->co_resident = (nranks + ndev - 1) / ndev;
  }
  // a camera is part of the problem if ANY rank observes it: sum the per-rank "seen" flags
  std::vector<double> flags(128, 0.0);
  for (int64_t c = 0; c < h->C; ++c) flags[(size_t)c] = h->seen[(size_t)c] ? 1.0 : 0.0;
  double* d_io = nullptr;
  int* d_ok = nullptr;
  CC_HIP(hipMalloc(&d_io, 256 * sizeof(double)));
  CC_HIP(hipMalloc(&d_ok, sizeof(int)));
  CC_HIP(hipMemcpyAsync(d_io, flags.data(), 128 * sizeof(double), hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(k_rig_flag_exchange, dim3(1), dim3(128), 0, h->stream, h->d, d_io, d_io + 128, (int)h->C, d_ok);
  int ok = 0;
  hipError_t e1 = hipMemcpyAsync(flags.data(), d_io + 128, 128 * sizeof(double), hipMemcpyDeviceToHost, h->stream);
  hipError_t e2 = hipMemcpyAsync(&ok, d_ok, sizeof(int), hipMemcpyDeviceToHost, h->stream);
  hipError_t e3 = hipStreamSynchronize(h->stream);
  hipFree(d_io);
  hipFree(d_ok);
  CC_HIP(e1); CC_HIP(e2); CC_HIP(e3);
  if (!ok) return fail(CC_ERR_COMM, "cc_rig_exchange_attach: a peer rank did not attach within 10 s");
  return rig_adopt_global_cameras(h, flags);
}

#ifdef CC_RIG_TIMING
// timing-only builds: out[0..2] = ticks (100 MHz) and shader cycles per factorisation of a random SPD S x S system, checksum
int cc_rig_debug_chol_bench(int32_t S, int32_t reps, int32_t which, double* out) {
  using namespace cc;
  const int LD = (S + 1) | 1;
  std::vector<double> A((size_t)(S + 1) * LD + 256, 0.0);
  for (int i = 0; i < S; ++i) {
    for (int j = 0; j <= i; ++j) A[(size_t)i * LD + j] = i == j ? S + 1.0 : 1.0 / (1 + i + j);
    A[(size_t)S * LD + i] = 1.0 + i;
  }
  double *dA = nullptr, *dout = nullptr;
  CC_HIP(hipMalloc(&dA, A.size() * 8));
  CC_HIP(hipMalloc(&dout, 64));
  CC_HIP(hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice));
  const size_t lds = ((size_t)(S + 1) * LD + 5 * 128) * 8;
  CC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_chol_bench), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k_chol_bench, dim3(1), dim3(256), lds, 0, dA, S, reps, which, dout);
  CC_HIP(hipDeviceSynchronize());
  CC_HIP(hipMemcpy(out, dout, 24, hipMemcpyDeviceToHost));
  hipFree(dA); hipFree(dout);
  return CC_OK;
}
#endif

// Debug/test aid (not declared in the public header): copies a named device buffer to the host.
int cc_rig_debug_fetch(cc_rig* h, const char* name, double* out, int64_t n) {
  using namespace cc;
  if (!h || !name || !out) return fail(CC_ERR_BAD_ARGUMENT, "cc_rig_debug_fetch: bad arguments");
  CC_HIP(hipSetDevice(h->device));
  CC_HIP(hipStreamSynchronize(h->stream));
  const std::string k(name);
  const double* src = nullptr;
  const RigDev& d = h->d;
  if (k == "ss") src = d.ss; else if (k == "sp") src = d.sp; else if (k == "ds") src = d.ds;
  else if (k == "Y") src = d.Y; else if (k == "partial") src = d.partial; else if (k == "gblocks") src = d.gblocks;
  else if (k == "gcomp") src = d.gcomp; else if (k == "fsum") src = d.fsum;
  else if (k == "camrec") src = d.camrec; else if (k == "frec") src = d.frec; else if (k == "gstats") src = d.gstats;
  else if (k == "fstats") src = d.fstats; else if (k == "shared_stats") src = d.shared_stats; else if (k == "cam") src = d.cam; else if (k == "pose") src = d.pose;
  else if (k == "vec") src = d.vec; else if (k == "vec_stats") src = d.vec_stats;
  else return fail(CC_ERR_BAD_ARGUMENT, "cc_rig_debug_fetch: unknown buffer %s", name);
  CC_HIP(hipMemcpy(out, src, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
  return CC_OK;
}

int cc_rig_optimize(const cc_options* opt, int32_t device, int64_t C, int64_t F, int64_t n_world,
                    const int64_t* off, const uint32_t* obs_cam, const uint64_t* obs_world,
                    const float* obs_uv, const float* world_xyz, double* cam_q, double* cam_t,
                    const uint8_t* cam_frozen, double* frame_q, double* frame_t, double huber_a,
                    double* obs_cost, cc_summary* summary) {
  cc::last_call_status_reset();
  cc_rig* h = nullptr;
  cc::HostPhases hp("cc_rig_optimize");
  int rc = cc_rig_create(device, C, F, n_world, off, obs_cam, obs_world, obs_uv, world_xyz, cam_frozen, huber_a, &h);
  if (rc) return rc;
  hp.mark("create");
  rc = cc_rig_set_state(h, cam_q, cam_t, frame_q, frame_t);
  hp.mark("set_state");
  cc_options o;
  if (opt) o = *opt; else { cc_options_init(&o); o.max_iterations = 1000; }  // extrinsics_calibrator.cpp:211
  o.use_graph = 0;   // one solve per handle: capturing and instantiating a graph cannot pay off
  if (!rc) rc = cc_rig_solve(h, &o, summary);
  hp.mark("solve");
  if (!rc) rc = cc_rig_get_state(h, cam_q, cam_t, frame_q, frame_t, obs_cost);
  hp.mark("get_state");
  cc_rig_destroy(h);
  hp.mark("destroy");
  return rc;
}


// Multi-device rig solve driven by ONE host thread (SURVEY.md 8(b) thread model; cf. cc_intrinsics_optimize_multi):
// frames sharded contiguously by observation count, cameras and world points replicated, mailboxes wired inside the
// process, every device's chunk enqueued before any is waited for. A device id may appear several times.
int cc_rig_optimize_frames(const cc_options* opt, int32_t device, int64_t C, int64_t F, int64_t n_world,
                           void* const* frame_records, const int64_t* counts, const cc_obs_layout* layout,
                           const float* world_xyz, double* cam_q, double* cam_t, const uint8_t* cam_frozen,
                           double* frame_q, double* frame_t, double huber_a, cc_summary* summary) {
  cc::last_call_status_reset();
  using namespace cc;
  if (F < 1 || !frame_records || !counts || !layout) return fail(CC_ERR_BAD_ARGUMENT, "cc_rig_optimize_frames: bad arguments");
  HostPhases hp("cc_rig_optimize_frames");
  std::vector<int64_t> off((size_t)F + 1, 0);
  for (int64_t f = 0; f < F; ++f) {
    if (counts[f] < 0) return fail(CC_ERR_BAD_ARGUMENT, "cc_rig_optimize_frames: negative count");
    off[(size_t)f + 1] = off[(size_t)f] + counts[f];
  }
  RigObsSource src;
  src.frames = const_cast<const void* const*>(frame_records);
  src.lay = *layout;
  cc_rig* h = nullptr;
  int rc = rig_create_impl(device, C, F, n_world, off.data(), src, world_xyz, cam_frozen, huber_a, RIG_K_NONE, &h);
  if (rc) return rc;
  hp.mark("create");
  rc = cc_rig_set_state(h, cam_q, cam_t, frame_q, frame_t);
  cc_options o;
  if (opt) o = *opt; else { cc_options_init(&o); o.max_iterations = 1000; }  // extrinsics_calibrator.cpp:211
  o.use_graph = 0;
  if (!rc) rc = cc_rig_solve(h, &o, summary);
  hp.mark("solve");
  if (!rc) rc = cc_rig_get_state(h, cam_q, cam_t, frame_q, frame_t, nullptr);
  if (!rc && layout->cost_offset >= 0 && h->N > 0) {
    // per-observation costs (extrinsics_calibrator.cpp:219-225) through the pinned block, then into the caller's records on
    // several host threads (a frame's observations stay inside the frame's range of the regrouped order)
    rc = [&]() -> int {
      LmCtl c;
      if (int r2 = rig_read_ctl(h, &c)) return r2;
      hipLaunchKernelGGL(k_rig_obs_cost, dim3((unsigned)h->NG), dim3(256), 0, h->stream, h->d, c.cur & 1, h->d_cost);
      CC_HIP(hipGetLastError());
      bool st_cached = false;
      double* sorted = static_cast<double*>(staging_get((size_t)h->N * sizeof(double), &st_cached));
      if (!sorted) return fail(CC_ERR_HIP, "cc_rig_optimize_frames: pinned staging memory could not be allocated");
      struct StGuard { void* p; hipStream_t s; ~StGuard() { (void)hipStreamSynchronize(s); staging_put(p); } } stg{sorted, h->stream};
      const int64_t* perm = h->perm.data();
      const int parts = parallel_parts(h->N, (int64_t)1 << 15);
      std::vector<int64_t> pf((size_t)parts + 1, 0);
      if (int r2 = cc_partition_frames(F, off.data(), parts, pf.data())) return r2;
      // one transfer and one event per thread's range of frames: a thread writes its range into the records as soon as it has
      // arrived, under the transfers of the ranges behind it
      std::vector<hipEvent_t> ev((size_t)parts, nullptr);
      struct EvGuard { std::vector<hipEvent_t>& e; ~EvGuard() { for (auto x : e) if (x) (void)hipEventDestroy(x); } } evg{ev};
      for (int t = 0; t < parts; ++t) {
        const int64_t a = off[(size_t)pf[(size_t)t]], n = off[(size_t)pf[(size_t)t + 1]] - a;
        if (n > 0) CC_HIP(hipMemcpyAsync(sorted + a, h->d_cost + a, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        CC_HIP(hipEventCreateWithFlags(&ev[(size_t)t], hipEventDisableTiming));
        CC_HIP(hipEventRecord(ev[(size_t)t], h->stream));
      }
      const int64_t stride = layout->stride, oc = layout->cost_offset;
      std::vector<char> bad((size_t)parts, 0);
      parallel_tasks(parts, [&](int t) {
        if (hipSetDevice(h->device) != hipSuccess || hipEventSynchronize(ev[(size_t)t]) != hipSuccess) { bad[(size_t)t] = 1; return; }
        // record after record (the inverse permutation: sequential stores, the reads stay inside the frame's range of `sorted`)
        for (int64_t f = pf[(size_t)t]; f < pf[(size_t)t + 1]; ++f) {
          unsigned char* rec = static_cast<unsigned char*>(frame_records[f]) + oc;
          for (int64_t k = off[(size_t)f]; k < off[(size_t)f + 1]; ++k, rec += stride) std::memcpy(rec, &sorted[perm[k]], sizeof(double));
        }
      });
      CC_HIP(hipStreamSynchronize(h->stream));
      for (char b : bad) if (b) return fail(CC_ERR_HIP, "cc_rig_optimize_frames: the read-back of the costs failed");
      return CC_OK;
    }();
  }
  hp.mark("get_state");
  cc_rig_destroy(h);
  hp.mark("destroy");
  return rc;
}

int cc_rig_optimize_multi(const cc_options* opt, int32_t n_devices, const int32_t* devices, int64_t C, int64_t F,
                          int64_t n_world, const int64_t* off, const uint32_t* obs_cam, const uint64_t* obs_world,
                          const float* obs_uv, const float* world_xyz, double* cam_q, double* cam_t,
                          const uint8_t* cam_frozen, double* frame_q, double* frame_t, double huber_a,
                          double* obs_cost, cc_summary* summary) {
  cc::last_call_status_reset();
  using namespace cc;
  if (n_devices < 1 || !devices || n_devices > kP2pMaxRanks)
    return fail(CC_ERR_BAD_ARGUMENT, "cc_rig_optimize_multi: 1..%d devices", kP2pMaxRanks);
  if (!off || F < 1 || !cam_q || !cam_t || !frame_q || !frame_t) return fail(CC_ERR_BAD_ARGUMENT, "cc_rig_optimize_multi: bad arguments");
  const int n = (int)std::min<int64_t>(n_devices, F);
  if (n == 1)
    return cc_rig_optimize(opt, devices[0], C, F, n_world, off, obs_cam, obs_world, obs_uv, world_xyz, cam_q, cam_t, cam_frozen,
                           frame_q, frame_t, huber_a, obs_cost, summary);
  std::vector<int64_t> first((size_t)n + 1);
  if (int rc = cc_partition_frames(F, off, n, first.data())) return rc;
  std::vector<cc_rig*> hs((size_t)n, nullptr);
  int rc = 0;
  for (int r = 0; r < n && !rc; ++r) {
    const int64_t f0 = first[(size_t)r], f1 = first[(size_t)r + 1], o0 = off[f0];
    std::vector<int64_t> so((size_t)(f1 - f0) + 1);
    for (int64_t f = f0; f <= f1; ++f) so[(size_t)(f - f0)] = off[f] - o0;
    rc = cc_rig_create(devices[r], C, f1 - f0, n_world, so.data(), obs_cam ? obs_cam + o0 : nullptr, obs_world ? obs_world + o0 : nullptr,
                       obs_uv ? obs_uv + 2 * o0 : nullptr, world_xyz, cam_frozen, huber_a, &hs[(size_t)r]);
  }
  if (!rc) {
    // a camera is part of the problem if ANY shard observes it (all shards live in this process: no exchange needed)
    std::vector<double> flags((size_t)C, 0.0);
    for (auto* h : hs) for (int64_t c = 0; c < C; ++c) if (h->seen[(size_t)c]) flags[(size_t)c] = 1.0;
    std::vector<Mailbox*> boxes;
    std::vector<int> devs;
    for (auto* h : hs) { boxes.push_back(&h->mailbox); devs.push_back(h->device); }
    for (int r = 0; r < n && !rc; ++r) {
      cc_rig* h = hs[(size_t)r];
      int k0 = 0, k1 = 0;
      rig_exchange_bounds(h, &k0, &k1);
      rc = hipSetDevice(h->device) == hipSuccess ? mailbox_alloc(&h->mailbox, k0, k1) : fail(CC_ERR_HIP, "hipSetDevice failed");
    }
    for (int r = 0; r < n && !rc; ++r) {
      cc_rig* h = hs[(size_t)r];
      rc = mailbox_wire_local(&h->mailbox, r, n, boxes.data(), devs.data(), &h->d.x);
      if (!rc) {
        h->d.comm = 1; h->d.rank = r; h->d.nranks = n; h->exchange = true;
        h->co_resident = (int)std::count(devs.begin(), devs.end(), h->device);   // shards of this solve on the same device
        rc = rig_adopt_global_cameras(h, flags);
      }
    }
  }
  for (int r = 0; r < n && !rc; ++r) {
    const int64_t f0 = first[(size_t)r];
    rc = cc_rig_set_state(hs[(size_t)r], cam_q, cam_t, frame_q + 4 * f0, frame_t + 3 * f0);
  }
  cc_options o;
  if (opt) o = *opt; else { cc_options_init(&o); o.max_iterations = 1000; }  // extrinsics_calibrator.cpp:211
  o.use_graph = 0;
  std::vector<RigRun> runs((size_t)n);
  for (int r = 0; r < n && !rc; ++r) rc = rig_begin(hs[(size_t)r], &o, &runs[(size_t)r]);
  for (int chunk = 0; !rc; ++chunk) {
    for (int r = 0; r < n && !rc; ++r) rc = rig_launch(hs[(size_t)r], &runs[(size_t)r], chunk);
    for (int r = 0; r < n && !rc; ++r) rc = rig_wait(hs[(size_t)r], &runs[(size_t)r]);
    if (rc) break;
    bool all_done = true, any_done = false;
    for (auto& run : runs) { all_done = all_done && run.st.done; any_done = any_done || run.st.done; }
    if (all_done) break;
    if (any_done) rc = fail(CC_ERR_STATE, "cc_rig_optimize_multi: the shards disagree about termination");
  }
  if (!rc) rc = rig_finish(hs[0], &runs[0], summary);
  for (int r = 0; r < n && !rc; ++r) {
    const int64_t f0 = first[(size_t)r];
    rc = cc_rig_get_state(hs[(size_t)r], r == 0 ? cam_q : nullptr, r == 0 ? cam_t : nullptr, frame_q + 4 * f0, frame_t + 3 * f0,
                          obs_cost ? obs_cost + off[f0] : nullptr);
  }
  const std::string err = rc ? last_error() : std::string();
  for (auto* h : hs) if (h) { hipSetDevice(h->device); hipStreamSynchronize(h->stream); }   // nobody frees a mailbox a peer still writes
  for (auto* h : hs) cc_rig_destroy(h);
  if (rc) last_error() = err;
  return rc;
}

}  // extern "C"
