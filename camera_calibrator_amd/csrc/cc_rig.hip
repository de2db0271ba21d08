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
// 8 x 2000 x 500. (The library calls are kept as a variant for parity forensics: scripts/variants/exact_arith.patch.)
__device__ __forceinline__ void huber_outlier(double a, double s, double& r, double& sr) {
  const double y = rsqrt_pos(s);
  r = s * y;
  const double q = fmax(2.2250738585072014e-308, a * y);   // (a = 0: the same tiny weight the guarded divide gives)
  sr = q * rsqrt_pos(q);
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

#include "cc_rig_sweeps.hpp"
#include "cc_rig_steps.hpp"
#include "cc_rig_big.hpp"
#include "cc_rig_lean.hpp"


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
  bool perm_inverse = false;   // columns: the storage holds int32 entries, [k] = where the caller's observation k went, counted from its frame's first position -- instead of perm[i] = caller's index of position i
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
  bool persist_lean_forced = false;   // CC_RIG_PERSIST=1: the lean form wherever it FITS, also where the three kernels measured faster (tests)
  double* d_cam_backup = nullptr;   // [C][8] cameras of the starting point (the lean persistent solve may be run again in the three-kernel form)
  int p_teams = 4;              // frames per workgroup of the lean form
  bool persist_w_ok = false;    // ... and so can its lean form (k_rig_persist_w + k_rig_persist_ctl: <= 4 observed cameras, <= 24 shared coordinates)
  int ran_form = -1;               // the form the handle's LAST solve ran in to its end (-1: none yet): 0 three kernels per iteration, 2 the lean persistent pair
  int form_reruns = 0;             // lean persistent solves that gave up and were run again in the three-kernel form
  int lean_strikes = 0;            // ... of them in the first round, in a row (two demote the handle)
  std::string form_note;           // why (cc_rig_solver_status)
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
  if (const char* e = getenv("CC_RIG_PERSIST")) { h->persist_lean_allowed = atoi(e) != 0; h->persist_lean_forced = atoi(e) != 0; }
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
      // Where it FITS is not where it PAYS (round 6, profiles/r06/lean_vs_three_kernel_grid.txt: 2 and 4 cameras x 256 .. 1020
      // frames x 4 .. 500 points, both forms on one box). With one or two frames per workgroup (<= 510 frames) the lean form
      // takes 10 - 30 % less time per iteration at every shape measured; with FOUR (up to 1020 frames: sixteen waves a compute
      // unit, 128 registers, every round as long as its slowest of four frames) it ties at best and loses up to 30 % wherever
      // the solve rejects steps -- 4 x 1020 x 300: 63.4 against 50.4 us per iteration -- except for rigs of two observed cameras
      // with a handful of points per frame (the reference's own test, 2 x 1000 x 4: 34.8 against 40.2). CC_RIG_PERSIST=1 keeps
      // the old rule (wherever it fits) for the tests of the four-frame workers.
      const bool pays = h->p_teams <= 2 || (CO <= 2 && h->N <= 32 * F);
      if (!pays && !h->persist_lean_forced) h->persist_w_ok = false;
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
// (2000 frames: 124.4 us per iteration with 128, 127.2 with 64).
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
  constexpr int rcap = 128;
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
    if (d.kcm) hipLaunchKernelGGL(k_rig_sweep_k2, dim3((unsigned)h->NG), dim3(128), 0, h->stream, d);   // (a workgroup of two waves per group)
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
  const uint32_t* cam = nullptr; const uint64_t* world = nullptr; const float* uv = nullptr;   // flat arrays, or
  const cc_obs_columns* cols = nullptr;                                                         // the caller's per-frame columns, read in place
};
struct RigFlatFrame {
  const uint32_t* cam; const uint64_t* world; const float* uv;
  RigFlatFrame(const RigObsSource& s, int64_t, int64_t base) : cam(s.cam + base), world(s.world + base), uv(s.uv + 2 * base) {}
  uint64_t camera(int64_t i) const { return cam[i]; }
  uint64_t point(int64_t i) const { return world[i]; }
  void pixel(int64_t i, float* o) const { o[0] = uv[2 * i]; o[1] = uv[2 * i + 1]; }
};
struct RigColumnFrame {   // one frame of cc_obs_columns (records with a stride, or arrays: stride = width)
  const unsigned char *pc, *pw, *pu; int64_t sc, sw, su; bool c4, w4;
  RigColumnFrame(const RigObsSource& s, int64_t f, int64_t)
      : pc(static_cast<const unsigned char*>(s.cols->camera[f])), pw(static_cast<const unsigned char*>(s.cols->world[f])),
        pu(static_cast<const unsigned char*>(s.cols->uv[f])), sc(s.cols->camera_stride), sw(s.cols->world_stride), su(s.cols->uv_stride),
        c4(s.cols->camera_width == 4), w4(s.cols->world_width == 4) {}
  static uint64_t id(const unsigned char* p, bool four) {
    if (four) { uint32_t v; std::memcpy(&v, p, 4); return v; }
    uint64_t v; std::memcpy(&v, p, 8); return v;
  }
  uint64_t camera(int64_t i) const { return id(pc + i * sc, c4); }
  uint64_t point(int64_t i) const { return id(pw + i * sw, w4); }
  void pixel(int64_t i, float* o) const { std::memcpy(o, pu + i * su, 8); }
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
      if (INV) reinterpret_cast<int32_t*>(perm)[base + i] = (int32_t)(dst - base); else perm[dst] = base + i;   // (columns: where observation i of the frame went, 32 bits -- the costs are written frame by frame)
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
  if (N > 0 && ((!src.cols && (!src.cam || !src.world || !src.uv)) || !world_xyz)) return fail(CC_ERR_BAD_ARGUMENT, "cc_rig_create: NULL arrays");
  if (src.cols) {
    const cc_obs_columns& l = *src.cols;
    if (!l.camera || !l.world || !l.uv || (l.camera_width != 4 && l.camera_width != 8) || (l.world_width != 4 && l.world_width != 8) ||
        l.camera_stride < l.camera_width || l.world_stride < l.world_width || l.uv_stride < 8 || (l.cost && l.cost_stride < 8))
      return fail(CC_ERR_BAD_ARGUMENT, "cc_rig_optimize_columns: a column is missing, an id is not 4 or 8 bytes wide or a stride is below its field");
    for (int64_t f = 0; f < F; ++f)
      if (off[f + 1] > off[f] && (!l.camera[f] || !l.world[f] || !l.uv[f] || (l.cost && !l.cost[f])))
        return fail(CC_ERR_BAD_ARGUMENT, "cc_rig_optimize_columns: frame %lld has observations but a NULL column", (long long)f);
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
  // one pinned piece for both: the regrouped observations and (behind them) the handle's small host-built tables -- dev_upload
  // copies a table in and enqueues an asynchronous copy (per group 16 B, per frame 4 CO + 128 B, per shared column / direct entry
  // a few words: 4 MB covers BASELINE configs[4] several times over; a table that does not fit takes the synchronous copy).
  // ONE piece per call: the staging cache lends four, and a call that needed two would leave two concurrent callers without.
  const size_t up_cap = (size_t)4 << 20;
  const size_t b_obs = (b_uv + (size_t)N * sizeof(int32_t) + 255) & ~(size_t)255;
  char* st = static_cast<char*>(staging_get(b_obs + up_cap + 256, &st_cached));
  if (!st) return fail(CC_ERR_HIP, "cc_rig_create: pinned staging memory could not be allocated");
  struct StGuard {   // the uploads read the block until the handle's stream has drained
    void* p; cc_rig* h;
    ~StGuard() { if (h->stream) (void)hipStreamSynchronize(h->stream); h->up_stage = nullptr; h->up_cap = h->up_used = 0; staging_put(p); }
  } stg{st, h};
  float* uv_s = reinterpret_cast<float*>(st);
  int32_t* widx_s = reinterpret_cast<int32_t*>(st + b_uv);
  if (int rc = stream_get(device, &h->stream)) return rc;
  h->up_stage = st + b_obs; h->up_cap = up_cap; h->up_used = 0;
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
    h->perm_inverse = src.cols != nullptr;
    uint8_t* seen_p = seen.data();
    parallel_tasks(parts, [&](int t) {
      if (src.cols) rig_regroup_part<RigColumnFrame, true>(src, C, n_world, off, pf[(size_t)t], pf[(size_t)t + 1], perm, uv_s, widx_s, seen_p, part[(size_t)t], flush);
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
  if (int rc = dev_zeroed(h, &d.shared_stats, (size_t)64)) return rc;   // [0..3] statistics; the rest is what the timing variants leave their marks in (scripts/variants/timing.patch)
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
  if (h) cc::last_call_status_record(h->ran_form >= 0 ? h->ran_form : cc_rig_solver_form(h), h->form_reruns, h->form_note);   // (what a one-shot call's caller can still ask for)
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
    static int drop_left = -2;
    const bool drop_control = persist_test_drop_control("CC_RIG_PERSIST_TEST_NO_CONTROL", &drop_left);   // (test hook: the workers' first wait gives up)
    // Workers first; the control's candidates are launched when every worker is RESIDENT (the last worker to start stores the
    // solve's tag into a pinned word this thread spins on: ~10 us, once per solve), so that whichever candidate runs first sits
    // on a compute unit no worker needs. (First version of the round: hipStreamWaitValue32 on signal memory -- right, but a
    // command processor that finds the value not there yet goes to sleep: +154 us per solve at 250 workgroups of 1024
    // threads. Without any gate a candidate that starts BEFORE the workers can claim the last compute unit of an XCD that
    // 32 workers need: seen at once on the first box tried.)
    q.gate = const_cast<unsigned long long*>(h->host_pub) + 22;
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
                  "%u workers had started, control workgroup %s (candidate %u, XCD %d); the solve was run again with three kernels per "
                  "iteration and the handle %s (kernel-serialising tools, a CU mask, another tenant on the device or another host thread "
                  "in a call that waits for the device cause this)",
                  r->st.iter, first_round ? "42 ms" : "1.3 s", h->pq.G, h->p_teams * 256, w[13], w[12] ? "claimed a compute unit" : "NEVER RAN", w[12] >> 8,
                  (int)(w[12] & 0xffu) - 1,
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
  // What the DEVICE has said about lean solves lately (persist_device_try, cc_common.hpp): a one-shot caller's handle is new
  // every call, so after a give-up the device's back-off window -- not this handle's memory -- keeps the next solves on the
  // three-kernel form; one solve probes the lean form again when the window is over.
  const bool lean_wanted = h->persist_w_ok && !r.profile && !h->comm && !h->exchange && !h->big && h->co_resident <= 1;
  bool lean_tried = false;
  if (lean_wanted) {
    lean_tried = persist_device_try(h->device, 1);
    r.no_persist = !lean_tried;
  }
  // (the lean form is two launches that wait for each other inside their kernels: one such solve at a time per device and
  // process -- persist_mutex, cc_common.hpp; a second host thread waits here instead of inside a kernel for 1.3 s)
  std::unique_lock<std::mutex> lean_lock(persist_mutex(h->device), std::defer_lock);
  for (int chunk = 0;; ++chunk) {
    if (chunk == 0 && !r.no_persist && h->persist_w_ok && !r.profile && !h->comm && !h->exchange && !h->big && h->co_resident <= 1)
      lean_lock.lock();
    if (int rc = rig_launch(h, &r, chunk)) {
      if (chunk == 0 && lean_tried) persist_device_gave_up(h->device, 1);   // (a probe that never started is over too)
      return rc;
    }
    if (chunk == 0) hp.mark("launch0");
    if (int rc = rig_wait(h, &r)) {
      if (chunk == 0 && lean_tried && r.persist) persist_device_gave_up(h->device, 1);   // (the probe is over, whatever ended it)
      return rc;
    }
    // (a lean solve that gave up: its control launch on the second stream may still be queued or spinning -- the lock is kept
    // until both streams have drained below, or another thread's lean solve would start next to that late control workgroup)
    if (lean_lock.owns_lock() && !r.rerun) lean_lock.unlock();
    if (chunk == 0) hp.mark("wait0");
    if (r.rerun) persist_device_gave_up(h->device, 1);
    else if (chunk == 0 && lean_tried && r.persist) persist_device_completed(h->device, 1);
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
  h->ran_form = r.persist ? 2 : 0;
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
    const int64_t* perm = h->perm.data();
    if (h->perm_inverse) {   // (a handle made from columns: frame by frame)
      CC_HIP(hipMemcpyAsync(sorted, h->d_cost, (size_t)h->N * sizeof(double), hipMemcpyDeviceToHost, h->stream));
      CC_HIP(hipStreamSynchronize(h->stream));
      const int32_t* rel = reinterpret_cast<const int32_t*>(perm);
      for (int64_t f = 0; f < h->F; ++f) {
        const int64_t a = h->goff_h[(size_t)h->fgoff_h[(size_t)f]], b = h->goff_h[(size_t)h->fgoff_h[(size_t)f + 1]];
        for (int64_t i = a; i < b; ++i) obs_cost[i] = sorted[a + rel[i]];
      }
    } else {
      // one transfer and one event per worker's range: a worker scatters its range as soon as it has arrived, under the
      // transfers of the ranges behind it (as cc_rig_optimize_columns does)
      const int parts = parallel_parts(h->N, (int64_t)1 << 17);
      std::vector<hipEvent_t> ev((size_t)parts, nullptr);
      struct EvGuard { std::vector<hipEvent_t>& e; ~EvGuard() { for (auto x : e) if (x) (void)hipEventDestroy(x); } } evg{ev};
      for (int t = 0; t < parts; ++t) {
        const int64_t a = h->N * t / parts, b = h->N * (t + 1) / parts;
        if (b > a) CC_HIP(hipMemcpyAsync(sorted + a, h->d_cost + a, (size_t)(b - a) * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        CC_HIP(hipEventCreateWithFlags(&ev[(size_t)t], hipEventDisableTiming));
        CC_HIP(hipEventRecord(ev[(size_t)t], h->stream));
      }
      std::vector<char> bad((size_t)parts, 0);
      parallel_tasks(parts, [&](int t) {
        if (hipSetDevice(h->device) != hipSuccess || hipEventSynchronize(ev[(size_t)t]) != hipSuccess) { bad[(size_t)t] = 1; return; }
        for (int64_t i = h->N * t / parts, b = h->N * (t + 1) / parts; i < b; ++i) obs_cost[perm[i]] = sorted[i];
      });
      CC_HIP(hipStreamSynchronize(h->stream));
      for (char b : bad) if (b) return fail(CC_ERR_HIP, "cc_rig_get_state: the read-back of the costs failed");
    }
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


// The one-shot solve on ONE device with the observations as per-frame COLUMNS (what ExtrinsicsCalibrator::Optimize holds: one array
// of camera ids, one of point ids, one of image points and one of costs per frame, extrinsics_calibrator.hh): regrouped by
// (frame, camera) straight from those arrays, the costs scattered back into them. Replaces extrinsics_calibrator.cpp:92-225.
int cc_rig_optimize_columns(const cc_options* opt, int32_t device, int64_t C, int64_t F, int64_t n_world,
                            const cc_obs_columns* columns, const int64_t* counts,
                            const float* world_xyz, double* cam_q, double* cam_t, const uint8_t* cam_frozen,
                            double* frame_q, double* frame_t, double huber_a, cc_summary* summary) {
  cc::last_call_status_reset();
  using namespace cc;
  if (F < 1 || !columns || !counts) return fail(CC_ERR_BAD_ARGUMENT, "cc_rig_optimize_columns: bad arguments");
  HostPhases hp("cc_rig_optimize_columns");
  std::vector<int64_t> off((size_t)F + 1, 0);
  for (int64_t f = 0; f < F; ++f) {
    if (counts[f] < 0 || counts[f] >= INT32_MAX) return fail(CC_ERR_BAD_ARGUMENT, "cc_rig_optimize_columns: a frame's count is negative or beyond 2^31 - 2");
    off[(size_t)f + 1] = off[(size_t)f] + counts[f];
  }
  RigObsSource src;
  src.cols = columns;
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
  if (!rc && columns->cost && h->N > 0) {
    // per-observation costs (extrinsics_calibrator.cpp:219-225) through the pinned block, then into the caller's column on
    // several host threads (a frame's observations stay inside the frame's range of the regrouped order)
    rc = [&]() -> int {
      LmCtl c;
      if (int r2 = rig_read_ctl(h, &c)) return r2;
      hipLaunchKernelGGL(k_rig_obs_cost, dim3((unsigned)h->NG), dim3(256), 0, h->stream, h->d, c.cur & 1, h->d_cost);
      CC_HIP(hipGetLastError());
      bool st_cached = false;
      double* sorted = static_cast<double*>(staging_get((size_t)h->N * sizeof(double), &st_cached));
      if (!sorted) return fail(CC_ERR_HIP, "cc_rig_optimize_columns: pinned staging memory could not be allocated");
      struct StGuard { void* p; hipStream_t s; ~StGuard() { (void)hipStreamSynchronize(s); staging_put(p); } } stg{sorted, h->stream};
      const int32_t* rel = reinterpret_cast<const int32_t*>(h->perm.data());   // (perm_inverse: positions inside the frame)
      const int parts = parallel_parts(h->N, (int64_t)1 << 17);   // (a transfer of a megabyte and more per part)
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
      const int64_t stride = columns->cost_stride;
      std::vector<char> bad((size_t)parts, 0);
      parallel_tasks(parts, [&](int t) {
        if (hipSetDevice(h->device) != hipSuccess || hipEventSynchronize(ev[(size_t)t]) != hipSuccess) { bad[(size_t)t] = 1; return; }
        // record after record (the inverse permutation: sequential stores, the reads stay inside the frame's range of `sorted`)
        for (int64_t f = pf[(size_t)t]; f < pf[(size_t)t + 1]; ++f) {
          unsigned char* rec = static_cast<unsigned char*>(columns->cost[f]);
          const double* from = sorted + off[(size_t)f];
          for (int64_t k = off[(size_t)f]; k < off[(size_t)f + 1]; ++k, rec += stride) std::memcpy(rec, &from[rel[k]], sizeof(double));
        }
      });
      CC_HIP(hipStreamSynchronize(h->stream));
      for (char b : bad) if (b) return fail(CC_ERR_HIP, "cc_rig_optimize_columns: the read-back of the costs failed");
      return CC_OK;
    }();
  }
  hp.mark("get_state");
  cc_rig_destroy(h);
  hp.mark("destroy");
  return rc;
}

// The same for observations kept as one record each (stride, field offsets): columns that share their base pointers.
int cc_rig_optimize_frames(const cc_options* opt, int32_t device, int64_t C, int64_t F, int64_t n_world,
                           void* const* frame_records, const int64_t* counts, const cc_obs_layout* layout,
                           const float* world_xyz, double* cam_q, double* cam_t, const uint8_t* cam_frozen,
                           double* frame_q, double* frame_t, double huber_a, cc_summary* summary) {
  using namespace cc;
  if (F < 1 || !frame_records || !counts || !layout) { last_call_status_reset(); return fail(CC_ERR_BAD_ARGUMENT, "cc_rig_optimize_frames: bad arguments"); }
  const cc_obs_layout& l = *layout;
  if (l.stride < 8 || l.camera_offset < 0 || l.world_offset < 0 || l.uv_offset < 0 || l.camera_offset + 8 > l.stride ||
      l.world_offset + 8 > l.stride || l.uv_offset + 8 > l.stride || (l.cost_offset >= 0 && l.cost_offset + 8 > l.stride)) {
    last_call_status_reset();
    return fail(CC_ERR_BAD_ARGUMENT, "cc_rig_optimize_frames: record layout does not fit its stride");
  }
  std::vector<const void*> pc((size_t)F), pw((size_t)F), pu((size_t)F);
  std::vector<void*> pr((size_t)F);
  for (int64_t f = 0; f < F; ++f) {
    unsigned char* r = static_cast<unsigned char*>(frame_records[f]);
    if (!r && counts[f] > 0) { last_call_status_reset(); return fail(CC_ERR_BAD_ARGUMENT, "cc_rig_optimize_frames: frame %lld has no records", (long long)f); }
    pc[(size_t)f] = r ? r + l.camera_offset : nullptr;
    pw[(size_t)f] = r ? r + l.world_offset : nullptr;
    pu[(size_t)f] = r ? r + l.uv_offset : nullptr;
    pr[(size_t)f] = (r && l.cost_offset >= 0) ? r + l.cost_offset : nullptr;
  }
  cc_obs_columns cols{};
  cols.camera = pc.data(); cols.camera_stride = l.stride; cols.camera_width = 8;
  cols.world = pw.data(); cols.world_stride = l.stride; cols.world_width = 8;
  cols.uv = pu.data(); cols.uv_stride = l.stride;
  cols.cost = l.cost_offset >= 0 ? pr.data() : nullptr; cols.cost_stride = l.stride;
  return cc_rig_optimize_columns(opt, device, C, F, n_world, &cols, counts, world_xyz, cam_q, cam_t, cam_frozen, frame_q, frame_t, huber_a, summary);
}

// Multi-device rig solve driven by ONE host thread (SURVEY.md 8(b) thread model; cf. cc_intrinsics_optimize_multi):
// frames sharded contiguously by observation count, cameras and world points replicated, mailboxes wired inside the
// process, every device's chunk enqueued before any is waited for. A device id may appear several times.
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
