// cc_common.hpp -- shared host/device definitions of the MI355X LM solver (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <functional>
#include <mutex>
#include <string>

#include "../../include/cc_solver.h"

namespace cc {

// ---------------------------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------------------------
std::string& last_error();
int fail(int code, const char* fmt, ...);

#define CC_HIP(expr)                                                                         \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess)                                                                    \
      return ::cc::fail(CC_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),   \
                        __FILE__, __LINE__);                                                 \
  } while (0)

int select_device(int device);

// Small per-process caches of resources that are slow to create and destroy (hipHostMalloc/hipHostFree
// ~0.2-0.4 ms, streams ~0.1 ms): a one-shot solve should not pay for them every time. Pinned blocks
// are 512 bytes (control block / options staging); streams are per device. Thread-safe; entries live
// until the process ends.
void* pinned_block_get();
void pinned_block_put(void* p);
int zhang_on_device(hipStream_t stream, int64_t F, const int64_t* doff, const float* duv, const float* dxyz,
                    double* dgram, float* dH, float* dK, float* dq, float* dt);   // cc_zhang.hip
int stream_get(int device, hipStream_t* out);
void stream_put(int device, hipStream_t s);
// One reusable scratch buffer per device for the point kernels (Distort / Undistort are called once per frame by the
// Python workflow): grows to the largest request (<= 64 MiB kept), handed out to one caller at a time; a second
// concurrent caller simply gets a fresh allocation. scratch_put frees what was not taken from / cannot go back to the cache.
int arena_get(int device, size_t bytes, void** out, bool* cached);    // device arena of a solver handle (one cached, grow-only)
void arena_put(int device, void* p, bool cached);
void* staging_get(size_t bytes, bool* cached);                       // pinned host staging (one cached, grow-only)
// Device blocks of handles that allocate piece by piece (the rig path): a destroyed handle's blocks wait here (per device, at
// most kPoolKeep bytes in all) for the next handle that asks for the same size -- a caller that optimises again with a few more
// frames pays hipMalloc / hipFree for the difference only.
int pool_alloc(int device, size_t bytes, void** out, size_t* got);
void pool_free(int device, void* p, size_t bytes);
// fn(part, begin, end) over [0, n) cut into contiguous parts, one host thread each (at most 16, one when n < 2 min_per_part);
// returns after all parts. parallel_parts gives the number of parts fn will see.
// One persistent launch at a time per device and process: a persistent solve (the intrinsics kernel, the lean rig form's two
// launches) waits INSIDE its kernels for workgroups of its own grid, so two of them in flight from two host threads can hold
// each other's compute units or hardware queues (a process has four; streams share them) until both give up after 1.3 s.
// The second caller waits here instead. Not taken by the sharded forms (their ranks must run together by construction).
std::mutex& persist_mutex(int device);
// What a device has told this PROCESS about persistent solves (the one-shot entry points create and destroy a handle per
// call, so a handle's own memory of a give-up dies with it: a second tenant on the GPU, or a tool that serialises kernels,
// would otherwise make EVERY Calibrator::Estimate / ExtrinsicsCalibrator::Optimize wait out the kernel's timeout). After a
// give-up the device is in a back-off window: persist_device_try() says no for the next `calls` solves AND `seconds`
// (both must have passed), then lets exactly one solve probe the persistent form again; a second give-up doubles the
// window (8 solves / 2 s at first, at most 1024 solves / 10 min), a solve that completes ends it. kind: 0 the intrinsics
// kernel, 1 the lean rig pair.
// Test hook of the persistent forms (CC_INTR_PERSIST_TEST_NO_CONTROL / CC_RIG_PERSIST_TEST_NO_CONTROL: launch the grid WITHOUT its
// control workgroup so that the workers' first wait gives up): "1" every launch of the process, "firstN" its first N only,
// unset / "0" none. `remaining` is the caller's static counter (-2: not parsed yet).
bool persist_test_drop_control(const char* env_name, int* remaining);
bool persist_device_try(int device, int kind);
void persist_device_gave_up(int device, int kind);
void persist_device_completed(int device, int kind);
void rig_release_host_caches();   // (cc_rig.hip: the permutation storage kept for the next rig handle; part of cc_release_caches)
int parallel_parts(int64_t n, int64_t min_per_part);
void parallel_tasks(int parts, const std::function<void(int)>& fn);   // fn(0 .. parts-1), one host thread each
void parallel_ranges(int64_t n, int64_t min_per_part, const std::function<void(int, int64_t, int64_t)>& fn);
void last_call_status_reset();                                                   // one-shot entry points: at their start
void last_call_status_record(int form, int reruns, const std::string& note);    // ... and for every handle they destroy
void parallel_pool_release();   // joins the pool's threads (cc_release_caches); the next parallel_tasks starts them again
int parallel_pool_threads();
void staging_put(void* p);
double* last_timing();                                               // [5] phases of this thread's last one-shot call (ms)
int scratch_get(int device, size_t bytes, void** out, bool* cached);
void scratch_put(int device, void* p, size_t bytes, bool cached);

// ---------------------------------------------------------------------------------------------
// Mailbox exchange between the ranks of one node (device side: cc_device.hpp; host side: cc_comm.cpp)
// ---------------------------------------------------------------------------------------------
constexpr int kP2pMaxRanks = 8;
constexpr long long kP2pTimeoutTicks = 1000000000LL;  // 10 s of the 100 MHz wall clock

struct P2pDev {
  unsigned long long* box[kP2pMaxRanks];   // box[r]: rank r's mailbox as mapped here (box[rank] is local)
  unsigned long long* seq;                 // [2] last epoch per kind (device memory of this rank)
  int32_t sw[2];                           // words per slot of kind 0 / kind 1 (two words per double)
  int32_t on;
  int32_t pad;
};

struct Mailbox {  // host-side owner of one rank's mailbox and of its mappings of the peers' mailboxes
  unsigned long long* local = nullptr;
  unsigned long long* peer[kP2pMaxRanks] = {};
  bool peer_ipc[kP2pMaxRanks] = {};        // peer[r] was opened with hipIpcOpenMemHandle (and must be closed)
  unsigned long long* seq = nullptr;
  int sw[2] = {0, 0};
};
int mailbox_alloc(Mailbox* m, int doubles_kind0, int doubles_kind1);   // on the current device
int mailbox_export(Mailbox* m, int doubles_kind0, int doubles_kind1, uint8_t handle[64]);
int mailbox_attach(Mailbox* m, int rank, int nranks, const uint8_t* handles, P2pDev* out);
// Same wiring inside ONE process (one host thread driving several devices): the peers' mailboxes are used through
// their own pointers; peer access between distinct devices is switched on. all[r] / devices[r]: rank r's mailbox
// and device.
int mailbox_wire_local(Mailbox* m, int rank, int nranks, Mailbox* const* all, const int* devices, P2pDev* out);
void mailbox_release(Mailbox* m);
std::string mailbox_describe(const Mailbox* m, int rank, int nranks);   // diagnostics after a timed-out wait

// ---------------------------------------------------------------------------------------------
// Device-resident LM state machine.  All control decisions (step acceptance, radius update,
// convergence tests) are taken on the device by a single thread of the `decide` kernel, so an
// LM iteration is a fixed sequence of kernel launches that can be enqueued ahead / replayed
// from a hipGraph; kernels turn into no-ops once `done` is set.
// Semantics follow Ceres' TrustRegionMinimizer / LevenbergMarquardtStrategy /
// TrustRegionStepEvaluator, which is what the reference runs (calibrator.cpp:314-324).
// ---------------------------------------------------------------------------------------------
struct LmOpts {
  int32_t max_iterations, use_nonmonotonic_steps, max_consecutive_nonmonotonic_steps, jacobi_scaling;
  int32_t max_consecutive_invalid_steps, pad0_;
  double function_tolerance, gradient_tolerance, parameter_tolerance;
  double initial_radius, max_radius, min_radius, min_relative_decrease, min_lm_diagonal, max_lm_diagonal;
};

// Scalar control block of the LM state machine (lives in HBM, 144 bytes; the deciding thread works
// on a register copy: one global round trip each way).
struct LmCtl {
  int32_t done, term, iter, n_success, n_invalid;
  int32_t cur;           // index (0/1) of the buffers holding the accepted point and its blocks
  int32_t phase;         // 0: initial evaluation pending, 1: iterating
  int32_t step_valid;    // set by the solve kernel: the linear solve succeeded
  int32_t cand_pending;  // a candidate point has been evaluated and awaits the decision
  int32_t num_nonmono, log_len, sweeps;
  double radius, decrease_factor, x_cost, x_norm, gmax, initial_cost;
  double minimum_cost, current_cost, reference_cost, candidate_cost, acc_ref, acc_cand;
};
static_assert(sizeof(LmCtl) == 144, "LmCtl layout");

__host__ inline void opts_from_public(const cc_options& o, LmOpts* d) {
  d->max_iterations = o.max_iterations;
  d->use_nonmonotonic_steps = o.use_nonmonotonic_steps;
  d->max_consecutive_nonmonotonic_steps = o.max_consecutive_nonmonotonic_steps;
  d->jacobi_scaling = o.jacobi_scaling;
  d->max_consecutive_invalid_steps = o.max_consecutive_invalid_steps;
  d->pad0_ = 0;
  d->function_tolerance = o.function_tolerance;
  d->gradient_tolerance = o.gradient_tolerance;
  d->parameter_tolerance = o.parameter_tolerance;
  d->initial_radius = o.initial_radius;
  d->max_radius = o.max_radius;
  d->min_radius = o.min_radius;
  d->min_relative_decrease = o.min_relative_decrease;
  d->min_lm_diagonal = o.min_lm_diagonal;
  d->max_lm_diagonal = o.max_lm_diagonal;
}

// profile_kernels = 1: event pairs of a solve -> per-kind totals. Rounds are enqueued a chunk at a time, so the launches after
// the terminating decision (made in round `iter`) return at once: they go to kernel_idle_*, not into the per-launch averages.
// A launch of round r did work if r < iter, or r == iter for the kinds ahead of the decision (sweep, statistics).
template <class Event, class Elapsed>
inline void summarise_probes(const std::vector<Event>& events, const std::vector<int>& kinds, const std::vector<int>& rounds, int iter,
                             cc_summary* s, Elapsed elapsed_ms) {
  for (int i = 0; i < CC_K_COUNT; ++i) { s->kernel_ms[i] = s->kernel_idle_ms[i] = 0.0; s->kernel_launches[i] = s->kernel_idle_launches[i] = 0; }
  for (size_t i = 0; i < kinds.size(); ++i) {
    float ms = 0.f;
    if (!elapsed_ms(&ms, events[2 * i], events[2 * i + 1])) continue;
    const int k = kinds[i], r = i < rounds.size() ? rounds[i] : 0;
    const bool worked = r < iter || (r == iter && (k == CC_K_SWEEP || k == CC_K_DECIDE));
    if (worked) { s->kernel_ms[k] += ms; s->kernel_launches[k]++; }
    else { s->kernel_idle_ms[k] += ms; s->kernel_idle_launches[k]++; }
  }
}

#if defined(__HIPCC__)

// One log record per LM iteration. lm_decide fills `e` (when the caller passes one); the caller stores it at
// log[st.log_len - 1] (after the gradient of an accepted point is known: solve step).
__device__ inline void lm_log(LmCtl& st, cc_iteration* e, double cost, double cc_,
                              double mcc, double rd, double sn, int acc, int valid) {
  if (e) {
    e->cost = cost; e->cost_change = cc_; e->model_cost_change = mcc; e->relative_decrease = rd;
    e->gradient_max_norm = st.gmax; e->step_norm = sn; e->radius = st.radius; e->accepted = acc;
    e->valid = valid;
  }
  st.log_len++;
}

// Initial evaluation bookkeeping (TrustRegionMinimizer::Init + IterationZero).
// The gradient test of iteration 0 happens in the first solve kernel (it owns the gradient).
__device__ inline void lm_init(LmCtl& st, const LmOpts& o, double cost, double x_norm) {
  st.x_cost = cost; st.initial_cost = cost; st.x_norm = x_norm; st.gmax = 0.0;
  st.minimum_cost = st.current_cost = st.reference_cost = st.candidate_cost = cost;
  st.acc_ref = st.acc_cand = 0.0; st.num_nonmono = 0;
  st.radius = o.initial_radius; st.decrease_factor = 2.0;
  st.iter = 0; st.n_success = 0; st.n_invalid = 0; st.phase = 1; st.step_valid = 0; st.cand_pending = 0;
  st.sweeps = 1;
  if (o.max_iterations <= 0) { st.done = 1; st.term = CC_NO_CONVERGENCE; }
}

// One LM iteration's decision, in two halves with exactly the arithmetic of the single function they replace.
// lm_trial: is the step valid, did a tolerance fire, is the candidate accepted (TrustRegionStepEvaluator::StepQuality).
// It does not touch the state, so a kernel can act on `accept` (which buffer holds the point to eliminate next) while
// lm_apply -- radius update, non-monotonic bookkeeping, log record, iteration limit -- is still running.
struct LmTrial {
  int valid;        // the linear solve succeeded and the model predicts a decrease
  int conv;         // 0 none, 1 parameter tolerance, 2 function tolerance
  int accept;       // candidate accepted (st.cur flips)
  double mcc, step_norm, cand_cost, cost_change, quality;
};

__device__ inline LmTrial lm_trial(const LmCtl& st, const LmOpts& o, double cand_cost, double q_model, double step2) {
  LmTrial t;
  t.mcc = -q_model;
  t.valid = st.step_valid && (t.mcc > 0.0) && isfinite(t.mcc);
  t.conv = 0; t.accept = 0; t.step_norm = 0.0; t.cost_change = 0.0; t.quality = 0.0;
  t.cand_cost = cand_cost;
  if (!t.valid) return t;
  t.step_norm = sqrt(step2);
  if (!isfinite(t.cand_cost)) t.cand_cost = 1.7976931348623157e308;
  t.cost_change = st.x_cost - t.cand_cost;
  if (t.step_norm <= o.parameter_tolerance * (st.x_norm + o.parameter_tolerance)) { t.conv = 1; return t; }
  if (fabs(t.cost_change) <= o.function_tolerance * st.x_cost) { t.conv = 2; return t; }
  // TrustRegionStepEvaluator::StepQuality
  if (!(t.cand_cost < 1.7976931348623157e308)) {
    t.quality = -1.7976931348623157e308;
  } else {
    const double rel = (st.current_cost - t.cand_cost) / t.mcc;
    const double hist = (st.reference_cost - t.cand_cost) / (st.acc_ref + t.mcc);
    t.quality = fmax(rel, hist);
  }
  t.accept = t.quality > o.min_relative_decrease;
  return t;
}

// An accepted candidate flips st.cur. The gradient-tolerance test of the new point is made by the
// solve step, which owns the reduced gradient (lm_finalize; it also completes the log record).
__device__ inline void lm_apply(LmCtl& st, const LmOpts& o, cc_iteration* e, const LmTrial& t, double xnorm2_cand) {
  st.iter++;
  if (st.step_valid) st.sweeps++;
  const double mcc = t.mcc;
  if (!t.valid) {
    // HandleInvalidStep + LevenbergMarquardtStrategy::StepIsInvalid
    st.n_invalid++;
    st.radius /= st.decrease_factor;
    st.decrease_factor *= 2.0;
    lm_log(st, e, st.x_cost, 0.0, mcc, 0.0, 0.0, 0, 0);
    if (st.n_invalid >= o.max_consecutive_invalid_steps) { st.done = 1; st.term = CC_FAILURE_INVALID_STEPS; }
  } else {
    st.n_invalid = 0;
    const double step_norm = t.step_norm, cand_cost = t.cand_cost, cost_change = t.cost_change, quality = t.quality;
    if (t.conv == 1) {
      st.done = 1; st.term = CC_CONVERGENCE_PARAMETER;
      lm_log(st, e, st.x_cost, cost_change, mcc, 0.0, step_norm, 0, 1);
    } else if (t.conv == 2) {
      st.done = 1; st.term = CC_CONVERGENCE_FUNCTION;
      lm_log(st, e, st.x_cost, cost_change, mcc, 0.0, step_norm, 0, 1);
    } else {
      if (t.accept) {
        st.cur ^= 1;
        st.x_cost = cand_cost;
        st.x_norm = sqrt(xnorm2_cand);
        const double q3 = 2.0 * quality - 1.0;
        st.radius = fmin(o.max_radius, st.radius / fmax(1.0 / 3.0, 1.0 - q3 * q3 * q3));
        st.decrease_factor = 2.0;
        // TrustRegionStepEvaluator::StepAccepted
        st.current_cost = cand_cost;
        st.acc_cand += mcc;
        st.acc_ref += mcc;
        if (st.current_cost < st.minimum_cost) {
          st.minimum_cost = st.current_cost; st.num_nonmono = 0;
          st.candidate_cost = st.current_cost; st.acc_cand = 0.0;
        } else {
          st.num_nonmono++;
          if (st.current_cost > st.candidate_cost) { st.candidate_cost = st.current_cost; st.acc_cand = 0.0; }
        }
        const int maxn = o.use_nonmonotonic_steps ? o.max_consecutive_nonmonotonic_steps : 0;
        if (st.num_nonmono == maxn) { st.reference_cost = st.candidate_cost; st.acc_ref = st.acc_cand; }
        st.n_success++;
        lm_log(st, e, st.x_cost, cost_change, mcc, quality, step_norm, 1, 1);
      } else {
        st.radius /= st.decrease_factor;
        st.decrease_factor *= 2.0;
        lm_log(st, e, st.x_cost, cost_change, mcc, quality, step_norm, 0, 1);
      }
    }
  }
  // TrustRegionMinimizer::FinalizeIterationAndCheckIfMinimizerCanContinue tests max iterations, then the
  // gradient tolerance, then the minimum radius: the first is made here, the other two by the solve step
  // (lm_finalize), which owns the gradient of the accepted point.
  if (!st.done && st.iter >= o.max_iterations) { st.done = 1; st.term = CC_NO_CONVERGENCE; }
  st.step_valid = 0;
  st.cand_pending = 0;
}

__device__ inline void lm_decide(LmCtl& st, const LmOpts& o, cc_iteration* e,
                                 double cand_cost, double q_model, double step2, double xnorm2_cand) {
  const LmTrial t = lm_trial(st, o, cand_cost, q_model, step2);
  lm_apply(st, o, e, t, xnorm2_cand);
}

// Second half of FinalizeIterationAndCheckIfMinimizerCanContinue, run by the solve step once the gradient
// max-norm of the accepted point is known: gradient tolerance, then minimum trust-region radius. Returns
// true when the minimiser goes on (the caller then computes the next step).
__device__ inline bool lm_finalize(LmCtl& st, const LmOpts& o, double gmax) {
  st.gmax = gmax;
  if (gmax <= o.gradient_tolerance) { st.done = 1; st.term = CC_CONVERGENCE_GRADIENT; return false; }
  if (st.radius < o.min_radius) { st.done = 1; st.term = CC_MIN_RADIUS; return false; }
  return true;
}

// ---- small device math -------------------------------------------------------------------

__device__ inline double rfl(double x) {  // wave-uniform value -> SGPR pair
  const int lo = __builtin_amdgcn_readfirstlane(__double2loint(x));
  const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(x));
  return __hiloint2double(hi, lo);
}

// 1 / sqrt(d), d > 0 and finite: the hardware estimate with the library's third-order correction, without the library's
// special-case selects (four dependent instructions; sqrt followed by a division is twenty-five, each waiting ~20 cycles
// for the one before it on a wave that has its SIMD to itself)
__device__ __forceinline__ double rsqrt_pos(double d) {
  const double y0 = __builtin_amdgcn_rsq(d);
  const double e = fma(y0 * -d, y0, 1.0);
  return fma(y0 * e, fma(e, 0.375, 0.5), y0);
}

// Rotation matrix of q/|q| (ceres::QuaternionRotatePoint normalises; calibrator.cpp:201)
__device__ inline void quat_to_R(const double* q, double* R) {
  const double n = rsqrt_pos(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);   // (a pose's quaternion is never zero)
  const double w = q[0] * n, x = q[1] * n, y = q[2] * n, z = q[3] * n;
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z);     R[2] = 2 * (x * z + w * y);
  R[3] = 2 * (x * y + w * z);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
  R[6] = 2 * (x * z - w * y);     R[7] = 2 * (y * z + w * x);     R[8] = 1 - 2 * (x * x + y * y);
}

// ceres::QuaternionManifold::Plus (calibrator.cpp:298)
__device__ inline void quat_plus(const double* x, const double* d, double* out) {
  const double n2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
  if (n2 == 0.0) { out[0] = x[0]; out[1] = x[1]; out[2] = x[2]; out[3] = x[3]; return; }
  double s, a0;
  if (n2 < 0.0625) {
    // LM steps are small rotations: cos(n) and sin(n)/n as even power series in n^2, truncated below
    // 1 ulp for n < 0.25 (next terms n^20/20! ~ 4e-31 and n^18/19! ~ 1e-28); ~20 FMAs on the
    // single-lane critical path of the sweep prologue instead of two libm calls and a square root
    a0 = 1.0 + n2 * (-1.0 / 2 + n2 * (1.0 / 24 + n2 * (-1.0 / 720 + n2 * (1.0 / 40320 + n2 * (-1.0 / 3628800 +
         n2 * (1.0 / 479001600 + n2 * (-1.0 / 87178291200.0 + n2 * (1.0 / 20922789888000.0))))))));
    s = 1.0 + n2 * (-1.0 / 6 + n2 * (1.0 / 120 + n2 * (-1.0 / 5040 + n2 * (1.0 / 362880 + n2 * (-1.0 / 39916800 +
        n2 * (1.0 / 6227020800.0 + n2 * (-1.0 / 1307674368000.0 + n2 * (1.0 / 355687428096000.0))))))));
  } else {
    const double nd = sqrt(n2);
    s = sin(nd) / nd;
    a0 = cos(nd);
  }
  const double a1 = s * d[0], a2 = s * d[1], a3 = s * d[2];
  out[0] = a0 * x[0] - a1 * x[1] - a2 * x[2] - a3 * x[3];
  out[1] = a0 * x[1] + a1 * x[0] + a2 * x[3] - a3 * x[2];
  out[2] = a0 * x[2] - a1 * x[3] + a2 * x[0] + a3 * x[1];
  out[3] = a0 * x[3] + a1 * x[2] - a2 * x[1] + a3 * x[0];
}

__device__ inline double clampd(double v, double lo, double hi) { return fmin(fmax(v, lo), hi); }

// Contribution of one pose block [q(4) t(3)] with tangent gradient g = [g_rot(3) g_t(3)] to Ceres' gradient_max_norm,
// || x - Plus(x, -g) ||_inf (TrustRegionMinimizer: the gradient tolerance is tested on the step a unit gradient descent
// would take THROUGH the manifold). Translation: |g_t|. Quaternion: q - Plus(q, -g_rot) with Plus as in quat_plus,
//   d_w = (1 - cos n) w - s (g.v),   d_v = (1 - cos n) v + s (w g + g x v),   n = |g_rot|, s = sin(n) / n, q = [w v],
// evaluated with 1 - cos n as its own series (Ceres subtracts two quaternions that agree to sixteen digits when g is at
// the tolerance; this is the same number without the cancellation). For |g_rot| >= 1/4 the block reports the tangent
// max-norm instead: sin / cos of a "rotation" by 1e6 radians per frame and iteration is not worth a number that is only
// ever compared with a tolerance of 1e-10 -- every decision is the same for tolerances below 0.14 (oracle: the same rule).
__device__ inline double pose_grad_proj_max(const double* q, const double* g) {
  const double gt = fmax(fmax(fabs(g[3]), fabs(g[4])), fabs(g[5]));
  const double n2 = g[0] * g[0] + g[1] * g[1] + g[2] * g[2];
  if (!(n2 < 0.0625)) return fmax(gt, fmax(fmax(fabs(g[0]), fabs(g[1])), fabs(g[2])));
  const double c1 = n2 * (1.0 / 2 + n2 * (-1.0 / 24 + n2 * (1.0 / 720 + n2 * (-1.0 / 40320 + n2 * (1.0 / 3628800 +
                    n2 * (-1.0 / 479001600 + n2 * (1.0 / 87178291200.0 + n2 * (-1.0 / 20922789888000.0))))))));
  const double s = 1.0 + n2 * (-1.0 / 6 + n2 * (1.0 / 120 + n2 * (-1.0 / 5040 + n2 * (1.0 / 362880 + n2 * (-1.0 / 39916800 +
                   n2 * (1.0 / 6227020800.0 + n2 * (-1.0 / 1307674368000.0 + n2 * (1.0 / 355687428096000.0))))))));
  const double w = q[0], v0 = q[1], v1 = q[2], v2 = q[3];
  const double dw = c1 * w - s * (g[0] * v0 + g[1] * v1 + g[2] * v2);
  const double d0 = c1 * v0 + s * (w * g[0] + (g[1] * v2 - g[2] * v1));
  const double d1 = c1 * v1 + s * (w * g[1] + (g[2] * v0 - g[0] * v2));
  const double d2 = c1 * v2 + s * (w * g[2] + (g[0] * v1 - g[1] * v0));
  return fmax(fmax(gt, fabs(dw)), fmax(fmax(fabs(d0), fabs(d1)), fabs(d2)));
}


#endif  // __HIPCC__

}  // namespace cc
