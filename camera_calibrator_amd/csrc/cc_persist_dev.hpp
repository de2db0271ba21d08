// cc_persist_dev.hpp -- the seams of the persistent per-solve kernels (cc_intrinsics_persist.hip, cc_rig.hip): everything
// that crosses workgroups inside a launch travels as self-validating 8-byte words {epoch32 : half of a double}, stored and
// polled with agent-scope (sc1) accesses -- an aligned 8-byte store is single-copy atomic, so there is no flag, no fence,
// no drain (MI355X guide, Guideline 16 R2). Device code only.
#pragma once
#include "cc_common.hpp"
#include "cc_device.hpp"

namespace cc {

typedef unsigned long long u64;

// 2^shift ticks of the 100 MHz wall clock (PersistDev::timeout_shift) -- a shift and a compare against zero: a 64-bit
// literal to compare with gets hoisted into a register pair that then sits there across the sweep's main loop
__device__ __forceinline__ bool timed_out(long long t0, int shift) { return ((wall_clock64() - t0) >> shift) != 0; }

__device__ __forceinline__ u64 ag_ld(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void ag_st(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned ag_ld32(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u64 granule(unsigned tag, double v, int half) {
  const u64 bits = (u64)__double_as_longlong(v);
  return ((u64)tag << 32) | (half ? (bits >> 32) : (bits & 0xffffffffull));
}
__device__ __forceinline__ double ungranule(u64 lo, u64 hi) { return __longlong_as_double((long long)((hi << 32) | (lo & 0xffffffffull))); }

// The trust-region radius after an accepted step of quality >= 0.937: Ceres' r / max(1/3, 1 - (2 rho - 1)^3) with the
// maximum taken by its first argument (lm_apply, cc_common.hpp). Workers and control evaluate this one expression.
__device__ __forceinline__ double persist_spec_radius(double radius, double max_radius) { return fmin(max_radius, radius / (1.0 / 3.0)); }

// One wave waits until the n doubles of a broadcast box carry `tag` and leaves them in dst[0..n) (LDS); lane l polls
// word l. false: gave up (timeout, or somebody else already failed); the failure word is set.
__device__ __forceinline__ bool bcast_wait(const u64* box, unsigned tag, int n, double* dst, unsigned* fail, int lane, int tshift) {
  const bool mine = lane < 2 * n;
  const u64* p = box + (mine ? lane : 0);
  const long long t0 = wall_clock64();
  u64 v;
  for (unsigned spins = 0;; ++spins) {
    v = ag_ld(p);
    const int ok = !mine || (unsigned)(v >> 32) == tag;
    if (__all(ok)) break;
    if ((spins & 63u) == 63u && (timed_out(t0, tshift) || ag_ld32(fail) != 0u)) {
      if (lane == 0) __hip_atomic_store(fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return false;
    }
    __builtin_amdgcn_s_sleep(1);
  }
  if (mine) reinterpret_cast<unsigned*>(dst)[lane] = (unsigned)v;   // word 2i = low half of double i
  return true;
}

// ceres::QuaternionManifold::Plus with the series coefficients of cc::quat_plus (cc_common.hpp: same values, same
// Horner order, same bits) read from constant memory through a pointer the compiler cannot see through: written as
// literals they are hoisted out of the round loop -- sixteen 64-bit constants parked in vector registers across the
// sweep's main loop, which promptly spills them.
static __constant__ double kPlusCoef[16] = {-1.0 / 2, 1.0 / 24, -1.0 / 720, 1.0 / 40320, -1.0 / 3628800, 1.0 / 479001600, -1.0 / 87178291200.0,
                                     1.0 / 20922789888000.0,   // cos(n) - 1 in n^2
                                     -1.0 / 6, 1.0 / 120, -1.0 / 5040, 1.0 / 362880, -1.0 / 39916800, 1.0 / 6227020800.0,
                                     -1.0 / 1307674368000.0, 1.0 / 355687428096000.0};   // sin(n) / n - 1 in n^2
__device__ __forceinline__ void quat_plus_tab(const double* x, const double* d, double* out) {
  const double n2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
  if (n2 == 0.0) { out[0] = x[0]; out[1] = x[1]; out[2] = x[2]; out[3] = x[3]; return; }
  double s, a0;
  if (n2 < 0.0625) {
    const double* c = kPlusCoef;
    asm volatile("" : "+s"(c));
    a0 = 1.0 + n2 * (c[0] + n2 * (c[1] + n2 * (c[2] + n2 * (c[3] + n2 * (c[4] + n2 * (c[5] + n2 * (c[6] + n2 * c[7])))))));
    s = 1.0 + n2 * (c[8] + n2 * (c[9] + n2 * (c[10] + n2 * (c[11] + n2 * (c[12] + n2 * (c[13] + n2 * (c[14] + n2 * c[15])))))));
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

// cc::pose_grad_proj_max (cc_common.hpp: same values, same order, same bits -- 1 - cos n is the negated Horner sum of the
// negated coefficients) with the series read from the same table, for the same reason
__device__ __forceinline__ double pose_grad_proj_max_tab(const double* q, const double* g) {
  const double gt = fmax(fmax(fabs(g[3]), fabs(g[4])), fabs(g[5]));
  const double n2 = g[0] * g[0] + g[1] * g[1] + g[2] * g[2];
  if (!(n2 < 0.0625)) return fmax(gt, fmax(fmax(fabs(g[0]), fabs(g[1])), fabs(g[2])));
  const double* c = kPlusCoef;
  asm volatile("" : "+s"(c));
  const double c1 = -(n2 * (c[0] + n2 * (c[1] + n2 * (c[2] + n2 * (c[3] + n2 * (c[4] + n2 * (c[5] + n2 * (c[6] + n2 * c[7]))))))));
  const double s = 1.0 + n2 * (c[8] + n2 * (c[9] + n2 * (c[10] + n2 * (c[11] + n2 * (c[12] + n2 * (c[13] + n2 * (c[14] + n2 * c[15])))))));
  const double w = q[0], v0 = q[1], v1 = q[2], v2 = q[3];
  const double dw = c1 * w - s * (g[0] * v0 + g[1] * v1 + g[2] * v2);
  const double d0 = c1 * v0 + s * (w * g[0] + (g[1] * v2 - g[2] * v1));
  const double d1 = c1 * v1 + s * (w * g[1] + (g[2] * v0 - g[0] * v2));
  const double d2 = c1 * v2 + s * (w * g[2] + (g[0] * v1 - g[1] * v0));
  return fmax(fmax(gt, fabs(dw)), fmax(fmax(fabs(d0), fabs(d1)), fabs(d2)));
}

}  // namespace cc
