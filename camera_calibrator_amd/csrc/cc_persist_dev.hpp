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

}  // namespace cc
