// cc_device.hpp -- device helpers shared by the intrinsics and the rig kernels (gfx950).
#pragma once
#include "cc_common.hpp"

namespace cc {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

constexpr int kStageDoublesPerWave = 64 * 16;  // 64 observations x one row (u or v) x 16 components

__device__ __forceinline__ int tri(int i, int j) { return i * (i + 1) / 2 + j; }  // packed lower, i >= j

// Cross-lane sums without the LDS crossbar (`__shfl_xor` = two ds_bpermute and a wait per step): DPP moves inside a
// 16-lane row, and across rows the lane-swap instructions of gfx950 -- v_permlane16_swap exchanges the odd rows of its
// first register with the even rows of the second, v_permlane32_swap lanes 32..63 of the first with lanes 0..31 of the
// second; fed two copies of v, first + second is v + (v of the partner row / half) in every lane.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row_pair_sum(double v) {    // lanes l and l ^ 16
  const auto l = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(v), (unsigned)__double2loint(v), false, false);
  const auto h = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(v), (unsigned)__double2hiint(v), false, false);
  return __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
}
__device__ __forceinline__ double half_pair_sum(double v) {   // lanes l and l ^ 32
  const auto l = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(v), (unsigned)__double2loint(v), false, false);
  const auto h = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(v), (unsigned)__double2hiint(v), false, false);
  return __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
}
// sum over the lanes of a wave that agree in the low LOG2 bits of the lane index (LOG2 = 0: all 64), in every lane
template <int LOG2 = 0>
__device__ __forceinline__ double wave_sum_mod(double v) {
  static_assert(LOG2 >= 0 && LOG2 <= 3, "strides 1, 2, 4, 8");
  if (LOG2 == 0) v += dpp_f64<0xB1>(v);    // quad_perm:[1,0,3,2]
  if (LOG2 <= 1) v += dpp_f64<0x4E>(v);    // quad_perm:[2,3,0,1]
  if (LOG2 <= 2) v += dpp_f64<0x124>(v);   // row_ror:4
  v += dpp_f64<0x128>(v);                  // row_ror:8
  return half_pair_sum(row_pair_sum(v));
}
__device__ __forceinline__ double wave_sum(double v) { return wave_sum_mod<0>(v); }
// sum / maximum over the sixteen lanes of a row (lanes 16k .. 16k+15), in every lane
__device__ __forceinline__ double row16_sum(double v) {
  v += dpp_f64<0xB1>(v);    // quad_perm:[1,0,3,2]
  v += dpp_f64<0x4E>(v);    // quad_perm:[2,3,0,1]
  v += dpp_f64<0x141>(v);   // row_half_mirror: the other quad of the half row
  v += dpp_f64<0x140>(v);   // row_mirror: the other half row
  return v;
}
__device__ __forceinline__ double row16_max(double v) {
  v = fmax(v, dpp_f64<0xB1>(v));
  v = fmax(v, dpp_f64<0x4E>(v));
  v = fmax(v, dpp_f64<0x141>(v));
  v = fmax(v, dpp_f64<0x140>(v));
  return v;
}

// value of lane l (a constant once the caller's loops are unrolled) in every lane: two v_readlane_b32 into scalar registers
__device__ __forceinline__ double lane_bcast(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

// Dense SPD solve A x = b with the matrix distributed BY ROWS over the lanes of one wave: lane i < S holds row i of A in
// a[0..i] (what it holds beyond the diagonal is ignored, lanes >= S hold anything finite) and b_i. Right-looking Cholesky,
// fully unrolled: the pivot and the multipliers of column j are read from their lanes into scalar registers
// (v_readlane), every lane updates its own row -- no LDS, no barrier, S (S + 1) / 2 lane reads for the factor and as many
// again for the two substitutions. x comes back wave-uniform (the same in every lane). Returns whether every pivot was
// positive and finite. The elimination order of every element is that of the usual left-looking loop (k ascending).
template <int S>
__device__ __forceinline__ bool chol_solve_rows(double (&a)[S], double b, double (&x)[S]) {
  bool ok = true;
  double inv[S];
#pragma unroll
  for (int j = 0; j < S; ++j) {
    const double d = lane_bcast(a[j], j);
    ok = ok && (d > 0.0) && isfinite(d);
    const double r = rsqrt(d);
    inv[j] = r;
    a[j] *= r;   // column j of L (lane j: d * rsqrt(d) = L_jj)
#pragma unroll
    for (int k = j + 1; k < S; ++k) a[k] = fma(-a[j], lane_bcast(a[j], k), a[k]);
  }
#pragma unroll
  for (int j = 0; j < S; ++j) {   // L y = b
    const double y = lane_bcast(b, j) * inv[j];
    x[j] = y;
    b = fma(-a[j], y, b);
  }
#pragma unroll
  for (int i = S - 1; i >= 0; --i) {   // L^T x = y: L_ki is element i of lane k's row
    double acc = x[i];
#pragma unroll
    for (int k = i + 1; k < S; ++k) acc = fma(-lane_bcast(a[i], k), x[k], acc);
    x[i] = acc * inv[i];
  }
  return ok;
}

// Workgroup barrier for data handed over through LDS ONLY: waits for this wave's LDS operations, not for its global loads.
// __syncthreads() is a workgroup fence + s_barrier, and the fence drains EVERY outstanding memory operation (s_waitcnt vmcnt(0)):
// a prefetch of the next pass's observations issued above it is then waited for at the barrier, its whole latency exposed in
// every pass (k_rig_sweep_k2: 201 -> ... us at 8 x 2000 x 500). The "memory" clobber keeps the compiler from moving LDS accesses
// across it; registers a pending global load writes are still tracked (the wait comes at their first use).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// LDS accesses of one wave execute in order; the fence only stops the compiler from moving
// the staged-row reads above the writes of other lanes (no instruction is emitted).
__device__ __forceinline__ void wave_lds_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }

// stage one row per lane (16 doubles at row `lane`), 16-byte slots XOR-swizzled with (row & 7):
// the ds_write_b128 groups (8 consecutive lanes) and the ds_read_b64 operand reads (two 32-lane
// halves, each two consecutive rows) are both bank-conflict free.
__device__ __forceinline__ void stage_row(double* stage, int lane, const double* v) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    d2 val;
    val.x = v[2 * j];
    val.y = v[2 * j + 1];
    *reinterpret_cast<d2*>(&stage[lane * 16 + ((j ^ (lane & 7)) << 1)]) = val;
  }
}

// 16 MFMAs over the 64 staged rows: MFMA m consumes rows 4m..4m+3; lane l supplies component
// (l & 15) of row 4m + (l >> 4) as both the A[i][k] and the B[k][j] operand of the Gram product.
__device__ __forceinline__ void gram_rows(const double* stage, int lane, d4& acc0, d4& acc1) {
  const int c = lane & 15, sub = lane >> 4;
#pragma unroll
  for (int m = 0; m < 16; m += 2) {
    const int r0 = 4 * m + sub, r1 = 4 * (m + 1) + sub;
    const double a0 = stage[r0 * 16 + (((c >> 1) ^ (r0 & 7)) << 1) + (c & 1)];
    const double a1 = stage[r1 * 16 + (((c >> 1) ^ (r1 & 7)) << 1) + (c & 1)];
    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, a0, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, a1, acc1, 0, 0, 0);
  }
}

// The same contraction with its schedule pinned: all sixteen operand reads are issued first, then the sixteen matrix
// products, each waiting only for its own operand (the scheduler may not move anything across the barrier between the
// two groups). Inside the persistent kernel's round loop the compiler otherwise interleaves read / wait / product one
// operand at a time -- every product then sits behind a full LDS round trip (13.5 us instead of 8.4 for the two passes
// of a 500-point frame at four waves per SIMD).
__device__ __forceinline__ void gram_rows_ahead(const double* stage, int lane, d4& acc0, d4& acc1) {
  const int c = lane & 15, sub = lane >> 4;
  double a[16];
#pragma unroll
  for (int m = 0; m < 16; ++m) {
    const int r = 4 * m + sub;
    a[m] = stage[r * 16 + (((c >> 1) ^ (r & 7)) << 1) + (c & 1)];
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int m = 0; m < 16; m += 2) {
    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], a[m], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m + 1], a[m + 1], acc1, 0, 0, 0);
  }
  __builtin_amdgcn_sched_barrier(0);
}


// ---------------------------------------------------------------------------------------------
// Mailbox exchange between the ranks of one node (one process per GPU). Every rank owns a mailbox in
// its own HBM (uncached, exported with hipIpcGetMemHandle); peers map it and STORE their partial
// sums into it over xGMI; the owner only ever reads local memory. This replaces a library
// all-reduce for the two tiny (<= 1 KB) latency-bound reductions of an LM iteration and keeps the
// whole iteration inside one captured graph.
//   slot(kind, parity, rank): self-validating 8-byte words (two per payload double; the slot sizes
//   of the two kinds are chosen by the problem type); payload double i travels as
//   word 2i = {epoch32 : lo32(v)} and word 2i+1 = {epoch32 : hi32(v)}. An aligned 8-byte store is
//   single-copy atomic, so a word either carries the expected epoch and its data or it does not:
//   no fence, no separate flag, one store round trip + one load round trip per exchange.
//   kind 0: elimination sums (solve), kind 1: sweep statistics (decide); parity = epoch & 1
// A rank posts epoch e+2 of a kind only after it finished e+1, which needs every peer's e+1 post,
// which a peer makes only after it has read all of epoch e: two parities are enough.
// ---------------------------------------------------------------------------------------------
// (P2pDev, kP2pMaxRanks and the host-side mailbox management live in cc_common.hpp / cc_comm.cpp)

// word offset of slot (kind, parity, rank); kind 0 slots hold X.sw[0] words, kind 1 slots X.sw[1]
__device__ __forceinline__ int p2p_slot(const P2pDev& X, int kind, unsigned long long epoch, int rank) {
  const int base = kind == 0 ? 0 : 2 * kP2pMaxRanks * X.sw[0];
  return base + ((int)(epoch & 1ull) * kP2pMaxRanks + rank) * X.sw[kind];
}

// all threads of the block call; src holds payload doubles [first, first + n) of this rank's slot.
// No barrier inside.
__device__ inline void p2p_post(const P2pDev& X, int kind, unsigned long long epoch, int rank, int nranks,
                                const double* src, int n, int first = 0) {
  const int off = p2p_slot(X, kind, epoch, rank) + 2 * first;
  const unsigned long long tag = (epoch & 0xffffffffull) << 32;
  const int words = 2 * n;
  for (int idx = threadIdx.x; idx < words * nranks; idx += blockDim.x) {
    const int r = idx / words, w = idx - r * words;
    const unsigned long long bits = (unsigned long long)__double_as_longlong(src[w >> 1]);
    const unsigned long long half = (w & 1) ? (bits >> 32) : (bits & 0xffffffffull);
    __hip_atomic_store(X.box[r] + off + w, tag | half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// payload i of every rank's slot, added in rank order; *ok is cleared when the words do not show up
__device__ __forceinline__ double p2p_poll_sum(const P2pDev& X, int kind, unsigned long long epoch, int rank, int nranks,
                                               int i, long long t0, int* s_ok) {
  const unsigned long long tag = epoch & 0xffffffffull;
  const unsigned long long* base = X.box[rank] + 2 * i;
  unsigned long long lo[kP2pMaxRanks], hi[kP2pMaxRanks];
  for (;;) {
    bool all = true;
#pragma unroll
    for (int r = 0; r < kP2pMaxRanks; ++r) {
      if (r < nranks) {
        const unsigned long long* p = base + p2p_slot(X, kind, epoch, r);
        lo[r] = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        hi[r] = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
#pragma unroll
    for (int r = 0; r < kP2pMaxRanks; ++r)
      if (r < nranks) all = all && (lo[r] >> 32) == tag && (hi[r] >> 32) == tag;
    if (all) break;
    if (wall_clock64() - t0 > kP2pTimeoutTicks) { *s_ok = 0; break; }
    __builtin_amdgcn_s_sleep(1);
  }
  double a = 0.0;
#pragma unroll
  for (int r = 0; r < kP2pMaxRanks; ++r)
    if (r < nranks) {
      const double v = __longlong_as_double((long long)((hi[r] << 32) | (lo[r] & 0xffffffffull)));
      a = r == 0 ? v : a + v;
    }
  return a;
}

// Thread i < n returns payload i of every rank's slot summed in rank order (identical on all ranks);
// all threads of the block call. *s_ok (one LDS int) ends 0 when a peer's words did not arrive in
// time; the return value is then meaningless. Contains two barriers.
__device__ inline double p2p_collect(const P2pDev& X, int kind, unsigned long long epoch, int rank, int nranks,
                                     int n, int* s_ok) {
  if (threadIdx.x == 0) *s_ok = 1;
  __syncthreads();
  double a = 0.0;
  if ((int)threadIdx.x < n) a = p2p_poll_sum(X, kind, epoch, rank, nranks, threadIdx.x, wall_clock64(), s_ok);
  __syncthreads();
  return a;
}

// Same for n > blockDim.x payload words: the rank-ordered sums are written to dst[0..n) (global or LDS).
__device__ inline void p2p_collect_to(const P2pDev& X, int kind, unsigned long long epoch, int rank, int nranks,
                                      int n, double* dst, int* s_ok) {
  if (threadIdx.x == 0) *s_ok = 1;
  __syncthreads();
  const long long t0 = wall_clock64();
  for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = p2p_poll_sum(X, kind, epoch, rank, nranks, i, t0, s_ok);
  __syncthreads();
}

}  // namespace cc
