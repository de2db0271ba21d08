// cc_device.hpp -- device helpers shared by the intrinsics and the rig kernels (gfx950).
#pragma once
#include "cc_common.hpp"

namespace cc {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

constexpr int kStageDoublesPerWave = 64 * 16;  // 64 observations x one row (u or v) x 16 components

__device__ __forceinline__ int tri(int i, int j) { return i * (i + 1) / 2 + j; }  // packed lower, i >= j

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// LDS accesses of one wave execute in order; the fence only stops the compiler from moving
// the staged-row reads above the writes of other lanes (no instruction is emitted).
__device__ __forceinline__ void wave_lds_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }

// stage one row per lane (16 doubles at row `lane`), 16-byte slots XOR-swizzled with (row & 7):
// the ds_write_b128 groups (8 consecutive lanes) and the ds_read_b64 operand reads (two 32-lane
// halves, each two consecutive rows) are both bank-conflict free.
__device__ __forceinline__ void stage_row(double* stage, int lane, const double* v) {
#if defined(CC_ABLATE) && CC_ABLATE == 2
  asm volatile("" ::"v"(v[0] + v[5] + v[9] + v[12] + v[15]));
  return;
#endif
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
#if defined(CC_ABLATE) && CC_ABLATE == 1
  asm volatile("" ::"v"(stage[lane]));
  return;
#endif
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


// ---------------------------------------------------------------------------------------------
// Mailbox exchange between the ranks of one node (one process per GPU). Every rank owns a mailbox in
// its own HBM (uncached, exported with hipIpcGetMemHandle); peers map it and STORE their partial
// sums into it over xGMI, then raise a flag; the owner only ever reads local memory. This replaces
// a library all-reduce for the two tiny (<= 1 KB) latency-bound reductions of an LM iteration and
// keeps the whole iteration inside one captured graph.
//   slot(kind, parity, rank): 128 doubles = 127 payload + 1 flag word holding the epoch
//   kind 0: elimination sums (solve), kind 1: sweep statistics (decide); parity = epoch & 1
// A rank posts epoch e+2 of a kind only after it finished e+1, which needs every peer's e+1 post,
// which a peer makes only after it has read all of epoch e: two parities are enough.
// ---------------------------------------------------------------------------------------------
constexpr int kP2pMaxRanks = 8;
constexpr int kP2pSlot = 128;
constexpr int kP2pFlag = 127;
constexpr int kP2pDoubles = 2 * 2 * kP2pMaxRanks * kP2pSlot;
constexpr long long kP2pTimeoutTicks = 200000000LL;  // 2 s of the 100 MHz wall clock

struct P2pDev {
  double* box[kP2pMaxRanks];   // box[r]: rank r's mailbox as mapped here (box[rank] is local)
  unsigned long long* seq;     // [2] last completed epoch per kind (device memory of this rank)
  int32_t on;
  int32_t pad;
};

__device__ __forceinline__ int p2p_slot(int kind, unsigned long long epoch, int rank) {
  return ((kind * 2 + (int)(epoch & 1ull)) * kP2pMaxRanks + rank) * kP2pSlot;
}

// all threads of the block call; src (LDS or registers spilled to LDS) holds n <= 127 doubles
__device__ inline void p2p_post(const P2pDev& X, int kind, unsigned long long epoch, int rank, int nranks,
                                const double* src, int n) {
  const int off = p2p_slot(kind, epoch, rank);
  for (int idx = threadIdx.x; idx < n * nranks; idx += blockDim.x) {
    const int r = idx / n, i = idx - r * n;
    __hip_atomic_store(X.box[r] + off + i, src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __threadfence_system();
  __syncthreads();
  if ((int)threadIdx.x < nranks)
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(X.box[threadIdx.x] + off + kP2pFlag), epoch,
                       __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// all threads of the block call; returns false when a peer did not show up in time (s_ok: one LDS int)
__device__ inline bool p2p_wait(const P2pDev& X, int kind, unsigned long long epoch, int rank, int nranks, int* s_ok) {
  if (threadIdx.x == 0) *s_ok = 1;
  __syncthreads();
  if ((int)threadIdx.x < nranks) {
    const unsigned long long* flag =
        reinterpret_cast<const unsigned long long*>(X.box[rank] + p2p_slot(kind, epoch, threadIdx.x) + kP2pFlag);
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != epoch) {
      if (wall_clock64() - t0 > kP2pTimeoutTicks) { *s_ok = 0; break; }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
  return *s_ok != 0;
}

// payload word i of every rank's slot, summed in rank order (identical on all ranks)
__device__ __forceinline__ double p2p_sum(const P2pDev& X, int kind, unsigned long long epoch, int rank, int nranks, int i) {
  double v[kP2pMaxRanks];
#pragma unroll
  for (int r = 0; r < kP2pMaxRanks; ++r)
    v[r] = r < nranks ? __hip_atomic_load(X.box[rank] + p2p_slot(kind, epoch, r) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0.0;
  double a = v[0];
#pragma unroll
  for (int r = 1; r < kP2pMaxRanks; ++r) a += v[r];
  return a;
}

}  // namespace cc
