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


}  // namespace cc
