// cc_intrinsics_dev.hpp -- device-side definitions shared by the two forms of the intrinsics solver (gfx950):
// the two-kernels-per-iteration path (cc_intrinsics.hip) and the persistent per-solve kernel (cc_intrinsics_persist.hip).
#pragma once
#include "cc_common.hpp"
#include "cc_device.hpp"

namespace cc {

constexpr int kSweepThreads = 256;
constexpr int kSweepLdsBytes = (4 * kStageDoublesPerWave + 256) * 8;  // staging (reused for the block reduction) + prologue scratch
constexpr int kStatsCols = 4;      // cost, q_model, step2, xnorm2
constexpr int kYStride = 64;       // Y = A_pp^-1 [H_ps | g_p] (6 x 10) per frame
constexpr int kPartialCols = 80;   // see k_intr_decide_elim
constexpr int kElimMaxBlocks = 64;
constexpr int kVecSolve = kPartialCols + 32;  // + one gmax slot per rank

enum { ST_COST = 0, ST_QMODEL = 1, ST_STEP2 = 2, ST_XNORM2 = 3 };
// partial / vec_solve columns
enum { PC_S = 0, PC_B = 45, PC_HDIAG = 54, PC_FAIL = 63, PC_GS = 64, PC_GMAXP = 73 };

struct IntrDev {
  int64_t F, N;
  int32_t T, tpad_;   // sweep workgroups (tiles of observations) per frame: 1, or more when there are too few frames to fill the chip
  const float* uv;
  const float* xyz;
  const int64_t* off;
  double* intr;       // [2][16]
  const double* init_intr;   // [16]   state of the last set_state (restart source)
  const double* init_pose;   // [F][8]
  double* pose;       // [2][F][8]
  double* blocks;     // [2][F][T][256]: per-tile partial Gram blocks (the consumers add the T tiles)
  double* stats;      // [F*T][4]
  double* hd0;        // [F*T][16] diag of H_ss (per tile) at the initial point (Jacobi scaling)
  double* sp;         // [F][8]  Jacobi scale of the pose block
  double* Y;          // [F][64]
  double* partial;    // [kElimMaxBlocks][80]
  double* vec_solve;  // [112] reduced elimination sums (all-reduced across ranks)
  double* vec_decide; // [16]  reduced sweep statistics (all-reduced across ranks)
  double* ds;         // [16] scaled shared step
  double* ss;         // [16] Jacobi scale of the shared block
  LmCtl* ctl;         // read by sweep / decide_elim, written by the solve step
  LmCtl* ctl_next;    // written by the solve step (RCCL route: also by block 0 of decide_elim); read by the host
  unsigned* arrive;   // [1] blocks of decide_elim that have stored their partial row (last-block-done)
  unsigned long long* pub_seq;   // [1] device: chunks published so far
  unsigned long long* host_pub;  // pinned host memory: [0] sequence word, [2..19] copy of the control block
  LmOpts* opts;
  cc_iteration* log;
  int32_t log_cap;
  uint32_t mask;
  int32_t rank, nranks;
  P2pDev x;           // mailbox exchange (cc_device.hpp); x.on == 0 on a single GPU / with RCCL
};

// ---------------------------------------------------------------------------------------------
// Per-observation model: residual and the two rows of [J_intr(9) J_pose(6) r].
// Restates ReprojectionError::operator() + DistortPixels/DistortNormalized
// (calibrator.cpp:70-95,183-219) with analytic derivatives.
// k = fx fy px py k1 k2 p1 p2 k3 (calibrator.cpp:168-179); R = R(q/|q|).
// ---------------------------------------------------------------------------------------------
struct ObsCommon {  // per-observation quantities shared by the u-row and the v-row
  double a0, a1, a2, x, y, iz, xy, r2, r4, r6, ax, ay, xd, yd, dxx, dxy, dyy;
};

__device__ __forceinline__ void obs_common(const double* k, const double* R, const double* t,
                                           double X0, double X1, double X2, ObsCommon& c) {
  c.a0 = R[0] * X0 + R[1] * X1 + R[2] * X2;
  c.a1 = R[3] * X0 + R[4] * X1 + R[5] * X2;
  c.a2 = R[6] * X0 + R[7] * X1 + R[8] * X2;
  const double xc = c.a0 + t[0], yc = c.a1 + t[1], zc = c.a2 + t[2];
  c.iz = 1.0 / zc;
  c.x = xc * c.iz;
  c.y = yc * c.iz;
  const double k1 = k[4], k2 = k[5], p1 = k[6], p2 = k[7], k3 = k[8];
  const double xx = c.x * c.x, yy = c.y * c.y;
  c.xy = c.x * c.y;
  c.r2 = xx + yy;
  c.r4 = c.r2 * c.r2;
  c.r6 = c.r4 * c.r2;
  const double m = 1.0 + k1 * c.r2 + k2 * c.r4 + k3 * c.r6;
  c.ax = c.r2 + 2.0 * xx;
  c.ay = c.r2 + 2.0 * yy;
  c.xd = c.x * m + 2.0 * p1 * c.xy + p2 * c.ax;
  c.yd = c.y * m + 2.0 * p2 * c.xy + p1 * c.ay;
  const double mp = k1 + 2.0 * k2 * c.r2 + 3.0 * k3 * c.r4;
  c.dxx = m + 2.0 * mp * xx + 2.0 * p1 * c.y + 6.0 * p2 * c.x;
  c.dxy = 2.0 * mp * c.xy + 2.0 * p1 * c.x + 2.0 * p2 * c.y;
  c.dyy = m + 2.0 * mp * yy + 2.0 * p2 * c.x + 6.0 * p1 * c.y;
}

// row of the u residual: d (fx xd + px - u) / d [fx fy px py k1 k2 p1 p2 k3 | rot(3) t(3)], then r.
// w = 1: the row; w = 0 (a lane beyond the frame's last observation; it evaluates a real observation of the frame, so everything
// is finite): a row of zeros. The weight rides on the row's factors -- fx w takes the ten entries that carry fx, three more
// products take the rest -- where zeroing the 14 entries afterwards took 28 conditional moves (round 5; with w = 1 the bits
// are the same as before). (Written as selects -- `valid ? fx : 0` -- the compiler builds a divergent region around the row and
// parks all sixteen entries in scratch: 90 spilled registers at four teams.)
// Coordinates held constant (SubsetManifold, calibrator.cpp:305-312) are NOT zeroed here, nine selects a row: their rows and
// columns of the frame's Gram block are zeroed once, where the block is assembled (gram_entry_held) -- sums of zeros are zero.
__device__ __forceinline__ void row_u(const double* k, const ObsCommon& c, double u, double* v, double w = 1.0) {
  const double fx = k[0] * w;
  v[15] = (k[0] * c.xd + k[2] - u) * w;
  v[0] = c.xd * w; v[1] = 0.0; v[2] = w; v[3] = 0.0;
  const double fxx = fx * c.x;
  v[4] = fxx * c.r2; v[5] = fxx * c.r4; v[6] = fx * 2.0 * c.xy; v[7] = fx * c.ax; v[8] = fxx * c.r6;
  const double b0 = fx * c.dxx * c.iz, b1 = fx * c.dxy * c.iz, b2 = -(b0 * c.x + b1 * c.y);
  v[9] = 2.0 * (b2 * c.a1 - b1 * c.a2); v[10] = 2.0 * (b0 * c.a2 - b2 * c.a0); v[11] = 2.0 * (b1 * c.a0 - b0 * c.a1);
  v[12] = b0; v[13] = b1; v[14] = b2;
}

__device__ __forceinline__ void row_v(const double* k, const ObsCommon& c, double vm, double* v, double w = 1.0) {
  const double fy = k[1] * w;
  v[15] = (k[1] * c.yd + k[3] - vm) * w;
  v[0] = 0.0; v[1] = c.yd * w; v[2] = 0.0; v[3] = w;
  const double fyy = fy * c.y;
  v[4] = fyy * c.r2; v[5] = fyy * c.r4; v[6] = fy * c.ay; v[7] = fy * 2.0 * c.xy; v[8] = fyy * c.r6;
  const double b0 = fy * c.dxy * c.iz, b1 = fy * c.dyy * c.iz, b2 = -(b0 * c.x + b1 * c.y);
  v[9] = 2.0 * (b2 * c.a1 - b1 * c.a2); v[10] = 2.0 * (b0 * c.a2 - b2 * c.a0); v[11] = 2.0 * (b1 * c.a0 - b0 * c.a1);
  v[12] = b0; v[13] = b1; v[14] = b2;
}
// entry e = 16 row + col of a frame's 16 x 16 Gram block: does it belong to a coordinate held constant (mask bit j = intrinsic j)?
__device__ __forceinline__ bool gram_entry_held(uint32_t mask, int e) { return (((mask >> (e >> 4)) | (mask >> (e & 15))) & 1u) != 0; }

// Hands the control block to the host without a copy engine in the way: payload words first, then the
// sequence word the host spins on (system-scope stores into pinned host memory; one thread).
__device__ __forceinline__ void publish_to_host(const IntrDev& P, const LmCtl& c) {
  if (!P.host_pub) return;
  const unsigned long long* w = reinterpret_cast<const unsigned long long*>(&c);
#pragma unroll
  for (int i = 0; i < (int)(sizeof(LmCtl) / 8); ++i)
    __hip_atomic_store(P.host_pub + 2 + i, w[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  const unsigned long long seq = *P.pub_seq + 1ull;
  *P.pub_seq = seq;
  __hip_atomic_store(P.host_pub, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace cc
