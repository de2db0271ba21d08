// cc_intrinsics.hip -- single-camera intrinsics bundle adjustment on MI355X (gfx950).
//
// Replaces the ceres::Problem/ceres::Solve block of Calibrator::Optimize
// (/root/reference/src/calibrator.cpp:236-324). Four kernels per LM iteration:
//
//   elim   : per frame, damped 6x6 pose block -> Cholesky, Z = L^-1 [H_ps | g_p]; partial sums of
//            the reduced (Schur) 9x9 system over frames.
//   solve  : reduce the partials, add the LM diagonal, 9x9 Cholesky solve -> scaled shared step.
//   sweep  : one workgroup per frame. Prologue back-substitutes the frame's pose step, forms the
//            candidate point (QuaternionManifold::Plus) and the frame's model-cost term; the main
//            loop evaluates pinhole + radial-tangential projection and the analytic 2x15 Jacobian
//            per observation (one lane = one observation), stages [J r] rows through LDS and
//            contracts them with v_mfma_f64_16x16x4_f64 into the frame's 16x16 Gram block
//            G_f = sum_rows [J r]^T [J r]  (= H_ss,f H_sp,f g_s,f / H_pp,f g_p,f / 2 cost_f).
//   decide : reduce the per-frame statistics, Ceres trust-region logic on one thread.
//
// HBM layout (per handle): uv float2[N], xyz float[3N] exactly as the API hands them over
// (20 B / observation, read once per sweep, coalesced: lane i of a wave reads observation
// base+i); poses double[2][F][8]; Gram blocks double[2][F][256] (2 KiB per frame, double
// buffered: accepted point / candidate); per-frame stats double[F][32]; Z,L double[F][96].
#include "cc_common.hpp"

#include <algorithm>
#include <chrono>
#include <vector>

namespace cc {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

constexpr int kSweepThreads = 256;
constexpr int kStageDoublesPerWave = 64 * 32;                       // 64 observations x 2 rows x 16
constexpr int kSweepLdsBytes = 4 * kStageDoublesPerWave * 8 + 1024; // 4 waves of staging + scratch
constexpr int kStatsCols = 32;
constexpr int kZLStride = 96;
constexpr int kElimMaxBlocks = 64;

// stats columns written by the sweep, reduced by decide
enum { ST_COST = 0, ST_QMODEL = 1, ST_STEP2 = 2, ST_XNORM2 = 3, ST_GMAXP = 4, ST_GS = 7, ST_HDIAG = 16 };

struct IntrDev {
  int64_t F, N;
  const float* uv;
  const float* xyz;
  const int64_t* off;
  double* intr;     // [2][16]
  double* pose;     // [2][F][8]
  double* blocks;   // [2][F][256]
  double* stats;    // [F][32]
  double* sp;       // [F][8]  Jacobi scale of the pose block
  double* ZL;       // [F][96] Z (6x10) then packed lower L (21)
  double* partial;  // [kElimMaxBlocks][64]
  double* vec_solve;   // [64]  reduced elimination sums (all-reduced across ranks)
  double* vec_decide;  // [64]  reduced sweep statistics (all-reduced across ranks)
  LmState* state;
  LmOpts* opts;
  cc_iteration* log;
  int32_t log_cap;
  uint32_t mask;
  int32_t rank, nranks;
};

// ---------------------------------------------------------------------------------------------
// Per-observation model: residual and the two rows of [J_intr(9) J_pose(6) r].
// Restates ReprojectionError::operator() + DistortPixels/DistortNormalized
// (calibrator.cpp:70-95,183-219) with analytic derivatives.
// k = fx fy px py k1 k2 p1 p2 k3 (calibrator.cpp:168-179); R = R(q/|q|).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void intr_rows(const double* k, const double* R, const double* t,
                                          double X0, double X1, double X2, double u, double v,
                                          uint32_t mask, double* vu, double* vv) {
  const double a0 = R[0] * X0 + R[1] * X1 + R[2] * X2;
  const double a1 = R[3] * X0 + R[4] * X1 + R[5] * X2;
  const double a2 = R[6] * X0 + R[7] * X1 + R[8] * X2;
  const double xc = a0 + t[0], yc = a1 + t[1], zc = a2 + t[2];
  const double iz = 1.0 / zc;
  const double x = xc * iz, y = yc * iz;
  const double fx = k[0], fy = k[1], px = k[2], py = k[3];
  const double k1 = k[4], k2 = k[5], p1 = k[6], p2 = k[7], k3 = k[8];
  const double xx = x * x, yy = y * y, xy = x * y;
  const double r2 = xx + yy, r4 = r2 * r2, r6 = r4 * r2;
  const double m = 1.0 + k1 * r2 + k2 * r4 + k3 * r6;
  const double ax = r2 + 2.0 * xx, ay = r2 + 2.0 * yy;
  const double xd = x * m + 2.0 * p1 * xy + p2 * ax;
  const double yd = y * m + 2.0 * p2 * xy + p1 * ay;
  vu[15] = fx * xd + px - u;
  vv[15] = fy * yd + py - v;
  vu[0] = xd;  vu[1] = 0.0; vu[2] = 1.0; vu[3] = 0.0;
  vv[0] = 0.0; vv[1] = yd;  vv[2] = 0.0; vv[3] = 1.0;
  const double fxx = fx * x, fyy = fy * y;
  vu[4] = fxx * r2; vu[5] = fxx * r4; vu[6] = fx * 2.0 * xy; vu[7] = fx * ax; vu[8] = fxx * r6;
  vv[4] = fyy * r2; vv[5] = fyy * r4; vv[6] = fy * ay; vv[7] = fy * 2.0 * xy; vv[8] = fyy * r6;
  const double mp = k1 + 2.0 * k2 * r2 + 3.0 * k3 * r4;
  const double dxx = m + 2.0 * mp * xx + 2.0 * p1 * y + 6.0 * p2 * x;
  const double dxy = 2.0 * mp * xy + 2.0 * p1 * x + 2.0 * p2 * y;
  const double dyy = m + 2.0 * mp * yy + 2.0 * p2 * x + 6.0 * p1 * y;
  const double b00 = fx * dxx * iz, b01 = fx * dxy * iz, b02 = -(b00 * x + b01 * y);
  const double b10 = fy * dxy * iz, b11 = fy * dyy * iz, b12 = -(b10 * x + b11 * y);
  vu[9] = 2.0 * (b02 * a1 - b01 * a2); vu[10] = 2.0 * (b00 * a2 - b02 * a0); vu[11] = 2.0 * (b01 * a0 - b00 * a1);
  vu[12] = b00; vu[13] = b01; vu[14] = b02;
  vv[9] = 2.0 * (b12 * a1 - b11 * a2); vv[10] = 2.0 * (b10 * a2 - b12 * a0); vv[11] = 2.0 * (b11 * a0 - b10 * a1);
  vv[12] = b10; vv[13] = b11; vv[14] = b12;
#pragma unroll
  for (int c = 0; c < 9; ++c)
    if (mask & (1u << c)) { vu[c] = 0.0; vv[c] = 0.0; }  // SubsetManifold (calibrator.cpp:305-312)
}

// ---------------------------------------------------------------------------------------------
// sweep: one workgroup (4 waves) per frame
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kSweepThreads, 2) void k_intr_sweep(IntrDev P) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* s_stage = reinterpret_cast<double*>(smem_raw);                 // [4][2048]
  double* sm = s_stage + 4 * kStageDoublesPerWave;                        // [128] scratch
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t f = blockIdx.x;
  LmState* st = P.state;
  if (st->done) return;
  const int phase = st->phase;
  if (phase != 0 && !st->step_valid) return;
  const int cur = st->cur, dst = phase == 0 ? cur : (cur ^ 1);
  const double* intr_cur = P.intr + cur * 16;
  const double* pose_cur = P.pose + ((size_t)cur * P.F + f) * 8;

  // ---- prologue: candidate point of this frame -------------------------------------------
  // sm[0..14] unscaled step (9 shared, 6 pose); sm[16..24] R; sm[25..27] t; sm[28..36] intr;
  // sm[38] step^2; sm[39] |x_cand|^2 (pose part); sm[40..45] u
  if (phase != 0 && tid < 6) {
    const double* Z = P.ZL + f * kZLStride + tid * 10;
    double u = Z[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) u += Z[j] * st->ds[j];
    sm[40 + tid] = u;
  }
  __syncthreads();
  if (tid == 0) {
    double q[4], t[3], kk[9];
    for (int i = 0; i < 4; ++i) q[i] = pose_cur[i];
    for (int i = 0; i < 3; ++i) t[i] = pose_cur[4 + i];
    for (int i = 0; i < 9; ++i) kk[i] = intr_cur[i];
    double step2 = 0.0;
    if (phase != 0) {
      const double* L = P.ZL + f * kZLStride + 60;  // packed lower: L[i(i+1)/2 + j]
      double xs[6];
      for (int i = 5; i >= 0; --i) {
        double s = sm[40 + i];
        for (int k2 = i + 1; k2 < 6; ++k2) s -= L[k2 * (k2 + 1) / 2 + i] * xs[k2];
        xs[i] = s / L[i * (i + 1) / 2 + i];
      }
      double dp[6];
      for (int i = 0; i < 6; ++i) { dp[i] = -xs[i] * P.sp[f * 8 + i]; sm[9 + i] = dp[i]; }
      for (int j = 0; j < 9; ++j) {
        const double d = (P.mask & (1u << j)) ? 0.0 : st->ds[j] * st->ss[j];
        sm[j] = d;
        kk[j] += d;
      }
      double qn[4];
      quat_plus(q, dp, qn);
      for (int i = 0; i < 4; ++i) { const double d = qn[i] - q[i]; step2 += d * d; q[i] = qn[i]; }
      for (int i = 0; i < 3; ++i) { const double tn = t[i] + dp[3 + i]; const double d = tn - t[i]; step2 += d * d; t[i] = tn; }
      double* pose_dst = P.pose + ((size_t)dst * P.F + f) * 8;
      for (int i = 0; i < 4; ++i) pose_dst[i] = q[i];
      for (int i = 0; i < 3; ++i) pose_dst[4 + i] = t[i];
      if (f == 0) for (int i = 0; i < 9; ++i) P.intr[dst * 16 + i] = kk[i];
    }
    double R[9];
    quat_to_R(q, R);
    for (int i = 0; i < 9; ++i) sm[16 + i] = R[i];
    for (int i = 0; i < 3; ++i) sm[25 + i] = t[i];
    for (int i = 0; i < 9; ++i) sm[28 + i] = kk[i];
    sm[38] = step2;
    sm[39] = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3] + t[0] * t[0] + t[1] * t[1] + t[2] * t[2];
  }
  __syncthreads();

  // model-cost term of this frame: q_f = d^T g_f + 1/2 d^T H_f d over the frame's 15x15 block at
  // the accepted point (Ceres: model_cost_change = -(J d)^T (r + J d / 2))
  double qterm = 0.0;
  if (phase != 0) {
    const int a = tid >> 4, b = tid & 15;
    const double g = P.blocks[((size_t)cur * P.F + f) * 256 + tid];
    if (a < 15) qterm = b < 15 ? 0.5 * sm[a] * g * sm[b] : sm[a] * g;
  }

  double R[9], tt[3], kk[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) R[i] = rfl(sm[16 + i]);
#pragma unroll
  for (int i = 0; i < 3; ++i) tt[i] = rfl(sm[25 + i]);
#pragma unroll
  for (int i = 0; i < 9; ++i) kk[i] = rfl(sm[28 + i]);
  const uint32_t mask = P.mask;

  // ---- main loop: 64 observations per wave per pass ----------------------------------------
  const int64_t s0 = P.off[f], s1 = P.off[f + 1];
  const int npass = (int)((s1 - s0 + kSweepThreads - 1) / kSweepThreads);
  double* stage = s_stage + wave * kStageDoublesPerWave;
  const float2* uv2 = reinterpret_cast<const float2*>(P.uv);
  d4 acc = {0.0, 0.0, 0.0, 0.0};
  for (int p = 0; p < npass; ++p) {
    const int64_t idx = s0 + (int64_t)p * kSweepThreads + tid;
    double vu[16], vv[16];
    if (idx < s1) {
      const float2 m = uv2[idx];
      const float X0 = P.xyz[idx * 3], X1 = P.xyz[idx * 3 + 1], X2 = P.xyz[idx * 3 + 2];
      intr_rows(kk, R, tt, (double)X0, (double)X1, (double)X2, (double)m.x, (double)m.y, mask, vu, vv);
    } else {
#pragma unroll
      for (int c = 0; c < 16; ++c) { vu[c] = 0.0; vv[c] = 0.0; }
    }
    // stage rows: lane o owns doubles [32 o, 32 o + 32); 16-byte slot j is XOR-swizzled with
    // (o & 15) so both the b128 writes and the b64 MFMA-operand reads are bank-conflict free.
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      d2 val;
      if (j < 8) { val.x = vu[2 * j]; val.y = vu[2 * j + 1]; }
      else { val.x = vv[2 * j - 16]; val.y = vv[2 * j - 15]; }
      *reinterpret_cast<d2*>(&stage[lane * 32 + ((j ^ (lane & 15)) << 1)]) = val;
    }
    __syncthreads();
    // MFMA m consumes rows 4m..4m+3 = observations 2m, 2m+1; lane l supplies component (l & 15)
    // of row (l >> 4): A[i][k] and B[k][j] coincide for the Gram product.
    const int pp = lane & 31, jj = pp >> 1, hh = pp & 1;
#pragma unroll 8
    for (int m = 0; m < 32; ++m) {
      const int o = 2 * m + (lane >> 5);
      const double a = stage[o * 32 + ((jj ^ (o & 15)) << 1) + hh];
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, acc, 0, 0, 0);
    }
    __syncthreads();
  }

  // ---- cross-wave reduction of the 16x16 block + model-cost term ---------------------------
  // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
  double* red = s_stage;  // [4][256] + [256] q terms
#pragma unroll
  for (int r = 0; r < 4; ++r) red[wave * 256 + ((lane >> 4) + 4 * r) * 16 + (lane & 15)] = acc[r];
  red[1024 + tid] = qterm;
  __syncthreads();
  const double g = red[tid] + red[256 + tid] + red[512 + tid] + red[768 + tid];
  P.blocks[((size_t)dst * P.F + f) * 256 + tid] = g;
  // q_f: tree over 256 terms (deterministic order)
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) red[1024 + tid] += red[1024 + tid + s];
    __syncthreads();
  }
  double* G = s_stage + 2048;  // full block for the stats threads
  G[tid] = g;
  __syncthreads();
  if (tid < kStatsCols) {
    double v = 0.0;
    if (tid == ST_COST) v = 0.5 * G[255];
    else if (tid == ST_QMODEL) v = red[1024];
    else if (tid == ST_STEP2) v = sm[38];
    else if (tid == ST_XNORM2) v = sm[39];
    else if (tid == ST_GMAXP) {
      for (int i = 0; i < 6; ++i) v = fmax(v, fabs(G[(9 + i) * 16 + 15]));
    } else if (tid >= ST_GS && tid < ST_GS + 9) v = G[(tid - ST_GS) * 16 + 15];
    else if (tid >= ST_HDIAG && tid < ST_HDIAG + 9) v = G[(tid - ST_HDIAG) * 17];
    P.stats[f * kStatsCols + tid] = v;
  }
  if (phase == 0 && tid < 6)
    P.sp[f * 8 + tid] = P.opts->jacobi_scaling ? 1.0 / (1.0 + sqrt(G[(9 + tid) * 17])) : 1.0;
}

// ---------------------------------------------------------------------------------------------
// decide: MODE 0 = reduce + decide (single GPU), 1 = reduce only (-> vec_decide), 2 = decide only
// vec_decide layout: [0..31] column sums of stats (col ST_GMAXP unused), [32 + rank] local gmax_p
// ---------------------------------------------------------------------------------------------
constexpr int kDecideThreads = 1024;
constexpr int kStateDoubles = (int)(sizeof(LmState) / sizeof(double));
static_assert(sizeof(LmState) % sizeof(double) == 0, "LmState must be a whole number of doubles");

template <int MODE>
__global__ __launch_bounds__(kDecideThreads) void k_intr_decide(IntrDev P) {
  __shared__ double red[32][kStatsCols];
  __shared__ double sv[64];
  __shared__ LmState sst;  // the state machine works on an LDS copy: one global round trip each way
  const int tid = threadIdx.x;
  if (tid < kStateDoubles) reinterpret_cast<double*>(&sst)[tid] = reinterpret_cast<const double*>(P.state)[tid];
  __syncthreads();
  if (sst.done) return;
  LmState* st = &sst;
  if (MODE != 2) {
    const bool skip = st->phase != 0 && !st->step_valid;  // sweep did not run: stats are stale
    const int col = tid & 31, grp = tid >> 5;
    double a = 0.0;
    if (!skip) {
      // 32 row groups; 8 independent loads in flight per thread
      const double* base = P.stats + col;
      if (col == ST_GMAXP) {
        for (int64_t f = grp; f < P.F; f += 32) a = fmax(a, base[f * kStatsCols]);
      } else {
        double acc8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int64_t f = grp;
        for (; f + 7 * 32 < P.F; f += 8 * 32) {
#pragma unroll
          for (int u = 0; u < 8; ++u) acc8[u] += base[(f + u * 32) * kStatsCols];
        }
        for (int u = 0; f < P.F; f += 32, ++u) acc8[u] += base[f * kStatsCols];
        a = ((acc8[0] + acc8[1]) + (acc8[2] + acc8[3])) + ((acc8[4] + acc8[5]) + (acc8[6] + acc8[7]));
      }
    }
    red[grp][col] = a;
    if (tid < 64) sv[tid] = 0.0;
    __syncthreads();
    if (tid < 32) {
      double v = 0.0;
      if (tid == ST_GMAXP) {
        for (int g2 = 0; g2 < 32; ++g2) v = fmax(v, red[g2][tid]);
        sv[32 + P.rank] = v;  // per-rank slot: a sum all-reduce then carries the max
      } else {
        for (int g2 = 0; g2 < 32; ++g2) v += red[g2][tid];
        sv[tid] = v;
      }
    }
    __syncthreads();
    if (MODE == 1) {
      if (tid < 64) P.vec_decide[tid] = sv[tid];
      return;
    }
  } else {
    if (tid < 64) sv[tid] = P.vec_decide[tid];
    __syncthreads();
  }
  if (tid == 0) {
    const double* V = sv;
    const LmOpts o = *P.opts;
    double gmax_p = 0.0;
    for (int r = 0; r < P.nranks && r < 32; ++r) gmax_p = fmax(gmax_p, V[32 + r]);
    if (st->phase == 0) {
      const double* k = P.intr + st->cur * 16;
      double xn2 = V[ST_XNORM2], gmax = gmax_p;
      for (int i = 0; i < 9; ++i) {
        xn2 += k[i] * k[i];
        if (!(P.mask & (1u << i))) gmax = fmax(gmax, fabs(V[ST_GS + i]));
        st->ss[i] = o.jacobi_scaling ? 1.0 / (1.0 + sqrt(V[ST_HDIAG + i])) : 1.0;
      }
      st->sweeps = 1;
      lm_init(*st, o, V[ST_COST], sqrt(xn2), gmax);
    } else {
      const int cand = st->cur ^ 1;
      const double* kc = P.intr + cand * 16;
      const double* k0 = P.intr + st->cur * 16;
      double step2 = V[ST_STEP2], xn2 = V[ST_XNORM2], gmax = gmax_p;
      if (st->step_valid) {
        st->sweeps++;
        for (int i = 0; i < 9; ++i) {
          const double d = kc[i] - k0[i];
          step2 += d * d;
          xn2 += kc[i] * kc[i];
          if (!(P.mask & (1u << i))) gmax = fmax(gmax, fabs(V[ST_GS + i]));
        }
      }
      lm_decide(*st, o, P.log, P.log_cap, V[ST_COST], V[ST_QMODEL], step2, xn2, gmax);
    }
  }
  __syncthreads();
  if (tid < kStateDoubles) reinterpret_cast<double*>(P.state)[tid] = reinterpret_cast<const double*>(&sst)[tid];
}

// ---------------------------------------------------------------------------------------------
// elim: 16 lanes per frame. Output slots (64): [0..44] upper triangle of the reduced 9x9 system
// (row-major pairs j<=k), [45..53] reduced rhs, [54..62] diag of the scaled H_ss, [63] failures.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int tri(int i, int j) { return i * (i + 1) / 2 + j; }  // i >= j

__global__ __launch_bounds__(256) void k_intr_elim(IntrDev P) {
  __shared__ double Zs[16][64];
  __shared__ double red[16][64];
  __shared__ double s_ss[16];
  __shared__ unsigned char pj[48], pk[48];
  LmState* st = P.state;
  if (st->done) return;
  const int tid = threadIdx.x, g = tid >> 4, l = tid & 15;
  const int cur = st->cur;
  const double radius = st->radius;
  const double mn = P.opts->min_lm_diagonal, mx = P.opts->max_lm_diagonal;
  if (tid < 16) s_ss[tid] = tid < 9 ? st->ss[tid] : 0.0;
  if (tid == 0) {
    int o = 0;
    for (int j = 0; j < 9; ++j)
      for (int k = j; k < 9; ++k) { pj[o] = (unsigned char)j; pk[o] = (unsigned char)k; ++o; }
  }
  __syncthreads();
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  for (int64_t base = (int64_t)blockIdx.x * 16; base < P.F; base += (int64_t)gridDim.x * 16) {
    const int64_t f = base + g;
    const bool valid = f < P.F;
    const double* G = P.blocks + ((size_t)cur * P.F + (valid ? f : 0)) * 256;
    double fail = 0.0;
    if (valid) {
      double s[6], L[21];
#pragma unroll
      for (int i = 0; i < 6; ++i) s[i] = P.sp[f * 8 + i];
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) L[tri(i, j)] = s[i] * G[(9 + i) * 16 + 9 + j] * s[j];
#pragma unroll
      for (int i = 0; i < 6; ++i) L[tri(i, i)] += clampd(L[tri(i, i)], mn, mx) / radius;
      // in-place Cholesky (lower), fully unrolled so L stays in registers
      bool ok = true;
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        double d = L[tri(j, j)];
#pragma unroll
        for (int k = 0; k < j; ++k) d -= L[tri(j, k)] * L[tri(j, k)];
        ok = ok && (d > 0.0) && isfinite(d);
        d = sqrt(d);
        L[tri(j, j)] = d;
        const double inv = 1.0 / d;
#pragma unroll
        for (int i = j + 1; i < 6; ++i) {
          double a = L[tri(i, j)];
#pragma unroll
          for (int k = 0; k < j; ++k) a -= L[tri(i, k)] * L[tri(j, k)];
          L[tri(i, j)] = a * inv;
        }
      }
      if (!ok) fail = 1.0;
      if (l < 10) {
        const int col = l < 9 ? l : 15;
        const double sc = l < 9 ? s_ss[l] : 1.0;
        double z[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          double a = s[i] * G[(9 + i) * 16 + col] * sc;
#pragma unroll
          for (int k = 0; k < i; ++k) a -= L[tri(i, k)] * z[k];
          z[i] = a / L[tri(i, i)];
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          Zs[g][i * 10 + l] = z[i];
          P.ZL[f * kZLStride + i * 10 + l] = z[i];
        }
      } else {
#pragma unroll
        for (int e = 0; e < 21; ++e)
          if (l == 10 + (e >> 2)) P.ZL[f * kZLStride + 60 + e] = L[e];
      }
    }
    __syncthreads();
    if (valid) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int o = l * 4 + r;
        double a;
        if (o < 45) {
          const int j = pj[o], k = pk[o];
          a = s_ss[j] * G[j * 16 + k] * s_ss[k];
#pragma unroll
          for (int i = 0; i < 6; ++i) a -= Zs[g][i * 10 + j] * Zs[g][i * 10 + k];
        } else if (o < 54) {
          const int j = o - 45;
          a = s_ss[j] * G[j * 16 + 15];
#pragma unroll
          for (int i = 0; i < 6; ++i) a -= Zs[g][i * 10 + j] * Zs[g][i * 10 + 9];
        } else if (o < 63) {
          const int j = o - 54;
          a = s_ss[j] * s_ss[j] * G[j * 17];
        } else {
          a = fail;
        }
        acc[r] += a;
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) red[g][l * 4 + r] = acc[r];
  __syncthreads();
  if (tid < 64) {
    double a = 0.0;
    for (int g2 = 0; g2 < 16; ++g2) a += red[g2][tid];
    P.partial[blockIdx.x * 64 + tid] = a;
  }
}

// ---------------------------------------------------------------------------------------------
// solve: MODE 0 = reduce + solve, 1 = reduce only (-> vec_solve), 2 = solve only
// ---------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(64) void k_intr_solve(IntrDev P, int nblk) {
  __shared__ double sv[64];
  __shared__ double S[81];
  __shared__ double b[9];
  LmState* st = P.state;
  if (st->done) return;
  const int tid = threadIdx.x;
  const double radius = st->radius;
  const double mn = P.opts->min_lm_diagonal, mx = P.opts->max_lm_diagonal;
  if (MODE != 2) {
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int bb = 0;
    for (; bb + 3 < nblk; bb += 4) {
      a0 += P.partial[bb * 64 + tid]; a1 += P.partial[(bb + 1) * 64 + tid];
      a2 += P.partial[(bb + 2) * 64 + tid]; a3 += P.partial[(bb + 3) * 64 + tid];
    }
    for (; bb < nblk; ++bb) a0 += P.partial[bb * 64 + tid];
    const double a = (a0 + a1) + (a2 + a3);
    sv[tid] = a;
    if (MODE == 1) { P.vec_solve[tid] = a; return; }
  } else {
    sv[tid] = P.vec_solve[tid];
  }
  __syncthreads();
  if (tid != 0) return;
  const double* V = sv;
  bool ok = !(V[63] > 0.0);
  int idx = 0;
  for (int j = 0; j < 9; ++j)
    for (int k = j; k < 9; ++k) { S[j * 9 + k] = S[k * 9 + j] = V[idx]; ++idx; }
  for (int j = 0; j < 9; ++j) {
    S[j * 9 + j] += clampd(V[54 + j], mn, mx) / radius;
    b[j] = V[45 + j];
  }
  for (int j = 0; j < 9; ++j)
    if (P.mask & (1u << j)) {
      for (int k = 0; k < 9; ++k) S[j * 9 + k] = S[k * 9 + j] = 0.0;
      S[j * 9 + j] = 1.0;
      b[j] = 0.0;
    }
  for (int j = 0; j < 9 && ok; ++j) {
    double d = S[j * 9 + j];
    for (int k = 0; k < j; ++k) d -= S[j * 9 + k] * S[j * 9 + k];
    if (!(d > 0.0) || !isfinite(d)) { ok = false; break; }
    d = sqrt(d);
    S[j * 9 + j] = d;
    const double inv = 1.0 / d;
    for (int i = j + 1; i < 9; ++i) {
      double a = S[i * 9 + j];
      for (int k = 0; k < j; ++k) a -= S[i * 9 + k] * S[j * 9 + k];
      S[i * 9 + j] = a * inv;
    }
  }
  if (ok) {
    for (int i = 0; i < 9; ++i) {
      double a = b[i];
      for (int k = 0; k < i; ++k) a -= S[i * 9 + k] * b[k];
      b[i] = a / S[i * 9 + i];
    }
    for (int i = 8; i >= 0; --i) {
      double a = b[i];
      for (int k = i + 1; k < 9; ++k) a -= S[k * 9 + i] * b[k];
      b[i] = a / S[i * 9 + i];
    }
    for (int i = 0; i < 9; ++i) {
      st->ds[i] = -b[i];
      ok = ok && isfinite(b[i]);
    }
  }
  st->step_valid = ok ? 1 : 0;
}

}  // namespace cc

// =============================================================================================
// host side
// =============================================================================================
namespace cc {

// RCCL, resolved lazily with dlopen so that single-GPU use never loads it (cc_comm.cpp)
struct Comm;
int comm_create(const uint8_t id[128], int rank, int nranks, Comm** out);
void comm_destroy(Comm* c);
int comm_allreduce_sum(Comm* c, double* buf, int n, hipStream_t stream);

}  // namespace cc

struct cc_intrinsics {
  int device = 0;
  hipStream_t stream = nullptr;
  cc::IntrDev d{};
  int64_t F = 0, N = 0;
  int elim_blocks = 1;
  double* init_intr = nullptr;  // [16]
  double* init_pose = nullptr;  // [F][8]
  bool have_state = false;
  cc::LmState* h_state = nullptr;  // pinned
  hipGraphExec_t graph = nullptr;
  int graph_iters = 0;
  cc::Comm* comm = nullptr;
  std::vector<hipEvent_t> events;
  std::vector<int> event_kind;
};

namespace cc {

static void enqueue_kernel(cc_intrinsics* h, int kind, int variant, bool profile) {
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (profile) {
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0, h->stream);
  }
  switch (kind) {
    case CC_K_SWEEP:
      hipLaunchKernelGGL(k_intr_sweep, dim3((unsigned)h->F), dim3(kSweepThreads), kSweepLdsBytes, h->stream, h->d);
      break;
    case CC_K_DECIDE:
      if (variant == 0) hipLaunchKernelGGL(k_intr_decide<0>, dim3(1), dim3(kDecideThreads), 0, h->stream, h->d);
      else if (variant == 1) hipLaunchKernelGGL(k_intr_decide<1>, dim3(1), dim3(kDecideThreads), 0, h->stream, h->d);
      else hipLaunchKernelGGL(k_intr_decide<2>, dim3(1), dim3(kDecideThreads), 0, h->stream, h->d);
      break;
    case CC_K_ELIM:
      hipLaunchKernelGGL(k_intr_elim, dim3(h->elim_blocks), dim3(256), 0, h->stream, h->d);
      break;
    case CC_K_SOLVE:
      if (variant == 0) hipLaunchKernelGGL(k_intr_solve<0>, dim3(1), dim3(64), 0, h->stream, h->d, h->elim_blocks);
      else if (variant == 1) hipLaunchKernelGGL(k_intr_solve<1>, dim3(1), dim3(64), 0, h->stream, h->d, h->elim_blocks);
      else hipLaunchKernelGGL(k_intr_solve<2>, dim3(1), dim3(64), 0, h->stream, h->d, h->elim_blocks);
      break;
    default: break;
  }
  if (profile) {
    hipEventRecord(e1, h->stream);
    h->events.push_back(e0);
    h->events.push_back(e1);
    h->event_kind.push_back(kind);
  }
}

static int enqueue_allreduce(cc_intrinsics* h, double* buf, bool profile) {
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (profile) { hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0, h->stream); }
  const int rc = comm_allreduce_sum(h->comm, buf, 64, h->stream);
  if (profile) {
    hipEventRecord(e1, h->stream);
    h->events.push_back(e0); h->events.push_back(e1); h->event_kind.push_back(CC_K_ALLREDUCE);
  }
  return rc;
}

// sweep + decide (also used for the initial evaluation)
static int enqueue_sweep_decide(cc_intrinsics* h, bool profile) {
  enqueue_kernel(h, CC_K_SWEEP, 0, profile);
  if (h->comm) {
    enqueue_kernel(h, CC_K_DECIDE, 1, profile);
    if (int rc = enqueue_allreduce(h, h->d.vec_decide, profile)) return rc;
    enqueue_kernel(h, CC_K_DECIDE, 2, profile);
  } else {
    enqueue_kernel(h, CC_K_DECIDE, 0, profile);
  }
  return 0;
}

static int enqueue_iteration(cc_intrinsics* h, bool profile) {
  enqueue_kernel(h, CC_K_ELIM, 0, profile);
  if (h->comm) {
    enqueue_kernel(h, CC_K_SOLVE, 1, profile);
    if (int rc = enqueue_allreduce(h, h->d.vec_solve, profile)) return rc;
    enqueue_kernel(h, CC_K_SOLVE, 2, profile);
  } else {
    enqueue_kernel(h, CC_K_SOLVE, 0, profile);
  }
  return enqueue_sweep_decide(h, profile);
}

static int reset_device_state(cc_intrinsics* h, const cc_options& o) {
  LmOpts lo;
  opts_from_public(o, &lo);
  CC_HIP(hipMemcpyAsync(h->d.opts, &lo, sizeof(lo), hipMemcpyHostToDevice, h->stream));
  CC_HIP(hipMemsetAsync(h->d.state, 0, sizeof(LmState), h->stream));
  CC_HIP(hipMemsetAsync(h->d.vec_decide, 0, 64 * sizeof(double), h->stream));
  CC_HIP(hipMemsetAsync(h->d.vec_solve, 0, 64 * sizeof(double), h->stream));
  // current point lives in buffer 0
  CC_HIP(hipMemcpyAsync(h->d.intr, h->init_intr, 16 * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  CC_HIP(hipMemcpyAsync(h->d.pose, h->init_pose, (size_t)h->F * 8 * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  return 0;
}

}  // namespace cc

extern "C" {

int cc_intrinsics_create(int32_t device, int64_t F, const int64_t* off, const float* uv,
                         const float* xyz, cc_intrinsics** out) {
  using namespace cc;
  if (!out || !off || F <= 0) return fail(CC_ERR_BAD_ARGUMENT, "cc_intrinsics_create: bad arguments");
  if (off[0] != 0) return fail(CC_ERR_BAD_ARGUMENT, "frame_offsets[0] must be 0");
  for (int64_t f = 0; f < F; ++f)
    if (off[f + 1] < off[f]) return fail(CC_ERR_BAD_ARGUMENT, "frame_offsets must be non-decreasing");
  const int64_t N = off[F];
  if (N > 0 && (!uv || !xyz)) return fail(CC_ERR_BAD_ARGUMENT, "uv/xyz are NULL");
  if (int rc = select_device(device)) return rc;
  cc_intrinsics* h = new cc_intrinsics();
  h->device = device; h->F = F; h->N = N;
  CC_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  IntrDev& d = h->d;
  d.F = F; d.N = N; d.rank = 0; d.nranks = 1; d.mask = 0;
  float *duv, *dxyz; int64_t* doff;
  const size_t n1 = (size_t)std::max<int64_t>(N, 1);
  CC_HIP(hipMalloc(&duv, n1 * 2 * sizeof(float)));
  CC_HIP(hipMalloc(&dxyz, n1 * 3 * sizeof(float)));
  CC_HIP(hipMalloc(&doff, (size_t)(F + 1) * sizeof(int64_t)));
  if (N > 0) {
    CC_HIP(hipMemcpy(duv, uv, (size_t)N * 2 * sizeof(float), hipMemcpyHostToDevice));
    CC_HIP(hipMemcpy(dxyz, xyz, (size_t)N * 3 * sizeof(float), hipMemcpyHostToDevice));
  }
  CC_HIP(hipMemcpy(doff, off, (size_t)(F + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
  d.uv = duv; d.xyz = dxyz; d.off = doff;
  CC_HIP(hipMalloc(&d.intr, 2 * 16 * sizeof(double)));
  CC_HIP(hipMalloc(&d.pose, (size_t)2 * F * 8 * sizeof(double)));
  CC_HIP(hipMalloc(&d.blocks, (size_t)2 * F * 256 * sizeof(double)));
  CC_HIP(hipMalloc(&d.stats, (size_t)F * kStatsCols * sizeof(double)));
  CC_HIP(hipMalloc(&d.sp, (size_t)F * 8 * sizeof(double)));
  CC_HIP(hipMalloc(&d.ZL, (size_t)F * kZLStride * sizeof(double)));
  CC_HIP(hipMalloc(&d.partial, (size_t)kElimMaxBlocks * 64 * sizeof(double)));
  CC_HIP(hipMalloc(&d.vec_solve, 64 * sizeof(double)));
  CC_HIP(hipMalloc(&d.vec_decide, 64 * sizeof(double)));
  CC_HIP(hipMalloc(&d.state, sizeof(LmState)));
  CC_HIP(hipMalloc(&d.opts, sizeof(LmOpts)));
  d.log_cap = 4096;
  CC_HIP(hipMalloc(&d.log, (size_t)d.log_cap * sizeof(cc_iteration)));
  CC_HIP(hipMalloc(&h->init_intr, 16 * sizeof(double)));
  CC_HIP(hipMalloc(&h->init_pose, (size_t)F * 8 * sizeof(double)));
  CC_HIP(hipMemset(d.intr, 0, 2 * 16 * sizeof(double)));
  CC_HIP(hipMemset(d.pose, 0, (size_t)2 * F * 8 * sizeof(double)));
  CC_HIP(hipMemset(d.sp, 0, (size_t)F * 8 * sizeof(double)));
  CC_HIP(hipMemset(d.ZL, 0, (size_t)F * kZLStride * sizeof(double)));
  CC_HIP(hipHostMalloc(&h->h_state, sizeof(LmState), hipHostMallocDefault));
  h->elim_blocks = (int)std::min<int64_t>(kElimMaxBlocks, (F + 15) / 16);
  CC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_intr_sweep),
                             hipFuncAttributeMaxDynamicSharedMemorySize, kSweepLdsBytes));
  *out = h;
  return CC_OK;
}

void cc_intrinsics_destroy(cc_intrinsics* h) {
  if (!h) return;
  hipSetDevice(h->device);
  if (h->stream) hipStreamSynchronize(h->stream);
  if (h->graph) hipGraphExecDestroy(h->graph);
  if (h->comm) cc::comm_destroy(h->comm);
  cc::IntrDev& d = h->d;
  hipFree((void*)d.uv); hipFree((void*)d.xyz); hipFree((void*)d.off);
  hipFree(d.intr); hipFree(d.pose); hipFree(d.blocks); hipFree(d.stats); hipFree(d.sp);
  hipFree(d.ZL); hipFree(d.partial); hipFree(d.vec_solve); hipFree(d.vec_decide);
  hipFree(d.state); hipFree(d.opts); hipFree(d.log);
  hipFree(h->init_intr); hipFree(h->init_pose);
  if (h->h_state) hipHostFree(h->h_state);
  if (h->stream) hipStreamDestroy(h->stream);
  delete h;
}

int cc_intrinsics_set_state(cc_intrinsics* h, const double* intr9, uint32_t mask, const double* q,
                            const double* t) {
  using namespace cc;
  if (!h || !intr9 || !q || !t) return fail(CC_ERR_BAD_ARGUMENT, "cc_intrinsics_set_state: NULL argument");
  CC_HIP(hipSetDevice(h->device));
  std::vector<double> pose((size_t)h->F * 8, 0.0);
  for (int64_t f = 0; f < h->F; ++f) {
    for (int i = 0; i < 4; ++i) pose[f * 8 + i] = q[f * 4 + i];
    for (int i = 0; i < 3; ++i) pose[f * 8 + 4 + i] = t[f * 3 + i];
  }
  double k[16] = {0};
  for (int i = 0; i < 9; ++i) k[i] = intr9[i];
  CC_HIP(hipStreamSynchronize(h->stream));
  CC_HIP(hipMemcpy(h->init_intr, k, sizeof(k), hipMemcpyHostToDevice));
  CC_HIP(hipMemcpy(h->init_pose, pose.data(), pose.size() * sizeof(double), hipMemcpyHostToDevice));
  if (h->d.mask != (mask & 0x1ffu) && h->graph) {  // kernel arguments are baked into the graph
    hipGraphExecDestroy(h->graph);
    h->graph = nullptr;
  }
  h->d.mask = mask & 0x1ffu;
  h->have_state = true;
  return cc_intrinsics_reset(h);
}

int cc_intrinsics_reset(cc_intrinsics* h) {
  using namespace cc;
  if (!h || !h->have_state) return fail(CC_ERR_STATE, "cc_intrinsics_reset: no state set");
  CC_HIP(hipSetDevice(h->device));
  cc_options o;
  cc_options_init(&o);
  if (int rc = reset_device_state(h, o)) return rc;
  CC_HIP(hipStreamSynchronize(h->stream));
  return CC_OK;
}

int cc_intrinsics_get_state(cc_intrinsics* h, double* intr9, double* q, double* t) {
  using namespace cc;
  if (!h) return fail(CC_ERR_BAD_ARGUMENT, "cc_intrinsics_get_state: NULL handle");
  CC_HIP(hipSetDevice(h->device));
  CC_HIP(hipStreamSynchronize(h->stream));
  LmState st;
  CC_HIP(hipMemcpy(&st, h->d.state, sizeof(st), hipMemcpyDeviceToHost));
  const int cur = st.cur & 1;
  if (intr9) CC_HIP(hipMemcpy(intr9, h->d.intr + cur * 16, 9 * sizeof(double), hipMemcpyDeviceToHost));
  if (q || t) {
    std::vector<double> pose((size_t)h->F * 8);
    CC_HIP(hipMemcpy(pose.data(), h->d.pose + (size_t)cur * h->F * 8, pose.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (int64_t f = 0; f < h->F; ++f) {
      if (q) for (int i = 0; i < 4; ++i) q[f * 4 + i] = pose[f * 8 + i];
      if (t) for (int i = 0; i < 3; ++i) t[f * 3 + i] = pose[f * 8 + 4 + i];
    }
  }
  return CC_OK;
}

int cc_intrinsics_eval(cc_intrinsics* h, double* blocks, double* cost) {
  using namespace cc;
  if (!h || !h->have_state) return fail(CC_ERR_STATE, "cc_intrinsics_eval: no state set");
  CC_HIP(hipSetDevice(h->device));
  CC_HIP(hipStreamSynchronize(h->stream));
  // evaluate at the accepted point without disturbing it: phase 0 sweep writes into buffer `cur`
  LmState st;
  CC_HIP(hipMemcpy(&st, h->d.state, sizeof(st), hipMemcpyDeviceToHost));
  LmState ev = st;
  ev.done = 0; ev.phase = 0;
  CC_HIP(hipMemcpy(h->d.state, &ev, sizeof(ev), hipMemcpyHostToDevice));
  enqueue_kernel(h, CC_K_SWEEP, 0, false);
  CC_HIP(hipGetLastError());
  CC_HIP(hipStreamSynchronize(h->stream));
  const int cur = st.cur & 1;
  if (blocks)
    CC_HIP(hipMemcpy(blocks, h->d.blocks + (size_t)cur * h->F * 256, (size_t)h->F * 256 * sizeof(double), hipMemcpyDeviceToHost));
  if (cost) {
    std::vector<double> stats((size_t)h->F * kStatsCols);
    CC_HIP(hipMemcpy(stats.data(), h->d.stats, stats.size() * sizeof(double), hipMemcpyDeviceToHost));
    double c = 0.0;
    for (int64_t f = 0; f < h->F; ++f) c += stats[f * kStatsCols + ST_COST];
    *cost = c;
  }
  CC_HIP(hipMemcpy(h->d.state, &st, sizeof(st), hipMemcpyHostToDevice));
  return CC_OK;
}

int cc_intrinsics_solve(cc_intrinsics* h, const cc_options* opt, cc_summary* summary) {
  using namespace cc;
  if (!h || !h->have_state) return fail(CC_ERR_STATE, "cc_intrinsics_solve: no state set");
  const auto t0 = std::chrono::steady_clock::now();
  cc_options o;
  if (opt) o = *opt; else cc_options_init(&o);
  if (o.check_interval < 1) o.check_interval = 1;
  if (o.max_iterations > h->d.log_cap - 1) o.max_iterations = h->d.log_cap - 1;
  const bool profile = o.profile_kernels != 0;
  const bool use_graph = o.use_graph && !profile && !h->comm;
  CC_HIP(hipSetDevice(h->device));
  // restart from the accepted point of the previous run (buffer `cur`), not from buffer 0
  {
    CC_HIP(hipStreamSynchronize(h->stream));
    LmState st;
    CC_HIP(hipMemcpy(&st, h->d.state, sizeof(st), hipMemcpyDeviceToHost));
    if (st.cur & 1) {
      CC_HIP(hipMemcpy(h->d.intr, h->d.intr + 16, 16 * sizeof(double), hipMemcpyDeviceToDevice));
      CC_HIP(hipMemcpy(h->d.pose, h->d.pose + (size_t)h->F * 8, (size_t)h->F * 8 * sizeof(double), hipMemcpyDeviceToDevice));
    }
    LmOpts lo;
    opts_from_public(o, &lo);
    CC_HIP(hipMemcpyAsync(h->d.opts, &lo, sizeof(lo), hipMemcpyHostToDevice, h->stream));
    CC_HIP(hipMemsetAsync(h->d.state, 0, sizeof(LmState), h->stream));
    CC_HIP(hipMemsetAsync(h->d.vec_decide, 0, 64 * sizeof(double), h->stream));
    CC_HIP(hipMemsetAsync(h->d.vec_solve, 0, 64 * sizeof(double), h->stream));
  }
  for (auto e : h->events) hipEventDestroy(e);
  h->events.clear();
  h->event_kind.clear();

  if (int rc = enqueue_sweep_decide(h, profile)) return rc;
  CC_HIP(hipGetLastError());

  if (use_graph && (!h->graph || h->graph_iters != o.check_interval)) {
    if (h->graph) { hipGraphExecDestroy(h->graph); h->graph = nullptr; }
    hipGraph_t g = nullptr;
    CC_HIP(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < o.check_interval; ++i) enqueue_iteration(h, false);
    CC_HIP(hipStreamEndCapture(h->stream, &g));
    CC_HIP(hipGraphInstantiate(&h->graph, g, nullptr, nullptr, 0));
    hipGraphDestroy(g);
    h->graph_iters = o.check_interval;
  }

  int launched = 0;
  while (true) {
    CC_HIP(hipMemcpyAsync(h->h_state, h->d.state, sizeof(LmState), hipMemcpyDeviceToHost, h->stream));
    CC_HIP(hipStreamSynchronize(h->stream));
    if (h->h_state->done) break;
    if (launched > o.max_iterations + o.check_interval)
      return fail(CC_ERR_STATE, "LM loop did not terminate (iter=%d)", h->h_state->iter);
    if (use_graph) {
      CC_HIP(hipGraphLaunch(h->graph, h->stream));
    } else {
      for (int i = 0; i < o.check_interval; ++i)
        if (int rc = enqueue_iteration(h, profile)) return rc;
      CC_HIP(hipGetLastError());
    }
    launched += o.check_interval;
  }
  const LmState& st = *h->h_state;
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
    for (int i = 0; i < CC_K_COUNT; ++i) { summary->kernel_ms[i] = 0.0; summary->kernel_launches[i] = 0; }
    if (profile) {
      for (size_t i = 0; i < h->event_kind.size(); ++i) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, h->events[2 * i], h->events[2 * i + 1]) == hipSuccess) {
          summary->kernel_ms[h->event_kind[i]] += ms;
          summary->kernel_launches[h->event_kind[i]]++;
        }
      }
    }
    summary->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  for (auto e : h->events) hipEventDestroy(e);
  h->events.clear();
  h->event_kind.clear();
  return CC_OK;
}

int cc_intrinsics_profile_sweep(cc_intrinsics* h, int32_t n, double* avg_ms) {
  using namespace cc;
  if (!h || n < 1 || !avg_ms) return fail(CC_ERR_BAD_ARGUMENT, "cc_intrinsics_profile_sweep: bad arguments");
  CC_HIP(hipSetDevice(h->device));
  CC_HIP(hipStreamSynchronize(h->stream));
  LmState st;
  CC_HIP(hipMemcpy(&st, h->d.state, sizeof(st), hipMemcpyDeviceToHost));
  if (st.phase != 1 || st.iter < 1) return fail(CC_ERR_STATE, "cc_intrinsics_profile_sweep: run cc_intrinsics_solve first");
  LmState run = st;
  run.done = 0; run.step_valid = 1;
  CC_HIP(hipMemcpy(h->d.state, &run, sizeof(run), hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CC_HIP(hipEventCreate(&e0));
  CC_HIP(hipEventCreate(&e1));
  enqueue_kernel(h, CC_K_SWEEP, 0, false);  // warm
  CC_HIP(hipEventRecord(e0, h->stream));
  for (int i = 0; i < n; ++i) enqueue_kernel(h, CC_K_SWEEP, 0, false);
  CC_HIP(hipEventRecord(e1, h->stream));
  CC_HIP(hipGetLastError());
  CC_HIP(hipStreamSynchronize(h->stream));
  float ms = 0.f;
  CC_HIP(hipEventElapsedTime(&ms, e0, e1));
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  *avg_ms = (double)ms / n;
  CC_HIP(hipMemcpy(h->d.state, &st, sizeof(st), hipMemcpyHostToDevice));
  return CC_OK;
}

int cc_intrinsics_optimize(const cc_options* opt, int32_t device, int64_t F, const int64_t* off,
                           const float* uv, const float* xyz, double* intr9, uint32_t mask,
                           double* q, double* t, cc_summary* summary) {
  cc_intrinsics* h = nullptr;
  int rc = cc_intrinsics_create(device, F, off, uv, xyz, &h);
  if (rc) return rc;
  rc = cc_intrinsics_set_state(h, intr9, mask, q, t);
  if (!rc) rc = cc_intrinsics_solve(h, opt, summary);
  if (!rc) rc = cc_intrinsics_get_state(h, intr9, q, t);
  cc_intrinsics_destroy(h);
  return rc;
}

int cc_intrinsics_comm_init(cc_intrinsics* h, const uint8_t id[128], int32_t rank, int32_t nranks) {
  using namespace cc;
  if (!h || !id || rank < 0 || nranks < 1 || rank >= nranks || nranks > 32)
    return fail(CC_ERR_BAD_ARGUMENT, "cc_intrinsics_comm_init: bad arguments (nranks must be 1..32)");
  CC_HIP(hipSetDevice(h->device));
  if (h->comm) { comm_destroy(h->comm); h->comm = nullptr; }
  if (h->graph) { hipGraphExecDestroy(h->graph); h->graph = nullptr; }
  if (int rc = comm_create(id, rank, nranks, &h->comm)) return rc;
  h->d.rank = rank;
  h->d.nranks = nranks;
  return CC_OK;
}

}  // extern "C"
