// cc_rig.hip -- multi-camera rig pose bundle adjustment on MI355X (gfx950).
//
// Replaces the ceres::Problem/ceres::Solve block of ExtrinsicsCalibrator::Optimize
// (/root/reference/src/extrinsics_calibrator.cpp:92-225): shared block = one 6-dof pose per camera
// (camera_T_rig), one 6-dof pose per observation frame (rig_T_world), constant world points,
// residuals in normalised coordinates, ceres::HuberLoss(3/500).
//
// Observations are regrouped on the host into (frame, camera) groups; one workgroup sweeps one
// group and produces its 16x16 Gram block of [J_cam(6) J_frame(6) r 0 0 0] rows with the same
// LDS-staged v_mfma_f64_16x16x4_f64 contraction as the intrinsics problem. Per LM iteration:
//   solve  : reduce elimination partials, dense (6C)x(6C) Cholesky in LDS, camera candidates
//   update : per frame, back-substitute the pose step, candidate pose (QuaternionManifold::Plus)
//   sweep  : per group, residuals + Jacobian rows + Huber scaling -> Gram block, cost, model term
//   init   : (first evaluation only) Jacobi scaling of the shared block, trust-region state
//   decide+elim : trust-region decision; per frame 6x6 Cholesky and Schur complement partials
//   reduce : column sums of the elimination partials (and, multi-GPU, their posting to the mailboxes)
//
// EXTENSION (cc_rigk_*, SURVEY 8f rank 4, no counterpart in the reference): the same kernels, templated
// where it matters, with 9 intrinsics shared by all cameras appended to the shared block and pixel
// observations (see k_rig_sweep<true>).
#include <algorithm>
#include <chrono>
#include <numeric>
#include <vector>

#include "cc_common.hpp"
#include "cc_device.hpp"

namespace cc {

constexpr int kRigMaxCams = 10;
constexpr int kRigMaxS = 6 * kRigMaxCams;
constexpr int kRigThreads = 256;
constexpr int kRigOwn = 8;  // partial-row columns owned per thread of the elim kernel (PC <= 2048)
constexpr int kRigSweepLdsBytes = (4 * kStageDoublesPerWave + 256) * 8;
constexpr int kRigSweepLdsBytesK = (8 * kStageDoublesPerWave + 256) * 8;  // with intrinsics: two staged tiles per wave
constexpr int kRigK = 9;  // shared intrinsics columns of the extension (0 in the reference's problem)
constexpr int kRigMaxElimBlocks = 512;

struct RigDev {
  int64_t F, N, NG;
  int32_t C, S, SW, NP, PC;  // cameras, 6C, 6C+1, S(S+1)/2, partial-row columns
  int32_t pc_b, pc_hd, pc_fail, pc_gs, pc_gmax;
  int32_t nblk;
  const float* uv;        // [N] float2, (frame, camera)-sorted
  const int32_t* widx;    // [N] world point index
  const float* wxyz;      // [3P]
  const int64_t* goff;    // [NG+1] observation range of each group
  const int32_t* gframe;  // [NG]
  const int32_t* gcam;    // [NG]
  const int64_t* fgoff;   // [F+1] group range of each frame
  const int32_t* cam_goff;   // [C+1]
  const int32_t* cam_glist;  // [NG] groups of each camera
  const uint8_t* cam_fixed;  // [C] frozen or unobserved
  const uint8_t* pair_p;     // [NP] (p,q), p <= q, row-major upper triangle
  const uint8_t* pair_q;
  double* cam;      // [2][C][8] q(4) t(3)
  double* pose;     // [2][F][8]
  double* camrec;   // [C][32] R(9) t(3) unscaled step(6)
  double* frec;     // [F][32] R(9) t(3) unscaled step(6)
  double* gblocks;  // [2][NG][gstride]: 256 (poses only) or 3 x 256 (AA | AB | BB tiles, with intrinsics)
  double* gstats;   // [NG][2] cost, model term
  double* fstats;   // [F][2] step^2, |x|^2
  double* ghd0;     // [NG][8] diag of H_cc at the initial point
  double* sp;       // [F][8]
  double* ss;       // [64]
  double* ds;       // [64] scaled shared step
  double* Y;        // [F][6*SW]
  double* partial;  // [nblk][PC]
  double* vec;      // [PC + 32] column sums of the partial rows (k_rig_reduce) + one max-gradient slot per rank
  double* vec_stats;  // [4 + 6C] globally reduced sweep statistics (multi-GPU only)
  int32_t comm, rank, nranks;  // comm != 0: statistics / partial sums pass through an all-reduce
  P2pDev x;                    // mailbox exchange (cc_device.hpp); x.on != 0 replaces the RCCL all-reduces
  double* shared_stats;  // [4] step^2 and |x|^2 of the shared block (candidate)
  LmCtl* ctl;
  LmCtl* ctl_next;
  LmOpts* opts;
  cc_iteration* log;
  int32_t log_cap;
  double huber_a;
  // EXTENSION: 9 intrinsics shared by all cameras, pixel observations. K = 0: off.
  // Shared tangent = [cam 0 (6) ... cam C-1 (6) | k (9)], S = 6C + K, S6 = 6C.
  int32_t K, S6, gstride;
  uint32_t kmask;     // bit i: intrinsic i is held constant
  double* intr;       // [2][16]
  double* krec;       // [32]: candidate intrinsics [0..8], unscaled step [16..24]
  double* ghdk;       // [NG][16] diag of the intrinsics block of each group at the initial point
};

// ceres::HuberLoss(a) + Corrector (rho'' <= 0): residual and Jacobian scaled by sqrt(rho')
__device__ __forceinline__ void huber(double a, double s, double& rho, double& sr) {
  const double b = a * a;
  if (s > b) {
    const double r = sqrt(s);
    rho = 2.0 * a * r - b;
    sr = sqrt(fmax(2.2250738585072014e-308, a / r));
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

// row = sr * [d res / d cam rot(3) t(3) | d res / d frame rot(3) t(3) | res | 0 0 0],
// B = d res / d x_cam for this row.
__device__ __forceinline__ void rig_row(const RigObs& o, const double* Rc, double B0, double B1, double B2,
                                        double res, double sr, bool cam_fixed, double* v) {
  v[0] = 2.0 * (B2 * o.a1 - B1 * o.a2); v[1] = 2.0 * (B0 * o.a2 - B2 * o.a0); v[2] = 2.0 * (B1 * o.a0 - B0 * o.a1);
  v[3] = B0; v[4] = B1; v[5] = B2;
  const double m0 = B0 * Rc[0] + B1 * Rc[3] + B2 * Rc[6];
  const double m1 = B0 * Rc[1] + B1 * Rc[4] + B2 * Rc[7];
  const double m2 = B0 * Rc[2] + B1 * Rc[5] + B2 * Rc[8];
  v[6] = 2.0 * (m2 * o.b1 - m1 * o.b2); v[7] = 2.0 * (m0 * o.b2 - m2 * o.b0); v[8] = 2.0 * (m1 * o.b0 - m0 * o.b1);
  v[9] = m0; v[10] = m1; v[11] = m2;
  v[12] = res;
  v[13] = 0.0; v[14] = 0.0; v[15] = 0.0;
#pragma unroll
  for (int c = 0; c < 13; ++c) v[c] *= sr;
  if (cam_fixed) {
#pragma unroll
    for (int c = 0; c < 6; ++c) v[c] = 0.0;
  }
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

// Gram tile between two staged 16-column halves of the same 64 rows: D[i][j] += sum_rows a[i] b[j]
__device__ __forceinline__ void gram_rows_ab(const double* sa, const double* sb, int lane, d4& acc0, d4& acc1) {
  const int c = lane & 15, sub = lane >> 4;
#pragma unroll
  for (int m = 0; m < 16; m += 2) {
    const int r0 = 4 * m + sub, r1 = 4 * (m + 1) + sub;
    const int o0 = r0 * 16 + (((c >> 1) ^ (r0 & 7)) << 1) + (c & 1), o1 = r1 * 16 + (((c >> 1) ^ (r1 & 7)) << 1) + (c & 1);
    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(sa[o0], sb[o0], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(sa[o1], sb[o1], acc1, 0, 0, 0);
  }
}

// ---------------------------------------------------------------------------------------------
// sweep: one workgroup per (frame, camera) group. HK = false: the reference's problem (normalised
// observations, poses only, one 16x16 Gram tile). HK = true (extension): pixel observations through 9
// shared intrinsics; the row is [J_cam(6) J_frame(6) r 0 0 0 | J_k(9) 0...] and the group block has
// three tiles: AA (as before), AB (first half x intrinsics), BB (intrinsics x intrinsics).
// ---------------------------------------------------------------------------------------------
template <bool HK>
__global__ __launch_bounds__(kRigThreads, HK ? 2 : 4) void k_rig_sweep(RigDev P) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  constexpr int kTiles = HK ? 2 : 1;
  double* s_stage = reinterpret_cast<double*>(smem_raw);
  double* s_blk = s_stage;
  double* sm = s_stage + 4 * kTiles * kStageDoublesPerWave;  // [256]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t g = blockIdx.x;
  const LmCtl* ctl = P.ctl;
  if (ctl->done) return;
  const int phase = ctl->phase;
  if (phase != 0 && !ctl->step_valid) return;
  const int cur = ctl->cur, dst = phase == 0 ? cur : (cur ^ 1);
  const int f = P.gframe[g], c = P.gcam[g];
  const bool fixed = P.cam_fixed[c] != 0;
  // observations are fetched one pass ahead: pixel + world index, then the gathered world point (two
  // dependent round trips); the first pass is issued here, under the prologue
  const int64_t s0 = P.goff[g], s1 = P.goff[g + 1];
  // passes of THIS wave: a wave whose 64 slots of a pass all lie beyond the group skips that pass
  // (wave-uniform; the main loop holds no workgroup barrier)
  const int64_t wrem = s1 - s0 - (tid >> 6) * 64;
  const int npass = wrem > 0 ? (int)((wrem + kRigThreads - 1) / kRigThreads) : 0;
  const float2* uv2 = reinterpret_cast<const float2*>(P.uv);
  float2 nm = make_float2(0.f, 0.f);
  float nX0 = 0.f, nX1 = 0.f, nX2 = 1.f;
  if (npass > 0) {
    const int64_t idx = s0 + tid;
    const int64_t ic = idx < s1 ? idx : s0;
    nm = uv2[ic];
    const int64_t w = P.widx[ic];
    nX0 = P.wxyz[w * 3]; nX1 = P.wxyz[w * 3 + 1]; nX2 = P.wxyz[w * 3 + 2];
  }
  // sm[0..31] camera record, sm[32..63] frame record, sm[64..95] intrinsics record (candidate, step)
  if (tid < 32) sm[tid] = P.camrec[c * 32 + tid];
  else if (tid < 64) sm[tid] = P.frec[(size_t)f * 32 + (tid - 32)];
  else if (HK && tid < 96) sm[tid] = P.krec[tid - 64];
  const size_t gs = (size_t)P.gstride;
  double g_old = 0.0, g_ab = 0.0, g_bb = 0.0;
  if (phase != 0) {
    const double* old = P.gblocks + ((size_t)cur * P.NG + g) * gs;
    g_old = old[tid];
    if (HK) { g_ab = old[256 + tid]; g_bb = old[512 + tid]; }
  }
  __syncthreads();
  // model-cost term of the group: d = [dc(6) df(6) (dk(9))], q = d^T g + 1/2 d^T H d over its block
  double qterm = 0.0;
  if (phase != 0) {
    const int a = tid >> 4, b = tid & 15;
    const double da = a < 6 ? sm[12 + a] : (a < 12 ? sm[32 + 12 + (a - 6)] : 0.0);
    if (a < 12) {
      if (b < 12) {
        const double db = b < 6 ? sm[12 + b] : sm[32 + 12 + (b - 6)];
        qterm = 0.5 * da * g_old * db;
      } else if (b == 12) {
        qterm = da * g_old;
      }
    }
    if (HK) {
      const double dkb = b < 9 ? sm[64 + 16 + b] : 0.0;
      if (a < 12) qterm += da * g_ab * dkb;            // cross term, counted once (1/2 * 2)
      else if (a == 12) qterm += g_ab * dkb;           // gradient with respect to the intrinsics
      if (a < 9) qterm += 0.5 * sm[64 + 16 + a] * g_bb * dkb;
    }
  }
  double Rc[9], tc[3], Rf[9], tf[3], kk[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) { Rc[i] = rfl(sm[i]); Rf[i] = rfl(sm[32 + i]); kk[i] = HK ? rfl(sm[64 + i]) : 0.0; }
#pragma unroll
  for (int i = 0; i < 3; ++i) { tc[i] = rfl(sm[9 + i]); tf[i] = rfl(sm[32 + 9 + i]); }
  const double ha = P.huber_a;
  const uint32_t kmask = P.kmask;

  double* stage = s_stage + wave * kTiles * kStageDoublesPerWave;
  double* stage_b = stage + kStageDoublesPerWave;
  d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
  d4 ab0 = {0.0, 0.0, 0.0, 0.0}, ab1 = {0.0, 0.0, 0.0, 0.0}, bb0 = {0.0, 0.0, 0.0, 0.0}, bb1 = {0.0, 0.0, 0.0, 0.0};
  double cost = 0.0;
  for (int p = 0; p < npass; ++p) {
    const int64_t idx = s0 + (int64_t)p * kRigThreads + tid;
    const bool valid = idx < s1;
    const float2 m = nm;
    const float X0 = nX0, X1 = nX1, X2 = nX2;
    if (p + 1 < npass) {
      const int64_t idn = idx + kRigThreads;
      const int64_t ic = idn < s1 ? idn : s0;
      nm = uv2[ic];
      const int64_t w = P.widx[ic];
      nX0 = P.wxyz[w * 3]; nX1 = P.wxyz[w * 3 + 1]; nX2 = P.wxyz[w * 3 + 2];
    }
    RigObs o;
    rig_common(Rf, tf, Rc, tc, (double)X0, (double)X1, (double)X2, (double)m.x, (double)m.y, o);
    RigKObs ko;
    double ru = o.ru, rv = o.rv;
    double Bu0 = o.iz, Bu1 = 0.0, Bu2 = -o.x * o.iz, Bv0 = 0.0, Bv1 = o.iz, Bv2 = -o.y * o.iz;
    if (HK) {
      rigk_obs(kk, o, (double)m.x, (double)m.y, ko);
      ru = ko.ru; rv = ko.rv;
      Bu0 = ko.Bu0; Bu1 = ko.Bu1; Bu2 = ko.Bu2; Bv0 = ko.Bv0; Bv1 = ko.Bv1; Bv2 = ko.Bv2;
    }
    double rho, sr;
    huber(ha, ru * ru + rv * rv, rho, sr);
    if (valid) cost += 0.5 * rho;
    double v[16], vb[16];
    rig_row(o, Rc, Bu0, Bu1, Bu2, ru, sr, fixed, v);
    if (!valid) {
#pragma unroll
      for (int k = 0; k < 16; ++k) v[k] = 0.0;
    }
    stage_row(stage, lane, v);
    if (HK) {
#pragma unroll
      for (int k = 0; k < 16; ++k) vb[k] = (k < 9 && valid && !(kmask & (1u << k))) ? sr * ko.ju[k] : 0.0;
      stage_row(stage_b, lane, vb);
    }
    wave_lds_fence();
    gram_rows(stage, lane, acc0, acc1);
    if (HK) { gram_rows_ab(stage, stage_b, lane, ab0, ab1); gram_rows(stage_b, lane, bb0, bb1); }
    wave_lds_fence();
    rig_row(o, Rc, Bv0, Bv1, Bv2, rv, sr, fixed, v);
    if (!valid) {
#pragma unroll
      for (int k = 0; k < 16; ++k) v[k] = 0.0;
    }
    stage_row(stage, lane, v);
    if (HK) {
#pragma unroll
      for (int k = 0; k < 16; ++k) vb[k] = (k < 9 && valid && !(kmask & (1u << k))) ? sr * ko.jv[k] : 0.0;
      stage_row(stage_b, lane, vb);
    }
    wave_lds_fence();
    gram_rows(stage, lane, acc0, acc1);
    if (HK) { gram_rows_ab(stage, stage_b, lane, ab0, ab1); gram_rows(stage_b, lane, bb0, bb1); }
    wave_lds_fence();
  }
  __syncthreads();
  const int slot = ((lane >> 4)) * 16 + (lane & 15);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    s_blk[wave * 256 + slot + 64 * r] = acc0[r] + acc1[r];
    if (HK) {
      s_blk[1024 + wave * 256 + slot + 64 * r] = ab0[r] + ab1[r];
      s_blk[2048 + wave * 256 + slot + 64 * r] = bb0[r] + bb1[r];
    }
  }
  const double qw = wave_sum(qterm), cw = wave_sum(cost);
  if (lane == 0) { sm[140 + wave] = qw; sm[144 + wave] = cw; }
  __syncthreads();
  double* out = P.gblocks + ((size_t)dst * P.NG + g) * gs;
  const double gv = (s_blk[tid] + s_blk[256 + tid]) + (s_blk[512 + tid] + s_blk[768 + tid]);
  out[tid] = gv;
  if (HK) {
    out[256 + tid] = (s_blk[1024 + tid] + s_blk[1280 + tid]) + (s_blk[1536 + tid] + s_blk[1792 + tid]);
    const double bv = (s_blk[2048 + tid] + s_blk[2304 + tid]) + (s_blk[2560 + tid] + s_blk[2816 + tid]);
    out[512 + tid] = bv;
    if (phase == 0 && (tid >> 4) < 9 && (tid & 15) == (tid >> 4)) P.ghdk[g * 16 + (tid >> 4)] = bv;  // diag of H_kk
  }
  if (tid == 0) {
    P.gstats[g * 2] = (sm[144] + sm[145]) + (sm[146] + sm[147]);
    P.gstats[g * 2 + 1] = (sm[140] + sm[141]) + (sm[142] + sm[143]);
  }
  if (phase == 0 && (tid >> 4) < 6 && (tid & 15) == (tid >> 4)) P.ghd0[g * 8 + (tid >> 4)] = gv;  // diag of H_cc
}

// ---------------------------------------------------------------------------------------------
// update: per frame, back-substitute the pose step and form the candidate pose. 16 lanes/frame.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rig_update(RigDev P) {
  const LmCtl* ctl = P.ctl;
  if (ctl->done) return;
  const int phase = ctl->phase;
  if (phase != 0 && !ctl->step_valid) return;
  const int cur = ctl->cur, dst = phase == 0 ? cur : (cur ^ 1);
  const int tid = threadIdx.x, l = tid & 15;
  const int64_t f = (int64_t)blockIdx.x * 16 + (tid >> 4);
  const bool valid = f < P.F;
  const int64_t fc = valid ? f : 0;
  double u[6] = {0, 0, 0, 0, 0, 0};
  if (phase != 0) {
    const double* Yf = P.Y + (size_t)fc * 6 * P.SW;
    for (int k = l; k < P.SW; k += 16) {
      const double d = k < P.S ? P.ds[k] : 1.0;
#pragma unroll
      for (int i = 0; i < 6; ++i) u[i] += Yf[i * P.SW + k] * d;
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) u[i] += __shfl_xor(u[i], o, 64);
    }
  }
  if (!valid || l != 0) return;
  const bool active = P.fgoff[f + 1] > P.fgoff[f];
  const double* pc = P.pose + ((size_t)cur * P.F + f) * 8;
  double q[4] = {pc[0], pc[1], pc[2], pc[3]}, t[3] = {pc[4], pc[5], pc[6]};
  double dp[6] = {0, 0, 0, 0, 0, 0};
  double step2 = 0.0;
  if (phase != 0) {
    if (active) {
#pragma unroll
      for (int i = 0; i < 6; ++i) dp[i] = -u[i] * P.sp[f * 8 + i];
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

// deterministic block-wide sum of one value per thread (256 threads); result valid for thread 0
__device__ __forceinline__ double block_sum256(double v, double* s4) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s4[threadIdx.x >> 6] = v;
  __syncthreads();
  return (s4[0] + s4[1]) + (s4[2] + s4[3]);
}

// column sums of gstats[NG][2] and fstats[F][2] -> out[0..3] = cost, q, step2, xnorm2 (all threads)
__device__ __forceinline__ void rig_reduce_stats(const RigDev& P, bool want, double* s4, double* out) {
  const int tid = threadIdx.x;
  double a[4] = {0, 0, 0, 0};
  if (want) {
    for (int64_t i = tid; i < P.NG; i += 256) { a[0] += P.gstats[i * 2]; a[1] += P.gstats[i * 2 + 1]; }
    for (int64_t i = tid; i < P.F; i += 256) { a[2] += P.fstats[i * 2]; a[3] += P.fstats[i * 2 + 1]; }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const double s = block_sum256(a[k], s4);
    if (tid == 0) out[k] = s;
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// stats (multi-GPU only, one block): local sums of the sweep statistics -> vec_stats, which is then
// all-reduced. [0..3] cost, model term, step^2, |x|^2; in phase 0 also [4 + 6c + i] = sum of
// diag(H_cc) of camera c (Jacobi scaling of the shared block).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rig_stats(RigDev P) {
  __shared__ double s4[4];
  __shared__ double s_out[4];
  const LmCtl* ctl = P.ctl;
  if (ctl->done) return;
  const int tid = threadIdx.x, phase = ctl->phase;
  const bool need = phase == 0 || (ctl->cand_pending && ctl->step_valid);
  rig_reduce_stats(P, need, s4, s_out);
  if (tid < 4) P.vec_stats[tid] = need ? s_out[tid] : 0.0;
  for (int c = 0; c < P.C; ++c) {
    double h[6] = {0, 0, 0, 0, 0, 0};
    if (phase == 0)
      for (int k = P.cam_goff[c] + tid; k < P.cam_goff[c + 1]; k += 256) {
        const int g = P.cam_glist[k];
#pragma unroll
        for (int i = 0; i < 6; ++i) h[i] += P.ghd0[(size_t)g * 8 + i];
      }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const double s = block_sum256(h[i], s4);
      if (tid == 0) P.vec_stats[4 + c * 6 + i] = s;
    }
  }
  for (int j = 0; j < P.K; ++j) {   // extension: diagonal of the intrinsics block (Jacobi scaling)
    double hk = 0.0;
    if (phase == 0)
      for (int64_t g = tid; g < P.NG; g += 256) hk += P.ghdk[g * 16 + j];
    const double sum = block_sum256(hk, s4);
    if (tid == 0) P.vec_stats[4 + P.S6 + j] = sum;
  }
  if (P.x.on) {
    // mailbox exchange (kind 1): post the local statistics, wait for every rank's, write the sums back
    // (k_rig_init / k_rig_decide_elim read vec_stats as they do after an all-reduce)
    __shared__ double s_post[4 + kRigMaxS];
    __shared__ int s_ok;
    __syncthreads();
    const int n = 4 + P.S;
    if (tid < n) s_post[tid] = P.vec_stats[tid];
    __syncthreads();
    const unsigned long long epoch = P.x.seq[1] + 1ull;
    p2p_post(P.x, 1, epoch, P.rank, P.nranks, s_post, n);
    const double a = p2p_collect(P.x, 1, epoch, P.rank, P.nranks, n, &s_ok);
    if (tid < n) P.vec_stats[tid] = a;
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
__global__ __launch_bounds__(64) void k_rig_flag_exchange(RigDev P, const double* in, double* out, int n, int* ok) {
  __shared__ double s_post[64];
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
  __shared__ double s4[4];
  __shared__ double s_out[4];
  __shared__ double s_ss[kRigMaxS];
  const LmCtl* ctl = P.ctl;
  if (ctl->done || ctl->phase != 0) return;
  const int tid = threadIdx.x;
  const bool jac = P.opts->jacobi_scaling != 0;
  if (P.comm) {
    if (tid < 4) s_out[tid] = P.vec_stats[tid];
    if (tid < P.S) s_ss[tid] = jac ? 1.0 / (1.0 + sqrt(P.vec_stats[4 + tid])) : 1.0;
  } else {
    rig_reduce_stats(P, true, s4, s_out);
    for (int c = 0; c < P.C; ++c) {
      double h[6] = {0, 0, 0, 0, 0, 0};
      for (int k = P.cam_goff[c] + tid; k < P.cam_goff[c + 1]; k += 256) {
        const int g = P.cam_glist[k];
#pragma unroll
        for (int i = 0; i < 6; ++i) h[i] += P.ghd0[(size_t)g * 8 + i];
      }
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const double s = block_sum256(h[i], s4);
        if (tid == 0) s_ss[c * 6 + i] = jac ? 1.0 / (1.0 + sqrt(s)) : 1.0;
      }
    }
  }
  if (P.K && !P.comm) {
    // extension: diagonal of the intrinsics block summed over all groups
    for (int j = 0; j < P.K; ++j) {
      double h = 0.0;
      for (int64_t g = tid; g < P.NG; g += 256) h += P.ghdk[g * 16 + j];
      const double sum = block_sum256(h, s4);
      if (tid == 0) s_ss[P.S6 + j] = jac ? 1.0 / (1.0 + sqrt(sum)) : 1.0;
    }
  }
  __syncthreads();
  if (tid < P.S) P.ss[tid] = s_ss[tid];
  if (tid == 0) {
    LmCtl c = *ctl;
    const LmOpts o = *P.opts;
    double xn2 = s_out[3];
    for (int cc2 = 0; cc2 < P.C; ++cc2)
      if (!P.cam_fixed[cc2])
        for (int i = 0; i < 7; ++i) { const double v = P.cam[((size_t)c.cur * P.C + cc2) * 8 + i]; xn2 += v * v; }
    for (int j = 0; j < P.K; ++j) { const double v = P.intr[c.cur * 16 + j]; xn2 += v * v; }
    lm_init(c, o, s_out[0], sqrt(xn2));
    *P.ctl = c;
    *P.ctl_next = c;
  }
}

// ---------------------------------------------------------------------------------------------
// decide + elim. All 256 threads of a block work on one frame at a time; the block loops over its
// frames. Each thread owns up to kRigOwn columns of the partial row and accumulates them in
// registers across frames (no atomics, deterministic).
// Partial row: [0..NP) upper triangle of the reduced (6C)x(6C) system, [pc_b..) rhs, [pc_hd..) diag of
// the scaled H_ss, [pc_fail] Cholesky failures, [pc_gs..) unscaled shared gradient, [pc_gmax] max |g_frame|.
// ---------------------------------------------------------------------------------------------
template <bool HK>
__global__ __launch_bounds__(256) void k_rig_decide_elim(RigDev P) {
  __shared__ double LG[kRigMaxCams][256];
  // extension (P.K != 0): AB tiles of the frame's groups, and the frame's sums over its groups of the
  // intrinsics x intrinsics tile, of the frame-pose x intrinsics rows and of the intrinsics gradient
  __shared__ double LGB[kRigMaxCams][256];
  __shared__ double s_BB[256];
  __shared__ double s_B[6][16];
  __shared__ double s_gk[16];
  __shared__ double Zl[6][kRigMaxS + 4];
  __shared__ double s_A[28];  // 21 packed H_ff entries + 6 g_f
  __shared__ double s_ss[kRigMaxS];
  __shared__ double s4[4];
  __shared__ double s_tot[4];
  __shared__ int s_slot[kRigMaxCams];   // camera -> slot of its group in this frame (-1: absent)
  __shared__ int s_cam[kRigMaxCams];    // slot -> camera
  __shared__ LmCtl s_ctl;
  const int tid = threadIdx.x;
  const LmCtl* ctl = P.ctl;
  if (ctl->done || ctl->phase == 0) return;
  const bool pending = ctl->cand_pending != 0;
  if (P.comm) {
    if (tid < 4) s_tot[tid] = P.vec_stats[tid];
    __syncthreads();
  } else {
    rig_reduce_stats(P, pending && ctl->step_valid, s4, s_tot);
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
  __syncthreads();
  if (s_ctl.done) return;
  const int cur = s_ctl.cur;
  const double radius = s_ctl.radius;
  const double mn = P.opts->min_lm_diagonal, mx = P.opts->max_lm_diagonal;

  // static description of the columns this thread owns
  int op[kRigOwn], oq[kRigOwn];
#pragma unroll
  for (int r = 0; r < kRigOwn; ++r) {
    const int o = tid + 256 * r;
    op[r] = -1; oq[r] = -1;
    if (o < P.NP) { op[r] = P.pair_p[o]; oq[r] = P.pair_q[o]; }
  }
  double acc[kRigOwn];
#pragma unroll
  for (int r = 0; r < kRigOwn; ++r) acc[r] = 0.0;

  for (int64_t f = blockIdx.x; f < P.F; f += gridDim.x) {
    const int64_t g0 = P.fgoff[f];
    const int ng = (int)(P.fgoff[f + 1] - g0);
    if (ng == 0) continue;  // uniform across the block
    if (tid < kRigMaxCams) s_slot[tid] = -1;
    __syncthreads();
    if (tid < ng) { const int cam = P.gcam[g0 + tid]; s_cam[tid] = cam; s_slot[cam] = tid; }
    for (int s = 0; s < ng; ++s) LG[s][tid] = P.gblocks[((size_t)cur * P.NG + g0 + s) * P.gstride + tid];
    if (HK) {
      double bb = 0.0;
      for (int s = 0; s < ng; ++s) {
        const double* gb = P.gblocks + ((size_t)cur * P.NG + g0 + s) * P.gstride;
        LGB[s][tid] = gb[256 + tid];
        bb += gb[512 + tid];
      }
      s_BB[tid] = bb;
    }
    __syncthreads();
    // frame block: A = sum over groups of H_ff (rows/cols 6..11), g_f = sum of column 12
    if (tid < 21 + 6) {
      double a = 0.0;
      if (tid < 21) {
        int i = 0;
        while (tri(i + 1, 0) <= tid) ++i;
        const int j = tid - tri(i, 0);
        for (int s = 0; s < ng; ++s) a += LG[s][(6 + i) * 16 + 6 + j];
      } else {
        const int i = tid - 21;
        for (int s = 0; s < ng; ++s) a += LG[s][(6 + i) * 16 + 12];
      }
      s_A[tid] = a;
    } else if (HK && tid >= 32 && tid < 32 + 54) {
      const int i = (tid - 32) / 9, j = (tid - 32) - i * 9;
      double a = 0.0;
      for (int s = 0; s < ng; ++s) a += LGB[s][(6 + i) * 16 + j];
      s_B[i][j] = a;
    } else if (HK && tid >= 96 && tid < 96 + 9) {
      double a = 0.0;
      for (int s = 0; s < ng; ++s) a += LGB[s][12 * 16 + (tid - 96)];
      s_gk[tid - 96] = a;
    }
    __syncthreads();
    double sf[6], L[21], Li[6];
    if (ctl->phase == 1 && s_ctl.iter == 0 && !pending) {
      // first elimination after the initial evaluation: Jacobi scale of this frame's pose block
#pragma unroll
      for (int i = 0; i < 6; ++i) sf[i] = P.opts->jacobi_scaling ? 1.0 / (1.0 + sqrt(s_A[tri(i, i)])) : 1.0;
      if (tid < 6) P.sp[f * 8 + tid] = sf[tid];
    } else {
#pragma unroll
      for (int i = 0; i < 6; ++i) sf[i] = P.sp[f * 8 + i];
    }
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = 0; j <= i; ++j) L[tri(i, j)] = sf[i] * s_A[tri(i, j)] * sf[j];
#pragma unroll
    for (int i = 0; i < 6; ++i) L[tri(i, i)] += clampd(L[tri(i, i)], mn, mx) / radius;
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
      Li[j] = inv;
#pragma unroll
      for (int i = j + 1; i < 6; ++i) {
        double a = L[tri(i, j)];
#pragma unroll
        for (int k = 0; k < j; ++k) a -= L[tri(i, k)] * L[tri(j, k)];
        L[tri(i, j)] = a * inv;
      }
    }
    // columns of [H_fs | g_f]: column k < S belongs to camera k/6 (zero if absent), column S is g_f
    if (tid < P.SW) {
      const int k = tid;
      double w[6];
      if (k < P.S6) {
        const int cam = k / 6, a = k - cam * 6, slot = s_slot[cam];
#pragma unroll
        for (int i = 0; i < 6; ++i) w[i] = slot >= 0 ? sf[i] * LG[slot][a * 16 + 6 + i] * s_ss[k] : 0.0;
      } else if (HK && k < P.S) {
#pragma unroll
        for (int i = 0; i < 6; ++i) w[i] = sf[i] * s_B[i][k - P.S6] * s_ss[k];
      } else {
#pragma unroll
        for (int i = 0; i < 6; ++i) w[i] = sf[i] * s_A[21 + i];
      }
      double z[6], y[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        double a = w[i];
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
        Zl[i][k] = z[i];
        P.Y[((size_t)f * 6 + i) * P.SW + k] = y[i];
      }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kRigOwn; ++r) {
      const int o = tid + 256 * r;
      if (o >= P.PC) continue;
      double a = 0.0;
      if (o < P.NP) {
        // pair (p, q), p <= q, of shared columns: camera columns exist in this frame only if the camera
        // has a group here, intrinsics columns always do
        const int p = op[r], q = oq[r];
        const bool pk = HK && p >= P.S6, qk = HK && q >= P.S6;
        const int cp = pk ? 0 : p / 6, cq = qk ? 0 : q / 6;
        const int sp_ = pk ? 0 : s_slot[cp], sq_ = qk ? 0 : s_slot[cq];
        if (sp_ >= 0 && sq_ >= 0) {
          if (pk) a = s_ss[p] * s_BB[(p - P.S6) * 16 + (q - P.S6)] * s_ss[q];                 // k x k
          else if (qk) a = s_ss[p] * LGB[sp_][(p - cp * 6) * 16 + (q - P.S6)] * s_ss[q];       // camera x k
          else if (cp == cq) a = s_ss[p] * LG[sp_][(p - cp * 6) * 16 + (q - cq * 6)] * s_ss[q];
#pragma unroll
          for (int i = 0; i < 6; ++i) a -= Zl[i][p] * Zl[i][q];
        }
        acc[r] += a;
      } else if (o < P.pc_hd) {
        const int p = o - P.pc_b;
        const bool pk = HK && p >= P.S6;
        const int cp = pk ? 0 : p / 6, sl = pk ? 0 : s_slot[cp];
        if (sl >= 0) {
          a = pk ? s_ss[p] * s_gk[p - P.S6] : s_ss[p] * LG[sl][(p - cp * 6) * 16 + 12];
#pragma unroll
          for (int i = 0; i < 6; ++i) a -= Zl[i][p] * Zl[i][P.S];
        }
        acc[r] += a;
      } else if (o < P.pc_fail) {
        const int p = o - P.pc_hd;
        const bool pk = HK && p >= P.S6;
        const int cp = pk ? 0 : p / 6, sl = pk ? 0 : s_slot[cp];
        if (sl >= 0) a = pk ? s_ss[p] * s_ss[p] * s_BB[(p - P.S6) * 17] : s_ss[p] * s_ss[p] * LG[sl][(p - cp * 6) * 17];
        acc[r] += a;
      } else if (o == P.pc_fail) {
        acc[r] += ok ? 0.0 : 1.0;
      } else if (o < P.pc_gmax) {
        const int p = o - P.pc_gs;
        const bool pk = HK && p >= P.S6;
        const int cp = pk ? 0 : p / 6, sl = pk ? 0 : s_slot[cp];
        if (sl >= 0) a = pk ? s_gk[p - P.S6] : LG[sl][(p - cp * 6) * 16 + 12];
        acc[r] += a;
      } else {
        for (int i = 0; i < 6; ++i) a = fmax(a, fabs(s_A[21 + i]));
        acc[r] = fmax(acc[r], a);
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int r = 0; r < kRigOwn; ++r) {
    const int o = tid + 256 * r;
    if (o < P.PC) P.partial[(size_t)blockIdx.x * P.PC + o] = acc[r];
  }
}

// ---------------------------------------------------------------------------------------------
// reduce: column sums (max for the last column) of the elimination partial rows, 16 columns per
// block, 16 row groups per column, 16 loads in flight per thread. Deterministic.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rig_reduce(RigDev P) {
  __shared__ double s_r[16][16];
  __shared__ double s_post[48];
  __shared__ double s_tail;
  const LmCtl* cn = P.ctl_next;
  if (cn->done || cn->phase == 0) return;
  const int tid = threadIdx.x, c = tid & 15, grp = tid >> 4;  // 16 columns x 16 row groups per block
  const int o = blockIdx.x * 16 + c;
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
  s_r[grp][c] = a;
  __syncthreads();
  if (tid < 16 && o < P.PC) {
    double r = 0.0;
    if (is_max) { for (int g2 = 0; g2 < 16; ++g2) r = fmax(r, s_r[g2][c]); }
    else {
      r = (((s_r[0][c] + s_r[1][c]) + (s_r[2][c] + s_r[3][c])) + ((s_r[4][c] + s_r[5][c]) + (s_r[6][c] + s_r[7][c]))) +
          (((s_r[8][c] + s_r[9][c]) + (s_r[10][c] + s_r[11][c])) + ((s_r[12][c] + s_r[13][c]) + (s_r[14][c] + s_r[15][c])));
    }
    if (is_max) { P.vec[P.PC + P.rank] = r; s_tail = r; r = 0.0; }  // a sum all-reduce then carries the max
    P.vec[o] = r;
    s_post[c] = r;
  }
  if (P.x.on) {
    // mailbox exchange (kind 0): every block posts its 16 column sums straight into all ranks'
    // mailboxes; the block that owns the max column also posts the 32 per-rank max slots (ours set,
    // the others zero). k_rig_solve collects. The epoch is stable here: only k_rig_solve advances it.
    __syncthreads();
    const unsigned long long epoch = P.x.seq[0] + 1ull;
    const int first = blockIdx.x * 16;
    const int ncol = P.PC - first < 16 ? P.PC - first : 16;
    p2p_post(P.x, 0, epoch, P.rank, P.nranks, s_post, ncol, first);
    if (first <= P.pc_gmax && P.pc_gmax < first + 16) {
      if (tid < 32) s_post[16 + tid] = tid == P.rank ? s_tail : 0.0;
      __syncthreads();
      p2p_post(P.x, 0, epoch, P.rank, P.nranks, s_post + 16, 32, P.PC);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// solve (one block of 256): dense Cholesky of the reduced (6C)x(6C) system in LDS (all four waves,
// one barrier per column), then wave 0 alone: substitutions with lane i owning b[i] (cross-lane
// values through v_readlane, no barrier), gradient test, camera candidates. In phase 0 it only
// prepares the camera records. Tried and dropped (C4, S = 24): a single-wave factorisation (33 us vs
// 20: the read-modify-write chain through LDS has nothing to hide behind) and a register-tiled one
// with only the pivot column crossing threads through LDS (23 us).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double readlane_d(double x, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(x), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(x), l);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  return v;
}

__global__ __launch_bounds__(256) void k_rig_solve(RigDev P) {
  __shared__ double A[kRigMaxS][kRigMaxS + 1];
  __shared__ double s_inv[64];
  __shared__ int s_ok;
  const int tid = threadIdx.x, lane = tid & 63, i = tid;
  const LmCtl* cn = P.ctl_next;
  const int done = cn->done, phase = cn->phase, cur = cn->cur;
  const int S = P.S;
  if (done) {
    if (tid == 0) *P.ctl = *cn;
    return;
  }
  const bool my_cam_fixed = lane < P.C && P.cam_fixed[lane] != 0;
  const unsigned long long fixed_mask = __ballot(my_cam_fixed);  // bit c: camera c is held constant (every wave)
  const bool row = tid < S;                                       // rows live in wave 0 (S <= 60)
  // columns 0..S6-1 belong to cameras (6 each), S6..S-1 to the shared intrinsics (extension)
  const int S6 = P.S6;
  const uint32_t kmask = P.kmask;
  const bool row_fixed = row && (i < S6 ? ((fixed_mask >> (i / 6)) & 1ull) != 0 : ((kmask >> (i - S6)) & 1u) != 0);
  bool step_ok = false, converged = false;
  int early_term = CC_CONVERGENCE_GRADIENT;
  double gmax = 0.0;
  double bi = 0.0;
  if (phase != 0 && P.x.on) {
    // mailbox exchange (kind 0): wait for every rank's column sums (posted by k_rig_reduce), add them
    // in rank order into P.vec; everything below then reads the global sums like on one GPU
    const unsigned long long epoch = P.x.seq[0] + 1ull;
    p2p_collect_to(P.x, 0, epoch, P.rank, P.nranks, P.PC + 32, P.vec, &s_ok);
    if (tid == 0) P.x.seq[0] = epoch;
    if (s_ok == 0) {
      if (tid == 0) {
        LmCtl c = *cn;
        c.done = 1; c.term = CC_FAILURE_EXCHANGE;
        *P.ctl = c; *P.ctl_next = c;
      }
      return;
    }
    __syncthreads();
  }
  if (phase != 0) {
    // ---- reduced sums from k_rig_reduce: packed upper triangle -> lower triangle in LDS; rows and
    // columns of constant cameras become identity. Everything is issued in one round trip.
    const double b_in = row ? P.vec[P.pc_b + i] : 0.0;
    const double gs_i = row ? P.vec[P.pc_gs + i] : 0.0;
    const double fail = P.vec[P.pc_fail];
    const double gm_r = (tid < P.nranks && tid < 32) ? P.vec[P.PC + tid] : 0.0;
    const LmOpts o = *P.opts;
    const double radius = cn->radius;
    for (int idx = tid; idx < P.NP; idx += 256) {
      const int p = P.pair_p[idx], q = P.pair_q[idx];  // p <= q
      double a = P.vec[idx];
      const bool pf = p < S6 ? ((fixed_mask >> (p / 6)) & 1ull) != 0 : ((kmask >> (p - S6)) & 1u) != 0;
      const bool qf = q < S6 ? ((fixed_mask >> (q / 6)) & 1ull) != 0 : ((kmask >> (q - S6)) & 1u) != 0;
      if (pf || qf) a = p == q ? 1.0 : 0.0;
      else if (p == q) a += clampd(P.vec[P.pc_hd + p], o.min_lm_diagonal, o.max_lm_diagonal) / radius;
      A[q][p] = a;
    }
    if (tid == 0) s_ok = fail > 0.0 ? 0 : 1;
    // gradient test (wave 0 holds the rows; the other waves compute the same uniform answer from LDS)
    __shared__ double s_g;
    if (tid < 64) {
      const double g = wave_max(fmax(gm_r, (row && !row_fixed) ? fabs(gs_i) : 0.0));
      if (tid == 0) s_g = g;
    }
    __syncthreads();
    gmax = s_g;
#if CC_ABLATE_RS == 1
    return;
#endif
    // second half of FinalizeIterationAndCheckIfMinimizerCanContinue (cf. lm_finalize): gradient tolerance, then
    // minimum trust-region radius
    converged = gmax <= o.gradient_tolerance;
    if (!converged && radius < o.min_radius) { converged = true; early_term = CC_MIN_RADIUS; }
    if (!converged) {
      // right-looking Cholesky, lower triangle in place, ONE barrier per step: every thread derives
      // 1/sqrt(pivot) itself, the trailing update uses the unscaled column times inv^2, and the
      // scaled column is written in the same step by the threads that own it.
      const int ti = tid >> 4, tk = tid & 15;  // 16 x 16 thread tile over the trailing block
      for (int j = 0; j < S; ++j) {
        const double d = A[j][j];
        if (tid == 0 && (!(d > 0.0) || !isfinite(d))) s_ok = 0;
        const double inv = rsqrt(d), inv2 = inv * inv;
        for (int r = j + 1 + ti; r < S; r += 16) {
          const double arj = A[r][j] * inv2;
          for (int k = j + 1 + tk; k <= r; k += 16) A[r][k] -= arj * A[k][j];
        }
        __syncthreads();
        if (tid > j && tid < S) A[tid][j] *= inv;
        if (tid == j) { A[j][j] = d * inv; s_inv[j] = inv; }
        // (column j is read again only by the substitutions, after the final barrier)
      }
      __syncthreads();
#if CC_ABLATE_RS == 2
      return;
#endif
      if (tid < 64) {
        // forward / backward substitution on wave 0: lane i owns b[i]
        const double inv_own = row ? s_inv[i] : 0.0;
        bi = row_fixed ? 0.0 : b_in;
        // the factor entries a lane needs do not depend on the running solution: fetch them eight
        // steps ahead so that only the lane reads and the FMA sit on the dependent chain
        for (int j0 = 0; j0 < S; j0 += 8) {
          double a[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) a[u] = (row && j0 + u < S && i > j0 + u) ? A[i][j0 + u] : 0.0;
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int j = j0 + u;
            if (j < S) {
              const double yj = readlane_d(bi, j) * readlane_d(inv_own, j);
              if (i == j) bi = yj;
              else bi -= a[u] * yj;
            }
          }
        }
        for (int j0 = S - 1; j0 >= 0; j0 -= 8) {
          double a[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) a[u] = (j0 - u >= 0 && i < j0 - u) ? A[j0 - u][i] : 0.0;
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int j = j0 - u;
            if (j >= 0) {
              const double xj = readlane_d(bi, j) * readlane_d(inv_own, j);
              if (i == j) bi = xj;
              else bi -= a[u] * xj;
            }
          }
        }
        step_ok = s_ok != 0 && __all(!row || isfinite(bi));
        if (row) P.ds[i] = -bi;
      }
    }
  }
#if CC_ABLATE_RS == 3
  if (phase != 0) return;
#endif
  if (tid >= 64) return;
  // ---- wave 0: camera candidates / records: lane c < C gathers its six step components from lanes 6c..6c+5
  const int dst = phase == 0 ? cur : (cur ^ 1);
  double step2 = 0.0, xn2 = 0.0;
  double dcv[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) dcv[k] = __shfl(bi, (lane < P.C ? lane : 0) * 6 + k, 64);
  if (lane < P.C && (phase == 0 || step_ok)) {
    const int c = lane;
    const double* pc = P.cam + ((size_t)cur * P.C + c) * 8;
    double q[4] = {pc[0], pc[1], pc[2], pc[3]}, t[3] = {pc[4], pc[5], pc[6]};
    double dc[6] = {0, 0, 0, 0, 0, 0};
    if (phase != 0) {
      if (!my_cam_fixed) {
        for (int k = 0; k < 6; ++k) dc[k] = -dcv[k] * P.ss[c * 6 + k];
        double qn[4];
        quat_plus(q, dc, qn);
        for (int k = 0; k < 4; ++k) { const double d = qn[k] - q[k]; step2 += d * d; q[k] = qn[k]; }
        for (int k = 0; k < 3; ++k) { const double tn = t[k] + dc[3 + k]; const double d = tn - t[k]; step2 += d * d; t[k] = tn; }
        xn2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3] + t[0] * t[0] + t[1] * t[1] + t[2] * t[2];
      }
      double* pd = P.cam + ((size_t)dst * P.C + c) * 8;
      for (int k = 0; k < 4; ++k) pd[k] = q[k];
      for (int k = 0; k < 3; ++k) pd[4 + k] = t[k];
    }
    double R[9];
    quat_to_R(q, R);
    double* rec = P.camrec + c * 32;
    for (int k = 0; k < 9; ++k) rec[k] = R[k];
    for (int k = 0; k < 3; ++k) rec[9 + k] = t[k];
    for (int k = 0; k < 6; ++k) rec[12 + k] = dc[k];
  }
  if (P.K && lane >= S6 && lane < S6 + P.K) {
    // extension: candidate intrinsics (lane S6 + j owns the step of intrinsic j); every intrinsic counts
    // in |x| (cf. IntrinsicsProblem), frozen ones do not move
    const int j = lane - S6;
    const double kc = P.intr[cur * 16 + j];
    double dk = 0.0;
    if (phase != 0 && step_ok && !((kmask >> j) & 1u)) dk = -bi * P.ss[lane];
    const double kn = kc + dk;
    if (phase == 0 || step_ok) {
      if (phase != 0) { P.intr[dst * 16 + j] = kn; step2 += dk * dk; xn2 = kn * kn; }
      P.krec[j] = kn;
      P.krec[16 + j] = dk;
    }
  }
  const double st = wave_sum(step2);
  const double xs = wave_sum(xn2);
  if (lane == 0) {
    LmCtl c = *cn;
    if (phase != 0) {
      c.gmax = gmax;
      if (c.log_len > 0 && c.log_len <= P.log_cap && P.log[c.log_len - 1].accepted)
        P.log[c.log_len - 1].gradient_max_norm = gmax;
      if (converged) { c.done = 1; c.term = early_term; }
      else {
        c.step_valid = step_ok ? 1 : 0;
        c.cand_pending = 1;
        P.shared_stats[0] = st;
        P.shared_stats[1] = xs;
      }
    }
    *P.ctl = c;
    *P.ctl_next = c;
  }
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
  for (int64_t idx = P.goff[g] + threadIdx.x; idx < P.goff[g + 1]; idx += blockDim.x) {
    const float2 m = uv2[idx];
    const int64_t w = P.widx[idx];
    RigObs o;
    rig_common(Rf, pf + 4, Rc, pc + 4, (double)P.wxyz[w * 3], (double)P.wxyz[w * 3 + 1], (double)P.wxyz[w * 3 + 2],
               (double)m.x, (double)m.y, o);
    double ru = o.ru, rv = o.rv;
    if (P.K) {
      RigKObs ko;
      rigk_obs(P.intr + cur * 16, o, (double)m.x, (double)m.y, ko);
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

struct cc_rig {
  int device = 0;
  hipStream_t stream = nullptr;
  cc::RigDev d{};
  int64_t C = 0, F = 0, N = 0, NG = 0, P = 0;
  std::vector<int64_t> perm;  // sorted position -> caller's observation index
  std::vector<void*> allocs;    // the chunks dev_alloc carves buffers from
  char* chunk_cur = nullptr;
  size_t chunk_left = 0;
  double* init_cam = nullptr;
  double* init_pose = nullptr;
  double* d_cost = nullptr;
  bool have_state = false;
  cc::LmCtl* h_ctl = nullptr;
  hipGraphExec_t graph[2] = {nullptr, nullptr};
  int graph_iters = 0;
  cc::Comm* comm = nullptr;
  cc::Mailbox mailbox;          // mailbox exchange (cc_rig_exchange_export / _attach)
  double* init_intr = nullptr;  // [16] (extension)
  bool have_intr = false;
  bool exchange = false;
  uint8_t* d_cam_fixed = nullptr;          // same memory as d.cam_fixed
  std::vector<uint8_t> frozen, seen;       // host copies (user freeze flags, locally observed cameras)
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
    CC_HIP(hipMalloc(&c, sz));
    h->allocs.push_back(c);
    h->chunk_cur = static_cast<char*>(c);
    h->chunk_left = sz;
  }
  *p = reinterpret_cast<T*>(h->chunk_cur);
  h->chunk_cur += bytes;
  h->chunk_left -= bytes;
  return 0;
}
template <class T>
static int dev_upload(cc_rig* h, const T** p, const std::vector<T>& v) {
  T* q = nullptr;
  if (int rc = dev_alloc(h, &q, v.size())) return rc;
  if (!v.empty()) CC_HIP(hipMemcpy(q, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  *p = q;
  return 0;
}

static void rig_drop_graphs(cc_rig* h) {
  for (auto& g : h->graph)
    if (g) { hipGraphExecDestroy(g); g = nullptr; }
}

// One round: reduce -> solve -> update -> sweep -> decide+elim. The very first round of a solve is the
// initial evaluation: there is nothing to reduce yet (k_rig_reduce would return at once), and only
// that round needs k_rig_init (Jacobi scale of the shared block, trust-region state).
static int rig_enqueue_round(cc_rig* h, bool initial) {
  const RigDev& d = h->d;
  if (!initial) {
    hipLaunchKernelGGL(k_rig_reduce, dim3((unsigned)((d.PC + 15) / 16)), dim3(256), 0, h->stream, d);
    if (h->comm) if (int rc = comm_allreduce_sum(h->comm, d.vec, d.PC + 32, h->stream)) return rc;  // (mailbox: posted by the kernel)
  }
  hipLaunchKernelGGL(k_rig_solve, dim3(1), dim3(256), 0, h->stream, d);
  hipLaunchKernelGGL(k_rig_update, dim3((unsigned)((h->F + 15) / 16)), dim3(256), 0, h->stream, d);
  if (d.K) hipLaunchKernelGGL(k_rig_sweep<true>, dim3((unsigned)h->NG), dim3(kRigThreads), kRigSweepLdsBytesK, h->stream, d);
  else hipLaunchKernelGGL(k_rig_sweep<false>, dim3((unsigned)h->NG), dim3(kRigThreads), kRigSweepLdsBytes, h->stream, d);
  if (h->comm || h->exchange) {
    hipLaunchKernelGGL(k_rig_stats, dim3(1), dim3(256), 0, h->stream, d);
    if (h->comm) if (int rc = comm_allreduce_sum(h->comm, d.vec_stats, 4 + d.S, h->stream)) return rc;
  }
  if (initial) hipLaunchKernelGGL(k_rig_init, dim3(1), dim3(256), 0, h->stream, d);
  if (d.K) hipLaunchKernelGGL(k_rig_decide_elim<true>, dim3(d.nblk), dim3(256), 0, h->stream, d);
  else hipLaunchKernelGGL(k_rig_decide_elim<false>, dim3(d.nblk), dim3(256), 0, h->stream, d);
  return 0;
}

static int rig_write_ctl(cc_rig* h, const LmCtl& c) {
  CC_HIP(hipMemcpyAsync(h->d.ctl, &c, sizeof(c), hipMemcpyHostToDevice, h->stream));
  CC_HIP(hipMemcpyAsync(h->d.ctl_next, &c, sizeof(c), hipMemcpyHostToDevice, h->stream));
  return 0;
}
static int rig_read_ctl(cc_rig* h, LmCtl* c) {
  CC_HIP(hipMemcpyAsync(h->h_ctl, h->d.ctl_next, sizeof(LmCtl), hipMemcpyDeviceToHost, h->stream));
  CC_HIP(hipStreamSynchronize(h->stream));
  *c = *h->h_ctl;
  return 0;
}

}  // namespace cc

extern "C" void cc_rig_destroy(cc_rig* h);

namespace cc {
struct RigCreateGuard {  // releases a half-built handle on every early return
  cc_rig* h;
  bool ok = false;
  ~RigCreateGuard() { if (!ok) cc_rig_destroy(h); }
};
}  // namespace cc

extern "C" {

static int rig_create_impl(int32_t device, int64_t C, int64_t F, int64_t n_world, const int64_t* off,
                           const uint32_t* obs_cam, const uint64_t* obs_world, const float* obs_uv,
                           const float* world_xyz, const uint8_t* cam_frozen, double huber_a, int K, cc_rig** out) {
  using namespace cc;
  if (!out || !off || C < 1 || F < 1 || n_world < 0) return fail(CC_ERR_BAD_ARGUMENT, "cc_rig_create: bad arguments");
  if (C > kRigMaxCams) return fail(CC_ERR_BAD_ARGUMENT, "cc_rig_create: at most %d cameras", kRigMaxCams);
  if (6 * C + K > kRigMaxS) return fail(CC_ERR_BAD_ARGUMENT, "cc_rigk_create: at most %d cameras with shared intrinsics", (kRigMaxS - K) / 6);
  if (off[0] != 0) return fail(CC_ERR_BAD_ARGUMENT, "obs_frame_offsets[0] must be 0");
  const int64_t N = off[F];
  for (int64_t f = 0; f < F; ++f)
    if (off[f + 1] < off[f]) return fail(CC_ERR_BAD_ARGUMENT, "obs_frame_offsets must be non-decreasing");
  if (N > 0 && (!obs_cam || !obs_world || !obs_uv || !world_xyz)) return fail(CC_ERR_BAD_ARGUMENT, "cc_rig_create: NULL arrays");
  for (int64_t k = 0; k < N; ++k) {
    if (obs_cam[k] >= (uint32_t)C) return fail(CC_ERR_BAD_ARGUMENT, "observation %lld: camera id out of range", (long long)k);
    if (obs_world[k] >= (uint64_t)n_world) return fail(CC_ERR_BAD_ARGUMENT, "observation %lld: world point id out of range", (long long)k);
  }
  if (int rc = select_device(device)) return rc;
  cc_rig* h = new cc_rig();
  RigCreateGuard guard{h};
  h->device = device; h->C = C; h->F = F; h->N = N; h->P = n_world;
  // ---- regroup: within each frame, stable sort by camera -> (frame, camera) groups
  h->perm.resize((size_t)N);
  std::vector<int64_t> goff{0}, fgoff((size_t)F + 1, 0);
  std::vector<int32_t> gframe, gcam;
  std::vector<uint8_t> seen((size_t)C, 0);
  {
    // counting sort by camera inside each frame (stable: observation order is kept within a group)
    int64_t pos = 0;
    std::vector<int64_t> cnt((size_t)C), start((size_t)C);
    for (int64_t f = 0; f < F; ++f) {
      std::fill(cnt.begin(), cnt.end(), 0);
      for (int64_t k = off[f]; k < off[f + 1]; ++k) cnt[obs_cam[k]]++;
      for (int64_t c = 0; c < C; ++c) {
        start[(size_t)c] = pos;
        if (cnt[(size_t)c] > 0) {
          gframe.push_back((int32_t)f);
          gcam.push_back((int32_t)c);
          seen[(size_t)c] = 1;
          pos += cnt[(size_t)c];
          goff.push_back(pos);
        }
      }
      for (int64_t k = off[f]; k < off[f + 1]; ++k) h->perm[(size_t)start[obs_cam[k]]++] = k;
      fgoff[(size_t)f + 1] = (int64_t)gframe.size();
    }
  }
  const int64_t NG = (int64_t)gframe.size();
  h->NG = NG;
  if (NG == 0) return fail(CC_ERR_BAD_ARGUMENT, "cc_rig_create: no observations");
  std::vector<float> uv((size_t)N * 2);
  std::vector<int32_t> widx((size_t)N);
  for (int64_t i = 0; i < N; ++i) {
    const int64_t k = h->perm[(size_t)i];
    uv[2 * i] = obs_uv[2 * k]; uv[2 * i + 1] = obs_uv[2 * k + 1];
    widx[(size_t)i] = (int32_t)obs_world[k];
  }
  std::vector<int32_t> cam_goff((size_t)C + 1, 0), cam_glist((size_t)NG);
  for (int64_t g = 0; g < NG; ++g) cam_goff[(size_t)gcam[g] + 1]++;
  for (int64_t c = 0; c < C; ++c) cam_goff[c + 1] += cam_goff[c];
  {
    std::vector<int32_t> fill(cam_goff.begin(), cam_goff.end() - 1);
    for (int64_t g = 0; g < NG; ++g) cam_glist[(size_t)fill[gcam[g]]++] = (int32_t)g;
  }
  std::vector<uint8_t> fixed((size_t)C);
  for (int64_t c = 0; c < C; ++c) fixed[c] = ((cam_frozen && cam_frozen[c]) || !seen[c]) ? 1 : 0;
  const int S = (int)(6 * C) + K;
  std::vector<uint8_t> pp, pq;
  for (int p = 0; p < S; ++p) for (int q = p; q < S; ++q) { pp.push_back((uint8_t)p); pq.push_back((uint8_t)q); }

  if (int rc = stream_get(device, &h->stream)) return rc;
  RigDev& d = h->d;
  d.F = F; d.N = N; d.NG = NG; d.C = (int32_t)C; d.S = S; d.SW = S + 1; d.NP = S * (S + 1) / 2;
  d.pc_b = d.NP; d.pc_hd = d.NP + S; d.pc_fail = d.NP + 2 * S; d.pc_gs = d.pc_fail + 1; d.pc_gmax = d.pc_gs + S; d.PC = d.pc_gmax + 1;
  d.nblk = (int)std::min<int64_t>(kRigMaxElimBlocks, F);
  d.huber_a = (K && !(huber_a > 0.0)) ? 1e300 : huber_a;   // extension: a <= 0 switches the loss off
  d.comm = 0; d.rank = 0; d.nranks = 1;
  d.K = K; d.S6 = (int32_t)(6 * C); d.gstride = K ? 768 : 256; d.kmask = 0;
  h->frozen.assign((size_t)C, 0);
  if (cam_frozen) for (int64_t c = 0; c < C; ++c) h->frozen[(size_t)c] = cam_frozen[c] ? 1 : 0;
  h->seen = seen;
  if (d.PC > 256 * kRigOwn) return fail(CC_ERR_BAD_ARGUMENT, "cc_rig_create: too many cameras");
  { const float* p; if (int rc = dev_upload(h, &p, uv)) return rc; d.uv = p; }
  if (int rc = dev_upload(h, &d.widx, widx)) return rc;
  {
    float* w = nullptr;
    if (int rc = dev_alloc(h, &w, (size_t)n_world * 3)) return rc;
    if (n_world > 0) CC_HIP(hipMemcpy(w, world_xyz, (size_t)n_world * 3 * sizeof(float), hipMemcpyHostToDevice));
    d.wxyz = w;
  }
  if (int rc = dev_upload(h, &d.goff, goff)) return rc;
  if (int rc = dev_upload(h, &d.gframe, gframe)) return rc;
  if (int rc = dev_upload(h, &d.gcam, gcam)) return rc;
  if (int rc = dev_upload(h, &d.fgoff, fgoff)) return rc;
  if (int rc = dev_upload(h, &d.cam_goff, cam_goff)) return rc;
  if (int rc = dev_upload(h, &d.cam_glist, cam_glist)) return rc;
  if (int rc = dev_upload(h, &d.cam_fixed, fixed)) return rc;
  h->d_cam_fixed = const_cast<uint8_t*>(d.cam_fixed);
  if (int rc = dev_upload(h, &d.pair_p, pp)) return rc;
  if (int rc = dev_upload(h, &d.pair_q, pq)) return rc;
  if (int rc = dev_alloc(h, &d.cam, (size_t)2 * C * 8)) return rc;
  if (int rc = dev_alloc(h, &d.pose, (size_t)2 * F * 8)) return rc;
  if (int rc = dev_alloc(h, &d.camrec, (size_t)C * 32)) return rc;
  if (int rc = dev_alloc(h, &d.frec, (size_t)F * 32)) return rc;
  if (int rc = dev_alloc(h, &d.gblocks, (size_t)2 * NG * d.gstride)) return rc;
  if (int rc = dev_alloc(h, &d.intr, (size_t)32)) return rc;
  if (int rc = dev_alloc(h, &d.krec, (size_t)32)) return rc;
  if (int rc = dev_alloc(h, &d.ghdk, (size_t)(K ? NG * 16 : 16))) return rc;
  if (int rc = dev_alloc(h, &h->init_intr, (size_t)16)) return rc;
  CC_HIP(hipMemset(d.intr, 0, 32 * sizeof(double)));
  CC_HIP(hipMemset(d.krec, 0, 32 * sizeof(double)));
  CC_HIP(hipMemset(d.ghdk, 0, (size_t)(K ? NG * 16 : 16) * sizeof(double)));
  CC_HIP(hipMemset(h->init_intr, 0, 16 * sizeof(double)));
  if (int rc = dev_alloc(h, &d.gstats, (size_t)NG * 2)) return rc;
  if (int rc = dev_alloc(h, &d.fstats, (size_t)F * 2)) return rc;
  if (int rc = dev_alloc(h, &d.ghd0, (size_t)NG * 8)) return rc;
  if (int rc = dev_alloc(h, &d.sp, (size_t)F * 8)) return rc;
  if (int rc = dev_alloc(h, &d.ss, (size_t)64)) return rc;
  if (int rc = dev_alloc(h, &d.ds, (size_t)64)) return rc;
  if (int rc = dev_alloc(h, &d.Y, (size_t)F * 6 * d.SW)) return rc;
  if (int rc = dev_alloc(h, &d.partial, (size_t)d.nblk * d.PC)) return rc;
  if (int rc = dev_alloc(h, &d.vec, (size_t)d.PC + 32)) return rc;
  if (int rc = dev_alloc(h, &d.vec_stats, (size_t)4 + kRigMaxS)) return rc;
  if (int rc = dev_alloc(h, &d.shared_stats, (size_t)4)) return rc;
  if (int rc = dev_alloc(h, &d.ctl, (size_t)1)) return rc;
  if (int rc = dev_alloc(h, &d.ctl_next, (size_t)1)) return rc;
  if (int rc = dev_alloc(h, &d.opts, (size_t)1)) return rc;
  d.log_cap = 4096;
  if (int rc = dev_alloc(h, &d.log, (size_t)d.log_cap)) return rc;
  if (int rc = dev_alloc(h, &h->init_cam, (size_t)C * 8)) return rc;
  if (int rc = dev_alloc(h, &h->init_pose, (size_t)F * 8)) return rc;
  if (int rc = dev_alloc(h, &h->d_cost, (size_t)N)) return rc;
  CC_HIP(hipMemset(d.cam, 0, (size_t)2 * C * 8 * sizeof(double)));
  CC_HIP(hipMemset(d.pose, 0, (size_t)2 * F * 8 * sizeof(double)));
  CC_HIP(hipMemset(d.sp, 0, (size_t)F * 8 * sizeof(double)));
  CC_HIP(hipMemset(d.ss, 0, 64 * sizeof(double)));
  CC_HIP(hipMemset(d.ds, 0, 64 * sizeof(double)));
  CC_HIP(hipMemset(d.Y, 0, (size_t)F * 6 * d.SW * sizeof(double)));
  CC_HIP(hipMemset(d.gstats, 0, (size_t)NG * 2 * sizeof(double)));
  CC_HIP(hipMemset(d.fstats, 0, (size_t)F * 2 * sizeof(double)));
  CC_HIP(hipMemset(d.ghd0, 0, (size_t)NG * 8 * sizeof(double)));
  CC_HIP(hipMemset(d.frec, 0, (size_t)F * 32 * sizeof(double)));
  CC_HIP(hipMemset(d.camrec, 0, (size_t)C * 32 * sizeof(double)));
  CC_HIP(hipMemset(d.partial, 0, (size_t)d.nblk * d.PC * sizeof(double)));
  CC_HIP(hipMemset(d.vec, 0, ((size_t)d.PC + 32) * sizeof(double)));
  CC_HIP(hipMemset(d.vec_stats, 0, ((size_t)4 + kRigMaxS) * sizeof(double)));
  CC_HIP(hipMemset(d.shared_stats, 0, 4 * sizeof(double)));
  CC_HIP(hipMemset(d.ctl, 0, sizeof(LmCtl)));
  CC_HIP(hipMemset(d.ctl_next, 0, sizeof(LmCtl)));
  h->h_ctl = reinterpret_cast<LmCtl*>(pinned_block_get());
  if (!h->h_ctl) return fail(CC_ERR_HIP, "hipHostMalloc failed");
  CC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_rig_sweep<false>), hipFuncAttributeMaxDynamicSharedMemorySize, kRigSweepLdsBytes));
  CC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_rig_sweep<true>), hipFuncAttributeMaxDynamicSharedMemorySize, kRigSweepLdsBytesK));
  guard.ok = true;
  *out = h;
  return CC_OK;
}

int cc_rig_create(int32_t device, int64_t C, int64_t F, int64_t n_world, const int64_t* off,
                  const uint32_t* obs_cam, const uint64_t* obs_world, const float* obs_uv,
                  const float* world_xyz, const uint8_t* cam_frozen, double huber_a, cc_rig** out) {
  return rig_create_impl(device, C, F, n_world, off, obs_cam, obs_world, obs_uv, world_xyz, cam_frozen, huber_a, 0, out);
}

// EXTENSION: the same handle with 9 intrinsics shared by all cameras; obs_uv in pixels
int cc_rigk_create(int32_t device, int64_t C, int64_t F, int64_t n_world, const int64_t* off,
                   const uint32_t* obs_cam, const uint64_t* obs_world, const float* obs_uv_pixels,
                   const float* world_xyz, const uint8_t* cam_frozen, double huber_a, cc_rig** out) {
  return rig_create_impl(device, C, F, n_world, off, obs_cam, obs_world, obs_uv_pixels, world_xyz, cam_frozen, huber_a,
                         cc::kRigK, out);
}

int cc_rigk_set_intrinsics(cc_rig* h, const double* intr9, uint32_t const_mask) {
  using namespace cc;
  if (!h || !intr9) return fail(CC_ERR_BAD_ARGUMENT, "cc_rigk_set_intrinsics: NULL argument");
  if (!h->d.K) return fail(CC_ERR_STATE, "cc_rigk_set_intrinsics: the handle was created without intrinsics (cc_rig_create)");
  CC_HIP(hipSetDevice(h->device));
  CC_HIP(hipStreamSynchronize(h->stream));
  double k16[16] = {0};
  for (int i = 0; i < 9; ++i) k16[i] = intr9[i];
  CC_HIP(hipMemcpy(h->init_intr, k16, sizeof(k16), hipMemcpyHostToDevice));
  if (h->d.kmask != (const_mask & 0x1ffu)) rig_drop_graphs(h);   // the mask is a kernel argument
  h->d.kmask = const_mask & 0x1ffu;
  h->have_intr = true;
  return h->have_state ? cc_rig_reset(h) : CC_OK;
}

int cc_rigk_get_intrinsics(cc_rig* h, double* intr9) {
  using namespace cc;
  if (!h || !intr9) return fail(CC_ERR_BAD_ARGUMENT, "cc_rigk_get_intrinsics: NULL argument");
  if (!h->d.K) return fail(CC_ERR_STATE, "cc_rigk_get_intrinsics: the handle was created without intrinsics");
  CC_HIP(hipSetDevice(h->device));
  LmCtl c;
  if (int rc = rig_read_ctl(h, &c)) return rc;
  double k16[16];
  CC_HIP(hipMemcpy(k16, h->d.intr + (size_t)(c.cur & 1) * 16, sizeof(k16), hipMemcpyDeviceToHost));
  for (int i = 0; i < 9; ++i) intr9[i] = k16[i];
  return CC_OK;
}

void cc_rig_destroy(cc_rig* h) {
  if (!h) return;
  hipSetDevice(h->device);
  if (h->stream) hipStreamSynchronize(h->stream);
  cc::rig_drop_graphs(h);
  if (h->comm) cc::comm_destroy(h->comm);
  cc::mailbox_release(&h->mailbox);
  for (void* p : h->allocs) hipFree(p);
  cc::pinned_block_put(h->h_ctl);
  cc::stream_put(h->device, h->stream);   // synchronised above
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
  CC_HIP(hipMemcpyAsync(h->d.cam, h->init_cam, (size_t)h->C * 8 * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  CC_HIP(hipMemcpyAsync(h->d.pose, h->init_pose, (size_t)h->F * 8 * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  if (h->d.K) CC_HIP(hipMemcpyAsync(h->d.intr, h->init_intr, 16 * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  return CC_OK;
}

int cc_rig_solve(cc_rig* h, const cc_options* opt, cc_summary* summary) {
  using namespace cc;
  if (!h || !h->have_state) return fail(CC_ERR_STATE, "cc_rig_solve: no state set");
  if (h->d.K && !h->have_intr) return fail(CC_ERR_STATE, "cc_rig_solve: cc_rigk_set_intrinsics has not been called");
  const auto t0 = std::chrono::steady_clock::now();
  cc_options o;
  if (opt) o = *opt; else { cc_options_init(&o); o.max_iterations = 1000; }  // extrinsics_calibrator.cpp:211
  if (o.check_interval < 1) o.check_interval = 1;
  if (o.max_iterations > h->d.log_cap - 1) o.max_iterations = h->d.log_cap - 1;
  const bool use_graph = o.use_graph != 0 && !h->comm;
  CC_HIP(hipSetDevice(h->device));
  {
    LmCtl st;
    if (int rc = rig_read_ctl(h, &st)) return rc;
    if (st.cur & 1) {
      CC_HIP(hipMemcpyAsync(h->d.cam, h->d.cam + (size_t)h->C * 8, (size_t)h->C * 8 * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
      CC_HIP(hipMemcpyAsync(h->d.pose, h->d.pose + (size_t)h->F * 8, (size_t)h->F * 8 * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
      if (h->d.K) CC_HIP(hipMemcpyAsync(h->d.intr, h->d.intr + 16, 16 * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    }
    LmOpts lo;
    opts_from_public(o, &lo);
    CC_HIP(hipMemcpyAsync(h->d.opts, &lo, sizeof(lo), hipMemcpyHostToDevice, h->stream));
    LmCtl c{};
    if (int rc = rig_write_ctl(h, c)) return rc;
  }
  if (use_graph && (!h->graph[0] || h->graph_iters != o.check_interval)) {
    rig_drop_graphs(h);
    for (int gi = 0; gi < 2; ++gi) {
      hipGraph_t g = nullptr;
      CC_HIP(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
      const int n = o.check_interval + (gi == 0 ? 1 : 0);
      for (int i = 0; i < n; ++i) rig_enqueue_round(h, gi == 0 && i == 0);
      CC_HIP(hipStreamEndCapture(h->stream, &g));
      CC_HIP(hipGraphInstantiate(&h->graph[gi], g, nullptr, nullptr, 0));
      hipGraphDestroy(g);
    }
    h->graph_iters = o.check_interval;
  }
  int launched = 0;
  LmCtl st;
  for (int chunk = 0;; ++chunk) {
    const int n = o.check_interval + (chunk == 0 ? 1 : 0);
    if (use_graph) {
      CC_HIP(hipGraphLaunch(h->graph[chunk == 0 ? 0 : 1], h->stream));
    } else {
      for (int i = 0; i < n; ++i)
        if (int rc = rig_enqueue_round(h, chunk == 0 && i == 0)) return rc;
      CC_HIP(hipGetLastError());
    }
    launched += n;
    if (int rc = rig_read_ctl(h, &st)) return rc;
    if (st.done && st.term == CC_FAILURE_EXCHANGE)
      return fail(CC_ERR_COMM, "mailbox exchange timed out: a peer rank did not post within 10 s (iteration %d)", st.iter);
    if (st.done) break;
    if (launched > o.max_iterations + 2 * o.check_interval + 2)
      return fail(CC_ERR_STATE, "rig LM loop did not terminate (iter=%d)", st.iter);
  }
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
    summary->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
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
    std::vector<double> sorted((size_t)h->N);
    CC_HIP(hipMemcpyAsync(sorted.data(), h->d_cost, sorted.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    CC_HIP(hipStreamSynchronize(h->stream));
    for (int64_t i = 0; i < h->N; ++i) obs_cost[h->perm[(size_t)i]] = sorted[(size_t)i];
  }
  return CC_OK;
}

int cc_rig_eval(cc_rig* h, double* cost) {
  using namespace cc;
  if (!h || !h->have_state || !cost) return fail(CC_ERR_STATE, "cc_rig_eval: no state set");
  if (h->d.K && !h->have_intr) return fail(CC_ERR_STATE, "cc_rig_eval: cc_rigk_set_intrinsics has not been called");
  std::vector<double> oc((size_t)h->N);
  if (int rc = cc_rig_get_state(h, nullptr, nullptr, nullptr, nullptr, oc.data())) return rc;
  double c = 0.0;
  for (double v : oc) c += v;
  *cost = c;
  return CC_OK;
}

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
  std::vector<double> flags((size_t)4 + kRigMaxS, 0.0);
  for (int64_t c = 0; c < h->C; ++c) flags[(size_t)c] = h->seen[(size_t)c] ? 1.0 : 0.0;
  CC_HIP(hipMemcpyAsync(h->d.vec_stats, flags.data(), flags.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
  if (int rc = comm_allreduce_sum(h->comm, h->d.vec_stats, (int)h->C, h->stream)) return rc;
  CC_HIP(hipMemcpyAsync(flags.data(), h->d.vec_stats, flags.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  CC_HIP(hipStreamSynchronize(h->stream));
  std::vector<uint8_t> fixed((size_t)h->C);
  for (int64_t c = 0; c < h->C; ++c) fixed[(size_t)c] = (h->frozen[(size_t)c] || !(flags[(size_t)c] > 0.0)) ? 1 : 0;
  CC_HIP(hipMemcpy(h->d_cam_fixed, fixed.data(), fixed.size(), hipMemcpyHostToDevice));
  return CC_OK;
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
  return mailbox_export(&h->mailbox, h->d.PC + 32, 4 + kRigMaxS, handle);
}

// collective: every rank must call it (the "camera seen" flags are summed through the mailboxes)
int cc_rig_exchange_attach(cc_rig* h, int32_t rank, int32_t nranks, const uint8_t* handles) {
  using namespace cc;
  if (!h || !handles || rank < 0 || nranks < 1 || rank >= nranks || nranks > kP2pMaxRanks)
    return fail(CC_ERR_BAD_ARGUMENT, "cc_rig_exchange_attach: bad arguments (nranks must be 1..%d)", kP2pMaxRanks);
  if (!h->mailbox.local) return fail(CC_ERR_STATE, "cc_rig_exchange_attach: call cc_rig_exchange_export first");
  if (h->comm) return fail(CC_ERR_STATE, "cc_rig_exchange_attach: an RCCL communicator is already attached");
  CC_HIP(hipSetDevice(h->device));
  rig_drop_graphs(h);
  if (int rc = mailbox_attach(&h->mailbox, rank, nranks, handles, &h->d.x)) return rc;
  h->d.comm = 1; h->d.rank = rank; h->d.nranks = nranks;
  h->exchange = true;
  // a camera is part of the problem if ANY rank observes it: sum the per-rank "seen" flags
  std::vector<double> flags(64, 0.0);
  for (int64_t c = 0; c < h->C; ++c) flags[(size_t)c] = h->seen[(size_t)c] ? 1.0 : 0.0;
  double* d_io = nullptr;
  int* d_ok = nullptr;
  CC_HIP(hipMalloc(&d_io, 128 * sizeof(double)));
  CC_HIP(hipMalloc(&d_ok, sizeof(int)));
  CC_HIP(hipMemcpyAsync(d_io, flags.data(), 64 * sizeof(double), hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(k_rig_flag_exchange, dim3(1), dim3(64), 0, h->stream, h->d, d_io, d_io + 64, (int)h->C, d_ok);
  int ok = 0;
  hipError_t e1 = hipMemcpyAsync(flags.data(), d_io + 64, 64 * sizeof(double), hipMemcpyDeviceToHost, h->stream);
  hipError_t e2 = hipMemcpyAsync(&ok, d_ok, sizeof(int), hipMemcpyDeviceToHost, h->stream);
  hipError_t e3 = hipStreamSynchronize(h->stream);
  hipFree(d_io);
  hipFree(d_ok);
  CC_HIP(e1); CC_HIP(e2); CC_HIP(e3);
  if (!ok) return fail(CC_ERR_COMM, "cc_rig_exchange_attach: a peer rank did not attach within 10 s");
  std::vector<uint8_t> fixed((size_t)h->C);
  for (int64_t c = 0; c < h->C; ++c) fixed[(size_t)c] = (h->frozen[(size_t)c] || !(flags[(size_t)c] > 0.0)) ? 1 : 0;
  CC_HIP(hipMemcpy(h->d_cam_fixed, fixed.data(), fixed.size(), hipMemcpyHostToDevice));
  return CC_OK;
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
  else if (k == "camrec") src = d.camrec; else if (k == "frec") src = d.frec; else if (k == "gstats") src = d.gstats;
  else if (k == "fstats") src = d.fstats; else if (k == "shared_stats") src = d.shared_stats; else if (k == "cam") src = d.cam; else if (k == "pose") src = d.pose;
  else return fail(CC_ERR_BAD_ARGUMENT, "cc_rig_debug_fetch: unknown buffer %s", name);
  CC_HIP(hipMemcpy(out, src, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
  return CC_OK;
}

int cc_rig_optimize(const cc_options* opt, int32_t device, int64_t C, int64_t F, int64_t n_world,
                    const int64_t* off, const uint32_t* obs_cam, const uint64_t* obs_world,
                    const float* obs_uv, const float* world_xyz, double* cam_q, double* cam_t,
                    const uint8_t* cam_frozen, double* frame_q, double* frame_t, double huber_a,
                    double* obs_cost, cc_summary* summary) {
  cc_rig* h = nullptr;
  int rc = cc_rig_create(device, C, F, n_world, off, obs_cam, obs_world, obs_uv, world_xyz, cam_frozen, huber_a, &h);
  if (rc) return rc;
  rc = cc_rig_set_state(h, cam_q, cam_t, frame_q, frame_t);
  cc_options o;
  if (opt) o = *opt; else { cc_options_init(&o); o.max_iterations = 1000; }  // extrinsics_calibrator.cpp:211
  o.use_graph = 0;   // one solve per handle: capturing and instantiating a graph cannot pay off
  if (!rc) rc = cc_rig_solve(h, &o, summary);
  if (!rc) rc = cc_rig_get_state(h, cam_q, cam_t, frame_q, frame_t, obs_cost);
  cc_rig_destroy(h);
  return rc;
}

}  // extern "C"
