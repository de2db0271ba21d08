// cc_rig_sweeps.hpp -- the sweeps of the rig path: per (frame, camera) group (k_rig_sweep_adj), per frame (k_rig_sweep_frame), with intrinsics on the matrix pipe / tiles (k_rig_sweep_adjk) and on plain FMAs / compact records (k_rig_sweep_k2).
// Part of cc_rig.hip (round 5: the 6.8 k-line file split by subject; included by it inside namespace cc, in this order:
// cc_rig_sweeps.hpp, cc_rig_steps.hpp, cc_rig_big.hpp, cc_rig_lean.hpp -- one translation unit, nothing else includes these).
#pragma once

// ---------------------------------------------------------------------------------------------
// sweep of the reference's problem (poses only) WITHOUT the matrix pipe. Inside a (frame, camera) group both poses are
// constants, and every row's frame columns are one 6 x 6 matrix applied to its camera columns (rig_row):
//     J_frame = J_cam * M,   M = | Rc            0  |   rows: camera (rotation, translation), columns: frame,
//                                | 2 [Rc_i x tf] Rc |   Rc_i = i-th row of the camera rotation, tf = frame translation
// (m = Rc^T B and b x m = Rc^T (a x B) - tf x m, a = Rc (b + tf)). So a group needs the Gram of SEVEN columns
// [J_cam(6) r], 28 unique numbers accumulated by each lane on its own observations with plain FMAs (21 per row: the
// normalised-image rows have one structural zero each) -- against 2 x 16 matrix instructions of 64 cycles per 64
// observations for the 16 x 16 product, half of which is the redundant triangle and a quarter padding. The block
// the other kernels read (same 16 x 16 layout [cam frame r]) is assembled once per group: CF = CC M, FF = M^T CC M,
// g_f = M^T g_c. A fixed camera zeroes its own blocks AFTER the frame blocks were derived from them.
// Rounding differs from the 13-column product by O(eps (|a| / |b|)^2) in the frame-rotation block (a: point relative to
// the camera pose's origin, b: rotated world point).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void untri(int idx, int& i, int& j) {  // packed lower index -> (i >= j)
  i = 0;
  while (tri(i + 1, 0) <= idx) ++i;
  j = idx - tri(i, 0);
}
// 1 / z for the depth of a point in front of the camera (z far from the ends of the exponent range): the hardware
// estimate and two Newton steps, five instructions instead of the twelve of the IEEE division sequence (scaling, fix-up).
// Not correctly rounded: within an ulp or two of 1 / z, so the adjoint sweeps (the default) are not bit-identical to
// the division form that k_rig_obs_cost and the oracle keep (parity is to the stated tolerances under either;
// the division form is kept as a variant, scripts/variants/exact_arith.patch). Degenerate depths: z = 0 (and z = +-inf) give NaN here (0 * inf inside
// the first fma) where the division gives +-inf / 0. Both are "not finite" to everything downstream -- the candidate
// cost fails isfinite() in lm_trial and counts as DBL_MAX, a Gram block holding either fails the Cholesky's
// `d > 0 && isfinite(d)` test -> invalid step -> the radius shrinks -- so a point that lands on the camera plane is
// rejected the same way in both forms; a select on the result would cost three instructions per observation of ~165.
// A point BEHIND the camera (z < 0) is an ordinary finite value in both.
__device__ __forceinline__ double recip_depth(double z) {
  double r = __builtin_amdgcn_rcp(z);
  r = fma(fma(-z, r, 1.0), r, r);
  r = fma(fma(-z, r, 1.0), r, r);
  return r;
}

template <int SKIP>
__device__ __forceinline__ void adj_accumulate(const double* w, double* acc) {
#pragma unroll
  for (int i = 0; i < 7; ++i) {
#pragma unroll
    for (int j = 0; j <= i; ++j) {
      if (i != SKIP && j != SKIP) acc[i * (i + 1) / 2 + j] = fma(w[i], w[j], acc[i * (i + 1) / 2 + j]);
    }
  }
}

// 32 per-lane values -> their 64-lane sums, value e left in lanes 2e and 2e + 1. Each exchange halves the values a lane
// carries (31 exchanges and adds instead of 32 x 6), and none of them goes through the LDS crossbar: the two widest
// are the lane-swap instructions of gfx950 (v_permlane32_swap: lanes 32..63 of the first register <-> lanes 0..31 of the
// second; v_permlane16_swap: odd 16-lane rows of the first <-> even rows of the second -- after either, first + second
// is the pairwise sum of the first register's values in the lower lanes / even rows and of the second's in the others),
// the rest DPP moves inside a row. Partner masks 32, 16, 8, 7 (half-row mirror), 2, 1 are independent, so every value
// collects all 64 lanes; the lane bit that picks the half kept in each step (5, 4, 3, 2, 1) differs between partners and
// all earlier ones agree.
template <int N>
__device__ __forceinline__ void reduce_swap32(double* p) {
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const auto l = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(p[i]), (unsigned)__double2loint(p[i + N]), false, false);
    const auto h = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(p[i]), (unsigned)__double2hiint(p[i + N]), false, false);
    p[i] = __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
  }
}
template <int N>
__device__ __forceinline__ void reduce_swap16(double* p) {
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const auto l = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(p[i]), (unsigned)__double2loint(p[i + N]), false, false);
    const auto h = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(p[i]), (unsigned)__double2hiint(p[i + N]), false, false);
    p[i] = __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
  }
}
template <int N, int CTRL, int BIT>
__device__ __forceinline__ void reduce_dpp(double* p, int lane) {
  const bool up = (lane & BIT) != 0;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const double lo = p[i], hi = p[i + N];
    const double send = up ? lo : hi, keep = up ? hi : lo;
    p[i] = keep + dpp_f64<CTRL>(send);
  }
}
__device__ __forceinline__ void reduce_scatter32(double* p, int lane) {
  reduce_swap32<16>(p);
  reduce_swap16<8>(p);
  reduce_dpp<4, 0x128, 8>(p, lane);   // row_ror:8
  reduce_dpp<2, 0x141, 4>(p, lane);   // row_half_mirror
  reduce_dpp<1, 0x4E, 2>(p, lane);    // quad_perm:[2,3,0,1]
  p[0] += dpp_f64<0xB1>(p[0]);        // quad_perm:[1,0,3,2]
}

constexpr int kRigAdjWaves = 4;    // waves per SIMD the one-wave-per-group sweep is compiled for (128 registers)
// The sweep of one group as a function: k_rig_sweep_adj (one workgroup per group) and the persistent per-solve kernel
// (k_rig_persist: WL -- "wave-local": the caller is ONE wave of a larger workgroup sweeping the groups of its frame one
// after the other, so there is no workgroup barrier in here, the scratch `lds` is the wave's own, and the camera records
// come from `camrec` -- there: the copy the control workgroup broadcast, in LDS).
constexpr int kRigSweepAdjLds(int NW) { return 64 + 64 + 8 + NW * 32 + 32 + 36; }   // doubles of scratch
// where a group's sweep reads and leaves things: global memory (the stand-alone kernels, k_rig_persist) or the LDS of a
// workgroup that keeps its frames resident (k_rig_persist_w)
struct RigSweepIO {
  const double* camrec;     // [C][32] camera records of the point to evaluate
  const double* frec;       // [32]    record of the group's frame
  const double* comp_old;   // [64]    compact record of the group at the accepted point
  double* block_out;        // [256]   the group's 16 x 16 block at the evaluated point
  double* comp_out;         // [64]    its compact record
  double* stats_out;        // [2]     cost, model-cost term
  double* hd0_out;          // [8]     diagonal of H_cc (first evaluation)
};
__device__ __forceinline__ RigSweepIO rig_sweep_io_global(const RigDev& P, int64_t g, int cur, int dst) {
  return RigSweepIO{P.camrec, P.frec + (size_t)P.gframe[g] * 32, P.gcomp + ((size_t)cur * P.NG + g) * 64,
                    P.gblocks + ((size_t)dst * P.NG + g) * (size_t)P.gstride, P.gcomp + ((size_t)dst * P.NG + g) * 64, P.gstats + g * 2, P.ghd0 + g * 8};
}
template <int NW, bool WL>
__device__ __forceinline__ void rig_sweep_adj_body(const RigDev& P, const int64_t g, const int phase, const int cur, double* lds, const RigSweepIO io) {
  constexpr int NT = NW * 64;      // threads
  double* sm = lds;                    // [64] camera record [0..31], frame record [32..63]
  double* s_old = sm + 64;             // [64] the accepted point's compact record of this group (gcomp)
  double* s_e = s_old + 64;            // [8]  step of the seven columns: e = dc + M_old df, 1
  double* s_red = s_e + 8;             // [NW * 32] per wave: 28 Gram sums, cost, model-cost term
  double* s_g = s_red + NW * 32;       // [32] their totals
  double* s_m = s_g + 32;              // [36] M
  auto sync = [] { if (WL) wave_lds_fence(); else __syncthreads(); };
  int tid_ = WL ? (int)(threadIdx.x & 63) : (int)threadIdx.x;
  if (WL) asm volatile("" : "+v"(tid_));   // (a fresh copy per call: nothing derived from it is hoisted out of the persistent kernel's round loop)
  const int tid = tid_, lane = tid & 63, wave = tid >> 6;
  const int c = P.gcam[g];
  const int64_t s0 = P.goff[g], s1 = P.goff[g + 1];
  const bool fixed = P.cam_fixed[c] != 0;
  // chunks dealt to the waves starting at wave (g mod NW), observations fetched one pass ahead by unconditional loads:
  // (idle slots re-read the group's first observation)
  const int otid = (((tid >> 6) - (int)(g & (NW - 1))) & (NW - 1)) * 64 + lane;
  const int n = (int)(s1 - s0);                                   // observations of the group (never empty)
  const int wrem = n - (otid >> 6) * 64;
  const int npass = wrem > 0 ? (wrem + NT - 1) / NT : 0;
  // Observations are fetched TWO passes ahead (two register sets, the loop unrolled by two): one pass of the other
  // waves on the SIMD does not cover the memory latency once the whole chip streams. Unconditional loads (idle slots
  // re-read the group's first observation), 32-bit offsets from the group's uniform base.
  const float2* uvg = reinterpret_cast<const float2*>(P.uv) + s0;
  struct F3 { float x, y, z; };
  const F3* xg = reinterpret_cast<const F3*>(P.oxyz) + s0;
  struct ObsRaw { float2 m; F3 X; };
  auto fetch = [&](int k, ObsRaw& r) {
    const int kc = k < n ? k : 0;
    r.m = uvg[kc];
    r.X = xg[kc];
  };
  ObsRaw oa, ob;
  fetch(otid, oa);
  fetch(otid + NT, ob);
  if (tid < 64) {
    // (both buffers hold valid memory: the record of `cur` is read whatever the phase, used only behind phase != 0)
    const double rec = tid < 32 ? io.camrec[c * 32 + tid] : io.frec[tid - 32];
    const double old = io.comp_old[tid];
    sm[tid] = rec;
    s_old[tid] = old;
  }
  sync();
  double acc[32];
#pragma unroll
  for (int e = 0; e < 32; ++e) acc[e] = 0.0;
  // the chain of both poses as one: a = Rc (Rf X + tf) = Rca X + tca (the frame columns are not formed row by row, so
  // the rotated world point is not needed on its own). Uniform addresses: scalar loads, the values live in SGPRs.
  double Rca[9], tca[3], tcs[3];
  {
    const double* cr = io.camrec + (size_t)c * 32;
    const double* fr = io.frec;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
      for (int j = 0; j < 3; ++j) Rca[3 * i + j] = rfl(cr[3 * i] * fr[j] + cr[3 * i + 1] * fr[3 + j] + cr[3 * i + 2] * fr[6 + j]);
      tca[i] = rfl(cr[3 * i] * fr[9] + cr[3 * i + 1] * fr[10] + cr[3 * i + 2] * fr[11]);
      tcs[i] = cr[9 + i];
    }
  }
  if (tid < 48) {   // M[a][b], a = tid >> 3, b = tid & 7 < 6
    const int a = tid >> 3, b = tid & 7;
    const int a3 = a < 3 ? a : a - 3, b3 = b < 3 ? b : b - 3;
    const int b1 = b3 == 2 ? 0 : b3 + 1, b2 = b3 == 0 ? 2 : b3 - 1;
    const double diag = sm[3 * a3 + (b3 < 3 ? b3 : 0)];
    const double cross = 2.0 * (sm[3 * a3 + b1] * sm[32 + 9 + b2] - sm[3 * a3 + b2] * sm[32 + 9 + b1]);   // 2 (Rc_i x tf)_b
    const double v = (a < 3) == (b < 3) ? diag : (a >= 3 ? cross : 0.0);
    if (b < 6) s_m[a * 6 + b] = v;
  }
  // model-cost term of the group at the accepted point, q = d^T g + 1/2 d^T H d with d = [dc df]: every row's J d is
  // J_cam e, e = dc + M_old df (dc = 0 for a fixed camera), so q = 1/2 (e' G7 e' - G7[6][6]) with e' = [e 1]
  if (phase != 0 && tid < 8) {
    double e = tid < 6 ? (fixed ? 0.0 : sm[12 + tid]) : (tid == 6 ? 1.0 : 0.0);
    const int row = tid < 6 ? tid : 0;
#pragma unroll
    for (int b = 0; b < 6; ++b) e = fma(tid < 6 ? s_old[28 + row * 6 + b] : 0.0, sm[32 + 12 + b], e);
    s_e[tid] = e;
  }
  const double ha = P.huber_a;
  struct ObsD { double u, v, X0, X1, X2; };
  auto widen = [](const ObsRaw& r, ObsD& d) { d.u = (double)r.m.x; d.v = (double)r.m.y; d.X0 = (double)r.X.x; d.X1 = (double)r.X.y; d.X2 = (double)r.X.z; };
  auto pass = [&](int k, const ObsD& r) {
    const bool valid = k < n;
    const double X0 = r.X0, X1 = r.X1, X2 = r.X2;
    RigObs o;
    o.a0 = Rca[0] * X0 + Rca[1] * X1 + Rca[2] * X2 + tca[0];
    o.a1 = Rca[3] * X0 + Rca[4] * X1 + Rca[5] * X2 + tca[1];
    o.a2 = Rca[6] * X0 + Rca[7] * X1 + Rca[8] * X2 + tca[2];
    o.iz = recip_depth(o.a2 + tcs[2]);
    o.x = (o.a0 + tcs[0]) * o.iz;
    o.y = (o.a1 + tcs[1]) * o.iz;
    o.ru = o.x - r.u;
    o.rv = o.y - r.v;
    double rho, sr;
    huber(ha, o.ru * o.ru + o.rv * o.rv, rho, sr);
    if (valid) acc[28] += 0.5 * rho;
    if (!valid) sr = 0.0;
    // rows as rig_row forms them, camera columns and residual only, with B_u = (iz, 0, -x iz), B_v = (0, iz, -y iz):
    // u: sr [2 Bu2 a1, 2 (Bu0 a2 - Bu2 a0), -2 Bu0 a1, Bu0, 0, Bu2, ru], v: sr [2 (Bv2 a1 - Bv1 a2), -2 Bv2 a0, 2 Bv1 a0, 0, Bv1, Bv2, rv]
    const double pz = sr * o.iz, qu = -(pz * o.x), qv = -(pz * o.y);     // sr Bu0 = sr Bv1, sr Bu2, sr Bv2
    const double pz2 = pz + pz, qu2 = qu + qu, qv2 = qv + qv;
    double w[7];
    w[0] = qu2 * o.a1; w[1] = pz2 * o.a2 - qu2 * o.a0; w[2] = -(pz2 * o.a1);
    w[3] = pz; w[4] = 0.0; w[5] = qu; w[6] = sr * o.ru;
    adj_accumulate<4>(w, acc);
    w[0] = qv2 * o.a1 - pz2 * o.a2; w[1] = -(qv2 * o.a0); w[2] = pz2 * o.a0;
    w[3] = 0.0; w[4] = pz; w[5] = qv; w[6] = sr * o.rv;
    adj_accumulate<3>(w, acc);
  };
  // (each register set is widened to doubles BEFORE it is refilled: the loaded registers are then dead and the refill
  // reuses them -- a set kept alive across its own refill would be rotated by copies that wait for every load in flight)
  int p = 0;
  for (; p + 1 < npass; p += 2) {    // pairs of passes, no branch inside (a conditional second half brings the copies back)
    const int k = p * NT + otid;
    ObsD d;
    widen(oa, d);
    fetch(k + 2 * NT, oa);
    pass(k, d);
    widen(ob, d);
    fetch(k + 3 * NT, ob);
    pass(k + NT, d);
  }
  if (p < npass) {
    ObsD d;
    widen(oa, d);
    pass(p * NT + otid, d);
  }
  if (phase != 0 && tid < 27) {
    int i, j;
    untri(tid, i, j);
    acc[29] = (i == j ? 0.5 : 1.0) * s_e[i] * s_e[j] * s_old[tid];
  }
  reduce_scatter32(acc, lane);   // value e in lanes 2e, 2e + 1
  const int ve = lane >> 1;
  if (NW > 1) {
    if ((lane & 1) == 0) s_red[wave * 32 + ve] = acc[0];
    sync();
    if (tid < 32) {
      double t = s_red[tid];
#pragma unroll
      for (int w2 = 1; w2 < NW; ++w2) t += s_red[w2 * 32 + tid];
      s_g[tid] = t;
    }
  } else if ((lane & 1) == 0) {
    s_g[ve] = acc[0];
  }
  sync();
  if (wave != 0) return;
  // The block the other kernels read, [cam frame r]^2 in a 16 x 16 tile, is N^T G7 N with N (7 x 13) = [I6 M 0; 0 0 1]
  // (the identity zeroed for a fixed camera): two matrix products, T = G7 N and N^T T. The first product's result rows
  // k and k + 4 sit in the very lanes that feed them to the second as its B operand.
  const int k0 = lane >> 4, j = lane & 15;
  const double mv0 = s_m[k0 * 6 + (j >= 6 && j < 12 ? j - 6 : 0)];
  const double mv1 = s_m[(k0 < 2 ? k0 + 4 : 0) * 6 + (j >= 6 && j < 12 ? j - 6 : 0)];
  const double id = fixed ? 0.0 : 1.0;
  const double n0 = j < 6 ? (j == k0 ? id : 0.0) : (j < 12 ? mv0 : 0.0);                       // N[k0][j], k0 = 0..3
  const double n1 = k0 < 2 ? (j < 6 ? (j == k0 + 4 ? id : 0.0) : (j < 12 ? mv1 : 0.0))       // N[k0 + 4][j]: rows 4, 5,
                           : (k0 == 2 && j == 12 ? 1.0 : 0.0);                                //   the residual row 6, nothing
  const int gi = j < 7 ? j : 0;
  const int h0 = gi > k0 ? gi : k0, l0 = gi > k0 ? k0 : gi;
  const int k1 = k0 + 4 < 7 ? k0 + 4 : 0;
  const int h1 = gi > k1 ? gi : k1, l1 = gi > k1 ? k1 : gi;
  const double gv0 = s_g[h0 * (h0 + 1) / 2 + l0], gv1 = s_g[h1 * (h1 + 1) / 2 + l1];
  const double a0 = j < 7 ? gv0 : 0.0;                       // G7[j][k0]
  const double a1 = (j < 7 && k0 + 4 < 7) ? gv1 : 0.0;       // G7[j][k0 + 4]
  d4 T = {0.0, 0.0, 0.0, 0.0};
  T = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, n0, T, 0, 0, 0);
  T = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, n1, T, 0, 0, 0);
  d4 B = {0.0, 0.0, 0.0, 0.0};
  B = __builtin_amdgcn_mfma_f64_16x16x4f64(n0, T[0], B, 0, 0, 0);
  B = __builtin_amdgcn_mfma_f64_16x16x4f64(n1, T[1], B, 0, 0, 0);
  double* out = io.block_out;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = k0 + 4 * r;
    out[row * 16 + j] = B[r];
    if (phase == 0 && row < 6 && row == j) io.hd0_out[row] = B[r];  // diag of H_cc
  }
  // compact record of this point for the next sweep's model-cost term: G7 (28) and M (36)
  io.comp_out[lane] = lane < 28 ? s_g[lane < 28 ? lane : 0] : s_m[lane >= 28 ? lane - 28 : 0];
  if (lane == 0) {
    io.stats_out[0] = s_g[28];
    io.stats_out[1] = s_g[29];
  }
}

template <int NW>
__global__ __launch_bounds__(NW * 64, NW == 1 ? kRigAdjWaves : 3) void k_rig_sweep_adj(RigDev P) {   // (NW > 1: few, large groups -- registers rather than residency)
  __shared__ double s_lds[kRigSweepAdjLds(NW)];
  const LmCtl* ctl = P.ctl;
  const int done = ctl->done, phase = ctl->phase, step_valid = ctl->step_valid, cur = ctl->cur;
  if (done) return;
  if (phase != 0 && !step_valid) return;
  rig_sweep_adj_body<NW, false>(P, blockIdx.x, phase, cur, s_lds, rig_sweep_io_global(P, blockIdx.x, cur, phase == 0 ? cur : (cur ^ 1)));
}

// ---------------------------------------------------------------------------------------------
// FRAME form of the poses-only sweep (round 4; the default of the three-kernel path): one workgroup per FRAME, its NWF waves
// deal the frame's (frame, camera) groups among themselves and sweep them one after the other with the main loop of
// k_rig_sweep_adj (7-column Gram per group, plain FMAs). What changes is everything AROUND that loop:
//   * the frame record is read once per wave, not per group; a wave requests its NEXT group's first observations before it
//     reduces the current one;
//   * nothing but the 28 numbers of G7 leaves the group: the 16 x 16 tile N^T G7 N of k_rig_sweep_adj (four matrix
//     instructions behind ~150 instructions of operand set-up per group, 2 KB written per group and read back by the
//     elimination) is never formed. Wave 0 ends the frame with ONE assembly for all its groups, eight lanes per group:
//     T = G_cc M (the 6 x 6 coupling block the elimination's camera columns are made of), the group's share M^T T of the frame
//     block and M^T g_c of its gradient, added over the groups by lane exchanges. A group's record is [G7 (28) | T (36)],
//     the frame's [H_ff (21) | g_f (6)]: 64 + 32/CO doubles per group where the tile form wrote 256 + 64;
//   * the model-cost term of the step needs no adjoint of the accepted point any more: with the OLD records
//     q = sum_g (1/2 dc' G_cc dc + dc' g_c + dc' T df) + 1/2 df' H_ff df + df' g_f   (dc = 0 for a camera held constant);
//   * cost and model-cost term are ONE row per frame (gstats[f]): the elimination's statistics pass reads F rows instead
//     of NG (BASELINE configs[4]: 2000 instead of 16000 in each of its 256 blocks).
// Arithmetic of a row and of G7: k_rig_sweep_adj's, instruction for instruction (same sums in the same order per lane, same
// butterfly). LDS (dynamic): per group slot 32 doubles (G7, cost), staging 64 per group of an assembly pass, small scratch.
// ---------------------------------------------------------------------------------------------
constexpr int kRigFrameLdsDoubles(int CO) { return CO * 32 + 8 * 64 + 64 + CO * 16; }
// ONE: every wave sweeps at most ONE group (a frame has no more groups than the workgroup has waves: rigs of up to eight
// observed cameras) -- no loop over groups, and the kernel fits the 128 registers of four waves per SIMD like k_rig_sweep_adj<1>
// does; with the loop (more groups than waves) the passes spill 12 - 19 registers at 128, so that variant is compiled for
// three waves per SIMD (141 registers).
template <int NWF, bool ONE>
__global__ __launch_bounds__(NWF * 64, ONE ? 4 : (NWF <= 4 ? 3 : 2)) void k_rig_sweep_frame(RigDev P) {
  extern __shared__ __attribute__((aligned(16))) double sf_lds[];
  double* s_G = sf_lds;                     // [CO][32]  G7 (28), cost (28) of every group of the frame
  double* s_rec = s_G + (size_t)P.CO * 32;  // [8][64]   records of an assembly pass, staged for one coalesced store
  double* s_fr = s_rec + 8 * 64;            // [64]      frame record of the evaluated point (32), then scratch
  double* s_cam = s_fr + 64;                // [CO][16]  per group: its camera's rotation (9), unscaled step (6), held-constant flag -- left here by the
                                            //           wave that sweeps the group, at its START, for the assembly at the workgroup's end (round 5: the
                                            //           assembly fetched them itself, group -> camera -> record, two dependent round trips on the tail)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t f = blockIdx.x;
  const double* fr = P.frec + (size_t)f * 32;
  const double ha = P.huber_a;
  const double hb = P.huber_b, h2a = P.huber_2a, hha = P.huber_ha;
  struct F3 { float x, y, z; };
  struct ObsRaw { float2 m; F3 X; };
  struct ObsD { double u, v, X0, X1, X2; };
  auto widen = [](const ObsRaw& r, ObsD& d) { d.u = (double)r.m.x; d.v = (double)r.m.y; d.X0 = (double)r.X.x; d.X1 = (double)r.X.y; d.X2 = (double)r.X.z; };
  // ---- the wave's groups, one after the other. The first two passes' observations are requested BEFORE the control block
  // is looked at: a launch that returns at once wastes two loads per lane, every other one starts its longest chain
  // (slot record -> observations) with the kernel.
  ObsRaw oa, ob;
  int64_t g0, s0_one = 0;
  int ng, n_one = 0, c_one = 0;
  constexpr bool FW = ONE;
  if (FW) {
    const int4 sl = P.fwave[f * 8 + wave];
    s0_one = (int64_t)(((unsigned long long)(unsigned)sl.y << 32) | (unsigned)sl.x);
    n_one = __builtin_amdgcn_readfirstlane(sl.z);
    c_one = __builtin_amdgcn_readfirstlane(sl.w);
    const float2* uvg = reinterpret_cast<const float2*>(P.uv) + s0_one;
    const F3* xg = reinterpret_cast<const F3*>(P.oxyz) + s0_one;
    const int k0 = lane < n_one ? lane : 0, k1 = lane + 64 < n_one ? lane + 64 : 0;
    oa.m = uvg[k0]; oa.X = xg[k0];
    ob.m = uvg[k1]; ob.X = xg[k1];
    g0 = P.fgoff[f];
    ng = (int)(P.fgoff[f + 1] - g0);            // groups of this frame (0: no observation)
  } else {
    g0 = P.fgoff[f];
    ng = (int)(P.fgoff[f + 1] - g0);
    const int64_t gq = g0 + (wave < ng ? wave : 0);
    const int64_t s0 = P.goff[gq < P.NG ? gq : 0], s1 = P.goff[(gq < P.NG ? gq : 0) + 1];
    const int n = (int)(s1 - s0);
    const float2* uvg = reinterpret_cast<const float2*>(P.uv) + s0;
    const F3* xg = reinterpret_cast<const F3*>(P.oxyz) + s0;
    const int k0 = lane < n ? lane : 0, k1 = lane + 64 < n ? lane + 64 : 0;
    oa.m = uvg[k0]; oa.X = xg[k0];
    ob.m = uvg[k1]; ob.X = xg[k1];
  }
  const LmCtl* ctl = P.ctl;
  const int done = ctl->done, phase = ctl->phase, step_valid = ctl->step_valid, cur = ctl->cur;
  if (done) return;
  if (phase != 0 && !step_valid) return;
  const int dst = phase == 0 ? cur : (cur ^ 1);
  auto sweep_group = [&](const int j) {
    const int64_t g = g0 + j;
    // (the camera index is uniform, and the compiler must know it: the camera record then comes by scalar loads)
    const int c = FW ? c_one : __builtin_amdgcn_readfirstlane(P.gcam[g]);
    const int64_t s0 = FW ? s0_one : P.goff[g];
    const int n = FW ? n_one : (int)(P.goff[g + 1] - s0);
    const int npass = (n + 63) >> 6;
    const float2* uvg = reinterpret_cast<const float2*>(P.uv) + s0;
    const F3* xg = reinterpret_cast<const F3*>(P.oxyz) + s0;
    auto fetch = [&](int k, ObsRaw& r) {
      const int kc = k < n ? k : 0;
      r.m = uvg[kc];
      r.X = xg[kc];
    };
    if (lane < 16) {   // the group's camera for the assembly (one coalesced load, requested with the first observations)
      const double v = lane < 15 ? P.camrec[(size_t)c * 32 + (lane < 9 ? lane : lane + 3)] : (double)P.cam_fixed[c];
      s_cam[j * 16 + lane] = v;
    }
    // the chain of both poses as one: a = Rc (Rf X + tf) = Rca X + tca. Uniform addresses: scalar loads, values in SGPRs.
    double Rca[9], tca[3], tcs[3];
    {
      const double* cr = P.camrec + (size_t)c * 32;
      // (the frame record is re-read -- scalar loads, twelve values -- for every group: read once above the loop, the copies
      // the products need in vector registers stay alive across the whole loop and are spilled: 28 registers of scratch)
      const double* frl = fr;
      asm volatile("" : "+s"(frl));
#pragma unroll
      for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int k = 0; k < 3; ++k) Rca[3 * i + k] = rfl(cr[3 * i] * frl[k] + cr[3 * i + 1] * frl[3 + k] + cr[3 * i + 2] * frl[6 + k]);
        tca[i] = rfl(cr[3 * i] * frl[9] + cr[3 * i + 1] * frl[10] + cr[3 * i + 2] * frl[11]);
        tcs[i] = rfl(cr[9 + i]);
      }
    }
    double acc[32];
#pragma unroll
    for (int e = 0; e < 32; ++e) acc[e] = 0.0;
    auto pass = [&](int k, const ObsD& r) {
      const bool valid = k < n;
      const double X0 = r.X0, X1 = r.X1, X2 = r.X2;
      RigObs o;
      o.a0 = Rca[0] * X0 + Rca[1] * X1 + Rca[2] * X2 + tca[0];
      o.a1 = Rca[3] * X0 + Rca[4] * X1 + Rca[5] * X2 + tca[1];
      o.a2 = Rca[6] * X0 + Rca[7] * X1 + Rca[8] * X2 + tca[2];
      o.iz = recip_depth(o.a2 + tcs[2]);
      o.x = (o.a0 + tcs[0]) * o.iz;
      o.y = (o.a1 + tcs[1]) * o.iz;
      o.ru = o.x - r.u;
      o.rv = o.y - r.v;
      double rho, sr;
      {
        const double ss = o.ru * o.ru + o.rv * o.rv;
        if (ss > hb) {
          double rr;
          huber_outlier(ha, ss, rr, sr);
          rho = h2a * (rr - hha);
        } else { rho = ss; sr = 1.0; }
      }
      if (valid) acc[28] += 0.5 * rho;
      if (!valid) sr = 0.0;
      const double pz = sr * o.iz, qu = -(pz * o.x), qv = -(pz * o.y);
      const double pz2 = pz + pz, qu2 = qu + qu, qv2 = qv + qv;
      double w[7];
      w[0] = qu2 * o.a1; w[1] = pz2 * o.a2 - qu2 * o.a0; w[2] = -(pz2 * o.a1);
      w[3] = pz; w[4] = 0.0; w[5] = qu; w[6] = sr * o.ru;
      adj_accumulate<4>(w, acc);
      w[0] = qv2 * o.a1 - pz2 * o.a2; w[1] = -(qv2 * o.a0); w[2] = pz2 * o.a0;
      w[3] = 0.0; w[4] = pz; w[5] = qv; w[6] = sr * o.rv;
      adj_accumulate<3>(w, acc);
    };
    int p = 0;
    for (; p + 1 < npass; p += 2) {
      const int k = p * 64 + lane;
      ObsD d;
      widen(oa, d);
      fetch(k + 128, oa);
      pass(k, d);
      widen(ob, d);
      fetch(k + 192, ob);
      pass(k + 64, d);
    }
    if (p < npass) {
      ObsD d;
      widen(oa, d);
      pass(p * 64 + lane, d);
    }
    reduce_scatter32(acc, lane);   // value e in lanes 2e, 2e + 1
    if ((lane & 1) == 0 && (lane >> 1) < 29) s_G[j * 32 + (lane >> 1)] = acc[0];
    // (loop form only) the NEXT group's first two passes (requested behind the reduction: held across it, the ten registers of the two sets
    // push the butterfly over the kernel's 128 and spill)
    if (!ONE) {
      const int jn = j + NWF < ng ? j + NWF : j;
      const int64_t gn = g0 + jn;
      const int64_t t0 = P.goff[gn], t1 = P.goff[gn + 1];
      const int nn = (int)(t1 - t0);
      const float2* uvn = reinterpret_cast<const float2*>(P.uv) + t0;
      const F3* xn = reinterpret_cast<const F3*>(P.oxyz) + t0;
      const int k0 = lane < nn ? lane : 0, k1 = lane + 64 < nn ? lane + 64 : 0;
      oa.m = uvn[k0]; oa.X = xn[k0];
      ob.m = uvn[k1]; ob.X = xn[k1];
    }
  };
  // (ONE: straight-line code -- as a loop, even one that runs once, the compiler hoists the Huber constants and lane
  // predicates out of it and keeps them in registers across the passes: sixteen spilled at 128)
  if (ONE) { if (wave < ng) sweep_group(wave); }
  else for (int j = wave; j < ng; j += NWF) sweep_group(j);
  if (tid < 32) s_fr[tid] = fr[tid];
  if (NWF > 1) __syncthreads(); else wave_lds_fence();
  // model-cost term of the step at the accepted point: per group from its OLD record (lanes l < 6: row a = l), 1/2 dc_a (G_cc dc)_a
  // + dc_a g_c,a + dc_a (T df)_a, and the frame's own block from the old frame record. It needs nothing of THIS sweep's sums, so
  // in a workgroup of several waves WAVE 1 forms it while wave 0 assembles the frame (round 5: the old records' round trip was
  // on wave 0's chain, behind the barrier).
  const double* comp_old = P.gcomp + (size_t)cur * P.NG * 64;
  auto model_cost_group = [&](int j, int gi_, int l_) -> double {
    if (phase == 0) return 0.0;
    const bool live = j < ng;
    const int64_t g = g0 + (live ? j : 0);
    const double* crl = s_cam + (size_t)(live ? j : 0) * 16;
    const bool fixed = crl[15] != 0.0;
    const int a = l_ < 6 ? l_ : 0;
    const double* old = comp_old + (size_t)g * 64;
    double gd = 0.0, td = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const int hi = a > k ? a : k, lo = a > k ? k : a;
      gd = fma(old[hi * (hi + 1) / 2 + lo], fixed ? 0.0 : crl[9 + k], gd);
      td = fma(old[28 + a * 6 + k], s_fr[12 + k], td);
    }
    const double dca = fixed ? 0.0 : crl[9 + a];
    (void)gi_;
    return (live && l_ < 6) ? dca * (0.5 * gd + old[21 + a] + td) : 0.0;
  };
  auto model_cost_frame = [&](int gi_, int l_) -> double {
    if (!(phase != 0 && gi_ == 0 && l_ < 6 && ng > 0)) return 0.0;
    const double* fo = P.fsum + ((size_t)cur * P.F + f) * 32;
    double hd = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const int hi = l_ > k ? l_ : k, lo = l_ > k ? k : l_;
      hd = fma(fo[hi * (hi + 1) / 2 + lo], s_fr[12 + k], hd);
    }
    return s_fr[12 + l_] * (0.5 * hd + fo[21 + l_]);
  };
  if (NWF > 1 && wave == 1) {
    int lane_q = threadIdx.x & 63;
    asm volatile("" : "+v"(lane_q));
    const int gq = lane_q >> 3, lq = lane_q & 7;
    double q = model_cost_frame(gq, lq);
    for (int jb = 0; jb < ng; jb += 8) q += model_cost_group(jb + gq, gq, lq);
    q = wave_sum(q);
    if (lane_q == 0) P.gstats[f * 2 + 1] = q;
    return;
  }
  if (wave != 0) return;
  // ---- the frame's assembly: eight lanes per group, eight groups per pass. Lane (gi, l): l < 6 owns column l of T and of the
  // group's share of H_ff; l == 6 the gradient column (g_c -> M^T g_c); l == 7 idles.
  // (every per-lane index below comes from a LAUNDERED copy of the lane id: derived from the original they are hoisted above
  // the group loop and kept alive -- spilled -- across its passes)
  int lane_a = threadIdx.x & 63;
  asm volatile("" : "+v"(lane_a));
  const int gi = lane_a >> 3, l = lane_a & 7;
  const double tf0 = s_fr[9], tf1 = s_fr[10], tf2 = s_fr[11];
  double hsum[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};   // column l of H_ff (l < 6) / g_f (l == 6), over this lane's groups
  double cost = 0.0, qm = 0.0;
  double* comp_dst = P.gcomp + (size_t)dst * P.NG * 64;
  for (int jb = 0; jb < ng; jb += 8) {
    const int j = jb + gi;
    const bool live = j < ng;
    const int64_t g = g0 + (live ? j : 0);
    const double* crl = s_cam + (size_t)(live ? j : 0) * 16;   // [0..8] rotation, [9..14] step, [15] held constant
    const bool fixed = crl[15] != 0.0;
    if (NWF == 1) qm += model_cost_group(j, gi, l);   // (workgroups of several waves: wave 1's, below)
    double Rc[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) Rc[i] = crl[i];
    const double* G = s_G + (size_t)(live ? j : 0) * 32;
    // K[i][b] = 2 (Rc_i x tf)_b: the rotation block of the adjoint's lower left
    double K[9];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      K[3 * i + 0] = 2.0 * (Rc[3 * i + 1] * tf2 - Rc[3 * i + 2] * tf1);
      K[3 * i + 1] = 2.0 * (Rc[3 * i + 2] * tf0 - Rc[3 * i + 0] * tf2);
      K[3 * i + 2] = 2.0 * (Rc[3 * i + 0] * tf1 - Rc[3 * i + 1] * tf0);
    }
    // column l of M: M = [Rc 0; K Rc]
    double mc[6];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int b3 = l < 3 ? l : (l < 6 ? l - 3 : 0);
      const double rv = b3 == 0 ? Rc[3 * k] : (b3 == 1 ? Rc[3 * k + 1] : Rc[3 * k + 2]);
      const double kv = b3 == 0 ? K[3 * k] : (b3 == 1 ? K[3 * k + 1] : K[3 * k + 2]);
      mc[k] = l < 3 ? rv : 0.0;
      mc[3 + k] = l < 3 ? kv : rv;
    }
    // tcol = column l of T = G_cc M (l < 6), or g_c (l == 6)
    double tcol[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      double t = 0.0;
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const int hi = r > k ? r : k, lo = r > k ? k : r;
        t = fma(G[hi * (hi + 1) / 2 + lo], mc[k], t);
      }
      tcol[r] = l < 6 ? t : G[21 + r];
    }
    // u = M^T tcol: u[a'] = sum_k M[k][a'] tcol[k]
    double u[6];
#pragma unroll
    for (int ap = 0; ap < 3; ++ap) {
      u[ap] = Rc[ap] * tcol[0] + Rc[3 + ap] * tcol[1] + Rc[6 + ap] * tcol[2] + K[ap] * tcol[3] + K[3 + ap] * tcol[4] + K[6 + ap] * tcol[5];
      u[3 + ap] = Rc[ap] * tcol[3] + Rc[3 + ap] * tcol[4] + Rc[6 + ap] * tcol[5];
    }
    if (live && l < 7) {
#pragma unroll
      for (int r = 0; r < 6; ++r) hsum[r] += u[r];
    }
    if (live && l == 7) cost += G[28];
    // stage the record [G7 | T] of the pass's groups, then one coalesced store per group
    if (l < 6) {
#pragma unroll
      for (int r = 0; r < 6; ++r) s_rec[gi * 64 + 28 + r * 6 + l] = tcol[r];
    }
    for (int e = l; e < 28; e += 8) s_rec[gi * 64 + e] = G[e];
    wave_lds_fence();
    {
      const int nb = ng - jb < 8 ? ng - jb : 8;
      double* out = comp_dst + (size_t)(g0 + jb) * 64;
      for (int e = lane_a; e < nb * 64; e += 64) out[e] = s_rec[e];
      if (phase == 0 && live && l < 6) P.ghd0[g * 8 + l] = fixed ? 0.0 : G[l * (l + 1) / 2 + l];
    }
    wave_lds_fence();
  }
  // ---- sums over the lanes that share l (the groups of the frame): lane bits 3, 4, 5
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    hsum[r] += __shfl_xor(hsum[r], 8, 64);
    hsum[r] += __shfl_xor(hsum[r], 16, 64);
    hsum[r] += __shfl_xor(hsum[r], 32, 64);
  }
  // frame record of the evaluated point: H_ff (packed lower triangle: entry (r, l), r >= l, from column l) and g_f
  double* fs = P.fsum + ((size_t)dst * P.F + f) * 32;
  if (gi == 0 && l < 6) {
#pragma unroll
    for (int r = 0; r < 6; ++r)
      if (r >= l) fs[r * (r + 1) / 2 + l] = hsum[r];
  }
  if (gi == 0 && l == 6) {
#pragma unroll
    for (int r = 0; r < 6; ++r) fs[21 + r] = hsum[r];
  }
  if (NWF == 1) qm += model_cost_frame(gi, l);
  cost = wave_sum(cost);
  if (NWF == 1) qm = wave_sum(qm);
  if (lane_a == 0) {
    P.gstats[f * 2] = cost;
    if (NWF == 1) P.gstats[f * 2 + 1] = qm;
  }
}

// ---------------------------------------------------------------------------------------------
// EXTENSION (pixel observations through the camera's intrinsics): the same idea with the matrix pipe. A row's 22 columns
// [J_cam(6) J_frame(6) r | J_k(9)] carry only SIXTEEN independent ones, X = [J_cam(6) r J_k(9)]: one 16 x 16 product
// per row set instead of the two of the first formulation (all 22 columns of a row through the matrix pipe: retired in round 5), one staged tile instead of two, no frame columns to
// form. The three tiles the other kernels read are assembled per group from G = X^T X and N (7 x 13) = [I6 M 0; 0 0 1]:
// AA = N^T G[0:7, 0:7] N, AB = N^T G[0:7, 7:16], BB = G[7:16, 7:16]. The compact record kept for the next sweep's
// model-cost term is G itself and M: q = 1/2 (e'^T G e' - G[6][6]), e' = [dc + M_old df, 1, dk].
// ---------------------------------------------------------------------------------------------
constexpr int kRigAdjkWaves = 3;   // waves per SIMD the sweep with intrinsics is compiled for
constexpr int kRigCompK = 320;   // doubles per group and buffer of the compact record with intrinsics: G (256), M (36)
// NW = waves per workgroup: one when the groups alone fill the chip (every wave then amortises the prologue, the
// cross-lane epilogue and the assembly over all passes of its group and there is no cross-wave reduction), four otherwise.
template <int NW>
__global__ __launch_bounds__(NW * 64, kRigAdjkWaves) void k_rig_sweep_adjk(RigDev P) {
  constexpr int NT = NW * 64, EPT = 256 / NT;
  __shared__ __attribute__((aligned(16))) double s_stage[NW * kStageDoublesPerWave];   // per wave 64 x 16; then the partial products
  __shared__ double sm[96];        // camera record, frame record, intrinsics record (candidate [0..8], step [16..24])
  __shared__ double s_G[256];      // G
  __shared__ double s_mold[36];    // M of the accepted point
  __shared__ double s_e[16];       // e'
  __shared__ double s_m[36];       // M
  __shared__ double s_w[8];        // per wave: model-cost term, cost
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t g = blockIdx.x;
  const int f = P.gframe[g], c = P.gcam[g];
  const int64_t s0 = P.goff[g], s1 = P.goff[g + 1];
  const LmCtl* ctl = P.ctl;
  const int done = ctl->done, phase = ctl->phase, step_valid = ctl->step_valid, cur = ctl->cur;
  if (done) return;
  if (phase != 0 && !step_valid) return;
  const int dst = phase == 0 ? cur : (cur ^ 1);
  const bool fixed = P.cam_fixed[c] != 0;
  const int ks = P.kset[c];
  const int otid = (((tid >> 6) - (int)(g & (NW - 1))) & (NW - 1)) * 64 + lane;
  const int n = (int)(s1 - s0);
  const int wrem = n - (otid >> 6) * 64;
  const int npass = wrem > 0 ? (wrem + NT - 1) / NT : 0;
  const float2* uvg = reinterpret_cast<const float2*>(P.uv) + s0;
  struct F3 { float x, y, z; };
  const F3* xg = reinterpret_cast<const F3*>(P.oxyz) + s0;
  struct ObsRaw { float2 m; F3 X; };
  auto fetch = [&](int k, ObsRaw& r) {
    const int kc = k < n ? k : 0;
    r.m = uvg[kc];
    r.X = xg[kc];
  };
  ObsRaw oa, ob;
  fetch(otid, oa);
  fetch(otid + NT, ob);
  const double* comp_old = P.gcomp + ((size_t)cur * P.NG + g) * kRigCompK;
  {
    // records: camera [0..31], frame [32..63], intrinsics [64..95]; M of the accepted point
    const double r0 = tid < 32 ? P.camrec[c * 32 + tid] : (tid < 64 ? P.frec[(size_t)f * 32 + (tid - 32)] : P.krec[ks * 32 + ((tid - 64) & 31)]);
    const double r1 = P.krec[ks * 32 + (tid & 31)];
    const double mo = comp_old[256 + (tid < 36 ? tid : 0)];
    if (tid < 96) sm[tid] = r0;
    if (NW == 1 && tid < 32) sm[64 + tid] = r1;
    if (tid < 36) s_mold[tid] = mo;
  }
  double g_old[EPT];
#pragma unroll
  for (int e = 0; e < EPT; ++e) g_old[e] = comp_old[tid + e * NT];
  __syncthreads();
  double Rca[9], tca[3], tcs[3], kk[9];
  {
    const double* cr = P.camrec + (size_t)c * 32;
    const double* fr = P.frec + (size_t)f * 32;
    const double* kr = P.krec + (size_t)ks * 32;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
      for (int j = 0; j < 3; ++j) Rca[3 * i + j] = rfl(cr[3 * i] * fr[j] + cr[3 * i + 1] * fr[3 + j] + cr[3 * i + 2] * fr[6 + j]);
      tca[i] = rfl(cr[3 * i] * fr[9] + cr[3 * i + 1] * fr[10] + cr[3 * i + 2] * fr[11]);
      tcs[i] = cr[9 + i];
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) kk[i] = kr[i];
  }
  if (tid < 48) {   // M[a][b], a = tid >> 3, b = tid & 7 < 6 (see k_rig_sweep_adj)
    const int a = tid >> 3, b = tid & 7;
    const int a3 = a < 3 ? a : a - 3, b3 = b < 3 ? b : b - 3;
    const int b1 = b3 == 2 ? 0 : b3 + 1, b2 = b3 == 0 ? 2 : b3 - 1;
    const double diag = sm[3 * a3 + (b3 < 3 ? b3 : 0)];
    const double cross = 2.0 * (sm[3 * a3 + b1] * sm[32 + 9 + b2] - sm[3 * a3 + b2] * sm[32 + 9 + b1]);
    const double v = (a < 3) == (b < 3) ? diag : (a >= 3 ? cross : 0.0);
    if (b < 6) s_m[a * 6 + b] = v;
  } else if (tid < 64) {   // e'
    const int t = tid - 48;
    double e = t < 6 ? (fixed ? 0.0 : sm[12 + t]) : (t == 6 ? 1.0 : sm[64 + 16 + (t - 7)]);
    const int row = t < 6 ? t : 0;
#pragma unroll
    for (int b = 0; b < 6; ++b) e = fma(t < 6 ? s_mold[row * 6 + b] : 0.0, sm[32 + 12 + b], e);
    s_e[t] = e;
  }
  __syncthreads();
  double qterm = 0.0;
  if (phase != 0) {
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int t = tid + e * NT;
      if (t != 6 * 16 + 6) qterm += 0.5 * s_e[t >> 4] * s_e[t & 15] * g_old[e];
    }
  }
  const double ha = P.huber_a;
  const uint32_t kmask = P.kmask[ks];
  double* stage = s_stage + wave * kStageDoublesPerWave;
  d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
  double cost = 0.0;
  struct ObsD { double u, v, X0, X1, X2; };
  auto widen = [](const ObsRaw& r, ObsD& d) { d.u = (double)r.m.x; d.v = (double)r.m.y; d.X0 = (double)r.X.x; d.X1 = (double)r.X.y; d.X2 = (double)r.X.z; };
  auto pass = [&](int k, const ObsD& r) {
    const bool valid = k < n;
    RigObs o;
    o.a0 = Rca[0] * r.X0 + Rca[1] * r.X1 + Rca[2] * r.X2 + tca[0];
    o.a1 = Rca[3] * r.X0 + Rca[4] * r.X1 + Rca[5] * r.X2 + tca[1];
    o.a2 = Rca[6] * r.X0 + Rca[7] * r.X1 + Rca[8] * r.X2 + tca[2];
    o.iz = recip_depth(o.a2 + tcs[2]);
    o.x = (o.a0 + tcs[0]) * o.iz;
    o.y = (o.a1 + tcs[1]) * o.iz;
    RigKObs ko;
    rigk_obs(kk, o, r.u, r.v, ko);
    double rho, sr;
    huber(ha, ko.ru * ko.ru + ko.rv * ko.rv, rho, sr);
    if (valid) cost += 0.5 * rho;
    if (!valid) sr = 0.0;
    double w[16];
    w[0] = sr * (2.0 * (ko.Bu2 * o.a1 - ko.Bu1 * o.a2)); w[1] = sr * (2.0 * (ko.Bu0 * o.a2 - ko.Bu2 * o.a0)); w[2] = sr * (2.0 * (ko.Bu1 * o.a0 - ko.Bu0 * o.a1));
    w[3] = sr * ko.Bu0; w[4] = sr * ko.Bu1; w[5] = sr * ko.Bu2; w[6] = sr * ko.ru;
#pragma unroll
    for (int q = 0; q < 9; ++q) w[7 + q] = (kmask & (1u << q)) ? 0.0 : sr * ko.ju[q];
    stage_row(stage, lane, w);
    wave_lds_fence();
    gram_rows(stage, lane, acc0, acc1);
    wave_lds_fence();
    w[0] = sr * (2.0 * (ko.Bv2 * o.a1 - ko.Bv1 * o.a2)); w[1] = sr * (2.0 * (ko.Bv0 * o.a2 - ko.Bv2 * o.a0)); w[2] = sr * (2.0 * (ko.Bv1 * o.a0 - ko.Bv0 * o.a1));
    w[3] = sr * ko.Bv0; w[4] = sr * ko.Bv1; w[5] = sr * ko.Bv2; w[6] = sr * ko.rv;
#pragma unroll
    for (int q = 0; q < 9; ++q) w[7 + q] = (kmask & (1u << q)) ? 0.0 : sr * ko.jv[q];
    stage_row(stage, lane, w);
    wave_lds_fence();
    gram_rows(stage, lane, acc0, acc1);
    wave_lds_fence();
  };
  int p = 0;
  for (; p + 1 < npass; p += 2) {
    const int k = p * NT + otid;
    ObsD d;
    widen(oa, d);
    fetch(k + 2 * NT, oa);
    pass(k, d);
    widen(ob, d);
    fetch(k + 3 * NT, ob);
    pass(k + NT, d);
  }
  if (p < npass) {
    ObsD d;
    widen(oa, d);
    pass(p * NT + otid, d);
  }
  if (NW > 1) __syncthreads();   // (every wave done with its staging tile before the partial products overwrite them)
  {
    const int slot = (lane >> 4) * 16 + (lane & 15);
    double* dstp = NW > 1 ? s_stage + wave * 256 : s_G;
#pragma unroll
    for (int r = 0; r < 4; ++r) dstp[slot + 64 * r] = acc0[r] + acc1[r];
  }
  const double qw = wave_sum(qterm), cw = wave_sum(cost);
  if (lane == 0) { s_w[wave] = qw; s_w[4 + wave] = cw; }
  __syncthreads();
  if (NW > 1) {
    s_G[tid] = (s_stage[tid] + s_stage[256 + tid]) + (s_stage[512 + tid] + s_stage[768 + tid]);
    __syncthreads();
  }
  double* out = P.gblocks + ((size_t)dst * P.NG + g) * (size_t)P.gstride;
  const int k0 = lane >> 4, j = lane & 15;
  auto role = [&](int what) {
    if (what < 2) {
      const double mv0 = s_m[k0 * 6 + (j >= 6 && j < 12 ? j - 6 : 0)];
      const double mv1 = s_m[(k0 < 2 ? k0 + 4 : 0) * 6 + (j >= 6 && j < 12 ? j - 6 : 0)];
      const double id = fixed ? 0.0 : 1.0;
      const double n0 = j < 6 ? (j == k0 ? id : 0.0) : (j < 12 ? mv0 : 0.0);                       // N[k0][j]
      const double n1 = k0 < 2 ? (j < 6 ? (j == k0 + 4 ? id : 0.0) : (j < 12 ? mv1 : 0.0))       // N[k0 + 4][j]
                               : (k0 == 2 && j == 12 ? 1.0 : 0.0);
      const int k1 = k0 < 3 ? k0 + 4 : 0;
      d4 B = {0.0, 0.0, 0.0, 0.0};
      if (what == 0) {          // AA = N^T (G7 N)
        const int gi = j < 7 ? j : 0;
        const double gv0 = s_G[gi * 16 + k0], gv1 = s_G[gi * 16 + k1];
        const double a0 = j < 7 ? gv0 : 0.0, a1 = (j < 7 && k0 < 3) ? gv1 : 0.0;
        d4 T = {0.0, 0.0, 0.0, 0.0};
        T = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, n0, T, 0, 0, 0);
        T = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, n1, T, 0, 0, 0);
        B = __builtin_amdgcn_mfma_f64_16x16x4f64(n0, T[0], B, 0, 0, 0);
        B = __builtin_amdgcn_mfma_f64_16x16x4f64(n1, T[1], B, 0, 0, 0);
      } else {                  // AB = N^T G[0:7, 7:16]
        const int hj = j < 9 ? 7 + j : 7;
        const double hv0 = s_G[k0 * 16 + hj], hv1 = s_G[k1 * 16 + hj];
        const double h0 = j < 9 ? hv0 : 0.0, h1 = (j < 9 && k0 < 3) ? hv1 : 0.0;
        B = __builtin_amdgcn_mfma_f64_16x16x4f64(n0, h0, B, 0, 0, 0);
        B = __builtin_amdgcn_mfma_f64_16x16x4f64(n1, h1, B, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = k0 + 4 * r;
        out[what * 256 + row * 16 + j] = B[r];
        if (what == 0 && phase == 0 && row < 6 && row == j) P.ghd0[g * 8 + row] = B[r];   // diag of H_cc
      }
    } else if (what == 2) {     // BB = G[7:16, 7:16]
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = k0 + 4 * r;
        const bool in = row < 9 && j < 9;
        const double gv = s_G[(in ? 7 + row : 0) * 16 + (in ? 7 + j : 0)];
        const double val = in ? gv : 0.0;
        out[512 + row * 16 + j] = val;
        if (phase == 0 && row < 9 && row == j) P.ghdk[g * 16 + row] = val;   // diag of H_kk
      }
    } else {
      if (lane == 0) {
        P.gstats[g * 2] = NW > 1 ? (s_w[4] + s_w[5]) + (s_w[6] + s_w[7]) : s_w[4];
        P.gstats[g * 2 + 1] = NW > 1 ? (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]) : s_w[0];
      }
      if (lane < 36) P.gcomp[((size_t)dst * P.NG + g) * kRigCompK + 256 + lane] = s_m[lane];
    }
  };
  if (NW > 1) {
    role(wave);
  } else {
    role(0); role(1); role(2); role(3);
  }
#pragma unroll
  for (int e = 0; e < EPT; ++e) P.gcomp[((size_t)dst * P.NG + g) * kRigCompK + tid + e * NT] = s_G[tid + e * NT];
}

// ---------------------------------------------------------------------------------------------
// EXTENSION, round 5: the sweep with intrinsics WITHOUT the matrix pipe, and with compact records (k_rig_sweep_k2).
// On gfx950 v_mfma_f64_16x16x4_f64 runs at the rate of v_fma_f64 (32 flop per clock and SIMD) and the two share one datapath
// (profiles/r01/microbench_f64.txt), so the 16 x 16 product of k_rig_sweep_adjk pays 2 x 256 multiply-adds per observation
// for 2 x 136 useful ones, plus a staging round trip through LDS per row set. Here the 16-column Gram of
// X = [J_cam(6) r J_k(9)] is accumulated by plain FMAs on its lower triangle, skipping the structural zeros of the pixel
// model (d u / d (fy, py) = d v / d (fx, px) = 0: 105 products per row instead of 256): 210 FMAs per observation.
// 132 accumulators do not fit one lane's 256 registers next to the projection, so a group is swept by TWO waves that split
// the pairs (kK2 below, 66 each) and show each other their rows through LDS: per pass each wave evaluates 64 observations,
// leaves the 28 non-zero row entries of each in LDS, and accumulates ITS pairs over both waves' 128 observations.
// What leaves the group is ONE record of 256 doubles (P.gcomp) -- everything the elimination reads of a group:
//   [0..134]   the direct sums in dmap order: G_cc (21) g_c (6) H_ck (54) H_kk (45) g_k (9)      [135] r^2
//   [136..171] T = G_cc M (camera columns of the frame's coupling)   [172..225] H_fk = M^T H_ck (its intrinsics columns)
//   [226..246] the group's share M^T G_cc M of the frame block       [247..252] its share M^T g_c of the frame gradient
// instead of three 16 x 16 tiles and a 320-double compact record (8.5 KB per group and buffer -> 2 KB). The model-cost term of
// a step is a weighted sum of the OLD record's entries (weights: products of the step's components, k2_qcoef).
// ---------------------------------------------------------------------------------------------
constexpr int kRigRecK = 256;
constexpr int kRkR2 = 135, kRkT = 136, kRkFK = 172, kRkHff = 226, kRkGf = 247, kRkEnd = 253;
constexpr bool k2_in_u(int c) { return c != 8 && c != 10; }   // columns with a non-zero entry in the u row / the v row
constexpr bool k2_in_v(int c) { return c != 7 && c != 9; }
struct K2Split {
  signed char owner[136];   // wave that accumulates pair p = tri(i, j); -1: structurally zero
  unsigned char slot[136];  // its accumulator: 0..63 summed by the butterfly (the sum ends in lane `slot`), 64.. by wave_sum
  unsigned char inv[2][64]; // pair of butterfly slot s of wave w (255: none)
  unsigned char dir[136];   // direct entry e (dmap order, [135] = r^2) -> pair
  unsigned char extra[2][4];   // pair of accumulator 64 + x of wave w (255: none)
  int n[2];
};
constexpr K2Split k2_make_split() {
  K2Split s{};
  int cost[2] = {0, 0};
  s.n[0] = s.n[1] = 0;
  for (int w = 0; w < 2; ++w) for (int k = 0; k < 64; ++k) s.inv[w][k] = 255;
  for (int w = 0; w < 2; ++w) for (int k = 0; k < 4; ++k) s.extra[w][k] = 255;
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j <= i; ++j) {
      const int p = i * (i + 1) / 2 + j;
      const int c = ((k2_in_u(i) && k2_in_u(j)) ? 1 : 0) + ((k2_in_v(i) && k2_in_v(j)) ? 1 : 0);
      if (c == 0) { s.owner[p] = -1; s.slot[p] = 0; continue; }
      const int w = cost[0] <= cost[1] ? 0 : 1;
      s.owner[p] = (signed char)w;
      s.slot[p] = (unsigned char)s.n[w];
      if (s.n[w] < 64) s.inv[w][s.n[w]] = (unsigned char)p; else s.extra[w][s.n[w] - 64] = (unsigned char)p;
      s.n[w]++;
      cost[w] += c;
    }
  // direct entries: columns 0..5 camera, 6 residual, 7..15 intrinsics
  int e = 0;
  for (int i = 0; i < 6; ++i) for (int j = 0; j <= i; ++j) s.dir[e++] = (unsigned char)(i * (i + 1) / 2 + j);
  for (int i = 0; i < 6; ++i) s.dir[e++] = (unsigned char)(6 * 7 / 2 + i);
  for (int i = 0; i < 6; ++i) for (int j = 0; j < 9; ++j) s.dir[e++] = (unsigned char)((7 + j) * (8 + j) / 2 + i);
  for (int i = 0; i < 9; ++i) for (int j = 0; j <= i; ++j) s.dir[e++] = (unsigned char)((7 + i) * (8 + i) / 2 + 7 + j);
  for (int j = 0; j < 9; ++j) s.dir[e++] = (unsigned char)((7 + j) * (8 + j) / 2 + 6);
  s.dir[e++] = (unsigned char)(6 * 7 / 2 + 6);
  return s;
}
constexpr K2Split kK2 = k2_make_split();
static_assert(kK2.n[0] <= 68 && kK2.n[1] <= 68 && kK2.n[0] + kK2.n[1] == 132, "pairs per wave");
constexpr int kK2Acc = 68;

// the products of ONE row (ROW 0: u, 1: v) that wave W accumulates
template <int W, int ROW>
__device__ __forceinline__ void k2_accumulate(const double* w, double* acc) {
#pragma unroll
  for (int i = 0; i < 16; ++i) {
#pragma unroll
    for (int j = 0; j <= i; ++j) {
      const int p = i * (i + 1) / 2 + j;
      if (kK2.owner[p] == W) {
        const int s = kK2.slot[p];
        if (ROW == 0 ? (k2_in_u(i) && k2_in_u(j)) : (k2_in_v(i) && k2_in_v(j))) acc[s] = fma(w[i], w[j], acc[s]);
      }
    }
  }
}
// 64 per-lane values -> their 64-lane sums, value e in lane e (reduce_scatter32 with one more halving in front and the last
// step a halving too)
__device__ __forceinline__ void reduce_scatter64(double* p, int lane) {
  reduce_swap32<32>(p);
  reduce_swap16<16>(p);
  reduce_dpp<8, 0x128, 8>(p, lane);   // row_ror:8
  reduce_dpp<4, 0x141, 4>(p, lane);   // row_half_mirror
  reduce_dpp<2, 0x4E, 2>(p, lane);    // quad_perm:[2,3,0,1]
  reduce_dpp<1, 0xB1, 1>(p, lane);    // quad_perm:[1,0,3,2]
}
// Weight of entry e of a group's OLD record in the model-cost term q = g^T d + 1/2 d^T H d of the step d = (dc, df, dk): a product
// of (at most) two step components and 1/2 or 1. The steps lie where the records lie in LDS (sm: camera [12..17], frame
// [44..49], intrinsics [80..88]); WHICH two, per entry, is a compile-time table (one word per entry: index of the first factor
// | index of the second << 8 | flags << 16; index 255 = the constant one; flag 1 / 2: the first / second factor is a camera
// component and vanishes for a camera held constant; flag 4: weight 1/2; flag 8: weight 0). Computed by index arithmetic per
// thread and group (division by 6 and 9, two triangular-index searches) it was 2.5 k cycles of a group's 27 k.
constexpr unsigned k2_qdesc(int e) {
  int ka = 0, ia = 0, kb = 0, ib = 0, half = 0, zero = 0;
  auto tri_i = [](int idx) { int i = 0; while ((i + 1) * (i + 2) / 2 <= idx) ++i; return i; };
  if (e < 21) { const int i = tri_i(e), j = e - i * (i + 1) / 2; ka = 1; ia = i; kb = 1; ib = j; half = i == j; }
  else if (e < 27) { ka = 1; ia = e - 21; }
  else if (e < 81) { const int i = (e - 27) / 9, j = (e - 27) - 9 * i; ka = 1; ia = i; kb = 3; ib = j; }
  else if (e < 126) { const int i = tri_i(e - 81), j = (e - 81) - i * (i + 1) / 2; ka = 3; ia = i; kb = 3; ib = j; half = i == j; }
  else if (e < 135) { ka = 3; ia = e - 126; }
  else if (e < kRkT) { zero = 1; }
  else if (e < kRkFK) { const int i = (e - kRkT) / 6, j = (e - kRkT) - 6 * i; ka = 1; ia = i; kb = 2; ib = j; }
  else if (e < kRkHff) { const int i = (e - kRkFK) / 9, j = (e - kRkFK) - 9 * i; ka = 2; ia = i; kb = 3; ib = j; }
  else if (e < kRkGf) { const int i = tri_i(e - kRkHff), j = (e - kRkHff) - i * (i + 1) / 2; ka = 2; ia = i; kb = 2; ib = j; half = i == j; }
  else if (e < kRkEnd) { ka = 2; ia = e - kRkGf; }
  else zero = 1;
  const unsigned xa = ka == 0 ? 255u : (unsigned)((ka == 1 ? 12 : (ka == 2 ? 44 : 80)) + ia);
  const unsigned xb = kb == 0 ? 255u : (unsigned)((kb == 1 ? 12 : (kb == 2 ? 44 : 80)) + ib);
  return xa | (xb << 8) | ((unsigned)((ka == 1 ? 1 : 0) | (kb == 1 ? 2 : 0) | (half ? 4 : 0) | (zero ? 8 : 0)) << 16);
}
// The tables a LANE indexes at run time, as one array of words that a workgroup copies into LDS once, under its first round
// trip (read from device memory where they are needed -- behind the lane sums, in the record assembly -- each was a memory
// round trip on the tail's chain): [0..255] k2_qdesc, [256..287] inv (bytes), [288..321] dir (bytes), [322..323] extra (bytes).
constexpr int kK2TabWords = 324;
struct K2Tab { unsigned w[kK2TabWords]; };
constexpr K2Tab k2_make_tab() {
  K2Tab t{};
  for (int e = 0; e < 256; ++e) t.w[e] = k2_qdesc(e);
  const K2Split s = k2_make_split();
  for (int i = 0; i < 128; ++i) t.w[256 + i / 4] |= (unsigned)s.inv[i / 64][i % 64] << (8 * (i % 4));
  for (int i = 0; i < 136; ++i) t.w[288 + i / 4] |= (unsigned)s.dir[i] << (8 * (i % 4));
  for (int i = 0; i < 8; ++i) t.w[322 + i / 4] |= (unsigned)s.extra[i / 4][i % 4] << (8 * (i % 4));
  return t;
}
__device__ const K2Tab kK2Tab = k2_make_tab();
__device__ __forceinline__ int k2_tab_byte(const unsigned* tab, int word0, int i) { return (int)((tab[word0 + (i >> 2)] >> (8 * (i & 3))) & 255u); }
__device__ __forceinline__ double k2_qcoef(const unsigned* tab, int e, const double* sm, double cam_on) {
  const unsigned d = tab[e];
  const int xa = (int)(d & 255u), xb = (int)((d >> 8) & 255u);
  const unsigned fl = d >> 16;
  double fa = sm[xa == 255 ? 0 : xa], fb = sm[xb == 255 ? 0 : xb];
  fa = xa == 255 ? 1.0 : ((fl & 1u) ? fa * cam_on : fa);
  fb = xb == 255 ? 1.0 : ((fl & 2u) ? fb * cam_on : fb);
  const double w = (fl & 8u) ? 0.0 : ((fl & 4u) ? 0.5 : 1.0);
  return w * fa * fb;
}

constexpr int kRigK2Waves = 2;     // waves per SIMD the kernel is compiled for (256 registers)
// ONE workgroup (two waves) per group. What a group needs before its first pass -- its indices, then its records, old record and
// first observations -- are two dependent round trips; a persistent form (four workgroups per compute unit looping over groups, the
// next group's loads under the current one's tail) was built in round 5 and measured SLOWER (241 against 201 us at 8 x 2000 x 500:
// what a group costs beyond its passes is instructions -- lane sums, assembly -- not latency, and the other workgroups of a compute
// unit already cover the latency); removed in round 6.
struct K2Group {   // what is known about a group before its sweep starts
  int f, c, ks, n, fixed;
  int64_t s0;
  uint32_t kmask;
};
__global__ __launch_bounds__(128, kRigK2Waves) void k_rig_sweep_k2(RigDev P) {
  __shared__ __attribute__((aligned(16))) d2 s_rows[2 * 14 * 64];   // [wave][q][lane]: u row entries (q < 7), v row entries
  __shared__ double sm[96];        // camera record, frame record, intrinsics record (candidate [0..8], step [16..24])
  __shared__ double s_G[144];      // lower triangle of G by pair, [136] cost of wave 0, [137] of wave 1, [138..139] model-cost sums
  __shared__ double s_m[36];       // M
  __shared__ double s_T[128];      // T (36) | H_fk (54) | g_f share (6) | H_ff share (21)
  __shared__ unsigned s_tab[kK2TabWords];   // the lane-indexed tables (kK2Tab), staged once
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const LmCtl* ctl = P.ctl;
  const int done = ctl->done, phase = ctl->phase, step_valid = ctl->step_valid, cur = ctl->cur;
  if (done) return;
  if (phase != 0 && !step_valid) return;
  const int dst = phase == 0 ? cur : (cur ^ 1);
  {   // (the first group's loop-top barrier is in front of every read of the tables)
    const unsigned t0 = kK2Tab.w[tid], t1 = kK2Tab.w[tid + 128], t2 = kK2Tab.w[tid + 256 < kK2TabWords ? tid + 256 : 0];
    s_tab[tid] = t0; s_tab[tid + 128] = t1;
    if (tid + 256 < kK2TabWords) s_tab[tid + 256] = t2;
  }
  const int64_t NG = P.NG;
  struct F3 { float x, y, z; };
  struct ObsRaw { float2 m; F3 X; };
  struct ObsD { double u, v, X0, X1, X2; };
  auto widen = [](const ObsRaw& r, ObsD& d) { d.u = (double)r.m.x; d.v = (double)r.m.y; d.X0 = (double)r.X.x; d.X1 = (double)r.X.y; d.X2 = (double)r.X.z; };
  auto group_indices = [&](int64_t g, K2Group& q) {   // ONE round trip: the group's record (then the mask of its intrinsics set)
    const int64_t gc = g < NG ? g : 0;
    const int4 a = P.gk2[2 * gc], b = P.gk2[2 * gc + 1];
    q.s0 = (int64_t)(((unsigned long long)(unsigned)a.y << 32) | (unsigned)a.x);
    q.n = a.z; q.f = a.w;
    q.c = b.x; q.ks = b.y; q.fixed = b.z;
    q.kmask = P.kmask[q.ks];
  };
  // vector loads of a group: the lane's first observation, one value of the three records, two of the group's old record
  auto group_loads = [&](int64_t g, const K2Group& q, ObsRaw& o0, double& recv, double& old0, double& old1) {
    const float2* uvg = reinterpret_cast<const float2*>(P.uv) + q.s0;
    const F3* xg = reinterpret_cast<const F3*>(P.oxyz) + q.s0;
    const int kc = tid < q.n ? tid : 0;
    o0.m = uvg[kc];
    o0.X = xg[kc];
    recv = tid < 32 ? P.camrec[q.c * 32 + tid] : (tid < 64 ? P.frec[(size_t)q.f * 32 + (tid - 32)] : P.krec[q.ks * 32 + ((tid - 64) & 31)]);
    const double* rec_old = P.gcomp + ((size_t)cur * NG + (g < NG ? g : 0)) * kRigRecK;
    old0 = rec_old[tid];
    old1 = rec_old[tid + 128];
  };
  const double ha = P.huber_a;
  // ONE branch on the wave for the whole sweep (each arm its own register allocation: with a branch per row the arms
  // met four times a pass, ~90 register moves each to reconcile them)
  auto sweep = [&](auto wtag) {
  constexpr int W = decltype(wtag)::value;
  const int64_t g = blockIdx.x;
  if (g >= NG) return;   // (uniform; the launch has one workgroup per group)
  K2Group q;
  group_indices(g, q);
  ObsRaw oa;
  double recv, old0, old1;
  group_loads(g, q, oa, recv, old0, old1);
  d2* mine = s_rows + (size_t)W * 14 * 64 + lane;
  const d2* theirs = s_rows + (size_t)(W ^ 1) * 14 * 64 + lane;
  {
    // ---- the group's records into LDS
    // (every per-lane index of the group's head and tail comes from a LAUNDERED copy of the thread id: derived from the original
    // they are kept alive -- spilled -- across the passes)
    int tid_h = threadIdx.x;
    asm volatile("" : "+v"(tid_h));
    if (tid_h < 96) sm[tid_h] = recv;
    lds_barrier();
    const int n = q.n, npass = (n + 127) >> 7;      // (the same for both waves: they meet at two barriers per pass)
    const bool fixed = q.fixed != 0;
    const uint32_t kmask = q.kmask;
    const float2* uvg = reinterpret_cast<const float2*>(P.uv) + q.s0;
    const F3* xg = reinterpret_cast<const F3*>(P.oxyz) + q.s0;
    auto fetch = [&](int k, ObsRaw& r) {
      const int kc = k < n ? k : 0;
      r.m = uvg[kc];
      r.X = xg[kc];
    };
    // the chain of both poses as one (k_rig_sweep_adj), from the records in LDS (uniform addresses), values in scalar registers
    double Rca[9], tca[3], tcs[3], kk[9];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
      for (int j = 0; j < 3; ++j) Rca[3 * i + j] = rfl(sm[3 * i] * sm[32 + j] + sm[3 * i + 1] * sm[32 + 3 + j] + sm[3 * i + 2] * sm[32 + 6 + j]);
      tca[i] = rfl(sm[3 * i] * sm[32 + 9] + sm[3 * i + 1] * sm[32 + 10] + sm[3 * i + 2] * sm[32 + 11]);
      tcs[i] = rfl(sm[9 + i]);
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) kk[i] = rfl(sm[64 + i]);
    if (tid_h < 48) {   // M[a][b], a = tid >> 3, b = tid & 7 < 6 (see k_rig_sweep_adj)
      const int a = tid_h >> 3, b = tid_h & 7;
      const int a3 = a < 3 ? a : a - 3, b3 = b < 3 ? b : b - 3;
      const int b1 = b3 == 2 ? 0 : b3 + 1, b2 = b3 == 0 ? 2 : b3 - 1;
      const double diag = sm[3 * a3 + (b3 < 3 ? b3 : 0)];
      const double cross = 2.0 * (sm[3 * a3 + b1] * sm[32 + 9 + b2] - sm[3 * a3 + b2] * sm[32 + 9 + b1]);
      const double v = (a < 3) == (b < 3) ? diag : (a >= 3 ? cross : 0.0);
      if (b < 6) s_m[a * 6 + b] = v;
    }
    // model-cost term of the step at the accepted point, from the group's old record (two entries per thread)
    double qterm = 0.0;
    if (phase != 0) qterm = k2_qcoef(s_tab, tid_h, sm, fixed ? 0.0 : 1.0) * old0 + k2_qcoef(s_tab, tid_h + 128, sm, fixed ? 0.0 : 1.0) * old1;
    double acc[kK2Acc];
#pragma unroll
    for (int e = 0; e < kK2Acc; ++e) acc[e] = 0.0;
    double cost = 0.0;
    auto pass = [&](int k, const ObsD& r) {
      const bool valid = k < n;
      RigObs o;
      o.a0 = Rca[0] * r.X0 + Rca[1] * r.X1 + Rca[2] * r.X2 + tca[0];
      o.a1 = Rca[3] * r.X0 + Rca[4] * r.X1 + Rca[5] * r.X2 + tca[1];
      o.a2 = Rca[6] * r.X0 + Rca[7] * r.X1 + Rca[8] * r.X2 + tca[2];
      o.iz = recip_depth(o.a2 + tcs[2]);
      o.x = (o.a0 + tcs[0]) * o.iz;
      o.y = (o.a1 + tcs[1]) * o.iz;
      // pixel model (rigk_obs, DistortNormalized / DistortPixels of calibrator.cpp:70-95) with the Huber weight folded into
      // the factors every row entry carries anyway (sr fx, sr fy, sr fx / z, sr fy / z): an entry costs one instruction
      const double x = o.x, y = o.y;
      const double x2 = x * x, y2 = y * y, xy = x * y;
      const double r2 = x2 + y2, r4 = r2 * r2, r6 = r4 * r2;
      const double m = 1.0 + kk[4] * r2 + kk[5] * r4 + kk[8] * r6;
      const double ax = r2 + 2.0 * x2, ay = r2 + 2.0 * y2;
      const double xd = x * m + 2.0 * kk[6] * xy + kk[7] * ax;
      const double yd = y * m + 2.0 * kk[7] * xy + kk[6] * ay;
      const double ru = kk[0] * xd + kk[2] - r.u, rv = kk[1] * yd + kk[3] - r.v;
      const double mp = kk[4] + 2.0 * kk[5] * r2 + 3.0 * kk[8] * r4;
      const double dxx = m + 2.0 * mp * x2 + 2.0 * kk[6] * y + 6.0 * kk[7] * x;
      const double dxy = 2.0 * mp * xy + 2.0 * kk[6] * x + 2.0 * kk[7] * y;
      const double dyy = m + 2.0 * mp * y2 + 2.0 * kk[7] * x + 6.0 * kk[6] * y;
      double rho, sr;
      huber(ha, ru * ru + rv * rv, rho, sr);
      if (valid) cost += 0.5 * rho;
      if (!valid) sr = 0.0;
      const double sfx = sr * kk[0], sfy = sr * kk[1];
      const double a0d = o.a0 + o.a0, a1d = o.a1 + o.a1, a2d = o.a2 + o.a2;   // (2 a: the rotation columns are 2 a x B)
      // Row by row, each used up at once: formed, left in LDS for the other wave (its 14 non-zero entries: u skips columns 8 and
      // 10, v columns 7 and 9), accumulated -- so that only one row of 14 is alive next to the accumulators at any time.
      {
        double w[16];
        const double gz = sfx * o.iz;
        const double B0 = gz * dxx, B1 = gz * dxy, B2 = -(B0 * x + B1 * y);
        w[0] = B2 * a1d - B1 * a2d; w[1] = B0 * a2d - B2 * a0d; w[2] = B1 * a0d - B0 * a1d;
        w[3] = B0; w[4] = B1; w[5] = B2; w[6] = sr * ru;
        const double tx = sfx * x;
        w[7] = sr * xd; w[8] = 0.0; w[9] = sr; w[10] = 0.0;
        w[11] = tx * r2; w[12] = tx * r4; w[13] = tx * (y + y); w[14] = sfx * ax; w[15] = tx * r6;
        if (kmask != 0u) {   // (uniform: intrinsics held constant have no column)
#pragma unroll
          for (int qq = 0; qq < 9; ++qq) w[7 + qq] = (kmask & (1u << qq)) ? 0.0 : w[7 + qq];
        }
        mine[0 * 64] = d2{w[0], w[1]}; mine[1 * 64] = d2{w[2], w[3]}; mine[2 * 64] = d2{w[4], w[5]}; mine[3 * 64] = d2{w[6], w[7]};
        mine[4 * 64] = d2{w[9], w[11]}; mine[5 * 64] = d2{w[12], w[13]}; mine[6 * 64] = d2{w[14], w[15]};
        k2_accumulate<W, 0>(w, acc);
      }
      {
        double w[16];
        const double gz = sfy * o.iz;
        const double B0 = gz * dxy, B1 = gz * dyy, B2 = -(B0 * x + B1 * y);
        w[0] = B2 * a1d - B1 * a2d; w[1] = B0 * a2d - B2 * a0d; w[2] = B1 * a0d - B0 * a1d;
        w[3] = B0; w[4] = B1; w[5] = B2; w[6] = sr * rv;
        const double ty = sfy * y;
        w[7] = 0.0; w[8] = sr * yd; w[9] = 0.0; w[10] = sr;
        w[11] = ty * r2; w[12] = ty * r4; w[13] = sfy * ay; w[14] = ty * (x + x); w[15] = ty * r6;
        if (kmask != 0u) {
#pragma unroll
          for (int qq = 0; qq < 9; ++qq) w[7 + qq] = (kmask & (1u << qq)) ? 0.0 : w[7 + qq];
        }
        mine[7 * 64] = d2{w[0], w[1]}; mine[8 * 64] = d2{w[2], w[3]}; mine[9 * 64] = d2{w[4], w[5]}; mine[10 * 64] = d2{w[6], w[8]};
        mine[11 * 64] = d2{w[10], w[11]}; mine[12 * 64] = d2{w[12], w[13]}; mine[13 * 64] = d2{w[14], w[15]};
        k2_accumulate<W, 1>(w, acc);
      }
      lds_barrier();
      {
        double w[16];
        d2 t;
        t = theirs[0 * 64]; w[0] = t.x; w[1] = t.y;  t = theirs[1 * 64]; w[2] = t.x; w[3] = t.y;
        t = theirs[2 * 64]; w[4] = t.x; w[5] = t.y;  t = theirs[3 * 64]; w[6] = t.x; w[7] = t.y;
        t = theirs[4 * 64]; w[9] = t.x; w[11] = t.y; t = theirs[5 * 64]; w[12] = t.x; w[13] = t.y;
        t = theirs[6 * 64]; w[14] = t.x; w[15] = t.y;
        w[8] = 0.0; w[10] = 0.0;
        k2_accumulate<W, 0>(w, acc);
      }
      {
        double w[16];
        d2 t;
        t = theirs[7 * 64]; w[0] = t.x; w[1] = t.y;  t = theirs[8 * 64]; w[2] = t.x; w[3] = t.y;
        t = theirs[9 * 64]; w[4] = t.x; w[5] = t.y;  t = theirs[10 * 64]; w[6] = t.x; w[8] = t.y;
        t = theirs[11 * 64]; w[10] = t.x; w[11] = t.y; t = theirs[12 * 64]; w[12] = t.x; w[13] = t.y;
        t = theirs[13 * 64]; w[14] = t.x; w[15] = t.y;
        w[7] = 0.0; w[9] = 0.0;
        k2_accumulate<W, 1>(w, acc);
      }
      lds_barrier();   // (both waves have read: the rows may be overwritten by the next pass)
    };
    // one register set, observations one pass ahead (two sets -- the pair of passes unrolled -- spill: 607 us against 209 at 8 x 2000 x 500)
    // (measured and dropped: one word of the observations two passes ahead, loaded and thrown away so that the real prefetch finds
    // its lines in L2 -- 210.7 us against 207.9 at 8 x 2000 x 500: the ~0.7 k cycles a pass waits for its observations are not
    // cache misses of the prefetch)
    for (int p = 0; p < npass; ++p) {
      const int k = p * 128 + tid;
      ObsD d;
      widen(oa, d);
      fetch(k + 128, oa);
      pass(k, d);
    }
    // ---- sums over the lanes: 64 accumulators by the butterfly (value s ends in lane s), the others and the scalars by wave_sum
    int tid_t = threadIdx.x;
    asm volatile("" : "+v"(tid_t));
    const int lane_t = tid_t & 63;
    reduce_scatter64(acc, lane_t);
    {
      const int pr = k2_tab_byte(s_tab, 256, W * 64 + lane_t);
      if (pr != 255) s_G[pr] = acc[0];
    }
    {
      // the (at most four) accumulators beyond the butterfly, the cost and the model-cost term: eight values through three halving
      // steps and three plain ones (value e in lanes with bits 5, 4, 3 = e), instead of six full wave sums
      double v8[8] = {acc[64], acc[65], acc[66], acc[67], cost, qterm, 0.0, 0.0};
      reduce_swap32<4>(v8);
      reduce_swap16<2>(v8);
      reduce_dpp<1, 0x128, 8>(v8, lane_t);   // row_ror:8
      double t = v8[0];
      t += dpp_f64<0x141>(t);                // row_half_mirror (partner l ^ 7: stays inside the eight lanes that share bits 5, 4, 3)
      t += dpp_f64<0x4E>(t);                 // quad_perm:[2,3,0,1]
      t += dpp_f64<0xB1>(t);                 // quad_perm:[1,0,3,2]
      const int e8 = ((lane_t >> 5) & 1) * 4 + ((lane_t >> 4) & 1) * 2 + ((lane_t >> 3) & 1);
      if ((lane_t & 7) == 0) {
        if (e8 < 4) { const int pr = k2_tab_byte(s_tab, 322, W * 4 + e8); if (pr != 255) s_G[pr] = t; }
        else if (e8 == 4) s_G[136 + W] = t;
        else if (e8 == 5) s_G[138 + W] = t;
      }
    }
    if (tid_t < 4) s_G[tid_t == 0 ? 43 : (tid_t == 1 ? 62 : (tid_t == 2 ? 53 : 64))] = 0.0;   // the structurally zero pairs (fy, fx) (py, fx) (px, fy) (py, px)
    lds_barrier();
    auto G = [&](int i, int j) { const int hi = i > j ? i : j, lo = i > j ? j : i; return s_G[hi * (hi + 1) / 2 + lo]; };
    // ---- the frame's couplings through the group's adjoint: T = G_cc M, H_fk = M^T H_ck, g_f = M^T g_c
    if (tid_t < 96) {
      double t = 0.0;
      if (tid_t < 36) {
        const int r = tid_t / 6, l = tid_t - 6 * r;
#pragma unroll
        for (int k = 0; k < 6; ++k) t = fma(G(r, k), s_m[k * 6 + l], t);
      } else if (tid_t < 90) {
        const int a = (tid_t - 36) / 9, j = (tid_t - 36) - 9 * a;
#pragma unroll
        for (int k = 0; k < 6; ++k) t = fma(s_m[k * 6 + a], G(k, 7 + j), t);
      } else {
        const int a = tid_t - 90;
#pragma unroll
        for (int k = 0; k < 6; ++k) t = fma(s_m[k * 6 + a], G(k, 6), t);
      }
      s_T[tid_t] = t;
    } else if (tid_t < 117) {   // share of the frame block, (M^T G_cc M)[i][j], i >= j -- in the same phase, straight from G (no T: no second barrier)
      const int e = tid_t - 96;
      int i = 0;
      while ((i + 1) * (i + 2) / 2 <= e) ++i;
      const int j = e - i * (i + 1) / 2;
      double t = 0.0;
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        double u = 0.0;
#pragma unroll
        for (int l = 0; l < 6; ++l) u = fma(G(k, l), s_m[l * 6 + j], u);
        t = fma(s_m[k * 6 + i], u, t);
      }
      s_T[tid_t] = t;
    }
    lds_barrier();
    double* rec = P.gcomp + ((size_t)dst * NG + g) * kRigRecK;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int e = tid_t + 128 * h;
      double v = 0.0;
      if (e <= kRkR2) v = s_G[k2_tab_byte(s_tab, 288, e)];
      else if (e < kRkHff) v = s_T[e - kRkT];
      else if (e < kRkGf) v = s_T[96 + (e - kRkHff)];
      else if (e < kRkEnd) v = s_T[90 + (e - kRkGf)];
      rec[e] = v;
    }
    if (phase == 0) {
      if (tid_t < 6) P.ghd0[g * 8 + tid_t] = fixed ? 0.0 : G(tid_t, tid_t);
      else if (tid_t >= 64 && tid_t < 73) P.ghdk[g * 16 + (tid_t - 64)] = G(7 + (tid_t - 64), 7 + (tid_t - 64));
    }
    if (tid_t == 0) {
      P.gstats[g * 2] = s_G[136] + s_G[137];
      P.gstats[g * 2 + 1] = s_G[138] + s_G[139];
    }
  }
  };
  if (wave == 0) sweep(std::integral_constant<int, 0>{}); else sweep(std::integral_constant<int, 1>{});
}

