// cc_intrinsics.hip -- single-camera intrinsics bundle adjustment on MI355X (gfx950).
//
// Replaces the ceres::Problem/ceres::Solve block of Calibrator::Optimize
// (/root/reference/src/calibrator.cpp:236-324). Two kernels per LM iteration (single GPU / mailbox exchange;
// the RCCL route keeps the solve step in a kernel of its own around the all-reduce):
//
//   decide+elim+solve : reduce the per-frame statistics, Ceres trust-region logic; per frame, damped 6x6
//            pose block -> Cholesky, Z = L^-1 [H_ps | g_p]; partial sums of the reduced (Schur) 9x9
//            system over frames; the LAST block to finish reduces the partials, adds the LM diagonal and
//            solves the 9x9 system -> scaled shared step, gradient / radius tests, control block.
//   sweep  : one workgroup per frame. Prologue back-substitutes the frame's pose step, forms the
//            candidate point (QuaternionManifold::Plus) and the frame's model-cost term; the main
//            loop evaluates pinhole + radial-tangential projection and the analytic 2x15 Jacobian
//            per observation (one lane = one observation), stages [J r] rows through LDS and
//            contracts them with v_mfma_f64_16x16x4_f64 into the frame's 16x16 Gram block
//            G_f = sum_rows [J r]^T [J r]  (= H_ss,f H_sp,f g_s,f / H_pp,f g_p,f / 2 cost_f).
//
// HBM layout (per handle): uv float2[N], xyz float[3N] exactly as the API hands them over
// (20 B / observation, read once per sweep, coalesced: lane i of a wave reads observation
// base+i); poses double[2][F][8]; Gram blocks double[2][F][256] (2 KiB per frame, double
// buffered: accepted point / candidate); per-frame stats double[F][32]; Z,L double[F][96].
#include "cc_common.hpp"
#include "cc_device.hpp"
#include "cc_intrinsics_dev.hpp"
#include "cc_intrinsics_persist.hpp"


#include <algorithm>
#include <chrono>
#include <climits>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace cc {

// ---------------------------------------------------------------------------------------------
// sweep: one workgroup (4 waves) per frame -- or, when the frames alone cannot fill the chip (a shard of a
// strong-scaled problem: 125 frames per GPU at eight GPUs), T workgroups per frame, each sweeping a contiguous
// tile of the frame's observations into a partial Gram block of its own (the elimination adds the T tiles).
// Every tile repeats the (cheap) pose update of its frame; tile 0 alone publishes it and owns the model-cost term.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kSweepThreads, 4) void k_intr_sweep(IntrDev P, int flags) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* s_stage = reinterpret_cast<double*>(smem_raw);       // [4][1024]
  double* s_blk = s_stage;                                      // [2048] cross-wave reduce + block copy (after the loop)
  double* sm = s_stage + 4 * kStageDoublesPerWave;              // [256] prologue scratch
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // (timing-only builds, scripts/time_intr_decide.py: the middle workgroup leaves wall-clock marks in vec_solve[48..])
  const int T = P.T;
  const int64_t f = (int64_t)blockIdx.x / T;
  const int tile = (int)(blockIdx.x - f * T);
  // ---- one global round trip gathers every input of the prologue; nothing in it depends on the
  // control block (both buffers of the ping-pong state are fetched, the right one is picked later)
  // sm[0..59] Y, [60..66] pose buf 0, [67..73] pose buf 1, [74..82] intr buf 0, [83..91] intr buf 1,
  // [92..100] ds (scaled), [101..109] ss, [110..115] sp
  // sm[120..134] unscaled step (9 shared, 6 pose); sm[136..144] R; [145..147] t; [148..156] intr_cand
  // sm[158] step^2 (pose part), sm[159] |x_cand|^2 (pose part); sm[160..169] pose/intr of `cur`
  // flags: bit 0 = part of a solve (mailbox epochs advance), bit 2 = fetch only the current Gram buffer, bit 1 = RESTART: first launch of a solve that starts
  // from the state of the last set_state. The restart sweep ignores the stale control block (it behaves as if
  // it were zero: initial evaluation into buffer 0), takes poses and intrinsics from the initial-state arrays and
  // restores buffer 0 from them on the way, so a restart costs no kernel of its own.
  const bool in_solve = (flags & 1) != 0, restart = (flags & 2) != 0;
  const LmCtl* ctl = P.ctl;
  // (read unconditionally, then select: loads behind `restart ? 0 :` would each wait on their own)
  const int c_done = ctl->done, c_phase = ctl->phase, c_valid = ctl->step_valid, c_cur = ctl->cur;
  const int done = restart ? 0 : c_done, phase = restart ? 0 : c_phase, step_valid = restart ? 0 : c_valid,
            cur = restart ? 0 : c_cur;
  int64_t s0 = P.off[f], s1 = P.off[f + 1];
  if (T > 1) {   // this workgroup's tile of the frame
    const int64_t len = (s1 - s0 + T - 1) / T;
    s0 = s0 + tile * len < s1 ? s0 + tile * len : s1;
    s1 = s0 + len < s1 ? s0 + len : s1;
  }
  // first pass of observations: issued FIRST (needs only the frame's offsets), consumed after the prologue. It used to
  // sit behind the gather below, whose accumulation loop over the tiles made the compiler wait for the old Gram block
  // before issuing anything else: a second dependent round trip in front of every workgroup's main loop.
  const float2* uv2 = reinterpret_cast<const float2*>(P.uv);
  // passes of THIS wave: a wave whose 64 slots of a pass all lie beyond the frame skips that pass
  // (wave-uniform; the main loop holds no workgroup barrier)
  const int64_t wrem = s1 - s0 - wave * 64;
  const int npass = wrem > 0 ? (int)((wrem + kSweepThreads - 1) / kSweepThreads) : 0;
  // UNCONDITIONAL loads (idle slots re-read a valid observation; the arena holds one slot even when N = 0): inside an
  // `if` the loaded registers are merged with the defaults at the end of the region, and that merge waits for the
  // load -- the "prefetch" then stalls the wave for a full memory round trip where it is issued.
  const int64_t safe0 = s0 < P.N ? s0 : 0;
  float2 nm;
  float nX0, nX1, nX2;
  {
    const int64_t idx = s0 + tid;
    const int64_t ic = idx < s1 ? idx : safe0;
    nm = uv2[ic];
    nX0 = P.xyz[ic * 3]; nX1 = P.xyz[ic * 3 + 1]; nX2 = P.xyz[ic * 3 + 2];
  }
  // gather: one load per thread from a selected address; it is stored to LDS only after the loads of the old Gram
  // block below have been issued (the store needs the data: placed here it made them a round trip of their own)
  double gv;
  {
    const double* src;
    if (tid < 60) src = P.Y + f * kYStride + tid;
    else if (tid < 67) src = (restart ? P.init_pose : P.pose) + (size_t)f * 8 + (tid - 60);
    else if (tid < 74) src = P.pose + ((size_t)P.F + f) * 8 + (tid - 67);
    else if (tid < 83) src = (restart ? P.init_intr : P.intr) + (tid - 74);
    else if (tid < 92) src = P.intr + 16 + (tid - 83);
    else if (tid < 101) src = P.ds + (tid - 92);
    else if (tid < 110) src = P.ss + (tid - 101);
    else if (tid < 116) src = P.sp + f * 8 + (tid - 110);
    else src = P.ss;   // (threads without a slot: any readable word)
    gv = *src;
  }
  // previous Gram block of the frame (model-cost term): sum of its tiles, from the current buffer (flags bit 2; without it
  // both ping-pong buffers are fetched and the right one is picked later: round 1's form, kept for A/B)
  double g_old0 = 0.0, g_old1 = 0.0;
  if (tile == 0) {
    if (T == 1) {   // (straight line: no loop-carried sum, so no wait here)
      if (flags & 4) {
        g_old0 = P.blocks[((cur ? (size_t)P.F : 0) + f) * 256 + tid];   // (selected below: a copy here would wait)
      } else {
        g_old0 = P.blocks[(size_t)f * 256 + tid];
        g_old1 = P.blocks[((size_t)P.F + f) * 256 + tid];
      }
    } else if (flags & 4) {
      const size_t base = cur ? (size_t)P.F : 0;
      for (int k = 0; k < T; ++k) g_old0 += P.blocks[((base + f) * T + k) * 256 + tid];
    } else {
      for (int k = 0; k < T; ++k) {
        g_old0 += P.blocks[((size_t)f * T + k) * 256 + tid];
        g_old1 += P.blocks[(((size_t)P.F + f) * T + k) * 256 + tid];
      }
    }
  }
  if (tid < 116) sm[tid] = gv;
  if (done) return;
  if (phase != 0 && !step_valid) return;
  // mailbox exchange: this round's statistics will be exchanged by decide_elim<3> (which evaluates the
  // same predicate); advance their epoch here so that it cannot change while that kernel reads it.
  // Only sweeps that belong to a solve count (cc_intrinsics_eval / profile_sweep are rank-local).
  if (P.x.on && in_solve && blockIdx.x == 0 && tid == 0) P.x.seq[1] += 1ull;
  if (restart) {   // restore buffer 0 (what the restore kernel did) and clear the arrival counter of the elimination
    if (tile == 0 && tid >= 60 && tid < 67) P.pose[(size_t)f * 8 + (tid - 60)] = sm[tid];
    if (blockIdx.x == 0 && tid >= 74 && tid < 83) P.intr[tid - 74] = sm[tid];
    if (blockIdx.x == 0 && tid == 0) *P.arrive = 0u;
  }
  const int dst = phase == 0 ? cur : (cur ^ 1);
  const double g_old = (flags & 4) ? g_old0 : (cur ? g_old1 : g_old0);
  __syncthreads();
  const int pose_o = cur ? 67 : 60, intr_o = cur ? 83 : 74;
  if (tid < 6) {
    const double* Yr = sm + tid * 10;
    double a = Yr[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) a += Yr[j] * sm[92 + j];
    sm[129 + tid] = phase != 0 ? -a * sm[110 + tid] : 0.0;
  } else if (tid >= 8 && tid < 17) {
    const int j = tid - 8;
    const double d = (phase == 0 || (P.mask & (1u << j))) ? 0.0 : sm[92 + j] * sm[101 + j];
    sm[120 + j] = d;
    const double kc = sm[intr_o + j] + d;
    sm[148 + j] = kc;
    if (blockIdx.x == 0 && phase != 0) P.intr[dst * 16 + j] = kc;
  }
  __syncthreads();
  if (tid == 0) {
    double q[4], t[3], dp[6];
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = sm[pose_o + i];
#pragma unroll
    for (int i = 0; i < 3; ++i) t[i] = sm[pose_o + 4 + i];
#pragma unroll
    for (int i = 0; i < 6; ++i) dp[i] = sm[129 + i];
    double step2 = 0.0;
    if (phase != 0) {
      double qn[4];
      quat_plus(q, dp, qn);
#pragma unroll
      for (int i = 0; i < 4; ++i) { const double d = qn[i] - q[i]; step2 += d * d; q[i] = qn[i]; }
#pragma unroll
      for (int i = 0; i < 3; ++i) { const double tn = t[i] + dp[3 + i]; const double d = tn - t[i]; step2 += d * d; t[i] = tn; }
      if (tile == 0) {
        double* pose_dst = P.pose + ((size_t)dst * P.F + f) * 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) pose_dst[i] = q[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) pose_dst[4 + i] = t[i];
      }
    }
    double R[9];
    quat_to_R(q, R);
#pragma unroll
    for (int i = 0; i < 9; ++i) sm[136 + i] = R[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) sm[145 + i] = t[i];
    sm[158] = step2;
    sm[159] = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3] + t[0] * t[0] + t[1] * t[1] + t[2] * t[2];
  }
  __syncthreads();

  // model-cost term of this frame: q_f = d^T g_f + 1/2 d^T H_f d over the frame's 15x15 block at
  // the accepted point (Ceres: model_cost_change = -(J d)^T (r + J d / 2))
  double qterm = 0.0;
  if (phase != 0 && tile == 0) {
    const int a = tid >> 4, b = tid & 15;
    if (a < 15) qterm = b < 15 ? 0.5 * sm[120 + a] * g_old * sm[120 + b] : sm[120 + a] * g_old;
  }
  // reduced HERE, not next to its use behind the main loop: kept alive across the loop it was the one value that did
  // not fit the 128 registers of four waves per SIMD (a spill: 1 KB of scratch traffic per workgroup)
  {
    const double qw = wave_sum(qterm);
    if (lane == 0) sm[170 + wave] = qw;
  }

  double R[9], tt[3], kk[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) R[i] = rfl(sm[136 + i]);
#pragma unroll
  for (int i = 0; i < 3; ++i) tt[i] = rfl(sm[145 + i]);
#pragma unroll
  for (int i = 0; i < 9; ++i) kk[i] = rfl(sm[148 + i]);
  const uint32_t mask = P.mask;

  // ---- main loop: 64 observations per wave per pass, no workgroup barrier; the next pass's
  // observations are fetched while the current one is processed
  double* stage = s_stage + wave * kStageDoublesPerWave;
  d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
  for (int p = 0; p < npass; ++p) {
    const int64_t idx = s0 + (int64_t)p * kSweepThreads + tid;
    const bool valid = idx < s1;  // only the last pass of a frame has idle lanes
    const float2 m = nm;
    const float X0 = nX0, X1 = nX1, X2 = nX2;
    {   // next pass, unconditionally (the last pass fetches a slot nobody uses: cheaper than the wait a branch costs)
      const int64_t nidx = idx + kSweepThreads;
      const int64_t ic = nidx < s1 ? nidx : safe0;
      nm = uv2[ic];
      nX0 = P.xyz[ic * 3]; nX1 = P.xyz[ic * 3 + 1]; nX2 = P.xyz[ic * 3 + 2];
    }
    ObsCommon oc;
    obs_common(kk, R, tt, (double)X0, (double)X1, (double)X2, oc);
    double v[16];
    const double wrow = valid ? 1.0 : 0.0;   // (an idle lane's rows are zero: the weight rides on the rows' factors)
    row_u(kk, oc, (double)m.x, v, wrow);
    stage_row(stage, lane, v);
    wave_lds_fence();
    gram_rows(stage, lane, acc0, acc1);
    wave_lds_fence();
    row_v(kk, oc, (double)m.y, v, wrow);
    stage_row(stage, lane, v);
    wave_lds_fence();
    gram_rows(stage, lane, acc0, acc1);
    wave_lds_fence();
  }

  // ---- cross-wave reduction of the 16x16 block + model-cost term ---------------------------
  // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
  __syncthreads();  // s_blk aliases the staging buffers
#pragma unroll
  for (int r = 0; r < 4; ++r) s_blk[wave * 256 + ((lane >> 4) + 4 * r) * 16 + (lane & 15)] = acc0[r] + acc1[r];
  __syncthreads();
  const double g = gram_entry_held(mask, tid) ? 0.0 : (s_blk[tid] + s_blk[256 + tid]) + (s_blk[512 + tid] + s_blk[768 + tid]);   // (constant coordinates: zero rows and columns)
  P.blocks[(((size_t)dst * P.F + f) * T + tile) * 256 + tid] = g;
  // per-tile statistics row (the per-frame quantities ride on tile 0): written by the thread that holds entry (15, 15)
  // = sum r^2 of the block, no third barrier and no LDS copy of the block
  if (tid == 255) {
    double* st = P.stats + (size_t)blockIdx.x * kStatsCols;
    st[ST_COST] = 0.5 * g;
    st[ST_QMODEL] = (sm[170] + sm[171]) + (sm[172] + sm[173]);
    st[ST_STEP2] = tile == 0 ? sm[158] : 0.0;
    st[ST_XNORM2] = tile == 0 ? sm[159] : 0.0;
  }
  // initial evaluation: diagonal of the shared block for its Jacobi scale, by the threads that hold it (the pose
  // blocks' scale needs the sum over the tiles: the first elimination derives it, k_intr_decide_elim)
  if (phase == 0 && tid < 9 * 17 && tid % 17 == 0) P.hd0[(size_t)blockIdx.x * 16 + tid / 17] = g;
}

// ---------------------------------------------------------------------------------------------
// stats reduction over frames (column sums of stats[F][4] and, in phase 0, hd0[F][16]).
// Deterministic: fixed assignment of rows to threads, fixed combination tree.
// Result: out[0..3] stats sums, out[4..12] sum of hd0 columns 0..8 (zero when !want_hd).
// All threads of the 256-thread block must call; result valid for every thread after return.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void reduce_frame_stats(const IntrDev& P, bool want_stats, bool want_hd,
                                                   double* s_w /*[4][16]*/, double* out /*[16] shared*/,
                                                   unsigned long long* s_in = nullptr, unsigned long long word = 0ull) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // stats: thread t -> column t & 3, row group t >> 2 (64 groups)
  double a = 0.0;
  if (want_stats) {
    // 16 independent loads in flight per thread: one round trip per 1024 frames
    const int col = tid & 3;
    const int64_t R = P.F * P.T;   // one row per sweep workgroup
    for (int64_t fb = tid >> 2; fb < R; fb += 16 * 64) {
      double v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int64_t f = fb + u * 64;
        v[u] = f < R ? P.stats[f * kStatsCols + col] : 0.0;
      }
      a += (((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]))) +
           (((v[8] + v[9]) + (v[10] + v[11])) + ((v[12] + v[13]) + (v[14] + v[15])));
    }
  }
  // combine the 16 row groups of a wave that share a column (lane bits 2..5)
  a = wave_sum_mod<2>(a);
  // hd0 (initial round only): thread t -> column pair t & 7 (16-byte loads), row group t >> 3 of 32;
  // 16 loads in flight per thread, one round trip per 512 frames
  double h0 = 0.0, h1 = 0.0;
  if (want_hd) {
    const int cp = tid & 7;
    const d2* hd2 = reinterpret_cast<const d2*>(P.hd0);
    const int64_t R = P.F * P.T;
    for (int64_t fb = tid >> 3; fb < R; fb += 16 * 32) {
      d2 v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int64_t f = fb + u * 32;
        v[u] = f < R ? hd2[f * 8 + cp] : d2{0.0, 0.0};
      }
      h0 += (((v[0].x + v[1].x) + (v[2].x + v[3].x)) + ((v[4].x + v[5].x) + (v[6].x + v[7].x))) +
            (((v[8].x + v[9].x) + (v[10].x + v[11].x)) + ((v[12].x + v[13].x) + (v[14].x + v[15].x)));
      h1 += (((v[0].y + v[1].y) + (v[2].y + v[3].y)) + ((v[4].y + v[5].y) + (v[6].y + v[7].y))) +
            (((v[8].y + v[9].y) + (v[10].y + v[11].y)) + ((v[12].y + v[13].y) + (v[14].y + v[15].y)));
    }
  }
  h0 = wave_sum_mod<3>(h0);
  h1 = wave_sum_mod<3>(h1);
  if (lane < 4) s_w[wave * 16 + lane] = a;
  if (lane < 8) { s_w[64 + wave * 16 + 2 * lane] = h0; s_w[64 + wave * 16 + 2 * lane + 1] = h1; }
  if (s_in && tid < 48) s_in[tid] = word;   // the caller's gathered word rides on this barrier (k_intr_decide_elim)
  __syncthreads();
  if (tid < 4) out[tid] = (s_w[tid] + s_w[16 + tid]) + (s_w[32 + tid] + s_w[48 + tid]);
  else if (tid >= 4 && tid < 13) {
    const int c = tid - 4;
    out[tid] = (s_w[64 + c] + s_w[64 + 16 + c]) + (s_w[64 + 32 + c] + s_w[64 + 48 + c]);
  }
  __syncthreads();
}

// RCCL path only: local reduction -> vec_decide, which is then all-reduced
__global__ __launch_bounds__(256) void k_intr_stats_reduce(IntrDev P) {
  __shared__ double s_w[128];
  __shared__ double s_out[16];
  const LmCtl* ctl = P.ctl;
  if (ctl->done) return;
  const int phase = ctl->phase;
  const bool need = phase == 0 || (ctl->cand_pending && ctl->step_valid);
  reduce_frame_stats(P, need, phase == 0, s_w, s_out);
  if (threadIdx.x < 16) P.vec_decide[threadIdx.x] = (need && threadIdx.x < 13) ? s_out[threadIdx.x] : 0.0;
}

// ---------------------------------------------------------------------------------------------
// Solve step (one thread): gradient / radius tests of the accepted point (lm_finalize), then the reduced
// 9x9 system. V: the kVecSolve reduced sums (LDS); c: the control block after the decision; e: this
// iteration's log record (NULL when there is none). Writes the scaled shared step P.ds.
// V layout: [0..79] column sums (col 73 unused), [80 + rank] each rank's max |pose gradient|.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void intr_solve_step(const IntrDev& P, const double* V, LmCtl& c, cc_iteration* e, const LmOpts& o) {
  // gradient of the accepted point: max-norm over the tangent coordinates
  double gmax = 0.0;
  for (int r = 0; r < P.nranks && r < 32; ++r) gmax = fmax(gmax, V[kPartialCols + r]);
#pragma unroll
  for (int j = 0; j < 9; ++j)
    if (!(P.mask & (1u << j))) gmax = fmax(gmax, fabs(V[PC_GS + j]));
  if (e && e->accepted) e->gradient_max_norm = gmax;
  if (!lm_finalize(c, o, gmax)) return;
  bool ok = !(V[PC_FAIL] > 0.0);
  double A[45], b[9], inv[9];
  {
    int idx = 0;
#pragma unroll
    for (int j = 0; j < 9; ++j)
#pragma unroll
      for (int k = j; k < 9; ++k) { A[tri(k, j)] = V[idx]; ++idx; }
  }
#pragma unroll
  for (int j = 0; j < 9; ++j) {
    A[tri(j, j)] += clampd(V[PC_HDIAG + j], o.min_lm_diagonal, o.max_lm_diagonal) / c.radius;
    b[j] = V[PC_B + j];
  }
#pragma unroll
  for (int j = 0; j < 9; ++j)
    if (P.mask & (1u << j)) {
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        if (k < j) A[tri(j, k)] = 0.0;
        if (k > j) A[tri(k, j)] = 0.0;
      }
      A[tri(j, j)] = 1.0;
      b[j] = 0.0;
    }
  // Cholesky with reciprocal square roots; the substitutions multiply by 1 / L_jj
#pragma unroll
  for (int j = 0; j < 9; ++j) {
    double d = A[tri(j, j)];
#pragma unroll
    for (int k = 0; k < j; ++k) d -= A[tri(j, k)] * A[tri(j, k)];
    ok = ok && (d > 0.0) && isfinite(d);
    const double r = rsqrt_pos(d);   // (same bits as rsqrt for d > 0 finite -- anything else discards the step --, three dependent instructions fewer per pivot)
    inv[j] = r;
#pragma unroll
    for (int i = j + 1; i < 9; ++i) {
      double a = A[tri(i, j)];
#pragma unroll
      for (int k = 0; k < j; ++k) a -= A[tri(i, k)] * A[tri(j, k)];
      A[tri(i, j)] = a * r;
    }
  }
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    double a = b[i];
#pragma unroll
    for (int k = 0; k < i; ++k) a -= A[tri(i, k)] * b[k];
    b[i] = a * inv[i];
  }
#pragma unroll
  for (int i = 8; i >= 0; --i) {
    double a = b[i];
#pragma unroll
    for (int k = i + 1; k < 9; ++k) a -= A[tri(k, i)] * b[k];
    b[i] = a * inv[i];
  }
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    P.ds[i] = -b[i];
    ok = ok && isfinite(b[i]);
  }
  c.step_valid = ok ? 1 : 0;
  c.cand_pending = 1;
}

// ---------------------------------------------------------------------------------------------
// decide + elim (+ solve).  Every block reduces the (small) per-frame statistics itself and takes the
// same trust-region decision on a register copy of the control block. Then the block eliminates the
// pose blocks of its frames: 16 lanes per frame.
// Output row per block (80 columns): [0..44] upper triangle of the reduced 9x9 system (row-major
// pairs j<=k), [45..53] reduced rhs, [54..62] diag of the scaled H_ss, [63] Cholesky failures,
// [64..72] unscaled shared gradient, [73] max |pose gradient| (max-combined), rest 0.
// MODE 0 (single GPU) and MODE 3 (mailbox exchange): the row is stored write-through, the block arrives on
// a counter, and the LAST block to arrive sums the rows (sc1 loads: no fence needed, MI355X guide, valid
// hand-off forms), [MODE 3: exchanges the sums with the other ranks,] runs the solve step and publishes
// the control block: nobody else writes it, and everybody has read it before the last arrival.
// MODE 2 (RCCL): statistics come all-reduced in vec_decide; block 0 publishes the decision and the solve
// step runs in k_intr_solve<1> -> all-reduce -> k_intr_solve<2>.
// publish != 0: last kernel of a host chunk -> also hand the control block to the host (pinned memory).
// ---------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void k_intr_decide_elim(IntrDev P, int publish) {
  __shared__ double Zs[16][64];
  __shared__ double red[16][kPartialCols];
  __shared__ double Gs[16 * 256];   // tiled frames only: the 16 frames' Gram blocks summed over their tiles
  __shared__ double s_w[128];
  __shared__ double s_tot[16];
  __shared__ double s_ss[16];
  __shared__ LmCtl s_ctl;
  __shared__ cc_iteration s_log;
  __shared__ int s_logged;
  __shared__ int s_last;
  __shared__ unsigned char pj[48], pk[48];
  constexpr bool kFused = MODE != 2;
  const int tid = threadIdx.x, g = tid >> 4, l = tid & 15;
  const LmCtl* ctl = P.ctl;
  // bit 2 of `publish`: RESTART, first elimination of a solve from the initial state: the control block in memory
  // is stale and counts as zero (cf. the restart sweep)
  const bool restart = (publish & 4) != 0;
  publish &= 3;
  // Everything the decision needs comes in ONE vector round trip: thread t fetches one 8-byte word of
  // [control block (18) | options (12) | intrinsics, both buffers (9 + 9)] into LDS, and -- single GPU -- the
  // statistics rows are requested in the same round trip, before anybody knows whether this launch needs them (that
  // is written in the control block still on its way). Reading the fields one by one (`restart ? 0 : ctl->phase`,
  // `*P.opts` in thread 0's branch, ...) compiled to seven dependent scalar-load round trips, ~2.5 us of a 14 us kernel.
  // The kernel's arguments are one large by-value struct; the compiler fetches its fields from the kernarg segment
  // where they are first used, each fetch a scalar load with a full wait behind it (eight dependent rounds before the
  // first global load left). Naming the fields here makes them live at this point: one batch of scalar loads, one wait.
  asm volatile("" ::"s"(P.F), "s"(P.T), "s"(P.stats), "s"(P.hd0), "s"(P.ctl), "s"(P.opts), "s"(P.intr), "s"(P.blocks),
               "s"(P.sp), "s"(P.Y), "s"(P.partial), "s"(P.ss), "s"(P.arrive));
  __shared__ unsigned long long s_in[64];
  unsigned long long word;
  {
    const unsigned long long* src = reinterpret_cast<const unsigned long long*>(ctl);
    if (tid < 18) src = reinterpret_cast<const unsigned long long*>(ctl) + tid;
    else if (tid < 30) src = reinterpret_cast<const unsigned long long*>(P.opts) + (tid - 18);
    else if (tid < 39) src = reinterpret_cast<const unsigned long long*>(P.intr) + (tid - 30);
    else if (tid < 48) src = reinterpret_cast<const unsigned long long*>(P.intr) + 16 + (tid - 39);
    word = *src;
  }
  static_assert(sizeof(LmCtl) == 18 * 8 && sizeof(LmOpts) == 12 * 8, "layout of the gathered decision inputs");
  if (MODE == 0) {
    reduce_frame_stats(P, true, restart, s_w, s_tot, s_in, word);   // (restart: the initial evaluation, known from the launch)
  } else {
    if (tid < 48) s_in[tid] = word;
    __syncthreads();
  }
  const LmCtl& c_in = *reinterpret_cast<const LmCtl*>(s_in);
  const LmOpts& o_in = *reinterpret_cast<const LmOpts*>(s_in + 18);
  const double* k_in0 = reinterpret_cast<const double*>(s_in + 30);
  const double* k_in1 = reinterpret_cast<const double*>(s_in + 39);
  if (!restart && c_in.done) {
    if (publish && blockIdx.x == 0 && tid == 0) publish_to_host(P, c_in);
    return;
  }
  const int phase = restart ? 0 : c_in.phase;
  const bool pending = !restart && c_in.cand_pending != 0;
  const bool need = phase == 0 || (pending && c_in.step_valid);
  bool exchange_ok = true;
  if (MODE == 0) {
    // (an initial evaluation that was not announced by the launch flag -- the first solve after set_state -- needs
    // the diagonal sums too: reduce again, the rare case)
    if (phase == 0 && !restart) reduce_frame_stats(P, true, true, s_w, s_tot);
  } else if (MODE == 3) {
    // mailbox exchange (kind 1): block 0 reduces this rank's statistics and posts them into every
    // rank's mailbox (its own included); all blocks then poll the local words and add the slots
    // in rank order. The epoch was advanced by this round's sweep, so it is stable while we run.
    __shared__ int s_ok;
    if (need) {
      const unsigned long long epoch = P.x.seq[1];
      if (blockIdx.x == 0) {
        if (tid >= 13 && tid < 16) s_tot[tid] = 0.0;
        reduce_frame_stats(P, true, phase == 0, s_w, s_tot);
        p2p_post(P.x, 1, epoch, P.rank, P.nranks, s_tot, 16);
        __syncthreads();
      }
      const double a = p2p_collect(P.x, 1, epoch, P.rank, P.nranks, 16, &s_ok);
      exchange_ok = s_ok != 0;
      if (tid < 16) s_tot[tid] = exchange_ok ? a : 0.0;
    } else if (tid < 16) {
      s_tot[tid] = 0.0;
    }
    __syncthreads();
  } else {
    if (tid < 16) s_tot[tid] = P.vec_decide[tid];
    __syncthreads();
  }
  if (tid == 0) {
    LmCtl c{};
    if (!restart) c = c_in;
    const LmOpts o = o_in;
    const int len0 = c.log_len;
    // (single GPU: the statistics were reduced before `need` was known; a launch that does not need them sees zeros,
    // as it did when the reduction was skipped)
    double tot[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) tot[i] = (MODE != 0 || need) ? s_tot[i] : 0.0;
    const double* kc0 = c.cur ? k_in1 : k_in0;   // accepted intrinsics
    const double* kc1 = c.cur ? k_in0 : k_in1;   // candidate
    if (phase == 0) {
      double xn2 = tot[ST_XNORM2];
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        const double ki = kc0[i];
        xn2 += ki * ki;
        const double sc = o.jacobi_scaling ? 1.0 / (1.0 + sqrt(s_tot[4 + i])) : 1.0;
        s_ss[i] = sc;
        if (blockIdx.x == 0) P.ss[i] = sc;
      }
      lm_init(c, o, tot[ST_COST], sqrt(xn2));
    } else if (pending) {
      double step2 = tot[ST_STEP2], xn2 = tot[ST_XNORM2];
      if (c.step_valid) {
#pragma unroll
        for (int i = 0; i < 9; ++i) {
          const double kc = kc1[i], k0 = kc0[i];
          const double d = kc - k0;
          step2 += d * d;
          xn2 += kc * kc;
        }
      }
      lm_decide(c, o, &s_log, tot[ST_COST], tot[ST_QMODEL], step2, xn2);
    }
    if (!exchange_ok) { c.done = 1; c.term = CC_FAILURE_EXCHANGE; }
    s_ctl = c;
    s_logged = c.log_len != len0;
    if (!kFused && blockIdx.x == 0) {
      if (s_logged && c.log_len <= P.log_cap) P.log[c.log_len - 1] = s_log;
      *P.ctl_next = c;
    }
  }
  if (phase != 0 && tid < 9) s_ss[tid] = P.ss[tid];
  if (tid == 32) {
    int o = 0;
    for (int j = 0; j < 9; ++j)
      for (int k = j; k < 9; ++k) { pj[o] = (unsigned char)j; pk[o] = (unsigned char)k; ++o; }
  }
  __syncthreads();
  const bool stop = s_ctl.done != 0;   // the same answer in every block
  if (stop && !kFused) return;
  if (!stop) {
  const int cur = s_ctl.cur;
  const int T = P.T;
  const bool jac = o_in.jacobi_scaling != 0;
  const double inv_radius = 1.0 / s_ctl.radius;
  const double mn = o_in.min_lm_diagonal, mx = o_in.max_lm_diagonal;

  // Output slots l*5 + r of this lane: what they read is the same for every frame, so the source
  // index, the scales and the Z columns live in registers and the loads go out with the Cholesky's.
  int gi[5], zj[5], zk[5];
  double sa[5], sb[5];
  bool use_z[5];
#pragma unroll
  for (int r = 0; r < 5; ++r) {
    const int o = l * 5 + r;
    gi[r] = 0; zj[r] = 0; zk[r] = 0; sa[r] = 0.0; sb[r] = 0.0; use_z[r] = false;
    if (o < 45) {
      const int j = pj[o], k = pk[o];
      gi[r] = j * 16 + k; zj[r] = j; zk[r] = k; sa[r] = s_ss[j]; sb[r] = s_ss[k]; use_z[r] = true;
    } else if (o < 54) {
      const int j = o - 45;
      gi[r] = j * 16 + 15; zj[r] = j; zk[r] = 9; sa[r] = s_ss[j]; sb[r] = 1.0; use_z[r] = true;
    } else if (o < 63) {
      const int j = o - 54;
      gi[r] = j * 17; sa[r] = s_ss[j] * s_ss[j]; sb[r] = 1.0;
    } else if (o >= PC_GS && o < PC_GS + 9) {
      gi[r] = (o - PC_GS) * 16 + 15; sa[r] = 1.0; sb[r] = 1.0;
    }  // PC_FAIL, PC_GMAXP and the padding columns: 0 here, the two live ones are filled in below
  }
  const int l6 = l < 6 ? l : l - 6 < 6 ? l - 6 : l - 12;
  double acc[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
  double gacc = 0.0, facc = 0.0;
  for (int64_t base = (int64_t)blockIdx.x * 16; base < P.F; base += (int64_t)gridDim.x * 16) {
    const int64_t f = base + g;
    const bool valid = f < P.F;
    const double* G = P.blocks + ((size_t)cur * P.F + (valid ? f : 0)) * T * 256;
    // entry `idx` of the frame's Gram block. Tiled frames (T > 1): the block's 16 frames are summed over their
    // tiles into LDS first (T round trips of 16 coalesced loads per thread), then read from there.
    if (T > 1) {
      __syncthreads();   // (previous iteration's readers)
      double a[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) a[u] = 0.0;
      for (int k = 0; k < T; ++k) {
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const int64_t fu = base + u;
          v[u] = fu < P.F ? P.blocks[(((size_t)cur * P.F + fu) * T + k) * 256 + tid] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) a[u] += v[u];
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) Gs[u * 256 + tid] = a[u];
      __syncthreads();
    }
    const double* Gl = Gs + g * 256;
    auto gsum = [&](int idx) { return T > 1 ? Gl[idx] : G[idx]; };
    double fail = 0.0;
    double gv[5];
#pragma unroll
    for (int r = 0; r < 5; ++r) gv[r] = gsum(gi[r]);
    const double gpe = gsum((9 + l6) * 16 + 15);   // entry l6 of the pose block's gradient (lanes 0..5 of the sixteen: entry l)
    const double* qf = P.pose + ((size_t)cur * P.F + (valid ? f : 0)) * 8;
    const double qw = qf[0], qx = qf[1], qy = qf[2], qz = qf[3];
    // column l (< 10) of [H_ps | g_p], loaded with everything else in one round trip
    const int col = l < 9 ? l : 15;
    double w[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) w[i] = gsum((9 + i) * 16 + col);
    if (valid) {
      double s[6], L[21];
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) L[tri(i, j)] = gsum((9 + i) * 16 + 9 + j);
      if (phase == 0) {
        // first elimination: Jacobi scale of this frame's pose block (Ceres: 1 / (1 + sqrt(diag J^T J)), once)
#pragma unroll
        for (int i = 0; i < 6; ++i) s[i] = jac ? 1.0 / (1.0 + sqrt(L[tri(i, i)])) : 1.0;
        if (l < 6) P.sp[f * 8 + l] = jac ? 1.0 / (1.0 + sqrt(gsum((9 + l) * 17))) : 1.0;
      } else {
#pragma unroll
        for (int i = 0; i < 6; ++i) s[i] = P.sp[f * 8 + i];
      }
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) L[tri(i, j)] = s[i] * L[tri(i, j)] * s[j];
#pragma unroll
      for (int i = 0; i < 6; ++i) L[tri(i, i)] += clampd(L[tri(i, i)], mn, mx) * inv_radius;
      // in-place Cholesky (lower), fully unrolled so L stays in registers (redundant per lane);
      // one rsqrt per pivot: L_jj = d * rsqrt(d), 1 / L_jj = rsqrt(d)
      bool ok = true;
      double Li[6];  // 1 / L_jj
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        double d = L[tri(j, j)];
#pragma unroll
        for (int k = 0; k < j; ++k) d -= L[tri(j, k)] * L[tri(j, k)];
        ok = ok && (d > 0.0) && isfinite(d);
        const double inv = rsqrt_pos(d);
        L[tri(j, j)] = d * inv;
        Li[j] = inv;
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
        // z = L^-1 w (for the Schur sums), y = L^-T z (for the back-substitution)
        const double sc = l < 9 ? s_ss[l] : 1.0;
        double z[6], y[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          double a = s[i] * w[i] * sc;
#pragma unroll
          for (int k = 0; k < i; ++k) a -= L[tri(i, k)] * z[k];
          z[i] = a * Li[i];
        }
#pragma unroll
        for (int i = 5; i >= 0; --i) {
          double a = z[i];
#pragma unroll
          for (int k = i + 1; k < 6; ++k) a -= L[tri(k, i)] * y[k];
          y[i] = a * Li[i];
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          Zs[g][i * 10 + l] = z[i];
          P.Y[f * kYStride + i * 10 + l] = y[i];
        }
      }
    }
    __syncthreads();
    if (valid) {
#pragma unroll
      for (int r = 0; r < 5; ++r) {
        double zz = 0.0;
#pragma unroll
        for (int i = 0; i < 6; ++i) zz += Zs[g][i * 10 + zj[r]] * Zs[g][i * 10 + zk[r]];
        acc[r] += sa[r] * gv[r] * sb[r] - (use_z[r] ? zz : 0.0);
      }
      {   // the frame's share of Ceres' gradient_max_norm, ||x - Plus(x, -g)||_inf (pose_grad_proj_max, cc_common.hpp)
        const int base16 = (threadIdx.x & 63) & ~15;
        double g6[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) g6[i] = __shfl(gpe, base16 + i, 64);
        const double q4[4] = {qw, qx, qy, qz};
        gacc = fmax(gacc, pose_grad_proj_max(q4, g6));
      }
      facc += fail;
    }
    __syncthreads();
  }
#pragma unroll
  for (int r = 0; r < 5; ++r)
    if (l * 5 + r != PC_FAIL && l * 5 + r != PC_GMAXP) red[g][l * 5 + r] = acc[r];
  gacc = row16_max(gacc);
  if (l == 0) { red[g][PC_FAIL] = facc; red[g][PC_GMAXP] = gacc; }
  __syncthreads();
  if (tid < kPartialCols) {
    double a = 0.0;
    if (tid == PC_GMAXP) { for (int g2 = 0; g2 < 16; ++g2) a = fmax(a, red[g2][tid]); }
    else { for (int g2 = 0; g2 < 16; ++g2) a += red[g2][tid]; }
    if (kFused)   // write-through: the last block reads these words with sc1 loads
      __hip_atomic_store(reinterpret_cast<unsigned long long*>(P.partial) + blockIdx.x * kPartialCols + tid,
                         (unsigned long long)__double_as_longlong(a), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
      P.partial[blockIdx.x * kPartialCols + tid] = a;
  }
  }  // !stop
  if (!kFused) return;

  // ---- last-block-done: every storing wave drains its stores, the block arrives, the last one goes on
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    const unsigned prev = __hip_atomic_fetch_add(P.arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = prev + 1u == gridDim.x;
  }
  __syncthreads();
  if (!s_last) return;
  if (tid == 0) __hip_atomic_store(P.arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // next launch
  double* sv = &red[0][0];            // [kVecSolve] reduced sums (the staging rows are no longer needed)
  double* s_part = &red[2][0];        // [3][kPartialCols]
  if (!stop) {
    const bool active = true;
    {
      // thread -> (column, row group of 3): all rows in one round trip, sc1 loads
      const int col = tid % kPartialCols, grp = tid / kPartialCols;
      const int nblk = (int)gridDim.x;
      if (grp < 3) {
        const unsigned long long* src = reinterpret_cast<const unsigned long long*>(P.partial) + col;
        double v[22];
#pragma unroll
        for (int u = 0; u < 22; ++u) {
          const int b = grp + 3 * u;
          v[u] = b < nblk ? __longlong_as_double((long long)__hip_atomic_load(src + b * kPartialCols, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) : 0.0;
        }
        double a = 0.0;
        if (col == PC_GMAXP) {
#pragma unroll
          for (int u = 0; u < 22; ++u) a = fmax(a, v[u]);
        } else {
#pragma unroll
          for (int u = 0; u < 22; ++u) a += v[u];
        }
        s_part[grp * kPartialCols + col] = a;
      }
    }
    __syncthreads();
    if (tid < kVecSolve) {   // sv = red[0..1] and s_part = red[2..4] do not overlap
      double a = 0.0;
      if (tid < kPartialCols) {
        if (tid != PC_GMAXP) a = (s_part[tid] + s_part[kPartialCols + tid]) + s_part[2 * kPartialCols + tid];
      } else if (tid == kPartialCols + P.rank) {
        a = fmax(fmax(s_part[PC_GMAXP], s_part[kPartialCols + PC_GMAXP]), s_part[2 * kPartialCols + PC_GMAXP]);
      }
      sv[tid] = a;
    }
    __syncthreads();
    if (MODE == 3 && active) {
      // mailbox all-reduce of the 112 sums (kind 0): post, wait for every rank, add in rank order
      __shared__ int s_ok2;
      const unsigned long long epoch = P.x.seq[0] + 1ull;
      p2p_post(P.x, 0, epoch, P.rank, P.nranks, sv, kVecSolve);
      const double a = p2p_collect(P.x, 0, epoch, P.rank, P.nranks, kVecSolve, &s_ok2);
      exchange_ok = s_ok2 != 0;
      if (tid < kVecSolve) sv[tid] = exchange_ok ? a : 0.0;
      if (tid == 0) P.x.seq[0] = epoch;
      __syncthreads();
    }
  }
  if (tid != 0) return;
  LmCtl c = s_ctl;
  cc_iteration* e = s_logged ? &s_log : nullptr;
  if (!exchange_ok) { c.done = 1; c.term = CC_FAILURE_EXCHANGE; }
  else if (!stop) intr_solve_step(P, sv, c, e, o_in);
  if (publish == 2) return;   // timing replay (cc_intrinsics_profile_kernel): the state stays as it is
  if (e && c.log_len <= P.log_cap) P.log[c.log_len - 1] = *e;
  *P.ctl = c;
  *P.ctl_next = c;
  if (publish) publish_to_host(P, c);
}

// ---------------------------------------------------------------------------------------------
// RCCL route only. MODE 1: reduce the elimination partials of this rank -> vec_solve (then all-reduced);
// MODE 2: the solve step on the all-reduced sums; publishes the control block for the sweep.
// vec_solve: [0..79] column sums (col 73 unused), [80 + rank] this rank's max |pose gradient|.
// ---------------------------------------------------------------------------------------------
constexpr int kSolveThreads = 8 * kPartialCols;

template <int MODE>
__global__ __launch_bounds__(kSolveThreads) void k_intr_solve(IntrDev P, int nblk, int publish) {
  static_assert(MODE == 1 || MODE == 2, "single-GPU and mailbox solves are fused into k_intr_decide_elim");
  __shared__ double sv[kVecSolve];
  __shared__ double s_part[8][kPartialCols];
  const int tid = threadIdx.x;
  const LmCtl* cn = P.ctl_next;
  const bool active = !cn->done && cn->phase != 0;
  if (MODE == 1) {
    // thread -> (column, row group): all rows of the partials are fetched in one round trip
    const int col = tid % kPartialCols, grp = tid / kPartialCols;
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int b = grp + 8 * u;
      v[u] = b < nblk ? P.partial[b * kPartialCols + col] : 0.0;
    }
    double a = 0.0;
    if (active) {
      if (col == PC_GMAXP) a = fmax(fmax(fmax(v[0], v[1]), fmax(v[2], v[3])), fmax(fmax(v[4], v[5]), fmax(v[6], v[7])));
      else a = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    }
    s_part[grp][col] = a;
    if (tid >= kPartialCols && tid < kVecSolve) sv[tid] = 0.0;  // rank slots; ours is written below
    __syncthreads();
    if (tid < kPartialCols) {
      double a2 = 0.0;
      if (tid == PC_GMAXP) {
        if (active) for (int g2 = 0; g2 < 8; ++g2) a2 = fmax(a2, s_part[g2][tid]);
        sv[kPartialCols + P.rank] = a2;
        sv[tid] = 0.0;
      } else {
        if (active)
          a2 = ((s_part[0][tid] + s_part[1][tid]) + (s_part[2][tid] + s_part[3][tid])) +
               ((s_part[4][tid] + s_part[5][tid]) + (s_part[6][tid] + s_part[7][tid]));
        sv[tid] = a2;
      }
    }
    __syncthreads();
    if (tid < kVecSolve) P.vec_solve[tid] = sv[tid];
    return;
  }
  if (tid < kVecSolve) sv[tid] = P.vec_solve[tid];
  __syncthreads();
  if (tid != 0) return;
  LmCtl c = *cn;
  if (active) {
    // the log record of this iteration was stored by decide_elim<2>; complete it in place
    cc_iteration* e = (c.log_len > 0 && c.log_len <= P.log_cap) ? &P.log[c.log_len - 1] : nullptr;
    cc_iteration rec;
    if (e) rec = *e;
    const LmOpts o = *P.opts;
    intr_solve_step(P, sv, c, e ? &rec : nullptr, o);
    if (e && rec.accepted) e->gradient_max_norm = rec.gradient_max_norm;
  }
  *P.ctl = c;
  *P.ctl_next = c;
  if (publish) publish_to_host(P, c);
}

// one-off (attach time): sum of one number per rank through the mailboxes (kind 1) -- do ALL ranks fit the persistent kernel?
__global__ __launch_bounds__(64) void k_intr_flag_exchange(IntrDev P, double mine, double* out, int* ok) {
  __shared__ double s_post[16];
  __shared__ int s_ok;
  const int tid = threadIdx.x;
  if (tid < 16) s_post[tid] = tid == 0 ? mine : 0.0;
  __syncthreads();
  const unsigned long long epoch = P.x.seq[1] + 1ull;
  p2p_post(P.x, 1, epoch, P.rank, P.nranks, s_post, 16);
  const double a = p2p_collect(P.x, 1, epoch, P.rank, P.nranks, 16, &s_ok);
  if (tid == 0) { *out = a; P.x.seq[1] = epoch; *ok = s_ok; }
}

// restores the initial point into buffer 0 and clears the control blocks and the arrival counter (first
// node of a solve-from-the-initial-state graph, or one launch per restart)
__global__ __launch_bounds__(256) void k_intr_reset(IntrDev P, const double* init_intr, const double* init_pose) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < P.F * 8) P.pose[i] = init_pose[i];
  if (i < 16) P.intr[i] = init_intr[i];
  if (i < (int64_t)(sizeof(LmCtl) / sizeof(double))) {
    reinterpret_cast<double*>(P.ctl)[i] = 0.0;
    reinterpret_cast<double*>(P.ctl_next)[i] = 0.0;
  }
  if (i == 0) *P.arrive = 0u;
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
  void* arena = nullptr;        // every device buffer of the handle lives in this one allocation (cc::arena_get: cached between handles)
  bool arena_cached = false;
  size_t extra_bytes = 0;       // set before creation: scratch appended to the arena (cc_intrinsics_estimate's initialisation)
  char* extra = nullptr;
  bool one_shot = false;        // the creator keeps uv / xyz alive until it has synchronised: no wait at the end of create
  bool no_obs_upload = false;   // ... and uploads the observations itself
  double* init_intr = nullptr;  // [16]
  double* init_pose = nullptr;  // [F][8]
  bool have_state = false;
  void* pinned = nullptr;      // cached 512-byte pinned block: host_pub (160 B) | h_ctl | h_opts
  volatile unsigned long long* host_pub = nullptr;  // [0] sequence word written by the device, [2..19] control block
  unsigned long long pub_count = 0;                 // chunks published so far (what the sequence word will read next: + 1)
  cc::LmCtl* h_ctl = nullptr;  // pinned
  cc::LmOpts* h_opts = nullptr;  // pinned staging of the options
  cc::LmOpts cached_opts{};     // what the device currently holds
  bool opts_valid = false;
  bool ctl_fresh = false;       // the next solve starts from the point of the last set_state
  bool reset_pending = false;   // ... and the device buffers have not been restored yet (lazy: the first round of the
                                // solve runs as the restart round, cc_intrinsics_reset costs no launch of its own)
  // [0]: initial evaluation as the restart round + check_interval iterations, [1]: check_interval iterations,
  // [2]: initial evaluation + check_interval iterations (continuing from the accepted point of the last solve)
  hipGraphExec_t graph[3] = {nullptr, nullptr, nullptr};
  int graph_iters = 0;
  cc::Comm* comm = nullptr;
  // mailbox exchange (cc_intrinsics_exchange_export / _attach): our mailbox and the peers' mappings
  cc::Mailbox mailbox;
  bool exchange = false;
  std::vector<hipEvent_t> events;
  std::vector<int> event_kind;
  std::vector<int> event_round;   // round of the solve a probed launch belongs to (summarise_probes)
  int enq_round = 0;
  // persistent per-solve kernel (cc_intrinsics_persist.hip): usable when every frame gets a team of a resident workgroup
  int ran_form = -1;            // the form the handle's LAST solve ran in to its end (-1: none yet): 0 two kernels per iteration, else frames per persistent workgroup
  int form_reruns = 0;          // persistent solves that gave up and were run again in the two-kernel form (cc_intrinsics_solver_status)
  std::string form_note;        // why
  bool persist_ok = false;      // ... on a device of its own
  bool persist_x_ok = false;    // ... as one rank of an exchange: every rank fits next to the ranks it shares its device with,
                                // and every rank said so (cc_intrinsics_exchange_attach / cc_intrinsics_optimize_multi agree on it)
  int p_resident = 0;           // resident workgroups of the chosen shape on this device
  cc::PersistDev pq{};
  uint32_t p_epoch = 0;         // last epoch handed to a launch (the seam words only ever see growing epochs)
  size_t p_box_bytes = 0;       // seam mailboxes (re-zeroed before the epochs wrap)
};

namespace cc {

static void drop_graphs(cc_intrinsics* h) {
  for (auto& g : h->graph)
    if (g) { hipGraphExecDestroy(g); g = nullptr; }
}

static void exchange_release(cc_intrinsics* h) {
  mailbox_release(&h->mailbox);
  h->d.x = P2pDev{};
  h->exchange = false;
}

struct Probe {  // optional hipEvent bracket around one launch
  cc_intrinsics* h; int kind; bool on; int round_shift; hipEvent_t e0 = nullptr, e1 = nullptr;
  Probe(cc_intrinsics* h_, int kind_, bool on_, int round_shift_ = 0) : h(h_), kind(kind_), on(on_), round_shift(round_shift_) {
    if (on) { hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0, h->stream); }
  }
  ~Probe() {
    if (on) { hipEventRecord(e1, h->stream); h->events.push_back(e0); h->events.push_back(e1); h->event_kind.push_back(kind); h->event_round.push_back(h->enq_round + round_shift); }
  }
};

static void launch_sweep(cc_intrinsics* h, bool profile, int flags = 0) {
  Probe p(h, CC_K_SWEEP, profile);
  // Only the CURRENT Gram buffer of a frame is fetched for the model-cost term (flag bit 2). Round 1 fetched both ping-pong
  // buffers so as not to wait for the control block; with the control block read by scalar loads at the top of the kernel
  // the index is there when the gather is issued: A/B at 1000 x 500, sweep 17.41 vs 17.42 us, iteration 42.50 vs 42.41 us,
  // and 2 KB per frame less traffic.
  flags |= 4;
  hipLaunchKernelGGL(k_intr_sweep, dim3((unsigned)(h->F * h->d.T)), dim3(kSweepThreads), kSweepLdsBytes, h->stream, h->d, flags);
}

static void launch_reset(cc_intrinsics* h) {
  const unsigned blocks = (unsigned)((h->F * 8 + 255) / 256);
  hipLaunchKernelGGL(k_intr_reset, dim3(blocks), dim3(256), 0, h->stream, h->d, h->init_intr, h->init_pose);
}

// One round = sweep -> decide + elim + solve step (two kernels; the very first round of a solve is the initial
// evaluation, same launches). RCCL route: the solve step is a pair of kernels around the all-reduce of the
// reduced sums at the head of the round (left out of the initial round: nothing to solve yet).
// publish: last round of a host chunk -> its final kernel hands the control block to the host.
static int enqueue_round(cc_intrinsics* h, bool profile, bool initial, bool publish, bool restart = false) {
  const int pub = (publish ? 1 : 0) | (restart ? 4 : 0);
  const int sweep_flags = 1 | (restart ? 2 : 0);
  struct RoundCount { cc_intrinsics* h; ~RoundCount() { h->enq_round++; } } count_round{h};
  if (h->comm) {
    if (!initial) {   // (the solve step at the head of round r belongs to the iteration decided in round r - 1)
      { Probe p(h, CC_K_SOLVE, profile, -1); hipLaunchKernelGGL(k_intr_solve<1>, dim3(1), dim3(kSolveThreads), 0, h->stream, h->d, h->elim_blocks, 0); }
      { Probe p(h, CC_K_ALLREDUCE, profile, -1); if (int rc = comm_allreduce_sum(h->comm, h->d.vec_solve, kVecSolve, h->stream)) return rc; }
      { Probe p(h, CC_K_SOLVE, profile, -1); hipLaunchKernelGGL(k_intr_solve<2>, dim3(1), dim3(kSolveThreads), 0, h->stream, h->d, h->elim_blocks, 0); }
    }
    launch_sweep(h, profile, 1);
    { Probe p(h, CC_K_DECIDE, profile); hipLaunchKernelGGL(k_intr_stats_reduce, dim3(1), dim3(256), 0, h->stream, h->d); }
    { Probe p(h, CC_K_ALLREDUCE, profile); if (int rc = comm_allreduce_sum(h->comm, h->d.vec_decide, 16, h->stream)) return rc; }
    { Probe p(h, CC_K_ELIM, profile); hipLaunchKernelGGL(k_intr_decide_elim<2>, dim3(h->elim_blocks), dim3(256), 0, h->stream, h->d, 0); }
    return 0;
  }
  launch_sweep(h, profile, sweep_flags);
  Probe p(h, CC_K_ELIM, profile);
  if (h->exchange) hipLaunchKernelGGL(k_intr_decide_elim<3>, dim3(h->elim_blocks), dim3(256), 0, h->stream, h->d, pub);
  else hipLaunchKernelGGL(k_intr_decide_elim<0>, dim3(h->elim_blocks), dim3(256), 0, h->stream, h->d, pub);
  return 0;
}

static int write_ctl(cc_intrinsics* h, const LmCtl& c) {
  CC_HIP(hipMemcpyAsync(h->d.ctl, &c, sizeof(c), hipMemcpyHostToDevice, h->stream));
  CC_HIP(hipMemcpyAsync(h->d.ctl_next, &c, sizeof(c), hipMemcpyHostToDevice, h->stream));
  return 0;
}

// restores the point of the last set_state on the device if that is still owed (see reset_pending)
static int flush_reset(cc_intrinsics* h) {
  if (!h->reset_pending) return 0;
  launch_reset(h);
  CC_HIP(hipGetLastError());
  h->reset_pending = false;
  return 0;
}

static int read_ctl(cc_intrinsics* h, LmCtl* c) {
  if (int rc = flush_reset(h)) return rc;
  CC_HIP(hipMemcpyAsync(h->h_ctl, h->d.ctl_next, sizeof(LmCtl), hipMemcpyDeviceToHost, h->stream));
  CC_HIP(hipStreamSynchronize(h->stream));
  *c = *h->h_ctl;
  return 0;
}

// Waits for the chunk just enqueued: spins on the sequence word its last kernel stores into pinned host memory
// (no copy engine, no stream synchronisation on the way), then takes the control block from next to it. A
// stream that has gone idle without the word showing up (a kernel fault, a stale counter) falls back to a copy.
static int wait_published(cc_intrinsics* h, LmCtl* c) {
  const unsigned long long want = ++h->pub_count;
  for (unsigned spins = 0;; ++spins) {
    if (__atomic_load_n(const_cast<const unsigned long long*>(h->host_pub), __ATOMIC_ACQUIRE) == want) break;
    if ((spins & 0xfffu) == 0xfffu) {
      const hipError_t q = hipStreamQuery(h->stream);
      if (q == hipSuccess) {
        if (__atomic_load_n(const_cast<const unsigned long long*>(h->host_pub), __ATOMIC_ACQUIRE) == want) break;
        h->pub_count = __atomic_load_n(const_cast<const unsigned long long*>(h->host_pub), __ATOMIC_ACQUIRE);
        return read_ctl(h, c);
      }
      if (q != hipErrorNotReady) return fail(CC_ERR_HIP, "stream failed while waiting for the solver: %s", hipGetErrorString(q));
    }
  }
  std::memcpy(c, const_cast<const unsigned long long*>(h->host_pub) + 2, sizeof(LmCtl));
  return 0;
}

}  // namespace cc

namespace cc {
int intr_create_impl(cc_intrinsics* h, const int64_t* off, const float* uv, const float* xyz);
// (how the one-shot entry points -- cc_intrinsics_optimize / _estimate -- ask cc_intrinsics_create for a handle of their own
// kind: the upload is not waited for, `extra` bytes of scratch ride in the same arena)
static thread_local bool g_create_one_shot = false;
static thread_local bool g_create_no_obs = false;   // the creator uploads uv / xyz itself (in pieces, as it packs them)
static thread_local size_t g_create_extra = 0;
struct OneShotCreate {
  OneShotCreate(size_t extra, bool no_obs = false) { g_create_one_shot = true; g_create_extra = extra; g_create_no_obs = no_obs; }
  ~OneShotCreate() { g_create_one_shot = false; g_create_extra = 0; g_create_no_obs = false; }
};
static double ms_since(std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
}

extern "C" {

void cc_intrinsics_destroy(cc_intrinsics* h);

int cc_intrinsics_create(int32_t device, int64_t F, const int64_t* off, const float* uv,
                         const float* xyz, cc_intrinsics** out) {
  using namespace cc;
  if (!out || !off || F <= 0) return fail(CC_ERR_BAD_ARGUMENT, "cc_intrinsics_create: bad arguments");
  if (off[0] != 0) return fail(CC_ERR_BAD_ARGUMENT, "frame_offsets[0] must be 0");
  for (int64_t f = 0; f < F; ++f)
    if (off[f + 1] < off[f]) return fail(CC_ERR_BAD_ARGUMENT, "frame_offsets must be non-decreasing");
  const int64_t N = off[F];
  if (F >= ((int64_t)1 << 28) || N >= ((int64_t)1 << 40))   // launch grids are 32-bit (F * tiles workgroups)
    return fail(CC_ERR_BAD_ARGUMENT, "cc_intrinsics_create: problem too large (frames < 2^28, observations < 2^40)");
  if (N > 0 && (!uv || !xyz) && !cc::g_create_no_obs) return fail(CC_ERR_BAD_ARGUMENT, "uv/xyz are NULL");
  if (int rc = select_device(device)) return rc;
  cc_intrinsics* h = new cc_intrinsics();
  h->device = device; h->F = F; h->N = N;
  h->one_shot = cc::g_create_one_shot;
  h->no_obs_upload = cc::g_create_no_obs;
  h->extra_bytes = cc::g_create_extra;
  const int rc_init = cc::intr_create_impl(h, off, uv, xyz);
  if (rc_init != CC_OK) {  // release whatever was allocated before the failure
    cc_intrinsics_destroy(h);
    return rc_init;
  }
  *out = h;
  return CC_OK;
}

int cc_intrinsics_exchange_export(cc_intrinsics* h, uint8_t handle[64]) {
  using namespace cc;
  if (!h || !handle) return fail(CC_ERR_BAD_ARGUMENT, "cc_intrinsics_exchange_export: NULL argument");
  CC_HIP(hipSetDevice(h->device));
  CC_HIP(hipStreamSynchronize(h->stream));
  drop_graphs(h);
  exchange_release(h);
  return mailbox_export(&h->mailbox, kVecSolve, 16, handle);
}

int cc_intrinsics_exchange_attach(cc_intrinsics* h, int32_t rank, int32_t nranks, const uint8_t* handles) {
  using namespace cc;
  if (!h || !handles || rank < 0 || nranks < 1 || rank >= nranks || nranks > kP2pMaxRanks)
    return fail(CC_ERR_BAD_ARGUMENT, "cc_intrinsics_exchange_attach: bad arguments (nranks must be 1..%d)", kP2pMaxRanks);
  if (!h->mailbox.local) return fail(CC_ERR_STATE, "cc_intrinsics_exchange_attach: call cc_intrinsics_exchange_export first");
  if (h->comm) return fail(CC_ERR_STATE, "cc_intrinsics_exchange_attach: an RCCL communicator is already attached");
  CC_HIP(hipSetDevice(h->device));
  drop_graphs(h);
  if (int rc = mailbox_attach(&h->mailbox, rank, nranks, handles, &h->d.x)) return rc;
  h->d.rank = rank;
  h->d.nranks = nranks;
  h->exchange = true;
  // Which form of the solver the ranks run must be ONE decision (the two forms add the reduced system up in different
  // orders and solve it by different code: mixed, the ranks' intrinsics would drift apart in the last bits). A rank fits
  // the persistent kernel when its workgroups are resident next to those of the ranks it shares its device with (the
  // pigeonhole bound over the devices this process sees: one rank per GPU -> 1); the ranks exchange that bit. COLLECTIVE
  // from here on: every rank must attach (10 s); a single rank skips it.
  h->persist_x_ok = false;
  if (h->persist_ok) {
    int ndev = 1;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) ndev = 1;
    const int co = getenv("CC_INTR_CO_RESIDENT") ? std::max(1, atoi(getenv("CC_INTR_CO_RESIDENT"))) : (nranks + ndev - 1) / ndev;
    h->persist_x_ok = (int64_t)(h->pq.G + 1) * co <= h->p_resident;
  }
  if (nranks > 1) {
    double* d_out = nullptr;
    int* d_ok = nullptr;
    CC_HIP(hipMalloc(&d_out, sizeof(double)));
    CC_HIP(hipMalloc(&d_ok, sizeof(int)));
    hipLaunchKernelGGL(k_intr_flag_exchange, dim3(1), dim3(64), 0, h->stream, h->d, h->persist_x_ok ? 1.0 : 0.0, d_out, d_ok);
    double sum = 0.0;
    int ok = 0;
    const hipError_t e1 = hipMemcpyAsync(&sum, d_out, sizeof(double), hipMemcpyDeviceToHost, h->stream);
    const hipError_t e2 = hipMemcpyAsync(&ok, d_ok, sizeof(int), hipMemcpyDeviceToHost, h->stream);
    const hipError_t e3 = hipStreamSynchronize(h->stream);
    hipFree(d_out);
    hipFree(d_ok);
    CC_HIP(e1); CC_HIP(e2); CC_HIP(e3);
    if (!ok) return fail(CC_ERR_COMM, "cc_intrinsics_exchange_attach: a peer rank did not attach within 10 s");
    h->persist_x_ok = sum == (double)nranks;
  }
  return CC_OK;
}

}  // extern "C"

namespace cc {
int intr_create_impl(cc_intrinsics* h, const int64_t* off, const float* uv, const float* xyz) {
  const int64_t F = h->F, N = h->N;
  if (int rc = stream_get(h->device, &h->stream)) return rc;
  IntrDev& d = h->d;
  d.F = F; d.N = N; d.rank = 0; d.nranks = 1; d.mask = 0;
  {
    // Sweep workgroups per frame. A frame is cut into tiles only when that shortens the critical path of the
    // iteration: few frames (every tile still gets a CU of its own) AND long ones (a tile keeps at least two passes
    // of 256 observations). Measured on MI355X (scripts/time_sweep_tiles.py, profiles/r02/sweep_tiles.jsonl): at
    // 125 x 500 two tiles take the sweep from 8.5 to 7.2 us but cost the elimination as much (it adds the tiles up),
    // so shards of BASELINE configs[2] stay untiled; 32 frames x 4000 points is where tiles pay.
    // CC_SWEEP_TILES overrides (tests, experiments).
    const int64_t per_frame = F > 0 ? (N + F - 1) / F : 0;
    int64_t T = std::min<int64_t>((per_frame + 2 * kSweepThreads - 1) / (2 * kSweepThreads), 256 / std::max<int64_t>(F, 1));
    if (const char* env = getenv("CC_SWEEP_TILES")) T = atoi(env);
    d.T = (int32_t)std::max<int64_t>(1, std::min<int64_t>(T, 8));
    d.tpad_ = 0;
    if (F * d.T >= (int64_t)INT32_MAX)   // the sweep's grid is F * T workgroups, a 32-bit launch dimension
      return fail(CC_ERR_BAD_ARGUMENT, "cc_intrinsics_create: %lld frames x %d sweep tiles exceed the launch grid", (long long)F, (int)d.T);
  }
  const size_t FT = (size_t)F * d.T;
  // One device arena for every buffer of the handle: a one-shot caller (Calibrator::Optimize) pays for
  // one hipMalloc / hipMemset / hipFree instead of two dozen of each (destroy: 1.3 ms -> 0.1 ms).
  // Layout: [zero-initialised state | observations], 256-byte aligned pieces.
  const size_t n1 = (size_t)std::max<int64_t>(N, 1);
  size_t cursor = 0;
  auto take = [&](size_t bytes) { const size_t at = cursor; cursor += (bytes + 255) & ~(size_t)255; return at; };
  d.log_cap = 4096;
  const size_t o_intr = take(2 * 16 * sizeof(double));
  const size_t o_pose = take((size_t)2 * F * 8 * sizeof(double));
  const size_t o_stats = take(FT * kStatsCols * sizeof(double));
  const size_t o_hd0 = take(FT * 16 * sizeof(double));
  const size_t o_sp = take((size_t)F * 8 * sizeof(double));
  const size_t o_Y = take((size_t)F * kYStride * sizeof(double));
  const size_t o_partial = take((size_t)kElimMaxBlocks * kPartialCols * sizeof(double));
  const size_t o_vs = take(kVecSolve * sizeof(double));
  const size_t o_vd = take(16 * sizeof(double));
  const size_t o_ds = take(16 * sizeof(double));
  const size_t o_ss = take(16 * sizeof(double));
  const size_t o_ctl = take(sizeof(LmCtl));
  const size_t o_ctln = take(sizeof(LmCtl));
  const size_t o_sync = take(64);                     // arrival counter (+0), published-chunk counter (+8)
  const size_t o_opts = take(sizeof(LmOpts));
  const size_t o_iintr = take(16 * sizeof(double));
  const size_t o_ipose = take((size_t)F * 8 * sizeof(double));
  // persistent kernel: frames per worker workgroup = the fewest (1, 2, 4) that still gives every frame a team of a resident
  // workgroup -- few frames spread over the chip (a 125-frame shard: one frame per compute unit), many share them four
  // to a unit; seam mailboxes: one statistics row and one elimination row per worker workgroup, one row per leader
  int p_teams = 0;
  const bool want_persist = d.T == 1 && !(getenv("CC_INTR_PERSIST") && atoi(getenv("CC_INTR_PERSIST")) == 0);
  if (want_persist) {
    for (int t = 1; t <= kPMaxTeams && !p_teams; t *= 2) {
      int resident = 0;
      if (int rc = persist_resident_workgroups(h->device, t, &resident)) return rc;
      // (kPMaxWorkers: the control's statistics gather reads one row per four threads, gather_stats4 -- a device with more
      // than 257 compute units must not get a grid whose last rows would be dropped from the sums)
      if ((F + t - 1) / t + 1 <= resident && (F + t - 1) / t <= kPMaxWorkers) { p_teams = t; h->p_resident = resident; }
    }
    if (const char* e = getenv("CC_INTR_PERSIST_TEAMS")) {   // (A/B: force a shape that fits)
      const int t = atoi(e);
      int resident = 0;
      if ((t == 1 || t == 2 || t == 4) && !persist_resident_workgroups(h->device, t, &resident) && (F + t - 1) / t + 1 <= resident && (F + t - 1) / t <= kPMaxWorkers) { p_teams = t; h->p_resident = resident; }
    }
  }
  const int64_t PG = p_teams ? (F + p_teams - 1) / p_teams : 1;
  const size_t pgn = (size_t)PG;
  const size_t o_pbox0 = cursor;
  const size_t o_sbox = take(pgn * 2 * kPStatCols * sizeof(unsigned long long));
  const size_t o_pbox = take(pgn * 2 * kPartialCols * sizeof(unsigned long long));
  const size_t o_rbox = take(pgn * 2 * kPartialCols * sizeof(unsigned long long));
  const size_t o_lbox = take(((pgn + kPLeaderRows - 1) / kPLeaderRows) * 2 * kPartialCols * sizeof(unsigned long long));
  const size_t o_xbox = take(kPBcastWords * sizeof(unsigned long long));
  const size_t o_pfail = take(64);
  h->p_box_bytes = cursor - o_pbox0;
  const size_t zeroed = cursor;                       // everything above starts as zeros
  const size_t o_blocks = take((size_t)2 * FT * 256 * sizeof(double));
  const size_t o_log = take((size_t)d.log_cap * sizeof(cc_iteration));
  const size_t o_uv = take(n1 * 2 * sizeof(float));
  const size_t o_xyz = take(n1 * 3 * sizeof(float));
  const size_t o_off = take((size_t)(F + 1) * sizeof(int64_t));
  const size_t o_extra = take(h->extra_bytes);
  if (int rc = arena_get(h->device, cursor, &h->arena, &h->arena_cached)) return rc;
  char* base = static_cast<char*>(h->arena);
  h->extra = h->extra_bytes ? base + o_extra : nullptr;
  CC_HIP(hipMemsetAsync(base, 0, zeroed, h->stream));
  if (N > 0 && !h->no_obs_upload) {
    CC_HIP(hipMemcpyAsync(base + o_uv, uv, (size_t)N * 2 * sizeof(float), hipMemcpyHostToDevice, h->stream));
    CC_HIP(hipMemcpyAsync(base + o_xyz, xyz, (size_t)N * 3 * sizeof(float), hipMemcpyHostToDevice, h->stream));
  }
  CC_HIP(hipMemcpyAsync(base + o_off, off, (size_t)(F + 1) * sizeof(int64_t), hipMemcpyHostToDevice, h->stream));
  if (!h->one_shot) CC_HIP(hipStreamSynchronize(h->stream));            // the caller's arrays may go away after create
  d.uv = reinterpret_cast<const float*>(base + o_uv);
  d.xyz = reinterpret_cast<const float*>(base + o_xyz);
  d.off = reinterpret_cast<const int64_t*>(base + o_off);
  d.intr = reinterpret_cast<double*>(base + o_intr);
  d.pose = reinterpret_cast<double*>(base + o_pose);
  d.blocks = reinterpret_cast<double*>(base + o_blocks);
  d.stats = reinterpret_cast<double*>(base + o_stats);
  d.hd0 = reinterpret_cast<double*>(base + o_hd0);
  d.sp = reinterpret_cast<double*>(base + o_sp);
  d.Y = reinterpret_cast<double*>(base + o_Y);
  d.partial = reinterpret_cast<double*>(base + o_partial);
  d.vec_solve = reinterpret_cast<double*>(base + o_vs);
  d.vec_decide = reinterpret_cast<double*>(base + o_vd);
  d.ds = reinterpret_cast<double*>(base + o_ds);
  d.ss = reinterpret_cast<double*>(base + o_ss);
  d.ctl = reinterpret_cast<LmCtl*>(base + o_ctl);
  d.ctl_next = reinterpret_cast<LmCtl*>(base + o_ctln);
  d.arrive = reinterpret_cast<unsigned*>(base + o_sync);
  d.pub_seq = reinterpret_cast<unsigned long long*>(base + o_sync + 8);
  d.opts = reinterpret_cast<LmOpts*>(base + o_opts);
  d.log = reinterpret_cast<cc_iteration*>(base + o_log);
  h->init_intr = reinterpret_cast<double*>(base + o_iintr);
  h->init_pose = reinterpret_cast<double*>(base + o_ipose);
  d.init_intr = h->init_intr;
  d.init_pose = h->init_pose;
  static_assert(16 + sizeof(LmCtl) <= 192 && sizeof(LmCtl) <= 160 && sizeof(LmOpts) <= 160, "one cached 512-byte pinned block holds all three");
  h->pinned = pinned_block_get();
  if (!h->pinned) return fail(CC_ERR_HIP, "hipHostMalloc failed");
  h->host_pub = reinterpret_cast<volatile unsigned long long*>(h->pinned);
  h->h_ctl = reinterpret_cast<LmCtl*>(static_cast<char*>(h->pinned) + 192);
  h->h_opts = reinterpret_cast<LmOpts*>(static_cast<char*>(h->pinned) + 352);
  h->host_pub[0] = 0ull;     // a recycled block may carry an old sequence number; the device counter starts at 0
  h->pub_count = 0;
  {
    void* dev_view = nullptr;
    CC_HIP(hipHostGetDevicePointer(&dev_view, h->pinned, 0));
    d.host_pub = reinterpret_cast<unsigned long long*>(dev_view);
  }
  h->pq.sbox = reinterpret_cast<unsigned long long*>(base + o_sbox);
  h->pq.pbox = reinterpret_cast<unsigned long long*>(base + o_pbox);
  h->pq.rbox = reinterpret_cast<unsigned long long*>(base + o_rbox);
  h->pq.xbox = reinterpret_cast<unsigned long long*>(base + o_xbox);
  h->pq.fail = reinterpret_cast<unsigned*>(base + o_pfail);
  h->pq.lbox = reinterpret_cast<unsigned long long*>(base + o_lbox);
  h->pq.G = (int32_t)PG;
  h->pq.teams = p_teams;
  h->p_epoch = 0;
  // every workgroup of the persistent launch waits for the others inside the kernel: it is only used when all of them
  // (workers + control) are resident at once, alone on the device (shards that share a device keep the two-kernel path)
  h->persist_ok = p_teams != 0;
  h->elim_blocks = (int)std::min<int64_t>(kElimMaxBlocks, (F + 15) / 16);
  CC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_intr_sweep),
                             hipFuncAttributeMaxDynamicSharedMemorySize, kSweepLdsBytes));
  return CC_OK;
}
}  // namespace cc

extern "C" {

void cc_intrinsics_destroy(cc_intrinsics* h) {
  if (h) cc::last_call_status_record(h->ran_form >= 0 ? h->ran_form : cc_intrinsics_solver_form(h), h->form_reruns, h->form_note);   // (what a one-shot call's caller can still ask for)
  if (!h) return;
  hipSetDevice(h->device);
  bool stream_ok = true;
  if (h->stream) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    stream_ok = hipStreamSynchronize(h->stream) == hipSuccess &&
                hipStreamIsCapturing(h->stream, &cap) == hipSuccess && cap == hipStreamCaptureStatusNone;
    (void)hipGetLastError();
  }
  cc::drop_graphs(h);
  for (auto e : h->events) hipEventDestroy(e);
  if (h->comm) cc::comm_destroy(h->comm);
  cc::exchange_release(h);
  if (h->arena) cc::arena_put(h->device, h->arena, h->arena_cached);   // (the stream was synchronised above: nothing of this handle is in flight)
  cc::pinned_block_put(h->pinned);
  if (stream_ok) cc::stream_put(h->device, h->stream);   // idle and reusable
  else if (h->stream) hipStreamDestroy(h->stream);      // never hand a failed / capturing stream to the next handle
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
  if (h->d.mask != (mask & 0x1ffu)) drop_graphs(h);  // kernel arguments are baked into the graphs
  h->d.mask = mask & 0x1ffu;
  h->have_state = true;
  return cc_intrinsics_reset(h);
}

int cc_intrinsics_reset(cc_intrinsics* h) {
  using namespace cc;
  if (!h || !h->have_state) return fail(CC_ERR_STATE, "cc_intrinsics_reset: no state set");
  // lazy: the restore kernel is the first node of the next solve's graph (or is launched by whoever reads the
  // device state first), so a restart costs neither a launch of its own nor a host round trip
  h->ctl_fresh = true;
  h->reset_pending = true;
  return CC_OK;
}

int cc_intrinsics_get_state(cc_intrinsics* h, double* intr9, double* q, double* t) {
  using namespace cc;
  if (!h) return fail(CC_ERR_BAD_ARGUMENT, "cc_intrinsics_get_state: NULL handle");
  CC_HIP(hipSetDevice(h->device));
  LmCtl c;
  if (int rc = read_ctl(h, &c)) return rc;
  const int cur = c.cur & 1;
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
  // evaluate at the accepted point without disturbing it: a phase-0 sweep writes into buffer `cur`
  LmCtl st;
  if (int rc = read_ctl(h, &st)) return rc;
  h->ctl_fresh = false;
  LmCtl ev{};
  ev.cur = st.cur & 1;
  CC_HIP(hipMemcpyAsync(h->d.ctl, &ev, sizeof(ev), hipMemcpyHostToDevice, h->stream));
  launch_sweep(h, false);
  CC_HIP(hipGetLastError());
  CC_HIP(hipMemcpyAsync(h->d.ctl, &st, sizeof(st), hipMemcpyHostToDevice, h->stream));
  CC_HIP(hipStreamSynchronize(h->stream));
  const int cur = st.cur & 1;
  const int64_t T = h->d.T, FT = h->F * T;
  if (blocks) {
    if (T == 1) {
      CC_HIP(hipMemcpy(blocks, h->d.blocks + (size_t)cur * h->F * 256, (size_t)h->F * 256 * sizeof(double), hipMemcpyDeviceToHost));
    } else {   // a frame's block is the sum of its tiles
      std::vector<double> tiles((size_t)FT * 256);
      CC_HIP(hipMemcpy(tiles.data(), h->d.blocks + (size_t)cur * FT * 256, tiles.size() * sizeof(double), hipMemcpyDeviceToHost));
      for (int64_t f = 0; f < h->F; ++f)
        for (int i = 0; i < 256; ++i) {
          double a = tiles[(size_t)(f * T) * 256 + i];
          for (int64_t k = 1; k < T; ++k) a += tiles[(size_t)(f * T + k) * 256 + i];
          blocks[(size_t)f * 256 + i] = a;
        }
    }
  }
  if (cost) {
    std::vector<double> stats((size_t)FT * kStatsCols);
    CC_HIP(hipMemcpy(stats.data(), h->d.stats, stats.size() * sizeof(double), hipMemcpyDeviceToHost));
    double c = 0.0;
    for (int64_t f = 0; f < FT; ++f) c += stats[f * kStatsCols + ST_COST];
    *cost = c;
  }
  return CC_OK;
}

}  // extern "C"

namespace cc {
// Captures `rounds` rounds (the first one optionally as the restart round) into an executable graph. On any failure the
// stream is taken out of capture mode again and nothing is kept (a stream left capturing would poison the
// process-wide stream cache it returns to).
static int capture_chunk(cc_intrinsics* h, bool with_reset, bool initial, int rounds, hipGraphExec_t* out) {
  hipGraph_t g = nullptr;
  CC_HIP(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
  int rc = 0;
  for (int i = 0; i < rounds && !rc; ++i) rc = enqueue_round(h, false, initial && i == 0, i == rounds - 1, with_reset && i == 0);
  const hipError_t e_end = hipStreamEndCapture(h->stream, &g);
  if (rc || e_end != hipSuccess) {
    if (g) hipGraphDestroy(g);
    (void)hipGetLastError();
    return rc ? rc : fail(CC_ERR_HIP, "hipStreamEndCapture failed: %s", hipGetErrorString(e_end));
  }
  const hipError_t e_inst = hipGraphInstantiate(out, g, nullptr, nullptr, 0);
  hipGraphDestroy(g);
  if (e_inst != hipSuccess) { *out = nullptr; return fail(CC_ERR_HIP, "hipGraphInstantiate failed: %s", hipGetErrorString(e_inst)); }
  return 0;
}
}  // namespace cc

namespace cc {
// A solve in phases, so that ONE host thread can drive several handles (devices) in lock step:
// begin (state, options) -> { launch a chunk on every handle -> wait for every handle } ... -> finish.
struct SolveRun {
  cc_options o;
  bool profile = false, use_graph = false, host_word = false;
  int launched = 0;
  LmCtl st{};
  std::chrono::steady_clock::time_point t0;
};

static int solve_begin(cc_intrinsics* h, const cc_options* opt, SolveRun* r) {
  r->t0 = std::chrono::steady_clock::now();
  if (opt) r->o = *opt; else cc_options_init(&r->o);
  cc_options& o = r->o;
  if (o.check_interval < 1) o.check_interval = 1;
  if (o.max_iterations > h->d.log_cap - 1) o.max_iterations = h->d.log_cap - 1;
  r->profile = o.profile_kernels != 0;
  r->use_graph = o.use_graph && !r->profile && !h->comm;
  r->host_word = !h->comm;   // fused routes hand the control block over through pinned memory
  r->launched = 0;
  CC_HIP(hipSetDevice(h->device));
  if (!h->ctl_fresh) {
    // continue from the accepted point of the previous run: move it to buffer 0, fresh control block
    LmCtl st;
    if (int rc = read_ctl(h, &st)) return rc;
    if (st.cur & 1) {
      CC_HIP(hipMemcpyAsync(h->d.intr, h->d.intr + 16, 16 * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
      CC_HIP(hipMemcpyAsync(h->d.pose, h->d.pose + (size_t)h->F * 8, (size_t)h->F * 8 * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    }
    LmCtl c{};
    if (int rc = write_ctl(h, c)) return rc;
    CC_HIP(hipMemsetAsync(h->d.arrive, 0, sizeof(unsigned), h->stream));   // (a failed solve may have left arrivals behind)
  }
  h->ctl_fresh = false;
  LmOpts lo;
  opts_from_public(o, &lo);
  if (!h->opts_valid || std::memcmp(&lo, &h->cached_opts, sizeof(lo)) != 0) {
    CC_HIP(hipStreamSynchronize(h->stream));  // the pinned staging buffer may still be in flight
    *h->h_opts = lo;
    CC_HIP(hipMemcpyAsync(h->d.opts, h->h_opts, sizeof(lo), hipMemcpyHostToDevice, h->stream));
    h->cached_opts = lo;
    h->opts_valid = true;
  }
  for (auto e : h->events) hipEventDestroy(e);
  h->events.clear();
  h->event_kind.clear();
  h->event_round.clear();
  h->enq_round = 0;
  if (r->use_graph && h->graph_iters != o.check_interval) { drop_graphs(h); h->graph_iters = o.check_interval; }
  return 0;
}

// The first chunk holds the (restart and the) initial evaluation plus check_interval iterations.
static int solve_launch(cc_intrinsics* h, SolveRun* r, int chunk) {
  const cc_options& o = r->o;
  CC_HIP(hipSetDevice(h->device));
  const int n = o.check_interval + (chunk == 0 ? 1 : 0);
  if (r->use_graph) {
    const int which = chunk > 0 ? 1 : (h->reset_pending ? 0 : 2);   // 0: restart + initial + n, 1: n, 2: initial + n
    if (!h->graph[which]) {
      const int rounds = o.check_interval + (which == 1 ? 0 : 1);
      if (int rc = capture_chunk(h, which == 0, which != 1, rounds, &h->graph[which])) { drop_graphs(h); return rc; }
    }
    CC_HIP(hipGraphLaunch(h->graph[which], h->stream));
    h->reset_pending = false;
  } else {
    const bool restart = chunk == 0 && h->reset_pending && !h->comm;
    if (restart) h->reset_pending = false;
    else if (int rc = flush_reset(h)) return rc;
    for (int i = 0; i < n; ++i)
      if (int rc = enqueue_round(h, r->profile, chunk == 0 && i == 0, i == n - 1, restart && i == 0)) return rc;
    CC_HIP(hipGetLastError());
  }
  r->launched += n;
  return 0;
}

static int solve_wait(cc_intrinsics* h, SolveRun* r) {
  CC_HIP(hipSetDevice(h->device));
  if (int rc = r->host_word ? wait_published(h, &r->st) : read_ctl(h, &r->st)) return rc;
  if (r->st.done && r->st.term == CC_FAILURE_EXCHANGE) {
    (void)hipStreamSynchronize(h->stream);
    const std::string where = h->exchange ? mailbox_describe(&h->mailbox, h->d.rank, h->d.nranks) : std::string();
    return fail(CC_ERR_COMM, "mailbox exchange timed out: a peer rank did not post within 10 s (iteration %d). %s", r->st.iter, where.c_str());
  }
  if (!r->st.done && r->launched > r->o.max_iterations + 2 * r->o.check_interval + 2)
    return fail(CC_ERR_STATE, "LM loop did not terminate (iter=%d)", r->st.iter);
  return 0;
}

// The whole solve as one launch of the persistent kernel. Starts from buffer 0 (solve_begin moved a continued solve's
// accepted point there) or, after set_state / reset, from the initial-state arrays.
static bool use_persistent(const cc_intrinsics* h, const SolveRun* r) {
  return !h->comm && !r->profile && (h->exchange ? h->persist_x_ok : h->persist_ok);
}

static int persistent_launch(cc_intrinsics* h, SolveRun* r) {
  CC_HIP(hipSetDevice(h->device));
  PersistDev q = h->pq;
  q.max_rounds = r->o.max_iterations + 2;
  q.restart = h->reset_pending ? 1 : 0;
  const uint32_t need = 3u * (uint32_t)q.max_rounds + 3u;
  if (h->p_epoch > 0xf0000000u - need) {   // before the 32-bit epochs wrap: forget every word ever stored
    CC_HIP(hipMemsetAsync(h->pq.sbox, 0, h->p_box_bytes, h->stream));
    h->p_epoch = 0;
  }
  q.epoch0 = h->p_epoch;
  h->p_epoch += need;
  q.timeout_shift = h->exchange ? 30 : 27;
  q.first_shift = h->exchange ? 30 : 20;
  // (CC_INTR_PERSIST_TEST_NO_CONTROL: test hook for the rerun in the two-kernel form, tests/test_gpu_intrinsics.py)
  static int drop_left = -2;
  const bool drop_control = !h->exchange && persist_test_drop_control("CC_INTR_PERSIST_TEST_NO_CONTROL", &drop_left);
  persist_launch(h->d, q, drop_control, h->stream);
  CC_HIP(hipGetLastError());
  h->reset_pending = false;
  r->launched = q.max_rounds;
  return 0;
}

static int persistent_wait(cc_intrinsics* h, SolveRun* r) {
  CC_HIP(hipSetDevice(h->device));
  if (int rc = wait_published(h, &r->st)) return rc;
  if (r->st.done && r->st.term == CC_FAILURE_EXCHANGE) {
    CC_HIP(hipMemsetAsync(h->pq.fail, 0, sizeof(unsigned), h->stream));
    (void)hipStreamSynchronize(h->stream);
    const std::string where = h->exchange ? mailbox_describe(&h->mailbox, h->d.rank, h->d.nranks) : std::string();
    return fail(CC_ERR_COMM, "persistent solve: a wait inside the kernel timed out (iteration %d): its %d workgroups were not all "
                "resident, or a peer rank did not post within 10 s. %s", r->st.iter, h->pq.G + 1, where.c_str());
  }
  if (!r->st.done) {
    // the control workgroup never published: did the workers give up waiting for it?
    unsigned failed = 0;
    CC_HIP(hipStreamSynchronize(h->stream));
    CC_HIP(hipMemcpy(&failed, h->pq.fail, sizeof(failed), hipMemcpyDeviceToHost));
    if (failed) {
      CC_HIP(hipMemsetAsync(h->pq.fail, 0, sizeof(unsigned), h->stream));
      return fail(CC_ERR_COMM, "persistent solve: the workers' waits timed out and the control workgroup never ran (%d workgroups not all resident)", h->pq.G + 1);
    }
    return fail(CC_ERR_STATE, "persistent solve ended without a result (iter=%d)", r->st.iter);
  }
  return 0;
}

// The whole solve as one launch of the persistent kernel. Starts from buffer 0 (solve_begin moved a continued solve's
// accepted point there) or, after set_state / reset, from the initial-state arrays.
static int solve_persistent(cc_intrinsics* h, SolveRun* r) {
  // (one persistent launch per device and process at a time: persist_mutex, cc_common.hpp; the exchanging forms -- several
  // ranks that must run together -- go through cc_intrinsics_optimize_multi's own loop or one process per rank)
  std::unique_lock<std::mutex> lk(persist_mutex(h->device), std::defer_lock);
  if (!h->exchange && !h->comm) lk.lock();
  if (int rc = persistent_launch(h, r)) return rc;
  return persistent_wait(h, r);
}

static int solve_finish(cc_intrinsics* h, SolveRun* r, cc_summary* summary) {
  const LmCtl& st = r->st;
  CC_HIP(hipSetDevice(h->device));
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
    summarise_probes(h->events, r->profile ? h->event_kind : std::vector<int>(), h->event_round, st.iter, summary,
                     [](float* ms, hipEvent_t a, hipEvent_t b) { return hipEventElapsedTime(ms, a, b) == hipSuccess; });
    summary->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - r->t0).count();
  }
  for (auto e : h->events) hipEventDestroy(e);
  h->events.clear();
  h->event_kind.clear();
  h->event_round.clear();
  return CC_OK;
}
}  // namespace cc

extern "C" {

int cc_intrinsics_solve(cc_intrinsics* h, const cc_options* opt, cc_summary* summary) {
  using namespace cc;
  if (!h || !h->have_state) return fail(CC_ERR_STATE, "cc_intrinsics_solve: no state set");
  SolveRun r;
  if (int rc = solve_begin(h, opt, &r)) return rc;
  // (alone on its device the handle also asks what the DEVICE has said about persistent solves lately -- persist_device_try,
  // cc_common.hpp: a one-shot caller's handle is new every call, so after a give-up the device's back-off window, not this
  // handle's memory, keeps the next solves on the two-kernel form; one solve probes again when the window is over. The
  // sharded forms decide collectively, once, at attach time.)
  const bool alone = !h->exchange && !h->comm;
  if (use_persistent(h, &r) && (!alone || persist_device_try(h->device, 0))) {
    // ONE launch runs the whole solve (cc_intrinsics_persist.hip); the host waits for its publication
    const bool was_restart = h->reset_pending;
    const int rc = solve_persistent(h, &r);
    if (alone) { if (rc == CC_OK) persist_device_completed(h->device, 0); else persist_device_gave_up(h->device, 0); }
    if (rc == CC_OK) { h->ran_form = h->pq.teams; return solve_finish(h, &r, summary); }
    if (rc != CC_ERR_COMM || h->exchange) return rc;
    // A wait inside the kernel gave up after 1.3 s: its workgroups were not all resident (another process on the device,
    // a compute-unit mask). Nothing was written back, so the solve is run again -- and this handle keeps to -- the
    // two-kernel form, which needs no co-residency. (Another GPU form of the same arithmetic, not a fallback off the GPU.)
    h->persist_ok = false;
    h->form_reruns++;   // (not silently: cc_intrinsics_solver_status reports the demotion and what the kernel said)
    h->form_note = last_error() + "; the solve was run again with two kernels per iteration and the handle stays on that form";
    CC_HIP(hipStreamSynchronize(h->stream));
    CC_HIP(hipMemsetAsync(h->pq.fail, 0, sizeof(unsigned), h->stream));
    if (was_restart) {
      h->reset_pending = true;
    } else {
      LmCtl zero{};
      if (int rc2 = write_ctl(h, zero)) return rc2;
      CC_HIP(hipMemsetAsync(h->d.arrive, 0, sizeof(unsigned), h->stream));
    }
    h->ctl_fresh = true;
    if (int rc2 = solve_begin(h, opt, &r)) return rc2;
  }
  for (int chunk = 0;; ++chunk) {
    if (int rc = solve_launch(h, &r, chunk)) return rc;
    if (int rc = solve_wait(h, &r)) return rc;
    if (r.st.done) break;
  }
  h->ran_form = 0;
  return solve_finish(h, &r, summary);
}

// Multi-device solve driven by ONE host thread (SURVEY.md 8(b) thread model): the frames are split into contiguous
// shards by observation count, one handle + stream per device, mailboxes wired inside the process (peer access, no
// hipIpc), every device's chunk enqueued before any of them is waited for. A device id may appear several times
// (several shards on one GPU: what the one-GPU test box does). Replaces calibrator.cpp:236-324 like
// cc_intrinsics_optimize; every shard ends with bit-identical intrinsics.
int cc_intrinsics_optimize_multi(const cc_options* opt, int32_t n_devices, const int32_t* devices, int64_t F,
                                 const int64_t* off, const float* uv, const float* xyz, double* intr9, uint32_t mask,
                                 double* q, double* t, cc_summary* summary) {
  cc::last_call_status_reset();
  using namespace cc;
  if (n_devices < 1 || !devices || n_devices > kP2pMaxRanks)
    return fail(CC_ERR_BAD_ARGUMENT, "cc_intrinsics_optimize_multi: 1..%d devices", kP2pMaxRanks);
  if (!off || F <= 0 || !intr9 || !q || !t) return fail(CC_ERR_BAD_ARGUMENT, "cc_intrinsics_optimize_multi: bad arguments");
  const int n = (int)std::min<int64_t>(n_devices, F);      // never more shards than frames
  if (n == 1) return cc_intrinsics_optimize(opt, devices[0], F, off, uv, xyz, intr9, mask, q, t, summary);
  std::vector<int64_t> first((size_t)n + 1);
  if (int rc = cc_partition_frames(F, off, n, first.data())) return rc;
  std::vector<cc_intrinsics*> hs((size_t)n, nullptr);
  auto cleanup = [&]() {
    for (auto* h : hs) if (h) { hipSetDevice(h->device); hipStreamSynchronize(h->stream); }   // nobody frees a mailbox a peer still writes
    for (auto* h : hs) cc_intrinsics_destroy(h);
  };
  int rc = 0;
  for (int r = 0; r < n && !rc; ++r) {
    const int64_t f0 = first[(size_t)r], f1 = first[(size_t)r + 1], o0 = off[f0];
    std::vector<int64_t> so((size_t)(f1 - f0) + 1);
    for (int64_t f = f0; f <= f1; ++f) so[(size_t)(f - f0)] = off[f] - o0;
    rc = cc_intrinsics_create(devices[r], f1 - f0, so.data(), uv ? uv + 2 * o0 : nullptr, xyz ? xyz + 3 * o0 : nullptr, &hs[(size_t)r]);
    if (!rc) rc = cc_intrinsics_set_state(hs[(size_t)r], intr9, mask, q + 4 * f0, t + 3 * f0);
  }
  if (!rc) {
    std::vector<Mailbox*> boxes;
    std::vector<int> devs;
    for (auto* h : hs) { boxes.push_back(&h->mailbox); devs.push_back(h->device); }
    for (int r = 0; r < n && !rc; ++r) { rc = hipSetDevice(hs[(size_t)r]->device) == hipSuccess ? mailbox_alloc(&hs[(size_t)r]->mailbox, kVecSolve, 16) : fail(CC_ERR_HIP, "hipSetDevice failed"); }
    for (int r = 0; r < n && !rc; ++r) {
      cc_intrinsics* h = hs[(size_t)r];
      rc = mailbox_wire_local(&h->mailbox, r, n, boxes.data(), devs.data(), &h->d.x);
      if (!rc) { h->d.rank = r; h->d.nranks = n; h->exchange = true; }
    }
  }
  cc_options o;
  if (opt) o = *opt; else cc_options_init(&o);
  o.use_graph = 0;   // one solve per handle: capturing and instantiating a graph cannot pay off
  std::vector<SolveRun> runs((size_t)n);
  for (int r = 0; r < n && !rc; ++r) rc = solve_begin(hs[(size_t)r], &o, &runs[(size_t)r]);
  // the persistent form, when EVERY shard fits it next to the shards that share its device (all in this process: no
  // exchange needed to agree): one launch per shard, all enqueued before any is waited for
  bool all_persist = !rc;
  for (int r = 0; r < n && all_persist; ++r) {
    cc_intrinsics* h = hs[(size_t)r];
    int co = 0;
    for (auto* g : hs) if (g->device == h->device) ++co;
    all_persist = h->persist_ok && (int64_t)(h->pq.G + 1) * co <= h->p_resident && !runs[(size_t)r].profile;
  }
  for (auto* h : hs) if (h) h->persist_x_ok = all_persist;
  if (all_persist) {
    for (int r = 0; r < n && !rc; ++r) rc = persistent_launch(hs[(size_t)r], &runs[(size_t)r]);
    for (int r = 0; r < n && !rc; ++r) rc = persistent_wait(hs[(size_t)r], &runs[(size_t)r]);
  }
  for (int chunk = 0; !rc && !all_persist; ++chunk) {
    for (int r = 0; r < n && !rc; ++r) rc = solve_launch(hs[(size_t)r], &runs[(size_t)r], chunk);
    for (int r = 0; r < n && !rc; ++r) rc = solve_wait(hs[(size_t)r], &runs[(size_t)r]);
    if (rc) break;
    bool all_done = true, any_done = false;
    for (auto& run : runs) { all_done = all_done && run.st.done; any_done = any_done || run.st.done; }
    if (all_done) break;
    if (any_done) rc = fail(CC_ERR_STATE, "cc_intrinsics_optimize_multi: the shards disagree about termination");
  }
  if (!rc) rc = solve_finish(hs[0], &runs[0], summary);
  for (int r = 0; r < n && !rc; ++r) {
    const int64_t f0 = first[(size_t)r];
    rc = cc_intrinsics_get_state(hs[(size_t)r], r == 0 ? intr9 : nullptr, q + 4 * f0, t + 3 * f0);
  }
  const std::string err = rc ? last_error() : std::string();
  cleanup();
  if (rc) last_error() = err;
  return rc;
}

// Scripts only (not in the public header): copy of a device vector. name: "vec_solve" (timing marks of timing-only builds).
int cc_intrinsics_debug_fetch(cc_intrinsics* h, const char* name, double* out, int64_t n) {
  using namespace cc;
  if (!h || !name || !out || n < 0) return fail(CC_ERR_BAD_ARGUMENT, "cc_intrinsics_debug_fetch: bad arguments");
  const std::string k(name);
  const double* src = nullptr;
  int64_t cap = 0;
  if (k == "vec_solve") { src = h->d.vec_solve; cap = kVecSolve; }
  else if (k == "stats") { src = h->d.stats; cap = h->F * h->d.T * kStatsCols; }   // (timing builds of the persistent kernel: per-workgroup marks)
  if (!src || n > cap) return fail(CC_ERR_BAD_ARGUMENT, "cc_intrinsics_debug_fetch: unknown buffer or too long");
  CC_HIP(hipSetDevice(h->device));
  CC_HIP(hipStreamSynchronize(h->stream));
  CC_HIP(hipMemcpy(out, src, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
  return CC_OK;
}

int cc_intrinsics_solver_form(cc_intrinsics* h) {
  if (!h || h->comm) return 0;
  return (h->exchange ? h->persist_x_ok : h->persist_ok) ? h->pq.teams : 0;
}

int cc_intrinsics_solver_status(cc_intrinsics* h, int32_t* form, int32_t* reruns, char* note, int32_t note_capacity) {
  using namespace cc;
  if (!h) return fail(CC_ERR_BAD_ARGUMENT, "cc_intrinsics_solver_status: NULL handle");
  if (form) *form = cc_intrinsics_solver_form(h);
  if (reruns) *reruns = h->form_reruns;
  if (note && note_capacity > 0) std::snprintf(note, (size_t)note_capacity, "%s", h->form_note.c_str());
  return CC_OK;
}

int cc_intrinsics_profile_solve(cc_intrinsics* h, const cc_options* opt, int32_t n, double* avg_launch_ms, int32_t* sweeps_per_launch) {
  using namespace cc;
  if (!h || n < 1 || !avg_launch_ms) return fail(CC_ERR_BAD_ARGUMENT, "cc_intrinsics_profile_solve: bad arguments");
  if (!h->have_state) return fail(CC_ERR_STATE, "cc_intrinsics_profile_solve: no state set");
  if (!h->persist_ok || h->comm || h->exchange) return fail(CC_ERR_STATE, "cc_intrinsics_profile_solve: the handle does not use the persistent form on a device of its own");
  CC_HIP(hipSetDevice(h->device));
  hipEvent_t e0, e1;
  CC_HIP(hipEventCreate(&e0));
  CC_HIP(hipEventCreate(&e1));
  double total = 0.0;
  int sweeps = 0, rc = 0;
  for (int i = 0; i <= n && !rc; ++i) {   // (the first one warms up and is not counted)
    SolveRun r;
    if ((rc = cc_intrinsics_reset(h))) break;
    if ((rc = solve_begin(h, opt, &r))) break;
    if (r.profile) { rc = fail(CC_ERR_BAD_ARGUMENT, "cc_intrinsics_profile_solve: profile_kernels selects the two-kernel form"); break; }
    hipEventRecord(e0, h->stream);
    if ((rc = persistent_launch(h, &r))) break;
    hipEventRecord(e1, h->stream);
    if ((rc = persistent_wait(h, &r))) break;
    if (hipEventSynchronize(e1) != hipSuccess) { rc = fail(CC_ERR_HIP, "hipEventSynchronize failed"); break; }
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, e0, e1) != hipSuccess) { rc = fail(CC_ERR_HIP, "hipEventElapsedTime failed"); break; }
    if (i > 0) total += ms;
    sweeps = r.st.sweeps;
  }
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  if (rc) return rc;
  *avg_launch_ms = total / n;
  if (sweeps_per_launch) *sweeps_per_launch = sweeps;
  return CC_OK;
}

int cc_intrinsics_profile_sweep(cc_intrinsics* h, int32_t n, double* avg_ms) {
  using namespace cc;
  if (!h || n < 1 || !avg_ms) return fail(CC_ERR_BAD_ARGUMENT, "cc_intrinsics_profile_sweep: bad arguments");
  CC_HIP(hipSetDevice(h->device));
  LmCtl st;
  if (int rc = read_ctl(h, &st)) return rc;
  if (st.phase != 1) return fail(CC_ERR_STATE, "cc_intrinsics_profile_sweep: run cc_intrinsics_solve first");
  LmCtl run = st;
  run.done = 0; run.step_valid = 1;
  CC_HIP(hipMemcpyAsync(h->d.ctl, &run, sizeof(run), hipMemcpyHostToDevice, h->stream));
  hipEvent_t e0, e1;
  CC_HIP(hipEventCreate(&e0));
  CC_HIP(hipEventCreate(&e1));
  launch_sweep(h, false);  // warm
  CC_HIP(hipEventRecord(e0, h->stream));
  for (int i = 0; i < n; ++i) launch_sweep(h, false);
  CC_HIP(hipEventRecord(e1, h->stream));
  CC_HIP(hipGetLastError());
  CC_HIP(hipStreamSynchronize(h->stream));
  float ms = 0.f;
  CC_HIP(hipEventElapsedTime(&ms, e0, e1));
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  *avg_ms = (double)ms / n;
  CC_HIP(hipMemcpy(h->d.ctl, &st, sizeof(st), hipMemcpyHostToDevice));
  return CC_OK;
}

// developer aid (not declared in cc_solver.h): time the decide + elim + solve kernel in isolation, replayed
// n times from the state the last solve left behind (which must be 1; the solve step no longer has a kernel
// of its own on a single GPU).
int cc_intrinsics_profile_kernel(cc_intrinsics* h, int32_t which, int32_t n, double* avg_ms) {
  using namespace cc;
  if (!h || n < 1 || !avg_ms || which != 1) return fail(CC_ERR_BAD_ARGUMENT, "cc_intrinsics_profile_kernel: bad arguments");
  if (h->comm || h->exchange) return fail(CC_ERR_STATE, "cc_intrinsics_profile_kernel: single-GPU handles only");
  CC_HIP(hipSetDevice(h->device));
  LmCtl st;
  if (int rc = read_ctl(h, &st)) return rc;
  if (st.phase != 1) return fail(CC_ERR_STATE, "cc_intrinsics_profile_kernel: run cc_intrinsics_solve first");
  LmCtl run = st;
  run.done = 0; run.step_valid = 1; run.cand_pending = 1; run.log_len = 1; run.iter = 1;
  run.gmax = 1e30;
  // keep every launch on the full path: tolerances off, the candidate looks like a clear improvement
  run.x_cost = run.current_cost = run.reference_cost = run.candidate_cost = run.minimum_cost = 2.0 * st.x_cost + 1.0;
  LmOpts lo = h->cached_opts;
  lo.function_tolerance = lo.parameter_tolerance = lo.gradient_tolerance = -1.0;
  lo.max_iterations = 1 << 30;
  CC_HIP(hipMemcpy(h->d.opts, &lo, sizeof(lo), hipMemcpyHostToDevice));
  h->opts_valid = false;  // the next solve uploads its own again
  CC_HIP(hipMemcpyAsync(h->d.ctl, &run, sizeof(run), hipMemcpyHostToDevice, h->stream));
  CC_HIP(hipMemcpyAsync(h->d.ctl_next, &run, sizeof(run), hipMemcpyHostToDevice, h->stream));
  hipEvent_t e0, e1;
  CC_HIP(hipEventCreate(&e0));
  CC_HIP(hipEventCreate(&e1));
  auto launch = [&]() { hipLaunchKernelGGL(k_intr_decide_elim<0>, dim3(h->elim_blocks), dim3(256), 0, h->stream, h->d, 2); };
  launch();
  CC_HIP(hipEventRecord(e0, h->stream));
  for (int i = 0; i < n; ++i) launch();
  CC_HIP(hipEventRecord(e1, h->stream));
  CC_HIP(hipGetLastError());
  CC_HIP(hipStreamSynchronize(h->stream));
  float ms = 0.f;
  CC_HIP(hipEventElapsedTime(&ms, e0, e1));
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  *avg_ms = (double)ms / n;
  CC_HIP(hipMemcpy(h->d.ctl, &st, sizeof(st), hipMemcpyHostToDevice));
  CC_HIP(hipMemcpy(h->d.ctl_next, &st, sizeof(st), hipMemcpyHostToDevice));
  return CC_OK;
}

int cc_intrinsics_optimize(const cc_options* opt, int32_t device, int64_t F, const int64_t* off,
                           const float* uv, const float* xyz, double* intr9, uint32_t mask,
                           double* q, double* t, cc_summary* summary) {
  cc::last_call_status_reset();
  cc_intrinsics* h = nullptr;
  double* tm = cc::last_timing();
  for (int i = 0; i < 5; ++i) tm[i] = 0.0;
  auto t0 = std::chrono::steady_clock::now();
  int rc;
  { cc::OneShotCreate os(0); rc = cc_intrinsics_create(device, F, off, uv, xyz, &h); }
  if (rc) return rc;
  tm[0] = cc::ms_since(t0); t0 = std::chrono::steady_clock::now();
  if (hipStreamSynchronize(h->stream) != hipSuccess) { (void)hipGetLastError(); rc = cc::fail(CC_ERR_HIP, "upload failed"); }   // (uv / xyz are the caller's)
  tm[1] = cc::ms_since(t0); t0 = std::chrono::steady_clock::now();
  if (!rc) rc = cc_intrinsics_set_state(h, intr9, mask, q, t);
  cc_options o;
  if (opt) o = *opt; else cc_options_init(&o);
  o.use_graph = 0;   // one solve per handle: capturing and instantiating a graph cannot pay off
  if (!rc) rc = cc_intrinsics_solve(h, &o, summary);
  tm[3] = cc::ms_since(t0); t0 = std::chrono::steady_clock::now();
  if (!rc) rc = cc_intrinsics_get_state(h, intr9, q, t);
  cc_intrinsics_destroy(h);
  tm[4] = cc::ms_since(t0);
  return rc;
}

// Calibrator::Estimate in ONE call: the observations are uploaded once, Zhang's closed-form initialisation
// (calibrator.cpp:47-66) runs on the handle's own device arrays, its K and poses -- rounded to float exactly as the
// two-step path cc_zhang_init -> cc_intrinsics_optimize hands them over -- start the solve. Same results as the two
// calls, one 20-byte-per-observation upload, one allocation and one pair of host packing loops less.
namespace cc {
// scratch of the Zhang initialisation, appended to the handle's arena: gram double[F][256] | H float[9F] | K float[9] | q float[4F] | t float[3F]
struct ZhangScratch {
  size_t o_gram, o_H, o_K, o_q, o_t, bytes;
  explicit ZhangScratch(int64_t F) {
    size_t cursor = 0;
    auto take = [&](size_t b) { const size_t at = cursor; cursor += (b + 255) & ~(size_t)255; return at; };
    o_gram = take((size_t)F * 256 * sizeof(double)); o_H = take((size_t)F * 9 * sizeof(float)); o_K = take(9 * sizeof(float));
    o_q = take((size_t)F * 4 * sizeof(float)); o_t = take((size_t)F * 3 * sizeof(float));
    bytes = cursor;
  }
};
// Everything of cc_intrinsics_estimate behind the handle and its upload (enqueued on h->stream, not waited for): Zhang on the
// device arrays, its K and poses rounded to float as the two-call path hands them over, the solve, the read-back.
static int estimate_on_handle(cc_intrinsics* h, const ZhangScratch& zs, const cc_options* opt, int64_t F, const double* distortion5, uint32_t mask,
                              float* K_init9, double* intr9, double* q, double* t, cc_summary* summary, std::chrono::steady_clock::time_point t0) {
  double* tm = last_timing();
  char* sc = h->extra;
  float* dK = reinterpret_cast<float*>(sc + zs.o_K);
  float* dq = reinterpret_cast<float*>(sc + zs.o_q);
  float* dt = reinterpret_cast<float*>(sc + zs.o_t);
  // (the initialisation's kernels are enqueued behind the upload; the wait below covers both -- the split of the two in
  // cc_last_call_timing comes from an event between them)
  hipEvent_t ev_up = nullptr;
  if (hipEventCreate(&ev_up) == hipSuccess) (void)hipEventRecord(ev_up, h->stream); else { ev_up = nullptr; (void)hipGetLastError(); }
  int rc;
  if ((rc = zhang_on_device(h->stream, F, h->d.off, h->d.uv, h->d.xyz, reinterpret_cast<double*>(sc + zs.o_gram),
                            reinterpret_cast<float*>(sc + zs.o_H), dK, dq, dt))) {
    if (ev_up) hipEventDestroy(ev_up);
    return rc;
  }
  float K9[9];
  std::vector<float> qf((size_t)F * 4), tf((size_t)F * 3);
  CC_HIP(hipMemcpyAsync(K9, dK, sizeof(K9), hipMemcpyDeviceToHost, h->stream));
  CC_HIP(hipMemcpyAsync(qf.data(), dq, qf.size() * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  CC_HIP(hipMemcpyAsync(tf.data(), dt, tf.size() * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  if (ev_up) {
    (void)hipEventSynchronize(ev_up);
    tm[1] += ms_since(t0); t0 = std::chrono::steady_clock::now();
    hipEventDestroy(ev_up);
  }
  CC_HIP(hipStreamSynchronize(h->stream));
  tm[2] = ms_since(t0); t0 = std::chrono::steady_clock::now();
  if (K_init9) std::memcpy(K_init9, K9, sizeof(K9));
  // intrinsics order fx fy px py k1 k2 p1 p2 k3 (calibrator.cpp:168-179); the distortion starts from the caller's
  intr9[0] = K9[0]; intr9[1] = K9[4]; intr9[2] = K9[2]; intr9[3] = K9[5];
  for (int i = 0; i < 5; ++i) intr9[4 + i] = distortion5 ? distortion5[i] : 0.0;
  for (size_t i = 0; i < qf.size(); ++i) q[i] = qf[i];
  for (size_t i = 0; i < tf.size(); ++i) t[i] = tf[i];
  if ((rc = cc_intrinsics_set_state(h, intr9, mask, q, t))) return rc;
  cc_options o;
  if (opt) o = *opt; else cc_options_init(&o);
  o.use_graph = 0;   // one solve per handle: capturing and instantiating a graph cannot pay off
  if ((rc = cc_intrinsics_solve(h, &o, summary))) return rc;
  tm[3] = ms_since(t0); t0 = std::chrono::steady_clock::now();
  rc = cc_intrinsics_get_state(h, intr9, q, t);
  tm[4] = ms_since(t0);   // (+ the handle's teardown, a few microseconds with the cached arena, after this function returns)
  return rc;
}
}  // namespace cc

int cc_intrinsics_estimate(const cc_options* opt, int32_t device, int64_t F, const int64_t* off, const float* uv,
                           const float* xyz, const double* distortion5, uint32_t mask, float* K_init9, double* intr9,
                           double* q, double* t, cc_summary* summary) {
  cc::last_call_status_reset();
  using namespace cc;
  if (F < 3 || !off || !uv || !xyz || !intr9 || !q || !t)
    return fail(CC_ERR_BAD_ARGUMENT, "cc_intrinsics_estimate: needs >= 3 frames and non-NULL arrays");
  for (int64_t f = 0; f < F; ++f)
    if (off[f + 1] - off[f] < 4) return fail(CC_ERR_BAD_ARGUMENT, "cc_intrinsics_estimate: frame %lld has fewer than 4 points", (long long)f);
  const ZhangScratch zs(F);
  double* tm = last_timing();
  for (int i = 0; i < 5; ++i) tm[i] = 0.0;
  auto t0 = std::chrono::steady_clock::now();
  cc_intrinsics* h = nullptr;
  int rc;
  { OneShotCreate os(zs.bytes); rc = cc_intrinsics_create(device, F, off, uv, xyz, &h); }
  if (rc) return rc;
  struct Guard { cc_intrinsics* h; ~Guard() { cc_intrinsics_destroy(h); } } guard{h};
  tm[0] = ms_since(t0); t0 = std::chrono::steady_clock::now();
  return estimate_on_handle(h, zs, opt, F, distortion5, mask, K_init9, intr9, q, t, summary, t0);
}

// The same for a caller whose views are separate arrays (the reference's vector<Points2D> / vector<Points3D> arguments,
// calibrator.cpp:47-68): view i has counts[i] points at uv_views[i] (2 floats each) and xyz_views[i] (3 floats each). The
// library packs them into its cached pinned staging block IN PIECES and uploads every piece as soon as it is packed: the
// copy of piece k over PCIe runs under the packing of piece k + 1, and the first kernels behind the last one.
namespace cc {
struct ViewsUpload {
  void* staging = nullptr;
  cc_intrinsics* h = nullptr;
  ~ViewsUpload() { if (h) cc_intrinsics_destroy(h); staging_put(staging); }   // (the handle's stream is idle by then: every path below waits on it)
  int open(const char* who, int32_t device, int64_t F, const float* const* uv_views, const float* const* xyz_views,
           const int64_t* counts, int64_t min_points, size_t extra_bytes) {
    int64_t N = 0;
    for (int64_t f = 0; f < F; ++f) {
      if (counts[f] < min_points || (counts[f] > 0 && (!uv_views[f] || !xyz_views[f])))
        return fail(CC_ERR_BAD_ARGUMENT, "%s: view %lld has %lld points (needs >= %lld, non-NULL)", who, (long long)f, (long long)counts[f], (long long)min_points);
      N += counts[f];
    }
    double* tm = last_timing();
    const auto t0 = std::chrono::steady_clock::now();
    const size_t b_off = ((size_t)(F + 1) * sizeof(int64_t) + 255) & ~(size_t)255, b_uv = ((size_t)N * 2 * sizeof(float) + 255) & ~(size_t)255;
    bool st_cached = false;
    staging = staging_get(b_off + b_uv + (size_t)N * 3 * sizeof(float) + 256, &st_cached);
    if (!staging) return fail(CC_ERR_HIP, "%s: pinned staging memory could not be allocated", who);
    char* st = static_cast<char*>(staging);
    int64_t* off = reinterpret_cast<int64_t*>(st);
    float* uv = reinterpret_cast<float*>(st + b_off);
    float* xyz = reinterpret_cast<float*>(st + b_off + b_uv);
    off[0] = 0;
    for (int64_t f = 0; f < F; ++f) off[f + 1] = off[f] + counts[f];
    int rc;
    { OneShotCreate os(extra_bytes, true); rc = cc_intrinsics_create(device, F, off, nullptr, nullptr, &h); }
    if (rc) return rc;
    tm[0] = ms_since(t0);
    const auto t1 = std::chrono::steady_clock::now();
    // THREE pieces of a large problem (none below 32k observations = 640 KB): every hipMemcpyAsync costs about 7.5 us of host
    // time and the link moves ~36 GB/s either way, so more pieces lose what the overlap gains (measured at 500k observations,
    // pack + upload: 1 piece 0.47-0.49 ms, 2: 0.39-0.40, 3: 0.38-0.39, 4: 0.45, 8: 0.51, 32: 0.92; a second stream for xyz
    // or a kernel that pulls the pinned block over PCIe itself made no difference beyond box-to-box noise)
    const int64_t piece = std::max<int64_t>(32768, N / 3 + 1);
    float* duv = const_cast<float*>(h->d.uv);
    float* dxyz = const_cast<float*>(h->d.xyz);
    int64_t f0 = 0;
    while (f0 < F) {
      int64_t f1 = f0;
      while (f1 < F && off[f1] - off[f0] < piece) ++f1;
      for (int64_t f = f0; f < f1; ++f) {
        if (!counts[f]) continue;
        std::memcpy(uv + 2 * off[f], uv_views[f], (size_t)counts[f] * 2 * sizeof(float));
        std::memcpy(xyz + 3 * off[f], xyz_views[f], (size_t)counts[f] * 3 * sizeof(float));
      }
      const size_t n = (size_t)(off[f1] - off[f0]);
      if (n) {
        CC_HIP(hipMemcpyAsync(duv + 2 * off[f0], uv + 2 * off[f0], n * 2 * sizeof(float), hipMemcpyHostToDevice, h->stream));
        CC_HIP(hipMemcpyAsync(dxyz + 3 * off[f0], xyz + 3 * off[f0], n * 3 * sizeof(float), hipMemcpyHostToDevice, h->stream));
      }
      f0 = f1;
    }
    tm[1] = ms_since(t1);   // packing + enqueueing; the callers add their wait for the last piece
    return CC_OK;
  }
};
}  // namespace cc

int cc_intrinsics_estimate_views(const cc_options* opt, int32_t device, int64_t F, const float* const* uv_views,
                                 const float* const* xyz_views, const int64_t* counts, const double* distortion5, uint32_t mask,
                                 float* K_init9, double* intr9, double* q, double* t, cc_summary* summary) {
  cc::last_call_status_reset();
  using namespace cc;
  if (F < 3 || !uv_views || !xyz_views || !counts || !intr9 || !q || !t)
    return fail(CC_ERR_BAD_ARGUMENT, "cc_intrinsics_estimate_views: needs >= 3 views and non-NULL arrays");
  const ZhangScratch zs(F);
  double* tm = last_timing();
  for (int i = 0; i < 5; ++i) tm[i] = 0.0;
  ViewsUpload up;
  if (int rc = up.open("cc_intrinsics_estimate_views", device, F, uv_views, xyz_views, counts, 4, zs.bytes)) return rc;
  const auto t0 = std::chrono::steady_clock::now();
  const int rc = estimate_on_handle(up.h, zs, opt, F, distortion5, mask, K_init9, intr9, q, t, summary, t0);
  if (rc) (void)hipStreamSynchronize(up.h->stream);   // (an early return may have left copies from the staging block in flight)
  return rc;
}

// cc_intrinsics_optimize for views given as separate arrays (Calibrator::Optimize's arguments, calibrator.cpp:70-74).
int cc_intrinsics_optimize_views(const cc_options* opt, int32_t device, int64_t F, const float* const* uv_views,
                                 const float* const* xyz_views, const int64_t* counts, double* intr9, uint32_t mask,
                                 double* q, double* t, cc_summary* summary) {
  cc::last_call_status_reset();
  using namespace cc;
  if (F < 1 || !uv_views || !xyz_views || !counts || !intr9 || !q || !t)
    return fail(CC_ERR_BAD_ARGUMENT, "cc_intrinsics_optimize_views: needs >= 1 view and non-NULL arrays");
  double* tm = last_timing();
  for (int i = 0; i < 5; ++i) tm[i] = 0.0;
  ViewsUpload up;
  if (int rc = up.open("cc_intrinsics_optimize_views", device, F, uv_views, xyz_views, counts, 0, 0)) return rc;
  auto t0 = std::chrono::steady_clock::now();
  int rc = CC_OK;
  if (hipStreamSynchronize(up.h->stream) != hipSuccess) { (void)hipGetLastError(); rc = fail(CC_ERR_HIP, "upload failed"); }
  tm[1] += ms_since(t0); t0 = std::chrono::steady_clock::now();
  if (!rc) rc = cc_intrinsics_set_state(up.h, intr9, mask, q, t);
  cc_options o;
  if (opt) o = *opt; else cc_options_init(&o);
  o.use_graph = 0;
  if (!rc) rc = cc_intrinsics_solve(up.h, &o, summary);
  tm[3] = ms_since(t0); t0 = std::chrono::steady_clock::now();
  if (!rc) rc = cc_intrinsics_get_state(up.h, intr9, q, t);
  tm[4] = ms_since(t0);
  return rc;
}

int cc_intrinsics_comm_init(cc_intrinsics* h, const uint8_t id[128], int32_t rank, int32_t nranks) {
  using namespace cc;
  if (!h || !id || rank < 0 || nranks < 1 || rank >= nranks || nranks > 32)
    return fail(CC_ERR_BAD_ARGUMENT, "cc_intrinsics_comm_init: bad arguments (nranks must be 1..32)");
  CC_HIP(hipSetDevice(h->device));
  if (h->comm) { comm_destroy(h->comm); h->comm = nullptr; }
  drop_graphs(h);
  if (int rc = comm_create(id, rank, nranks, &h->comm)) return rc;
  h->d.rank = rank;
  h->d.nranks = nranks;
  return CC_OK;
}

}  // extern "C"
