// cc_intrinsics_persist.hip -- the single-camera LM solve as ONE persistent kernel per solve (gfx950).
//
// Same arithmetic as the two-kernels-per-iteration path of cc_intrinsics.hip (sweep -> decide + elim + solve; replaces
// the ceres::Solve call of Calibrator::Optimize, /root/reference/src/calibrator.cpp:314-324), restructured around what
// bounded that path: an LM iteration is a chain of short dependent steps, and every one of them re-paid a kernel
// boundary, a launch ramp and a gather of state out of HBM (profiles/r02/intr_stage_marks.jsonl).
//
// Here ONE launch runs the whole solve. Grid = G worker workgroups + 1 control workgroup, 1024 threads each, one per
// compute unit, all resident (the host checks that before choosing this path):
//   worker  : four TEAMS of four waves, one frame per team for the whole solve. The frame's pose (both buffers), its
//             Y = (H_pp + D)^-1 [H_ps g_p], and BOTH 16 x 16 Gram blocks (accepted point / candidate) stay in LDS;
//             the observations (20 B each) are the only thing re-read every round (L2 / Infinity-Cache resident).
//             Round: back-substitute the pose step, Plus, sweep (the same LDS-staged v_mfma_f64_16x16x4_f64 Gram
//             as k_intr_sweep) -> statistics row -> [seam 1] -> eliminate the pose block of the accepted point's
//             Gram block (16 lanes, 6 x 6 Cholesky in registers) -> elimination row -> [seam 2].
//   control : owns the trust-region state (LmCtl), the log and the publication to the host. Seam 1: gathers the G
//             statistics rows, takes the decision (lm_init / lm_decide), broadcasts {done, cur, radius (, Jacobi
//             scales)}. Seam 2: gathers the G elimination rows, gradient / radius tests, 9 x 9 reduced solve with
//             the matrix distributed by rows over nine lanes (v_readlane pivots), broadcasts {done, valid, step}.
// Seams are self-validating 8-byte words {epoch32 : half of a double}, stored and polled with agent-scope (sc1)
// accesses: an aligned 8-byte store is single-copy atomic, so there is no flag, no fence, no drain (MI355X guide,
// Guideline 16 R2). Sums over rows run in a fixed order: results do not depend on timing or placement. Every wait is
// bounded (10 s of the wall clock); a wait that gives up sets the failure word, everybody leaves, and the host
// returns CC_ERR_COMM.
#include "cc_intrinsics_persist.hpp"
#include "cc_persist_dev.hpp"

namespace cc {

// (Where a round's time goes is measured with a variant of this file that leaves wall-clock marks in vec_solve:
// scripts/variants/timing.patch, scripts/time_intr_persist.py.)

// LDS of a workgroup with TEAMS frames: one staging tile per wave, both Gram blocks and the scratch of every team, the
// workgroup's scratch. 157,696 B at four teams (one workgroup per compute unit), 42,496 B at one.
constexpr int persist_lds_doubles(int teams) { return teams * 4 * kStageDoublesPerWave + teams * 2 * 256 + teams * 192 + 512; }

// per-team scratch (doubles)
enum { TM_Y = 0, TM_POSE = 60, TM_SP = 76, TM_STEP = 84, TM_R = 100, TM_T = 109, TM_KC = 112, TM_STEP2 = 122, TM_XN2 = 123,
       TM_QW = 124, TM_STAT = 128 /* cost, q_model, step^2, |x|^2, then nine diagonal entries (first round) */ };
// workgroup scratch (doubles)
enum { WG_INTR = 0 /* [2][16] */, WG_X = 64 /* the control's last broadcast: flags, radius, nine doubles */, WG_DS = 66 /* ... the step */,
       WG_SS = 80 /* scales of the shared columns as the workers see them: 1 */, WG_R0 = 96 /* initial radius */, WG_OPT = 100 /* jacobi, min / max LM diagonal, max radius */,
       WG_TAB = 104 /* pj / pk byte tables */,
       // slot table of the elimination row (80 slots; built once per solve; no scale factors: the control scales the sums):
       // source entry of the Gram block, the two columns of Z whose product is subtracted (| 1 << 16: there is one)
       WG_TGI = 416 /* int[80] */, WG_TZ = 456 /* int[80] */ };

static_assert(TM_POSE - TM_Y >= 6 * 10 && TM_SP - TM_POSE >= 2 * 8 && TM_STEP - TM_SP >= 6 && TM_R - TM_STEP >= 15 && TM_T - TM_R >= 9 && TM_KC - TM_T >= 3 &&
              TM_STEP2 - TM_KC >= 9 && TM_QW - TM_XN2 >= 1 && TM_STAT - TM_QW >= 4 && TM_STAT + 4 + 9 <= 192,
              "per-team scratch: Y (6 x 10), two poses, pose scale, step (9 + 6), R, t, intrinsics, scalars, statistics row -- inside the 192 doubles persist_lds_doubles gives a team");
static_assert(WG_X - WG_INTR >= 2 * 16 && WG_DS - WG_X >= 2 && WG_SS - WG_DS >= 9 && WG_R0 - WG_SS >= 9 && WG_OPT - WG_R0 >= 1 && WG_TAB - WG_OPT >= 4 &&
              WG_TZ - WG_TGI >= 80 / 2 && WG_TZ + 80 / 2 <= 512,
              "workgroup scratch: both intrinsics, broadcast (2 + 9), scales, options, tables of the 80-slot row -- inside the 512 doubles of persist_lds_doubles");

// Control workgroup, all 1024 threads: column sums (maximum for column `maxcol`) of the G rows of a box whose words
// carry `tag`. Thread -> (column, row group): rows grp, grp + NG, ... are polled BATCH at a time (all 2 BATCH loads in
// flight, unconditional from clamped rows; BATCH = what sixteen rows need where there are never more) and added in that order; the NG group sums are then added in group order.
// out[0..ncols) valid for every thread after return. *s_ok (LDS) ends 0 when a row did not show up in time.
template <int NC, int NG, int BATCH = 8>
__device__ __forceinline__ void gather_rows(const u64* box, int G, unsigned tag, int ncols, int maxcol, double* s_part,
                                            double* out, unsigned* fail, int* s_ok, int tshift) {
  const int tid = threadIdx.x, col = tid % NC, grp = tid / NC;
  if (tid == 0) *s_ok = 1;
  __syncthreads();
  if (grp < NG && col < ncols) {
    double acc = 0.0;
    bool good = true;
    const long long t0 = wall_clock64();
    for (int r0 = grp; r0 < G && good; r0 += BATCH * NG) {
      u64 lo[BATCH], hi[BATCH];
      for (unsigned spins = 0;; ++spins) {
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
          const int r = r0 + u * NG;
          const u64* p = box + (size_t)(r < G ? r : r0) * (2 * NC) + 2 * col;
          lo[u] = ag_ld(p);
          hi[u] = ag_ld(p + 1);
        }
        bool ok = true;
#pragma unroll
        for (int u = 0; u < BATCH; ++u) ok = ok && (unsigned)(lo[u] >> 32) == tag && (unsigned)(hi[u] >> 32) == tag;
        if (ok) break;
        if ((spins & 63u) == 63u && (timed_out(t0, tshift) || ag_ld32(fail) != 0u)) { good = false; break; }
        __builtin_amdgcn_s_sleep(1);
      }
      if (!good) break;
#pragma unroll
      for (int u = 0; u < BATCH; ++u) {
        if (r0 + u * NG < G) {
          const double v = ungranule(lo[u], hi[u]);
          acc = col == maxcol ? fmax(acc, v) : acc + v;
        }
      }
    }
    s_part[grp * NC + col] = acc;
    if (!good) *s_ok = 0;
  }
  __syncthreads();
  if (tid < ncols) {
    double a = 0.0;
    if (tid == maxcol) { for (int g = 0; g < NG; ++g) a = fmax(a, s_part[g * NC + tid]); }
    else { for (int g = 0; g < NG; ++g) a += s_part[g * NC + tid]; }
    out[tid] = a;
  }
  __syncthreads();
}

// Control workgroup, rounds after the first: the four column sums of the G <= 256 statistics rows. Thread -> (row, column):
// every thread polls its own two words (all rows in ONE round trip), the rows of a wave are added by the fixed
// cross-lane tree of wave_sum_mod, the waves in order.
template <int THREADS>
__device__ __forceinline__ void gather_stats4(const u64* box, int G, unsigned tag, double* s_part, double* out, unsigned* fail, int* s_ok, int tshift) {
  constexpr int RPT = 1024 / THREADS;   // rows per thread
  static_assert(THREADS / 4 * RPT == kPMaxWorkers, "one statistics row per four threads: the host caps the worker grid at kPMaxWorkers");
  const int tid = threadIdx.x, col = tid & 3, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) *s_ok = 1;
  __syncthreads();
  u64 lo[RPT], hi[RPT];
  bool good = true;
  const long long t0 = wall_clock64();
  for (unsigned spins = 0;; ++spins) {
    bool ok = true;
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      const int r = (tid >> 2) + k * (THREADS / 4);
      const u64* p = box + (size_t)(r < G ? r : 0) * (2 * kPStatCols) + 2 * col;
      lo[k] = ag_ld(p);
      hi[k] = ag_ld(p + 1);
    }
#pragma unroll
    for (int k = 0; k < RPT; ++k) ok = ok && (unsigned)(lo[k] >> 32) == tag && (unsigned)(hi[k] >> 32) == tag;   // (rows >= G read row 0)
    if (ok) break;
    if ((spins & 63u) == 63u && (timed_out(t0, tshift) || ag_ld32(fail) != 0u)) { good = false; break; }
    __builtin_amdgcn_s_sleep(1);
  }
  double acc = 0.0;
#pragma unroll
  for (int k = 0; k < RPT; ++k)
    if ((tid >> 2) + k * (THREADS / 4) < G) acc += ungranule(lo[k], hi[k]);
  acc = wave_sum_mod<2>(acc);
  if (lane < 4) s_part[wave * 4 + lane] = acc;
  if (!good) *s_ok = 0;
  __syncthreads();
  if (tid < 4) {
    double a = 0.0;
#pragma unroll
    for (int w = 0; w < THREADS / 64; ++w) a += s_part[w * 4 + tid];
    out[tid] = a;
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// control workgroup
// ---------------------------------------------------------------------------------------------
template <int THREADS>
__device__ __forceinline__ void persist_control(const IntrDev& P, const PersistDev& Q, double* lds) {
  double* s_part = lds;                    // [1024] gather scratch
  double* sv = lds + 1024;                 // [kVecSolve] reduced elimination sums (layout of cc_intrinsics.hip)
  double* s_tot = lds + 1152;              // [16] reduced statistics
  double* s_bc = lds + 1168;               // [16] broadcast payload
  double* s_intr = lds + 1184;             // [2][16] accepted / candidate intrinsics
  double* s_ss = lds + 1216;               // [16] Jacobi scales of the shared block
  LmCtl* s_ctl = reinterpret_cast<LmCtl*>(lds + 1232);         // 18 doubles
  LmOpts* s_opts = reinterpret_cast<LmOpts*>(lds + 1252);      // 12 doubles
  cc_iteration* s_log = reinterpret_cast<cc_iteration*>(lds + 1264);
  int* s_int = reinterpret_cast<int*>(lds + 1296);             // [0] gather ok, [1] a record was logged this round
  const int tid = threadIdx.x, wave = tid >> 6;
  static_assert(sizeof(cc_iteration) <= 32 * 8, "log record scratch");
  // initial state: a fresh control block (every launch is a whole solve), options, intrinsics of the starting point
  if (tid < (int)(sizeof(LmCtl) / 8)) reinterpret_cast<double*>(s_ctl)[tid] = 0.0;
  if (tid >= 32 && tid < 32 + (int)(sizeof(LmOpts) / 8)) reinterpret_cast<u64*>(s_opts)[tid - 32] = reinterpret_cast<const u64*>(P.opts)[tid - 32];
  if (tid >= 64 && tid < 64 + 32) s_intr[tid - 64] = 0.0;
  __syncthreads();
  if (tid < 9) s_intr[tid] = (Q.restart ? P.init_intr : P.intr)[tid];
  if (tid == 0) s_int[1] = 0;
  __syncthreads();
  // (a reference into LDS, not a copy: twelve doubles alive across the round loop were spilled in the 128-register build of
  // four teams, and reloaded one by one inside the reduced solve)
  const LmOpts& o = *s_opts;
  const uint32_t mask = P.mask;

  for (int round = 0; round < Q.max_rounds; ++round) {
    const unsigned e1 = Q.epoch0 + 3u * (unsigned)round + 1u, e2 = e1 + 1u, e3 = e1 + 2u;
    const bool phase0 = round == 0;
    // ---- statistics -> decision. Meanwhile the workers eliminate on the ASSUMPTION that the candidate is accepted with
    // a quality >= 0.937, where Ceres' radius update r / max(1/3, 1 - (2 rho - 1)^3) is exactly r / (1/3): the usual
    // outcome of an LM step that works. A decision that says otherwise (rejection, a mediocre step, the first round) is
    // a MISS: it is broadcast on its own and the workers eliminate again with what it says.
    if (phase0) gather_rows<kPStatCols, THREADS / kPStatCols>(Q.sbox, Q.G, e1, 13, -1, s_part, s_tot, Q.fail, s_int, (phase0 ? Q.first_shift : Q.timeout_shift));
    else gather_stats4<THREADS>(Q.sbox, Q.G, e1, s_part, s_tot, Q.fail, s_int, (phase0 ? Q.first_shift : Q.timeout_shift));
    if (P.x.on) {
      // several GPUs (mailbox exchange, cc_device.hpp): this rank's sums go into every rank's mailbox, the slots are added
      // in rank order -- the same sequence of exchanges, payloads and sums as k_intr_decide_elim<3> makes, skipped like
      // there in a round that has no candidate to judge
      const bool need = phase0 || (s_ctl->cand_pending && s_ctl->step_valid);
      if (need) {
        if (tid >= 13 && tid < 16) s_tot[tid] = 0.0;
        __syncthreads();
        const unsigned long long epoch = P.x.seq[1] + 1ull;
        p2p_post(P.x, 1, epoch, P.rank, P.nranks, s_tot, 16);
        int* s_ok2 = s_int + 4;
        const double a = p2p_collect(P.x, 1, epoch, P.rank, P.nranks, 16, s_ok2);
        if (tid < 16) s_tot[tid] = a;
        if (tid == 0) { P.x.seq[1] = epoch; if (!*s_ok2) s_int[0] = 0; }
        __syncthreads();
      }
    }
    if (tid == 0) {
      LmCtl c = *s_ctl;
      const int len0 = c.log_len, prev_cur = c.cur & 1, was_valid = c.step_valid;
      const double prev_radius = c.radius;
      const double* kc0 = s_intr + (c.cur ? 16 : 0);   // accepted intrinsics
      const double* kc1 = s_intr + (c.cur ? 0 : 16);   // candidate
      if (!s_int[0]) {
        // (a gather that gave up ends the solve AS A FAILURE for everybody: the failure word stops the workers from writing
        // their poses back and this workgroup from writing the intrinsics, so that the host's rerun in the two-kernel form
        // starts from the untouched starting point -- ADVICE round 3: only a worker's own timed-out wait used to set it)
        __hip_atomic_store(Q.fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        c.done = 1; c.term = CC_FAILURE_EXCHANGE;
      } else if (phase0) {
        double xn2 = s_tot[ST_XNORM2];
#pragma unroll
        for (int i = 0; i < 9; ++i) {
          const double ki = kc0[i];
          xn2 += ki * ki;
          s_ss[i] = o.jacobi_scaling ? 1.0 / (1.0 + sqrt(s_tot[4 + i])) : 1.0;
        }
        lm_init(c, o, s_tot[ST_COST], sqrt(xn2));
      } else if (c.cand_pending) {
        double step2 = s_tot[ST_STEP2], xn2 = s_tot[ST_XNORM2];
        if (c.step_valid) {
#pragma unroll
          for (int i = 0; i < 9; ++i) {
            const double kc = kc1[i], k0 = kc0[i];
            const double d = kc - k0;
            step2 += d * d;
            xn2 += kc * kc;
          }
        }
        lm_decide(c, o, s_log, s_tot[ST_COST], s_tot[ST_QMODEL], step2, xn2);
      }
      if (!c.done && round + 1 >= Q.max_rounds) { c.done = 1; c.term = CC_NO_CONVERGENCE; }   // (never first: lm_apply counts iterations)
      *s_ctl = c;
      s_int[1] = c.log_len != len0;
      // did the workers' assumption hold? (the same expression they evaluate: persist_spec_radius)
      s_int[2] = phase0 ? (!c.done && c.radius == o.initial_radius)
                        : (was_valid && !c.done && (c.cur & 1) == (prev_cur ^ 1) && c.radius == persist_spec_radius(prev_radius, o.max_radius));
      s_bc[0] = (double)((c.done ? 1 : 0) | ((c.cur & 1) << 3));
      s_bc[1] = c.radius;
#pragma unroll
      for (int i = 0; i < 9; ++i) s_bc[2 + i] = 0.0;
    }
    __syncthreads();
    const bool hit = s_int[2] != 0;
    if (s_ctl->done || !hit) {
      if (tid < 22) ag_st(Q.xbox + tid, granule(e2, s_bc[tid >> 1], tid & 1));
      if (s_ctl->done) {
        // a decision that ends the solve (tolerance, iteration limit) still owes its log record
        if (tid == 0 && s_int[1] && s_ctl->log_len <= P.log_cap) P.log[s_ctl->log_len - 1] = *s_log;
        break;
      }
    }
    // ---- elimination rows (of the assumption, or of the second elimination after a miss) -> gradient tests, reduced solve
    const unsigned erow = hit ? e2 : e3;
    // (the leaders among the workers have added the elimination rows sixteen at a time: one round trip here)
    gather_rows<kPartialCols, THREADS / kPartialCols, (kPLeaderRows + THREADS / kPartialCols - 1) / (THREADS / kPartialCols)>(Q.lbox, (Q.G + kPLeaderRows - 1) / kPLeaderRows, erow, kPartialCols, PC_GMAXP, s_part, sv, Q.fail, s_int, (phase0 ? Q.first_shift : Q.timeout_shift));
    if (P.x.on) {
      // all-reduce of the 112 sums through the mailboxes (kind 0): the maximum of the pose gradients rides in a slot per rank
      if (tid >= kPartialCols && tid < kVecSolve) sv[tid] = 0.0;
      __syncthreads();
      if (tid == 0) { sv[kPartialCols + P.rank] = sv[PC_GMAXP]; sv[PC_GMAXP] = 0.0; }
      __syncthreads();
      const unsigned long long epoch = P.x.seq[0] + 1ull;
      p2p_post(P.x, 0, epoch, P.rank, P.nranks, sv, kVecSolve);
      int* s_ok2 = s_int + 4;
      const double a = p2p_collect(P.x, 0, epoch, P.rank, P.nranks, kVecSolve, s_ok2);
      if (tid < kVecSolve) sv[tid] = a;
      if (tid == 0) { P.x.seq[0] = epoch; if (!*s_ok2) s_int[0] = 0; }
      __syncthreads();
      if (tid == 0) {
        double g = 0.0;
        for (int r = 0; r < P.nranks && r < 32; ++r) g = fmax(g, sv[kPartialCols + r]);
        sv[PC_GMAXP] = g;
      }
      __syncthreads();
    }
    // the workers eliminated with unit scales on the shared columns: the Jacobi scaling of the reduced system happens here
    if (tid < 63) {
      double f2;
      if (tid < 45) {
        int j = 0, rem = tid;
        while (rem >= 9 - j) { rem -= 9 - j; ++j; }
        f2 = s_ss[j] * s_ss[j + rem];
      } else if (tid < 54) {
        f2 = s_ss[tid - 45];
      } else {
        f2 = s_ss[tid - 54] * s_ss[tid - 54];
      }
      sv[tid] *= f2;
    }
    __syncthreads();
    if (wave == 0) {
      // (the lane index of THIS round, opaque to the compiler: with the plain one it computes the nine per-lane selects and unit-row
      // constants of the pinned coordinates once, before the round loop, and keeps them in registers it does not have -- the
      // 128-register build of four teams reloaded them from scratch one at a time in front of the factorisation, 0.9 us per round)
      int lane = tid & 63;
      asm volatile("" : "+v"(lane));
      const int cur = s_ctl->cur & 1;
      const double radius = s_ctl->radius;
      double gmax = sv[PC_GMAXP];   // (the maximum over the pose gradients of all ranks)
#pragma unroll
      for (int j = 0; j < 9; ++j)
        if (!(mask & (1u << j))) gmax = fmax(gmax, fabs(sv[PC_GS + j]));
      const bool go = s_int[0] && !(gmax <= o.gradient_tolerance) && !(radius < o.min_radius);
      bool ok = !(sv[PC_FAIL] > 0.0);
      double x[9];
      {
        // row `lane` of the damped reduced system (sv[0..44]: upper triangle, row-major pairs j <= k). Built and solved
        // whether or not the tests above let the solve go on: nothing else waits on this wave, and a branch on `go`
        // would put the gradient maximum in front of the factorisation.
        const int i = lane < 9 ? lane : 8;
        const bool pin_i = (mask >> i) & 1u;
        const double damp = clampd(sv[PC_HDIAG + i], o.min_lm_diagonal, o.max_lm_diagonal) / radius;   // (ONE division per lane)
        double a[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
          const int kk = k <= i ? k : i;                           // (entries beyond the diagonal: ignored, keep them finite)
          const int idx = kk * 9 - kk * (kk - 1) / 2 + (i - kk);   // pair (kk, i)
          double v = sv[idx];
          if (k == i) v += damp;
          const bool pin_k = (mask >> kk) & 1u;
          if (pin_i || pin_k) v = (k == i) ? 1.0 : 0.0;           // SubsetManifold: unit row / column, zero right-hand side
          a[k] = v;
        }
        const double b = pin_i ? 0.0 : sv[PC_B + i];
        ok = chol_solve_rows<9>(a, b, x) && ok;
#pragma unroll
        for (int j = 0; j < 9; ++j) ok = ok && isfinite(x[j]);
      }
      if (lane < 9) {
        // scaled shared step and the candidate intrinsics every worker will form from it (same expression, same bits)
        double xs = 0.0;
#pragma unroll
        for (int j = 0; j < 9; ++j) xs = lane == j ? x[j] : xs;
        const double ds = go ? -xs : 0.0;
        const double d = ((mask >> lane) & 1u) ? 0.0 : ds * s_ss[lane];   // the step in the intrinsics' own units
        s_bc[2 + lane] = d;                                                 // (what the workers' unit-scaled Y and candidates take)
        s_intr[(cur ^ 1) * 16 + lane] = s_intr[cur * 16 + lane] + d;
      }
      if (lane == 0) {
        // the flags first (they are what the broadcast waits for), the control block and the log record behind them:
        // lm_finalize below sets `done` exactly when `go` is false
        s_bc[0] = (double)((go ? 0 : 1) | (go && ok ? 2 : 0) | (hit ? 4 : 0) | (cur << 3));
        s_bc[1] = radius;
      }
    }
    __syncthreads();
    if (tid < 22) ag_st(Q.xbox + tid, granule(erow, s_bc[tid >> 1], tid & 1));
    if (tid == 0) {
      // (on the control block in LDS: only the fields lm_finalize touches move, not 144 bytes each way)
      double gmax = sv[PC_GMAXP];
#pragma unroll
      for (int j = 0; j < 9; ++j)
        if (!(mask & (1u << j))) gmax = fmax(gmax, fabs(sv[PC_GS + j]));
      const bool ok = ((int)s_bc[0] & 2) != 0;
      cc_iteration* e = s_int[1] ? s_log : nullptr;
      if (e && e->accepted) e->gradient_max_norm = gmax;
      if (!s_int[0]) { __hip_atomic_store(Q.fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); s_ctl->done = 1; s_ctl->term = CC_FAILURE_EXCHANGE; }
      else if (lm_finalize(*s_ctl, o, gmax)) { s_ctl->step_valid = ok ? 1 : 0; s_ctl->cand_pending = 1; }
      if (e && s_ctl->log_len <= P.log_cap) P.log[s_ctl->log_len - 1] = *e;
    }
    __syncthreads();
    if (s_ctl->done) break;
  }
  // ---- the solve is over: control block, intrinsics, publication
  __syncthreads();
  // (after a wait that gave up nothing is written back: buffer 0 still holds the point the solve started from, so the host
  // can run it again in the other form)
  if (tid < 32 && ag_ld32(Q.fail) == 0u) P.intr[tid] = s_intr[tid];
  if (tid == 0) {
    LmCtl c = *s_ctl;
    if (ag_ld32(Q.fail) != 0u) { c.done = 1; c.term = CC_FAILURE_EXCHANGE; }
    if (!c.done) { c.done = 1; c.term = CC_NO_CONVERGENCE; }
    *P.ctl = c;
    *P.ctl_next = c;
    publish_to_host(P, c);
  }
}

// ---------------------------------------------------------------------------------------------
// the kernel
// ---------------------------------------------------------------------------------------------
template <int TEAMS>
__global__ __launch_bounds__(TEAMS * 256, TEAMS) void k_intr_persist(IntrDev P, PersistDev Q) {
  constexpr int THREADS = TEAMS * 256;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* s_stage = reinterpret_cast<double*>(smem_raw);          // [4 TEAMS][1024] staging, one tile per wave
  if ((int)blockIdx.x == Q.G) { persist_control<THREADS>(P, Q, s_stage); return; }
  double* s_G = s_stage + TEAMS * 4 * kStageDoublesPerWave;        // [TEAMS][2][256] Gram blocks of the teams' frames
  double* s_tm = s_G + TEAMS * 2 * 256;                            // [TEAMS][192] per-team scratch
  double* s_wg = s_tm + TEAMS * 192;                               // [512] workgroup scratch
  // Register budget: four teams are 1024 threads, which leaves 128 registers per lane, and the sweep's main loop needs
  // nearly all of them. Two things keep the rest of the kernel out of its way: (1) the round loop re-derives every
  // per-thread index from a FRESH copy of the thread id (an empty asm the compiler cannot see through), so that the
  // dozens of LDS addresses and predicates that are the same every round are recomputed where they are used instead of
  // being hoisted out of the round loop and held across the main loop; (2) wave-uniform values (wave, team, frame
  // range) are read through readfirstlane, so they sit in scalar registers.
#define CC_FRESH_TID(name) int name = tid0; asm volatile("" : "+v"(name))
  const int tid0 = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid0 >> 6), team = wave >> 2, twave = wave & 3;
  // The serial pieces of a team's round (pose step, Plus, the sixteen elimination lanes) run on wave `team` of the team,
  // i.e. on waves 0, 5, 10, 15 of the workgroup: a workgroup's waves go round the four SIMDs, so waves 0, 4, 8, 12 --
  // wave 0 of every team -- share ONE SIMD and would serialise what is already the critical path.
  const int sbase = (team & 3) * 64;   // first ttid of the team's serial wave
  const int64_t f = (int64_t)blockIdx.x * TEAMS + team;
  const bool has_frame = f < P.F;
  double* sm = s_tm + team * 192;
  double* tstage = s_stage + team * 4 * kStageDoublesPerWave;      // this team's four tiles
  double* s_blk = tstage;                                          // [4][256] cross-wave reduction (after the loop)
  double* Zs = tstage + 1024;                                      // [60] elimination: L^-1 [H_ps | g_p]
  double* red = tstage + 1100;                                     // [80] elimination row of the frame
  unsigned char* pj = reinterpret_cast<unsigned char*>(s_wg + WG_TAB);
  unsigned char* pk = pj + 48;
  const uint32_t mask = P.mask;

  // ---- start of the solve: pose and intrinsics of the starting point into LDS
  {
  const int tid = tid0, ttid = tid0 & 255;
  if (has_frame && ttid < 8) {
    const double v = ttid < 7 ? (Q.restart ? P.init_pose : P.pose)[(size_t)f * 8 + ttid] : 0.0;
    sm[TM_POSE + ttid] = v;
    sm[TM_POSE + 8 + ttid] = v;
  }
  if (ttid >= 64 && ttid < 64 + 60) sm[TM_Y + ttid - 64] = 0.0;
  if (tid < 32) s_wg[WG_INTR + tid] = 0.0;
  if (tid == 32) {
    const LmOpts* o = P.opts;
    s_wg[WG_OPT] = o->jacobi_scaling ? 1.0 : 0.0;
    s_wg[WG_OPT + 1] = o->min_lm_diagonal;
    s_wg[WG_OPT + 2] = o->max_lm_diagonal;
    s_wg[WG_OPT + 3] = o->max_radius;
    s_wg[WG_R0] = o->initial_radius;
  }
  if (tid == 33) {
    int oo = 0;
    for (int j = 0; j < 9; ++j)
      for (int k = j; k < 9; ++k) { pj[oo] = (unsigned char)j; pk[oo] = (unsigned char)k; ++oo; }
  }
  if (tid >= 64 && tid < 64 + 16) s_wg[WG_X + tid - 64] = 0.0;
  // The Jacobi scales of the NINE SHARED columns never reach the workers (they are sums over all frames, known to the
  // control after the first statistics): a worker eliminates with unit scales there -- the control scales the sums it
  // gathers, and broadcasts the step multiplied by them -- so the FIRST elimination need not wait for anything either.
  if (tid >= 80 && tid < 80 + 16) s_wg[WG_SS + tid - 80] = 1.0;
  __syncthreads();
  if (tid < 9) s_wg[WG_INTR + tid] = (Q.restart ? P.init_intr : P.intr)[tid];
  if (tid >= 128 && tid < 128 + kPartialCols) {   // slot o of the elimination row (layout of k_intr_decide_elim's partial row)
    const int o = tid - 128;
    int gi = 0, z = 0;
    if (o < 45) {
      const int j = pj[o], k = pk[o];
      gi = j * 16 + k; z = j | (k << 8) | (1 << 16);
    } else if (o < 54) {
      const int j = o - 45;
      gi = j * 16 + 15; z = j | (9 << 8) | (1 << 16);
    } else if (o < 63) {
      const int j = o - 54;
      gi = j * 17;
    } else if (o >= PC_GS && o < PC_GS + 9) {
      gi = (o - PC_GS) * 16 + 15;
    }
    reinterpret_cast<int*>(s_wg + WG_TGI)[o] = gi;
    reinterpret_cast<int*>(s_wg + WG_TZ)[o] = z;
  }
  }

  int64_t s0 = 0, s1 = 0;
  if (has_frame) { s0 = P.off[f]; s1 = P.off[f + 1]; }
  const float2* uv2 = reinterpret_cast<const float2*>(P.uv);
  const int64_t wrem = s1 - s0 - twave * 64;
  const int npass = wrem > 0 ? (int)((wrem + kSweepThreads - 1) / kSweepThreads) : 0;
  const int64_t safe0 = s0 < P.N ? s0 : 0;
  // first pass of observations: in registers across the seams (the last pass of a round prefetches it again)
  float2 nm;
  float nX0, nX1, nX2;
  {
    const int64_t first = s0 + (tid0 & 255);
    const int64_t firstc = first < s1 ? first : safe0;
    nm = uv2[firstc];
    nX0 = P.xyz[firstc * 3]; nX1 = P.xyz[firstc * 3 + 1]; nX2 = P.xyz[firstc * 3 + 2];
  }
  __syncthreads();

  int cur = 0;
  double radius = 1.0;   // trust-region radius of the accepted point (known from the first broadcast on)
  bool step_valid = true, failed = false;
  for (int round = 0; round < Q.max_rounds; ++round) {
    const unsigned e1 = Q.epoch0 + 3u * (unsigned)round + 1u, e2 = e1 + 1u, e3 = e1 + 2u;
    const bool phase0 = round == 0;
    const bool do_sweep = phase0 || step_valid;
    const int dst = phase0 ? cur : (cur ^ 1);
    // =========================== sweep (candidate point, or the starting point in the first round)
    d4 acc0, acc1;
    if (do_sweep) {
      CC_FRESH_TID(tid);
      const int ttid = tid & 255, lane = tid & 63;
      const double* ds = s_wg + WG_DS;
      const double* ss = s_wg + WG_SS;
      if (has_frame) {
        const int st = ttid - sbase;   // lane of the team's serial wave
        if (st >= 0 && st < 6) {
          const double* Yr = sm + TM_Y + st * 10;
          double a = Yr[9];
#pragma unroll
          for (int j = 0; j < 9; ++j) a += Yr[j] * ds[j];
          sm[TM_STEP + 9 + st] = phase0 ? 0.0 : -a * sm[TM_SP + st];
        } else if (st >= 8 && st < 17) {
          const int j = st - 8;
          const double d = (phase0 || (mask & (1u << j))) ? 0.0 : ds[j] * ss[j];
          sm[TM_STEP + j] = d;
          const double kc = s_wg[WG_INTR + cur * 16 + j] + d;
          sm[TM_KC + j] = kc;
          if (team == 0 && !phase0) s_wg[WG_INTR + dst * 16 + j] = kc;
        }
      }
      __syncthreads();
      if (has_frame && ttid == sbase) {
        double q[4], t[3], dp[6];
#pragma unroll
        for (int i = 0; i < 4; ++i) q[i] = sm[TM_POSE + cur * 8 + i];
#pragma unroll
        for (int i = 0; i < 3; ++i) t[i] = sm[TM_POSE + cur * 8 + 4 + i];
#pragma unroll
        for (int i = 0; i < 6; ++i) dp[i] = sm[TM_STEP + 9 + i];
        double step2 = 0.0;
        if (!phase0) {
          double qn[4];
          quat_plus_tab(q, dp, qn);
#pragma unroll
          for (int i = 0; i < 4; ++i) { const double d = qn[i] - q[i]; step2 += d * d; q[i] = qn[i]; }
#pragma unroll
          for (int i = 0; i < 3; ++i) { const double tn = t[i] + dp[3 + i]; const double d = tn - t[i]; step2 += d * d; t[i] = tn; }
#pragma unroll
          for (int i = 0; i < 4; ++i) sm[TM_POSE + dst * 8 + i] = q[i];
#pragma unroll
          for (int i = 0; i < 3; ++i) sm[TM_POSE + dst * 8 + 4 + i] = t[i];
        }
        double R[9];
        quat_to_R(q, R);
#pragma unroll
        for (int i = 0; i < 9; ++i) sm[TM_R + i] = R[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) sm[TM_T + i] = t[i];
        sm[TM_STEP2] = step2;
        sm[TM_XN2] = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3] + t[0] * t[0] + t[1] * t[1] + t[2] * t[2];
      }
      __syncthreads();
      // model-cost term of the frame over its 15 x 15 block at the accepted point (cf. k_intr_sweep)
      {
        double qterm = 0.0;
        if (has_frame && !phase0) {
          const int a = ttid >> 4, b = ttid & 15;
          const double g_old = s_G[(team * 2 + cur) * 256 + ttid];
          if (a < 15) qterm = b < 15 ? 0.5 * sm[TM_STEP + a] * g_old * sm[TM_STEP + b] : sm[TM_STEP + a] * g_old;
        }
        const double qw = wave_sum(qterm);
        if (lane == 0) sm[TM_QW + twave] = qw;
      }
      double R[9], tt[3], kk[9];
#pragma unroll
      for (int i = 0; i < 9; ++i) R[i] = rfl(sm[TM_R + i]);
#pragma unroll
      for (int i = 0; i < 3; ++i) tt[i] = rfl(sm[TM_T + i]);
#pragma unroll
      for (int i = 0; i < 9; ++i) kk[i] = rfl(sm[TM_KC + i]);
      // ---- main loop: 64 observations per wave per pass, no barrier (k_intr_sweep's loop)
      double* stage = s_stage + wave * kStageDoublesPerWave;
      acc0 = d4{0.0, 0.0, 0.0, 0.0};
      acc1 = d4{0.0, 0.0, 0.0, 0.0};
      // (Waves that share a SIMD -- one per team -- enter the loop in lockstep. A start staggered by k sleep units per team, so that one
      // team's LDS staging falls under another's matrix products, was measured at several k and never paid: removed in round 6.)
      for (int p = 0; p < npass; ++p) {
        const int64_t idx = s0 + (int64_t)p * kSweepThreads + ttid;
        const bool valid = idx < s1;
        const float2 m = nm;
        const float X0 = nX0, X1 = nX1, X2 = nX2;
        {   // next pass -- after the last one, the first pass of the next round -- unconditionally
          const int64_t nidx = p + 1 < npass ? idx + kSweepThreads : s0 + ttid;
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
        gram_rows_ahead(stage, lane, acc0, acc1);
        wave_lds_fence();
        row_v(kk, oc, (double)m.y, v, wrow);
        stage_row(stage, lane, v);
        wave_lds_fence();
        gram_rows_ahead(stage, lane, acc0, acc1);
        wave_lds_fence();
      }
      // ---- cross-wave reduction of the 16 x 16 block into the frame's LDS slot
      __syncthreads();   // s_blk aliases the staging tiles
    }
    if (do_sweep) {
      CC_FRESH_TID(tid);
      const int ttid = tid & 255, lane = tid & 63;
#pragma unroll
      for (int r = 0; r < 4; ++r) s_blk[twave * 256 + ((lane >> 4) + 4 * r) * 16 + (lane & 15)] = acc0[r] + acc1[r];
      __syncthreads();
      const double g = gram_entry_held(mask, ttid) ? 0.0 : (s_blk[ttid] + s_blk[256 + ttid]) + (s_blk[512 + ttid] + s_blk[768 + ttid]);   // (constant coordinates: zero rows and columns)
      s_G[(team * 2 + dst) * 256 + ttid] = g;
      if (ttid == 255) {
        sm[TM_STAT + ST_COST] = 0.5 * g;
        sm[TM_STAT + ST_QMODEL] = (sm[TM_QW] + sm[TM_QW + 1]) + (sm[TM_QW + 2] + sm[TM_QW + 3]);
        sm[TM_STAT + ST_STEP2] = sm[TM_STEP2];
        sm[TM_STAT + ST_XNORM2] = sm[TM_XN2];
      }
      if (phase0 && ttid < 9 * 17 && ttid % 17 == 0) sm[TM_STAT + 4 + ttid / 17] = g;
      __syncthreads();
    }
    // ---- statistics row of the workgroup (teams in order) -> control
    {
      CC_FRESH_TID(tid);
      const int ncols = phase0 ? 13 : 4;
      if (tid < 2 * ncols) {
        const int c = tid >> 1;
        double a = 0.0;
        if (do_sweep) {
#pragma unroll
          for (int k = 0; k < TEAMS; ++k)
            if ((int64_t)blockIdx.x * TEAMS + k < P.F) a += s_tm[k * 192 + TM_STAT + c];
        }
        ag_st(Q.sbox + (size_t)blockIdx.x * (2 * kPStatCols) + tid, granule(e1, a, tid & 1));
      }
    }
    // =========================== elimination of the frame's pose block at buffer cur_e under radius_e, the workgroup's
    // row -> box (tag), and -- leaders -- the sum of sixteen rows -> lbox (tag)
    auto eliminate_and_post = [&](const int cur_e, const double radius_e, const bool first, u64* box, const unsigned tag) {
    CC_FRESH_TID(tid_e);
    if (has_frame && (tid_e & 255) >= sbase && (tid_e & 255) < sbase + 16) {
      const int l = (tid_e & 255) - sbase;
      const double* Gl = s_G + (team * 2 + cur_e) * 256;
      const double* ss = s_wg + WG_SS;
      const bool jac = s_wg[WG_OPT] != 0.0;
      const double mn = s_wg[WG_OPT + 1], mx = s_wg[WG_OPT + 2];
      const double inv_radius = 1.0 / radius_e;
      double s[6], L[21];
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) L[tri(i, j)] = Gl[(9 + i) * 16 + 9 + j];
      if (first) {
#pragma unroll
        for (int i = 0; i < 6; ++i) s[i] = jac ? 1.0 / (1.0 + sqrt(L[tri(i, i)])) : 1.0;
        if (l < 6) {
          double sl = 0.0;
#pragma unroll
          for (int i = 0; i < 6; ++i) sl = l == i ? s[i] : sl;
          sm[TM_SP + l] = sl;
        }
      } else {
#pragma unroll
        for (int i = 0; i < 6; ++i) s[i] = sm[TM_SP + i];
      }
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) L[tri(i, j)] = s[i] * L[tri(i, j)] * s[j];
#pragma unroll
      for (int i = 0; i < 6; ++i) L[tri(i, i)] += clampd(L[tri(i, i)], mn, mx) * inv_radius;
      bool ok = true;
      double Li[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        double d = L[tri(j, j)];
#pragma unroll
        for (int k = 0; k < j; ++k) d -= L[tri(j, k)] * L[tri(j, k)];
        ok = ok && (d > 0.0) && isfinite(d);
        const double inv = rsqrt_pos(d);   // (same bits as rsqrt for d > 0 finite; the step is discarded otherwise)
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
      // the factor is the same in all sixteen lanes: from here on it lives in scalar registers (48 of them), not in 54
      // vector registers next to each lane's own columns
#pragma unroll
      for (int i = 0; i < 21; ++i) L[i] = rfl(L[i]);
#pragma unroll
      for (int i = 0; i < 6; ++i) { Li[i] = rfl(Li[i]); s[i] = rfl(s[i]); }
      if (l < 10) {
        const double sc = l < 9 ? ss[l] : 1.0;
        const int col = l < 9 ? l : 15;
        double w[6], z[6], y[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) w[i] = Gl[(9 + i) * 16 + col];
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
          Zs[i * 10 + l] = z[i];
          sm[TM_Y + i * 10 + l] = y[i];
        }
      }
      if (l == 0) red[PC_FAIL] = ok ? 0.0 : 1.0;
    }
    // ---- the frame's elimination row: slot o of eighty on lane o mod 64 of the SAME wave (its LDS operations execute in
    // order: a fence for the compiler, no barrier) -- two slots per lane instead of the five each of the sixteen
    // elimination lanes built. Slot table: source entry of the Gram block, the two columns of Z whose product is subtracted.
    if (has_frame && (tid_e & 255) >= sbase && (tid_e & 255) < sbase + 64) {
      const int l64 = (tid_e & 255) - sbase;
      const double* Gl = s_G + (team * 2 + cur_e) * 256;
      wave_lds_fence();
      const int* tgi = reinterpret_cast<const int*>(s_wg + WG_TGI);
      const int* tz = reinterpret_cast<const int*>(s_wg + WG_TZ);
      double accv[2];
      int zz_[2];
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int o = l64 + 64 * r, oc = o < kPartialCols ? o : 0;
        zz_[r] = tz[oc];
        accv[r] = Gl[tgi[oc]];
      }
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int zj = zz_[r] & 255, zk = (zz_[r] >> 8) & 255;
        double zz = 0.0;
#pragma unroll
        for (int i = 0; i < 6; ++i) zz += Zs[i * 10 + zj] * Zs[i * 10 + zk];
        const int o = l64 + 64 * r;
        const bool used = o < 63 || (o >= PC_GS && o < PC_GS + 9);   // (63: failures, 73: gradient maximum -- written elsewhere)
        if (o < kPartialCols && o != PC_FAIL && o != PC_GMAXP) red[o] = used ? accv[r] - ((zz_[r] >> 16) ? zz : 0.0) : 0.0;
      }
    }
    // The frame's share of Ceres' gradient_max_norm, ||x - Plus(x, -g)||_inf (pose_grad_proj_max, cc_common.hpp): one lane
    // of the team's NEXT wave -- idle here, on another SIMD -- so that the forty dependent operations run beside the
    // elimination instead of behind it (on the elimination's own lane 0: 35.7 -> 37.4 us per iteration at configs[2]).
    if (has_frame && (tid_e & 255) == (((team & 3) + 1) & 3) * 64) {
      const double* Gg = s_G + (team * 2 + cur_e) * 256;
      double q4[4], g6[6];
#pragma unroll
      for (int i = 0; i < 4; ++i) q4[i] = sm[TM_POSE + cur_e * 8 + i];
#pragma unroll
      for (int i = 0; i < 6; ++i) g6[i] = Gg[(9 + i) * 16 + 15];
      red[PC_GMAXP] = pose_grad_proj_max_tab(q4, g6);
    }
    __syncthreads();
    // ---- elimination row of the workgroup (teams in order)
    {
      CC_FRESH_TID(tid);
      if (tid < 2 * kPartialCols) {
        const int c = tid >> 1;
        double a = 0.0;
#pragma unroll
        for (int k = 0; k < TEAMS; ++k) {
          if ((int64_t)blockIdx.x * TEAMS + k < P.F) {
            const double v = s_stage[k * 4 * kStageDoublesPerWave + 1100 + c];
            a = c == PC_GMAXP ? fmax(a, v) : a + v;
          }
        }
        ag_st(box + (size_t)blockIdx.x * (2 * kPartialCols) + tid, granule(tag, a, tid & 1));
      }
    }
    // ---- every sixteenth workgroup is a LEADER: it adds up the rows of its sixteen (it would only be waiting for the
    // step otherwise) and posts ONE row for the control, which then reads G / 16 rows in a single round trip instead of G
    if ((blockIdx.x % kPLeaderRows) == 0) {
      int* s_lok = reinterpret_cast<int*>(s_wg + 250);
      const int g0 = (int)blockIdx.x, n = Q.G - g0 < kPLeaderRows ? Q.G - g0 : kPLeaderRows;
      double* lout = s_wg + 128;        // [80]; the group sums go through team 0's staging tiles (idle between sweeps)
      gather_rows<kPartialCols, THREADS / kPartialCols, (kPLeaderRows + THREADS / kPartialCols - 1) / (THREADS / kPartialCols)>(box + (size_t)g0 * (2 * kPartialCols), n, tag, kPartialCols, PC_GMAXP, s_stage + 2048, lout,
                                                        Q.fail, s_lok, (phase0 ? Q.first_shift : Q.timeout_shift));
      CC_FRESH_TID(tid);
      // (rows that did not arrive: nothing is posted, the control's own wait gives up and ends the solve -- as a failure)
      if (!*s_lok && tid == 0) __hip_atomic_store(Q.fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (*s_lok && tid < 2 * kPartialCols) ag_st(Q.lbox + (size_t)(g0 / kPLeaderRows) * (2 * kPartialCols) + tid, granule(tag, lout[tid >> 1], tid & 1));
    }
    };
    // one wave waits for the control's broadcast `tag` (flags, radius, nine doubles) -> s_wg[WG_X ..]
    auto wait_bcast = [&](const unsigned tag) {
      CC_FRESH_TID(tid);
      if (wave == 0 && !bcast_wait(Q.xbox, tag, 11, s_wg + WG_X, Q.fail, tid & 63, (phase0 ? Q.first_shift : Q.timeout_shift))) s_wg[WG_X] = 17.0;   // done + failed
      __syncthreads();
    };

    // =========================== the assumed decision: candidate accepted, radius at its clamp (see the control).
    // Eliminating NOW, next to the control's gathering and deciding, takes a seam out of the round when it holds.
    // (first round: the starting point is "accepted" at the initial radius unless the solve ends before it begins; its
    // elimination also computes the frame's Jacobi scale)
    const bool spec = do_sweep;
    const double radius_spec = phase0 ? s_wg[WG_R0] : persist_spec_radius(radius, s_wg[WG_OPT + 3]);
    if (spec) eliminate_and_post(dst, radius_spec, phase0, Q.pbox, e2);
    wait_bcast(e2);
    int fl = (int)s_wg[WG_X];
    if (fl & 16) failed = true;
    if (fl & 1) { cur = (fl >> 3) & 1; break; }
    if (fl & 4) {   // the assumption held: this broadcast IS the step
      cur = dst;
      radius = radius_spec;
    } else {        // it did not (or there was none): eliminate with what the decision says, then wait for the step
      cur = (fl >> 3) & 1;
      radius = s_wg[WG_X + 1];
      eliminate_and_post(cur, radius, false, Q.rbox, e3);   // (never the first round: that one always holds or ends the solve)
      wait_bcast(e3);
      fl = (int)s_wg[WG_X];
      if (fl & 16) failed = true;
      if (fl & 1) { cur = (fl >> 3) & 1; break; }
    }
    step_valid = (fl & 2) != 0;
  }
  // ---- the solve is over: the frame's accepted pose goes back to HBM (cc_intrinsics_get_state, the next solve)
  if (!failed && ag_ld32(Q.fail) == 0u && has_frame && (tid0 & 255) < 7) P.pose[((size_t)cur * P.F + f) * 8 + (tid0 & 255)] = sm[TM_POSE + cur * 8 + (tid0 & 255)];
  asm volatile("" ::"v"(nm.x), "v"(nX0), "v"(nX1), "v"(nX2));   // (the last prefetch has no consumer)
}

template <int TEAMS>
static int resident_of(int device, int* out) {
  const int lds = persist_lds_doubles(TEAMS) * 8;
  CC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_intr_persist<TEAMS>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  int per_cu = 0, cus = 0;
  CC_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_intr_persist<TEAMS>, TEAMS * 256, lds));
  CC_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
  // one workgroup per compute unit, whatever the query admits: a workgroup's teams are meant to have a CU to themselves
  *out = per_cu >= 1 ? cus : 0;
  return 0;
}

int persist_resident_workgroups(int device, int teams, int* out) {
  return teams == 1 ? resident_of<1>(device, out) : teams == 2 ? resident_of<2>(device, out) : resident_of<4>(device, out);
}

void persist_launch(const IntrDev& P, const PersistDev& Q, bool drop_control, hipStream_t stream) {
  // (drop_control: test hook -- the grid goes out WITHOUT its control workgroup, every worker's first wait gives up)
  const dim3 grid((unsigned)Q.G + (drop_control ? 0u : 1u));
  if (Q.teams == 1) hipLaunchKernelGGL(k_intr_persist<1>, grid, dim3(256), persist_lds_doubles(1) * 8, stream, P, Q);
  else if (Q.teams == 2) hipLaunchKernelGGL(k_intr_persist<2>, grid, dim3(512), persist_lds_doubles(2) * 8, stream, P, Q);
  else hipLaunchKernelGGL(k_intr_persist<4>, grid, dim3(1024), persist_lds_doubles(4) * 8, stream, P, Q);
}

}  // namespace cc
