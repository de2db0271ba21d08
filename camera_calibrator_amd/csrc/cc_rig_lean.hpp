// cc_rig_lean.hpp -- the lean persistent form: the whole solve of a small rig in one launch (k_rig_persist_w + k_rig_persist_ctl).
// Part of cc_rig.hip (round 5: the 6.8 k-line file split by subject; included by it inside namespace cc, in this order:
// cc_rig_sweeps.hpp, cc_rig_steps.hpp, cc_rig_big.hpp, cc_rig_lean.hpp -- one translation unit, nothing else includes these).
#pragma once

// =============================================================================================
// THE RIG SOLVE AS ONE PERSISTENT KERNEL (round 3; poses only, single GPU, at most four frames per compute unit).
// TWO FORMS. This one, k_rig_persist, GLUES the three kernels' bodies together: AN EXPERIMENT, OFF BY DEFAULT (CC_RIG_PERSIST=1,
// used where the lean form below does not fit): correct -- the rig test suite passes on it -- and SLOWER than the three
// kernels it replaces: 81 against 47 us per iteration at BASELINE configs[3] size (profiles/r03/rig_persist_marks.jsonl).
// The bodies need up to 444 registers per thread, so a compute unit holds ONE wave per SIMD: the sweep of a frame's groups
// runs one after the other with every memory round trip exposed (25.6 us where the stand-alone sweep, sixteen waves deep,
// takes 9), and pose update, elimination and solve step each run 1.3 - 2 x slower for the same reason.
// The LEAN form further down (k_rig_persist_w + k_rig_persist_ctl: small rigs, the per-frame state in LDS, one wave per
// group, the workers under 128 / 256 registers) is what runs BY DEFAULT where it fits: 39 us at configs[3] size. Seams,
// control workgroup and host side are shared.
// Three launches per LM iteration cost this path ~19 of its ~47 us at BASELINE configs[3] (ramp of a launch, dependent
// read of the control block, the gap; profiles/r03/rig_c4_kernel_stats.csv): here ONE launch runs the whole solve, built
// from the very functions the three kernels run (rig_update_body, rig_sweep_adj_body, rig_elim_body, rig_solve_block,
// rig_candidates), with the seams of cc_intrinsics_persist.hip between them (cc_persist_dev.hpp: self-validating words,
// no atomics, no flags, bounded waits).
//   grid    : G = ceil(F / 4) worker workgroups + 1 control workgroup, 256 threads each, all resident (host: occupancy).
//   worker b: frames 4b .. 4b + 3, one wave each, for the whole solve. Round: [broadcast B: step + camera records] ->
//             pose update of its frames -> sweep of their groups (one wave: the groups of its frame one after the other)
//             -> statistics row -> [broadcast A: decision] -> elimination of its four frames -> partial row, compacted
//             to the K entries the reduced system uses -> posts it; then adds up ITS share of the K columns over all G
//             rows (column c belongs to worker c mod G: every worker reads G x K / G words -- the column sums of
//             k_rig_reduce, spread over the workers) and posts the sums.
//   control : owns the trust-region state. Gathers the statistics rows -> decision (first round: Jacobi scales of the
//             shared columns, |x|, lm_init -- what k_rig_init does) -> broadcast A; gathers the K column sums ->
//             reduced solve, candidates, records (rig_solve_block, unchanged) -> broadcast B.
// What a workgroup writes to global memory for its own later use (poses, frame records, group blocks, Y, partial row)
// it reads back itself: plain stores and loads on one compute unit. Sums over rows run in a fixed order.
// A wait that gives up sets the failure word (arrive[3]): nothing further happens, the host returns CC_ERR_COMM and the
// handle goes back to the three-kernel form.
// =============================================================================================
struct RigPersistDev {
  u64* sbox;            // [G][KS][2]  statistics rows: cost, model term, step^2, |x|^2, S diagonal sums (first round)
  u64* abox;            // [2 + S][2]  broadcast A: flags (1 done | cur << 3), radius, Jacobi scales of the shared columns (first round)
  u64* rbox;            // [G][K][2]   elimination rows (compacted)
  u64* cbox;            // [K][2]      column sums
  u64* pbox, *pcbox;    // the same two for the elimination on the ASSUMED decision (k_rig_persist_w; null: nobody assumes)
  u64* ybox;            // [NB][2]     broadcast B: flags (1 done | 2 step valid | cur << 3), radius, step[S], camera records [C][32]
  const int32_t* comp;  // [K] entry of the partial-row layout behind compact index k (the last two: failures, gradient maximum)
  const int32_t* slots; // [K] the same entries for k_rig_persist_w: 1 << 30 | p << 8 | q: sum_i Z[i][p] Z[i][q]; c << 16 | offset: entry of observed
                        //     camera c's block; -1: failed factorisations; -2: gradient maximum
  int32_t G, K, KS, NB;
  unsigned epoch0;      // tags: epoch0 + round + 1 (boxes are zeroed when they would wrap)
  unsigned* claim;      // [1] the control candidate that exchanges epoch0 + 1 in first is the control workgroup (k_rig_persist_ctl)
  unsigned long long* gate;   // pinned host word: the worker that finds all G workers started stores epoch0 + 1 into it and the HOST then
                              //   launches the control (rig_launch); null: no gate (the candidates run when they run)
  int32_t max_rounds, timeout_shift, first_shift;   // (first_shift: the workers' wait for the control's FIRST broadcast)
};

constexpr int kRigPersistMaxS = 48;     // shared coordinates (8 optimised cameras)
constexpr int kRigPersistMaxC = 9;      // cameras (records travel in broadcast B)
constexpr int kRigPersistMaxNB = 2 + kRigPersistMaxS + 32 * kRigPersistMaxC;   // (the control workgroup's own copies are sized for 48 coordinates; the workers take kRpwMaxS)

// One wave waits until the n doubles of a broadcast box carry `tag` and leaves them in dst[0..n) (LDS). false: gave up.
__device__ __forceinline__ bool rig_bcast_wait(const u64* box, unsigned tag, int n, double* dst, unsigned* fail, int tshift) {
  int lane = threadIdx.x & 63;
  asm volatile("" : "+v"(lane));   // (a fresh copy: the eleven word addresses of a lane are not worth keeping across a round)
  constexpr int W = (2 * kRigPersistMaxNB + 63) / 64;
  const long long t0 = wall_clock64();
  u64 v[W];
  for (unsigned spins = 0;; ++spins) {
    bool ok = true;
#pragma unroll
    for (int j = 0; j < W; ++j) {
      const int w = lane + 64 * j;
      if (64 * j < 2 * n) {   // (uniform)
        v[j] = ag_ld(box + (w < 2 * n ? w : 0));
        ok = ok && (w >= 2 * n || (unsigned)(v[j] >> 32) == tag);
      }
    }
    if (__all(ok)) break;
    if ((spins & 63u) == 63u && (timed_out(t0, tshift) || ag_ld32(fail) != 0u)) {
      if (lane == 0) __hip_atomic_store(fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return false;
    }
    __builtin_amdgcn_s_sleep(1);
  }
#pragma unroll
  for (int j = 0; j < W; ++j) {
    const int w = lane + 64 * j;
    if (w < 2 * n) reinterpret_cast<unsigned*>(dst)[w] = (unsigned)v[j];   // word 2i = low half of double i
  }
  return true;
}

// All 256 threads: thread t (< G) waits for entry `col` of row t (NC columns per row, up to NB columns in one round trip),
// then the block adds the G values in a fixed order (maximum for is_max). out[j] valid for every thread after return.
template <int NB>
__device__ __forceinline__ bool rig_gather_cols(const u64* box, int G, int rowlen, const int* cols, int ncols, int maxcol, unsigned tag, double* s4, double* out,
                                                unsigned* fail, int tshift) {
  const int tid = threadIdx.x;
  __shared__ int s_good;
  if (tid == 0) s_good = 1;
  __syncthreads();
  u64 lo[NB], hi[NB];
  bool good = true;
  if (tid < G) {
    const long long t0 = wall_clock64();
    for (unsigned spins = 0;; ++spins) {
      bool ok = true;
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int c = cols[j < ncols ? j : 0];
        const u64* p = box + ((size_t)tid * rowlen + c) * 2;
        lo[j] = ag_ld(p);
        hi[j] = ag_ld(p + 1);
        ok = ok && (j >= ncols || ((unsigned)(lo[j] >> 32) == tag && (unsigned)(hi[j] >> 32) == tag));
      }
      if (ok) break;
      if ((spins & 63u) == 63u && (timed_out(t0, tshift) || ag_ld32(fail) != 0u)) { good = false; break; }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  if (!good) s_good = 0;
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    if (j < ncols) {   // (uniform)
      const double v = (tid < G && good) ? ungranule(lo[j], hi[j]) : 0.0;
      double r;
      if (cols[j] == maxcol) {
        const double m = wave_max(v);
        __syncthreads();
        if ((tid & 63) == 0) s4[tid >> 6] = m;
        __syncthreads();
        r = fmax(fmax(s4[0], s4[1]), fmax(s4[2], s4[3]));
      } else {
        r = block_sum256(v, s4);
      }
      out[j] = r;
    }
  }
  __syncthreads();
  return s_good != 0;
}

// The control workgroup of the persistent rig kernels (256 threads; dynamic LDS: the solve step's, rig_solve_block): block G of
// k_rig_persist, or a launch of its own next to the lean workers of k_rig_persist_w (k_rig_persist_ctl).
__device__ __forceinline__ void rig_persist_control(const RigDev& P, const RigPersistDev& Q, char* smem_raw) {
  __shared__ double s_bc[kRigPersistMaxNB];     // broadcast B of this round: flags, radius, step, camera records
  __shared__ double s_a[2 + kRigPersistMaxS];   // broadcast A
  __shared__ double s_ss[kRigPersistMaxS + 1];
  __shared__ double s4[4];
  __shared__ int s_cols[16];
  __shared__ int s_flag;
  const int tid = threadIdx.x;
  const int S = P.S, C = P.C, G = Q.G, K = Q.K, KS = Q.KS, NB = Q.NB;
  unsigned* fail = P.arrive + 3;
    // =========================================================================== control workgroup
    __shared__ LmCtl s_ctl;
    __shared__ cc_iteration s_rec;
    __shared__ double s_tot[4 + kRigPersistMaxS];
    __shared__ int s_has_rec, s_hit;
    double* smem = reinterpret_cast<double*>(smem_raw);
    double* vl = smem + (size_t)S * ((S + 1) | 1) + 5 * 128;   // [PC + 32] the reduced row in the layout rig_solve_block reads, behind its own LDS
    if (tid == 0) s_ctl = *P.ctl;   // (zeros: rig_begin)
    for (int i = tid; i < P.PC + 32; i += 256) vl[i] = 0.0;   // entries the compact rows never touch stay zero
    // the solve step's destination tables, copied to LDS once: rig_solve_block walks them every round, and here nothing
    // but this workgroup's latency is on the critical path (flat loads of LDS addresses through the same RigDev fields)
    RigDev Pc = P;
    {
      int32_t* t_tile = reinterpret_cast<int32_t*>(vl + P.PC + 32);
      int32_t* t_dd = t_tile + P.nT * 256;
      int32_t* t_dn = t_dd + P.ND;
      int16_t* t_sa = reinterpret_cast<int16_t*>(t_dn + P.ND);
      int16_t* t_sb = t_sa + P.ND;
      for (int i = tid; i < P.nT * 256; i += 256) t_tile[i] = P.tile_dst[i];
      for (int i = tid; i < P.ND; i += 256) { t_dd[i] = P.dir_dst[i]; t_dn[i] = P.dir_next[i]; t_sa[i] = P.dir_sa[i]; t_sb[i] = P.dir_sb[i]; }
      Pc.tile_dst = t_tile; Pc.dir_dst = t_dd; Pc.dir_next = t_dn; Pc.dir_sa = t_sa; Pc.dir_sb = t_sb;
    }
    __syncthreads();
    {   // records of the starting point (k_rig_records) -> broadcast B of round 0
      double a, b;
      rig_candidates(P, nullptr, nullptr, false, s_ctl.cur, s_ctl.cur, a, b);
    }
    __syncthreads();
    bool failed = false;
    for (int round = 0; round < Q.max_rounds; ++round) {
      const unsigned e = Q.epoch0 + (unsigned)round + 1u;
      const bool phase0 = round == 0;
      // ---- broadcast B(e): what this round's sweep evaluates
      if (tid == 0) {
        s_bc[0] = (double)((s_ctl.done ? 1 : 0) | ((phase0 || s_ctl.step_valid) ? 2 : 0) | ((s_ctl.cur & 1) << 3));
        s_bc[1] = s_ctl.radius;
      }
      if (!phase0 && tid < S) s_bc[2 + tid] = -smem[(size_t)S * ((S + 1) | 1) + tid];   // the shared step: -x of rig_solve_block (s_b)
      if (phase0 && tid < S) s_bc[2 + tid] = 0.0;
      for (int i = tid; i < 32 * C; i += 256) s_bc[2 + S + i] = P.camrec[i];
      __syncthreads();
      for (int w = tid; w < 2 * NB; w += 256) ag_st(Q.ybox + w, granule(e, s_bc[w >> 1], w & 1));
      if (s_ctl.done) break;
      // ---- statistics rows -> decision
      const bool swept = phase0 || s_ctl.step_valid;
      {
        const int nst = phase0 ? KS : 4;
        for (int c0 = 0; c0 < nst; c0 += 8) {
          if (tid < 8) s_cols[tid] = c0 + tid;
          __syncthreads();
          double out8[8];
          const int nc = nst - c0 < 8 ? nst - c0 : 8;
          if (!rig_gather_cols<8>(Q.sbox, G, KS, s_cols, nc, -1, e, s4, out8, fail, Q.timeout_shift)) failed = true;
          if (tid == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
              if (j < nc) s_tot[c0 + j] = out8[j];
          }
          __syncthreads();
        }
      }
      if (failed) break;
      // |x|^2 of the shared block at the starting point (k_rig_init)
      double x2_shared = 0.0;
      if (phase0) {
        double x2 = 0.0;
        const int cur0 = s_ctl.cur;
        for (int i = tid; i < C * 7; i += 256) {
          const int cc2 = i / 7;
          const double v = P.cam[((size_t)cur0 * C + cc2) * 8 + (i - cc2 * 7)];
          x2 += P.cam_fixed[cc2] ? 0.0 : v * v;
        }
        x2_shared = block_sum256(x2, s4);
      }
      if (tid == 0) {
        LmCtl c = s_ctl;
        const LmOpts o = *P.opts;
        const int prev_cur = c.cur & 1, was_valid = c.step_valid;
        const double prev_radius = c.radius;
        s_has_rec = 0;
        if (phase0) {
          for (int k = 0; k < S; ++k) s_ss[k] = o.jacobi_scaling ? 1.0 / (1.0 + sqrt(s_tot[4 + k])) : 1.0;
          lm_init(c, o, s_tot[0], sqrt(s_tot[3] + x2_shared));
        } else if (c.cand_pending) {
          double step2 = swept ? s_tot[2] : 0.0, xn2 = swept ? s_tot[3] : 0.0;
          if (c.step_valid) { step2 += P.shared_stats[0]; xn2 += P.shared_stats[1]; }
          const int len0 = c.log_len;
          lm_decide(c, o, &s_rec, swept ? s_tot[0] : 0.0, swept ? s_tot[1] : 0.0, step2, xn2);
          s_has_rec = (c.log_len != len0 && c.log_len <= P.log_cap) ? 1 : 0;
          if (s_has_rec) P.log[c.log_len - 1] = s_rec;
        }
        if (!c.done && round + 1 >= Q.max_rounds) { c.done = 1; c.term = CC_NO_CONVERGENCE; }
        s_ctl = c;
        // did the workers' assumption hold? (the expression they evaluate: persist_spec_radius)
        s_hit = Q.pbox != nullptr && !phase0 && swept && was_valid && !c.done && (c.cur & 1) == (prev_cur ^ 1) &&
                c.radius == persist_spec_radius(prev_radius, o.max_radius);
        s_a[0] = (double)((c.done ? 1 : 0) | (s_hit ? 4 : 0) | ((c.cur & 1) << 3));
        s_a[1] = c.radius;
      }
      __syncthreads();
      if (tid < S) { s_a[2 + tid] = phase0 ? s_ss[tid] : 0.0; if (phase0) P.ss[tid] = s_ss[tid]; }
      __syncthreads();
      for (int w = tid; w < 2 * (2 + S); w += 256) ag_st(Q.abox + w, granule(e, s_a[w >> 1], w & 1));
      if (s_ctl.done) {
        if (tid == 0) { *P.ctl = s_ctl; *P.ctl_next = s_ctl; }
        break;
      }
      // ---- the K column sums -> the layout rig_solve_block reads (P.vec), then the solve step
      if (tid == 0) s_flag = 0;
      __syncthreads();
      {
        const long long t0 = wall_clock64();
        bool good = true;
        for (int k0 = tid; k0 < K && good; k0 += 256 * 4) {
          u64 lo[4], hi[4];
          for (unsigned spins = 0;; ++spins) {
            bool ok = true;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int k = k0 + 256 * u;
              const u64* p = (s_hit ? Q.pcbox : Q.cbox) + (size_t)(k < K ? k : k0) * 2;
              lo[u] = ag_ld(p);
              hi[u] = ag_ld(p + 1);
              ok = ok && (unsigned)(lo[u] >> 32) == e && (unsigned)(hi[u] >> 32) == e;
            }
            if (ok) break;
            if ((spins & 63u) == 63u && (timed_out(t0, Q.timeout_shift) || ag_ld32(fail) != 0u)) { good = false; break; }
            __builtin_amdgcn_s_sleep(1);
          }
          if (!good) break;
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int k = k0 + 256 * u;
            if (k < K) {
              const double v = ungranule(lo[u], hi[u]);
              if (k == K - 1) vl[P.PC + P.rank] = v;   // the gradient maximum rides in the rank's slot (k_rig_reduce)
              else vl[Q.comp[k]] = v;
            }
          }
        }
        if (!good) s_flag = 1;
      }
      __syncthreads();
      if (s_flag == 1) { failed = true; break; }
      if (tid == 0) { *P.ctl = s_ctl; *P.ctl_next = s_ctl; }   // (rig_solve_block finishes the record of this round in P.log)
      __syncthreads();
      rig_solve_block<3>(Pc, smem, &s_ctl, vl);
      __syncthreads();
      if (tid == 0) s_ctl = *P.ctl;   // as the solve step left it (this workgroup wrote it)
      __syncthreads();
    }
    // ---- the solve is over
    __syncthreads();
    if (tid == 0) {
      LmCtl c = s_ctl;
      if (failed || ag_ld32(fail) != 0u) {
        __hip_atomic_store(fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        c.done = 1; c.term = CC_FAILURE_EXCHANGE;
      }
      if (!c.done) { c.done = 1; c.term = CC_NO_CONVERGENCE; }
      *P.ctl = c;
      *P.ctl_next = c;
      rig_publish(P, c);
    }
}

// ---------------------------------------------------------------------------------------------
// The LEAN workers of the persistent rig solve (k_rig_persist_w; small rigs: at most 4 observed cameras, 24 shared
// coordinates) with the control workgroup as a launch of its own beside them (k_rig_persist_ctl: rig_persist_control, on a
// second stream -- it needs 230 registers a thread, the workers must stay under 128 to put SIXTEEN waves on a compute
// unit). Same seams, same rows, same control as k_rig_persist; what differs is where a worker keeps its four frames:
// in LDS -- poses, frame records, the 16 x 16 blocks and compact records of their groups (both buffers), Y, the Jacobi
// scales -- and how it works on them: one WAVE PER GROUP in the sweep (sixteen at once; k_rig_persist: four, one after
// the other), sixteen lanes per frame in the pose update, one wave per frame in the elimination, which builds the
// compact row straight from a slot table (cc_intrinsics_persist.hip's way) instead of going through the partial-row layout.
// ---------------------------------------------------------------------------------------------
constexpr int kRpwMaxS = 24;        // shared coordinates: four optimised cameras (none of them frozen)
constexpr int kRpwMaxCO = 4;        // observed cameras = groups of a frame = sweep waves of a team
constexpr int kRpwYS = 28;          // row stride of Y / Z (S + 1 <= 25 columns)
constexpr int kRpwMaxK = 448;       // compact row entries (S = 24: 325 Schur + 108 direct + 2)
constexpr int kRpwInfoG = 32;       // s_info: [0..31] colinfo of the shared columns, [32..47] group of (team, slot)
// Per-team scratch (doubles). Every region is placed from the SIZE of the one before it, and the sizes from the capacities
// above: a capacity cannot be raised without the layout following (round 3: a slot sized by hand overflowed beyond four
// cameras in all and was found late -- c533722).
enum { RPW_Y = 0,                                  // [6][kRpwYS]   Y
       RPW_Z = RPW_Y + 6 * kRpwYS,                 // [6][kRpwYS]   Z
       RPW_POSE = RPW_Z + 6 * kRpwYS,              // [2][8]        the frame's pose, both buffers
       RPW_FREC = RPW_POSE + 2 * 8,                // [32]          frame record: R(9) t(3) step(6)
       RPW_SP = RPW_FREC + 32,                     // [8]           Jacobi scale of the pose block (6)
       RPW_A = RPW_SP + 8,                         // [32]          damped frame block (21) + its gradient (6)
       RPW_GST = RPW_A + 32,                       // [kRpwMaxCO][2] group statistics
       RPW_FST = RPW_GST + 2 * kRpwMaxCO,          // [2]           frame statistics
       RPW_HD0 = RPW_FST + 2,                      // [kRpwMaxCO][8] diagonals of the camera blocks (first round)
       RPW_ROW = ((RPW_HD0 + 8 * kRpwMaxCO + 7) / 8) * 8,   // [kRpwMaxK] the frame's compact row
       RPW_TEAM = RPW_ROW + kRpwMaxK };
// Workgroup scratch (doubles): broadcast B (flags, radius, step[S], records of ALL C cameras), broadcast A (flags, radius, S
// scales), Jacobi scales (S + 1), statistics row (4 + S), slot table int[kRpwMaxK], sums, s_info int[kRpwInfoG + 4 teams x
// kRpwMaxCO], column list int[8] + good flag.
constexpr int kRpwBcDoubles = ((2 + kRpwMaxS + 32 * kRigPersistMaxC + 7) / 8) * 8 + 24;   // (+ 24: round 3's slot was 344 for nine cameras; kept)
enum { RPW_BC = 0,
       RPW_AB = RPW_BC + kRpwBcDoubles,
       RPW_SS = RPW_AB + 32,
       RPW_SROW = RPW_SS + 32,
       RPW_SLOT = RPW_SROW + 32,                   // int[kRpwMaxK]
       RPW_S16 = RPW_SLOT + kRpwMaxK / 2,
       RPW_INFO = RPW_S16 + 16,                    // int[kRpwInfoG + 16]
       RPW_COLS = RPW_INFO + (kRpwInfoG + 4 * kRpwMaxCO) / 2,   // int[8], int good
       RPW_WG = RPW_COLS + 8 };
static_assert(kRpwYS >= kRpwMaxS + 1, "a row of Y / Z holds the S shared columns and the right-hand side");
static_assert(kRpwInfoG >= kRpwMaxS + 1, "s_info[0..kRpwInfoG) holds colinfo of the S + 1 columns");
static_assert(kRpwMaxK % 2 == 0 && kRpwMaxK >= (kRpwMaxS + 1) * (kRpwMaxS + 2) / 2 + kRpwMaxCO * kDE0 + 2, "compact row: Schur entries of S + 1 columns, 27 direct entries per observed camera, failures, gradient maximum");
static_assert(2 + kRpwMaxS + 32 * kRigPersistMaxC <= RPW_AB - RPW_BC, "broadcast B (step + records of every camera) must fit its LDS slot");
static_assert(2 + kRpwMaxS <= RPW_SS - RPW_AB && kRpwMaxS + 1 <= RPW_SROW - RPW_SS && 4 + kRpwMaxS <= RPW_SLOT - RPW_SROW, "broadcast A, scales and statistics row must fit their LDS slots");
static_assert(27 <= RPW_GST - RPW_A && 6 <= RPW_A - RPW_SP && 18 <= RPW_SP - RPW_FREC, "frame block + gradient, pose scale and frame record must fit their LDS slots");
static_assert(RPW_AB == 344 && RPW_SS == 376 && RPW_SROW == 408 && RPW_SLOT == 440 && RPW_TEAM == ((12 * kRpwYS + 16 + 32 + 8 + 32 + 8 + 2 + 32 + 7) / 8) * 8 + kRpwMaxK,
              "layout as measured in round 3 (profiles/r03/rig_persist_marks.jsonl); a change of capacity moves it knowingly");
constexpr int rpw_lds_doubles(int teams) { return teams * (2048 + 512 + 1024 + RPW_TEAM) + RPW_WG; }

// column sums over the G rows of a box, for a workgroup of NW waves (cf. rig_gather_cols; thread t < G polls row t)
template <int NB, int NW>
__device__ __forceinline__ bool rig_gather_cols_w(const u64* box, int G, int rowlen, const int* cols, int ncols, int maxcol, unsigned tag, double* s16, int* s_good,
                                                  double* out, unsigned* fail, int tshift) {
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  if (tid == 0) *s_good = 1;
  __syncthreads();
  u64 lo[NB], hi[NB];
  bool good = true;
  if (tid < G) {
    const long long t0 = wall_clock64();
    for (unsigned spins = 0;; ++spins) {
      bool ok = true;
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int c = cols[j < ncols ? j : 0];
        const u64* p = box + ((size_t)tid * rowlen + c) * 2;
        lo[j] = ag_ld(p);
        hi[j] = ag_ld(p + 1);
        ok = ok && (j >= ncols || ((unsigned)(lo[j] >> 32) == tag && (unsigned)(hi[j] >> 32) == tag));
      }
      if (ok) break;
      if ((spins & 63u) == 63u && (timed_out(t0, tshift) || ag_ld32(fail) != 0u)) { good = false; break; }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  if (!good) *s_good = 0;
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    if (j < ncols) {   // (uniform)
      const double v = (tid < G && good) ? ungranule(lo[j], hi[j]) : 0.0;
      const bool is_max = cols[j] == maxcol;
      const double w = is_max ? wave_max(v) : wave_sum(v);
      __syncthreads();
      if ((tid & 63) == 0) s16[tid >> 6] = w;
      __syncthreads();
      double r = s16[0];
#pragma unroll
      for (int u = 1; u < NW; ++u) r = is_max ? fmax(r, s16[u]) : r + s16[u];
      out[j] = r;
    }
  }
  __syncthreads();
  return *s_good != 0;
}

// The control workgroup FINDS its compute unit (round 4). Workgroups of a launch are dealt round-robin over the eight XCDs and
// never move; with G = 250 workers two XCDs hold 32 of them -- every compute unit -- and WHICH two is not fixed (the XCD
// block 0 of a launch goes to varies: MI355X guide, workgroup dispatch), so no block index can be told in advance to land
// next to a free compute unit (round 3 launched (G mod 8) + 1 blocks and let the last one work: right only when both
// launches start their round on the same XCD). This launch has kRigCtlCandidates blocks -- two per XCD -- and the FIRST one
// that gets to run claims the solve (one exchange on a word tagged with the solve's epoch) and is the control; the others
// leave as soon as they run (those queued on a full XCD: when the workers are gone). A claim needs a free compute unit
// somewhere, which G <= 255 leaves; the XCD that gave it is recorded (arrive[12]) for the host's diagnostics.
constexpr int kRigCtlCandidates = 16;
__global__ __launch_bounds__(256) void k_rig_persist_ctl(RigDev P, RigPersistDev Q) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  __shared__ int s_mine;
  if (threadIdx.x == 0) {
    const unsigned tag = Q.epoch0 + 1u;
    const unsigned prev = __hip_atomic_exchange(Q.claim, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_mine = prev != tag;
    if (prev != tag) {
      unsigned xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
      __hip_atomic_store(P.arrive + 12, ((unsigned)blockIdx.x << 8) | (xcc + 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __syncthreads();
  if (!s_mine) return;
  rig_persist_control(P, Q, smem_raw);
}

template <int TEAMS>
__global__ __launch_bounds__(TEAMS * 256) void k_rig_persist_w(RigDev P, RigPersistDev Q) {
  extern __shared__ __attribute__((aligned(16))) double rpw_lds[];
  constexpr int NT = TEAMS * 256;           // threads
  double* s_tile = rpw_lds;                 // [TEAMS][4 slots][2][256]
  double* s_comp = s_tile + TEAMS * 2048;   // [TEAMS][4][2][64]
  double* s_sw = s_comp + TEAMS * 512;      // [TEAMS * 4 waves][256] sweep scratch
  double* s_tm = s_sw + TEAMS * 1024;       // [TEAMS][RPW_TEAM]
  double* s_wg = s_tm + TEAMS * RPW_TEAM;   // [RPW_WG]
  const int tid0 = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid0 >> 6), team = wave >> 2, twave = wave & 3, lane = tid0 & 63;
  const int S = P.S, SW = P.SW, CO = P.CO, G = Q.G, K = Q.K, KS = Q.KS, NB = Q.NB;
  unsigned* fail = P.arrive + 3;
  const int64_t f = (int64_t)blockIdx.x * TEAMS + team;
  const bool has_frame = f < P.F;
  const int g_mine = __builtin_amdgcn_readfirstlane((has_frame && twave < CO) ? P.fslot[f * CO + twave] : -1);   // the group this wave sweeps
  double* tm = s_tm + team * RPW_TEAM;
  double* s_bc = s_wg + RPW_BC;
  double* s_a = s_wg + RPW_AB;
  double* s_ss = s_wg + RPW_SS;
  double* s_row = s_wg + RPW_SROW;
  int* s_slot = reinterpret_cast<int*>(s_wg + RPW_SLOT);
  double* s16 = s_wg + RPW_S16;
  int* s_info = reinterpret_cast<int*>(s_wg + RPW_INFO);   // [0..31] colinfo, [32..47] group of (team, slot)
  int* s_cols = reinterpret_cast<int*>(s_wg + RPW_COLS);
  int* s_good = s_cols + 8;
  // ---- every worker is RESIDENT once all G have passed this point: the last one tells the host, which launches the control
  // only then -- its candidates therefore only ever run on compute units the workers left free (k_rig_persist_ctl)
  if (Q.gate && tid0 == 0) {
    const unsigned prev = __hip_atomic_fetch_add(P.arrive + 13, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (prev + 1u == (unsigned)G) __hip_atomic_store(Q.gate, (unsigned long long)(Q.epoch0 + 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  // ---- start of the solve: poses of the workgroup's frames (buffer 0 holds the starting point: rig_begin), tables
  for (int i = tid0; i < TEAMS * (2048 + 512); i += NT) s_tile[i] = 0.0;
  for (int i = tid0; i < TEAMS * RPW_TEAM; i += NT) s_tm[i] = 0.0;
  for (int i = tid0; i < K; i += NT) s_slot[i] = Q.slots[i];
  if (tid0 < SW) s_info[tid0] = P.colinfo[tid0];
  if (tid0 >= 64 && tid0 < 64 + 4 * TEAMS) {
    const int t = (tid0 - 64) >> 2, j = (tid0 - 64) & 3;
    const int64_t ff = (int64_t)blockIdx.x * TEAMS + t;
    s_info[kRpwInfoG + t * 4 + j] = (ff < P.F && j < CO) ? P.fslot[ff * CO + j] : -1;
  }
  __syncthreads();
  if (has_frame && twave == 0 && lane < 8) {
    const double v = lane < 7 ? P.pose[(size_t)f * 8 + lane] : 0.0;
    tm[RPW_POSE + lane] = v;
    tm[RPW_POSE + 8 + lane] = v;
  }
  __syncthreads();
  int cur = 0;
  double radius = 1.0;
  for (int round = 0; round < Q.max_rounds; ++round) {
    const unsigned e = Q.epoch0 + (unsigned)round + 1u;
    const bool phase0 = round == 0;
    // ---- broadcast B: step and camera records
    if (wave == 0 && !rig_bcast_wait(Q.ybox, e, NB, s_bc, fail, phase0 ? Q.first_shift : Q.timeout_shift)) s_bc[0] = 1.0;
    __syncthreads();
    const int flb = (int)s_bc[0];
    if (flb & 1) { cur = (flb >> 3) & 1; break; }
    cur = (flb >> 3) & 1;
    const bool swept = (flb & 2) != 0;
    const int dst = phase0 ? cur : (cur ^ 1);
    if (swept) {
      // ---- pose update of the frame (rig_update_body's arithmetic): sixteen lanes of the team's first wave
      int ul = tid0 & 63;
      asm volatile("" : "+v"(ul));
      if (has_frame && twave == 0 && ul < 16) {
        const int lane = ul;
        double u[6] = {0, 0, 0, 0, 0, 0};
        if (!phase0) {
          for (int k = lane; k < SW; k += 16) {
            const double d = k < S ? s_bc[2 + k] : 1.0;
#pragma unroll
            for (int i = 0; i < 6; ++i) u[i] += tm[RPW_Y + i * kRpwYS + k] * d;
          }
#pragma unroll
          for (int i = 0; i < 6; ++i) u[i] = row16_sum(u[i]);
        }
        if (lane == 0) {
          bool active = false;
          for (int j = 0; j < 4; ++j) active = active || s_info[kRpwInfoG + team * 4 + j] >= 0;
          double q[4], t[3], dp[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
          for (int i = 0; i < 4; ++i) q[i] = tm[RPW_POSE + cur * 8 + i];
#pragma unroll
          for (int i = 0; i < 3; ++i) t[i] = tm[RPW_POSE + cur * 8 + 4 + i];
          double step2 = 0.0;
          if (!phase0) {
            if (active) {
#pragma unroll
              for (int i = 0; i < 6; ++i) dp[i] = -u[i] * tm[RPW_SP + i];
              double qn[4];
              quat_plus_tab(q, dp, qn);   // (series coefficients from a table: as literals they are hoisted out of the round loop and spilled)
#pragma unroll
              for (int i = 0; i < 4; ++i) { const double d = qn[i] - q[i]; step2 += d * d; q[i] = qn[i]; }
#pragma unroll
              for (int i = 0; i < 3; ++i) { const double tn = t[i] + dp[3 + i]; const double d = tn - t[i]; step2 += d * d; t[i] = tn; }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) tm[RPW_POSE + dst * 8 + i] = q[i];
#pragma unroll
            for (int i = 0; i < 3; ++i) tm[RPW_POSE + dst * 8 + 4 + i] = t[i];
          }
          double R[9];
          quat_to_R(q, R);
#pragma unroll
          for (int i = 0; i < 9; ++i) tm[RPW_FREC + i] = R[i];
#pragma unroll
          for (int i = 0; i < 3; ++i) tm[RPW_FREC + 9 + i] = t[i];
#pragma unroll
          for (int i = 0; i < 6; ++i) tm[RPW_FREC + 12 + i] = dp[i];
          tm[RPW_FST] = step2;
          tm[RPW_FST + 1] = active ? q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3] + t[0] * t[0] + t[1] * t[1] + t[2] * t[2] : 0.0;
        }
      }
      __syncthreads();
      // ---- sweep: one wave per group
      if (g_mine >= 0) {
        const RigSweepIO io{s_bc + 2 + S, tm + RPW_FREC, s_comp + ((team * 4 + twave) * 2 + cur) * 64, s_tile + ((team * 4 + twave) * 2 + dst) * 256,
                            s_comp + ((team * 4 + twave) * 2 + dst) * 64, tm + RPW_GST + 2 * twave, tm + RPW_HD0 + 8 * twave};
        rig_sweep_adj_body<1, true>(P, g_mine, phase0 ? 0 : 1, cur, s_sw + wave * 256, io);
      }
      __syncthreads();
    }
    // ---- statistics row of the workgroup (teams and slots in order)
    {
      int tid = tid0;
      asm volatile("" : "+v"(tid));
      if (tid == 0) {
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        if (swept) {
          for (int t = 0; t < TEAMS; ++t) {
            for (int j = 0; j < 4; ++j)
              if (s_info[kRpwInfoG + t * 4 + j] >= 0) { a0 += s_tm[t * RPW_TEAM + RPW_GST + 2 * j]; a1 += s_tm[t * RPW_TEAM + RPW_GST + 2 * j + 1]; }
            if ((int64_t)blockIdx.x * TEAMS + t < P.F) { a2 += s_tm[t * RPW_TEAM + RPW_FST]; a3 += s_tm[t * RPW_TEAM + RPW_FST + 1]; }
          }
        }
        s_row[0] = a0; s_row[1] = a1; s_row[2] = a2; s_row[3] = a3;
      }
      if (phase0 && tid >= 64 && tid < 64 + S) {   // diagonal of H_cc per shared column (Jacobi scaling)
        const int k = tid - 64, info = s_info[k], j = info >> 8, comp = info & 15;
        double d = 0.0;
        for (int t = 0; t < TEAMS; ++t)
          if (s_info[kRpwInfoG + t * 4 + j] >= 0) d += s_tm[t * RPW_TEAM + RPW_HD0 + 8 * j + comp];
        s_row[4 + k] = d;
      }
      __syncthreads();
      const int nst = phase0 ? KS : 4;
      if (tid < 2 * nst) ag_st(Q.sbox + ((size_t)blockIdx.x * KS) * 2 + tid, granule(e, s_row[tid >> 1], tid & 1));
    }
    // ---- the assumed decision (cf. cc_intrinsics_persist.hip): candidate accepted, radius at its clamp -- the normal outcome of a
    // step that works. The workers eliminate the candidate NOW, next to the control's gathering and deciding; when the
    // decision is what was assumed (broadcast A says so) the rows are already where the control looks for them.
    const bool spec = !phase0 && swept;
    const double radius_spec = persist_spec_radius(radius, P.opts->max_radius);
    auto eliminate_and_post = [&](const int cur_e, const double radius_e, const bool first_e, u64* rowbox, u64* colbox, const bool is_spec) {
    // ---- elimination of the frame: the team's first wave
    if (twave == 0) {
      int ln = lane;
      asm volatile("" : "+v"(ln));
      double* rowt = tm + RPW_ROW;
      bool exists[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) exists[j] = has_frame && s_info[kRpwInfoG + team * 4 + j] >= 0;
      const bool live = exists[0] || exists[1] || exists[2] || exists[3];
      const double* T0 = s_tile + ((team * 4) * 2 + cur_e) * 256;   // slot j: T0 + j * 512
      bool ok = true;
      double gmaxp = 0.0;
      if (live) {
        // frame block A = sum over the groups: lanes 0..26 (21 entries of H_ff, 6 of g_f)
        if (ln < 27) {
          int a_off;
          if (ln < 21) { int i = 0; while (tri(i + 1, 0) <= ln) ++i; a_off = (6 + i) * 16 + 6 + (ln - tri(i, 0)); }
          else a_off = (6 + (ln - 21)) * 16 + 12;
          double a_e = 0.0;
#pragma unroll
          for (int j = 0; j < 4; ++j) a_e += exists[j] ? T0[j * 512 + a_off] : 0.0;
          tm[RPW_A + ln] = a_e;
        }
        wave_lds_fence();
        const bool jac = P.opts->jacobi_scaling != 0;
        const double mn = P.opts->min_lm_diagonal, mx = P.opts->max_lm_diagonal;
        const double inv_radius = 1.0 / radius_e;
        double sf[6], L[21], Li[6], gf[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) gf[i] = tm[RPW_A + 21 + i];
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
          for (int j = 0; j <= i; ++j) L[tri(i, j)] = tm[RPW_A + tri(i, j)];
        if (first_e) {
#pragma unroll
          for (int i = 0; i < 6; ++i) sf[i] = jac ? 1.0 / (1.0 + sqrt(L[tri(i, i)])) : 1.0;
          if (ln < 6) {
            double sl = 0.0;
#pragma unroll
            for (int i = 0; i < 6; ++i) sl = ln == i ? sf[i] : sl;
            tm[RPW_SP + ln] = sl;
          }
        } else {
#pragma unroll
          for (int i = 0; i < 6; ++i) sf[i] = tm[RPW_SP + i];
        }
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
          for (int j = 0; j <= i; ++j) L[tri(i, j)] = sf[i] * L[tri(i, j)] * sf[j];
#pragma unroll
        for (int i = 0; i < 6; ++i) L[tri(i, i)] += clampd(L[tri(i, i)], mn, mx) * inv_radius;
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
        {   // the frame's share of Ceres' gradient_max_norm (pose_grad_proj_max, cc_common.hpp)
          double q4[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) q4[i] = tm[RPW_POSE + cur_e * 8 + i];
          gmaxp = pose_grad_proj_max_tab(q4, gf);
        }
        // the factor is the same in every lane: scalar registers from here on
#pragma unroll
        for (int i = 0; i < 21; ++i) L[i] = rfl(L[i]);
#pragma unroll
        for (int i = 0; i < 6; ++i) { Li[i] = rfl(Li[i]); sf[i] = rfl(sf[i]); }
        if (ln < SW) {   // shared column ln (ln == S: the right-hand side)
          const int info = s_info[ln], kind = (info >> 4) & 15, j = info >> 8, comp = info & 15;
          const double sc = ln < S ? s_ss[ln] : 1.0;
          const double* Tj = T0 + (kind == 0 ? j : 0) * 512;
          const bool ex = kind == 0 && ((j == 0 && exists[0]) || (j == 1 && exists[1]) || (j == 2 && exists[2]) || (j == 3 && exists[3]));
          double z[6], y[6];
#pragma unroll
          for (int i = 0; i < 6; ++i) {
            const double w = Tj[comp * 16 + 6 + i];
            double a = kind == 3 ? sf[i] * gf[i] : (ex ? sf[i] * w * sc : 0.0);
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
          for (int i = 0; i < 6; ++i) { tm[RPW_Z + i * kRpwYS + ln] = z[i]; tm[RPW_Y + i * kRpwYS + ln] = y[i]; }
        }
        wave_lds_fence();
      }
      // the frame's compact row, slot k on lane k mod 64
      for (int k = ln; k < K; k += 64) {
        const int code = s_slot[k];
        double v = 0.0;
        if (live) {
          if (code == -1) v = ok ? 0.0 : 1.0;
          else if (code == -2) v = gmaxp;
          else if (code & (1 << 30)) {
            const int pcol = (code >> 8) & 255, qcol = code & 255;
#pragma unroll
            for (int i = 0; i < 6; ++i) v += tm[RPW_Z + i * kRpwYS + pcol] * tm[RPW_Z + i * kRpwYS + qcol];
          } else {
            const int j = code >> 16;
            const bool ex = (j == 0 && exists[0]) || (j == 1 && exists[1]) || (j == 2 && exists[2]) || (j == 3 && exists[3]);
            v = ex ? T0[j * 512 + (code & 0xffff)] : 0.0;
          }
        }
        rowt[k] = v;
      }
    }
    __syncthreads();
    // ---- the workgroup's row (teams in order) -> granules
    {
      int tid = tid0;
      asm volatile("" : "+v"(tid));
      for (int k = tid; k < K; k += NT) {
        double v = s_tm[RPW_ROW + k];
        for (int t = 1; t < TEAMS; ++t) {
          const double w = s_tm[t * RPW_TEAM + RPW_ROW + k];
          v = k == K - 1 ? fmax(v, w) : v + w;
        }
        u64* q = rowbox + ((size_t)blockIdx.x * K + k) * 2;
        ag_st(q, granule(e, v, 0));
        ag_st(q + 1, granule(e, v, 1));
      }
    }
    // ---- this workgroup's share of the column sums: columns b, b + G, ...
    for (int c0 = (int)blockIdx.x; c0 < K; c0 += 8 * G) {
      int tidc = tid0;
      asm volatile("" : "+v"(tidc));
      if (tidc < 8) s_cols[tidc] = c0 + tidc * G < K ? c0 + tidc * G : c0;
      __syncthreads();
      int nc = 0;
      for (int j = 0; j < 8; ++j) nc += c0 + j * G < K ? 1 : 0;
      double out8[8];
      const bool okg = rig_gather_cols_w<8, TEAMS * 4>(rowbox, G, K, s_cols, nc, K - 1, e, s16, s_good, out8, fail, Q.timeout_shift);
      if (!okg && tidc == 0) __hip_atomic_store(fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (okg && tidc < 2 * nc) {
        const int j = tidc >> 1;
        double v = 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u) v = j == u ? out8[u] : v;
        ag_st(colbox + (size_t)s_cols[j] * 2 + (tidc & 1), granule(e, v, tidc & 1));
      }
      __syncthreads();
    }
    };
    if (spec) eliminate_and_post(dst, radius_spec, false, Q.pbox, Q.pcbox, true);
    // ---- broadcast A: the decision
    if (wave == 0 && !rig_bcast_wait(Q.abox, e, 2 + S, s_a, fail, Q.timeout_shift)) s_a[0] = 1.0;
    __syncthreads();
    const int fla = (int)s_a[0];
    if (fla & 1) { cur = (fla >> 3) & 1; break; }
    if (fla & 4) {   // the assumption held
      cur = dst;
      radius = radius_spec;
    } else {
      cur = (fla >> 3) & 1;
      radius = s_a[1];
      {
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        if (phase0 && tid < S) s_ss[tid] = s_a[2 + tid];
      }
      __syncthreads();
      eliminate_and_post(cur, radius, phase0, Q.rbox, Q.cbox, false);
    }
  }
  // ---- the solve is over: the frame's accepted pose goes back to global memory (cc_rig_get_state, the next solve)
  __syncthreads();
  if (ag_ld32(fail) == 0u && has_frame && twave == 0 && lane < 7) P.pose[((size_t)cur * P.F + f) * 8 + lane] = tm[RPW_POSE + cur * 8 + lane];
}

