// cc_rig_steps.hpp -- the steps of an LM iteration behind the sweep: pose update, statistics, first-round scaling (k_rig_init), elimination of the frame poses (k_rig_elim), Cholesky routines of the reduced system, solve step (rig_solve_block), column sums + fused solve + update (k_rig_reduce).
// Part of cc_rig.hip (round 5: the 6.8 k-line file split by subject; included by it inside namespace cc, in this order:
// cc_rig_sweeps.hpp, cc_rig_steps.hpp, cc_rig_big.hpp, cc_rig_lean.hpp -- one translation unit, nothing else includes these).
#pragma once

// ---------------------------------------------------------------------------------------------
// update: per frame, back-substitute the pose step and form the candidate pose. 16 lanes/frame, 16 frames per
// 256-thread block (`fblk` = which sixteen). SC1: the shared step `ds` was written by another workgroup of the
// SAME launch (fused into k_rig_reduce): read it with sc1 loads.
// ---------------------------------------------------------------------------------------------
// What the update of a frame needs besides the shared step: fetched by the blocks of k_rig_reduce WHILE they wait for
// the solving block's flag (SW <= 32: two Y columns per lane), so that only the step itself is read behind the flag.
// Everything unconditional (all sixteen lanes of a frame fetch the frame's scalars: same addresses, one transaction).
struct RigUpdPre {   // (y: columns l, l + 16, l + 32, l + 48 of the frame's six rows of Y -- up to 64 shared columns)
  double y[4][6], p0[7], p1[7], sp[6];
  int g0, g1;
};
__device__ __forceinline__ void rig_update_prefetch(const RigDev& P, int64_t f, RigUpdPre& x) {   // f: frame of this thread's sixteen lanes
  const int tid = threadIdx.x, l = tid & 15;
  const int64_t fc = f < P.F ? f : 0;
  const double* Yf = P.Y + (size_t)fc * 6 * P.SW;
#pragma unroll
  for (int h = 0; h < 4; ++h) {
    const int k = l + 16 * h, kc = k < P.SW ? k : 0;
    if (16 * h < P.SW) {   // (uniform)
#pragma unroll
      for (int i = 0; i < 6; ++i) x.y[h][i] = Yf[i * P.SW + kc];
    } else {
#pragma unroll
      for (int i = 0; i < 6; ++i) x.y[h][i] = 0.0;
    }
  }
#pragma unroll
  for (int i = 0; i < 7; ++i) { x.p0[i] = P.pose[(size_t)fc * 8 + i]; x.p1[i] = P.pose[((size_t)P.F + fc) * 8 + i]; }
#pragma unroll
  for (int i = 0; i < 6; ++i) x.sp[i] = P.sp[fc * 8 + i];
  x.g0 = P.fgoff[fc]; x.g1 = P.fgoff[fc + 1];
}

template <bool SC1, bool PRE = false>
// f: frame of this thread's sixteen lanes; ds_lds: the shared step in LDS (persistent kernel), else read from P.ds
__device__ __forceinline__ void rig_update_body(const RigDev& P, int phase, int cur, int64_t f, const RigUpdPre& pre, const double* ds_lds = nullptr) {
  const int dst = phase == 0 ? cur : (cur ^ 1);
  int tid_ = threadIdx.x;
  if (ds_lds) asm volatile("" : "+v"(tid_));
  const int tid = tid_, l = tid & 15;
  const bool valid = f < P.F;
  const int64_t fc = valid ? f : 0;
  double u[6] = {0, 0, 0, 0, 0, 0};
  if (phase != 0 && PRE) {
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const int k = l + 16 * h;
      if (16 * h >= P.SW) continue;   // (uniform)
      double d = 1.0;   // (column S is the right-hand side)
      const double dk = __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(P.ds) + (k < P.S ? k : 0),
                                                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      if (k < P.S) d = dk;
      if (k < P.SW) {
#pragma unroll
        for (int i = 0; i < 6; ++i) u[i] += pre.y[h][i] * d;
      }
    }
  } else if (phase != 0) {
    // Eight columns per lane (128 of the shared step) in ONE round trip: every load of the batch -- 48 of Y, 8 of the step -- is
    // issued before the first product (round 6; the rolled loop waited for each column's seven loads on their own: eight
    // dependent round trips behind the flag at S = 114, 14 us of the reduce launch). Same products in the same order.
    const double* Yf = P.Y + (size_t)fc * 6 * P.SW;
    if (ds_lds) {   // (the lean persistent workers: at most 25 columns, a register budget of their own -- the plain loop)
      for (int k = l; k < P.SW; k += 16) {
        const double d = k < P.S ? ds_lds[k] : 1.0;
#pragma unroll
        for (int i = 0; i < 6; ++i) u[i] += Yf[i * P.SW + k] * d;
      }
    } else
    for (int k0 = l; k0 < P.SW; k0 += 128) {
      double y[8][6], d[8];
#pragma unroll
      for (int h = 0; h < 8; ++h) {
        const int k = k0 + 16 * h, kc = k < P.SW ? k : 0, ks = k < P.S ? k : 0;
#pragma unroll
        for (int i = 0; i < 6; ++i) y[h][i] = Yf[i * P.SW + kc];
        const double dk = SC1 ? __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(P.ds) + ks, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                        : P.ds[ks];
        d[h] = k < P.S ? dk : 1.0;   // (column S is the right-hand side)
      }
#pragma unroll
      for (int h = 0; h < 8; ++h) {
        if (k0 + 16 * h < P.SW) {
#pragma unroll
          for (int i = 0; i < 6; ++i) u[i] += y[h][i] * d[h];
        }
      }
    }
  }
  if (phase != 0) {   // the sixteen lanes of a frame add up their columns
#pragma unroll
    for (int i = 0; i < 6; ++i) u[i] = row16_sum(u[i]);
  }
  if (!valid || l != 0) return;
  bool active;
  double q[4], t[3], spf[6];
  if (PRE) {
    active = pre.g1 > pre.g0;
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = cur ? pre.p1[i] : pre.p0[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) t[i] = cur ? pre.p1[4 + i] : pre.p0[4 + i];
#pragma unroll
    for (int i = 0; i < 6; ++i) spf[i] = pre.sp[i];
  } else {
    active = P.fgoff[f + 1] > P.fgoff[f];
    const double* pc = P.pose + ((size_t)cur * P.F + f) * 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = pc[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) t[i] = pc[4 + i];
#pragma unroll
    for (int i = 0; i < 6; ++i) spf[i] = phase != 0 ? P.sp[f * 8 + i] : 0.0;
  }
  double dp[6] = {0, 0, 0, 0, 0, 0};
  double step2 = 0.0;
  if (phase != 0) {
    if (active) {
#pragma unroll
      for (int i = 0; i < 6; ++i) dp[i] = -u[i] * spf[i];
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

__global__ __launch_bounds__(256) void k_rig_update(RigDev P) {
  rig_progress(P, RIG_PROG_UPDATE);
  const LmCtl* ctl = P.ctl;
  if (ctl->done) return;
  const int phase = ctl->phase;
  if (phase != 0 && !ctl->step_valid) return;
  RigUpdPre none;   // (unused: PRE = false)
  rig_update_body<false, false>(P, phase, ctl->cur, (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4), none);
}

// deterministic block-wide sum of one value per thread (256 threads); result valid for thread 0
__device__ __forceinline__ double block_sum256(double v, double* s4) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s4[threadIdx.x >> 6] = v;
  __syncthreads();
  return (s4[0] + s4[1]) + (s4[2] + s4[3]);
}

// column sums of gstats[NG][2] and fstats[F][2] -> out[0..3] = cost, q, step2, xnorm2 (all threads after return)
// pre_g / pre_f (optional): the rows i = u * 256 + tid, u < 8, of gstats / fstats requested by the caller at kernel start
// (at most 2048 rows each: the frame form's one row per frame) -- same sums in the same order, one round trip earlier
__device__ __forceinline__ void rig_reduce_stats(const RigDev& P, bool want, double* s16, double* out, const d2* pre_g = nullptr,
                                                 const d2* pre_f = nullptr) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double a[4] = {0, 0, 0, 0};
  if (want && pre_g) {
#pragma unroll
    for (int u = 0; u < 8; ++u) { a[0] += pre_g[u].x; a[1] += pre_g[u].y; }
#pragma unroll
    for (int u = 0; u < 8; ++u) { a[2] += pre_f[u].x; a[3] += pre_f[u].y; }
  } else if (want) {
    const d2* gs2 = reinterpret_cast<const d2*>(P.gstats);
    const d2* fs2 = reinterpret_cast<const d2*>(P.fstats);
    // up to sixteen loads in flight per thread: one round trip per 4096 groups instead of one per 256 (the plain loop waited
    // for every load: 20 us for the 16000 groups of BASELINE configs[4], in every block of the elimination)
    const int64_t nrows = P.fmode ? P.F : P.NG;   // (frame form: one row of cost / model-cost term per FRAME)
    for (int64_t i0 = 0; i0 < nrows; i0 += 16 * 256) {
      d2 v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) { const int64_t i = i0 + u * 256 + tid; v[u] = i < nrows ? gs2[i] : d2{0.0, 0.0}; }
#pragma unroll
      for (int u = 0; u < 16; ++u) { a[0] += v[u].x; a[1] += v[u].y; }
    }
    for (int64_t i0 = 0; i0 < P.F; i0 += 8 * 256) {
      d2 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int64_t i = i0 + u * 256 + tid; v[u] = i < P.F ? fs2[i] : d2{0.0, 0.0}; }
#pragma unroll
      for (int u = 0; u < 8; ++u) { a[2] += v[u].x; a[3] += v[u].y; }
    }
    // (measured: clamped unconditional loads + selects are 1-2 us slower here than these selects on the loaded value)
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) a[k] = wave_sum(a[k]);
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) s16[wave * 4 + k] = a[k];
  }
  __syncthreads();
  if (tid < 4) out[tid] = (s16[tid] + s16[4 + tid]) + (s16[8 + tid] + s16[12 + tid]);
  __syncthreads();
}

// Sum over this rank's groups of the diagonal entry of shared column k at the initial point (Jacobi scaling of
// the shared block). All 256 threads call; the result is valid for thread 0.
// Diagonal of the shared block at the initial point, all S columns -> out[0..S) (LDS), for the Jacobi scale. The columns
// of one camera (6 pose coordinates, kind 0) or of one intrinsics set (9, kinds 1 and 2) are consecutive and sum over the
// same groups, so a run is reduced together: the group indices of four steps are fetched first, then their values (two
// round trips per 1024 groups and run, fixed summation order). The first version walked one column at a time with a
// dependent index -> value load pair per step: 138 us for BASELINE configs[4], once per solve.
__device__ __forceinline__ void rig_diag_sums(const RigDev& P, double* s4, double* out, int only_run = -1, int slice = 0, int nslices = 1) {
  const int tid = threadIdx.x;
  int run = 0;
  for (int k = 0; k < P.S; ++run) {
    const int info = P.colinfo[k], kind = (info >> 4) & 15, co = info >> 8;
    const int n = kind == 0 ? 6 : kRigK;
    if (only_run >= 0 && run != only_run) { k += n; continue; }   // (k_rig_init: one run per block)
    const double* src = kind == 0 ? P.ghd0 : P.ghdk;
    const int stride = kind == 0 ? 8 : 16;
    int64_t lo = 0, hi = P.NG;
    if (kind != 2) { const int c = P.obs_cam[co]; lo = P.cam_goff[c]; hi = P.cam_goff[c + 1]; }
    else if (nslices > 1) {   // (k_rig_init: a set shared by all cameras is summed by several blocks, each over a slice of the groups)
      const int64_t len = (P.NG + nslices - 1) / nslices;
      lo = (int64_t)slice * len < P.NG ? (int64_t)slice * len : P.NG;
      hi = lo + len < P.NG ? lo + len : P.NG;
    }
    double h[kRigK];
#pragma unroll
    for (int c = 0; c < kRigK; ++c) h[c] = 0.0;
    for (int64_t i0 = lo; i0 < hi; i0 += 4 * 256) {
      int64_t g[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t i = i0 + u * 256 + tid;
        const int64_t ic = i < hi ? i : lo;   // (unconditional loads; idle slots are selected away below)
        g[u] = kind == 2 ? ic : (int64_t)P.cam_glist[ic];
      }
      double v[4][kRigK];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int c = 0; c < kRigK; ++c) v[u][c] = c < n ? src[(size_t)g[u] * stride + c] : 0.0;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const bool live = i0 + u * 256 + tid < hi;
#pragma unroll
        for (int c = 0; c < kRigK; ++c) h[c] += live ? v[u][c] : 0.0;
      }
    }
#pragma unroll
    for (int c = 0; c < kRigK; ++c) {
      if (c < n) {   // (uniform)
        const double sum = block_sum256(h[c], s4);
        if (tid == 0) out[k + c] = sum;
      }
    }
    k += n;
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// stats (multi-GPU only, one block): local sums of the sweep statistics -> vec_stats, which is then
// exchanged. [0..3] cost, model term, step^2, |x|^2; in phase 0 also [4 + k] = diagonal sum of shared
// column k over this rank's groups (Jacobi scaling of the shared block).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rig_stats(RigDev P) {
  rig_progress(P, RIG_PROG_STATS);
  __shared__ double s4[4];
  __shared__ double s16[16];
  __shared__ double s_out[4];
  const LmCtl* ctl = P.ctl;
  if (ctl->done) return;
  const int tid = threadIdx.x, phase = ctl->phase;
  const bool need = phase == 0 || (ctl->cand_pending && ctl->step_valid);
  rig_reduce_stats(P, need, s16, s_out);
  if (tid < 4) P.vec_stats[tid] = need ? s_out[tid] : 0.0;
  {
    __shared__ double s_diag[256];
    if (phase == 0) rig_diag_sums(P, s4, s_diag);
    for (int k = tid; k < P.S; k += 256) P.vec_stats[4 + k] = phase == 0 ? s_diag[k] : 0.0;
  }
  if (P.x.on) {
    // mailbox exchange (kind 1): post the local statistics, wait for every rank's, write the sums back
    // (k_rig_init / k_rig_elim read vec_stats as they do after an all-reduce)
    __shared__ double s_post[4 + 256];
    __shared__ int s_ok;
    __syncthreads();
    const int n = 4 + P.S;   // (up to 259: large rigs)
    for (int k = tid; k < n; k += 256) s_post[k] = P.vec_stats[k];
    __syncthreads();
    const unsigned long long epoch = P.x.seq[1] + 1ull;
    p2p_post(P.x, 1, epoch, P.rank, P.nranks, s_post, n);
    p2p_collect_to(P.x, 1, epoch, P.rank, P.nranks, n, P.vec_stats, &s_ok);
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
__global__ __launch_bounds__(128) void k_rig_flag_exchange(RigDev P, const double* in, double* out, int n, int* ok) {
  __shared__ double s_post[128];
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
  rig_progress(P, RIG_PROG_INIT);
  __shared__ double s4[4];
  __shared__ double s16[16];
  __shared__ double s_out[4];
  __shared__ double s_ss[256];
  const LmCtl* ctl = P.ctl;
  // Only block 0 looks at the control block: it ends by storing lm_init's result (phase = 1) into it, so a block of this
  // launch that is dispatched late (busy or partitioned GPU) would see the flipped phase, return, and leave its run's
  // Jacobi scales unset and -- a sliced run -- the arrival counter short. The kernel is launched in the first round of a
  // solve only (rig_enqueue_round, `initial`), where the phase IS 0 unless an exchange has already failed; the scales a
  // failed solve computes for nothing are harmless.
  if (blockIdx.x == 0 && (ctl->done || ctl->phase != 0)) return;
  const int tid = threadIdx.x;
  const bool jac = P.opts->jacobi_scaling != 0;
  // Single GPU: 1 + (runs of columns) blocks. Block r > 0 sums the diagonal of run r - 1 (one camera's poses or one
  // intrinsics set) and writes its Jacobi scales; block 0 does the rest. Nothing is exchanged between the blocks. (One
  // block doing all runs took 130 us at BASELINE configs[4]: tens of thousands of scattered 8-byte loads through one CU.)
  if (blockIdx.x > 0) {
    if (P.comm) return;
    // which run, and which slice of it (a set of intrinsics shared by all cameras sums over EVERY group: 16000 at
    // BASELINE configs[4], 61 us in one block; init_slices blocks take a slice each and the last one to arrive adds the
    // partial sums up in slice order)
    int b = (int)blockIdx.x - 1, run = 0, slice = 0, ns = 1, k0 = 0;
    for (int k = 0; k < P.S; ++run) {
      const int kind = (P.colinfo[k] >> 4) & 15;
      const int cnt = kind == 2 ? P.init_slices : 1;
      if (b < cnt) { slice = b; ns = cnt; k0 = k; break; }
      b -= cnt;
      k += kind == 0 ? 6 : kRigK;
    }
    for (int k = tid; k < 256; k += 256) s_ss[k] = -1.0;
    __syncthreads();
    rig_diag_sums(P, s4, s_ss, run, slice, ns);
    if (ns == 1) {
      for (int k = tid; k < P.S; k += 256)
        if (s_ss[k] >= 0.0) P.ss[k] = jac ? 1.0 / (1.0 + sqrt(s_ss[k])) : 1.0;
      return;
    }
    __shared__ int s_last_slice;
    unsigned long long* part = reinterpret_cast<unsigned long long*>(P.partial);   // (free until the first elimination)
    if (tid < kRigK) __hip_atomic_store(part + slice * 16 + tid, (unsigned long long)__double_as_longlong(s_ss[k0 + tid]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) s_last_slice = __hip_atomic_fetch_add(P.arrive + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == (unsigned)ns;
    __syncthreads();
    if (!s_last_slice) return;
    if (tid == 0) __hip_atomic_store(P.arrive + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // next solve
    if (tid < kRigK) {
      double t = 0.0;
      for (int q = 0; q < ns; ++q) t += __longlong_as_double((long long)__hip_atomic_load(part + q * 16 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      P.ss[k0 + tid] = jac ? 1.0 / (1.0 + sqrt(t)) : 1.0;
    }
    return;
  }
  if (P.comm) {
    if (tid < 4) s_out[tid] = P.vec_stats[tid];
    if (tid < P.S) P.ss[tid] = jac ? 1.0 / (1.0 + sqrt(P.vec_stats[4 + tid])) : 1.0;
  } else {
    rig_reduce_stats(P, true, s16, s_out);
  }
  __syncthreads();
  // |x|^2 of the shared block: one value per thread and step (thread 0 walking the cameras alone waited for 65
  // dependent loads at BASELINE configs[4])
  double x2 = 0.0;
  {
    const int cur0 = ctl->cur;
    for (int i = tid; i < P.C * 7; i += 256) {
      const int cc2 = i / 7;
      const double v = P.cam[((size_t)cur0 * P.C + cc2) * 8 + (i - cc2 * 7)];
      x2 += P.cam_fixed[cc2] ? 0.0 : v * v;
    }
    for (int i = tid; i < P.CK * kRigK; i += 256) {   // every intrinsic of a set that is in the problem counts in |x|
      const int ks = i / kRigK;
      const double v = P.intr[((size_t)cur0 * P.CK + ks) * 16 + (i - ks * kRigK)];
      x2 += P.kscol[ks] >= 0 ? v * v : 0.0;
    }
  }
  const double x2_shared = block_sum256(x2, s4);
  if (tid == 0) {
    LmCtl c = *ctl;
    const LmOpts o = *P.opts;
    const double xn2 = s_out[3] + x2_shared;
    lm_init(c, o, s_out[0], sqrt(xn2));
    *P.ctl = c;
    *P.ctl_next = c;
  }
}

// ---------------------------------------------------------------------------------------------
// elim: trust-region decision (every block, same answer; block 0 publishes it), then the elimination of the
// frame poses. One WAVE per frame, four frames per block iteration:
//   lanes 0..26 sum the frame block A = sum_groups H_ff (21 entries) and g_f (6) over the frame's groups;
//   every lane factors the damped 6x6 block in registers; lane k owns shared column k (and k + 64):
//   w = column of [H_fs | g_f] (scaled), z = L^-1 w -> staged in LDS, y = L^-T z -> Y (back-substitution);
//   the wave also adds its frame's entries of the shared diagonal blocks into per-lane accumulators;
//   then the four waves contract the 24 staged rows of Z on the matrix cores: tile pair (ti <= tj) of
//   the (SW x SW) product Z^T Z goes to wave (index mod 4), 6 k-steps of v_mfma_f64_16x16x4_f64.
// Partial row of a block: [nT tiles x 256 | ND direct sums | Cholesky failures | max |g_frame|].
// ---------------------------------------------------------------------------------------------
// NR = direct-sum accumulators per lane: 8 covers ND <= 512 (the usual rigs: <= 18 observed cameras with poses only, 3 with
// intrinsics), 24 the full range; the small variant exists because the kernel sits at the register limit.
// The elimination as a function: k_rig_elim (a launch of its own: trust-region decision, then the elimination) and the
// persistent per-solve kernel (PS: the decision is the control workgroup's -- which buffer holds the point to eliminate,
// the radius, whether this is the first elimination, and the Jacobi scales of the shared columns come as arguments).
// FM: the sweep was k_rig_sweep_frame -- a group's record is [G7 (28) | T (36)] (P.gcomp), the frame block comes summed (P.fsum).
// KC: the sweep was k_rig_sweep_k2 (intrinsics, compact records of kRigRecK doubles in P.gcomp: offsets kRk*).
template <bool HK, int NR, bool PS, bool FM = false, bool KC = false>
__device__ __forceinline__ void rig_elim_body(const RigDev& P, char* smem_raw, const int ps_cur, const double ps_radius, const bool ps_first,
                                              const double* ps_ss) {
  static_assert(!(HK && FM), "the frame form is the poses-only sweep's");
  static_assert(!KC || HK, "compact K records belong to the sweep with intrinsics");
  double* s_Z = reinterpret_cast<double*>(smem_raw);         // [24][ZS] staged Z rows of the four frames
  double* s_A = s_Z + 24 * P.ZS;                             // [4][32] frame block broadcast, per wave
  double* s_red = s_A + 4 * 32;                              // [4][1024] cross-wave reduction scratch
  // large variant (NR > 8): the direct-sum accumulators of a wave live in LDS, one slot per lane and register index --
  // as registers they pushed the kernel over the 512-VGPR limit (48 spilled VGPRs, scratch traffic in the frame loop)
  constexpr bool kLdsAcc = NR > 8;
  double* s_dacc = s_red + 4 * 1024;                         // [4][NR * 64] (large variant only)
  __shared__ double s_ss[kRigMaxS + 1];
  __shared__ double s16[16];
  __shared__ double s_tot[4];
  __shared__ double s_fg[8];
  __shared__ LmCtl s_ctl;
  int tid_ = threadIdx.x;
  if (PS) asm volatile("" : "+v"(tid_));   // (a fresh copy per call: the lane tables below are rebuilt every round of the persistent kernel instead of
                                              //  being hoisted out of its round loop and kept -- spilled -- across the sweep)
  const int tid = tid_, lane = tid & 63, wave = tid >> 6;
  const LmCtl* ctl = P.ctl;
  const int ctl_done = PS ? 0 : ctl->done, ctl_phase = PS ? 1 : ctl->phase;
  // what thread 0 needs for the trust-region decision, fetched now instead of behind the statistics barrier
  LmCtl c_in;
  LmOpts o_in;
  double sh0 = 0.0, sh1 = 0.0;
  if constexpr (!PS) { c_in = *ctl; o_in = *P.opts; sh0 = P.shared_stats[0]; sh1 = P.shared_stats[1]; }
  // ... and, frame form, the statistics rows themselves (one per frame: eight per thread up to 2048 frames): requested next
  // to the control block instead of behind it
  constexpr bool kPre = FM && !PS;
  d2 pre_g[kPre ? 8 : 1], pre_f[kPre ? 8 : 1];
  const bool pre = kPre && P.fmode && P.F <= 2048 && !P.comm;
  if constexpr (kPre) {
    if (pre) {
      const d2* gs2 = reinterpret_cast<const d2*>(P.gstats);
      const d2* fs2 = reinterpret_cast<const d2*>(P.fstats);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int64_t i = u * 256 + tid;
        pre_g[u] = i < P.F ? gs2[i] : d2{0.0, 0.0};
        pre_f[u] = i < P.F ? fs2[i] : d2{0.0, 0.0};
      }
    }
  }
  // ---- loads that do not depend on the trust-region decision go out first, under the statistics round trip: the
  // lane's static tables and the group slots of the block's first four frames
  const int CO = P.CO;
  const int64_t f_first = (int64_t)blockIdx.x * 4 + wave;
  const int gj_first = (f_first < P.F && lane < CO) ? P.fslot[f_first * CO + lane] : -1;
  // static (frame-independent) description of what this lane owns
  // frame-block entry of lane e < 27 (offset inside a group's AA tile); lanes holding a diagonal entry also
  // store the frame's Jacobi scale in the first elimination
  int a_off = 0, sp_i = -1;
  if (lane < 21) {
    int i = 0;
    while (tri(i + 1, 0) <= lane) ++i;
    const int j = lane - tri(i, 0);
    a_off = KC ? kRkHff + lane : (6 + i) * 16 + 6 + j;
    if (i == j) sp_i = i;
  } else if (lane < 27) {
    a_off = KC ? kRkHff + lane : (6 + (lane - 21)) * 16 + 12;
  }
  // shared columns of this lane: k = lane and lane + 64
  int c_kind[2], c_co[2], c_comp[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int k = lane + 64 * h;
    c_kind[h] = -1; c_co[h] = 0; c_comp[h] = 0;
    if (k < P.SW) {
      const int info = P.colinfo[k];
      c_kind[h] = (info >> 4) & 15; c_co[h] = info >> 8; c_comp[h] = info & 15;
    }
  }
  // direct-sum entries of this lane: e = lane + 64 r -> (observed camera, offset inside the group block)
  int d_ent[NR];   // (observed camera << 16) | offset, -1: nothing (packed: registers are scarce here)
  double dacc[kLdsAcc ? 1 : NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int e = lane + 64 * r;
    if (kLdsAcc) s_dacc[(wave * NR + r) * 64 + lane] = 0.0; else dacc[r] = 0.0;
    const int t = P.dent[e < P.ND ? e : 0];   // (unconditional load + select: a conditional load waits on its own)
    d_ent[r] = e < P.ND ? t : -1;
  }
  // tile pairs of this wave's accumulators
  int t_ij[kRigTilesPerWave];   // ti | tj << 8, wave-uniform (scalar registers)
#pragma unroll
  for (int u = 0; u < kRigTilesPerWave; ++u) {
    const int idx = 4 * u + wave, ic = idx < P.nT ? idx : 0;
    t_ij[u] = __builtin_amdgcn_readfirstlane((int)P.tile_i[ic] | ((int)P.tile_j[ic] << 8));
  }
  if (ctl_done || ctl_phase == 0) return;
  bool pending = false;
  if constexpr (!PS) {
  pending = ctl->cand_pending != 0;
  if (P.comm) {
    if (tid < 4) s_tot[tid] = P.vec_stats[tid];
    __syncthreads();
  } else {
    if (kPre && pre) rig_reduce_stats(P, pending && ctl->step_valid, s16, s_tot, pre_g, pre_f);
    else rig_reduce_stats(P, pending && ctl->step_valid, s16, s_tot);
  }
  if (tid == 0) {
    LmCtl c = c_in;
    const LmOpts& o = o_in;
    if (pending) {
      double step2 = s_tot[2], xn2 = s_tot[3];
      if (c.step_valid) { step2 += sh0; xn2 += sh1; }
      cc_iteration rec;
      const int len0 = c.log_len;
      lm_decide(c, o, &rec, s_tot[0], s_tot[1], step2, xn2);
      if (blockIdx.x == 0 && c.log_len != len0 && c.log_len <= P.log_cap) P.log[c.log_len - 1] = rec;
    }
    s_ctl = c;
    if (blockIdx.x == 0) *P.ctl_next = c;
  }
  }   // (!PS)
  if (tid < P.S) s_ss[tid] = (PS ? ps_ss : P.ss)[tid];
  for (int i = tid; i < 24 * P.ZS; i += 256) s_Z[i] = 0.0;   // padding columns stay zero
  __syncthreads();
  if (!PS && s_ctl.done) return;
  const int cur = PS ? ps_cur : s_ctl.cur;
  const double inv_radius = 1.0 / (PS ? ps_radius : s_ctl.radius);
  const double mn = P.opts->min_lm_diagonal, mx = P.opts->max_lm_diagonal;
  const bool first_elim = PS ? ps_first : (ctl->phase == 1 && s_ctl.iter == 0 && !pending);   // Jacobi scale of the frame blocks
  const bool jac = P.opts->jacobi_scaling != 0;
  const int SW = P.SW, S = P.S, ZS = P.ZS;
  const size_t gs = FM ? (size_t)64 : (KC ? (size_t)kRigRecK : (size_t)P.gstride);
  const double* blocks = (FM || KC) ? P.gcomp + (size_t)cur * P.NG * gs : P.gblocks + (size_t)cur * P.NG * gs;

  double c_ss[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) { const int k = lane + 64 * h; c_ss[h] = k < S ? s_ss[k] : (k < SW ? 1.0 : 0.0); }
  d4 acc[kRigTilesPerWave];
#pragma unroll
  for (int u = 0; u < kRigTilesPerWave; ++u) acc[u] = d4{0.0, 0.0, 0.0, 0.0};
  double gmax = 0.0, nfail = 0.0;
  double* As = s_A + wave * 32;

  int gj_next = gj_first;
  const int nr = (P.ND + 63) >> 6;   // direct-sum registers in use (uniform)
  for (int64_t fb = (int64_t)blockIdx.x * 4; fb < P.F; fb += (int64_t)gridDim.x * 4) {
    const int64_t f = fb + wave;
    // group of (frame, observed camera j) on lane j: fetched one pass ahead (-1 beyond the last frame)
    const int gj = gj_next;
    {
      const int64_t fn = f + (int64_t)gridDim.x * 4;
      gj_next = (fn < P.F && lane < CO) ? P.fslot[fn * CO + lane] : -1;
    }
    const bool live = __any(gj >= 0);   // the frame has observations (wave-uniform)
    if (live) {
      // the frame's Jacobi scale (overwritten below in the first elimination, which computes it)
      double sf[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) sf[i] = P.sp[f * 8 + i];
      // (the frame's quaternion at the point being eliminated: for the gradient norm, same round trip)
      const double* fqp = P.pose + ((size_t)cur * P.F + f) * 8;
      const double fq0 = fqp[0], fq1 = fqp[1], fq2 = fqp[2], fq3 = fqp[3];
      // ---- loads: frame block entries (lanes < 27, summed over the groups), column data, direct entries
      double a_e = 0.0;
      if (FM) {
        a_e = P.fsum[((size_t)cur * P.F + f) * 32 + (lane < 27 ? lane : 0)];   // (the sweep summed the frame's groups)
        if (lane >= 27) a_e = 0.0;
      } else {
        for (int j0 = 0; j0 < CO; j0 += 8) {
          double v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int g = j0 + u < CO ? __builtin_amdgcn_readlane(gj, j0 + u) : -1;
            v[u] = (g >= 0 && lane < 27) ? blocks[(size_t)g * gs + a_off] : 0.0;
          }
          a_e += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        }
      }
      double w[2][6];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int i = 0; i < 6; ++i) w[h][i] = 0.0;
        const int gsel = __shfl(gj, c_co[h], 64);
        if ((c_kind[h] == 0 || c_kind[h] == 1) && gsel >= 0) {
          const double* G = blocks + (size_t)gsel * gs;
#pragma unroll
          for (int i = 0; i < 6; ++i)
            w[h][i] = FM ? G[28 + c_comp[h] * 6 + i]
                    : KC ? (c_kind[h] == 0 ? G[kRkT + c_comp[h] * 6 + i] : G[kRkFK + i * 9 + c_comp[h]])
                         : (c_kind[h] == 0 ? G[c_comp[h] * 16 + 6 + i] : G[256 + (6 + i) * 16 + c_comp[h]]);
        }
      }
      if (HK && P.kmode == RIG_K_SHARED) {
        // columns of the intrinsics shared by all cameras: sum of the frame's groups. Four groups per round trip, unconditional
        // loads (group 0 stands in for a camera that does not see the frame) and selects: written as a loop over the groups with
        // a `continue`, every group's six loads waited for on their own -- CO dependent round trips per frame (round 5: the
        // elimination at 8 x 2000 x 500 with shared intrinsics 43 -> ... us)
        for (int j0 = 0; j0 < CO; j0 += 4) {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            if (c_kind[h] != 2) continue;   // (nine lanes own such a column; the others skip the batch)
            double t[4][6];
            bool ok[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int g = j0 + u < CO ? __builtin_amdgcn_readlane(gj, j0 + u < CO ? j0 + u : 0) : -1;
              ok[u] = g >= 0;
              const double* G = blocks + (size_t)(ok[u] ? g : 0) * gs + (KC ? kRkFK : 256);
#pragma unroll
              for (int i = 0; i < 6; ++i) t[u][i] = KC ? G[i * 9 + c_comp[h]] : G[(6 + i) * 16 + c_comp[h]];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
              for (int i = 0; i < 6; ++i) w[h][i] += ok[u] ? t[u][i] : 0.0;
          }
        }
      }
      // direct entries, eight registers per round trip: unconditional loads (group 0 stands in where a lane has
      // nothing to fetch) followed by selects -- a load inside a lane-dependent `if` gets a wait of its own
      constexpr int DB = NR <= 8 ? 8 : 4;   // loads per round trip (the large variant has no registers to spare)
#pragma unroll
      for (int r0 = 0; r0 < NR; r0 += DB) {
        if (r0 < nr) {   // (uniform)
          double dx[DB];
          bool dk[DB];
#pragma unroll
          for (int u = 0; u < DB; ++u) {
            const int t = d_ent[r0 + u];
            const int g = __shfl(gj, t < 0 ? 0 : (t >> 16), 64);
            dk[u] = t >= 0 && g >= 0;
            dx[u] = blocks[(size_t)(dk[u] ? g : 0) * gs + (t & 0xffff)];
          }
#pragma unroll
          for (int u = 0; u < DB; ++u) {
            if (kLdsAcc) s_dacc[(wave * NR + r0 + u) * 64 + lane] += dk[u] ? dx[u] : 0.0;   // (own slot: no conflict)
            else dacc[r0 + u] += dk[u] ? dx[u] : 0.0;
          }
        }
      }
      // ---- broadcast the frame block
      if (lane < 27) As[lane] = a_e;
      wave_lds_fence();
      double A[27];
#pragma unroll
      for (int i = 0; i < 27; ++i) A[i] = As[i];
      wave_lds_fence();
      if (first_elim) {
#pragma unroll
        for (int i = 0; i < 6; ++i) sf[i] = jac ? 1.0 / (1.0 + sqrt(A[tri(i, i)])) : 1.0;
        if (sp_i >= 0) P.sp[f * 8 + sp_i] = jac ? 1.0 / (1.0 + sqrt(a_e)) : 1.0;
      }
      double L[21], Li[6];
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) L[tri(i, j)] = sf[i] * A[tri(i, j)] * sf[j];
#pragma unroll
      for (int i = 0; i < 6; ++i) L[tri(i, i)] += clampd(L[tri(i, i)], mn, mx) * inv_radius;
      bool ok = true;
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
      if (!ok) nfail += 1.0;
      {   // the frame's share of Ceres' gradient_max_norm, ||x - Plus(x, -g)||_inf (pose_grad_proj_max, cc_common.hpp)
        const double q4[4] = {fq0, fq1, fq2, fq3};
        gmax = fmax(gmax, pose_grad_proj_max(q4, &A[21]));
      }
      // ---- columns: z = L^-1 w, y = L^-T z
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int k = lane + 64 * h;
        if (k < SW) {
          double z[6], y[6];
#pragma unroll
          for (int i = 0; i < 6; ++i) {
            double a = c_kind[h] == 3 ? sf[i] * A[21 + i] : sf[i] * w[h][i] * c_ss[h];
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
            s_Z[(wave * 6 + i) * ZS + k] = z[i];
            P.Y[((size_t)f * 6 + i) * SW + k] = y[i];
          }
        }
      }
    } else {
      for (int k = lane; k < SW; k += 64)
#pragma unroll
        for (int i = 0; i < 6; ++i) s_Z[(wave * 6 + i) * ZS + k] = 0.0;
    }
    __syncthreads();
    // ---- Schur products of the four staged frames on the matrix cores
#pragma unroll
    for (int u = 0; u < kRigTilesPerWave; ++u) {
      const int idx = 4 * u + wave;
      if (idx < P.nT) {
        const int ti = t_ij[u] & 255, tj = t_ij[u] >> 8;
        const int col = lane & 15, sub = lane >> 4;
#pragma unroll
        for (int ksx = 0; ksx < 6; ++ksx) {
          const double* row = s_Z + (4 * ksx + sub) * ZS;
          const double a = row[16 * ti + col], b = row[16 * tj + col];
          acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[u], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }

  // ---- one partial row per block
  double* prow = P.partial + (size_t)blockIdx.x * P.PC;
  // tiles: each belongs to exactly one wave. C/D layout: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
  for (int u = 0; u < kRigTilesPerWave; ++u) {
    const int idx = 4 * u + wave;
    if (idx < P.nT) {
#pragma unroll
      for (int r = 0; r < 4; ++r) prow[(size_t)idx * 256 + ((lane >> 4) + 4 * r) * 16 + (lane & 15)] = acc[u][r];
    }
  }
  // direct sums: four waves each hold partial sums of the same entries
  if (kLdsAcc) {
    __syncthreads();
    for (int e = tid; e < P.ND; e += 256)
      prow[P.pc_dir + e] = (s_dacc[e] + s_dacc[NR * 64 + e]) + (s_dacc[2 * NR * 64 + e] + s_dacc[3 * NR * 64 + e]);
  }
#pragma unroll
  for (int r0 = 0; r0 < (kLdsAcc ? 0 : NR); r0 += 16) {
    if (r0 * 64 >= P.ND) continue;   // (uniform) nothing left in this chunk
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (r0 + r < NR) s_red[wave * 1024 + r * 64 + lane] = dacc[r0 + r];
    __syncthreads();
    for (int i = tid; i < 1024; i += 256) {
      const int e = r0 * 64 + i;
      if (e < P.ND) prow[P.pc_dir + e] = (s_red[i] + s_red[1024 + i]) + (s_red[2048 + i] + s_red[3072 + i]);
    }
  }
  gmax = wave_max(gmax);
  if (lane == 0) { s_fg[wave] = gmax; s_fg[4 + wave] = nfail; }
  __syncthreads();
  if (tid == 0) {
    prow[P.pc_fail] = (s_fg[4] + s_fg[5]) + (s_fg[6] + s_fg[7]);
    prow[P.pc_gmax] = fmax(fmax(s_fg[0], s_fg[1]), fmax(s_fg[2], s_fg[3]));
  }
}

template <bool HK, int NR, bool FM = false, bool KC = false>
__global__ __launch_bounds__(256) void k_rig_elim(RigDev P) {
  rig_progress(P, RIG_PROG_ELIM);
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  rig_elim_body<HK, NR, false, FM, KC>(P, smem_raw, 0, 1.0, false, nullptr);
}

// ---------------------------------------------------------------------------------------------
// The solve step, run by ONE block of 256 threads on the reduced sums (vec: [nT tiles | direct | fail | 0],
// then one max-gradient slot per rank): assembles the damped reduced system in LDS (lower triangle), dense
// Cholesky (all four waves, one barrier per column), then wave 0 alone: substitutions with lane i owning
// b[i] and b[i + 64] (cross-lane values through v_readlane, no barrier), gradient / radius tests
// (lm_finalize order), camera and intrinsics candidates, control block.
// SRC: where a reduced value comes from. 0: plain loads of P.vec (after an all-reduce, RCCL route);
// 1: write-through stores of the reduce blocks, read with sc1 loads (last-block-done, single GPU);
// 2: the ranks' mailbox slots, polled and added in rank order (last-block-done, mailbox exchange).
// Tried and dropped in round 1 (S = 24): a single-wave factorisation through LDS (33 us vs 20) and a
// register-tiled one with only the pivot column crossing threads through LDS (23 us).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double readlane_d(double x, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(x), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(x), l);
  return __hiloint2double(hi, lo);
}



// shared step: write-through, so that workgroups of the same launch can read it behind a flag (sc1 loads)
__device__ __forceinline__ void store_ds(double* p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct RigVal {   // reader of reduced value e
  const RigDev& P; unsigned long long epoch; long long t0; int* s_ok; const double* lds_vec;
  template <int SRC>
  __device__ __forceinline__ double get(int e) const {
    if (SRC == 3) return lds_vec[e];   // (persistent kernels: the control workgroup keeps the reduced row in LDS)
    if (SRC == 0) return P.vec[e];
    if (SRC == 1)
      return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(P.vec) + e,
                                                              __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    return p2p_poll_sum(P.x, 0, epoch, P.rank, P.nranks, e, t0, s_ok);
  }
};

// writes the camera / intrinsics records the sweep reads, from the current parameters plus (step_ok) the step
// x (LDS, scaled shared step with the sign of b: the step is -x * ss). Returns this thread's share of
// (step^2, |x_cand|^2) of the shared block. All 256 threads call.
__device__ __forceinline__ void rig_candidates(const RigDev& P, const double* x, const double* ss, bool have_step, int cur, int dst,
                                               double& step2, double& xn2) {
  const int tid = threadIdx.x;
  step2 = 0.0; xn2 = 0.0;
  for (int c = tid; c < P.C; c += 256) {
    const double* pc = P.cam + ((size_t)cur * P.C + c) * 8;
    double q[4] = {pc[0], pc[1], pc[2], pc[3]}, t[3] = {pc[4], pc[5], pc[6]};
    double dc[6] = {0, 0, 0, 0, 0, 0};
    const int p0 = P.pcol[c];
    if (have_step) {
      if (p0 >= 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) dc[k] = -x[p0 + k] * ss[p0 + k];
        double qn[4];
        quat_plus(q, dc, qn);
#pragma unroll
        for (int k = 0; k < 4; ++k) { const double d = qn[k] - q[k]; step2 += d * d; q[k] = qn[k]; }
#pragma unroll
        for (int k = 0; k < 3; ++k) { const double tn = t[k] + dc[3 + k]; const double d = tn - t[k]; step2 += d * d; t[k] = tn; }
        xn2 += q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3] + t[0] * t[0] + t[1] * t[1] + t[2] * t[2];
      }
      double* pd = P.cam + ((size_t)dst * P.C + c) * 8;
#pragma unroll
      for (int k = 0; k < 4; ++k) pd[k] = q[k];
#pragma unroll
      for (int k = 0; k < 3; ++k) pd[4 + k] = t[k];
    }
    double R[9];
    quat_to_R(q, R);
    double* rec = P.camrec + c * 32;
#pragma unroll
    for (int k = 0; k < 9; ++k) rec[k] = R[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) rec[9 + k] = t[k];
#pragma unroll
    for (int k = 0; k < 6; ++k) rec[12 + k] = dc[k];
  }
  // extension: candidate intrinsics, thread (set, j). Every intrinsic of a set that is in the problem counts in
  // |x| (cf. IntrinsicsProblem), frozen ones do not move.
  for (int i = tid; i < P.CK * kRigK; i += 256) {
    const int s = i / kRigK, j = i - s * kRigK;
    const int k0 = P.kscol[s];
    const double kc = P.intr[((size_t)cur * P.CK + s) * 16 + j];
    double dk = 0.0;
    if (have_step && k0 >= 0 && !((P.kmask[s] >> j) & 1u)) dk = -x[k0 + j] * ss[k0 + j];
    const double kn = kc + dk;
    if (have_step) {
      P.intr[((size_t)dst * P.CK + s) * 16 + j] = kn;
      if (k0 >= 0) { step2 += dk * dk; xn2 += kn * kn; }
    }
    P.krec[s * 32 + j] = kn;
    P.krec[s * 32 + 16 + j] = dk;
  }
}

// One panel (columns j0 .. j0 + nc - 1, nc <= 8) of the Cholesky factorisation of the S x S system in LDS, on ONE
// wave: lane i keeps the panel's entries of row i (and, TWO, of row i + 64) in registers. A column step takes the
// pivot with v_readlane, scales the column, puts it into the LDS vector `colbuf` and reads the multipliers of the
// panel's remaining columns back as uniform-address LDS reads (LDS operations of one wave execute in order: no
// barrier). The forward substitution of the right-hand side (b0 / b1: rows i / i + 64) rides along; v0 / v1 collect
// 1 / L_ii. The loop over panels is rolled (rig_solve_block), so the code stays a few hundred instructions:
// the fully unrolled whole-matrix-in-registers form this replaces ran 20 KB of straight-line code once per launch and
// was bound by instruction fetch (profiles/r02/rig_reduce_breakdown.txt).
template <bool TWO>
__device__ __forceinline__ void chol_panel(double* A, int S, int LD, int j0, int nc, double* colbuf, double& b0, double& b1,
                                           double& v0, double& v1, bool& okw) {
  const int lane = threadIdx.x & 63, i0 = lane, i1 = lane + 64;
  double p0[8], p1[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int col = j0 + c;
    // (unconditional loads from a clamped row, then a select: conditional loads become one branch and one wait each)
    const double x0 = A[(size_t)(i0 < S ? i0 : S - 1) * LD + col];
    const double x1 = TWO ? A[(size_t)(i1 < S ? i1 : S - 1) * LD + col] : 0.0;
    p0[c] = (c < nc && i0 < S && col <= i0) ? x0 : 0.0;
    p1[c] = (TWO && c < nc && i1 < S && col <= i1) ? x1 : 0.0;
  }
  // The pivot of the NEXT column is taken ahead of the column's own update (two lane reads and one FMA, the very
  // operation the update performs on that entry, so the value is the same bit for bit): its reciprocal square root is
  // then computed while the LDS round trip of the multipliers is in flight instead of behind it.
  double d = (!TWO || j0 < 64) ? readlane_d(p0[0], j0 & 63) : readlane_d(p1[0], j0 & 63);
  double inv = rsqrt_pos(d);
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    if (c < nc) {   // (uniform)
      const int col = j0 + c;
      okw = okw && (d > 0.0) && isfinite(d);
      const double l0 = i0 == col ? d * inv : (i0 > col ? p0[c] * inv : 0.0);
      const double l1 = TWO ? (i1 == col ? d * inv : (i1 > col ? p1[c] * inv : 0.0)) : 0.0;
      const double inv_c = inv;
      p0[c] = l0;
      if (i0 == col) v0 = inv;
      colbuf[i0] = l0;
      if (TWO) { p1[c] = l1; if (i1 == col) v1 = inv; colbuf[i1] = l1; }
      wave_lds_fence();
      if (c + 1 < 8 && c + 1 < nc) {
        const int cn = col + 1;
        const double ln = (!TWO || cn < 64) ? readlane_d(l0, cn & 63) : readlane_d(l1, cn & 63);          // L[cn][col]
        const double pn = (!TWO || cn < 64) ? readlane_d(p0[c + 1], cn & 63) : readlane_d(p1[c + 1], cn & 63);
        d = fma(-ln, ln, pn);
        inv = rsqrt_pos(d);
      }
      // (no test against nc here: columns beyond the panel's end are computed on whatever colbuf holds and never
      // stored -- a uniform branch per column would put every LDS read behind its own wait)
      double m[8];
#pragma unroll
      for (int c2 = c + 1; c2 < 8; ++c2) m[c2] = colbuf[j0 + c2];   // L[j0 + c2][col], same address in every lane
#pragma unroll
      for (int c2 = c + 1; c2 < 8; ++c2) {
        p0[c2] = fma(-l0, m[c2], p0[c2]);
        if (TWO) p1[c2] = fma(-l1, m[c2], p1[c2]);
      }
      // forward substitution: y_col = b_col / L_col,col, b_i -= L_i,col y_col (i > col)
      const double yj = ((!TWO || col < 64) ? readlane_d(b0, col & 63) : readlane_d(b1, col & 63)) * inv_c;
      b0 = i0 == col ? yj : (i0 > col ? b0 - l0 * yj : b0);
      if (TWO) b1 = i1 == col ? yj : (i1 > col ? b1 - l1 * yj : b1);
      wave_lds_fence();
    }
  }
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int col = j0 + c;
    if (c < nc && i0 < S && col <= i0) A[(size_t)i0 * LD + col] = p0[c];
    if (TWO && c < nc && i1 < S && col <= i1) A[(size_t)i1 * LD + col] = p1[c];
  }
}

// Backward substitution L^T x = y on the same wave (lane i: rows i and, TWO, i + 64; v = 1 / L_ii). The factor entries
// a lane needs do not depend on the running solution: they are fetched eight steps ahead and pre-multiplied by
// 1 / L_jj, so that a step is one lane read and one FMA on the dependent chain; x_i = b_i / L_ii is formed at the end.
template <bool TWO>
__device__ __forceinline__ void chol_backward(const double* A, int S, int LD, double& b0, double& b1, double v0, double v1) {
  const int lane = threadIdx.x & 63, i0 = lane, i1 = lane + 64;
  if (TWO) {
    // Rows S - 1 .. 64 first: their pivots live in b1 and nowhere else, so a step is two lane reads + two FMAs with nothing to
    // select (round 6: one loop over all rows chose between b0 and b1 on every step -- four lane reads and two scalar selects on
    // the dependent chain, 114 times). Same products, same order: same bits.
    for (int j0 = S - 1; j0 >= 64; j0 -= 8) {
      double a0[8], a1[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int j = j0 - u, jr = j >= 64 ? j : 64;
        const double x0 = A[(size_t)jr * LD + i0];
        const double x1 = A[(size_t)jr * LD + (i1 < LD ? i1 : 0)];
        const double vj = readlane_d(v1, jr - 64);
        a0[u] = j >= 64 ? x0 * vj : 0.0;              // (every row i0 < 64 <= j)
        a1[u] = (j >= 64 && i1 < j) ? x1 * vj : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int jr = j0 - u >= 64 ? j0 - u : 64;   // (steps below row 64 multiply by the zeros selected above)
        const double bj = readlane_d(b1, jr - 64);
        b0 -= a0[u] * bj;
        b1 -= a1[u] * bj;
      }
    }
    b1 *= v1;
  }
  for (int j0 = TWO ? 63 : S - 1; j0 >= 0; j0 -= 8) {
    double a0[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int j = j0 - u, jr = j >= 0 ? j : 0;
      // (unconditional loads from row jr, then a select: a conditional load is a branch and a wait of its own)
      const double x0 = A[(size_t)jr * LD + i0];
      const double vj = readlane_d(v0, jr);
      a0[u] = (j >= 0 && i0 < j) ? x0 * vj : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int jr = j0 - u >= 0 ? j0 - u : 0;   // (steps below row 0 multiply by the zeros selected above)
      const double bj = readlane_d(b0, jr);      // final: rows > j are done
      b0 -= a0[u] * bj;
    }
  }
  b0 *= v0;
}

// Trailing update A[t0.., t0..] -= P P^T (P = the panel's nc <= 4 KS columns from j0, rows t0..S-1) ON THE MATRIX PIPE: the
// lower 16 x 16 tiles of the trailing triangle are dealt to the four waves, KS v_mfma_f64_16x16x4_f64 per tile; per element 3
// LDS operations instead of the 18 of the element-wise form (S = 114: the trailing updates were a third of the solve
// step). Two tiles per round: every LDS read of both (operands and the elements to update) is issued before the first
// matrix instruction -- one LDS round trip and one matrix-pipe latency per pair instead of per tile. All 256 threads call.
// RE: one past the last ROW updated -- S, or S + 1 when the right-hand side rides along as row S of the matrix (chol_block4).
// PART / w / nw (round 6, look-ahead of the eight-column panels): which tiles and on which waves. PART 0: every tile; 1: the tiles
// of the FIRST tile column only (columns t0 .. t0 + 15: the next two panels' columns); 2: all the others. The caller's wave is
// number w of the NWV that take part (tiles dealt round robin, two per round as before).
template <int KS, int PART = 0, int NWV = 4>
__device__ __forceinline__ void chol_trail_mfma(double* A, int S, int LD, int j0, int nc, int t0, int RE, int w = -1) {
  const int tid = threadIdx.x;
  const int nt = RE - t0, n16 = (nt + 15) >> 4;
  const int ntile = PART == 0 ? n16 * (n16 + 1) / 2 : (PART == 1 ? n16 : (n16 - 1) * n16 / 2);
  const int wv = w >= 0 ? w : tid >> 6, ln = tid & 63, kq = ln >> 4, c16 = ln & 15;
  for (int tb = wv; tb < ntile; tb += 2 * NWV) {
    double am[2][KS], bm[2][KS], old[2][4];
    int at[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int t = tb + NWV * u;
      const bool live = t < ntile;
      const int tc = live ? t : 0;
      int ti, tj;
      if (PART == 1) { ti = tc; tj = 0; }
      else {
        ti = (int)((sqrtf(8.0f * (float)tc + 1.0f) - 1.0f) * 0.5f);
        ti = ti * (ti + 1) / 2 > tc ? ti - 1 : ti;
        ti = (ti + 1) * (ti + 2) / 2 <= tc ? ti + 1 : ti;
        tj = tc - ti * (ti + 1) / 2;
        if (PART == 2) { ti += 1; tj += 1; }   // (the lower triangle without its first column is a lower triangle again)
      }
      const int R = t0 + 16 * ti, Cc = t0 + 16 * tj;
      const bool ina = live && R + c16 < RE, inb = live && Cc + c16 < S;
      const int ra = ina ? R + c16 : S - 1, rb = inb ? Cc + c16 : S - 1;      // (unconditional loads, then selects)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int kc = 4 * ks + kq;                                            // panel column of this lane in k-step ks
        const int cc = j0 + kc < S ? j0 + kc : S - 1;                          // (stays inside the LDS block; selected away)
        const double xa = A[(size_t)ra * LD + cc], xb = A[(size_t)rb * LD + cc];
        am[u][ks] = (ina && kc < nc) ? xa : 0.0;
        bm[u][ks] = (inb && kc < nc) ? xb : 0.0;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = R + kq + 4 * r, col = Cc + c16;
        at[u][r] = (live && row < RE && col < S && col <= row) ? row * LD + col : -1;
        old[u][r] = A[at[u][r] >= 0 ? at[u][r] : 0];
      }
    }
    d4 T0 = {0.0, 0.0, 0.0, 0.0}, T1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      T0 = __builtin_amdgcn_mfma_f64_16x16x4f64(am[0][ks], bm[0][ks], T0, 0, 0, 0);
      T1 = __builtin_amdgcn_mfma_f64_16x16x4f64(am[1][ks], bm[1][ks], T1, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (at[0][r] >= 0) A[at[0][r]] = old[0][r] - T0[r];
      if (at[1][r] >= 0) A[at[1][r]] = old[1][r] - T1[r];
    }
  }
}

// Cholesky of the damped reduced system FOUR columns at a time with LOOK-AHEAD (round 4; medium systems, 24 < S <= 63:
// BASELINE configs[4] is S = 42), the right-hand side riding along as row S of the matrix. One barrier per block of four:
//   wave 0 owns the serial chain. Lane l holds the four entries of row j0 + l in the block's columns, fully updated; a
//     column is pivot (lane read) -> rsqrt -> scale -> up to three updates of (lane read + FMA) -- no LDS round trip and
//     no barrier on the chain. It stores the panel, and behind the barrier applies THIS panel's rank-4 update to the NEXT
//     block's four columns itself (sixteen FMAs per row, multipliers by uniform LDS reads) and goes straight on factoring;
//   waves 1..3 meanwhile give the REST of the trailing matrix (columns beyond the next block, the right-hand side's row
//     included) the same rank-4 update on the matrix pipe: one v_mfma_f64_16x16x4_f64 per 16 x 16 tile, operands straight
//     from LDS, tiles fixed for the whole factorisation (addresses and validity computed once per lane).
// What was measured on the way (scripts/time_chol.py, one workgroup, hot, S = 42, shader cycles at 2.41 GHz): round 3's
// eight-column panels on wave 0 + trailing updates 32.2 k (13.4 us; in the solving block 9.0 + 4.8 us); sixteen-column
// register-row panels with lane reads 13.0 + 3.0 us in the solving block (360 dependent lane-read / FMA triples per panel
// on one wave); four-column blocks with the 4 x 4 diagonal block in closed form on every thread, two barriers and the
// trailing update on all four waves 38.1 k, of which the trailing update 20 k (tiles re-anchored per step) / 15 k (fixed
// tiles) -- a dependent fp64 instruction costs ~20 cycles when a SIMD has one wave to run, so what counts is the LENGTH of
// the dependent chain (~12 instructions per column: 42 x 240 cycles = 4.2 us is the floor), and everything that can
// leave the chain's wave must. s_inv[j] receives 1 / L_jj (backward substitution). All 256 threads call; returns whether
// every pivot was positive and finite (valid in every thread).
__device__ __forceinline__ bool chol_block4(double* A, int S, int LD, double* s_inv) {
  const int tid = threadIdx.x, wv = tid >> 6, ln = tid & 63, kq = ln >> 4, c16 = ln & 15;
  __shared__ int s_okb;
  if (tid == 0) s_okb = 1;
  // ---- waves 1..3: the tiles of the trailing update, dealt round robin; fixed rows / columns 16 ti.. / 16 tj.. (ti >= tj)
  const int n16 = (S + 1 + 15) >> 4, ntile = n16 * (n16 + 1) / 2;
  constexpr int kMaxT = 4;   // tiles per wave: ntile <= 10 over three waves (S <= 63)
  int ra[kMaxT], rb[kMaxT], rowmin[kMaxT], colmin[kMaxT], e0[kMaxT];
  bool ina[kMaxT], inb[kMaxT], live[kMaxT];
#pragma unroll
  for (int u = 0; u < kMaxT; ++u) {
    const int t = (wv - 1) + 3 * u;
    live[u] = wv > 0 && t < ntile;
    const int tc = live[u] ? t : 0;
    int ti = (int)((sqrtf(8.0f * (float)tc + 1.0f) - 1.0f) * 0.5f);
    ti = ti * (ti + 1) / 2 > tc ? ti - 1 : ti;
    ti = (ti + 1) * (ti + 2) / 2 <= tc ? ti + 1 : ti;
    const int tj = tc - ti * (ti + 1) / 2;
    const int R = 16 * ti, Cc = 16 * tj;
    ina[u] = live[u] && R + c16 <= S;
    inb[u] = live[u] && Cc + c16 < S;
    ra[u] = (ina[u] ? R + c16 : S) * LD;
    rb[u] = (inb[u] ? Cc + c16 : S - 1) * LD;
    rowmin[u] = R;                        // tile rows R + kq + 4 r, column Cc + c16
    colmin[u] = Cc + c16;
    e0[u] = (R + kq) * LD + Cc + c16;     // element r of this lane: e0 + 4 r LD
  }
  // ---- wave 0: lane l is ROW l of the matrix for the whole factorisation (row S: the right-hand side); x = its entries in the
  // current block's columns, fully updated
  double x[4] = {0.0, 0.0, 0.0, 0.0};
  bool ok = true;
  const double* Row = A + (size_t)(ln <= S ? ln : S) * LD;
  if (wv == 0) {
    const int nb0 = S < 4 ? S : 4;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const double v = Row[c < nb0 ? c : 0];
      x[c] = (ln <= S && c < nb0 && c <= ln) ? v : 0.0;
    }
  }
  for (int j0 = 0; j0 < S; j0 += 4) {
    const int nb = S - j0 < 4 ? S - j0 : 4, t0 = j0 + nb;
    const int nbn = S - t0 < 4 ? S - t0 : 4;   // width of the next block (<= 0: there is none)
    if (wv == 0) {
      // ---- the block's columns, one after the other
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (c < nb) {   // (uniform)
          const double d = readlane_d(x[c], j0 + c);
          ok = ok && (d > 0.0) && isfinite(d);
          const double inv = rsqrt_pos(d);
          const double y = x[c] * inv;            // lane j0 + c: d * inv = L_cc; lanes above it: not part of the column
          x[c] = y;
          if (ln == j0 + c) s_inv[j0 + c] = inv;
#pragma unroll
          for (int c2 = c + 1; c2 < 4; ++c2) x[c2] = fma(-y, readlane_d(y, (j0 + c2) & 63), x[c2]);
        }
      }
      if (ln <= S) {
        double* W = A + (size_t)ln * LD + j0;
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (c < nb && j0 + c <= ln) W[c] = x[c];
      }
    }
    __syncthreads();   // panel j0 is in LDS; waves 1..3 have finished the previous block's trailing update
    if (t0 >= S) break;
    if (wv == 0) {
      // ---- look-ahead: this panel's update of the NEXT block's columns. The row's entries there (final but for this
      // panel: the barrier) and the sixteen multipliers L[t0 + c][j0 + k] (uniform addresses) come in ONE LDS round trip; the
      // row's own panel entries are the registers x[] (lane = row for the whole factorisation). (Multipliers by lane reads
      // instead, with the products under the round trip of the four entries: 21.4 k cycles against 19.9 k -- a lane read
      // into a scalar register followed by its use costs more than a broadcast LDS read.)
      double xn[4], m[4][4];
#pragma unroll
      for (int c = 0; c < 4; ++c) xn[c] = Row[(c < nbn ? t0 + c : 0)];
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int k = 0; k < 4; ++k) m[c][k] = A[(size_t)(t0 + (c < nbn ? c : 0)) * LD + j0 + (k < nb ? k : 0)];   // L[t0 + c][j0 + k]: uniform address, one round trip for all twenty
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        double a = xn[c];
#pragma unroll
        for (int k = 0; k < 4; ++k) a = fma(-(k < nb ? x[k] : 0.0), m[c][k], a);
        xn[c] = a;
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) x[c] = (ln <= S && c < nbn && t0 + c <= ln) ? xn[c] : 0.0;
    } else if (t0 + 4 < S) {
      // ---- waves 1..3: rank-nb update of the rest, columns >= t0 + 4 (the next block's are wave 0's), rows up to S
      const int kc = j0 + (kq < nb ? kq : 0);   // the lane's panel column (one k-step: column kq)
#pragma unroll
      for (int u0 = 0; u0 < kMaxT; u0 += 2) {
        if ((live[u0] && rowmin[u0] + 15 >= t0 + 4) || (u0 + 1 < kMaxT && live[u0 + 1] && rowmin[u0 + 1] + 15 >= t0 + 4)) {   // (uniform; tiles wholly above the corner are finished)
          double am[2], bm[2], old[2][4];
#pragma unroll
          for (int v = 0; v < 2; ++v) {
            const int u = u0 + v;
            const double xa = A[ra[u] + kc], xb = A[rb[u] + kc];
            am[v] = (ina[u] && kq < nb) ? xa : 0.0;
            bm[v] = (inb[u] && kq < nb) ? xb : 0.0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int e = e0[u] + 4 * r * LD;
              old[v][r] = A[(live[u] && e < (S + 1) * LD) ? e : 0];
            }
          }
          d4 T0 = {0.0, 0.0, 0.0, 0.0}, T1 = {0.0, 0.0, 0.0, 0.0};
          T0 = __builtin_amdgcn_mfma_f64_16x16x4f64(am[0], bm[0], T0, 0, 0, 0);
          T1 = __builtin_amdgcn_mfma_f64_16x16x4f64(am[1], bm[1], T1, 0, 0, 0);
#pragma unroll
          for (int v = 0; v < 2; ++v) {
            const int u = u0 + v;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int row = rowmin[u] + kq + 4 * r, col = colmin[u];
              if (live[u] && col >= t0 + 4 && col < S && row <= S && col <= row) A[e0[u] + 4 * r * LD] = old[v][r] - (v == 0 ? T0[r] : T1[r]);
            }
          }
        }
      }
    }
  }
  if (wv == 0 && ln == 0 && !ok) s_okb = 0;
  __syncthreads();
  return s_okb != 0;
}

template <int SRC>
__device__ void rig_solve_block(const RigDev& P, double* smem, const LmCtl* cn_in = nullptr, const double* vec_lds = nullptr, unsigned flag_epoch = 0u) {   // cn_in: the persistent kernels' control block (LDS); vec_lds: their reduced row (SRC 3)
  const int S = P.S, LD = (S + 1) | 1;   // odd row stride: a column walks all LDS banks
  double* A = smem;                       // [S][LD] lower triangle of the reduced system
  double* s_b = A + (size_t)S * LD;       // [128] right-hand side, then the solution x
  double* s_gs = s_b + 128;               // [128] unscaled shared gradient
  double* s_hd = s_gs + 128;              // [128] diagonal of the scaled H_ss
  double* s_inv = s_hd + 128;             // [128] 1 / L_jj
  double* s_ss = s_inv + 128;             // [128] Jacobi scale of the shared block
  __shared__ int s_ok, s_cholok, s_stepok, s_go;
  __shared__ double s4[4];
  __shared__ double s8[8];
  __shared__ LmCtl s_c;
  const int tid = threadIdx.x, lane = tid & 63;
  const LmCtl* cn = cn_in ? cn_in : P.ctl_next;
  const int cur = cn->cur, dst = cur ^ 1;
  const double radius = cn->radius;
  const LmOpts o = *P.opts;
  RigVal val{P, SRC == 2 ? P.x.seq[0] + 1ull : 0ull, wall_clock64(), &s_ok, vec_lds};
  if (tid == 0) { s_ok = 1; s_cholok = 1; s_stepok = 0; s_go = 0; s_c = *cn; }
  for (int i = tid; i < S * LD; i += 256) A[i] = 0.0;
  if (tid < 128) { s_b[tid] = 0.0; s_gs[tid] = 0.0; s_hd[tid] = 0.0; s_inv[tid] = 0.0; s_ss[tid] = tid < S ? P.ss[tid] : 0.0; }
  const int pin = tid < S ? P.colpin[tid] : -1;
  const bool pinned = pin >= 0 && ((P.kmask[pin >> 4] >> (pin & 15)) & 1u) != 0;
  __syncthreads();
  // ---- 1. one pass over the reduced values (loads batched eight deep: one round trip per batch, not per value;
  // where a value goes comes from host-built tables, fetched in the same round trip): per-camera sums of the
  // shared-block entries -> scaled H_ss (lower triangle) and unscaled gradient, minus the Schur products Z^T Z.
  // An element of A gets at most two contributions (one of each kind), added with LDS atomics onto zero: x + y is
  // commutative, so the result does not depend on who comes first. An intrinsics set shared by several cameras
  // is summed along its chain (dir_next), in camera order, by the first camera's thread.
  constexpr int NB = SRC == 2 ? 1 : 8;   // (a mailbox read is a polling loop of its own: no batching there)
  const double fail = val.get<SRC>(P.pc_fail);
  const double gm_r = (tid < P.nranks && tid < 32) ? val.get<SRC>(P.PC + tid) : 0.0;
  for (int e0 = 0; e0 < P.ND; e0 += NB * 256) {
    double v[NB];
    int d[NB], sa[NB], sb[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const int e = e0 + u * 256 + tid;
      v[u] = 0.0; d[u] = -1; sa[u] = 0; sb[u] = 0;
      if (e >= P.ND) continue;
      // (the value is requested together with its table entries, not behind them: one round trip instead of two; the
      // mailbox reader polls and stays conditional)
      double acc = SRC == 2 ? 0.0 : val.get<SRC>(P.pc_dir + e);
      d[u] = P.dir_dst[e];
      sa[u] = P.dir_sa[e]; sb[u] = P.dir_sb[e];
      const int nx = P.dir_next[e];
      if (d[u] == -1) continue;
      if (SRC == 2) acc = val.get<SRC>(P.pc_dir + e);
      for (int n = nx; n >= 0; n = P.dir_next[n]) acc += val.get<SRC>(P.pc_dir + n);
      v[u] = acc;
    }
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      if (d[u] >= 0) {
        const double x = s_ss[sa[u]] * v[u] * s_ss[sb[u]];
        __hip_atomic_fetch_add(&A[d[u]], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (sa[u] == sb[u]) s_hd[sa[u]] = x;
      } else if (d[u] <= -2) {
        s_gs[-2 - d[u]] = v[u];
      }
    }
  }
  for (int i0 = 0; i0 < P.nT * 256; i0 += NB * 256) {
    double v[NB];
    int d[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const int i = i0 + u * 256 + tid;
      v[u] = 0.0; d[u] = -1;
      if (i >= P.nT * 256) continue;
      if (SRC != 2) v[u] = val.get<SRC>(i);   // (with the table entry, not behind it)
      d[u] = P.tile_dst[i];
      if (SRC == 2 && d[u] != -1) v[u] = val.get<SRC>(i);
    }
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      if (d[u] >= 0) __hip_atomic_fetch_add(&A[d[u]], -v[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else if (d[u] <= -2) s_b[-2 - d[u]] = -v[u];
    }
  }
  __syncthreads();
  // ---- 2. right-hand side, LM diagonal, constant coordinates (held intrinsics) become identity rows
  if (tid < S) s_b[tid] = pinned ? 0.0 : s_b[tid] + s_ss[tid] * s_gs[tid];
  __syncthreads();
  if (tid < S) {
    if (pinned) {
      for (int k = 0; k < tid; ++k) A[(size_t)tid * LD + k] = 0.0;
      for (int k = tid + 1; k < S; ++k) A[(size_t)k * LD + tid] = 0.0;
      A[(size_t)tid * LD + tid] = 1.0;
    } else {
      A[(size_t)tid * LD + tid] += clampd(s_hd[tid], o.min_lm_diagonal, o.max_lm_diagonal) / radius;
    }
  }
  // gradient of the accepted point: max-norm over the tangent coordinates (frames: per-rank slots)
  {
    double g = gm_r;
    if (tid < S && !pinned) {
      // Ceres' gradient_max_norm, ||x - Plus(x, -g)||_inf: a camera's pose block through Plus (its first coordinate's thread;
      // pose_grad_proj_max, cc_common.hpp), every other coordinate as it is
      const int info = P.colinfo[tid], kind = (info >> 4) & 15, comp = info & 15;
      if (kind != 0) {
        g = fmax(g, fabs(s_gs[tid]));
      } else if (comp == 0) {
        const double* qc = P.cam + ((size_t)cur * P.C + P.obs_cam[info >> 8]) * 8;
        const double q4[4] = {qc[0], qc[1], qc[2], qc[3]};
        const double g6[6] = {s_gs[tid], s_gs[tid + 1], s_gs[tid + 2], s_gs[tid + 3], s_gs[tid + 4], s_gs[tid + 5]};
        g = fmax(g, pose_grad_proj_max(q4, g6));
      }
    }
    g = wave_max(g);
    if (lane == 0) s4[tid >> 6] = g;
  }
  __syncthreads();
  if (tid == 0) {
    LmCtl c = s_c;
    const double gmax = fmax(fmax(s4[0], s4[1]), fmax(s4[2], s4[3]));
    // (stored whether or not the step was accepted: after a rejected step the accepted point, hence its gradient and this
    // very value, is unchanged -- testing the record's `accepted` flag first was a global load on the solving block's path)
    if (c.log_len > 0 && c.log_len <= P.log_cap) P.log[c.log_len - 1].gradient_max_norm = gmax;
    if (s_ok == 0) { c.done = 1; c.term = CC_FAILURE_EXCHANGE; }
    else if (lm_finalize(c, o, gmax)) s_go = 1;
    if (fail > 0.0) s_cholok = 0;
    s_c = c;
  }
  __syncthreads();
  // Small reduced systems (poses of one to four optimised cameras: S = 6, 12, 18, 24 -- BASELINE configs[3] is S = 18): the
  // whole solve on wave 0 with the matrix distributed by rows over the lanes (chol_solve_rows, cc_device.hpp): pivots and
  // multipliers travel through v_readlane, no LDS vector, no panel loop, no barrier. S = 18: 7.5 -> ~3 us for the
  // factorisation and both substitutions (profiles/r03/rig_stage_marks.jsonl).
  const bool small_rows = S == 6 || S == 12 || S == 18 || S == 24;
  if (s_go && small_rows) {
    if (tid < 64) {
      bool okw = true;
      double xs = 0.0;
      auto solve_rows = [&](auto tag) {
        constexpr int SS = decltype(tag)::value;
        const int i = lane < SS ? lane : SS - 1;   // (lanes beyond the system repeat its last row: finite, never read)
        double a[SS], x[SS];
#pragma unroll
        for (int k = 0; k < SS; ++k) a[k] = A[(size_t)i * LD + (k <= i ? k : i)];
        okw = chol_solve_rows<SS>(a, s_b[i], x);
#pragma unroll
        for (int k = 0; k < SS; ++k) xs = lane == k ? x[k] : xs;
      };
      if (S == 6) solve_rows(std::integral_constant<int, 6>{});
      else if (S == 12) solve_rows(std::integral_constant<int, 12>{});
      else if (S == 18) solve_rows(std::integral_constant<int, 18>{});
      else solve_rows(std::integral_constant<int, 24>{});
      const bool fin = lane >= S || isfinite(xs);
      const bool step_ok = s_cholok != 0 && okw && __all(fin);
      if (lane < S) { s_b[lane] = xs; if (SRC != 3) store_ds(P.ds + lane, -xs); }
      if (SRC != 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the update blocks of this launch read ds behind a flag (SRC 3: the step travels in a broadcast)
      if (lane == 0) s_stepok = step_ok ? 1 : 0;
    }
    __syncthreads();
  } else if (SRC != 3 && s_go && S <= 63) {   // (chol_block4: the right-hand side is row S on lane S of wave 0; SRC 3 -- the lean form's control workgroup -- only ever has S <= 24: the larger routines are not compiled into its kernel)
    // ---- medium systems: four columns at a time on all four waves, right-hand side as row S (chol_block4)
    const bool okb = chol_block4(A, S, LD, s_inv);
    if (tid < 64) {
      const int i0 = lane;
      double b0 = i0 < S ? s_b[i0] : 0.0, b1 = 0.0;          // y = L^-1 b (row S of the matrix)
      const double v0 = i0 < S ? s_inv[i0] : 0.0;
      chol_backward<false>(A, S, LD, b0, b1, v0, 0.0);
      const bool fin = i0 >= S || isfinite(b0);
      const bool step_ok = s_cholok != 0 && okb && __all(fin);
      if (i0 < S) { s_b[i0] = b0; if (SRC != 3) store_ds(P.ds + i0, -b0); }
      if (SRC != 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the update blocks of this launch read ds behind a flag
      if (lane == 0) s_stepok = step_ok ? 1 : 0;
    }
    __syncthreads();
  } else if (SRC != 3 && s_go) {
    // ---- Cholesky of the damped reduced system in LDS, eight columns at a time (S <= 127):
    //   panel:    wave 0 (chol_panel), forward substitution included;
    //   trailing: all 256 threads, A[i][k] -= sum_c L[i][c] L[k][c] over the panel's columns; two barriers per panel.
    const int i0 = lane, i1 = lane + 64;
    double b0 = 0.0, b1 = 0.0, v0 = 0.0, v1 = 0.0;
    bool okw = true;
    if (tid < 64) { b0 = i0 < S ? s_b[i0] : 0.0; b1 = i1 < S ? s_b[i1] : 0.0; }
    if (S > 64) {
      // ---- 65 .. 127 coordinates (rigs of 11 .. 21 optimised cameras; cameras with their own intrinsics): trailing updates on the
      // matrix pipe, with LOOK-AHEAD since round 6. Before, a panel's whole trailing update (all four waves, ~2 us at S = 114)
      // stood between two panels (wave 0 alone, ~2.5 us): 68 us of factorisation. Now only the tiles of the first tile column --
      // the next panel's columns -- are updated by all four waves first (part 1: at most eight tiles, one round); then wave 0
      // factors the next panel WHILE waves 1..3 give the rest of the trailing matrix (part 2: columns beyond t0 + 15, which the
      // next panel neither reads nor writes) the same update. Every element still receives the panels' updates one panel after
      // the other, each as `old - T` with T summed over the panel's eight columns in the same two k-steps: the bits do not change.
      if (tid < 64) chol_panel<true>(A, S, LD, 0, 8, s_inv, b0, b1, v0, v1, okw);
      __syncthreads();
      for (int j0 = 0; j0 + 8 < S; j0 += 8) {
        const int t0 = j0 + 8, nn = S - t0 < 8 ? S - t0 : 8;
        chol_trail_mfma<2, 1, 4>(A, S, LD, j0, 8, t0, S, tid >> 6);
        __syncthreads();
        if (tid < 64) chol_panel<true>(A, S, LD, t0, nn, s_inv, b0, b1, v0, v1, okw);
        else chol_trail_mfma<2, 2, 3>(A, S, LD, j0, 8, t0, S, (tid >> 6) - 1);
        __syncthreads();
      }
    } else
    for (int j0 = 0; j0 < S; j0 += 8) {   // (S = 64: the element-wise trailing update)
      const int nc = S - j0 < 8 ? S - j0 : 8;
      if (tid < 64) chol_panel<false>(A, S, LD, j0, nc, s_inv, b0, b1, v0, v1, okw);
      __syncthreads();
      const int t0 = j0 + nc;
      if (t0 < S) {
        // trailing triangle rows t0..S-1, columns t0..row, as a flat list of elements dealt to the threads three at a
        // time: all LDS reads of a batch are issued before its first write (the elements are distinct and none lies
        // in the panel's columns, which the compiler cannot know), so a batch costs one LDS round trip, not three
        const int nt = S - t0, ne = nt * (nt + 1) / 2;
        for (int e0 = 0; e0 < ne; e0 += 3 * 256) {
          double acc[3], li[3][8], lk[3][8];
          int at[3];
#pragma unroll
          for (int u = 0; u < 3; ++u) {
            const int e = e0 + u * 256 + tid;
            at[u] = -1;
            acc[u] = 0.0;
#pragma unroll
            for (int c = 0; c < 8; ++c) { li[u][c] = 0.0; lk[u][c] = 0.0; }
            if (e < ne) {
              int n = (int)((sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
              while (n * (n + 1) / 2 > e) --n;
              while ((n + 1) * (n + 2) / 2 <= e) ++n;
              const int i = t0 + n, k = t0 + (e - n * (n + 1) / 2);
              at[u] = i * LD + k;
              acc[u] = A[at[u]];
#pragma unroll
              for (int c = 0; c < 8; ++c) {   // (loads past the panel's end stay inside the LDS block; selected away)
                const double x = A[(size_t)i * LD + j0 + c], y = A[(size_t)k * LD + j0 + c];
                li[u][c] = c < nc ? x : 0.0;
                lk[u][c] = c < nc ? y : 0.0;
              }
            }
          }
#pragma unroll
          for (int u = 0; u < 3; ++u) {
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[u] -= li[u][c] * lk[u][c];
          }
#pragma unroll
          for (int u = 0; u < 3; ++u)
            if (at[u] >= 0) A[at[u]] = acc[u];
        }
      }
      __syncthreads();
    }
    if (tid < 64) {
      if (S <= 64) chol_backward<false>(A, S, LD, b0, b1, v0, v1);
      else chol_backward<true>(A, S, LD, b0, b1, v0, v1);
      const bool fin = (i0 >= S || isfinite(b0)) && (i1 >= S || isfinite(b1));
      const bool step_ok = s_cholok != 0 && __all(okw) && __all(fin);
      if (i0 < S) { s_b[i0] = b0; if (SRC != 3) store_ds(P.ds + i0, -b0); }
      if (i1 < S) { s_b[i1] = b1; if (SRC != 3) store_ds(P.ds + i1, -b1); }
      if (SRC != 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the update blocks of this launch read ds behind a flag
      if (lane == 0) s_stepok = step_ok ? 1 : 0;
    }
    __syncthreads();
  }
  const bool have_step = s_go != 0 && s_stepok != 0;
  // Fused launch (k_rig_reduce<0>, flag_epoch != 0): the blocks waiting to update their frames need the shared step -- stored
  // and drained above -- and three bits of the control block that are final by now (done and cur do not change below,
  // step_valid is have_step): the flag goes up HERE, and the frame updates run under the camera candidates, block sums and
  // control block below instead of behind them (2.8 us at BASELINE configs[4] size). The next kernel reads the control block;
  // this one is not over before it is written.
  if (flag_epoch != 0u && tid == 0) {
    const unsigned fl = (flag_epoch << 3) | (s_c.done ? 4u : 0u) | ((s_go ? have_step : (s_c.step_valid != 0)) ? 2u : 0u) | (unsigned)(s_c.cur & 1);
    __hip_atomic_store(P.arrive + 1, fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // ---- camera / intrinsics candidates and records (nothing moves unless a valid step exists)
  double st2 = 0.0, xs2 = 0.0;
  if (have_step) rig_candidates(P, s_b, s_ss, true, cur, dst, st2, xs2);
  {   // both block sums behind one pair of barriers
    const double a = wave_sum(st2), b2 = wave_sum(xs2);
    __syncthreads();
    if (lane == 0) { s8[tid >> 6] = a; s8[4 + (tid >> 6)] = b2; }
    __syncthreads();
  }
  const double st = (s8[0] + s8[1]) + (s8[2] + s8[3]);
  const double xs = (s8[4] + s8[5]) + (s8[6] + s8[7]);
  if (tid == 0) {
    LmCtl c = s_c;
    if (s_go) {
      c.step_valid = have_step ? 1 : 0;
      c.cand_pending = 1;
      P.shared_stats[0] = st;
      P.shared_stats[1] = xs;
    }
    if (SRC == 2) P.x.seq[0] = val.epoch;
    *P.ctl = c;
    *P.ctl_next = c;
  }
}

// Hands the control block (and the failure word of the in-kernel waits) to the host without a copy engine in the way:
// payload first, then the sequence word the host spins on (system-scope stores into pinned host memory; one thread).
// Last reduce launch of a host chunk only (cf. publish_to_host, cc_intrinsics_dev.hpp).
__device__ __forceinline__ void rig_publish(const RigDev& P, const LmCtl& c) {
  if (!P.host_pub) return;
  const unsigned long long* w = reinterpret_cast<const unsigned long long*>(&c);
#pragma unroll
  for (int i = 0; i < (int)(sizeof(LmCtl) / 8); ++i)
    __hip_atomic_store(P.host_pub + 2 + i, w[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  const unsigned failed = __hip_atomic_load(P.arrive + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(P.host_pub + 2 + sizeof(LmCtl) / 8, (unsigned long long)failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  const unsigned long long seq = *P.pub_seq + 1ull;
  *P.pub_seq = seq;
  __hip_atomic_store(P.host_pub, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// The solve step as a kernel of its own.
//   SRC 0 (RCCL route): after the all-reduce of P.vec.
//   SRC 2 (mailbox exchange on a device this rank SHARES with other shards or processes: rig_enqueue_round): this ONE
//          block collects every rank's posts (k_rig_reduce<4> made ours) in rank order inside rig_solve_block, and -- last
//          launch of a host chunk but for the pose update -- publishes the control block to the host. No block of any
//          launch of this form waits for another block: the only waits are this block's polls of its own mailbox.
template <int SRC>
__global__ __launch_bounds__(256) void k_rig_solve(RigDev P, int publish) {
  rig_progress(P, RIG_PROG_SOLVE);
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const LmCtl* cn = P.ctl_next;
  if (cn->done) {
    if (threadIdx.x == 0) { *P.ctl = *cn; if (SRC == 2 && publish) rig_publish(P, *cn); }
    return;
  }
  if (cn->phase == 0) return;
  rig_solve_block<SRC>(P, reinterpret_cast<double*>(smem_raw));
  if (SRC == 2 && publish) {
    __syncthreads();
    if (threadIdx.x == 0) rig_publish(P, *P.ctl);   // (written by this very thread a moment ago)
  }
}

// Large rigs on the mailbox exchange: k_rig_reduce<4> posted this rank's column sums into every mailbox; ONE block collects
// all ranks' posts in rank order into P.vec, which k_rig_solve_big then reads as it does after an all-reduce.
__global__ __launch_bounds__(256) void k_rig_collect(RigDev P) {
  __shared__ int s_ok;
  const LmCtl* cn = P.ctl_next;
  if (cn->done || cn->phase == 0) return;
  const unsigned long long epoch = P.x.seq[0] + 1ull;
  p2p_collect_to(P.x, 0, epoch, P.rank, P.nranks, P.PC + 32, P.vec, &s_ok);
  if (threadIdx.x == 0) {
    P.x.seq[0] = epoch;
    if (s_ok == 0) {
      LmCtl c = *cn;
      c.done = 1; c.term = CC_FAILURE_EXCHANGE;
      *P.ctl = c; *P.ctl_next = c;
    }
  }
}

// first launch of a solve: camera / intrinsics records of the starting point (what the first sweep reads)
__global__ __launch_bounds__(256) void k_rig_records(RigDev P) {
  const LmCtl* ctl = P.ctl;
  if (ctl->done || ctl->phase != 0) return;
  double a, b;
  rig_candidates(P, nullptr, nullptr, false, ctl->cur, ctl->cur, a, b);
}

// ---------------------------------------------------------------------------------------------
// reduce (+ solve + update): column sums (max for the last column) of the elimination partial rows, 16 columns
// per block and step, 16 row groups per column, 16 loads in flight per thread. Deterministic.
// MODE 0 (single GPU): the sums are stored write-through, the block arrives on a counter and the LAST block
// to arrive runs the solve step on them (sc1 loads, no fence: MI355X guide, valid hand-off forms).
// MODE 3 (mailbox exchange): every block posts its sums straight into all ranks' mailboxes; the last block
// to arrive collects them in rank order inside the solve step. MODE 2 (RCCL): sums -> P.vec, nothing else.
// MODE 4 (mailbox exchange on a SHARED device): sums -> every rank's mailbox, nothing else -- the solve step and the pose
// update are launches of their own (k_rig_solve<2>, k_rig_update), so no block waits for another one.
// MODES 0 and 3 then run the POSE UPDATE in the same launch: the grid is at most one block per CU (all of them
// resident), the blocks that are not last wait for a flag word the solver stores (epoch | done | step_valid | cur,
// sc1, behind its drained sc1 stores of the shared step) and every block updates its share of the frames. The wait is
// bounded (10 s of the wall clock) like the mailbox polls.
// ---------------------------------------------------------------------------------------------

template <int MODE>
__global__ __launch_bounds__(256) void k_rig_reduce(RigDev P, int publish) {
  rig_progress(P, RIG_PROG_REDUCE);
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  __shared__ double s_r[16][16];
  __shared__ double s_post[48];
  __shared__ double s_tail;
  __shared__ int s_last;
  __shared__ unsigned s_flag;
  constexpr bool FUSED = MODE == 0 || MODE == 3;   // solve step + pose update in this launch (its blocks wait for each other)
  const LmCtl* cn = P.ctl_next;
  if (cn->done) {
    if (FUSED && blockIdx.x == 0 && threadIdx.x == 0) { *P.ctl = *cn; if (publish) rig_publish(P, *cn); }
    return;
  }
  if (cn->phase == 0) return;
  // an earlier launch of this solve gave up waiting (below): the state is half updated, the host will report it
  // (rig_wait); do not wait another ten seconds per remaining round of the chunk
  if (FUSED && __hip_atomic_load(P.arrive + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
    if (publish && blockIdx.x == 0 && threadIdx.x == 0) rig_publish(P, *cn);
    return;
  }
  const int tid = threadIdx.x, c = tid & 15, grp = tid >> 4;  // 16 columns x 16 row groups per step
  unsigned epoch0 = 0;
  if (FUSED) epoch0 = __hip_atomic_load(P.arrive + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 3;   // before we arrive
  for (int first = blockIdx.x * 16; first < P.PC; first += gridDim.x * 16) {
    const int o = first + c;
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
    __syncthreads();   // (readers of the previous step)
    s_r[grp][c] = a;
    __syncthreads();
    if (tid < 16 && o < P.PC) {
      double r = 0.0;
      if (is_max) { for (int g2 = 0; g2 < 16; ++g2) r = fmax(r, s_r[g2][c]); }
      else {
        r = (((s_r[0][c] + s_r[1][c]) + (s_r[2][c] + s_r[3][c])) + ((s_r[4][c] + s_r[5][c]) + (s_r[6][c] + s_r[7][c]))) +
            (((s_r[8][c] + s_r[9][c]) + (s_r[10][c] + s_r[11][c])) + ((s_r[12][c] + s_r[13][c]) + (s_r[14][c] + s_r[15][c])));
      }
      unsigned long long* vw = reinterpret_cast<unsigned long long*>(P.vec);
      if (is_max) {   // the per-rank slot carries the max (a sum exchange then keeps it); the column itself is 0
        if (MODE == 0) __hip_atomic_store(vw + P.PC + P.rank, (unsigned long long)__double_as_longlong(r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else P.vec[P.PC + P.rank] = r;
        s_tail = r;
        r = 0.0;
      }
      if (MODE == 0) __hip_atomic_store(vw + o, (unsigned long long)__double_as_longlong(r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else P.vec[o] = r;
      s_post[c] = r;
    }
    if (MODE == 3 || MODE == 4) {
      // mailbox exchange (kind 0): the block that owns the max column also posts the 32 per-rank max slots
      // (ours set, the others zero). The epoch is stable here: only the solve step advances it.
      __syncthreads();
      const unsigned long long epoch = P.x.seq[0] + 1ull;
      const int ncol = P.PC - first < 16 ? P.PC - first : 16;
      p2p_post(P.x, 0, epoch, P.rank, P.nranks, s_post, ncol, first);
      if (first <= P.pc_gmax && P.pc_gmax < first + 16) {
        if (tid < 32) s_post[16 + tid] = tid == P.rank ? s_tail : 0.0;
        __syncthreads();
        p2p_post(P.x, 0, epoch, P.rank, P.nranks, s_post + 16, 32, P.PC);
      }
    }
  }
  if (!FUSED) return;
  // ---- last-block-done: every storing wave drains its stores, the block arrives, the last one solves
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    const unsigned prev = __hip_atomic_fetch_add(P.arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = prev + 1u == gridDim.x;
  }
  __syncthreads();
  // every block -- the solving one included -- requests what the update of its first sixteen frames needs NOW, before it
  // waits (or solves): behind the flag only the shared step is still to be read
  RigUpdPre pre;
  const bool use_pre = P.SW <= 64 && (int64_t)blockIdx.x * 16 < P.F;
  rig_update_prefetch(P, (int64_t)blockIdx.x * 16 + (tid >> 4), pre);   // (unconditional: loads inside an `if` would be waited for at its end)
  if (s_last) {
    if (tid == 0) __hip_atomic_store(P.arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // next launch
    rig_solve_block<MODE == 0 ? 1 : 2>(P, reinterpret_cast<double*>(smem_raw), nullptr, nullptr, MODE == 0 ? epoch0 + 1u : 0u);   // (MODE 0: raises the flag itself, early)
    // the shared step (sc1 stores of wave 0) has been drained inside; hand the outcome to the waiting blocks
    __syncthreads();
    if (tid == 0) {
      const LmCtl* c = P.ctl;   // written by this very thread a moment ago
      s_flag = ((epoch0 + 1u) << 3) | (c->done ? 4u : 0u) | (c->step_valid ? 2u : 0u) | (unsigned)(c->cur & 1);
      __hip_atomic_store(P.arrive + 1, s_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // (the host only decides on it whether another chunk follows; whoever reads poses synchronises the stream first)
      if (publish) rig_publish(P, *c);
    }
  }
  if (!s_last && tid == 0) {
    const long long t0 = wall_clock64();
    unsigned f;
    for (;;) {
      f = __hip_atomic_load(P.arrive + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((f >> 3) == epoch0 + 1u) break;
      if (wall_clock64() - t0 > kP2pTimeoutTicks) {
        // The solving block did not publish within 10 s: the blocks of this launch were not all resident (the grid is sized
        // for that at launch, rig_reduce_blocks) or the solve step waits for a peer rank. Leave without updating and SAY SO:
        // the failure word makes every later launch a no-op and the host return CC_ERR_COMM (rig_wait).
        __hip_atomic_store(P.arrive + 3, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        f = 4u;
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
    s_flag = f;
  }
  __syncthreads();
  const unsigned flag = s_flag;
  if ((flag & 4u) || !(flag & 2u)) return;   // done, or no valid step: the poses stay
  const int cur = (int)(flag & 1u);
  if (use_pre) rig_update_body<true, true>(P, 1, cur, (int64_t)blockIdx.x * 16 + (tid >> 4), pre);
  for (int64_t fblk = use_pre ? (int64_t)blockIdx.x + gridDim.x : (int64_t)blockIdx.x; fblk * 16 < P.F; fblk += gridDim.x)
    rig_update_body<true, false>(P, 1, cur, fblk * 16 + (tid >> 4), pre);
}

