// cc_rig_big.hpp -- the plain kernels for reduced systems of 128 .. 255 coordinates (k_rig_elim_big, k_rig_solve_big).
// Part of cc_rig.hip (round 5: the 6.8 k-line file split by subject; included by it inside namespace cc, in this order:
// cc_rig_sweeps.hpp, cc_rig_steps.hpp, cc_rig_big.hpp, cc_rig_lean.hpp -- one translation unit, nothing else includes these).
#pragma once

// =============================================================================================
// LARGE reduced systems (128 <= S <= 255: more than 21 optimised cameras, or more than 8 with intrinsics of their own --
// or more direct sums than k_rig_elim keeps, 12+ observed cameras with intrinsics; the reference takes any number of
// cameras, extrinsics_calibrator.cpp:9-17). The kernels above are built around
// S + 1 <= 128 (two shared columns per lane, nine tile accumulators per wave, the reduced system in LDS with a row stride);
// rather than bend them, such problems run the same arithmetic in a plainer form -- correctness first, no tuning:
//   k_rig_elim_big : one block per frame at a time, thread k owns shared column k (S + 1 <= 256); the 6 x 6 factor is
//                    computed by every thread; Schur products Z^T Z accumulated per 16 x 16 tile with plain FMAs, entry
//                    `tid` of every tile in a register (<= 136 tiles); the direct sums in LDS. Same partial-row layout.
//   k_rig_reduce<2>: the column sums (unchanged) -> P.vec
//   k_rig_solve_big: one block; the reduced system as a lower triangle packed by rows in LDS (S <= 193) or column-major
//                    in global memory, the right-hand side as row S; left-looking Cholesky, thread i owns row i, sixteen
//                    columns of both rows per round trip, two barriers per column; backward substitution with one
//                    barrier per step; the tests, candidates and control block of rig_solve_block.
//   k_rig_update   : unchanged.
// Sweep, init, records, statistics: unchanged (their shared-column arrays hold 256 entries).
// =============================================================================================
constexpr int kRigBigMaxS = 255;
constexpr int kRigBigTiles = 136;   // upper tile pairs of a 16 x 16 tile grid
__host__ __device__ constexpr int big_tile(int a, int b) { return a * 16 - a * (a - 1) / 2 + (b - a); }   // (a <= b < 16)

template <bool HK>
__global__ __launch_bounds__(256) void k_rig_elim_big(RigDev P) {
  rig_progress(P, RIG_PROG_ELIM);
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* s_Z = reinterpret_cast<double*>(smem_raw);   // [6][256] staged Z rows of the block's current frame
  double* s_d = s_Z + 6 * 256;                          // [ND] direct sums of the block
  __shared__ double s_A[32];
  __shared__ double s_ss[kRigBigMaxS + 1];
  __shared__ double s16[16];
  __shared__ double s_tot[4];
  __shared__ double s_fg[8];
  __shared__ int s_g[64];
  __shared__ LmCtl s_ctl;
  const int tid = threadIdx.x;
  const LmCtl* ctl = P.ctl;
  if (ctl->done || ctl->phase == 0) return;
  // ---- trust-region decision: every block, same answer; block 0 publishes it (as in k_rig_elim)
  const bool pending = ctl->cand_pending != 0;
  if (P.comm) {   // (sharded: k_rig_stats exchanged them)
    if (tid < 4) s_tot[tid] = P.vec_stats[tid];
    __syncthreads();
  } else {
    rig_reduce_stats(P, pending && ctl->step_valid, s16, s_tot);
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
  for (int i = tid; i < 6 * 256; i += 256) s_Z[i] = 0.0;
  for (int i = tid; i < P.ND; i += 256) s_d[i] = 0.0;
  __syncthreads();
  if (s_ctl.done) return;
  const int cur = s_ctl.cur;
  const double inv_radius = 1.0 / s_ctl.radius;
  const double mn = P.opts->min_lm_diagonal, mx = P.opts->max_lm_diagonal;
  const bool first_elim = ctl->phase == 1 && s_ctl.iter == 0 && !pending;
  const bool jac = P.opts->jacobi_scaling != 0;
  const int SW = P.SW, S = P.S, CO = P.CO, T = P.T;
  const size_t gs = (size_t)P.gstride;
  const double* blocks = P.gblocks + (size_t)cur * P.NG * gs;
  // this thread's shared column and frame-block entry
  int c_kind = -1, c_co = 0, c_comp = 0;
  if (tid < SW) { const int info = P.colinfo[tid]; c_kind = (info >> 4) & 15; c_co = info >> 8; c_comp = info & 15; }
  const double c_ss = tid < S ? s_ss[tid] : (tid < SW ? 1.0 : 0.0);
  int a_off = 0, sp_i = -1;
  if (tid < 21) {
    int i = 0;
    while (tri(i + 1, 0) <= tid) ++i;
    const int j = tid - tri(i, 0);
    a_off = (6 + i) * 16 + 6 + j;
    if (i == j) sp_i = i;
  } else if (tid < 27) {
    a_off = (6 + (tid - 21)) * 16 + 12;
  }
  double acc[kRigBigTiles];
#pragma unroll
  for (int t = 0; t < kRigBigTiles; ++t) acc[t] = 0.0;
  // (failure count and gradient maximum of the block live in LDS, s_fg[0] / s_fg[1]: thread 0 alone touches them)
  if (tid == 0) { s_fg[0] = 0.0; s_fg[1] = 0.0; }
  const int tr = tid >> 4, tc = tid & 15;
  for (int64_t f = blockIdx.x; f < P.F; f += gridDim.x) {
    if (tid < 64) s_g[tid] = tid < CO ? P.fslot[f * CO + tid] : -1;
    __syncthreads();
    bool live = false;
    for (int j = 0; j < CO; ++j) live = live || s_g[j] >= 0;
    if (live) {
      if (tid < 27) {
        // (eight loads per round trip, unconditional from a clamped group, then selects: one load per wait took 20 us of a
        // frame's 32 with 40 observed cameras; the sum keeps its order)
        double a_e = 0.0;
        for (int j0 = 0; j0 < CO; j0 += 8) {
          double v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int g = j0 + u < CO ? s_g[j0 + u] : -1;
            const double x = blocks[(size_t)(g >= 0 ? g : 0) * gs + a_off];
            v[u] = g >= 0 ? x : 0.0;
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) a_e += v[u];
        }
        s_A[tid] = a_e;
        if (first_elim && sp_i >= 0) P.sp[f * 8 + sp_i] = jac ? 1.0 / (1.0 + sqrt(a_e)) : 1.0;
      }
      __syncthreads();
      double A[27], sf[6];
#pragma unroll
      for (int i = 0; i < 27; ++i) A[i] = s_A[i];
#pragma unroll
      for (int i = 0; i < 6; ++i) sf[i] = first_elim ? (jac ? 1.0 / (1.0 + sqrt(A[tri(i, i)])) : 1.0) : P.sp[f * 8 + i];
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
      if (tid == 0) {
        if (!ok) s_fg[0] += 1.0;
        const double* fqp = P.pose + ((size_t)cur * P.F + f) * 8;
        const double q4[4] = {fqp[0], fqp[1], fqp[2], fqp[3]};
        s_fg[1] = fmax(s_fg[1], pose_grad_proj_max(q4, &A[21]));   // Ceres' gradient_max_norm (cc_common.hpp)
      }
      if (tid < SW) {
        double w[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        if (c_kind == 0 || c_kind == 1) {
          const int g = s_g[c_co];
          if (g >= 0) {
            const double* G = blocks + (size_t)g * gs;
#pragma unroll
            for (int i = 0; i < 6; ++i) w[i] = c_kind == 0 ? G[c_comp * 16 + 6 + i] : G[256 + (6 + i) * 16 + c_comp];
          }
        } else if (HK && c_kind == 2) {
          for (int j = 0; j < CO; ++j) {
            const int g = s_g[j];
            if (g < 0) continue;
            const double* G = blocks + (size_t)g * gs + 256;
#pragma unroll
            for (int i = 0; i < 6; ++i) w[i] += G[(6 + i) * 16 + c_comp];
          }
        }
        double z[6], y[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          double a = c_kind == 3 ? sf[i] * A[21 + i] : sf[i] * w[i] * c_ss;
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
          s_Z[i * 256 + tid] = z[i];
          P.Y[((size_t)f * 6 + i) * SW + tid] = y[i];
        }
      }
      for (int e0 = tid; e0 < P.ND; e0 += 4 * 256) {   // (four entries per round trip)
        double v[4];
        bool k[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int e = e0 + 256 * u;
          const int t = P.dent[e < P.ND ? e : 0];
          const int g = s_g[t >> 16];
          k[u] = e < P.ND && g >= 0;
          v[u] = blocks[(size_t)(g >= 0 ? g : 0) * gs + (t & 0xffff)];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (k[u]) s_d[e0 + 256 * u] += v[u];
      }
      __syncthreads();
      // Schur products of this frame: entry (tr, tc) of every upper tile pair (a, b), a <= b < T. The loops run over the
      // largest tile grid with compile-time accumulator indices (no tables: 136 pairs of table entries in scalar registers
      // spilled hundreds of them); which pairs exist is a uniform test.
#pragma unroll
      for (int a = 0; a < 16; ++a) {
        if (a < T) {
          double za[6];
#pragma unroll
          for (int i = 0; i < 6; ++i) za[i] = s_Z[i * 256 + 16 * a + tr];
#pragma unroll
          for (int b = a; b < 16; ++b) {
            if (b < T) {
              double x = acc[big_tile(a, b)];
#pragma unroll
              for (int i = 0; i < 6; ++i) x = fma(za[i], s_Z[i * 256 + 16 * b + tc], x);
              acc[big_tile(a, b)] = x;
            }
          }
        }
      }
    }
    __syncthreads();
  }
  double* prow = P.partial + (size_t)blockIdx.x * P.PC;
  // (the partial row numbers the pairs of the T x T grid in the same order: a, then b)
#pragma unroll
  for (int a = 0; a < 16; ++a)
#pragma unroll
    for (int b = a; b < 16; ++b)
      if (b < T) prow[(size_t)(a * T - a * (a - 1) / 2 + (b - a)) * 256 + tid] = acc[big_tile(a, b)];
  for (int e = tid; e < P.ND; e += 256) prow[P.pc_dir + e] = s_d[e];
  if (tid == 0) { prow[P.pc_fail] = s_fg[0]; prow[P.pc_gmax] = s_fg[1]; }
}

// accessor of the reduced system's lower triangle (rows 0..S, row S = right-hand side; S columns). In LDS: packed by
// rows -- thread i owns row i, a batch of its entries is one base address plus immediates. In global memory (L2-resident:
// 0.5 MB at S = 255): ROW-major with the stride the host's destination tables use -- a thread's sixteen panel entries are
// 128 contiguous bytes, the sixteen columns of a trailing tile's row one transaction (round 3 kept it column-major for its
// left-looking factorisation, one row per thread).
constexpr int kRigBigPanelDoubles = 256 * 17 + 64;
template <bool PACKED>
struct BigA {
  double* p; int LD;
  __device__ __forceinline__ double& at(int i, int k) const {   // k <= i <= S, k < S
    return PACKED ? p[i * (i + 1) / 2 + k] : p[(size_t)i * LD + k];
  }
  // host-built destinations are row * LD + col (rig_layout)
  __device__ __forceinline__ double& at_dst(int dst) const { const int i = dst / LD, k = dst - i * LD; return at(i, k); }
};

template <bool PACKED>
__global__ __launch_bounds__(256) void k_rig_solve_big(RigDev P, double* Aglobal) {
  rig_progress(P, RIG_PROG_SOLVE);
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const LmCtl* cn = P.ctl_next;
  if (cn->done) {
    if (threadIdx.x == 0) *P.ctl = *cn;
    return;
  }
  if (cn->phase == 0) return;
  const int S = P.S, LD = (S + 1) | 1;
  double* lds = reinterpret_cast<double*>(smem_raw);
  double* s_b = lds;                  // [256] right-hand side -> y -> x
  double* s_gs = s_b + 256;           // [256] unscaled shared gradient
  double* s_hd = s_gs + 256;          // [256] diagonal of the scaled H_ss
  double* s_inv = s_hd + 256;         // [256] 1 / L_jj
  double* s_ss = s_inv + 256;         // [256]
  double* s_pan = s_ss + 256;         // [256][17] the factorisation's panel, then [64] micro-block words
  BigA<PACKED> A{PACKED ? s_pan + kRigBigPanelDoubles : Aglobal, LD};
  __shared__ int s_cholok, s_stepok, s_go;
  __shared__ double s4[4];
  __shared__ double s8[8];
  __shared__ LmCtl s_c;
  const int tid = threadIdx.x, lane = tid & 63;
  const int cur = cn->cur, dst = cur ^ 1;
  const double radius = cn->radius;
  const LmOpts o = *P.opts;
  if (tid == 0) { s_cholok = 1; s_stepok = 0; s_go = 0; s_c = *cn; }
  for (int i = tid; i <= S; i += 256)   // (row S: the right-hand side)
    for (int k = 0; k <= i && k < S; ++k) A.at(i, k) = 0.0;
  s_b[tid] = 0.0; s_gs[tid] = 0.0; s_hd[tid] = 0.0; s_inv[tid] = 0.0; s_ss[tid] = tid < S ? P.ss[tid] : 0.0;
  const int pin = tid < S ? P.colpin[tid] : -1;
  const bool pinned = pin >= 0 && ((P.kmask[pin >> 4] >> (pin & 15)) & 1u) != 0;
  __syncthreads();
  // ---- assembly from the column sums (P.vec): direct sums, then minus the Schur products. An element gets at most one
  // contribution of each kind; the two loops are separated by a barrier, so plain read-modify-write is safe. Loads are
  // batched eight deep (table entry and value together, then the eight elements): 30720 tile entries at S = 234 were 120
  // dependent round trips per thread one at a time -- the largest piece of the launch once the factorisation was blocked.
  for (int e0 = 0; e0 < P.ND; e0 += 8 * 256) {
    int d[8], sa[8], sb[8];
    double acc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + u * 256 + tid, ec = e < P.ND ? e : 0;
      d[u] = P.dir_dst[ec]; sa[u] = P.dir_sa[ec]; sb[u] = P.dir_sb[ec];
      acc[u] = P.vec[P.pc_dir + ec];
      if (e >= P.ND) d[u] = -1;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (d[u] == -1) continue;
      const int e = e0 + u * 256 + tid;
      for (int n = P.dir_next[e]; n >= 0; n = P.dir_next[n]) acc[u] += P.vec[P.pc_dir + n];
      if (d[u] >= 0) {
        const double x = s_ss[sa[u]] * acc[u] * s_ss[sb[u]];
        A.at_dst(d[u]) = x;          // (zeroed above, one direct contribution at most: a plain store)
        if (sa[u] == sb[u]) s_hd[sa[u]] = x;
      } else {
        s_gs[-2 - d[u]] = acc[u];
      }
    }
  }
  __syncthreads();
  for (int i0 = 0; i0 < P.nT * 256; i0 += 8 * 256) {
    int d[8];
    double v[8], old[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u * 256 + tid, ic = i < P.nT * 256 ? i : 0;
      d[u] = P.tile_dst[ic];
      v[u] = P.vec[ic];
      if (i >= P.nT * 256) d[u] = -1;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) old[u] = A.at_dst(d[u] >= 0 ? d[u] : 0);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (d[u] >= 0) A.at_dst(d[u]) = old[u] - v[u];
      else if (d[u] <= -2) s_b[-2 - d[u]] = -v[u];
    }
  }
  const double fail = P.vec[P.pc_fail];
  const double gm_r = (tid < P.nranks && tid < 32) ? P.vec[P.PC + tid] : 0.0;
  __syncthreads();
  if (tid < S) s_b[tid] = pinned ? 0.0 : s_b[tid] + s_ss[tid] * s_gs[tid];
  __syncthreads();
  if (tid < S) {
    if (pinned) {
      for (int k = 0; k < tid; ++k) A.at(tid, k) = 0.0;
      for (int k = tid + 1; k < S; ++k) A.at(k, tid) = 0.0;
      A.at(tid, tid) = 1.0;
    } else {
      A.at(tid, tid) += clampd(s_hd[tid], o.min_lm_diagonal, o.max_lm_diagonal) / radius;
    }
  }
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
    if (c.log_len > 0 && c.log_len <= P.log_cap) P.log[c.log_len - 1].gradient_max_norm = gmax;
    if (lm_finalize(c, o, gmax)) s_go = 1;
    if (fail > 0.0) s_cholok = 0;
    s_c = c;
  }
  __syncthreads();
  if (s_go) {
    // ---- blocked right-looking Cholesky (round 4), sixteen columns per panel, all 256 threads; the right-hand side is ROW S
    // of the matrix, so its forward substitution is what every other row undergoes. Per panel:
    //   thread i = row i holds the panel's sixteen entries of its row in registers; the panel is factored four columns at a
    //   time: the 4 x 4 diagonal block is published (sixteen LDS words), factored in closed form by every thread, every row
    //   below solves its four entries against it and takes the rank-4 update of its remaining panel entries with the
    //   multipliers the block's rows publish -- two barriers per FOUR columns, no dot product over finished columns;
    //   the panel goes back to the matrix and into an LDS tile [rows][17], and the trailing matrix takes its rank-16 update
    //   on the matrix pipe: 16 x 16 tiles dealt to the four waves, four at a time (every load of the four -- operands from
    //   the LDS panel, the elements themselves from LDS / L2 -- is issued before the first product).
    // Round 3's left-looking form, one row per thread and a dot product over all finished columns per entry, two barriers
    // per COLUMN: S = 234, 971 us per launch (profiles/r03/rig_big.jsonl).
    const int i = tid;
    if (tid < S) A.at(S, tid) = s_b[tid];
    __syncthreads();
    double* Pn = s_pan;        // [256][17] the panel, rows by matrix row
    double* s_d = s_pan + 256 * 17;   // [16] diagonal block of a micro-block, [16 + 12 * 4] multipliers of the panel's later rows
    bool ok = true;
    for (int j0 = 0; j0 < S; j0 += 16) {
      const int nc = S - j0 < 16 ? S - j0 : 16;
      const bool row_in = i >= j0 && i <= S;
      double pv[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const int col = j0 + (c < nc ? c : 0);
        const double x = A.at(row_in ? i : S, col <= (row_in ? i : S) ? col : 0);
        pv[c] = (row_in && c < nc && (j0 + c <= i || i == S)) ? x : 0.0;
      }
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) {
        const int c0 = 4 * mb;
        if (c0 < nc) {   // (uniform)
          const int nb = nc - c0 < 4 ? nc - c0 : 4;
          const int rb = i - (j0 + c0);   // row inside the micro-block: 0 .. nb - 1
          if (rb >= 0 && rb < nb) {
#pragma unroll
            for (int c = 0; c < 4; ++c) s_d[rb * 4 + c] = pv[c0 + c];
          }
          __syncthreads();
          double a00 = s_d[0], a10 = s_d[4], a11 = s_d[5], a20 = s_d[8], a21 = s_d[9], a22 = s_d[10], a30 = s_d[12], a31 = s_d[13], a32 = s_d[14], a33 = s_d[15];
          if (nb < 2) { a10 = 0.0; a11 = 1.0; }
          if (nb < 3) { a20 = 0.0; a21 = 0.0; a22 = 1.0; }
          if (nb < 4) { a30 = 0.0; a31 = 0.0; a32 = 0.0; a33 = 1.0; }
          const double i0 = rsqrt_pos(a00);
          ok = ok && (a00 > 0.0) && isfinite(a00);
          const double l10 = a10 * i0, l20 = a20 * i0, l30 = a30 * i0;
          const double d1 = fma(-l10, l10, a11);
          const double i1 = rsqrt_pos(d1);
          ok = ok && (d1 > 0.0) && isfinite(d1);
          const double l21 = fma(-l20, l10, a21) * i1, l31 = fma(-l30, l10, a31) * i1;
          const double d2 = fma(-l21, l21, fma(-l20, l20, a22));
          const double i2 = rsqrt_pos(d2);
          ok = ok && (d2 > 0.0) && isfinite(d2);
          const double l32 = fma(-l31, l21, fma(-l30, l20, a32)) * i2;
          const double d3 = fma(-l32, l32, fma(-l31, l31, fma(-l30, l30, a33)));
          const double i3 = rsqrt_pos(d3);
          ok = ok && (d3 > 0.0) && isfinite(d3);
          if (i == 0) { s_inv[j0 + c0] = i0; if (nb > 1) s_inv[j0 + c0 + 1] = i1; if (nb > 2) s_inv[j0 + c0 + 2] = i2; if (nb > 3) s_inv[j0 + c0 + 3] = i3; }
          // the row's four entries against the block (rows inside the block get their own row of L: the same recurrence
          // stopped at the diagonal)
          {
            const double x0 = pv[c0], x1 = nb > 1 ? pv[c0 + 1] : 0.0, x2 = nb > 2 ? pv[c0 + 2] : 0.0, x3 = nb > 3 ? pv[c0 + 3] : 0.0;
            const double y0 = x0 * i0;
            const double y1 = fma(-y0, l10, x1) * i1;
            const double y2 = fma(-y1, l21, fma(-y0, l20, x2)) * i2;
            const double y3 = fma(-y2, l32, fma(-y1, l31, fma(-y0, l30, x3))) * i3;
            const bool below = rb >= nb || (i == S);   // (row S: the right-hand side, below everything)
            // inside the block: row rb of L = entries up to the diagonal (y_c for c < rb is L[rb][c]; the diagonal is d * inv)
            pv[c0] = below ? y0 : (rb == 0 ? a00 * i0 : (rb > 0 ? y0 : pv[c0]));
            if (nb > 1) pv[c0 + 1] = below ? y1 : (rb == 1 ? d1 * i1 : (rb > 1 ? y1 : pv[c0 + 1]));
            if (nb > 2) pv[c0 + 2] = below ? y2 : (rb == 2 ? d2 * i2 : (rb > 2 ? y2 : pv[c0 + 2]));
            if (nb > 3) pv[c0 + 3] = below ? y3 : (rb == 3 ? d3 * i3 : pv[c0 + 3]);
          }
          // multipliers of the panel's later columns: rows j0 + c2 (c2 >= c0 + 4) publish their four new entries
          const int rl = i - j0;   // row inside the panel
          if (rl >= c0 + 4 && rl < 16 && rl < nc) {
#pragma unroll
            for (int c = 0; c < 4; ++c) s_d[16 + (rl - 4) * 4 + c] = pv[c0 + c];
          }
          __syncthreads();
#pragma unroll
          for (int c2 = c0 + 4; c2 < 16; ++c2) {
            if (c2 < nc) {   // (uniform)
              double acc = pv[c2];
#pragma unroll
              for (int c = 0; c < 4; ++c) acc = fma(-(c < nb ? pv[c0 + c] : 0.0), s_d[16 + (c2 - 4) * 4 + c], acc);
              pv[c2] = (row_in && (j0 + c2 <= i || i == S)) ? acc : 0.0;
            }
          }
        }
      }
      // the panel: back to the matrix, and into its LDS tile for the trailing update
      if (row_in) {
#pragma unroll
        for (int c = 0; c < 16; ++c)
          if (c < nc && (j0 + c <= i || i == S)) A.at(i, j0 + c) = pv[c];
      }
#pragma unroll
      for (int c = 0; c < 16; ++c) Pn[i * 17 + c] = (row_in && c < nc && i >= j0 + nc) ? pv[c] : 0.0;   // (rows below the panel: the update's operands)
      __syncthreads();
      // ---- trailing update: rows t0..S (right-hand side included), columns t0..S-1, 16 x 16 tiles at multiples of 16
      const int t0 = j0 + nc;
      if (t0 < S) {
        const int wv = tid >> 6, ln = tid & 63, kq = ln >> 4, c16 = ln & 15;
        const int tlo = t0 >> 4, n16 = (S + 1 + 15) >> 4;
        // tiles (ti, tj), tlo <= tj <= ti < n16, numbered row by row; wave w takes numbers w, w + 4, ...: four per round
        const int nrow = n16 - tlo, ntile = nrow * (nrow + 1) / 2;
        for (int tb = wv; tb < ntile; tb += 16) {
          double am[4][4], bm[4][4], old[4][4];
          int at_i[4][4], at_k[4][4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int t = tb + 4 * u;
            const bool live = t < ntile;
            const int tc = live ? t : 0;
            int tr = (int)((sqrtf(8.0f * (float)tc + 1.0f) - 1.0f) * 0.5f);
            tr = tr * (tr + 1) / 2 > tc ? tr - 1 : tr;
            tr = (tr + 1) * (tr + 2) / 2 <= tc ? tr + 1 : tr;
            const int tq = tc - tr * (tr + 1) / 2;
            const int R = 16 * (tlo + tr), Cc = 16 * (tlo + tq);
            const int ra = R + c16 <= S ? R + c16 : S, rb2 = Cc + c16 < S ? Cc + c16 : S - 1;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
              const double xa = Pn[ra * 17 + 4 * ks + kq], xb = Pn[rb2 * 17 + 4 * ks + kq];
              am[u][ks] = (live && R + c16 <= S) ? xa : 0.0;
              bm[u][ks] = (live && Cc + c16 < S) ? xb : 0.0;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int row = R + kq + 4 * r, col = Cc + c16;
              const bool v = live && row >= t0 && row <= S && col >= t0 && col < S && (col <= row);
              at_i[u][r] = v ? row : -1;
              at_k[u][r] = v ? col : 0;
              old[u][r] = A.at(v ? row : S, v ? col : 0);
            }
          }
          d4 T[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) T[u] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int u = 0; u < 4; ++u) T[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(am[u][ks], bm[u][ks], T[u], 0, 0, 0);
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (at_i[u][r] >= 0) A.at(at_i[u][r], at_k[u][r]) = old[u][r] - T[u][r];
        }
      }
      __syncthreads();
    }
    if (!ok && tid == 0) s_cholok = 0;
    if (tid < S) s_b[tid] = A.at(S, tid);
    __syncthreads();
    // ---- backward substitution L^T x = y in blocks of sixteen unknowns, from the last: wave 0 solves the block's triangle
    // (lane j holds y_j and column j of the block; sixteen steps of lane read + FMA, no barrier), publishes x, and every
    // row above the block subtracts its sixteen products at once -- two barriers per SIXTEEN unknowns (round 3: one per
    // unknown, each behind a dependent load)
    {
      double* s_x = s_pan;   // [16] the block's solution
      for (int kb = ((S - 1) >> 4) << 4; kb >= 0; kb -= 16) {
        const int nbk = S - kb < 16 ? S - kb : 16;
        if (tid < 64) {
          const int j = lane < nbk ? lane : 0;
          double bj = s_b[kb + j];
          double lcol[16];
#pragma unroll
          for (int k = 0; k < 16; ++k) {
            const int kr = kb + (k < nbk ? k : 0);
            const double x = A.at(kr > kb + j ? kr : kb + j, kb + j);   // L[kb + k][kb + j] for k > j
            lcol[k] = (k < nbk && k > j) ? x : 0.0;
          }
#pragma unroll
          for (int k = 15; k >= 0; --k) {
            if (k < nbk) {   // (uniform)
              const double xk = readlane_d(bj, k) * s_inv[kb + k];
              bj = lane == k ? xk : fma(-lcol[k], xk, bj);
            }
          }
          if (lane < nbk) { s_b[kb + lane] = bj; s_x[lane] = bj; }
        }
        __syncthreads();
        if (i < kb) {
          double acc = s_b[i];
          double l[16];
#pragma unroll
          for (int k = 0; k < 16; ++k) l[k] = A.at(kb + (k < nbk ? k : 0), i);
#pragma unroll
          for (int k = 0; k < 16; ++k) acc = fma(-(k < nbk ? l[k] : 0.0), s_x[k], acc);
          s_b[i] = acc;
        }
        __syncthreads();
      }
    }
    if (tid < 64) {
      bool fin = true;
      for (int k = lane; k < S; k += 64) {
        fin = fin && isfinite(s_b[k]);
        P.ds[k] = -s_b[k];
      }
      const bool step_ok = s_cholok != 0 && __all(fin);
      if (lane == 0) s_stepok = step_ok ? 1 : 0;
    }
    __syncthreads();
  }
  double st2 = 0.0, xs2 = 0.0;
  const bool have_step = s_go != 0 && s_stepok != 0;
  if (have_step) rig_candidates(P, s_b, s_ss, true, cur, dst, st2, xs2);
  {
    const double a = wave_sum(st2), b2 = wave_sum(xs2);
    __syncthreads();
    if (lane == 0) { s8[tid >> 6] = a; s8[4 + (tid >> 6)] = b2; }
    __syncthreads();
  }
  if (tid == 0) {
    LmCtl c = s_c;
    if (s_go) {
      c.step_valid = have_step ? 1 : 0;
      c.cand_pending = 1;
      P.shared_stats[0] = (s8[0] + s8[1]) + (s8[2] + s8[3]);
      P.shared_stats[1] = (s8[4] + s8[5]) + (s8[6] + s8[7]);
    }
    *P.ctl = c;
    *P.ctl_next = c;
  }
}

