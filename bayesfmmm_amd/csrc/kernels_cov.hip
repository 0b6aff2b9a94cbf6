// Covariate-adjusted models: the eta (mean effect) and Xi (covariance effect) blocks.
//
//   updateEta             UpdateEta.h:28-94    (d outer, j inner; prior tau_eta(j,d) * P_mat; pinv + symmetrise)
//   updateXiCovariateAdj  UpdateXi.h:26-93     ((j, m, d) order; prior diag(tilde_tau_xi(j,m,d) * gamma_xi_j(.,d,m)); inv)
//   updateTauEta          UpdateTau.h:75-95,   updateDeltaXi UpdateDelta.h:76-124,
//   updateAXi             UpdateA.h:137-205,   updateGammaXi UpdateGamma.h:48-72
//
// eta_j[:,d] and xi_jm[:,d] are directions with per-curve weight w_i = Z_ij * chit_{i,mt} * X_id.  All of them see the
// same Z, chi, X and sigma^2, so every conditional covariance C_a = ((1/sigma^2) sum_i w_i^2 G_i + Prior_a)^-1 is
// prepared up front in parallel (k_cov_w2 + k_cov_factor); the K*D + K*M*D draws are then sequential "direct"
// steps over the per-curve fitted coefficient c_i and g_i = G_i c_i kept in HBM:
//     rhs_a = (1/sigma^2) [ sum_i w_i (s_i - g_i) + (sum_i w_i^2 G_i) theta_a ],   theta_a ~ N(C_a rhs_a, C_a),
//     c_i += w_i (theta_a_new - theta_a_old),   g_i += w_i G_i (theta_a_new - theta_a_old)
// (the c_i / g_i update of step s is applied lazily at the start of step s+1): one launch per direction, k_cov_step.
#include "model.hpp"
#include "rng.hpp"
#include "scalar_jobs.hpp"
#include "factor_core.hpp"

namespace bfmmm {

struct Dir2 { int j, mt, dd; };

__host__ __device__ inline Dir2 dir2_of(const Dims& d, int a2) {
  Dir2 r;
  const int KD = d.K * d.D;
  if (a2 < KD) { r.dd = a2 / d.K; r.j = a2 - r.dd * d.K; r.mt = 0; }
  else {
    int q = a2 - KD;
    r.dd = q % d.D; q /= d.D;
    const int m = q % d.M;
    r.j = q / d.M;
    r.mt = m + 1;
  }
  return r;
}

__device__ inline double w_of(const Ctx& c, int i, const Dir2& a) {
  const int n = c.d.n;
  double w = c.Z[i + (size_t)n * a.j] * c.X[i + (size_t)n * a.dd];
  if (a.mt > 0) w *= c.chi[i + (size_t)n * (a.mt - 1)];
  return w;
}

__device__ inline bool dir2_updated(const Ctx& c, const Dir2& a) {
  if (a.mt == 0) return (c.mask & U_ETA) != 0;
  return (c.mask & U_XI) != 0 && c.d.MD > 1;
}

// ---- sum_i w_i^2 G_i, partial over curve chunks: grid (A2, NB2) --------------------------------
__global__ __launch_bounds__(256) void k_cov_w2(Ctx c) {
  const Dims& d = c.d;
  const int a2 = blockIdx.x, cb = blockIdx.y, tid = threadIdx.x;
  const Dir2 a = dir2_of(d, a2);
  if (!dir2_updated(c, a)) return;
  const int CH = (d.n + c.NB2 - 1) / c.NB2;
  const int i0 = cb * CH, i1 = min(d.n, i0 + CH);
  double acc0 = 0.0, acc1 = 0.0;
  // (batches of 16 curves: the loads of a batch are in flight together instead of one L2 round trip per curve)
  for (int ib = i0; ib < i1; ib += 16) {
    double w2[16], r0[16], r1[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int i = min(ib + u, i1 - 1);
      const double w = w_of(c, i, a);
      w2[u] = (ib + u < i1) ? w * w : 0.0;
      const double* r = c.rec + (size_t)i * d.LREC;
      r0[u] = (tid < d.LG) ? r[tid] : 0.0;
      r1[u] = (tid + 256 < d.LG) ? r[tid + 256] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) { acc0 += w2[u] * r0[u]; acc1 += w2[u] * r1[u]; }
  }
  double* out = c.w2_part + ((size_t)a2 * c.NB2 + cb) * d.LG;
  if (tid < d.LG) out[tid] = acc0;
  if (tid + 256 < d.LG) out[tid + 256] = acc1;
}

// ---- C_a, L_a z_a for every eta / Xi direction: grid A2 -----------------------------------------
template <int PP>
__global__ __launch_bounds__(256) void k_cov_factor(Ctx c) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const Dims& d = c.d;
  const int P = d.P, K = d.K, M = d.M, D = d.D, tid = threadIdx.x;
  const int a2 = blockIdx.x;
  const Dir2 a = dir2_of(d, a2);
  if (!dir2_updated(c, a)) return;
  double* S = smem;
  double* X = S + PP * PP;
  double* zv = X + PP * PP;
  double* hb = zv + PP;            // LG
  for (int e = tid; e < d.LG; e += 256) {
    double s = 0.0;
    for (int cb0 = 0; cb0 < c.NB2; cb0 += 16) {
      double v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = (cb0 + u < c.NB2) ? c.w2_part[((size_t)a2 * c.NB2 + cb0 + u) * d.LG + e] : 0.0;
#pragma unroll
      for (int u = 0; u < 16; ++u) s += v[u];
    }
    hb[e] = s;
    c.H2aa[(size_t)a2 * d.LG + e] = s;
  }
  __syncthreads();
  const Dyn* dyn = c.dyn;
  const double f = dyn->beta / dyn->sigma2;
  double tt = 1.0;
  if (a.mt > 0)     // tilde_tau_xi(j, m, d) = prod_{m' <= m} delta_xi(j, m', d), BFMMM.h:3983-3990
    for (int m2 = 0; m2 < a.mt; ++m2) tt *= c.delta_xi[a.j + (size_t)K * (m2 + (size_t)M * a.dd)];
  const double te = c.tau_eta[a.j + (size_t)K * a.dd];
  const double* gx = c.gamma_xi + (size_t)a.j * P * D * M;
  for (int e = tid; e < PP * PP; e += 256) {
    const int p = e & (PP - 1), q = e / PP;
    double v = 0.0;
    if (p < P && q < P) {
      const int lo = min(p, q), dd = max(p, q) - lo;
      v = (dd <= d.BW) ? f * hb[dd * P + lo] : 0.0;
      if (a.mt == 0) {
        if (d.mv) { if (p == q) v += 1.0 / te; }
        else v += te * c.Pmat[p + (size_t)P * q];                                   // UpdateEta.h:84
      } else if (p == q) {
        v += tt * gx[p + (size_t)P * (a.dd + (size_t)D * (a.mt - 1))];            // UpdateXi.h:76-78
      }
    }
    S[e] = v;
    X[e] = 0.0;
  }
  if (tid >= 64 && tid < 64 + P) {
    const RngKey key = make_key(c.seed, c.chain, dyn->iter, dyn->tt_step);
    const int p = tid - 64;
    if (a.mt == 0) zv[p] = rnorm(key, UPD_ETA, (uint32_t)((a.dd * K + a.j) * P + p));
    else zv[p] = rnorm(key, UPD_XI, (uint32_t)((((a.j * M + (a.mt - 1)) * D + a.dd) * P) + p));
  }
  __syncthreads();
  const bool bad = factor_core<PP>(S, X, zv, P, d.BWP, c.C2 + (size_t)a2 * P * P, nullptr, c.Lz2 + (size_t)a2 * P, tid);
  if (bad && tid == 0) atomicOr(&c.dyn->status, 1u);
}

// ---- one step of the eta / Xi sweep: ONE launch per direction -----------------------------------------
// k_cov_step(a_prev, a_next) is the whole dependent chain between two directions:
//   1. (a_prev >= 0) every workgroup sums the partial sums the previous launch left (fixed order), forms
//      rhs = (beta/sigma^2) [ sum_i w_i (s_i - g_i) + H_aa theta_old ] and draws theta_new = C rhs + L z  -- redundantly
//      and bit-identically in all workgroups, so that the launch boundary is the only grid-wide synchronisation of
//      the step (a second launch for the draw costs as much as the step itself); workgroup 0 records theta_new,
//   2. applies c_i += w_i delta, g_i += w_i G_i delta to its curves,
//   3. (a_next >= 0) accumulates sum_i w_i (s_i - g_i) of the next direction into the other parity of step_part;
//      (a_next < 0) leaves the residual sums of squares of the final state for the log-likelihood / sigma^2.
// State-independent operands (C_a, L_a z_a, H_aa, the curve records) are requested before the partial sums so
// that one memory round trip covers all of them.
template <int LPC>
__device__ inline double gsum_l(double v) {
#pragma unroll
  for (int o = LPC / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, LPC);
  return v;
}

constexpr int COV_CPG = 4;     // curves per lane group and launch

template <int BW, int LPC>
__global__ __launch_bounds__(256) void k_cov_step(Ctx c, int a_prev, int a_next, int par_prev) {
  constexpr int GPB = 256 / LPC;          // lane groups per workgroup
  constexpr int LR = 256 / LPC;           // lanes per row of C in the draw (row = tid / LR)
  constexpr int QPL = LPC / LR;           // columns per lane
  __shared__ double sDl[PMAX + 2 * BWMAX + 2];
  __shared__ double sTh[PMAX + 2 * BWMAX + 2];
  __shared__ double sRhs[PMAX];
  __shared__ double sAcc[GPB][PMAX];
  const Dims& d = c.d;
  const int n = d.n, P = d.P, tid = threadIdx.x;
  const int grp = tid / LPC, lp = tid % LPC;
  const Dyn* dyn = c.dyn;
  const bool act = lp < P;
  const int pc = min(lp, P - 1);

  // ---- requests that do not depend on the previous step ----
  const int row = tid / LR, rl = tid % LR;
  double cr[QPL];
  double lz = 0.0;
  int ax_prev = 0;
  if (a_prev >= 0) {
    const Dir2 ap = dir2_of(d, a_prev);
    ax_prev = (ap.j * (d.M + 1) + ap.mt) * d.D + ap.dd;
    const double* Cg = c.C2 + (size_t)a_prev * P * P;
#pragma unroll
    for (int u = 0; u < QPL; ++u) {
      const int q = rl + u * LR;
      cr[u] = (row < P && q < P) ? Cg[row + (size_t)P * q] : 0.0;
    }
    if (row < P) lz = c.Lz2[(size_t)a_prev * P + row];
    for (int e = tid; e < PMAX + 2 * BWMAX + 2; e += 256) {
      const int p = e - BW;
      sTh[e] = (p >= 0 && p < P) ? c.thetaX[(size_t)ax_prev * P + p] : 0.0;
    }
  }
  double g[COV_CPG][BW + 1], gl[COV_CPG][BW + 1], sv[COV_CPG], cf[COV_CPG], gv[COV_CPG], wp[COV_CPG], wn[COV_CPG], yy[COV_CPG];
  const int ibase = (blockIdx.x * GPB + grp) * COV_CPG;
#pragma unroll
  for (int u = 0; u < COV_CPG; ++u) {
    const int i = ibase + u;
    const bool valid = i < n;
    const int ic = valid ? i : 0;
    const double* rec = c.rec + (size_t)ic * d.LREC;
#pragma unroll
    for (int dd = 0; dd <= BW; ++dd) {
      const double vg = rec[dd * P + pc], vl = rec[dd * P + max(pc - dd, 0)];
      g[u][dd] = (act && valid) ? vg : 0.0;
      gl[u][dd] = (act && valid && dd > 0 && lp - dd >= 0) ? vl : 0.0;
    }
    const double s0 = rec[d.LG + pc], c0 = c.cfull[(size_t)ic * P + pc], g0 = c.gfull[(size_t)ic * P + pc];
    sv[u] = (act && valid) ? s0 : 0.0;
    cf[u] = (act && valid) ? c0 : 0.0;
    gv[u] = (act && valid) ? g0 : 0.0;
    yy[u] = valid ? rec[d.LG + P] : 0.0;
    wp[u] = (valid && a_prev >= 0) ? w_of(c, ic, dir2_of(d, a_prev)) : 0.0;
    wn[u] = (valid && a_next >= 0) ? w_of(c, ic, dir2_of(d, a_next)) : 0.0;
  }

  // ---- the draw of direction a_prev ----
  if (a_prev >= 0) {
    const double* sp = c.step_part + (size_t)par_prev * c.NBS * P;
    // (the partial sums were written by other XCDs: every load is a trip to memory, so they go out in batches)
    if (lp < P) {
      double s = 0.0;
      for (int b0 = grp; b0 < c.NBS; b0 += GPB * 16) {
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const int b = b0 + u * GPB;
          v[u] = (b < c.NBS) ? sp[(size_t)b * P + lp] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) s += v[u];
      }
      sAcc[grp][lp] = s;
    }
    __syncthreads();
    const double f = dyn->beta / dyn->sigma2;
    if (tid < P) {
      const double* Hb = c.H2aa + (size_t)a_prev * d.LG;
      const double* th = sTh + BW + tid;
      double hv = Hb[tid] * th[0];
#pragma unroll
      for (int dd = 1; dd <= BW; ++dd) hv += Hb[dd * P + tid] * th[dd] + Hb[dd * P + max(tid - dd, 0)] * th[-dd];
      double ps = 0.0;
#pragma unroll
      for (int gq = 0; gq < GPB; ++gq) ps += sAcc[gq][tid];
      sRhs[tid] = f * (ps + hv);
    }
    __syncthreads();
    double mean = 0.0;
#pragma unroll
    for (int u = 0; u < QPL; ++u) {
      const int q = rl + u * LR;
      mean += cr[u] * sRhs[min(q, P - 1)];
    }
    mean = gsum_l<LR>(mean);
    if (tid < BW || (tid >= 64 && tid < 64 + BWMAX + 2)) {     // zero pads of the delta row
      if (tid < BW) sDl[tid] = 0.0;
      else if (BW + P + (tid - 64) < PMAX + 2 * BWMAX + 2) sDl[BW + P + (tid - 64)] = 0.0;
    }
    if (rl == 0 && row < P) {
      const double nw = mean + lz;
      sDl[BW + row] = nw - sTh[BW + row];
      if (blockIdx.x == 0) c.thetaN[(size_t)ax_prev * P + row] = nw;
    }
    __syncthreads();
    // ---- apply it to this workgroup's curves ----
    const double* dl = sDl + BW + pc;
#pragma unroll
    for (int u = 0; u < COV_CPG; ++u) {
      double Gd = g[u][0] * dl[0];
#pragma unroll
      for (int dd = 1; dd <= BW; ++dd) Gd += g[u][dd] * dl[dd] + gl[u][dd] * dl[-dd];
      cf[u] += wp[u] * dl[0];
      gv[u] += wp[u] * Gd;
      const int i = ibase + u;
      if (act && i < n) { c.cfull[(size_t)i * P + lp] = cf[u]; c.gfull[(size_t)i * P + lp] = gv[u]; }
    }
  }

  // ---- partial sums for the next direction, or the residual sums of the final state ----
  double acc = 0.0;
  if (a_next >= 0) {
#pragma unroll
    for (int u = 0; u < COV_CPG; ++u) acc += wn[u] * (sv[u] - gv[u]);
  } else {
#pragma unroll
    for (int u = 0; u < COV_CPG; ++u) {
      const double cs = gsum_l<LPC>(cf[u] * sv[u]), cg = gsum_l<LPC>(cf[u] * gv[u]);
      acc += yy[u] - 2.0 * cs + cg;          // identical on every lane of the group
    }
  }
  if (lp < PMAX) sAcc[grp][lp] = acc;
  __syncthreads();
  if (a_next >= 0) {
    if (tid < P) {
      double s = 0.0;
#pragma unroll
      for (int gq = 0; gq < GPB; ++gq) s += sAcc[gq][tid];
      c.step_part[((size_t)(par_prev ^ 1) * c.NBS + blockIdx.x) * P + tid] = s;
    }
  } else {
    if (tid == 0) {
      double s = 0.0;
#pragma unroll
      for (int gq = 0; gq < GPB; ++gq) s += sAcc[gq][0];
      c.rss_part[blockIdx.x] = s;
    }
    // rss_part is read over nblk_curve entries by the log-likelihood / sigma^2 jobs
    if (blockIdx.x == 0)
      for (int b = c.NBS + tid; b < c.nblk_curve; b += 256) c.rss_part[b] = 0.0;
  }
}

// ---- tau_eta, delta_xi, A_xi, gamma_xi and the chain slots of the covariate blocks: one workgroup ----
__global__ __launch_bounds__(256) void k_cov_hyper(Ctx c) {
  __shared__ double red[256];
  __shared__ double Sk[KMAX * 16];
  const Dims& d = c.d;
  const int P = d.P, K = d.K, M = d.M, D = d.D, tid = threadIdx.x;
  Dyn* dyn = c.dyn;
  const uint32_t mask = c.mask, slot = dyn->slot;
  const RngKey key = make_key(c.seed, c.chain, dyn->iter, dyn->tt_step);
  if (tid == 0) dyn->pend_dir = -1;
  // ---- commit this iteration's eta / Xi draws (k_cov_step leaves them in thetaN) ----
  for (int e = tid; e < c.A2 * P; e += 256) {
    const int a2 = e / P, p = e - a2 * P;
    const Dir2 a = dir2_of(d, a2);
    if (!dir2_updated(c, a)) continue;
    const size_t ax = (size_t)((a.j * (M + 1) + a.mt) * D + a.dd);
    c.thetaX[ax * P + p] = c.thetaN[ax * P + p];
  }
  __syncthreads();
  // ---- tau_eta (UpdateTau.h:75-95; MV :106-124) ----
  if (mask & U_TAU_ETA) {
    for (int j = 0; j < K; ++j)
      for (int i = 0; i < D; ++i) {
        const double* e = c.thetaX + (size_t)((j * (M + 1)) * D + i) * P;
        double acc = 0.0;
        if (tid < P) {
          double s = 0.0;
          if (d.mv) s = e[tid];
          else
            for (int q = 0; q < P; ++q) s += c.Pmat[tid + (size_t)P * q] * e[q];
          acc = e[tid] * s;
        }
        const double qf = block_sum256(acc, red);
        if (tid == 0) {
          const double aa = c.h.alpha_eta + (P / 2);                         // integer division, UpdateTau.h:87
          const double bb = c.h.beta_eta + (0.5 * qf);
          const double g = rgamma(key, UPD_TAU_ETA, (uint32_t)(j * D + i), aa, 1.0 / bb);
          c.tau_eta[j + (size_t)K * i] = d.mv ? (1.0 / g) : g;
        }
      }
    __syncthreads();
  }
  const bool xi_on = c.covariance_adj && d.MD > 1;
  // ---- delta_xi (UpdateDelta.h:76-124), order (d, k, i) ----
  if ((mask & U_DELTA_XI) && xi_on) {
    for (int dd = 0; dd < D; ++dd) {
      if (tid < K * M) {
        const int k = tid / M, m = tid - k * M;
        const double* xk = c.thetaX + (size_t)((k * (M + 1) + m + 1) * D + dd) * P;
        const double* gx = c.gamma_xi + (size_t)k * P * D * M;
        double acc = 0.0;
        for (int p = 0; p < P; ++p) acc += gx[p + (size_t)P * (dd + (size_t)D * m)] * (xk[p] * xk[p]);
        Sk[tid] = acc;
      }
      __syncthreads();
      if (tid < K) {
        const int k = tid;
        for (int i = 0; i < M; ++i) {
          double param1, param2 = 1.0;
          auto DX = [&](int nn) { return c.delta_xi[k + (size_t)K * (nn + (size_t)M * dd)]; };
          if (i == 0) {
            param1 = c.A_xi[k + (size_t)K * (0 + 2 * (size_t)dd)] + ((P * M) * 0.5);
            param2 += 0.5 * Sk[k * M + 0];
            double tt = 1.0;
            for (int m = 1; m < M; ++m) { tt *= DX(m); param2 += 0.5 * tt * Sk[k * M + m]; }
          } else {
            param1 = c.A_xi[k + (size_t)K * (1 + 2 * (size_t)dd)] + ((P * (M - i)) * 0.5);
            for (int m = i; m < M; ++m) {
              double tt = 1.0;
              for (int nn = 0; nn <= m; ++nn)
                if (nn != i) tt *= DX(nn);
              param2 += 0.5 * tt * Sk[k * M + m];
            }
          }
          c.delta_xi[k + (size_t)K * (i + (size_t)M * dd)] =
              rgamma(key, UPD_DELTA_XI, (uint32_t)((dd * K + k) * M + i), param1, 1.0 / param2);
        }
      }
      __syncthreads();
    }
  }
  // ---- A_xi (UpdateA.h:137-205), cells (j, i, d) ----
  if ((mask & U_A_XI) && xi_on) {
    if (tid < K * 2 * D) {
      const int dd = tid % D, ji = tid / D, i = ji % 2, j = ji / 2;
      const bool first = (i == 0);
      const double sd = first ? (c.h.var_epsilon1 / c.h.beta1l) : (c.h.var_epsilon2 / c.h.beta2l);
      double* cell = &c.A_xi[j + (size_t)K * (i + 2 * (size_t)dd)];
      const double cur = *cell;
      const uint32_t idx = (uint32_t)((j * 2 + i) * D + dd);
      const double na = rtruncnorm_lo(key, UPD_AXI_PROP, idx, cur, sd, 0.0);
      const double* drow = c.delta_xi + j + (size_t)K * M * dd;     // delta_xi.slice(d).row(j), stride K
      double l0, l1;
      if (first) {
        l0 = -logGamma_ref(cur) + (cur - 1) * log(drow[0]) + (c.h.alpha1l - 1) * log(cur) - (cur * c.h.beta1l);
        l1 = -logGamma_ref(na) + (na - 1) * log(drow[0]) + (c.h.alpha1l - 1) * log(na) - (na * c.h.beta1l);
      } else {
        const double x = M - 1;
        l0 = -x * logGamma_ref(cur) + (c.h.alpha2l - 1) * log(cur) - (cur * c.h.beta2l);
        l1 = -x * logGamma_ref(na) + (c.h.alpha2l - 1) * log(na) - (na * c.h.beta2l);
        for (int q = 1; q < M; ++q) {
          const double lg = log(drow[(size_t)q * K]);
          l0 = l0 + (cur - 1) * lg;
          l1 = l1 + (na - 1) * lg;
        }
      }
      const double acc = (l1 + dtruncnorm_lo_log(cur, na, sd, 0.0)) - l0 - dtruncnorm_lo_log(na, cur, sd, 0.0);
      const double u = runif(key, UPD_AXI_ACC, idx);
      if (log(u) < acc) *cell = na;
    }
    __syncthreads();
  }
  // ---- gamma_xi (UpdateGamma.h:48-72), order (k, i = d, l = p, j = m) ----
  if ((mask & U_GAMMA_XI) && xi_on) {
    const int tot = K * D * P * M;
    for (int e = tid; e < tot; e += 256) {
      const int jm = e % M, r1 = e / M, l = r1 % P, r2 = r1 / P, i = r2 % D, k = r2 / D;
      double ph = 1.0;
      for (int j2 = 0; j2 <= jm; ++j2) ph *= c.delta_xi[k + (size_t)K * (j2 + (size_t)M * i)];
      const double x = c.thetaX[(size_t)((k * (M + 1) + jm + 1) * D + i) * P + l];
      c.gamma_xi[(size_t)k * P * D * M + l + (size_t)P * (i + (size_t)D * jm)] =
          rgamma(key, UPD_GAMMA_XI, (uint32_t)e, (c.h.nu_1 + 1) / 2, 2 / (c.h.nu_1 + ph * (x * x)));
    }
  }
  __syncthreads();
  // ---- chain slots (reference layouts: eta P x D x K; xi / gamma_xi K cubes P x D x M; ...) ----
  {
    double* s_eta = c.c_eta + (size_t)slot * P * D * K;
    for (int e = tid; e < P * D * K; e += 256) {
      const int p = e % P, r = e / P, dd = r % D, k = r / D;
      s_eta[e] = c.thetaX[(size_t)((k * (M + 1)) * D + dd) * P + p];
    }
    double* s_xi = c.c_xi + (size_t)slot * K * P * D * M;
    double* s_gx = c.c_gamma_xi + (size_t)slot * K * P * D * M;
    for (int e = tid; e < K * P * D * M; e += 256) {
      const int p = e % P, r = e / P, dd = r % D, r2 = r / D, m = r2 % M, k = r2 / M;
      s_xi[e] = c.thetaX[(size_t)((k * (M + 1) + m + 1) * D + dd) * P + p];
      s_gx[e] = c.gamma_xi[e];
    }
    for (int e = tid; e < K * D; e += 256) c.c_tau_eta[(size_t)slot * K * D + e] = c.tau_eta[e];
    for (int e = tid; e < K * M * D; e += 256) c.c_delta_xi[(size_t)slot * K * M * D + e] = c.delta_xi[e];
    for (int e = tid; e < K * 2 * D; e += 256) c.c_A_xi[(size_t)slot * K * 2 * D + e] = c.A_xi[e];
  }
}

// ---- host launchers -------------------------------------------------------------------------
template <int BW>
static void launch_step_bw(const Ctx& c, int a_prev, int a_next, int par_prev, hipStream_t st) {
  if (c.d.P <= 32) hipLaunchKernelGGL((k_cov_step<BW, 32>), dim3(c.NBS), dim3(256), 0, st, c, a_prev, a_next, par_prev);
  else hipLaunchKernelGGL((k_cov_step<BW, 64>), dim3(c.NBS), dim3(256), 0, st, c, a_prev, a_next, par_prev);
}

static void launch_step(const Ctx& c, int a_prev, int a_next, int par_prev, hipStream_t st) {
  switch (c.d.BW) {
    case 0: launch_step_bw<0>(c, a_prev, a_next, par_prev, st); break;
    case 1: launch_step_bw<1>(c, a_prev, a_next, par_prev, st); break;
    case 2: launch_step_bw<2>(c, a_prev, a_next, par_prev, st); break;
    case 3: launch_step_bw<3>(c, a_prev, a_next, par_prev, st); break;
    case 4: launch_step_bw<4>(c, a_prev, a_next, par_prev, st); break;
    default: launch_step_bw<5>(c, a_prev, a_next, par_prev, st); break;
  }
}

// curve blocks of k_cov_step for nblk_curve blocks of the per-curve kernels
int cov_step_blocks(int nblk_curve) { return (nblk_curve + COV_CPG - 1) / COV_CPG; }

// the eta / Xi part of one iteration (after k_curve_chi has stored c_i, g_i)
void launch_cov_block(const Ctx& c, hipStream_t st) {
  const Dims& d = c.d;
  const bool do_eta = (c.mask & U_ETA) != 0;
  const bool do_xi = (c.mask & U_XI) != 0 && c.covariance_adj && d.MD > 1;
  int a_prev = -1, par = 0;
  if (do_eta || do_xi) {
    hipLaunchKernelGGL(k_cov_w2, dim3(c.A2, c.NB2), dim3(256), 0, st, c);
    const int PP = (d.P <= 32) ? 32 : 64;
    const size_t lds = (2 * (size_t)PP * PP + PP + d.LG) * sizeof(double);
    if (PP == 32) hipLaunchKernelGGL(k_cov_factor<32>, dim3(c.A2), dim3(256), lds, st, c);
    else hipLaunchKernelGGL(k_cov_factor<64>, dim3(c.A2), dim3(256), lds, st, c);
    const int n_eta = d.K * d.D;
    for (int a2 = 0; a2 < c.A2; ++a2) {
      if (a2 < n_eta ? !do_eta : !do_xi) continue;
      launch_step(c, a_prev, a2, par, st);      // partial sums of a2 go to parity par ^ 1
      a_prev = a2;
      par ^= 1;
    }
  }
  launch_step(c, a_prev, -1, par, st);      // the last draw, and the residual sums for the log-likelihood
  hipLaunchKernelGGL(k_cov_hyper, dim3(1), dim3(256), 0, st, c);
}

void prepare_cov_kernels() {
  set_max_lds((const void*)k_cov_factor<32>);
  set_max_lds((const void*)k_cov_factor<64>);
}

}  // namespace bfmmm
