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
// (the c_i / g_i update of step s is applied lazily at the start of step s+1).
// This round's implementation favours simplicity over speed: two small launches per direction.
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
  for (int i = i0; i < i1; ++i) {
    const double w = w_of(c, i, a);
    const double w2 = w * w;
    const double* r = c.rec + (size_t)i * d.LREC;
    if (tid < d.LG) acc0 += w2 * r[tid];
    if (tid + 256 < d.LG) acc1 += w2 * r[tid + 256];
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
    for (int cb = 0; cb < c.NB2; ++cb) s += c.w2_part[((size_t)a2 * c.NB2 + cb) * d.LG + e];
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

// ---- per-curve part of one step: apply the pending update, accumulate w (s - g) ------------------
template <int LPC>
__device__ inline double gsum_l(double v) {
#pragma unroll
  for (int o = LPC / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, LPC);
  return v;
}

// a2 >= 0: step for direction a2 (partials -> step_part); a2 == -1: final pass (residual sums -> rss_part)
template <int BW, int LPC>
__global__ __launch_bounds__(256) void k_cov_accum(Ctx c, int a2) {
  __shared__ double sDl[PMAX + 2 * BWMAX + 2];
  __shared__ double sAcc[8][PMAX];
  constexpr int GPB = 256 / LPC;
  const Dims& d = c.d;
  const int n = d.n, P = d.P, tid = threadIdx.x;
  const int grp = tid / LPC, lp = tid % LPC;
  const Dyn* dyn = c.dyn;
  const int pd = dyn->pend_dir;
  for (int e = tid; e < PMAX + 2 * BWMAX + 2; e += 256) {
    const int p = e - BW;
    sDl[e] = (pd >= 0 && p >= 0 && p < P) ? c.delta_cur[p] : 0.0;
  }
  __syncthreads();
  const int i = blockIdx.x * GPB + grp;
  const bool valid = i < n, act = lp < P;
  double acc = 0.0;
  if (valid) {
    // band of G_i in registers (same access pattern as the per-curve kernels)
    const double* rec = c.rec + (size_t)i * d.LREC;
    const int pc = min(lp, P - 1);
    double g[BW + 1], gl[BW + 1];
#pragma unroll
    for (int dd = 0; dd <= BW; ++dd) {
      const double vg = rec[dd * P + pc], vl = rec[dd * P + max(pc - dd, 0)];
      g[dd] = act ? vg : 0.0;
      gl[dd] = (act && dd > 0 && lp - dd >= 0) ? vl : 0.0;
    }
    const double sv = act ? rec[d.LG + pc] : 0.0;
    double cf = act ? c.cfull[(size_t)i * P + pc] : 0.0;
    double gv = act ? c.gfull[(size_t)i * P + pc] : 0.0;
    if (pd >= 0) {
      const Dir2 ap = dir2_of(d, pd);
      const double wp = w_of(c, i, ap);
      const double* dl = sDl + BW + lp;
      double Gd = g[0] * dl[0];
#pragma unroll
      for (int dd = 1; dd <= BW; ++dd) Gd += g[dd] * dl[dd] + gl[dd] * dl[-dd];
      cf += wp * dl[0];
      gv += wp * Gd;
      if (act) { c.cfull[(size_t)i * P + lp] = cf; c.gfull[(size_t)i * P + lp] = gv; }
    }
    if (a2 >= 0) {
      const Dir2 a = dir2_of(d, a2);
      acc = w_of(c, i, a) * (sv - gv);
    } else {
      const double yy = rec[d.LG + P];
      const double cs = gsum_l<LPC>(cf * sv), cg = gsum_l<LPC>(cf * gv);
      acc = yy - 2.0 * cs + cg;          // identical on every lane of the group
    }
  }
  if (lp < PMAX) sAcc[grp][lp] = acc;
  __syncthreads();
  if (a2 >= 0) {
    if (tid < P) {
      double s = 0.0;
      for (int gq = 0; gq < GPB; ++gq) s += sAcc[gq][tid];
      c.step_part[(size_t)blockIdx.x * P + tid] = s;
    }
  } else if (tid == 0) {
    double s = 0.0;
    for (int gq = 0; gq < GPB; ++gq) s += sAcc[gq][0];
    c.rss_part[blockIdx.x] = s;
  }
}

// ---- the draw of one step: one workgroup -----------------------------------------------------------
__global__ __launch_bounds__(256) void k_cov_draw(Ctx c, int a2) {
  __shared__ double part[4][PMAX], rhs[PMAX], thold[PMAX + 2 * BWMAX + 2];
  const Dims& d = c.d;
  const int P = d.P, D = d.D, M = d.M, tid = threadIdx.x;
  Dyn* dyn = c.dyn;
  const Dir2 a = dir2_of(d, a2);
  const int ax = (a.j * (M + 1) + a.mt) * D + a.dd;
  const int p0 = tid & 63, seg = tid >> 6;
  if (p0 < P) {
    double s = 0.0;
    for (int b = seg; b < c.nblk_curve; b += 4) s += c.step_part[(size_t)b * P + p0];
    part[seg][p0] = s;
  }
  for (int e = tid; e < PMAX + 2 * BWMAX + 2; e += 256) {
    const int p = e - d.BW;
    thold[e] = (p >= 0 && p < P) ? c.thetaX[(size_t)ax * P + p] : 0.0;
  }
  __syncthreads();
  const double f = dyn->beta / dyn->sigma2;
  if (tid < P) {
    const double* Hb = c.H2aa + (size_t)a2 * d.LG;
    const double* th = thold + d.BW + tid;
    double hv = Hb[tid] * th[0];
    for (int dd = 1; dd <= d.BW; ++dd) hv += Hb[dd * P + tid] * th[dd] + Hb[dd * P + tid - dd] * th[-dd];
    rhs[tid] = f * (((part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid])) + hv);
  }
  __syncthreads();
  if (tid < P) {
    const double* Cg = c.C2 + (size_t)a2 * P * P;
    double mean = 0.0;
    for (int q = 0; q < P; ++q) mean += Cg[tid + (size_t)P * q] * rhs[q];
    const double nw = mean + c.Lz2[(size_t)a2 * P + tid];
    c.delta_cur[tid] = nw - thold[d.BW + tid];
    c.thetaX[(size_t)ax * P + tid] = nw;
  }
  if (tid == 0) dyn->pend_dir = a2;
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
static void launch_accum_bw(const Ctx& c, int a2, hipStream_t st) {
  if (c.d.P <= 32) hipLaunchKernelGGL((k_cov_accum<BW, 32>), dim3(c.nblk_curve), dim3(256), 0, st, c, a2);
  else hipLaunchKernelGGL((k_cov_accum<BW, 64>), dim3(c.nblk_curve), dim3(256), 0, st, c, a2);
}

static void launch_accum(const Ctx& c, int a2, hipStream_t st) {
  switch (c.d.BW) {
    case 0: launch_accum_bw<0>(c, a2, st); break;
    case 1: launch_accum_bw<1>(c, a2, st); break;
    case 2: launch_accum_bw<2>(c, a2, st); break;
    case 3: launch_accum_bw<3>(c, a2, st); break;
    case 4: launch_accum_bw<4>(c, a2, st); break;
    default: launch_accum_bw<5>(c, a2, st); break;
  }
}

// the eta / Xi part of one iteration (after k_curve_chi has stored c_i, g_i)
void launch_cov_block(const Ctx& c, hipStream_t st) {
  const Dims& d = c.d;
  const bool do_eta = (c.mask & U_ETA) != 0;
  const bool do_xi = (c.mask & U_XI) != 0 && c.covariance_adj && d.MD > 1;
  if (do_eta || do_xi) {
    hipLaunchKernelGGL(k_cov_w2, dim3(c.A2, c.NB2), dim3(256), 0, st, c);
    const int PP = (d.P <= 32) ? 32 : 64;
    const size_t lds = (2 * (size_t)PP * PP + PP + d.LG) * sizeof(double);
    if (PP == 32) hipLaunchKernelGGL(k_cov_factor<32>, dim3(c.A2), dim3(256), lds, st, c);
    else hipLaunchKernelGGL(k_cov_factor<64>, dim3(c.A2), dim3(256), lds, st, c);
    const int n_eta = d.K * d.D;
    for (int a2 = 0; a2 < c.A2; ++a2) {
      if (a2 < n_eta ? !do_eta : !do_xi) continue;
      launch_accum(c, a2, st);
      hipLaunchKernelGGL(k_cov_draw, dim3(1), dim3(256), 0, st, c, a2);
    }
  }
  launch_accum(c, -1, st);      // applies the last pending update and leaves the residual sums for the log-likelihood
  hipLaunchKernelGGL(k_cov_hyper, dim3(1), dim3(256), 0, st, c);
}

void prepare_cov_kernels() {
  set_max_lds((const void*)k_cov_factor<32>);
  set_max_lds((const void*)k_cov_factor<64>);
}

}  // namespace bfmmm
