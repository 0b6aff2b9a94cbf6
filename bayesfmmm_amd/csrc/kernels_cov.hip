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
// The directions are visited in the reference's order, D consecutive directions ("group") per launch: inside a group a
// direction sees the earlier draws through the in-group pair blocks H_su = sum_i w_s w_u G_i (k_cov_w2), the c_i / g_i
// update of a group is applied at the start of the next launch (k_cov_group).  Launch sequence of one iteration:
//   k_cov_prep -> k_cov_w2 -> k_cov_factor -> (K + K M) x k_cov_group -> k_cov_hyper.
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

constexpr int DMAX_COV = 8;     // covariates (bfmmm_set_covariates)

// in-group pair q = v(v+1)/2 + u  (u <= v < D)
__host__ __device__ inline void pair_uv(int q, int& u, int& v) {
  v = 0;
  while ((v + 1) * (v + 2) / 2 <= q) ++v;
  u = q - v * (v + 1) / 2;
}

// ---- preparation, one launch: (a) w_{a,i} for every eta / Xi direction (Z, chi, X are fixed during the block):
//      Wdir[i][a];  (b) the standard gamma variates of tau_eta, delta_xi, gamma_xi -- their shapes do not depend on
//      anything this iteration samples (delta_xi's uses A_xi, which is updated after it), so k_cov_hyper only scales
//      them: rgamma(shape, scale) == rgamma(shape, 1) * scale bit for bit (rng.hpp).  gstd2 = [K*D | K*M*D | K*D*P*M].
__global__ __launch_bounds__(256) void k_cov_prep(Ctx c0, int n_wblocks) {
  const Ctx c = chain_view(c0);      // chain blockIdx.z of the batch
  const Dims& d = c.d;
  if ((int)blockIdx.x < n_wblocks) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t tot = (size_t)d.n * c.A2;
    if (e >= tot) return;
    const int i = (int)(e / c.A2), a2 = (int)(e - (size_t)i * c.A2);
    c.Wdir[e] = w_of(c, i, dir2_of(d, a2));
    return;
  }
  const int P = d.P, K = d.K, M = d.M, D = d.D;
  const int n_tau = K * D, n_del = K * M * D, n_gam = K * D * P * M;
  const int e = ((int)blockIdx.x - n_wblocks) * 256 + threadIdx.x;
  const RngKey key = make_key(c.seed, c.chain, c.dyn->iter, c.dyn->tt_step);
  const bool xi_on = c.covariance_adj && d.MD > 1;
  if (e < n_tau) {
    if (c.mask & U_TAU_ETA) c.gstd2[e] = rgamma(key, UPD_TAU_ETA, (uint32_t)e, c.h.alpha_eta + (P / 2), 1.0);   // integer division, UpdateTau.h:87
  } else if (e < n_tau + n_del) {
    if ((c.mask & U_DELTA_XI) && xi_on) {
      const int idx = e - n_tau;                    // (dd*K + k)*M + i
      const int i = idx % M, k = (idx / M) % K, dd = idx / (M * K);
      const double shape = (i == 0) ? c.A_xi[k + (size_t)K * (0 + 2 * (size_t)dd)] + ((P * M) * 0.5)
                                    : c.A_xi[k + (size_t)K * (1 + 2 * (size_t)dd)] + ((P * (M - i)) * 0.5);
      c.gstd2[e] = rgamma(key, UPD_DELTA_XI, (uint32_t)idx, shape, 1.0);
    }
  } else if (e < n_tau + n_del + n_gam) {
    if ((c.mask & U_GAMMA_XI) && xi_on) c.gstd2[e] = rgamma(key, UPD_GAMMA_XI, (uint32_t)(e - n_tau - n_del), (c.h.nu_1 + 1) / 2, 1.0);
  }
}

// ---- sum_i w_a w_b G_i over the in-group direction pairs, partial over curve chunks: grid (groups, NB2) ----
// One workgroup = one group of D directions x one chunk of curves: a thread owns one record element e (two "halves"
// of the chunk side by side when LG <= 128) and keeps the NPG pair accumulators of the group in registers, so a
// record element is read once for all D(D+1)/2 pairs; the D weights of a curve are read from LDS, the pair weights
// formed in registers.
constexpr int NPG_MAX = DMAX_COV * (DMAX_COV + 1) / 2;
constexpr int W2_CH = 128;       // curves per chunk (NB2 = ceil(n / W2_CH))

// DX = D, a compile-time constant: with a run-time D the "v < D" guards of the pair loops are uniform branches around every
// group of FMAs (44 -> 23 us at config 3, D 5).
template <int DX>
__global__ __launch_bounds__(256) void k_cov_w2(Ctx c0) {
  const Ctx c = chain_view(c0);      // chain blockIdx.z of the batch
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const Dims& d = c.d;
  const int g = blockIdx.x, cb = blockIdx.y, tid = threadIdx.x;
  constexpr int D = DX, NPG = DX * (DX + 1) / 2;
  const int LG = d.LG;
  if (!dir2_updated(c, dir2_of(d, g * D))) return;
  const int i0 = cb * W2_CH, nc = min(d.n - i0, W2_CH);
  double* sW = sm;                       // W2_CH x D weights of the chunk (zero rows beyond its curves)
  double* sRed = sW + W2_CH * D;         // NPG x 128 (second half's sums)
  for (int e = tid; e < W2_CH * D; e += 256) {
    const int cl = e / D, sdir_ = e - cl * D;
    sW[e] = (cl < nc) ? c.Wdir[(size_t)(i0 + cl) * c.A2 + g * D + sdir_] : 0.0;
  }
  __syncthreads();
  const int LGR = (LG + 63) & ~63;
  const int NH = (LGR <= 128) ? 2 : 1;
  const int half = (NH == 2) ? tid / 128 : 0;
  const int e0 = (NH == 2) ? tid % 128 : tid;
  const int per = W2_CH / NH;                      // curves per half
  const int cl0 = half * per;
  for (int ep = 0; ep < LG; ep += 256) {
    const int e = ep + e0;
    const bool on = e < LG;
    double acc[NPG];
#pragma unroll
    for (int q = 0; q < NPG; ++q) acc[q] = 0.0;
    for (int cb0 = 0; cb0 < per; cb0 += 16) {
      double r[16];
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int cl = min(cl0 + cb0 + t, nc - 1);
        r[t] = on ? c.rec[(size_t)(i0 + max(cl, 0)) * d.LREC + e] : 0.0;
      }
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        // D broadcast LDS reads per curve; the D(D+1)/2 pair weights are formed in registers (one LDS read per FMA
        // bound the first version of this kernel)
        const double* wr = sW + (cl0 + cb0 + t) * D;
        double w[D];
#pragma unroll
        for (int v = 0; v < D; ++v) w[v] = wr[v];
#pragma unroll
        for (int v = 0; v < D; ++v) {
          const double wv = w[v] * r[t];
#pragma unroll
          for (int u = 0; u <= v; ++u) acc[v * (v + 1) / 2 + u] += w[u] * wv;
        }
      }
    }
    if (NH == 2) {
      __syncthreads();
      if (half == 1)
#pragma unroll
        for (int q = 0; q < NPG; ++q) sRed[q * 128 + e0] = acc[q];
      __syncthreads();
      if (half == 0 && on)
#pragma unroll
        for (int q = 0; q < NPG; ++q) c.w2_part[((size_t)(g * NPG + q) * c.NB2 + cb) * LG + e] = acc[q] + sRed[q * 128 + e0];
    } else if (on) {
#pragma unroll
      for (int q = 0; q < NPG; ++q) c.w2_part[((size_t)(g * NPG + q) * c.NB2 + cb) * LG + e] = acc[q];
    }
  }
}

// ---- H_ab for every in-group pair; C_a, L_a z_a for every eta / Xi direction: grid NPAIR -----------------
template <int PP>
__global__ __launch_bounds__(256) void k_cov_factor(Ctx c0) {
  const Ctx c = chain_view(c0);      // chain blockIdx.z of the batch
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const Dims& d = c.d;
  const int P = d.P, K = d.K, M = d.M, D = d.D, tid = threadIdx.x;
  const int pid = blockIdx.x;
  const int g = pid / c.NPG;
  int pu, pv;
  pair_uv(pid - g * c.NPG, pu, pv);
  const int a2 = g * D + pu;
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
      for (int u = 0; u < 16; ++u) v[u] = (cb0 + u < c.NB2) ? c.w2_part[((size_t)pid * c.NB2 + cb0 + u) * d.LG + e] : 0.0;
#pragma unroll
      for (int u = 0; u < 16; ++u) s += v[u];
    }
    hb[e] = s;
    c.H2aa[(size_t)pid * d.LG + e] = s;
  }
  if (pu != pv) return;      // off-diagonal pair: only the reduction
  __syncthreads();
  const Dyn* dyn = c.dyn;
  const double f = dyn->beta / dyn->sigma2;
  double tt = 1.0;
  if (a.mt > 0)     // tilde_tau_xi(j, m, d) = prod_{m' <= m} delta_xi(j, m', d), BFMMM.h:3983-3990
    for (int m2 = 0; m2 < a.mt; ++m2) tt *= c.delta_xi[a.j + (size_t)K * (m2 + (size_t)M * a.dd)];
  const double te = c.tau_eta[a.j + (size_t)K * a.dd];
  const double* gx = c.gamma_xi + (size_t)a.j * P * D * M;
  auto build_prec = [&]() {        // the full symmetric precision (zero beyond P)
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
  };
  build_prec();
  if (tid >= 64 && tid < 64 + P) {
    const RngKey key = make_key(c.seed, c.chain, dyn->iter, dyn->tt_step);
    const int p = tid - 64;
    if (a.mt == 0) zv[p] = rnorm(key, UPD_ETA, (uint32_t)((a.dd * K + a.j) * P + p));
    else zv[p] = rnorm(key, UPD_XI, (uint32_t)((((a.j * M + (a.mt - 1)) * D + a.dd) * P) + p));
  }
  __syncthreads();
  if (factor_core<PP>(S, X, zv, P, d.BWP, c.C2 + (size_t)a2 * P * P, nullptr, c.Lz2 + (size_t)a2 * P, tid)) {
    // singular to working accuracy: the reference's arma::pinv route (UpdateEta.h:85-87), factor_pinv here
    build_prec();
    __syncthreads();
    factor_pinv<PP>(S, X, zv, P, c.C2 + (size_t)a2 * P * P, nullptr, c.Lz2 + (size_t)a2 * P, tid, hb + d.LG);
  }
}

// ---- one GROUP of the eta / Xi sweep per launch ---------------------------------------------------------
// The K*D + K*M*D directions are visited in the reference's order, D consecutive directions ("group") per launch.
// k_cov_group(g_prev, g_next):
//   1. (g_prev >= 0) every workgroup sums the partial sums r_s = sum_i w_{s,i} (s_i - g_i) the previous launch left
//      for the D directions of g_prev (fixed order) and runs the D sequential draws itself -- redundantly and
//      bit-identically in all workgroups, so that the launch boundary is the only grid-wide synchronisation of a
//      group: inside the group, direction s sees the earlier draws through the pair blocks H_su = sum_i w_s w_u G_i,
//        r_s <- r_s - sum_{u<s} H_su delta_u,   rhs = (beta/sigma^2) (r_s + H_ss theta_old),   theta_new = C_s rhs + L_s z_s;
//      workgroup 0 records theta_new,
//   2. applies c_i += v_i, g_i += G_i v_i with v_i = sum_s w_{s,i} delta_s to its curves,
//   3. (g_next >= 0) accumulates the D partial sums of the next group into the other parity of step_part;
//      (g_next < 0) leaves the residual sums of squares of the final state for the log-likelihood / sigma^2.
// State-independent operands (the curve records, weights, pair blocks, C_s, L_s z_s) are requested before the partial
// sums so that they share one memory round trip.
template <int LPC>
__device__ inline double gsum_l(double v) {
#pragma unroll
  for (int o = LPC / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, LPC);
  return v;
}

// sum over the LR (8 or 4) adjacent lanes of a row, on the DPP path (an LDS-pipe shuffle costs ~100 clk on the chain)
template <int LR>
__device__ inline double rsum_dpp(double v) {
  v = dpp_add<0xB1>(v);                   // quad_perm [1,0,3,2]
  v = dpp_add<0x4E>(v);                   // quad_perm [2,3,0,1]
  if (LR == 8) v = dpp_add<0x141>(v);     // row_half_mirror
  return v;
}

constexpr int COV_CPG = 2;     // curves per lane group of k_cov_group
constexpr int COV_CPB32 = 32;  // curves per workgroup of k_cov_group (P <= 32: 32 lane groups of 32; P > 32: 16 groups of 64)
constexpr int DMAX = DMAX_COV;
constexpr int GT = 1024;       // threads of k_cov_group

// 1024 threads = four waves per SIMD: every phase of this kernel is bound by instruction issue and latency, not by
// bytes, and a wave alone on a SIMD issues only every ~5 clk.  One curve per lane group; the D sequential draws are
// done by the first 256 threads.
typedef double dbl2 __attribute__((ext_vector_type(2)));     // one 16-byte memory instruction

// DX > 0: D == DX exactly, a compile-time constant (built for the cubic-spline model, BW 3): 14.2 -> 13.4 us per launch at config 3
template <int BW, int LPC, int DX = 0>
__global__ __launch_bounds__(GT) void k_cov_group(Ctx c0, int g_prev, int g_next, int par_prev) {
  Ctx cx = chain_view(c0);           // chain blockIdx.z of the batch
  if constexpr (DX > 0) { cx.d.D = DX; cx.NPG = DX * (DX + 1) / 2; }
  const Ctx& c = cx;
  constexpr int GPB = GT / LPC;           // lane groups
  constexpr int CPB = GPB * COV_CPG;      // curves per workgroup
  constexpr int LR = 256 / LPC;           // lanes per row of C in the draw (row = tid / LR, tid < 256)
  constexpr int QPL = LPC / LR;           // columns per lane
  constexpr int NSEG = GT / 32;           // segments of the partial-sum reduction
  constexpr int PADW = PMAX + 2 * BW + 2;  // a coefficient row with BW zeros either side
  extern __shared__ __attribute__((aligned(16))) double sHp[];     // NPG x LG pair blocks of g_prev
  __shared__ double sDl[DMAX][PADW];
  __shared__ double sTh[DMAX][PADW];
  __shared__ double sR[DMAX * PMAX];
  __shared__ double sLz[DMAX * PMAX];
  __shared__ double sRhs[PMAX];
  __shared__ __attribute__((aligned(16))) double sAcc[GPB * DMAX * LPC];     // [grp][s][lp]; also the 32 x 160 scratch of the partial-sum reduction
  __shared__ double sWp[CPB][DMAX], sWn[CPB][DMAX];
  const Dims& d = c.d;
  const int n = d.n, P = d.P, D = d.D, DP = D * P, tid = threadIdx.x;
  const int grp = tid / LPC, lp = tid % LPC;
  const Dyn* dyn = c.dyn;
  const bool act = lp < P;
  const int pc = min(lp, P - 1);
  const bool chain = tid < 256;                      // wave-uniform
  const int row = (tid & 255) / LR, rl = tid % LR;
  const int cbase = blockIdx.x * CPB;
#ifdef COV_STAMPS
#define CST(k) do { if (blockIdx.x == 1 && tid == 0 && g_prev == 5) c.dyn->stamps[40 + (k)] = clock64(); } while (0)
#else
#define CST(k) do { } while (0)
#endif
  CST(0);

  // ---- requests that do not depend on the previous group: into registers first, so that everything this launch
  //      reads shares one memory round trip (addresses are clamped instead of predicated: no branches) ----
  // C_s: P <= 32: all D matrices are staged in LDS (after the pair blocks); P > 32: rows in registers, one direction ahead
  constexpr bool ALLC = (LPC == 32);
  constexpr int NCV = ALLC ? (DMAX * 32 * 32) / GT : 1;
  constexpr int NHR = 2;
  double cr[QPL];
  dbl2 cv2[(NCV + 1) / 2];
  double hpv[NHR], thv = 0.0, lzv = 0.0, wpv = 0.0, wnv = 0.0;
  const int totH = c.NPG * d.LG, totC = D * P * P;
  double* sC = sHp + ((totH + 1) & ~1);
  const int a0 = max(g_prev, 0) * D;
  const bool c_al = ((((size_t)a0 * P * P) & 1) == 0);       // C_a0 starts on a 16-byte boundary (uniform)
  if (g_prev >= 0) {
    if (ALLC) {
      const double* Cg = c.C2 + (size_t)a0 * P * P;
      if (c_al) {       // 16-byte loads: the kernel is bound by the number of memory instructions per CU
        const dbl2* Cg2 = (const dbl2*)Cg;
#pragma unroll
        for (int t = 0; t < NCV / 2; ++t) cv2[t] = Cg2[min(tid + GT * t, (totC + 1) / 2 - 1)];
      }
    } else if (chain) {
      const double* Cg = c.C2 + (size_t)a0 * P * P;
#pragma unroll
      for (int u = 0; u < QPL; ++u) {
        const int q = rl + u * LR;
        const double v = Cg[min(row, P - 1) + P * min(q, P - 1)];
        cr[u] = (row < P && q < P) ? v : 0.0;
      }
    }
    const double* Hg = c.H2aa + (size_t)g_prev * totH;
#pragma unroll
    for (int t = 0; t < NHR; ++t) hpv[t] = Hg[min(tid + GT * t, totH - 1)];
    {
      const int e = min(tid, D * PADW - 1);
      const int s = e / PADW, p = e - s * PADW - BW;
      const Dir2 as = dir2_of(d, a0 + s);
      const int ax = (as.j * (d.M + 1) + as.mt) * D + as.dd;
      const double v = c.thetaX[(size_t)ax * P + min(max(p, 0), P - 1)];
      thv = (p >= 0 && p < P) ? v : 0.0;
    }
    lzv = c.Lz2[(size_t)a0 * P + min(tid, DP - 1)];
  }
  {
    const int e = min(tid, CPB * D - 1);
    const int cl = e / D, s = e - cl * D;
    const int i = min(cbase + cl, n - 1);
    const double vp = c.Wdir[(size_t)i * c.A2 + a0 + s], vn = c.Wdir[(size_t)i * c.A2 + max(g_next, 0) * D + s];
    wpv = (cbase + cl < n && g_prev >= 0) ? vp : 0.0;
    wnv = (cbase + cl < n && g_next >= 0) ? vn : 0.0;
  }
  auto stage_to_lds = [&]() {
    if (g_prev >= 0) {
#pragma unroll
      for (int t = 0; t < NHR; ++t) { const int e = tid + GT * t; if (e < totH) sHp[e] = hpv[t]; }
      const double* Hg = c.H2aa + (size_t)g_prev * totH;
      for (int e = tid + GT * NHR; e < totH; e += GT) sHp[e] = Hg[e];
      if (ALLC) {
        if (c_al) {
#pragma unroll
          for (int t = 0; t < NCV / 2; ++t) { const int e = tid + GT * t; if (2 * e < totC) ((dbl2*)sC)[e] = cv2[t]; }
        } else {
          const double* Cg = c.C2 + (size_t)a0 * P * P;
          for (int e = tid; e < totC; e += GT) sC[e] = Cg[e];
        }
      }
      if (tid < D * PADW) { (&sTh[0][0])[tid] = thv; (&sDl[0][0])[tid] = 0.0; }
      if (tid < DP) sLz[tid] = lzv;
    }
    if (tid < CPB * D) { (&sWp[0][0])[(tid / D) * DMAX + tid % D] = wpv; (&sWn[0][0])[(tid / D) * DMAX + tid % D] = wnv; }
  };
  // this lane group's curve (requested after the partial sums: it is not needed before the draws are done, and the
  // registers are scarce while the partial sums are in flight)
  double g[COV_CPG][BW + 1], gl[COV_CPG][BW + 1], sv[COV_CPG], cf[COV_CPG], gv[COV_CPG], yy[COV_CPG];
  auto load_curve = [&]() {
#pragma unroll
    for (int u = 0; u < COV_CPG; ++u) {
      const int icv = cbase + grp * COV_CPG + u;
      const bool valid = icv < n;
      const int ic = valid ? icv : 0;
      const double* rec = c.rec + (size_t)ic * d.LREC;
#pragma unroll
      for (int dd = 0; dd <= BW; ++dd) {
        const double vg = rec[dd * P + pc];
        g[u][dd] = (act && valid && lp + dd < P) ? vg : 0.0;
      }
#pragma unroll
      for (int dd = 0; dd <= BW; ++dd) {       // G(p, p-dd) = G(p-dd, p): the band entry lane p-dd has just loaded
        const double vl = (dd > 0) ? __shfl_up(g[u][dd], dd, LPC) : 0.0;
        gl[u][dd] = (dd > 0 && lp - dd >= 0) ? vl : 0.0;
      }
      const double s0 = rec[d.LG + pc], c0 = c.cfull[(size_t)ic * P + pc], g0 = c.gfull[(size_t)ic * P + pc];
      const double y0 = (g_next < 0) ? rec[d.LG + P] : 0.0;      // (only the final pass needs yy_i)
      sv[u] = (act && valid) ? s0 : 0.0;
      cf[u] = (act && valid) ? c0 : 0.0;
      gv[u] = (act && valid) ? g0 : 0.0;
      yy[u] = valid ? y0 : 0.0;
    }
  };

  if (g_prev < 0) { stage_to_lds(); load_curve(); }
  if (g_prev >= 0) {
    // ---- r_s of the D directions: fixed-order sum of the previous launch's partial sums.  They were written by
    //      other XCDs (every load is a trip to memory), so a thread's 20 go out together ----
    CST(1);
    const int DPS = (DP + 1) & ~1;                                   // row stride of step_part (even: 16-byte loads)
    const dbl2* sp2 = (const dbl2*)(c.step_part + (size_t)par_prev * c.NBS * DPS);
    const int seg = tid >> 5, l32 = tid & 31;
    bool staged = false;
    for (int v0 = 0; v0 < DPS; v0 += 192) {         // 96 pairs of values per round
      dbl2 accv[3];
#pragma unroll
      for (int w = 0; w < 3; ++w) accv[w] = dbl2{0.0, 0.0};
      for (int b0 = seg; b0 < c.NBS; b0 += NSEG * 4) {
        dbl2 t[3][4];
#pragma unroll
        for (int w = 0; w < 3; ++w)
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int v2 = v0 / 2 + w * 32 + l32, b = b0 + u * NSEG;
            const dbl2 x = sp2[(size_t)min(b, c.NBS - 1) * (DPS / 2) + min(v2, DPS / 2 - 1)];
            const bool on = (2 * v2 < DPS && b < c.NBS);
            t[w][u] = dbl2{on ? x.x : 0.0, on ? x.y : 0.0};
          }
        if (!staged) { stage_to_lds(); staged = true; }
#pragma unroll
        for (int w = 0; w < 3; ++w)
#pragma unroll
          for (int u = 0; u < 4; ++u) { accv[w].x += t[w][u].x; accv[w].y += t[w][u].y; }
      }
      if (!staged) { stage_to_lds(); staged = true; }       // (workgroups with fewer partial blocks than segments)
      // the segments of a value meet in LDS (sAcc is free until the accumulation phase)
      __syncthreads();
#pragma unroll
      for (int w = 0; w < 3; ++w) ((dbl2*)sAcc)[seg * 96 + w * 32 + l32] = accv[w];
      __syncthreads();
      if (tid < 192 && v0 + tid < DP) {
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < NSEG; ++q) s += sAcc[q * 192 + tid];
        sR[v0 + tid] = s;
      }
    }
    load_curve();
    __syncthreads();
    CST(2);
    // ---- the D sequential draws (first 256 threads; the others wait at the barriers) ----
    const double f = dyn->beta / dyn->sigma2;
#pragma unroll
    for (int s = 0; s < DMAX; ++s) {
      if (s < D) {
        const int a2 = g_prev * D + s;
        if (chain) {
          // rhs_s: lanes of a row share the pairs (u, s), u <= s
          double part = 0.0;
          if (row < P) {
            for (int u = rl; u <= s; u += LR) {
              const double* Hb = sHp + (size_t)(s * (s + 1) / 2 + u) * d.LG;
              const double* vec = (u < s) ? &sDl[u][BW + row] : &sTh[s][BW + row];
              double hv = Hb[row] * vec[0];
#pragma unroll
              for (int dd = 1; dd <= BW; ++dd) hv += Hb[dd * P + row] * vec[dd] + Hb[dd * P + max(row - dd, 0)] * vec[-dd];
              part += (u < s) ? -hv : hv;
            }
          }
          part = rsum_dpp<LR>(part);
          if (rl == 0 && row < P) sRhs[row] = f * (sR[s * P + row] + part);
        }
        __syncthreads();
        if (chain) {
          double mean = 0.0;
          if (ALLC) {
            const double* Cs = sC + s * P * P + min(row, P - 1);
#pragma unroll
            for (int u = 0; u < QPL; ++u) {
              const int q = rl + u * LR;
              mean += ((row < P && q < P) ? Cs[P * min(q, P - 1)] : 0.0) * sRhs[min(q, P - 1)];
            }
          } else {
#pragma unroll
            for (int u = 0; u < QPL; ++u) mean += cr[u] * sRhs[min(rl + u * LR, P - 1)];
          }
          mean = rsum_dpp<LR>(mean);
          if (!ALLC && s + 1 < D) {       // C of the next direction is on its way while this one is finished
            const double* Cg = c.C2 + (size_t)(a2 + 1) * P * P;
#pragma unroll
            for (int u = 0; u < QPL; ++u) {
              const int q = rl + u * LR;
              cr[u] = (row < P && q < P) ? Cg[row + (size_t)P * q] : 0.0;
            }
          }
          if (rl == 0 && row < P) {
            const double nw = mean + sLz[s * P + row];
            sDl[s][BW + row] = nw - sTh[s][BW + row];
            if (blockIdx.x == 0) {
              const Dir2 as = dir2_of(d, a2);
              const int ax = (as.j * (d.M + 1) + as.mt) * D + as.dd;
              c.thetaN[(size_t)ax * P + row] = nw;
            }
          }
        }
        __syncthreads();
      }
    }
    CST(3);
    // ---- apply the group to this lane group's curves ----
#pragma unroll
    for (int u = 0; u < COV_CPG; ++u) {
      const int cl = grp * COV_CPG + u;
      double v = 0.0;
      for (int s = 0; s < D; ++s) v += sWp[cl][s] * sDl[s][BW + pc];
      if (!act) v = 0.0;
      double Gd = g[u][0] * v;
#pragma unroll
      for (int dd = 1; dd <= BW; ++dd) {
        const double vu = __shfl_down(v, dd, LPC), vd = __shfl_up(v, dd, LPC);
        Gd += g[u][dd] * ((lp + dd < P) ? vu : 0.0) + gl[u][dd] * ((lp - dd >= 0) ? vd : 0.0);
      }
      cf[u] += v;
      gv[u] += Gd;
      const int icv = cbase + cl;
      if (act && icv < n) { c.cfull[(size_t)icv * P + lp] = cf[u]; c.gfull[(size_t)icv * P + lp] = gv[u]; }
    }
  }

  CST(4);
  // ---- partial sums for the next group, or the residual sums of the final state ----
  if (g_next >= 0) {
    __syncthreads();
#pragma unroll
    for (int s = 0; s < DMAX; ++s)
      if (s < D) {
        double a = 0.0;
#pragma unroll
        for (int u = 0; u < COV_CPG; ++u) a += sWn[grp * COV_CPG + u][s] * (sv[u] - gv[u]);
        sAcc[(grp * DMAX + s) * LPC + lp] = a;
      }
    __syncthreads();
    for (int e = tid; e < DP; e += GT) {
      const int s = e / P, p = e - s * P;
      double t = 0.0;
#pragma unroll 8
      for (int gq = 0; gq < GPB; ++gq) t += sAcc[(gq * DMAX + s) * LPC + p];
      c.step_part[((size_t)(par_prev ^ 1) * c.NBS + blockIdx.x) * ((DP + 1) & ~1) + e] = t;
    }
    CST(5);
  } else {
    double acc = 0.0;
#pragma unroll
    for (int u = 0; u < COV_CPG; ++u) {
      const double cs = gsum_l<LPC>(cf[u] * sv[u]), cg = gsum_l<LPC>(cf[u] * gv[u]);
      acc += yy[u] - 2.0 * cs + cg;          // identical on every lane of the group
    }
    __syncthreads();
    if (lp == 0) sAcc[grp] = acc;
    __syncthreads();
    if (tid == 0) {
      double t = 0.0;
      for (int gq = 0; gq < GPB; ++gq) t += sAcc[gq];
      c.rss_part[blockIdx.x] = t;
    }
    // rss_part is read over nblk_curve entries by the log-likelihood / sigma^2 jobs
    if (blockIdx.x == 0)
      for (int b = c.NBS + tid; b < c.nblk_curve; b += GT) c.rss_part[b] = 0.0;
  }
}

// ---- tau_eta, delta_xi, A_xi, gamma_xi and the chain slots of the covariate blocks: one workgroup ----
// (the gamma variates arrive as standard draws from k_cov_prep; what is left is sums, products and the A_xi step)
constexpr int HT = 1024;       // threads of k_cov_hyper: its loops are chains of dependent global round trips, so more lanes = fewer trips
__global__ __launch_bounds__(HT) void k_cov_hyper(Ctx c0) {
  const Ctx c = chain_view(c0);      // chain blockIdx.z of the batch
  extern __shared__ __attribute__((aligned(16))) double hsm[];
  const Dims& d = c.d;
  const int P = d.P, K = d.K, M = d.M, D = d.D, tid = threadIdx.x;
  Dyn* dyn = c.dyn;
  const uint32_t mask = c.mask, slot = dyn->slot;
  const RngKey key = make_key(c.seed, c.chain, dyn->iter, dyn->tt_step);
  const int n_tau = K * D, n_del = K * M * D;
  double* sE = hsm;                 // K*D x P    eta rows
  double* sPm = sE + K * D * P;     // P x P      penalty
  double* sSk = sPm + P * P;        // D x K*M
  double* sDX = sSk + D * K * M;    // (k*D + dd)*M + m    delta_xi
  double* sGs = sDX + D * K * M;    // D x K*M             standard gamma variates of the delta_xi step
  if (tid == 0) dyn->pend_dir = -1;
  // ---- commit this iteration's eta / Xi draws (k_cov_group leaves them in thetaN) ----
  for (int e = tid; e < c.A2 * P; e += HT) {
    const int a2 = e / P, p = e - a2 * P;
    const Dir2 a = dir2_of(d, a2);
    if (!dir2_updated(c, a)) continue;
    const size_t ax = (size_t)((a.j * (M + 1) + a.mt) * D + a.dd);
    c.thetaX[ax * P + p] = c.thetaN[ax * P + p];
  }
  for (int e = tid; e < K * M * D; e += HT) {
    const int m = e % M, dd = (e / M) % D, k = e / (M * D);
    sDX[e] = c.delta_xi[k + (size_t)K * (m + (size_t)M * dd)];
  }
  if (!d.mv)
    for (int e = tid; e < P * P; e += HT) sPm[e] = c.Pmat[e];
  __syncthreads();
  // ---- tau_eta (UpdateTau.h:75-95; MV :106-124): the K*D quadratic forms side by side, 32 lanes each ----
  if (mask & U_TAU_ETA) {
    for (int e = tid; e < K * D * P; e += HT) {
      const int p = e % P, pr = e / P, j = pr / D, i = pr - j * D;
      sE[e] = c.thetaX[(size_t)((j * (M + 1)) * D + i) * P + p];
    }
    __syncthreads();
    // (on the UPPER half of the workgroup: the lower waves go straight on to the sums of the delta_xi step, two round trips to
    //  memory that do not depend on tau_eta)
    const int grp = (tid >> 5) - HT / 64, l = tid & 31;
    for (int pr = grp; pr >= 0 && pr < K * D; pr += HT / 64) {
      const double* ev = sE + pr * P;
      double acc = 0.0;
      for (int p = l; p < P; p += 32) {
        double sv = 0.0;
        if (d.mv) sv = ev[p];
        else
          for (int q = 0; q < P; ++q) sv += sPm[p + P * q] * ev[q];
        acc += ev[p] * sv;
      }
      const double qf = gsum_l<32>(acc);
      if (l == 0) {
        const int j = pr / D, i = pr - j * D;
        const double bb = c.h.beta_eta + (0.5 * qf);
        const double gv = c.gstd2[j * D + i] * (1.0 / bb);
        c.tau_eta[j + (size_t)K * i] = d.mv ? (1.0 / gv) : gv;
      }
    }
  }
  const bool xi_on = c.covariance_adj && d.MD > 1;
  // ---- delta_xi (UpdateDelta.h:76-124), order (d, k, i): the (d, k) cells are independent of each other ----
  if ((mask & U_DELTA_XI) && xi_on) {
    for (int e = tid; e < D * K * M; e += HT) {
      const int dd = e / (K * M), km = e - dd * K * M, k = km / M, m = km - k * M;
      const double* xk = c.thetaX + (size_t)((k * (M + 1) + m + 1) * D + dd) * P;
      const double* gx = c.gamma_xi + (size_t)k * P * D * M + (size_t)P * (dd + (size_t)D * m);
      const double gsv = c.gstd2[n_tau + e];       // ((dd K + k) M + m: the order of the delta_xi variates)
      double acc = 0.0;
      for (int p0 = 0; p0 < P; p0 += 16) {
        double gq[16], xq[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) { const int p = min(p0 + t, P - 1); gq[t] = gx[p]; xq[t] = xk[p]; }
#pragma unroll
        for (int t = 0; t < 16; ++t) if (p0 + t < P) acc += gq[t] * (xq[t] * xq[t]);
      }
      sSk[e] = acc;
      sGs[e] = gsv;
    }
    __syncthreads();
    if (tid < K * D) {
      const int dd = tid / K, k = tid - dd * K;
      const double* Sk = sSk + dd * K * M;
      double* DX = sDX + (k * D + dd) * M;
      const double* gsd = sGs + (dd * K + k) * M;      // (the cell's standard gamma variates, staged with the sums above: read from
                                                       //  global memory inside the recursion each was a trip of its own)
      for (int i = 0; i < M; ++i) {
        double param2 = 1.0;
        if (i == 0) {
          param2 += 0.5 * Sk[k * M + 0];
          double tt = 1.0;
          for (int m = 1; m < M; ++m) { tt *= DX[m]; param2 += 0.5 * tt * Sk[k * M + m]; }
        } else {
          for (int m = i; m < M; ++m) {
            double tt = 1.0;
            for (int nn = 0; nn <= m; ++nn)
              if (nn != i) tt *= DX[nn];
            param2 += 0.5 * tt * Sk[k * M + m];
          }
        }
        const double nv = gsd[i] * (1.0 / param2);
        DX[i] = nv;
        c.delta_xi[k + (size_t)K * (i + (size_t)M * dd)] = nv;
      }
    }
    __syncthreads();
  }
  // ---- A_xi (UpdateA.h:137-205), cells (j, i, d) ----
  if ((mask & U_A_XI) && xi_on) {
    // cells (j, i, d): the i = 0 cells on wave 0 and the i = 1 cells on wave 1 -- the two kinds take different branches (different
    // densities), and side by side in one wave each lane paid for both
    if ((tid & 63) < K * D && tid < 128) {
      const int i = tid >> 6, jd = tid & 63, dd = jd % D, j = jd / D;
      const bool first = (i == 0);
      const double sd = first ? (c.h.var_epsilon1 / c.h.beta1l) : (c.h.var_epsilon2 / c.h.beta2l);
      double* cell = &c.A_xi[j + (size_t)K * (i + 2 * (size_t)dd)];
      const double cur = *cell;
      const uint32_t idx = (uint32_t)((j * 2 + i) * D + dd);
      const double na = rtruncnorm_lo(key, UPD_AXI_PROP, idx, cur, sd, 0.0);
      const double* drow = sDX + (j * D + dd) * M;     // delta_xi.slice(d).row(j)
      double l0, l1;
      if (first) {
        l0 = -logGamma_ref(cur) + (cur - 1) * log(drow[0]) + (c.h.alpha1l - 1) * log(cur) - (cur * c.h.beta1l);
        l1 = -logGamma_ref(na) + (na - 1) * log(drow[0]) + (c.h.alpha1l - 1) * log(na) - (na * c.h.beta1l);
      } else {
        const double x = M - 1;
        l0 = -x * logGamma_ref(cur) + (c.h.alpha2l - 1) * log(cur) - (cur * c.h.beta2l);
        l1 = -x * logGamma_ref(na) + (c.h.alpha2l - 1) * log(na) - (na * c.h.beta2l);
        for (int q = 1; q < M; ++q) {
          const double lg = log(drow[q]);
          l0 = l0 + (cur - 1) * lg;
          l1 = l1 + (na - 1) * lg;
        }
      }
      const double acc = (l1 + dtruncnorm_lo_log(cur, na, sd, 0.0)) - l0 - dtruncnorm_lo_log(na, cur, sd, 0.0);
      const double u = runif(key, UPD_AXI_ACC, idx);
      if (log(u) < acc) *cell = na;
    }
  }
  // ---- gamma_xi (UpdateGamma.h:48-72), order (k, i = d, l = p, j = m) ----
  if ((mask & U_GAMMA_XI) && xi_on) {
    const int tot = K * D * P * M;
    for (int e = tid; e < tot; e += HT) {
      const int jm = e % M, r1 = e / M, l = r1 % P, r2 = r1 / P, i = r2 % D, k = r2 / D;
      double ph = 1.0;
      for (int j2 = 0; j2 <= jm; ++j2) ph *= sDX[(k * D + i) * M + j2];
      const double x = c.thetaX[(size_t)((k * (M + 1) + jm + 1) * D + i) * P + l];
      c.gamma_xi[(size_t)k * P * D * M + l + (size_t)P * (i + (size_t)D * jm)] =
          c.gstd2[n_tau + n_del + e] * (2 / (c.h.nu_1 + ph * (x * x)));
    }
  }
  __syncthreads();
  // ---- chain slots (reference layouts: eta P x D x K; xi / gamma_xi K cubes P x D x M; ...) ----
  {
    double* s_eta = c.c_eta + (size_t)slot * P * D * K;
    for (int e = tid; e < P * D * K; e += HT) {
      const int p = e % P, r = e / P, dd = r % D, k = r / D;
      s_eta[e] = c.thetaX[(size_t)((k * (M + 1)) * D + dd) * P + p];
    }
    double* s_xi = c.c_xi + (size_t)slot * K * P * D * M;
    double* s_gx = c.c_gamma_xi + (size_t)slot * K * P * D * M;
    for (int e = tid; e < K * P * D * M; e += HT) {
      const int p = e % P, r = e / P, dd = r % D, r2 = r / D, m = r2 % M, k = r2 / M;
      s_xi[e] = c.thetaX[(size_t)((k * (M + 1) + m + 1) * D + dd) * P + p];
      s_gx[e] = c.gamma_xi[e];
    }
    for (int e = tid; e < K * D; e += HT) c.c_tau_eta[(size_t)slot * K * D + e] = c.tau_eta[e];
    for (int e = tid; e < K * M * D; e += HT) c.c_delta_xi[(size_t)slot * K * M * D + e] = c.delta_xi[e];
    for (int e = tid; e < K * 2 * D; e += HT) c.c_A_xi[(size_t)slot * K * 2 * D + e] = c.A_xi[e];
  }
}

// ---- host launchers -------------------------------------------------------------------------
#define COV_EACH_D(X) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8)
static_assert(DMAX_COV == 8, "COV_EACH_D lists 1 .. DMAX_COV");
template <int BW>
static void launch_group_bw(const Ctx& c, int g_prev, int g_next, int par_prev, hipStream_t st) {
  const size_t lds = ((size_t)c.NPG * c.d.LG + 2 + (c.d.P <= 32 ? (size_t)c.d.D * c.d.P * c.d.P + 2 : 0)) * sizeof(double);
  if constexpr (BW == 3) {           // instances with D exact
#define X(dx)                                                                                                                    \
    if (g_exact_instances && c.d.D == dx) {                                                                                                           \
      if (c.d.P <= 32) hipLaunchKernelGGL((k_cov_group<BW, 32, dx>), dim3(c.NBS, 1, c.nch), dim3(GT), lds, st, c, g_prev, g_next, par_prev);   \
      else hipLaunchKernelGGL((k_cov_group<BW, 64, dx>), dim3(c.NBS, 1, c.nch), dim3(GT), lds, st, c, g_prev, g_next, par_prev);  \
      return;                                                                                                                    \
    }
    COV_EACH_D(X)
#undef X
  }
  if (c.d.P <= 32) hipLaunchKernelGGL((k_cov_group<BW, 32>), dim3(c.NBS, 1, c.nch), dim3(GT), lds, st, c, g_prev, g_next, par_prev);
  else hipLaunchKernelGGL((k_cov_group<BW, 64>), dim3(c.NBS, 1, c.nch), dim3(GT), lds, st, c, g_prev, g_next, par_prev);
}

static void launch_group(const Ctx& c, int g_prev, int g_next, int par_prev, hipStream_t st) {
  switch (c.d.BW) {
    case 0: launch_group_bw<0>(c, g_prev, g_next, par_prev, st); break;
    case 1: launch_group_bw<1>(c, g_prev, g_next, par_prev, st); break;
    case 2: launch_group_bw<2>(c, g_prev, g_next, par_prev, st); break;
    case 3: launch_group_bw<3>(c, g_prev, g_next, par_prev, st); break;
    case 4: launch_group_bw<4>(c, g_prev, g_next, par_prev, st); break;
    case 5: launch_group_bw<5>(c, g_prev, g_next, par_prev, st); break;
    case BWMID: launch_group_bw<BWMID>(c, g_prev, g_next, par_prev, st); break;
    default: launch_group_bw<BWWIDE>(c, g_prev, g_next, par_prev, st); break;
  }
}

// curve blocks of k_cov_group for nblk_curve blocks of the per-curve kernels
// (the per-curve kernels take 256 / LPC curves per block, k_cov_group 1024 / LPC)
int cov_step_blocks(int nblk_curve) { return (nblk_curve + 4 * COV_CPG - 1) / (4 * COV_CPG); }
int cov_w2_chunks(int n) { return (n + W2_CH - 1) / W2_CH; }

// the eta / Xi part of one iteration (after k_curve_chi has stored c_i, g_i)
void launch_cov_block(const Ctx& c, hipStream_t st) {
  const Dims& d = c.d;
  const bool do_eta = (c.mask & U_ETA) != 0;
  const bool do_xi = (c.mask & U_XI) != 0 && c.covariance_adj && d.MD > 1;
  int g_prev = -1, par = 0;
  {
    const size_t tot = (size_t)d.n * c.A2;
    const int n_wblocks = (do_eta || do_xi) ? (int)((tot + 255) / 256) : 0;
    const int n_draws = d.K * d.D + d.K * d.M * d.D + d.K * d.D * d.P * d.M;
    hipLaunchKernelGGL(k_cov_prep, dim3(n_wblocks + (n_draws + 255) / 256, 1, c.nch), dim3(256), 0, st, c, n_wblocks);
  }
  if (do_eta || do_xi) {
    const size_t lds_w2 = ((size_t)W2_CH * d.D + (size_t)c.NPG * 128) * sizeof(double);
    switch (d.D) {
#define X(dx) case dx: hipLaunchKernelGGL(k_cov_w2<dx>, dim3(c.A2 / d.D, c.NB2, c.nch), dim3(256), lds_w2, st, c); break;
      COV_EACH_D(X)
#undef X
    }
    const int PP = (d.P <= 32) ? 32 : 64;
    const size_t lds = (2 * (size_t)PP * PP + PP + d.LG + 4 * PP + 2) * sizeof(double);      // + scratch of the pseudo-inverse route
    if (PP == 32) hipLaunchKernelGGL(k_cov_factor<32>, dim3(c.NPAIR, 1, c.nch), dim3(256), lds, st, c);
    else hipLaunchKernelGGL(k_cov_factor<64>, dim3(c.NPAIR, 1, c.nch), dim3(256), lds, st, c);
    const int n_eta_groups = d.K, n_groups = c.A2 / d.D;
    for (int g = 0; g < n_groups; ++g) {
      if (g < n_eta_groups ? !do_eta : !do_xi) continue;
      launch_group(c, g_prev, g, par, st);      // partial sums of g go to parity par ^ 1
      g_prev = g;
      par ^= 1;
    }
  }
  launch_group(c, g_prev, -1, par, st);      // the last group's draws, and the residual sums for the log-likelihood
  const size_t lds_h = ((size_t)d.K * d.D * d.P + (size_t)d.P * d.P + 3 * (size_t)d.D * d.K * d.M) * sizeof(double);
  hipLaunchKernelGGL(k_cov_hyper, dim3(1, 1, c.nch), dim3(HT), lds_h, st, c);
}

template <int BW>
static void prepare_group_bw() {
  set_max_lds((const void*)k_cov_group<BW, 32>);
  set_max_lds((const void*)k_cov_group<BW, 64>);
  if constexpr (BW == 3) {
#define X(dx) set_max_lds((const void*)k_cov_group<BW, 32, dx>); set_max_lds((const void*)k_cov_group<BW, 64, dx>);
    COV_EACH_D(X)
#undef X
  }
}

void prepare_cov_kernels() {
#define X(dx) set_max_lds((const void*)k_cov_w2<dx>);
  COV_EACH_D(X)
#undef X
  set_max_lds((const void*)k_cov_hyper);
  set_max_lds((const void*)k_cov_factor<32>);
  set_max_lds((const void*)k_cov_factor<64>);
  prepare_group_bw<0>(); prepare_group_bw<1>(); prepare_group_bw<2>();
  prepare_group_bw<3>(); prepare_group_bw<4>(); prepare_group_bw<5>(); prepare_group_bw<BWMID>(); prepare_group_bw<BWWIDE>();
}

// does the covariate block of this model fit the kernels' LDS and staging assumptions? (host check at set-up)
template <int BW>
static bool group_fits_bw(const Ctx& c) {
  const Dims& d = c.d;
  hipFuncAttributes at;
  const void* fn = (d.P <= 32) ? (const void*)k_cov_group<BW, 32> : (const void*)k_cov_group<BW, 64>;
  if (hipFuncGetAttributes(&at, fn) != hipSuccess) { (void)hipGetLastError(); return false; }
  const size_t dyn = ((size_t)c.NPG * d.LG + 2 + (d.P <= 32 ? (size_t)d.D * d.P * d.P + 2 : 0)) * sizeof(double);
  return at.sharedSizeBytes + dyn <= 160 * 1024 && d.D * (PMAX + 2 * BW + 2) <= GT;
}

bool cov_block_fits(const Ctx& c) {
  const Dims& d = c.d;
  const int PP = (d.P <= 32) ? 32 : 64;
  if ((2 * (size_t)PP * PP + PP + d.LG) * sizeof(double) > 160 * 1024) return false;                 // k_cov_factor
  if (((size_t)W2_CH * d.D + (size_t)c.NPG * 128) * sizeof(double) > 160 * 1024) return false;       // k_cov_w2
  switch (d.BW) {
    case 0: return group_fits_bw<0>(c);
    case 1: return group_fits_bw<1>(c);
    case 2: return group_fits_bw<2>(c);
    case 3: return group_fits_bw<3>(c);
    case 4: return group_fits_bw<4>(c);
    case 5: return group_fits_bw<5>(c);
    case BWMID: return group_fits_bw<BWMID>(c);
    default: return group_fits_bw<BWWIDE>(c);
  }
}

}  // namespace bfmmm
