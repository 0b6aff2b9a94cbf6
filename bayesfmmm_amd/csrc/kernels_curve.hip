// Per-curve kernels: Z (mixed-membership Metropolis-Hastings) and chi (scores), plus the
// per-curve residual sums that feed sigma^2 and the log-likelihood.
//
//   k_curve_z   <- updateZ_PM / lpdf_z / Z_proposal_density / rdirichlet
//                  (UpdateMixedMembership.h:20-50, 102-113, 131-185; Distributions.h:22-60)
//   k_curve_chi <- updateChi (UpdateChi.h:19-64) and the residual pass of calcLikelihood
//                  (CalculateLikelihood.h:19-44)
//
// Mapping: curves are independent, so a group of LPC lanes (32 or 64, one lane per basis
// function) owns one curve; a 256-thread workgroup holds 256/LPC curves.  The banded Gram matrix
// G_i sits in registers (2*BW+1 values per lane); matrix-vector products exchange neighbours with
// wave shuffles, quadratic forms are butterfly reductions inside the group, so nothing but the
// shared parameter block theta goes through LDS.  All reductions have a fixed order: results do
// not depend on the launch geometry.
#include "model.hpp"
#include "rng.hpp"

namespace bfmmm {

template <int LPC>
__device__ inline double gsum(double v) {
#pragma unroll
  for (int o = LPC / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, LPC);
  return v;
}

template <int BW, int LPC>
struct Curve {
  double g[BW + 1];    // g[d]  = G[p][p+d]
  double gl[BW + 1];   // gl[d] = G[p-d][p]
  double s, yy;
  int p;
  __device__ inline void load(const double* __restrict__ rec, int P, int LG, int lane_in_group) {
    p = lane_in_group;
    const bool act = p < P;
#pragma unroll
    for (int d = 0; d <= BW; ++d) {
      g[d] = (act && p + d < P) ? rec[d * P + p] : 0.0;
      gl[d] = (act && d > 0 && p - d >= 0) ? rec[d * P + p - d] : 0.0;
    }
    s = act ? rec[LG + p] : 0.0;
    yy = rec[LG + P];
  }
  // (G u)[p]; u must be 0 on lanes p >= P
  __device__ inline double matvec(double u) const {
    double v = g[0] * u;
#pragma unroll
    for (int d = 1; d <= BW; ++d) {
      v += g[d] * __shfl(u, p + d, LPC);
      v += gl[d] * __shfl(u, p - d, LPC);
    }
    return v;
  }
};

// ------------------------------------------------------------------------------------------------
// Z update.  do_update == 0 only recomputes the partial sums of log Z (used when pi / alpha_3 are
// sampled with Z held fixed).
// ------------------------------------------------------------------------------------------------
template <int BW, int LPC>
__global__ __launch_bounds__(256) void k_curve_z(Ctx c, int do_update) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int GPB = 256 / LPC;
  const Dims& d = c.d;
  const int n = d.n, K = d.K, P = d.P, M = d.M, MD = d.MD;
  double* sTh = smem;                       // K*(M+1)*P
  double* sLog = smem + (size_t)K * (M + 1) * P;   // GPB*KMAX
  for (int q = threadIdx.x; q < K * (M + 1) * P; q += 256) sTh[q] = c.theta[q];
  __syncthreads();
  const int grp = threadIdx.x / LPC, lp = threadIdx.x % LPC;
  const int i = blockIdx.x * GPB + grp;
  const bool valid = i < n;
  const bool act = lp < P;
  const Dyn* dyn = c.dyn;
  const double sigma2 = dyn->sigma2, alpha3 = dyn->alpha3, beta = dyn->beta;
  double logz_mine = 0.0;
  if (valid) {
    Curve<BW, LPC> cv;
    cv.load(c.rec + (size_t)i * d.LREC, P, d.LG, lp);
    double Zold[KMAX], u[KMAX], Gu[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) Zold[k] = (k < K) ? c.Z[i + (size_t)n * k] : 0.0;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      double v = 0.0;
      if (k < K && act) {
        const double* th = sTh + (size_t)k * (M + 1) * P;
        v = th[lp];
        if (MD > 1)
          for (int m = 0; m < M; ++m) v += c.chi[i + (size_t)n * m] * th[(m + 1) * P + lp];
      }
      u[k] = v;
    }
#pragma unroll
    for (int k = 0; k < KMAX; ++k) Gu[k] = (k < K) ? cv.matvec(u[k]) : 0.0;
    double av[KMAX], Q[KMAX][KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      av[k] = (k < K) ? gsum<LPC>(u[k] * cv.s) : 0.0;
#pragma unroll
      for (int k2 = 0; k2 < KMAX; ++k2) {
        if (k2 >= k) Q[k][k2] = (k2 < K) ? gsum<LPC>(u[k] * Gu[k2]) : 0.0;
      }
    }
    double Zfin[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) Zfin[k] = Zold[k];
    if (do_update) {
      const RngKey key = make_key(c.seed, c.chain, dyn->iter, dyn->tt_step);
      // proposal: Dirichlet(a_Z_PM * Z_old) through K gamma draws, lane k draws component k
      double a_old[KMAX], a_new[KMAX], Znew[KMAX];
      double mygam = 0.0;
#pragma unroll
      for (int k = 0; k < KMAX; ++k) {
        a_old[k] = c.h.a_Z_PM * Zold[k];
        if (k < K && lp == k) {
          const double a = (a_old[k] <= 0) ? 10.0 : a_old[k];            // Distributions.h:24-28
          mygam = rgamma(key, UPD_Z_PROP, (uint32_t)(i * K + k), a, 1.0);
        }
      }
      double gsum_ = 0.0;
#pragma unroll
      for (int k = 0; k < KMAX; ++k) {
        Znew[k] = (k < K) ? __shfl(mygam, k, LPC) : 0.0;
        if (k < K) gsum_ += Znew[k];
      }
#pragma unroll
      for (int k = 0; k < KMAX; ++k) { Znew[k] = Znew[k] / gsum_; a_new[k] = c.h.a_Z_PM * Znew[k]; }
      // quadratic form of the residual sum of squares in Z
      double q_old = cv.yy, q_new = cv.yy, pr_old = 0.0, pr_new = 0.0, dn = 0.0, dold = 0.0;
#pragma unroll
      for (int k = 0; k < KMAX; ++k) {
        if (k < K) {
          q_old -= 2.0 * Zold[k] * av[k];
          q_new -= 2.0 * Znew[k] * av[k];
#pragma unroll
          for (int k2 = 0; k2 < KMAX; ++k2) {
            if (k2 < K) {
              const double qq = (k2 >= k) ? Q[k][k2] : Q[k2][k];
              q_old += Zold[k] * Zold[k2] * qq;
              q_new += Znew[k] * Znew[k2] * qq;
            }
          }
          const double lo = log(Zold[k]), ln = log(Znew[k]);
          pr_old += (alpha3 * dyn->pi[k] - 1.0) * lo;
          pr_new += (alpha3 * dyn->pi[k] - 1.0) * ln;
          dn += (a_old[k] - 1.0) * ln;       // density of proposing new from old
          dold += (a_new[k] - 1.0) * lo;     // density of proposing old from new
        }
      }
      const double z_lpdf = pr_old - beta * (q_old / (2.0 * sigma2));
      const double z_new_lpdf = pr_new - beta * (q_new / (2.0 * sigma2));
      const double lpdf_propose_new = dn - calc_lB(K, a_old);
      const double lpdf_propose_old = dold - calc_lB(K, a_new);
      double acceptance = z_new_lpdf - z_lpdf + lpdf_propose_old - lpdf_propose_new;
      const double uu = runif(key, UPD_Z_ACC, (uint32_t)i);
#pragma unroll
      for (int k = 0; k < KMAX; ++k)
        if (k < K && Zold[k] <= 0) acceptance = 1;                        // UpdateMixedMembership.h:170-174
      if (log(uu) < acceptance) {
#pragma unroll
        for (int k = 0; k < KMAX; ++k) Zfin[k] = Znew[k];
      }
      double* zslot = c.c_Z + (size_t)dyn->slot * n * K;
#pragma unroll
      for (int k = 0; k < KMAX; ++k)
        if (k < K && lp == k) { c.Z[i + (size_t)n * k] = Zfin[k]; zslot[i + (size_t)n * k] = Zfin[k]; }
    }
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
      if (k < K && lp == k) logz_mine = log(Zfin[k]);
  }
  // block partial of sum_i log Z_ik, fixed order over the groups of this block
  if (lp < KMAX) sLog[grp * KMAX + lp] = (lp < K) ? logz_mine : 0.0;
  __syncthreads();
  if (threadIdx.x < K) {
    double acc = 0.0;
    for (int g = 0; g < GPB; ++g) acc += sLog[g * KMAX + threadIdx.x];
    c.logz_part[(size_t)blockIdx.x * K + threadIdx.x] = acc;
  }
}

// ------------------------------------------------------------------------------------------------
// chi update + per-curve residual sum of squares.  do_update == 0 computes only the residuals.
// ------------------------------------------------------------------------------------------------
template <int BW, int LPC>
__global__ __launch_bounds__(256) void k_curve_chi(Ctx c, int do_update) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int GPB = 256 / LPC;
  const Dims& d = c.d;
  const int n = d.n, K = d.K, P = d.P, M = d.M, MD = d.MD;
  double* sTh = smem;
  double* sRss = smem + (size_t)K * (M + 1) * P;
  for (int q = threadIdx.x; q < K * (M + 1) * P; q += 256) sTh[q] = c.theta[q];
  __syncthreads();
  const int grp = threadIdx.x / LPC, lp = threadIdx.x % LPC;
  const int i = blockIdx.x * GPB + grp;
  const bool valid = i < n;
  const bool act = lp < P;
  const Dyn* dyn = c.dyn;
  const double sigma2 = dyn->sigma2, beta = dyn->beta;
  double rss = 0.0;
  if (valid) {
    Curve<BW, LPC> cv;
    cv.load(c.rec + (size_t)i * d.LREC, P, d.LG, lp);
    double Zi[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) Zi[k] = (k < K) ? c.Z[i + (size_t)n * k] : 0.0;
    // fitted coefficient c_i and g = G c_i
    double cf = 0.0;
    if (act) {
#pragma unroll
      for (int k = 0; k < KMAX; ++k)
        if (k < K) {
          const double* th = sTh + (size_t)k * (M + 1) * P;
          double v = th[lp];
          if (MD > 1)
            for (int m = 0; m < M; ++m) v += c.chi[i + (size_t)n * m] * th[(m + 1) * P + lp];
          cf += Zi[k] * v;
        }
    }
    double gv = cv.matvec(cf);
    if (do_update && MD > 1) {
      const RngKey key = make_key(c.seed, c.chain, dyn->iter, dyn->tt_step);
      double* cslot = c.c_chi + (size_t)dyn->slot * n * M;
      for (int m = 0; m < M; ++m) {
        double um = 0.0;
        if (act) {
#pragma unroll
          for (int k = 0; k < KMAX; ++k)
            if (k < K) um += Zi[k] * sTh[((size_t)k * (M + 1) + m + 1) * P + lp];
        }
        const double Gum = cv.matvec(um);
        const double W0 = gsum<LPC>(um * Gum);
        const double r1 = gsum<LPC>(um * (cv.s - gv));
        const double chi_old = c.chi[i + (size_t)n * m];
        const double w = ((r1 + chi_old * W0) * beta) / sigma2;
        const double W = 1.0 / (1.0 + ((W0 * beta) / sigma2));
        const double chi_new = W * w + sqrt(W) * rnorm(key, UPD_CHI, (uint32_t)(i * M + m));
        const double dl = chi_new - chi_old;
        cf += dl * um;
        gv += dl * Gum;
        if (lp == 0) { c.chi[i + (size_t)n * m] = chi_new; cslot[i + (size_t)n * m] = chi_new; }
      }
    }
    rss = cv.yy - 2.0 * gsum<LPC>(cf * cv.s) + gsum<LPC>(cf * gv);
  }
  if (lp == 0) sRss[grp] = rss;
  __syncthreads();
  if (threadIdx.x == 0) {
    double acc = 0.0;
    for (int g = 0; g < GPB; ++g) acc += sRss[g];
    c.rss_part[blockIdx.x] = acc;
  }
}

// ---- host launchers -------------------------------------------------------------------------
template <int BW>
static void launch_curve_bw(const Ctx& c, int which, int do_update, hipStream_t st) {
  const int LPC = (c.d.P <= 32) ? 32 : 64;
  const int GPB = 256 / LPC;
  const int nblk = (c.d.n + GPB - 1) / GPB;
  const size_t lds = ((size_t)c.d.K * (c.d.M + 1) * c.d.P + GPB * KMAX + 8) * sizeof(double);
  if (LPC == 32) {
    if (which == 0) hipLaunchKernelGGL((k_curve_z<BW, 32>), dim3(nblk), dim3(256), lds, st, c, do_update);
    else hipLaunchKernelGGL((k_curve_chi<BW, 32>), dim3(nblk), dim3(256), lds, st, c, do_update);
  } else {
    if (which == 0) hipLaunchKernelGGL((k_curve_z<BW, 64>), dim3(nblk), dim3(256), lds, st, c, do_update);
    else hipLaunchKernelGGL((k_curve_chi<BW, 64>), dim3(nblk), dim3(256), lds, st, c, do_update);
  }
}

int launch_curve(const Ctx& c, int which, int do_update, hipStream_t st) {
  switch (c.d.BW) {
    case 0: launch_curve_bw<0>(c, which, do_update, st); break;
    case 1: launch_curve_bw<1>(c, which, do_update, st); break;
    case 2: launch_curve_bw<2>(c, which, do_update, st); break;
    case 3: launch_curve_bw<3>(c, which, do_update, st); break;
    case 4: launch_curve_bw<4>(c, which, do_update, st); break;
    case 5: launch_curve_bw<5>(c, which, do_update, st); break;
    default: return 1;
  }
  return 0;
}

int curve_blocks(int n, int P) {
  const int LPC = (P <= 32) ? 32 : 64;
  const int GPB = 256 / LPC;
  return (n + GPB - 1) / GPB;
}

}  // namespace bfmmm
