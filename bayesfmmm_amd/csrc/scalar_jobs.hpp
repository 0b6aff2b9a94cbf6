// Scalar ("small") updates of the sweep, written as single-workgroup device jobs (256 threads)
// that ride as ONE EXTRA WORKGROUP of a wide kernel instead of being kernels of their own:
//   job_pi_alpha : pi (updatePi_PM, UpdatePi.h:84-116) and alpha_3 (updateAlpha3, UpdateAlpha3.h:36-63)
//                  -> extra workgroup of k_pair_gram (only the next Z update reads pi / alpha_3)
//   job_hyper    : delta (UpdateDelta.h:17-64), A (UpdateA.h:58-123), gamma (UpdateGamma.h:17-37),
//                  tau (UpdateTau.h:18-36; MV :47-63) in the reference's order
//                  -> extra workgroup of k_curve_chi (only the next factorisation reads them)
// so they cost neither a launch nor a cross-stream dependency, and run while the other 255 CUs do
// the per-curve / contraction work.  On a GPU a single lane runs ~3e8 dependent instructions/s, so
// the jobs are organised to spread their transcendental work (lgamma, log, gamma rejection
// loops) over lanes: every gamma variate is  scale * Gamma(shape, 1)  with a shape known up
// front, so all standard variates are drawn in one parallel phase and only O(M) scale
// recursions stay sequential.
#pragma once
#include "model.hpp"
#include "rng.hpp"

namespace bfmmm {

// deterministic tree reduction over blockDim.x == 256 values held in LDS scratch
__device__ inline double block_sum256(double v, double* scratch) {
  __syncthreads();
  scratch[threadIdx.x] = v;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) scratch[threadIdx.x] += scratch[threadIdx.x + o];
    __syncthreads();
  }
  const double r = scratch[0];
  __syncthreads();
  return r;
}

__device__ inline double logGamma_ref(double x) { return log(tgamma(x)); }   // Distributions.h:13-15

// ------------------------------------------------------------------------------------------------
// calcLikelihood (CalculateLikelihood.h:19-44; MV :150-160) of the iteration whose residual sums are pending: fixed-order
// sum of the per-block partials of k_curve_chi (or the sweep's quadratic-form RSS), then the closed form in sigma^2.
// 256 threads; `red` is 256 doubles of LDS scratch.
__device__ inline void deferred_loglik(const Ctx& c, double* red) {
  Dyn* dyn = c.dyn;
  const int tid = threadIdx.x;
  double rss = dyn->rss;
  if (dyn->ll_use_part) {
    double acc = 0.0;
    const int per = (c.nblk_curve + 255) / 256;
    for (int b = tid * per; b < min(c.nblk_curve, (tid + 1) * per); ++b) acc += c.rss_part[b];
    rss = block_sum256(acc, red);
  }
  if (tid == 0) {
    const double s2 = dyn->sigma2;
    double ll;
    if (c.d.mv)   // calcLikelihoodMV: (y_obs.n_cols / 2) is an integer division, CalculateLikelihood.h:155
      ll = -(double)c.d.n * ((c.d.P / 2) * log(2 * 3.14159265358979323846 * s2)) - (1 / (s2 * 2)) * rss;
    else
      ll = -(double)c.d.n_obs_total * (0.91893853320467274178 + log(sqrt(s2))) - rss / (2.0 * s2);
    dyn->rss = rss;
    dyn->loglik = ll;
    if (c.mask & U_LOGLIK) c.c_loglik[dyn->ll_slot] = ll;
    dyn->ll_pending = 0u;
  }
  __syncthreads();
}

constexpr int PI_ALPHA_LDS_DOUBLES = KMAX * 256 + 10 * KMAX + 16;
constexpr int PI_TAB_DOUBLES = 9 * KMAX + 16;      // g | lg | ph_s | lu | lp | dt  (contiguous, see pi_alpha_layout)

struct PiAlphaTabs { double *g, *lg, *ph_s, *lu, *lp, *dt; };
__device__ inline PiAlphaTabs pi_alpha_layout(double* base) {
  PiAlphaTabs t;
  t.g = base; t.lg = t.g + KMAX; t.ph_s = t.lg + 6 * KMAX + 8; t.lu = t.ph_s + 2; t.lp = t.lu + 2; t.dt = t.lp + 2 * KMAX;
  return t;
}

// Data-independent part of the pi / alpha_3 updates (updatePi_PM UpdatePi.h:84-116, updateAlpha3 UpdateAlpha3.h:36-63):
// the proposals, the acceptance uniforms and every lgamma / log / truncated-normal term of the two ratios.  Only the
// sums S_k = sum_i log Z_ik depend on the data.  256 threads; tables in LDS (t); ends with a barrier.
//   lgamma table rows (x K): 0 a_pi*pi_old, 1 a_pi*pi_new, 2 a3*pi_old, 3 a3*pi_new, 4 ph*pi_old, 5 ph*pi_new; then the
//   6 lgamma(sum) terms at 6K..6K+5.
__device__ inline void pi_alpha_tables(const Ctx& c, const RngKey& key, const PiAlphaTabs& t) {
  const int K = c.d.K, tid = threadIdx.x;
  const Dyn* dyn = c.dyn;
  const double alpha3 = dyn->alpha3;
  const double sd = c.h.var_alpha3;
  // one wave each so that the different code paths run side by side: wave 0 lanes k < K: the pi gammas; wave 1: the
  // alpha_3 truncated normal; wave 2: log(u) of the two MH tests
  if (tid < K) {
    const double a_old = c.h.a_pi_PM * dyn->pi[tid];
    t.g[tid] = rgamma(key, UPD_PI_PROP, (uint32_t)tid, (a_old <= 0) ? 10.0 : a_old, 1.0);
  } else if (tid == 64) {
    t.ph_s[0] = rtruncnorm_lo(key, UPD_A3_PROP, 0, alpha3, sd, 0.0);
  } else if (tid == 128 || tid == 129) {
    t.lu[tid - 128] = log(runif(key, (tid == 128) ? UPD_PI_ACC : UPD_A3_ACC, 0));
  }
  __syncthreads();
  // (fully unrolled over KMAX with k < K predicates: runtime-indexed local arrays would live in scratch memory, and a
  //  kernel that needs scratch pays for its set-up at every wave launch)
  double pi_old[KMAX], pi_new[KMAX];
  double gsum = 0.0;
#pragma unroll
  for (int k = 0; k < KMAX; ++k) if (k < K) gsum += t.g[k];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) { pi_old[k] = (k < K) ? dyn->pi[k] : 0.0; pi_new[k] = (k < K) ? t.g[k] / gsum : 0.0; }
  const double a3_ph = t.ph_s[0];
  if (tid < 6 * K + 6) {
    const bool is_sum = tid >= 6 * K;
    const int row = is_sum ? tid - 6 * K : tid / K, k = is_sum ? 0 : tid - row * K;
    const double sc = (row < 2) ? c.h.a_pi_PM : ((row < 4) ? alpha3 : a3_ph);
    double arg = 0.0;
#pragma unroll
    for (int k2 = 0; k2 < KMAX; ++k2)
      if (k2 < K) {
        const double term = sc * ((row & 1) ? pi_new[k2] : pi_old[k2]);
        if (is_sum) arg += term;
        else if (k2 == k) arg = term;
      }
    t.lg[tid] = lgamma_pos(arg);
  } else if (tid >= 64 && tid < 64 + 2 * K) {
    const int e = tid - 64;
    double arg = 1.0;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) { if (k < K && e == k) arg = pi_old[k]; if (k < K && e == K + k) arg = pi_new[k]; }
    t.lp[e] = log(arg);
  } else if (tid == 128 || tid == 129) {
    // d_truncnorm(x, x, sd, 0, Inf, log) evaluated at the *other* state (UpdateAlpha3.h:23-24)
    t.dt[tid - 128] = (tid == 128) ? dtruncnorm_lo_log(a3_ph, a3_ph, sd, 0.0) : dtruncnorm_lo_log(alpha3, alpha3, sd, 0.0);
  }
  __syncthreads();
}

// Spare workgroup of k_factor (iteration t): the tables of iteration t + 1's pi / alpha_3 job (pi, alpha_3 are final for
// iteration t once k_pair_gram has run).  Tagged like the Z proposals (Dyn::piprep_*).
__device__ inline void job_pi_prepare(const Ctx& c) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  Dyn* dyn = c.dyn;
  const PiAlphaTabs t = pi_alpha_layout(smem);
  pi_alpha_tables(c, make_key(c.seed, c.chain, dyn->iter + 1u, 0u), t);
  for (int e = threadIdx.x; e < PI_TAB_DOUBLES; e += 256) c.piprep[e] = smem[e];
  if (threadIdx.x == 0) { dyn->piprep_iter = dyn->iter + 1u; dyn->zprep_chain = c.chain; dyn->zprep_seed = c.seed; dyn->piprep_valid = 1u; }
}

__device__ inline void job_pi_alpha(const Ctx& c) {
  // scratch carved from k_pair_gram's dynamic LDS (the launcher guarantees PI_ALPHA_LDS_DOUBLES)
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double (*red)[256] = (double (*)[256])smem;       // KMAX x 256
  double* S = smem + KMAX * 256;                     // KMAX
  const PiAlphaTabs t = pi_alpha_layout(S + KMAX);
  double *g = t.g, *lg = t.lg, *ph_s = t.ph_s, *lu = t.lu, *lp = t.lp, *dt = t.dt;
  const Dims& d = c.d;
  const int K = d.K, n = d.n, tid = threadIdx.x;
  Dyn* dyn = c.dyn;
  const uint32_t mask = c.mask;
  if (dyn->ll_pending) deferred_loglik(c, smem);
  if (!(mask & (U_PI | U_ALPHA3))) {
    if (tid == 0) {
      c.c_alpha3[dyn->slot] = dyn->alpha3;
      for (int k = 0; k < K; ++k) c.c_pi[(size_t)dyn->slot * K + k] = dyn->pi[k];
    }
    return;
  }
  // the data-independent tables: prepared during the previous iteration's k_factor, or evaluated here
  const bool pre = dyn->piprep_valid && dyn->piprep_iter == dyn->iter && dyn->tt_step == 0 &&
                   dyn->zprep_chain == c.chain && dyn->zprep_seed == c.seed;
  if (pre) {
    for (int e = tid; e < PI_TAB_DOUBLES; e += 256) t.g[e] = c.piprep[e];
  } else {
    pi_alpha_tables(c, make_key(c.seed, c.chain, dyn->iter, dyn->tt_step), t);
  }
  // S_k = sum_i log Z_ik from the block partials of k_curve_z: one fixed-order tree for all k at once
  {
    double acc[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) acc[k] = 0.0;
    const int per = (c.nblk_curve + 255) / 256;
    for (int b = tid * per; b < min(c.nblk_curve, (tid + 1) * per); ++b)
#pragma unroll
      for (int k = 0; k < KMAX; ++k)
        if (k < K) acc[k] += c.logz_part[(size_t)b * K + k];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) red[k][tid] = acc[k];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (tid < o)
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
          if (k < K) red[k][tid] += red[k][tid + o];
      __syncthreads();
    }
    if (tid < K) S[tid] = red[tid][0];
  }
  __syncthreads();
  double alpha3 = dyn->alpha3;
  double pi_old[KMAX], pi_new[KMAX];
  double gsum = 0.0;
#pragma unroll
  for (int k = 0; k < KMAX; ++k) if (k < K) gsum += g[k];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) { pi_old[k] = (k < K) ? dyn->pi[k] : 0.0; pi_new[k] = (k < K) ? g[k] / gsum : 0.0; }
  const double a3_ph = ph_s[0];
  if (tid == 0) {
    auto lB = [&](int row) {   // calc_lB of (scale_row * pi_row), Distributions.h:51-60
      double s = 0.0;
      for (int k = 0; k < K; ++k) s += lg[row * K + k];
      return s - lg[6 * K + row];
    };
    double pi[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) pi[k] = pi_old[k];
    int pi_is_new = 0;
    if (mask & U_PI) {
      double lpdf_new = 0.0, lpdf_old = 0.0, pn = 0.0, po = 0.0;
#pragma unroll
      for (int k = 0; k < KMAX; ++k)
        if (k < K) {
          const double lo = lp[k], ln = lp[K + k];
          lpdf_new += (c.h.c[k] - 1) * ln + ((alpha3 * pi_new[k]) - 1) * S[k];
          lpdf_old += (c.h.c[k] - 1) * lo + ((alpha3 * pi_old[k]) - 1) * S[k];
          pn += (c.h.a_pi_PM * pi_old[k] - 1) * ln;
          po += (c.h.a_pi_PM * pi_new[k] - 1) * lo;
        }
      lpdf_new -= n * lB(3);
      lpdf_old -= n * lB(2);
      const double lpn = pn - lB(0);
      const double lpo = po - lB(1);
      const double acc = lpdf_new - lpdf_old + lpo - lpn;
      if (lu[0] < acc) {
        pi_is_new = 1;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) pi[k] = pi_new[k];
      }
#pragma unroll
      for (int k = 0; k < KMAX; ++k) if (k < K) dyn->pi[k] = pi[k];
    }
    if (mask & U_ALPHA3) {
      double l_old = (-c.h.b) * alpha3, l_new = (-c.h.b) * a3_ph;
#pragma unroll
      for (int k = 0; k < KMAX; ++k)
        if (k < K) {
          l_old += ((alpha3 * pi[k]) - 1) * S[k];
          l_new += ((a3_ph * pi[k]) - 1) * S[k];
        }
      l_old -= n * lB(2 + pi_is_new);
      l_new -= n * lB(4 + pi_is_new);
      l_old += dt[0];
      l_new += dt[1];
      if (lu[1] < l_new - l_old) alpha3 = a3_ph;
      dyn->alpha3 = alpha3;
    }
    c.c_alpha3[dyn->slot] = dyn->alpha3;
    for (int k = 0; k < K; ++k) c.c_pi[(size_t)dyn->slot * K + k] = dyn->pi[k];
  }
}

// ------------------------------------------------------------------------------------------------
// State-independent variates of job_hyper, drawn by spare workgroups of k_factor (same iteration):
//   gstd = [ gamma(k,p,m): K*P*M standard gammas | delta: K*M | tau: K | A proposals: 2K | A accept uniforms: 2K ]
// (the delta shapes and the A proposals depend on A, which the previous iteration's job_hyper
// finished updating before this iteration's k_factor started).
__device__ inline int hyper_gstd_count(const Dims& d) { return d.K * d.P * d.M + d.K * d.M + d.K + 4 * d.K; }

__device__ inline void job_hyper_draws(const Ctx& c, int first) {
  const Dims& d = c.d;
  const int P = d.P, K = d.K, M = d.M;
  const Dyn* dyn = c.dyn;
  const RngKey key = make_key(c.seed, c.chain, dyn->iter, dyn->tt_step);
  const int nG = K * P * M;
  const int e = first + threadIdx.x;
  if (e > hyper_gstd_count(d) + 8 * K) return;
  double v;
  if (e > hyper_gstd_count(d)) {
    // state-independent pieces of the A update's acceptance ratio (UpdateA.h:17-44, 79-112): cell = (j, i), piece q:
    //   0 / 1: logGamma(a), log(a) at the current value; 2 / 3: the same at the proposal (drawn above with the same key)
    const int x = e - hyper_gstd_count(d) - 1, cell = x >> 2, q = x & 3;
    const int j = cell / 2, i = cell - 2 * j;
    const double sd = (i == 0) ? (c.h.var_epsilon1 / c.h.beta1l) : (c.h.var_epsilon2 / c.h.beta2l);
    const double cur = c.Aa[j + (size_t)K * i];
    const double na = rtruncnorm_lo(key, UPD_A_PROP, (uint32_t)(j * 2 + i), cur, sd, 0.0);
    const double a = (q < 2) ? cur : na;
    v = (q & 1) ? log(a) : logGamma_ref(a);
  } else if (e == hyper_gstd_count(d)) {
    // standard gamma variate of the sigma^2 draw (its shape is state-independent): UpdateSigma.h:49 / :150
    const bool tempered = (dyn->tt_step != 0);
    double shape = tempered ? (dyn->beta * (double)d.n_obs_total) / 2 : (d.mv ? (double)(d.n_obs_total / 2) : (double)d.half_sum);
    shape += c.h.alpha_0;
    v = rgamma(key, UPD_SIGMA, 0, shape, 1.0);
  } else if (e < nG) {
    v = rgamma(key, UPD_GAMMA, (uint32_t)e, (c.h.nu_1 + 1) / 2, 1.0);                       // UpdateGamma.h:29
  } else if (e < nG + K * M) {
    const int q = e - nG, k = q / M, i = q - k * M;
    const double param1 = (i == 0) ? c.Aa[k] + ((P * M) / 2.0) : c.Aa[k + (size_t)K] + ((P * (M - i)) / 2.0);
    v = rgamma(key, UPD_DELTA, (uint32_t)(k * M + i), param1, 1.0);                         // UpdateDelta.h:28,42 / :45,57
  } else if (e < nG + K * M + K) {
    const int k = e - nG - K * M;
    v = rgamma(key, UPD_TAU, (uint32_t)k, c.h.alpha_nu + (P / 2), 1.0);                     // integer division, UpdateTau.h:29
  } else if (e < nG + K * M + K + 2 * K) {
    const int cell = e - nG - K * M - K, j = cell / 2, i = cell - 2 * j;
    const double sd = (i == 0) ? (c.h.var_epsilon1 / c.h.beta1l) : (c.h.var_epsilon2 / c.h.beta2l);
    v = rtruncnorm_lo(key, UPD_A_PROP, (uint32_t)(j * 2 + i), c.Aa[j + (size_t)K * i], sd, 0.0);   // UpdateA.h:79,98
  } else {
    const int cell = e - nG - K * M - K - 2 * K;
    v = runif(key, UPD_A_ACC, (uint32_t)cell);                                              // UpdateA.h:90,109
  }
  c.gstd[e] = v;
}

constexpr int HYPER_LDS_DOUBLES = 5 * KMAX * 16 + KMAX * 2 * 6 + KMAX + 2 + 2 * KMAX * PMAX + (2 * BWMAX + 1) * PMAX;

// delta recursion of one cluster (UpdateDelta.h:28-57) on registers: the old delta_km, S_km and the standard gamma variates are
// read from LDS up front (MT of each, clamped), so the O(M^2) chain of products is a chain of arithmetic only -- with the values
// read from LDS inside the loops every one of its M (M + 1) / 2 steps paid an LDS round trip.  tilde-tau products are carried
// along instead of rebuilt: the multiplication order is the reference's (UpdateDelta.h:30-37, 47-54), the work O(M^2) not O(M^3).
template <int MT>
__device__ inline void delta_recursion(double* dk, const double* sk, const double* gd, double* tp, int M, bool do_delta) {
  double dv[MT], sv[MT], gv[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) { const int mc = min(m, M - 1); dv[m] = dk[mc]; sv[m] = sk[mc]; gv[m] = gd[mc]; }
  if (do_delta) {
    double pre = 1.0;                                  // prod_{nn < i} delta_new(k, nn)
#pragma unroll
    for (int i = 0; i < MT; ++i)
      if (i < M) {
        double tt = pre;                               // prod_{nn <= m, nn != i} delta(k, nn), m = i
        double param2 = 1.0 + 0.5 * tt * sv[i];
#pragma unroll
        for (int m = i + 1; m < MT; ++m)
          if (m < M) { tt *= dv[m]; param2 += 0.5 * tt * sv[m]; }
        dv[i] = gv[i] * (1.0 / param2);
        pre *= dv[i];
      }
  }
  double tt = 1.0;                                     // tilde-tau(k, j) = prod_{j2 <= j} delta(k, j2), UpdateGamma.h:26-28
#pragma unroll
  for (int m = 0; m < MT; ++m)
    if (m < M) { tt *= dv[m]; tp[m] = tt; dk[m] = dv[m]; }
}

// One workgroup; a chain of dependent steps, so it is laid out for latency: EVERY global operand the job needs is requested
// before the first one is waited for (one round trip to memory instead of one per phase: S_km inputs, nu, delta, the penalty band,
// the A cells, the gamma-scaling inputs, the variates drawn ahead by job_hyper_draws), the phases between run on LDS and
// registers, and the chain slots are written from the values at hand instead of being read back from the state arrays.
// end-of-iteration bookkeeping (when the iteration has no k_loglik): the curve workgroups of k_curve_chi work from the sweep's
// snapshot (iter_hyper / slot_hyper), so the counters can advance beside them; the log-likelihood of the finished iteration is
// reduced by the next kernel that has an idle workgroup (deferred_loglik)
__device__ inline void job_hyper_counters(const Ctx& c) {
  Dyn* dyn = c.dyn;
  if (c.defer_loglik && threadIdx.x == 0) {
    dyn->ll_slot = dyn->slot_hyper;
    dyn->ll_use_part = (uint32_t)c.ll_use_part;
    dyn->ll_pending = 1u;
    dyn->iter = dyn->iter_hyper + 1u;
    dyn->slot = dyn->iter_hyper + 1u - dyn->slot_base;
  }
}

__device__ inline void job_hyper(const Ctx& c, bool counters = true) {
  // scratch carved from the host kernel's dynamic LDS (the job's workgroup does not use it otherwise; the launcher
  // guarantees HYPER_LDS_DOUBLES): no static LDS, so the job does not lower the occupancy of the curve workgroups
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* Skm = smem;                 // KMAX * 16
  double* dl = Skm + KMAX * 16;       // KMAX * 16
  double* lgd = dl + KMAX * 16;       // KMAX * 16
  double* tpre = lgd + KMAX * 16;     // KMAX * 16
  double* gDs = tpre + KMAX * 16;     // KMAX * 16
  double* aw = gDs + KMAX * 16;       // KMAX * 2 * 6
  double* slog = aw + KMAX * 2 * 6;   // KMAX (+2 pad; unused now, kept for the layout)
  double* qrow = slog + KMAX + 2;     // KMAX * PMAX
  double* snu = qrow + KMAX * PMAX;   // KMAX * PMAX
  double* sPb = snu + KMAX * PMAX;    // (2 BWMAX + 1) * PMAX: the band of the penalty, [u][p] = P[p][p - BWP + u]
  const Dims& d = c.d;
  const int P = d.P, K = d.K, M = d.M, MD = d.MD, tid = threadIdx.x;
  Dyn* dyn = c.dyn;
  const uint32_t mask = c.mask;
  const bool phi_on = MD > 1;
  const bool do_delta = (mask & U_DELTA) && phi_on, do_A = (mask & U_A) && phi_on, do_gamma = (mask & U_GAMMA) && phi_on;
  const bool do_tau = (mask & U_TAU) != 0;
  const int Mc = max(M, 1);
  const int nG = K * P * M, nGc = max(nG, 1);
  const double* gGam = c.gstd;
  const double* gD = c.gstd + nG;
  const double* gT = gD + K * M;
  const double* aProp = gT + K;
  const double* aUnif = aProp + 2 * K;
  const double* aTerm = c.gstd + hyper_gstd_count(d) + 1;      // 4 per cell: logGamma / log at cur, at the proposal
  // ================= all global loads (clamped addresses, no branches: the requests are issued back to back) =================
  const uint32_t slot = dyn->slot_hyper;
  // S_km = sum_p gamma_old(k,p,m) phi(k,p,m)^2: 8 lanes per (k, m); first 32 (k, m) here, the rest (K M > 32) in the loop below
  double s_ph[8], s_gm[8];
  {
    const int km = min(tid >> 3, max(K * M - 1, 0)), q = tid & 7;
    const int k = km / Mc, m = km - k * Mc;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int p = min(q + 8 * u, P - 1);
      s_ph[u] = c.theta[(size_t)(k * (M + 1) + min(m + 1, M)) * P + p];
      s_gm[u] = c.gamma[min(k + K * (p + P * m), nGc - 1)];
    }
  }
  // nu_k for nu_k' P nu_k, the old delta and its variates (staged in LDS for the K lanes of the recursion)
  const int KP = K * P;
  const int e0 = min(tid, KP - 1), k0 = e0 / P, p0 = e0 - k0 * P;
  const double nu0 = c.theta[(size_t)(k0 * (M + 1)) * P + p0];
  const int e1 = min(tid + 256, KP - 1), k1 = e1 / P, p1 = e1 - k1 * P;
  const double nu1 = c.theta[(size_t)(k1 * (M + 1)) * P + p1];
  const int kmc = min(tid, max(K * M - 1, 0)), kd = kmc / Mc, md = kmc - kd * Mc;
  const double dl0 = c.delta[kd + (size_t)K * md];
  const double gd0 = gD[kmc];
  // the band of the penalty (the wide-band case reads whole rows below): (2 BWMAX + 1) P <= 704 entries, three per thread, staged in LDS
  constexpr int NPB = ((2 * BWMAX + 1) * PMAX + 255) / 256;
  double pb[NPB];
#pragma unroll
  for (int r = 0; r < NPB; ++r) {
    const int idx = min(tid + 256 * r, (2 * BWMAX + 1) * P - 1), u = idx / P, p = idx - u * P;
    pb[r] = c.Pmat[p + (size_t)P * min(max(p - d.BWP + u, 0), P - 1)];
  }
  // A: the term threads (4 per cell) and the accept threads (1 per cell)
  const int cellT = min(tid >> 2, 2 * K - 1), jobT = tid & 3, jT = cellT / 2, iT = cellT - 2 * jT;
  const double curT = c.Aa[jT + (size_t)K * iT], naT = aProp[cellT];
  const double lgaT = aTerm[cellT * 4 + 2 * (jobT & 1)], laT = aTerm[cellT * 4 + 2 * (jobT & 1) + 1];      // drawn ahead (job_hyper_draws)
  const int cellA = min(tid, 2 * K - 1), idxA = (cellA >> 1) + K * (cellA & 1);
  const double curA = c.Aa[idxA], naA = aProp[cellA], unA = aUnif[cellA];
  const int kt = min(max(tid - 32, 0), K - 1);
  const double tau_old = dyn->tau[kt], gTk = gT[kt];
  // gamma scaling: e = (i*P + l)*M + j (reference loop order i, l, j); the first three trips here
  constexpr int GU = 3;
  double g_std[GU], g_phi[GU];
#pragma unroll
  for (int u = 0; u < GU; ++u) {
    const int e = min(tid + 256 * u, nGc - 1);
    const int jj = e % Mc, il = e / Mc, l = il % P, i = il / P;
    g_std[u] = gGam[e];
    g_phi[u] = c.theta[(size_t)(i * (M + 1) + min(jj + 1, M)) * P + l];
  }
  // ================= phase 1: S_km, staging =================
  if (do_delta) {
    {
      const int km = tid >> 3, q = tid & 7;
      const bool on = km < K * M;
      double acc = 0.0;
      double pr[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) pr[u] = (on && q + 8 * u < P) ? s_gm[u] * (s_ph[u] * s_ph[u]) : 0.0;
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += pr[u];
      acc = dpp_add<0xB1>(acc);
      acc = dpp_add<0x4E>(acc);
      acc = dpp_add<0x141>(acc);
      if (on && q == 0) Skm[km] = acc;
    }
    for (int g0 = 32; g0 < K * M; g0 += 32) {
      const int km = g0 + (tid >> 3), q = tid & 7;
      const bool on = km < K * M;
      const int k = on ? km / M : 0, m = on ? km - k * M : 0;
      double pr[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int p = min(q + 8 * u, P - 1);
        const double ph = c.theta[(size_t)(k * (M + 1) + m + 1) * P + p];
        const double gm = c.gamma[k + (size_t)K * (p + (size_t)P * m)];
        pr[u] = (on && q + 8 * u < P) ? gm * (ph * ph) : 0.0;
      }
      double acc = 0.0;
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += pr[u];
      acc = dpp_add<0xB1>(acc);
      acc = dpp_add<0x4E>(acc);
      acc = dpp_add<0x141>(acc);
      if (on && q == 0) Skm[km] = acc;
    }
  }
  if (do_tau) {
    if (tid < KP) snu[tid] = nu0;
    if (tid + 256 < KP) snu[tid + 256] = nu1;
  }
  if (tid < K * M && tid < KMAX * 16) { dl[kd * 16 + md] = dl0; gDs[kd * 16 + md] = gd0; }
#pragma unroll
  for (int r = 0; r < NPB; ++r)
    if (tid + 256 * r < (2 * BWMAX + 1) * P) sPb[tid + 256 * r] = pb[r];
  __syncthreads();
  // nu_k' P nu_k : lane (k, p) takes row p of P, summed per k below
  if (do_tau)
    for (int e = tid; e < KP; e += 256) {
      const int k = e / P, p = e - k * P;
      double s = 0.0;
      if (d.mv) s = snu[e];
      else if (d.BWP > BWMAX) {       // wide penalty band (tensor-product bases): the plain row product of UpdateTau.h:26-28
        for (int q = 0; q < P; ++q) s += c.Pmat[p + (size_t)P * q] * snu[k * P + q];
      } else {
        // the penalty is banded (half-width BWP, the same assumption the factorisation makes): the other products of
        // row p are exact zeros, so skipping them leaves the sum of UpdateTau.h:26-28 unchanged
#pragma unroll
        for (int u = 0; u < 2 * BWMAX + 1; ++u) {
          const int q = p - d.BWP + u;
          if (u <= 2 * d.BWP && q >= 0 && q < P) s += sPb[u * P + p] * snu[k * P + q];
        }
      }
      qrow[e] = snu[e] * s;
    }
  // ---- phase 2: delta recursion (K lanes) ----
  if (tid < K && M > 0) {
    if (M <= 8) delta_recursion<8>(dl + tid * 16, Skm + tid * M, gDs + tid * 16, tpre + tid * 16, M, do_delta);
    else {      // M > 8: the same recursion on LDS operands (48 more live doubles would spill in the kernels that host this job)
      double* dk = dl + tid * 16;
      const double* sk = Skm + tid * M;
      if (do_delta) {
        double pre = 1.0;
        for (int i = 0; i < M; ++i) {
          double tt = pre;
          double param2 = 1.0 + 0.5 * tt * sk[i];
          for (int m = i + 1; m < M; ++m) { tt *= dk[m]; param2 += 0.5 * tt * sk[m]; }
          dk[i] = gDs[tid * 16 + i] * (1.0 / param2);
          pre *= dk[i];
        }
      }
      double tt = 1.0;
      for (int m = 0; m < M; ++m) { tt *= dk[m]; tpre[tid * 16 + m] = tt; }
    }
  }
  __syncthreads();
  if (tid >= 32 && tid < 32 + K) {
    const int k = tid - 32;
    double tv = tau_old;
    if (do_tau) {
      double qf = 0.0;
      for (int p = 0; p < P; ++p) qf += qrow[k * P + p];
      const double b = c.h.beta_nu + (0.5 * qf);
      const double gg = gTk * (1.0 / b);
      tv = d.mv ? (1.0 / gg) : gg;
      dyn->tau[k] = tv;
    }
    c.c_tau[slot + (size_t)c.T * k] = tv;
  }
  if (tid < K * M) {
    const double dv = dl[kd * 16 + md];
    if (do_delta) c.delta[kd + (size_t)K * md] = dv;
    c.c_delta[(size_t)slot * K * M + kd + (size_t)K * md] = dv;
    lgd[kd * 16 + md] = log(dv);
  }
  // ---- gamma scaling (needs the tilde-tau products of phase 2) ----
  double* s_gam = c.c_gamma + (size_t)slot * K * P * M;
  if (do_gamma) {
#pragma unroll
    for (int u = 0; u < GU; ++u) {
      const int e = tid + 256 * u;
      if (e < nG) {
        const int jj = e % M, il = e / M, l = il % P, i = il / P;
        const double ph = tpre[i * 16 + jj];
        const double phi = g_phi[u];
        const double gnew = g_std[u] * (2 / (c.h.nu_1 + ph * (phi * phi)));
        const size_t at = i + (size_t)K * (l + (size_t)P * jj);
        c.gamma[at] = gnew;
        s_gam[at] = gnew;
      }
    }
    for (int e = tid + 256 * GU; e < nG; e += 256) {
      const int jj = e % M, il = e / M, l = il % P, i = il / P;
      const double ph = tpre[i * 16 + jj];
      const double phi = c.theta[(size_t)(i * (M + 1) + jj + 1) * P + l];
      const double gnew = gGam[e] * (2 / (c.h.nu_1 + ph * (phi * phi)));
      const size_t at = i + (size_t)K * (l + (size_t)P * jj);
      c.gamma[at] = gnew;
      s_gam[at] = gnew;
    }
  } else {
    for (int e = tid; e < nG; e += 256) s_gam[e] = c.gamma[e];
  }
  __syncthreads();
  // ---- phase 3: A terms (4 lanes per cell) ----
  if (do_A && tid < K * 2 * 4) {
    const int cell = tid >> 2, job = tid & 3;
    const int j = jT;
    const bool first = (iT == 0);
    const double sd = first ? (c.h.var_epsilon1 / c.h.beta1l) : (c.h.var_epsilon2 / c.h.beta2l);
    const double cur = curT, na = naT;
    double v;
    if (job < 2) {
      const double a = (job == 0) ? cur : na;
      const double lga = lgaT, la = laT;
      if (first) v = -lga + (a - 1) * lgd[j * 16] + (c.h.alpha1l - 1) * la - (a * c.h.beta1l);          // UpdateA.h:17-24
      else {
        double sl = 0.0;                                 // sum_{q >= 1} log delta(k, q), used by lpdf_a2
        for (int q = 1; q < M; ++q) sl += lgd[j * 16 + q];
        const double x = M - 1;
        v = -x * lga + (c.h.alpha2l - 1) * la - (a * c.h.beta2l) + (a - 1) * sl;                         // UpdateA.h:33-44
      }
    } else {
      v = (job == 2) ? dtruncnorm_lo_log(cur, na, sd, 0.0) : dtruncnorm_lo_log(na, cur, sd, 0.0);
    }
    aw[cell * 6 + 1 + job] = v;
  }
  __syncthreads();
  // ---- phase 4: A accept; chain slots ----
  if (tid < K * 2) {
    double av = curA;
    if (do_A) {
      const double* w = aw + tid * 6;
      const double acc = (w[2] + w[3]) - w[1] - w[4];
      if (log(unA) < acc) { av = naA; c.Aa[idxA] = naA; }
    }
    c.c_A[(size_t)slot * K * 2 + idxA] = av;
  }
  if (counters) job_hyper_counters(c);
}

}  // namespace bfmmm
