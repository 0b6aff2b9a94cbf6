/*
 * oracle/updates.c -- TEST INFRASTRUCTURE (see oracle.h).
 *
 * Reference-structure restatement of the full-conditional updates of ndmarco/BayesFMMM:
 * same loop nests (i -> l -> k -> n), same skip-if-Z==0 rules, same integer divisions,
 * same "copy slice iter to iter+1" carry as inst/include/BayesFMMM/Update*.h.
 * When d->D > 0 the extra terms of the *CovariateAdj variants are added at the places the
 * reference adds them (cited per function).  beta_i is the tempering temperature of the
 * *Tempered variants (1.0 = untempered; the arithmetic is then identical).
 * When d->mv != 0 the multivariate (BMVMMM) variants are followed: every "curve" has
 * exactly P observations and B_i = I (rows of d->B hold the identity).
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define NI(i)        ((int)(d->off[(i) + 1] - d->off[(i)]))
#define YOBS(i, l)   (d->y[d->off[(i)] + (l)])
#define BROW(i, l)   (d->B + (size_t)(d->off[(i)] + (l)) * P)
#define XCOV(i, dd)  (d->X[(i) + (size_t)n * (dd)])

#define SL_NU(c, t)     ((c)->nu + (size_t)K * P * (t))
#define SL_CHI(c, t)    ((c)->chi + (size_t)n * M * (t))
#define SL_Z(c, t)      ((c)->Z + (size_t)n * K * (t))
#define SL_PHI(c, t)    ((c)->Phi + (size_t)K * P * M * (t))
#define SL_GAMMA(c, t)  ((c)->gamma + (size_t)K * P * M * (t))
#define SL_DELTA(c, t)  ((c)->delta + (size_t)K * M * (t))
#define SL_A(c, t)      ((c)->A + (size_t)K * 2 * (t))
#define SL_PI(c, t)     ((c)->pi + (size_t)K * (t))
#define SL_ETA(c, t)    ((c)->eta + (size_t)P * D * K * (t))
#define SL_XI(c, t, k)  ((c)->xi + ((size_t)(t) * K + (k)) * P * D * M)
#define SL_GXI(c, t, k) ((c)->gamma_xi + ((size_t)(t) * K + (k)) * P * D * M)
#define SL_DXI(c, t)    ((c)->delta_xi + (size_t)K * M * D * (t))
#define SL_AXI(c, t)    ((c)->A_xi + (size_t)K * 2 * D * (t))
#define SL_TAUETA(c, t) ((c)->tau_eta + (size_t)K * D * (t))

#define DIMS const int n = d->n, K = d->K, P = d->P, M = d->M, D = d->D; (void)n; (void)K; (void)P; (void)M; (void)D

/* arma::dot(nu.row(k), B.row(l)) */
/* N(pinv(Prec) rhs, pinv(Prec)) for a precision that is singular to working accuracy (linalg.c): returns 1 and fills
 * `draw` when that route was taken, 0 when the caller should go on with the reference's inv / pinv + mvnrnd calls */
static int singular_route(const orc_rng* r, uint32_t upd, uint32_t idx0, int P, const double* Prec, const double* rhs, double* draw) {
  if (!orc_prec_is_singular(P, Prec)) return 0;
  double* z = (double*)malloc(sizeof(double) * (size_t)P);
  for (int p = 0; p < P; ++p) z[p] = orc_rnorm(r, upd, idx0 + (uint32_t)p);
  orc_pinv_draw(P, Prec, rhs, z, draw);
  free(z);
  return 1;
}

/* Optional, for the full-size parity tests only (orc_set_row_window(1); default 0 = the reference's dense row products, which
 * is what bench.py's cpu_baseline times): every loop over the entries of a basis row B_i(l, .) runs over the row's non-zero
 * window [W_LO, W_HI) only.  The skipped terms are exact zeros times finite numbers, so every sum is BIT-IDENTICAL to the dense
 * loop (tests/test_oracle_row_window.py); a cubic B-spline row has 4 non-zeros of 30, a multivariate "row" 1 of 50. */
static int g_row_window = 0;
static int W_LO = 0, W_HI = 0;
void orc_set_row_window(int on) { g_row_window = on; }
static inline void set_window(const double* b, int P) {
  W_LO = 0; W_HI = P;
  if (!g_row_window) return;
  while (W_LO < P && b[W_LO] == 0) ++W_LO;
  while (W_HI > W_LO && b[W_HI - 1] == 0) --W_HI;
}
#define WIN(b) set_window((b), P)
#define WIN_P(b, P_) set_window((b), (P_))

static inline double dot_nu(const double* nu_t, int K, int P, int k, const double* b) {
  double s = 0.0;
  for (int p = W_LO; p < W_HI; ++p) s += nu_t[k + (size_t)K * p] * b[p];
  return s;
}
/* arma::dot(Phi.slice(m).row(k), B.row(l)) */
static inline double dot_phi(const double* phi_t, int K, int P, int k, int m, const double* b) {
  const double* s0 = phi_t + (size_t)K * P * m;
  double s = 0.0;
  for (int p = W_LO; p < W_HI; ++p) s += s0[k + (size_t)K * p] * b[p];
  return s;
}
/* arma::dot(eta.slice(k) * X.row(i).t(), B.row(l)) */
static inline double dot_eta(const orc_data* d, const double* eta_t, int k, int i, const double* b) {
  DIMS;
  double s = 0.0;
  for (int p = W_LO; p < W_HI; ++p) {
    double e = 0.0;
    for (int dd = 0; dd < D; ++dd) e += eta_t[p + (size_t)P * (dd + (size_t)D * k)] * XCOV(i, dd);
    s += e * b[p];
  }
  return s;
}
/* arma::dot(xi(iter,k).slice(m) * X.row(i).t(), B.row(l)) */
static inline double dot_xi(const orc_data* d, const double* xi_tk, int m, int i, const double* b) {
  DIMS;
  double s = 0.0;
  for (int p = W_LO; p < W_HI; ++p) {
    double e = 0.0;
    for (int dd = 0; dd < D; ++dd) e += xi_tk[p + (size_t)P * (dd + (size_t)D * m)] * XCOV(i, dd);
    s += e * b[p];
  }
  return s;
}

/* ------------------------------------------------------------------------------------------
 * rdirichlet, Distributions.h:22-45 (alpha <= 0 replaced by 10, :24-28), and calc_lB :51-60
 * ---------------------------------------------------------------------------------------- */
static void rdirichlet(const orc_rng* r, uint32_t upd, uint32_t idx0, int K, const double* alpha_in, double* out) {
  double sum_term = 0.0;
  for (int j = 0; j < K; ++j) {
    double a = alpha_in[j];
    if (a <= 0) a = 10;
    double gam = orc_rgamma(r, upd, idx0 + (uint32_t)j, a, 1.0);
    out[j] = gam;
    sum_term += gam;
  }
  for (int j = 0; j < K; ++j) out[j] = out[j] / sum_term;
}

static double calc_lB(int K, const double* alpha) {
  double log_B = 0.0, acc = 0.0;
  for (int i = 0; i < K; ++i) { log_B = log_B + lgamma(alpha[i]); acc += alpha[i]; }
  log_B = log_B - lgamma(acc);
  return log_B;
}

void orc_rdirichlet(const orc_rng* r, uint32_t upd, uint32_t idx0, int K, const double* alpha, double* out) {
  rdirichlet(r, upd, idx0, K, alpha, out);
}

/* ------------------------------------------------------------------------------------------
 * lpdf_z / lpdf_zTempered, UpdateMixedMembership.h:20-50 / :64-95
 * (covariate version lpdf_z_CovariateAdj adds the eta and xi terms to the mean)
 * ---------------------------------------------------------------------------------------- */
static double lpdf_z(const orc_data* d, const orc_chain* c, int iter, int i, const double* Zrow,
                     double alpha_3, double sigma_sq, double beta_i) {
  DIMS;
  const double* nu_t = SL_NU(c, iter);
  const double* phi_t = SL_PHI(c, iter);
  const double* chi_t = SL_CHI(c, iter);
  const double* pi_t = SL_PI(c, iter);
  double lpdf = 0.0;
  for (int l = 0; l < K; ++l) lpdf = lpdf + ((alpha_3 * pi_t[l] - 1) * log(Zrow[l]));
  const int ni = NI(i);
  for (int l = 0; l < ni; ++l) {
    const double* b = BROW(i, l);
    WIN(b);
    double mean = 0.0;
    for (int k = 0; k < K; ++k) {
      mean = mean + Zrow[k] * dot_nu(nu_t, K, P, k, b);
      if (D > 0) mean = mean + Zrow[k] * dot_eta(d, SL_ETA(c, iter), k, i, b);
      for (int nn = 0; nn < M; ++nn) {
        double t = dot_phi(phi_t, K, P, k, nn, b);
        if (D > 0) t += dot_xi(d, SL_XI(c, iter, k), nn, i, b);
        mean = mean + Zrow[k] * chi_t[i + (size_t)n * nn] * t;
      }
    }
    double rr = YOBS(i, l) - mean;
    lpdf = lpdf - (beta_i * ((rr * rr) / (2 * sigma_sq)));
  }
  return lpdf;
}

/* Z_proposal_density, UpdateMixedMembership.h:102-113 */
static double Z_proposal_density(int K, const double* Z, const double* alpha) {
  double density = 0.0;
  for (int i = 0; i < K; ++i) density = density + (alpha[i] - 1) * log(Z[i]);
  density = density - calc_lB(K, alpha);
  return density;
}

/* updateZ_PM, UpdateMixedMembership.h:131-185 (Tempered :204-261; CovariateAdj :615-690) */
void orc_updateZ_PM(const orc_data* d, const orc_rng* r, double beta_i, int iter, int T, double a_Z_PM, orc_chain* c) {
  DIMS;
  double* Z_t = SL_Z(c, iter);
  const double sigma_sq = c->sigma[iter], alpha_3 = c->alpha3[iter];
  double Zold[16], Z_ph[16], a_old[16], a_new[16];
  for (int i = 0; i < n; ++i) {
    for (int k = 0; k < K; ++k) { Zold[k] = Z_t[i + (size_t)n * k]; a_old[k] = a_Z_PM * Zold[k]; }
    /* Propose new state */
    rdirichlet(r, UPD_Z_PROP, (uint32_t)(i * K), K, a_old, Z_ph);
    double z_lpdf = lpdf_z(d, c, iter, i, Zold, alpha_3, sigma_sq, beta_i);
    double z_new_lpdf = lpdf_z(d, c, iter, i, Z_ph, alpha_3, sigma_sq, beta_i);
    for (int k = 0; k < K; ++k) a_new[k] = a_Z_PM * Z_ph[k];
    double lpdf_propose_new = Z_proposal_density(K, Z_ph, a_old);
    double lpdf_propose_old = Z_proposal_density(K, Zold, a_new);
    double acceptance_prob = z_new_lpdf - z_lpdf + lpdf_propose_old - lpdf_propose_new;
    double rand_unif_var = orc_runif(r, UPD_Z_ACC, (uint32_t)i);
    for (int j = 0; j < K; ++j)
      if (Zold[j] <= 0) acceptance_prob = 1;   /* :170-174 */
    if (log(rand_unif_var) < acceptance_prob)
      for (int k = 0; k < K; ++k) Z_t[i + (size_t)n * k] = Z_ph[k];
  }
  if (iter < (T - 1)) memcpy(SL_Z(c, iter + 1), Z_t, sizeof(double) * (size_t)n * K);
}

/* lpdf_pi_PM UpdatePi.h:39-53; pi_proposal_density :60-71; updatePi_PM :84-116 */
static double lpdf_pi_PM(int n, int K, const double* cvec, double alpha_3, const double* pi, const double* Z_t) {
  double lpdf = 0.0;
  double ap[16];
  for (int k = 0; k < K; ++k) {
    lpdf = lpdf + ((cvec[k] - 1) * log(pi[k]));
    for (int i = 0; i < n; ++i) lpdf = lpdf + (((alpha_3 * pi[k]) - 1) * log(Z_t[i + (size_t)n * k]));
    ap[k] = alpha_3 * pi[k];
  }
  lpdf = lpdf - (n * calc_lB(K, ap));
  return lpdf;
}

void orc_updatePi_PM(const orc_data* d, const orc_rng* r, int iter, int T, const double* cvec, double a_pi_PM, orc_chain* c) {
  DIMS;
  double* pi_t = SL_PI(c, iter);
  const double* Z_t = SL_Z(c, iter);
  const double alpha_3 = c->alpha3[iter];
  double pi_ph[16], a_old[16], a_new[16];
  for (int k = 0; k < K; ++k) a_old[k] = a_pi_PM * pi_t[k];
  rdirichlet(r, UPD_PI_PROP, 0, K, a_old, pi_ph);
  double lpdf_new = lpdf_pi_PM(n, K, cvec, alpha_3, pi_ph, Z_t);
  double lpdf_old = lpdf_pi_PM(n, K, cvec, alpha_3, pi_t, Z_t);
  for (int k = 0; k < K; ++k) a_new[k] = a_pi_PM * pi_ph[k];
  double lpdf_propose_new = Z_proposal_density(K, pi_ph, a_old);
  double lpdf_propose_old = Z_proposal_density(K, pi_t, a_new);
  double acceptance_prob = lpdf_new - lpdf_old + lpdf_propose_old - lpdf_propose_new;
  double rand_unif_var = orc_runif(r, UPD_PI_ACC, 0);
  if (log(rand_unif_var) < acceptance_prob)
    for (int k = 0; k < K; ++k) pi_t[k] = pi_ph[k];
  if ((T - 1) > iter) memcpy(SL_PI(c, iter + 1), pi_t, sizeof(double) * (size_t)K);
}

/* lpdf_alpha3 UpdateAlpha3.h:10-27 (note d_truncnorm(alpha_3_ph, alpha_3_ph, ...), :23-24);
 * updateAlpha3 :36-63 */
static double lpdf_alpha3(int n, int K, const double* pi, double b, const double* Z_t,
                          double alpha_3, double alpha_3_ph, double sigma_alpha_3) {
  double lpdf = (-b) * alpha_3;
  double ap[16];
  for (int k = 0; k < K; ++k) {
    for (int i = 0; i < n; ++i) lpdf = lpdf + (((alpha_3 * pi[k]) - 1) * log(Z_t[i + (size_t)n * k]));
    ap[k] = alpha_3 * pi[k];
  }
  lpdf = lpdf - (n * calc_lB(K, ap));
  lpdf = lpdf + orc_dtruncnorm_log(alpha_3_ph, alpha_3_ph, sigma_alpha_3, 0, INFINITY);
  return lpdf;
}

void orc_updateAlpha3(const orc_data* d, const orc_rng* r, int iter, int T, double b, double var_alpha3, orc_chain* c) {
  DIMS;
  const double* pi_t = SL_PI(c, iter);
  const double* Z_t = SL_Z(c, iter);
  double cur = c->alpha3[iter];
  double alpha_3_ph = orc_rtruncnorm(r, UPD_A3_PROP, 0, cur, var_alpha3, 0, INFINITY);
  double lpdf_old = lpdf_alpha3(n, K, pi_t, b, Z_t, cur, alpha_3_ph, var_alpha3);
  double lpdf_new = lpdf_alpha3(n, K, pi_t, b, Z_t, alpha_3_ph, cur, var_alpha3);
  double acceptance_prob = lpdf_new - lpdf_old;
  double rand_unif_var = orc_runif(r, UPD_A3_ACC, 0);
  if (log(rand_unif_var) < acceptance_prob) c->alpha3[iter] = alpha_3_ph;
  if ((T - 1) > iter) c->alpha3[iter + 1] = c->alpha3[iter];
}

/* tilde_tau cumulative product, BFMMM.h:1514-1519 */
void orc_tilde_tau(int K, int M, const double* delta_slice, double* tilde_tau) {
  for (int k = 0; k < K; ++k) {
    tilde_tau[k] = delta_slice[k];
    for (int j = 1; j < M; ++j)
      tilde_tau[k + (size_t)K * j] = tilde_tau[k + (size_t)K * (j - 1)] * delta_slice[k + (size_t)K * j];
  }
}

/* BFMMM.h:3983-3990 */
void orc_tilde_tau_xi(int K, int M, int D, const double* dxi, double* tt) {
  for (int k = 0; k < K; ++k)
    for (int m = 0; m < D; ++m) {
      tt[k + (size_t)K * (0 + (size_t)M * m)] = dxi[k + (size_t)K * (0 + (size_t)M * m)];
      for (int j = 1; j < M; ++j)
        tt[k + (size_t)K * (j + (size_t)M * m)] =
            tt[k + (size_t)K * (j - 1 + (size_t)M * m)] * dxi[k + (size_t)K * (j + (size_t)M * m)];
    }
}

/* updatePhi, UpdatePhi.h:23-89 (Tempered :107-174: the factor is beta_i / sigma_sq;
 * CovariateAdj :351-445; MV :190-249 uses B = I) */
void orc_updatePhi(const orc_data* d, const orc_rng* r, double beta_i, int iter, int T, const double* tilde_tau, orc_chain* c) {
  DIMS;
  const double* nu_t = SL_NU(c, iter);
  double* phi_t = SL_PHI(c, iter);
  const double* gamma_t = SL_GAMMA(c, iter);
  const double* Z_t = SL_Z(c, iter);
  const double* chi_t = SL_CHI(c, iter);
  const double sigma_sq = c->sigma[iter];
  double* m_1 = (double*)malloc(sizeof(double) * (size_t)P);
  double* M_1 = (double*)malloc(sizeof(double) * (size_t)P * P);
  double* mean = (double*)malloc(sizeof(double) * (size_t)P);
  double* draw = (double*)malloc(sizeof(double) * (size_t)P);
  for (int j = 0; j < K; ++j) {
    for (int m = 0; m < M; ++m) {
      memset(m_1, 0, sizeof(double) * (size_t)P);
      memset(M_1, 0, sizeof(double) * (size_t)P * P);
      for (int i = 0; i < n; ++i) {
        const double zij = Z_t[i + (size_t)n * j];
        if (zij != 0) {
          const int ni = NI(i);
          const double chim = chi_t[i + (size_t)n * m];
          for (int l = 0; l < ni; ++l) {
            const double* b = BROW(i, l);
            WIN(b);
            double ph = YOBS(i, l) - zij * dot_nu(nu_t, K, P, j, b);
            if (D > 0) ph -= zij * dot_eta(d, SL_ETA(c, iter), j, i, b);
            const double w2 = zij * zij * (chim * chim);
            for (int q = W_LO; q < W_HI; ++q)
              for (int p = W_LO; p < W_HI; ++p) M_1[p + (size_t)P * q] += w2 * b[p] * b[q];
            for (int k = 0; k < K; ++k) {
              const double zik = Z_t[i + (size_t)n * k];
              for (int nn = 0; nn < M; ++nn) {
                const double chin = chi_t[i + (size_t)n * nn];
                if (k == j) {
                  if (nn != m) ph = ph - (zij * chin * dot_phi(phi_t, K, P, k, nn, b));
                  if (D > 0) ph = ph - (zij * chin * dot_xi(d, SL_XI(c, iter, k), nn, i, b));
                } else {
                  double t = dot_phi(phi_t, K, P, k, nn, b);
                  if (D > 0) t += dot_xi(d, SL_XI(c, iter, k), nn, i, b);
                  ph = ph - (zik * chin * t);
                }
              }
              if (k != j) {
                double t = dot_nu(nu_t, K, P, k, b);
                if (D > 0) t += dot_eta(d, SL_ETA(c, iter), k, i, b);
                ph = ph - zik * t;
              }
            }
            const double w = zij * chim * ph;
            for (int p = W_LO; p < W_HI; ++p) m_1[p] += w * b[p];
          }
        }
      }
      const double f = beta_i / sigma_sq;
      for (int p = 0; p < P; ++p) m_1[p] *= f;
      for (int q = 0; q < P * P; ++q) M_1[q] *= f;
      /* Add on diagonal component (:76-78) */
      for (int k = 0; k < P; ++k)
        M_1[k + (size_t)P * k] += tilde_tau[j + (size_t)K * m] * gamma_t[j + (size_t)K * (k + (size_t)P * m)];
      if (!singular_route(r, UPD_PHI, (uint32_t)((j * M + m) * P), P, M_1, m_1, draw)) {
      orc_inv(P, M_1);
      for (int p = 0; p < P; ++p) {
        double s = 0.0;
        for (int q = 0; q < P; ++q) s += M_1[p + (size_t)P * q] * m_1[q];
        mean[p] = s;
      }
      orc_mvnrnd(r, UPD_PHI, (uint32_t)((j * M + m) * P), P, mean, M_1, draw);
      }
      for (int p = 0; p < P; ++p) phi_t[j + (size_t)K * (p + (size_t)P * m)] = draw[p];
    }
  }
  if (iter < (T - 1)) memcpy(SL_PHI(c, iter + 1), phi_t, sizeof(double) * (size_t)K * P * M);
  free(m_1); free(M_1); free(mean); free(draw);
}

/* updateDelta, UpdateDelta.h:17-64 */
void orc_updateDelta(const orc_data* d, const orc_rng* r, int iter, int T, orc_chain* c) {
  DIMS;
  const double* phi = SL_PHI(c, iter);
  const double* gamma = SL_GAMMA(c, iter);
  const double* a = SL_A(c, iter);
  double* delta = SL_DELTA(c, iter);
#define PH3(k, j, m) phi[(k) + (size_t)K * ((j) + (size_t)P * (m))]
#define GA3(k, j, m) gamma[(k) + (size_t)K * ((j) + (size_t)P * (m))]
  for (int k = 0; k < K; ++k) {
    for (int i = 0; i < M; ++i) {
      double param1, param2, tilde_tau;
      if (i == 0) {
        param1 = a[k] + ((P * M) / 2.0);
        param2 = 1;
        for (int j = 0; j < P; ++j) {
          param2 = param2 + (0.5 * GA3(k, j, 0) * (PH3(k, j, 0) * PH3(k, j, 0)));
          for (int m = 1; m < M; ++m) {
            tilde_tau = 1;
            for (int nn = 1; nn <= m; ++nn) tilde_tau = tilde_tau * delta[k + (size_t)K * nn];
            param2 = param2 + (0.5 * GA3(k, j, m) * tilde_tau * (PH3(k, j, m) * PH3(k, j, m)));
          }
        }
      } else {
        param1 = a[k + (size_t)K * 1] + ((P * (M - i)) / 2.0);
        param2 = 1;
        for (int j = 0; j < P; ++j) {
          for (int m = i; m < M; ++m) {
            tilde_tau = 1;
            for (int nn = 0; nn <= m; ++nn)
              if (nn != i) tilde_tau = tilde_tau * delta[k + (size_t)K * nn];
            param2 = param2 + (0.5 * GA3(k, j, m) * tilde_tau * (PH3(k, j, m) * PH3(k, j, m)));
          }
        }
      }
      delta[k + (size_t)K * i] = orc_rgamma(r, UPD_DELTA, (uint32_t)(k * M + i), param1, 1 / param2);
    }
  }
#undef PH3
#undef GA3
  if (iter < (T - 1)) memcpy(SL_DELTA(c, iter + 1), delta, sizeof(double) * (size_t)K * M);
}

/* lpdf_a1 UpdateA.h:17-24, lpdf_a2 :33-44 (logGamma = log(tgamma(x)), Distributions.h:13-15) */
static double logGamma(double x) { return log(tgamma(x)); }
static double lpdf_a1(double alpha_1l, double beta_1l, double a, double delta) {
  return -logGamma(a) + (a - 1) * log(delta) + (alpha_1l - 1) * log(a) - (a * beta_1l);
}
static double lpdf_a2(double alpha_2l, double beta_2l, double a, int M, const double* delta_row, int stride) {
  double x = M - 1;
  double lpdf = -x * logGamma(a) + (alpha_2l - 1) * log(a) - (a * beta_2l);
  for (int i = 1; i < M; ++i) lpdf = lpdf + (a - 1) * log(delta_row[(size_t)i * stride]);
  return lpdf;
}

/* one (j,i) cell of updateA / updateAXi: UpdateA.h:75-117 */
static double update_a_cell(const orc_rng* r, uint32_t upd_prop, uint32_t upd_acc, uint32_t idx, int first,
                            const orc_hyper* h, double cur, double delta0, int M, const double* delta_row, int stride) {
  const double sd = first ? (h->var_epsilon1 / h->beta1l) : (h->var_epsilon2 / h->beta2l);
  double a_lpdf, a_new_lpdf;
  double new_a = orc_rtruncnorm(r, upd_prop, idx, cur, sd, 0, INFINITY);
  if (first) {
    a_lpdf = lpdf_a1(h->alpha1l, h->beta1l, cur, delta0);
    a_new_lpdf = lpdf_a1(h->alpha1l, h->beta1l, new_a, delta0);
  } else {
    a_lpdf = lpdf_a2(h->alpha2l, h->beta2l, cur, M, delta_row, stride);
    a_new_lpdf = lpdf_a2(h->alpha2l, h->beta2l, new_a, M, delta_row, stride);
  }
  double acceptance_prob = (a_new_lpdf + orc_dtruncnorm_log(cur, new_a, sd, 0, INFINITY)) - a_lpdf -
                           orc_dtruncnorm_log(new_a, cur, sd, 0, INFINITY);
  double rand_unif_var = orc_runif(r, upd_acc, idx);
  if (log(rand_unif_var) < acceptance_prob) return new_a;
  return cur;
}

/* updateA, UpdateA.h:58-123 */
void orc_updateA(const orc_data* d, const orc_rng* r, int iter, int T, const orc_hyper* h, orc_chain* c) {
  DIMS;
  double* a = SL_A(c, iter);
  const double* delta = SL_DELTA(c, iter);
  for (int j = 0; j < K; ++j)
    for (int i = 0; i < 2; ++i)
      a[j + (size_t)K * i] = update_a_cell(r, UPD_A_PROP, UPD_A_ACC, (uint32_t)(j * 2 + i), i == 0, h,
                                           a[j + (size_t)K * i], delta[j], M, delta + j, K);
  if (iter < (T - 1)) memcpy(SL_A(c, iter + 1), a, sizeof(double) * (size_t)K * 2);
}

/* updateGamma, UpdateGamma.h:17-37 */
void orc_updateGamma(const orc_data* d, const orc_rng* r, int iter, int T, double nu_gamma, orc_chain* c) {
  DIMS;
  const double* phi = SL_PHI(c, iter);
  const double* delta = SL_DELTA(c, iter);
  double* gamma = SL_GAMMA(c, iter);
  for (int i = 0; i < K; ++i)
    for (int l = 0; l < P; ++l) {
      double placeholder = 1;
      for (int j = 0; j < M; ++j) {
        placeholder = placeholder * delta[i + (size_t)K * j];
        const double ph = phi[i + (size_t)K * (l + (size_t)P * j)];
        gamma[i + (size_t)K * (l + (size_t)P * j)] =
            orc_rgamma(r, UPD_GAMMA, (uint32_t)((i * P + l) * M + j), (nu_gamma + 1) / 2,
                       2 / (nu_gamma + placeholder * (ph * ph)));
      }
    }
  if (iter < T - 1) memcpy(SL_GAMMA(c, iter + 1), gamma, sizeof(double) * (size_t)K * P * M);
}

/* updateNu, UpdateNu.h:24-74 (Tempered :93-144; CovariateAdj :287-347; MV :160-205 with
 * prior (1/tau) I instead of tau P) */
void orc_updateNu(const orc_data* d, const orc_rng* r, double beta_i, int iter, int T, orc_chain* c) {
  DIMS;
  double* nu_t = SL_NU(c, iter);
  const double* phi_t = SL_PHI(c, iter);
  const double* Z_t = SL_Z(c, iter);
  const double* chi_t = SL_CHI(c, iter);
  const double sigma = c->sigma[iter];
  double* b_1 = (double*)malloc(sizeof(double) * (size_t)P);
  double* B_1 = (double*)malloc(sizeof(double) * (size_t)P * P);
  double* mean = (double*)malloc(sizeof(double) * (size_t)P);
  double* draw = (double*)malloc(sizeof(double) * (size_t)P);
  for (int j = 0; j < K; ++j) {
    memset(b_1, 0, sizeof(double) * (size_t)P);
    memset(B_1, 0, sizeof(double) * (size_t)P * P);
    for (int i = 0; i < n; ++i) {
      const double zij = Z_t[i + (size_t)n * j];
      if (zij != 0) {
        const int ni = NI(i);
        for (int l = 0; l < ni; ++l) {
          const double* b = BROW(i, l);
          WIN(b);
          double ph = YOBS(i, l);
          const double w2 = zij * zij;
          for (int q = W_LO; q < W_HI; ++q)
            for (int p = W_LO; p < W_HI; ++p) B_1[p + (size_t)P * q] += w2 * b[p] * b[q];
          for (int k = 0; k < K; ++k) {
            const double zik = Z_t[i + (size_t)n * k];
            if (zik != 0) {
              if (k != j) ph = ph - zik * dot_nu(nu_t, K, P, k, b);
              if (D > 0) ph = ph - zik * dot_eta(d, SL_ETA(c, iter), k, i, b);
              for (int nn = 0; nn < M; ++nn) {
                double t = dot_phi(phi_t, K, P, k, nn, b);
                if (D > 0) t += dot_xi(d, SL_XI(c, iter, k), nn, i, b);
                ph = ph - zik * (chi_t[i + (size_t)n * nn] * t);
              }
            }
          }
          const double w = zij * ph;
          for (int p = W_LO; p < W_HI; ++p) b_1[p] += w * b[p];
        }
      }
    }
    const double f = beta_i / sigma;
    for (int p = 0; p < P; ++p) b_1[p] *= f;
    for (int q = 0; q < P * P; ++q) B_1[q] *= f;
    const double tau_j = c->tau[iter + (size_t)T * j];
    if (d->mv) {
      for (int p = 0; p < P; ++p) B_1[p + (size_t)P * p] += 1.0 / tau_j;   /* UpdateNu.h:197 */
    } else {
      for (int q = 0; q < P * P; ++q) B_1[q] += tau_j * d->Pmat[q];
    }
    if (!singular_route(r, UPD_NU, (uint32_t)(j * P), P, B_1, b_1, draw)) {
    orc_pinv_sym(P, B_1);
    for (int q = 0; q < P; ++q)
      for (int p = 0; p < q; ++p) {
        double s = (B_1[p + (size_t)P * q] + B_1[q + (size_t)P * p]) / 2;
        B_1[p + (size_t)P * q] = s;
        B_1[q + (size_t)P * p] = s;
      }
    for (int p = 0; p < P; ++p) {
      double s = 0.0;
      for (int q = 0; q < P; ++q) s += B_1[p + (size_t)P * q] * b_1[q];
      mean[p] = s;
    }
    orc_mvnrnd(r, UPD_NU, (uint32_t)(j * P), P, mean, B_1, draw);
    }
    for (int p = 0; p < P; ++p) nu_t[j + (size_t)K * p] = draw[p];
  }
  if (iter < (T - 1)) memcpy(SL_NU(c, iter + 1), nu_t, sizeof(double) * (size_t)K * P);
  free(b_1); free(B_1); free(mean); free(draw);
}

/* updateTau, UpdateTau.h:18-36 (integer division nu.n_cols / 2, :29); updateTauMV :47-63
 * stores the inverse (:58) and has no P matrix */
void orc_updateTau(const orc_data* d, const orc_rng* r, int iter, int T, double alpha, double beta, orc_chain* c) {
  DIMS;
  const double* nu_t = SL_NU(c, iter);
  for (int i = 0; i < K; ++i) {
    double a = alpha + (P / 2);
    double q = 0.0;
    if (d->mv) {
      for (int p = 0; p < P; ++p) q += nu_t[i + (size_t)K * p] * nu_t[i + (size_t)K * p];
    } else {
      for (int p = 0; p < P; ++p) {
        double s = 0.0;
        for (int p2 = 0; p2 < P; ++p2) s += d->Pmat[p + (size_t)P * p2] * nu_t[i + (size_t)K * p2];
        q += nu_t[i + (size_t)K * p] * s;
      }
    }
    double b = beta + (0.5 * q);
    double g = orc_rgamma(r, UPD_TAU, (uint32_t)i, a, 1 / b);
    c->tau[iter + (size_t)T * i] = d->mv ? (1 / g) : g;
  }
  if (iter < (T - 1))
    for (int i = 0; i < K; ++i) c->tau[(iter + 1) + (size_t)T * i] = c->tau[iter + (size_t)T * i];
}

/* fitted mean at observation (i,l): shared by updateSigma / calcLikelihood, which walk
 * k -> n with the Z(i,k) != 0 skip (UpdateSigma.h:39-46, CalculateLikelihood.h:30-39) */
static double fitted_skipzero(const orc_data* d, const orc_chain* c, int iter, int i, const double* b) {
  WIN_P(b, d->P);
  DIMS;
  const double* nu_t = SL_NU(c, iter);
  const double* phi_t = SL_PHI(c, iter);
  const double* Z_t = SL_Z(c, iter);
  const double* chi_t = SL_CHI(c, iter);
  double mean = 0.0;
  for (int k = 0; k < K; ++k) {
    const double zik = Z_t[i + (size_t)n * k];
    if (zik != 0) {
      mean = mean + zik * dot_nu(nu_t, K, P, k, b);
      if (D > 0) mean = mean + zik * dot_eta(d, SL_ETA(c, iter), k, i, b);
      for (int nn = 0; nn < M; ++nn) {
        double t = dot_phi(phi_t, K, P, k, nn, b);
        if (D > 0) t += dot_xi(d, SL_XI(c, iter, k), nn, i, b);
        mean = mean + zik * chi_t[i + (size_t)n * nn] * t;
      }
    }
  }
  return mean;
}

/* fitted mean and observation of (i, l) for the post-processing restatements (post.c) */
double orc_fitted(const orc_data* d, const orc_chain* c, int iter, int i, int l) {
  DIMS;
  return fitted_skipzero(d, c, iter, i, BROW(i, l));
}
double orc_yobs(const orc_data* d, int i, int l) { return YOBS(i, l); }
/* B_i.row(l) . (nu.row(k) + eta.slice(k) X.row(i)')  (mt = 0)  or  B_i.row(l) . (Phi.slice(mt-1).row(k) + xi(iter,k).slice(mt-1) X.row(i)') */
double orc_row_dot(const orc_data* d, const orc_chain* c, int iter, int i, int l, int k, int mt) {
  DIMS;
  const double* b = BROW(i, l);
  WIN(b);
  if (mt == 0) return dot_nu(SL_NU(c, iter), K, P, k, b) + (D > 0 ? dot_eta(d, SL_ETA(c, iter), k, i, b) : 0.0);
  return dot_phi(SL_PHI(c, iter), K, P, k, mt - 1, b) + (D > 0 ? dot_xi(d, SL_XI(c, iter, k), mt - 1, i, b) : 0.0);
}
int orc_ni(const orc_data* d, int i) { return NI(i); }

/* updateSigma, UpdateSigma.h:22-58: a += n_i / 2 with INTEGER division (:49);
 * Tempered :75-113 uses (beta_i * n_i) / 2 in floating point (:103) and beta_i/2 weights;
 * MV :127-156 uses y_obs.n_elem / 2 (integer, total element count, :150);
 * TemperedMV :172-205 uses (beta_i * y_obs.n_elem) / 2. */
void orc_updateSigma(const orc_data* d, const orc_rng* r, double beta_i, int tempered, int iter, int T,
                     double alpha_0, double beta_0, orc_chain* c) {
  DIMS;
  double a = 0, b_1 = 0;
  int64_t total = 0;
  for (int i = 0; i < n; ++i) {
    const int ni = NI(i);
    for (int l = 0; l < ni; ++l) {
      double b = YOBS(i, l) - fitted_skipzero(d, c, iter, i, BROW(i, l));
      b_1 = b_1 + (tempered ? (beta_i / 2) : 0.5) * (b * b);
    }
    total += ni;
    if (!d->mv) {
      if (tempered) a = a + ((beta_i * ni) / 2);
      else a = a + (ni / 2);
    }
  }
  if (d->mv) a = tempered ? ((beta_i * (double)total) / 2) : (double)(total / 2);
  b_1 = b_1 + beta_0;
  a = a + alpha_0;
  c->sigma[iter] = 1 / orc_rgamma(r, UPD_SIGMA, 0, a, 1 / b_1);
  if (iter < (T - 1)) c->sigma[iter + 1] = c->sigma[iter];
}

/* updateChi, UpdateChi.h:19-64 (Tempered :79-125; CovariateAdj :242-307) */
void orc_updateChi(const orc_data* d, const orc_rng* r, double beta_i, int iter, int T, orc_chain* c) {
  DIMS;
  const double* nu_t = SL_NU(c, iter);
  const double* phi_t = SL_PHI(c, iter);
  const double* Z_t = SL_Z(c, iter);
  double* chi_t = SL_CHI(c, iter);
  const double sigma = c->sigma[iter];
  for (int i = 0; i < n; ++i) {
    const int ni = NI(i);
    for (int m = 0; m < M; ++m) {
      double w = 0, W = 0;
      for (int l = 0; l < ni; ++l) {
        const double* b = BROW(i, l);
        WIN(b);
        double ph = 0;
        for (int k2 = 0; k2 < K; ++k2) {
          double t = dot_phi(phi_t, K, P, k2, m, b);
          if (D > 0) t += dot_xi(d, SL_XI(c, iter, k2), m, i, b);
          ph = ph + Z_t[i + (size_t)n * k2] * t;
        }
        w = w + ph * YOBS(i, l);
        W = W + ph * ph;
        for (int k1 = 0; k1 < K; ++k1) {
          const double z1 = Z_t[i + (size_t)n * k1];
          if (z1 != 0) {
            double t0 = dot_nu(nu_t, K, P, k1, b);
            if (D > 0) t0 += dot_eta(d, SL_ETA(c, iter), k1, i, b);
            w = w - z1 * ph * t0;
            for (int nn = 0; nn < M; ++nn)
              if (nn != m) {
                double t = dot_phi(phi_t, K, P, k1, nn, b);
                if (D > 0) t += dot_xi(d, SL_XI(c, iter, k1), nn, i, b);
                w = w - z1 * ph * chi_t[i + (size_t)n * nn] * t;
              }
          }
        }
      }
      w = (w * beta_i) / sigma;
      W = 1 + ((W * beta_i) / sigma);
      W = 1 / W;
      chi_t[i + (size_t)n * m] = orc_rnorm_ms(r, UPD_CHI, (uint32_t)(i * M + m), W * w, sqrt(W));      /* R::rnorm(W * w, sqrt(W)), :58 */
    }
  }
  if (iter < (T - 1)) memcpy(SL_CHI(c, iter + 1), chi_t, sizeof(double) * (size_t)n * M);
}

/* calcLikelihood, CalculateLikelihood.h:19-44 (R::dnorm(y, mean, sqrt(sigma), log));
 * calcLikelihoodMV :137-160 uses (y_obs.n_cols / 2) * log(2 pi sigma) with INTEGER division (:155) */
double orc_calcLikelihood(const orc_data* d, int iter, const orc_chain* c) {
  DIMS;
  const double sigma = c->sigma[iter];
  const double sd = sqrt(sigma);
  double log_lik = 0;
  for (int i = 0; i < n; ++i) {
    const int ni = NI(i);
    if (d->mv) {
      double ss = 0.0;
      for (int l = 0; l < ni; ++l) {
        double rr = YOBS(i, l) - fitted_skipzero(d, c, iter, i, BROW(i, l));
        ss += rr * rr;
      }
      log_lik = log_lik - ((P / 2) * log(2 * 3.14159265358979323846 * sigma)) - ((1 / (sigma * 2)) * ss);
      continue;
    }
    for (int l = 0; l < ni; ++l) {
      double mean = fitted_skipzero(d, c, iter, i, BROW(i, l));
      double z = (YOBS(i, l) - mean) / sd;
      log_lik = log_lik + (-(0.91893853320467274178 + 0.5 * z * z + log(sd)));
    }
  }
  return log_lik;
}

/* calculatePZetaCovariateAdj / calculatePZetaMVCovariateAdj, CalculateTTAcceptance.h:195-232, :296-330: the tempered
 * log-likelihood of state `iter`, per observation, with the covariate-adjusted mean (xi read at the state's own slot:
 * the documented deviation from the reference's wrong-index reads). */
double orc_calculatePZetaCov(const orc_data* d, double beta_i, int iter, const orc_chain* c) {
  DIMS;
  (void)K; (void)M; (void)D;
  const double sigma = c->sigma[iter];
  double logAcceptance = 0;
  for (int i = 0; i < n; ++i) {
    const int ni = NI(i);
    if (d->mv) {
      double ss = 0.0;
      for (int l = 0; l < ni; ++l) {
        const double rr = YOBS(i, l) - fitted_skipzero(d, c, iter, i, BROW(i, l));
        ss += rr * rr;
      }
      logAcceptance = logAcceptance + ((-(beta_i / 2) * log(sigma) * P) - (beta_i / (2 * sigma)) * ss);
      continue;
    }
    for (int l = 0; l < ni; ++l) {
      const double res = YOBS(i, l) - fitted_skipzero(d, c, iter, i, BROW(i, l));
      logAcceptance = logAcceptance + ((-(beta_i / 2) * log(sigma)) - (beta_i / (2 * sigma)) * (res * res));
    }
  }
  return logAcceptance;
}

/* ========================= covariate-adjusted extras ======================================= */

/* updateEta, UpdateEta.h:28-94 (d outer, j inner; pinv + symmetrise; Tempered :116-185) */
void orc_updateEta(const orc_data* d, const orc_rng* r, double beta_i, int iter, int T, orc_chain* c) {
  DIMS;
  const double* nu_t = SL_NU(c, iter);
  const double* phi_t = SL_PHI(c, iter);
  const double* Z_t = SL_Z(c, iter);
  const double* chi_t = SL_CHI(c, iter);
  double* eta_t = SL_ETA(c, iter);
  const double* tau_eta = SL_TAUETA(c, iter);
  const double sigma = c->sigma[iter];
  double* b_1 = (double*)malloc(sizeof(double) * (size_t)P);
  double* B_1 = (double*)malloc(sizeof(double) * (size_t)P * P);
  double* mean = (double*)malloc(sizeof(double) * (size_t)P);
  double* draw = (double*)malloc(sizeof(double) * (size_t)P);
  for (int dd = 0; dd < D; ++dd) {
    for (int j = 0; j < K; ++j) {
      memset(b_1, 0, sizeof(double) * (size_t)P);
      memset(B_1, 0, sizeof(double) * (size_t)P * P);
      for (int i = 0; i < n; ++i) {
        const double zij = Z_t[i + (size_t)n * j];
        if (zij != 0) {
          const int ni = NI(i);
          const double xid = XCOV(i, dd);
          for (int l = 0; l < ni; ++l) {
            const double* b = BROW(i, l);
            WIN(b);
            double ph = YOBS(i, l);
            const double w2 = zij * zij * xid * xid;
            for (int q = W_LO; q < W_HI; ++q)
              for (int p = W_LO; p < W_HI; ++p) B_1[p + (size_t)P * q] += w2 * b[p] * b[q];
            for (int rr = 0; rr < D; ++rr)
              if (rr != dd) {
                double s = 0.0;
                for (int p = W_LO; p < W_HI; ++p) s += eta_t[p + (size_t)P * (rr + (size_t)D * j)] * b[p];
                ph = ph - zij * XCOV(i, rr) * s;
              }
            for (int k = 0; k < K; ++k) {
              const double zik = Z_t[i + (size_t)n * k];
              if (zik != 0) {
                if (k != j) ph = ph - zik * dot_eta(d, eta_t, k, i, b);
                ph = ph - zik * dot_nu(nu_t, K, P, k, b);
                for (int nn = 0; nn < M; ++nn)
                  ph = ph - zik * chi_t[i + (size_t)n * nn] *
                                (dot_phi(phi_t, K, P, k, nn, b) + dot_xi(d, SL_XI(c, iter, k), nn, i, b));
              }
            }
            const double w = zij * xid * ph;
            for (int p = W_LO; p < W_HI; ++p) b_1[p] += w * b[p];
          }
        }
      }
      const double f = beta_i / sigma;
      for (int p = 0; p < P; ++p) b_1[p] *= f;
      for (int q = 0; q < P * P; ++q) B_1[q] *= f;
      const double te = tau_eta[j + (size_t)K * dd];
      if (d->mv) {
        for (int p = 0; p < P; ++p) B_1[p + (size_t)P * p] += 1.0 / te;
      } else {
        for (int q = 0; q < P * P; ++q) B_1[q] += te * d->Pmat[q];
      }
      if (!singular_route(r, UPD_ETA, (uint32_t)((dd * K + j) * P), P, B_1, b_1, draw)) {
      orc_pinv_sym(P, B_1);
      for (int q = 0; q < P; ++q)
        for (int p = 0; p < q; ++p) {
          double s = (B_1[p + (size_t)P * q] + B_1[q + (size_t)P * p]) / 2;
          B_1[p + (size_t)P * q] = s;
          B_1[q + (size_t)P * p] = s;
        }
      for (int p = 0; p < P; ++p) {
        double s = 0.0;
        for (int q = 0; q < P; ++q) s += B_1[p + (size_t)P * q] * b_1[q];
        mean[p] = s;
      }
      orc_mvnrnd(r, UPD_ETA, (uint32_t)((dd * K + j) * P), P, mean, B_1, draw);
      }
      for (int p = 0; p < P; ++p) eta_t[p + (size_t)P * (dd + (size_t)D * j)] = draw[p];
    }
  }
  if (iter < (T - 1)) memcpy(SL_ETA(c, iter + 1), eta_t, sizeof(double) * (size_t)P * D * K);
  free(b_1); free(B_1); free(mean); free(draw);
}

/* updateTauEta, UpdateTau.h:75-95 (eta.n_rows / 2 integer division, :87); MV :106-124 */
void orc_updateTauEta(const orc_data* d, const orc_rng* r, int iter, int T, double alpha, double beta, orc_chain* c) {
  DIMS;
  const double* eta_t = SL_ETA(c, iter);
  double* te = SL_TAUETA(c, iter);
  for (int j = 0; j < K; ++j)
    for (int i = 0; i < D; ++i) {
      double a = alpha + (P / 2);
      const double* e = eta_t + (size_t)P * (i + (size_t)D * j);
      double q = 0.0;
      if (d->mv) {
        for (int p = 0; p < P; ++p) q += e[p] * e[p];
      } else {
        for (int p = 0; p < P; ++p) {
          double s = 0.0;
          for (int p2 = 0; p2 < P; ++p2) s += d->Pmat[p + (size_t)P * p2] * e[p2];
          q += e[p] * s;
        }
      }
      double b = beta + (0.5 * q);
      double g = orc_rgamma(r, UPD_TAU_ETA, (uint32_t)(j * D + i), a, 1 / b);
      te[j + (size_t)K * i] = d->mv ? (1 / g) : g;
    }
  if (iter < (T - 1)) memcpy(SL_TAUETA(c, iter + 1), te, sizeof(double) * (size_t)K * D);
}

/* updateXiCovariateAdj, UpdateXi.h:26-93 ((j, m, d) order; inv, no symmetrisation;
 * residual = full residual with the own term added back, :67-69) */
void orc_updateXi(const orc_data* d, const orc_rng* r, double beta_i, int iter, int T, const double* tilde_tau_xi, orc_chain* c) {
  DIMS;
  const double* nu_t = SL_NU(c, iter);
  const double* phi_t = SL_PHI(c, iter);
  const double* Z_t = SL_Z(c, iter);
  const double* chi_t = SL_CHI(c, iter);
  const double* eta_t = SL_ETA(c, iter);
  const double sigma_sq = c->sigma[iter];
  double* m_1 = (double*)malloc(sizeof(double) * (size_t)P);
  double* M_1 = (double*)malloc(sizeof(double) * (size_t)P * P);
  double* mean = (double*)malloc(sizeof(double) * (size_t)P);
  double* draw = (double*)malloc(sizeof(double) * (size_t)P);
  for (int j = 0; j < K; ++j) {
    double* xi_j = SL_XI(c, iter, j);
    const double* gxi_j = SL_GXI(c, iter, j);
    for (int m = 0; m < M; ++m) {
      for (int dd = 0; dd < D; ++dd) {
        memset(m_1, 0, sizeof(double) * (size_t)P);
        memset(M_1, 0, sizeof(double) * (size_t)P * P);
        for (int i = 0; i < n; ++i) {
          const double zij = Z_t[i + (size_t)n * j];
          if (zij != 0) {
            const int ni = NI(i);
            const double xid = XCOV(i, dd);
            const double chim = chi_t[i + (size_t)n * m];
            for (int l = 0; l < ni; ++l) {
              const double* b = BROW(i, l);
              WIN(b);
              double ph = YOBS(i, l);
              const double w2 = zij * zij * xid * xid * (chim * chim);
              for (int q = W_LO; q < W_HI; ++q)
                for (int p = W_LO; p < W_HI; ++p) M_1[p + (size_t)P * q] += w2 * b[p] * b[q];
              for (int k = 0; k < K; ++k) {
                const double zik = Z_t[i + (size_t)n * k];
                ph = ph - (zik * (dot_nu(nu_t, K, P, k, b) + dot_eta(d, eta_t, k, i, b)));
                for (int nn = 0; nn < M; ++nn)
                  ph = ph - (zik * chi_t[i + (size_t)n * nn] *
                             (dot_phi(phi_t, K, P, k, nn, b) + dot_xi(d, SL_XI(c, iter, k), nn, i, b)));
              }
              double own = 0.0;
              for (int p = W_LO; p < W_HI; ++p) own += xi_j[p + (size_t)P * (dd + (size_t)D * m)] * b[p];
              ph = ph + (zij * chim * xid * own);
              const double w = zij * chim * xid * ph;
              for (int p = W_LO; p < W_HI; ++p) m_1[p] += w * b[p];
            }
          }
        }
        const double f = beta_i / sigma_sq;
        for (int p = 0; p < P; ++p) m_1[p] *= f;
        for (int q = 0; q < P * P; ++q) M_1[q] *= f;
        for (int k = 0; k < P; ++k)
          M_1[k + (size_t)P * k] += tilde_tau_xi[j + (size_t)K * (m + (size_t)M * dd)] *
                                    gxi_j[k + (size_t)P * (dd + (size_t)D * m)];
        if (!singular_route(r, UPD_XI, (uint32_t)(((j * M + m) * D + dd) * P), P, M_1, m_1, draw)) {
        orc_inv(P, M_1);
        for (int p = 0; p < P; ++p) {
          double s = 0.0;
          for (int q = 0; q < P; ++q) s += M_1[p + (size_t)P * q] * m_1[q];
          mean[p] = s;
        }
        orc_mvnrnd(r, UPD_XI, (uint32_t)(((j * M + m) * D + dd) * P), P, mean, M_1, draw);
        }
        for (int p = 0; p < P; ++p) xi_j[p + (size_t)P * (dd + (size_t)D * m)] = draw[p];
      }
    }
  }
  if (iter < (T - 1))
    for (int k = 0; k < K; ++k) memcpy(SL_XI(c, iter + 1, k), SL_XI(c, iter, k), sizeof(double) * (size_t)P * D * M);
  free(m_1); free(M_1); free(mean); free(draw);
}

/* updateDeltaXi, UpdateDelta.h:76-124 ((d, k, i) order) */
void orc_updateDeltaXi(const orc_data* d, const orc_rng* r, int iter, int T, orc_chain* c) {
  DIMS;
  double* delta = SL_DXI(c, iter);
  const double* a_xi = SL_AXI(c, iter);
#define DX(k, i, dd) delta[(k) + (size_t)K * ((i) + (size_t)M * (dd))]
  for (int dd = 0; dd < D; ++dd)
    for (int k = 0; k < K; ++k) {
      const double* xi_k = SL_XI(c, iter, k);
      const double* gx_k = SL_GXI(c, iter, k);
#define XI3(j, m) xi_k[(j) + (size_t)P * (dd + (size_t)D * (m))]
#define GX3(j, m) gx_k[(j) + (size_t)P * (dd + (size_t)D * (m))]
      for (int i = 0; i < M; ++i) {
        double param1, param2, tilde_tau;
        if (i == 0) {
          param1 = a_xi[k + (size_t)K * (0 + 2 * (size_t)dd)] + ((P * M) * 0.5);
          param2 = 1;
          for (int j = 0; j < P; ++j) {
            param2 = param2 + (0.5 * GX3(j, 0) * (XI3(j, 0) * XI3(j, 0)));
            for (int m = 1; m < M; ++m) {
              tilde_tau = 1;
              for (int nn = 1; nn <= m; ++nn) tilde_tau = tilde_tau * DX(k, nn, dd);
              param2 = param2 + (0.5 * GX3(j, m) * tilde_tau * (XI3(j, m) * XI3(j, m)));
            }
          }
        } else {
          param1 = a_xi[k + (size_t)K * (1 + 2 * (size_t)dd)] + ((P * (M - i)) * 0.5);
          param2 = 1;
          for (int j = 0; j < P; ++j)
            for (int m = i; m < M; ++m) {
              tilde_tau = 1;
              for (int nn = 0; nn <= m; ++nn)
                if (nn != i) tilde_tau = tilde_tau * DX(k, nn, dd);
              param2 = param2 + (0.5 * GX3(j, m) * tilde_tau * (XI3(j, m) * XI3(j, m)));
            }
        }
        DX(k, i, dd) = orc_rgamma(r, UPD_DELTA_XI, (uint32_t)((dd * K + k) * M + i), param1, 1 / param2);
      }
#undef XI3
#undef GX3
    }
#undef DX
  if (iter < (T - 1)) memcpy(SL_DXI(c, iter + 1), delta, sizeof(double) * (size_t)K * M * D);
}

/* updateAXi, UpdateA.h:137-205 ((j, i, d) order) */
void orc_updateAXi(const orc_data* d, const orc_rng* r, int iter, int T, const orc_hyper* h, orc_chain* c) {
  DIMS;
  double* a = SL_AXI(c, iter);
  const double* delta = SL_DXI(c, iter);
  for (int j = 0; j < K; ++j)
    for (int i = 0; i < 2; ++i)
      for (int dd = 0; dd < D; ++dd) {
        double* cell = &a[j + (size_t)K * (i + 2 * (size_t)dd)];
        const double* drow = delta + j + (size_t)K * M * dd; /* delta.slice(d).row(j) */
        *cell = update_a_cell(r, UPD_AXI_PROP, UPD_AXI_ACC, (uint32_t)((j * 2 + i) * D + dd), i == 0, h,
                              *cell, drow[0], M, drow, K);
      }
  if (iter < (T - 1)) memcpy(SL_AXI(c, iter + 1), a, sizeof(double) * (size_t)K * 2 * D);
}

/* updateGammaXi, UpdateGamma.h:48-72 ((k, i=d, l=p, j=m) order) */
void orc_updateGammaXi(const orc_data* d, const orc_rng* r, int iter, int T, double nu_gamma, orc_chain* c) {
  DIMS;
  const double* delta_xi = SL_DXI(c, iter);
  for (int k = 0; k < K; ++k) {
    const double* xi_k = SL_XI(c, iter, k);
    double* gx_k = SL_GXI(c, iter, k);
    for (int i = 0; i < D; ++i)
      for (int l = 0; l < P; ++l) {
        double placeholder = 1;
        for (int j = 0; j < M; ++j) {
          placeholder = placeholder * delta_xi[k + (size_t)K * (j + (size_t)M * i)];
          const double x = xi_k[l + (size_t)P * (i + (size_t)D * j)];
          gx_k[l + (size_t)P * (i + (size_t)D * j)] =
              orc_rgamma(r, UPD_GAMMA_XI, (uint32_t)(((k * D + i) * P + l) * M + j), (nu_gamma + 1) / 2,
                         2 / (nu_gamma + placeholder * (x * x)));
        }
      }
  }
  if (iter < T - 1)
    for (int k = 0; k < K; ++k) memcpy(SL_GXI(c, iter + 1, k), SL_GXI(c, iter, k), sizeof(double) * (size_t)P * D * M);
}
