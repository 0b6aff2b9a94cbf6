/*
 * oracle/drivers.c -- TEST INFRASTRUCTURE (see oracle.h).
 *
 * Sweep orders and initial states of the chain drivers in inst/include/BayesFMMM/BFMMM.h,
 * untempered part (n_temp_trans == 0, the reference default: UserFunctions.cpp:1533-1535
 * turns tempered transitions off).  Full chain kept in memory (r_stored_iters == T).
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

void orc_rdirichlet(const orc_rng* r, uint32_t upd, uint32_t idx0, int K, const double* alpha, double* out);

/* BFMMM_Nu_Z initial state, BFMMM.h:1039-1071 (Cov_Adj: :3690-3730):
 *   nu ~ randn, chi = 0, pi(0) ~ Dir(c), Z(0) rows ~ Dir(100 pi(0)), sigma = alpha_3 = tau =
 *   delta = A = gamma = 1, Phi = 0.  (Only slot 0 matters: every later slot is overwritten
 *   by the "copy to iter+1" carry before it is read.) */
void orc_init_nu_z(const orc_data* d, const orc_hyper* h, uint64_t seed, uint32_t chain, orc_chain* c) {
  const int n = d->n, K = d->K, P = d->P, M = d->M, D = d->D;
  orc_rng r = {seed, chain, 0, 0};
  for (int q = 0; q < K * P; ++q) c->nu[q] = orc_rnorm(&r, UPD_INIT_NU, (uint32_t)q);
  memset(c->chi, 0, sizeof(double) * (size_t)n * M);
  orc_rdirichlet(&r, UPD_INIT_PI, 0, K, h->c, c->pi);
  double a[16], z[16];
  for (int k = 0; k < K; ++k) a[k] = c->pi[k] * 100;
  for (int i = 0; i < n; ++i) {
    orc_rdirichlet(&r, UPD_INIT_Z, (uint32_t)(i * K), K, a, z);
    for (int k = 0; k < K; ++k) c->Z[i + (size_t)n * k] = z[k];
  }
  c->sigma[0] = 1; c->alpha3[0] = 1;
  for (int k = 0; k < K; ++k) c->tau[0 + (size_t)c->T * k] = 1;
  for (int q = 0; q < K * M; ++q) c->delta[q] = 1;
  for (int q = 0; q < K * 2; ++q) c->A[q] = 1;
  for (int q = 0; q < K * P * M; ++q) { c->gamma[q] = 1; c->Phi[q] = 0; }
  if (D > 0) {
    memset(c->eta, 0, sizeof(double) * (size_t)P * D * K);
    for (int q = 0; q < K * D; ++q) c->tau_eta[q] = 1;
    memset(c->xi, 0, sizeof(double) * (size_t)K * P * D * M);
    for (int q = 0; q < K * P * D * M; ++q) c->gamma_xi[q] = 1;
    for (int q = 0; q < K * M * D; ++q) c->delta_xi[q] = 1;
    for (int q = 0; q < K * 2 * D; ++q) c->A_xi[q] = 1;
  }
}

/* BFMMM_Theta initial state, BFMMM.h:1210-1250 (Cov_Adj :3884-3942): as Nu_Z but
 * chi ~ randn, Phi ~ randn and Z, nu (eta) pinned to the supplied estimates. */
void orc_init_theta(const orc_data* d, const orc_hyper* h, uint64_t seed, uint32_t chain,
                    const double* Z_est, const double* nu_est, const double* eta_est, orc_chain* c) {
  const int n = d->n, K = d->K, P = d->P, M = d->M, D = d->D;
  orc_init_nu_z(d, h, seed, chain, c);
  orc_rng r = {seed, chain, 0, 0};
  for (int q = 0; q < n * M; ++q) c->chi[q] = orc_rnorm(&r, UPD_INIT_CHI, (uint32_t)q);
  for (int q = 0; q < K * P * M; ++q) c->Phi[q] = orc_rnorm(&r, UPD_INIT_PHI, (uint32_t)q);
  memcpy(c->Z, Z_est, sizeof(double) * (size_t)n * K);
  memcpy(c->nu, nu_est, sizeof(double) * (size_t)K * P);
  if (D > 0 && eta_est) memcpy(c->eta, eta_est, sizeof(double) * (size_t)P * D * K);
}

/* carry a block that the sweep does not update, so that slot iter+1 starts from slot iter
 * (the reference gets this for free because it pre-fills every slot, e.g. BFMMM.h:1247-1250) */
static void carry(double* base, size_t len, int iter, int T) {
  if (iter < T - 1) memcpy(base + len * (size_t)(iter + 1), base + len * (size_t)iter, sizeof(double) * len);
}

void orc_run_sweeps(const orc_data* d, const orc_hyper* h, uint64_t seed, uint32_t chain,
                    int sweep, int covariance_adj, int T, int first_iter, int n_iter, orc_chain* c) {
  const int n = d->n, K = d->K, P = d->P, M = d->M, D = d->D;
  double* tilde_tau = (double*)malloc(sizeof(double) * (size_t)K * M);
  double* tilde_tau_xi = (double*)malloc(sizeof(double) * (size_t)K * M * (D > 0 ? D : 1));
  for (int i = first_iter; i < first_iter + n_iter && i < T; ++i) {
    orc_rng r = {seed, chain, (uint32_t)i, 0};
    if (sweep == ORC_SWEEP_NU_Z) {
      /* BFMMM.h:1073-1107; covariate version :3741-3780 */
      orc_updateZ_PM(d, &r, 1.0, i, T, h->a_Z_PM, c);
      orc_updatePi_PM(d, &r, i, T, h->c, h->a_pi_PM, c);
      orc_updateAlpha3(d, &r, i, T, h->b, h->var_alpha3, c);
      orc_updateNu(d, &r, 1.0, i, T, c);
      orc_updateTau(d, &r, i, T, h->alpha_nu, h->beta_nu, c);
      orc_updateSigma(d, &r, 1.0, 0, i, T, h->alpha_0, h->beta_0, c);
      if (D > 0) {
        orc_updateEta(d, &r, 1.0, i, T, c);
        orc_updateTauEta(d, &r, i, T, h->alpha_eta, h->beta_eta, c);
        carry(c->xi, (size_t)K * P * D * M, i, T);
      }
      /* blocks the Nu_Z sweep never updates */
      carry(c->chi, (size_t)n * M, i, T);
      carry(c->Phi, (size_t)K * P * M, i, T);
      carry(c->gamma, (size_t)K * P * M, i, T);
      carry(c->delta, (size_t)K * M, i, T);
      carry(c->A, (size_t)K * 2, i, T);
    } else if (sweep == ORC_SWEEP_THETA) {
      /* BFMMM.h:1253-1292; covariate version :3944-4010 */
      orc_tilde_tau(K, M, c->delta + (size_t)K * M * i, tilde_tau);
      orc_updatePhi(d, &r, 1.0, i, T, tilde_tau, c);
      orc_updateDelta(d, &r, i, T, c);
      orc_updateA(d, &r, i, T, h, c);
      orc_updateGamma(d, &r, i, T, h->nu_1, c);
      orc_updateTau(d, &r, i, T, h->alpha_nu, h->beta_nu, c);
      orc_updateSigma(d, &r, 1.0, 0, i, T, h->alpha_0, h->beta_0, c);
      orc_updateChi(d, &r, 1.0, i, T, c);
      if (D > 0) {
        orc_updateTauEta(d, &r, i, T, h->alpha_eta, h->beta_eta, c);
        if (covariance_adj) {
          orc_tilde_tau_xi(K, M, D, c->delta_xi + (size_t)K * M * D * i, tilde_tau_xi);
          orc_updateXi(d, &r, 1.0, i, T, tilde_tau_xi, c);
          orc_updateDeltaXi(d, &r, i, T, c);
          orc_updateAXi(d, &r, i, T, h, c);
          orc_updateGammaXi(d, &r, i, T, h->nu_1, c);
        } else {
          carry(c->xi, (size_t)K * P * D * M, i, T);
        }
        carry(c->eta, (size_t)P * D * K, i, T);
      }
      carry(c->Z, (size_t)n * K, i, T);
      carry(c->nu, (size_t)K * P, i, T);
      carry(c->pi, (size_t)K, i, T);
      if (i < T - 1) c->alpha3[i + 1] = c->alpha3[i];
    } else {
      /* BFMMM_MTT_warm_start untempered sweep, BFMMM.h:1502-1553;
       * MeanAdj :4248-4312, Mean_CovAdj :4809-4894 */
      orc_updateZ_PM(d, &r, 1.0, i, T, h->a_Z_PM, c);
      orc_updatePi_PM(d, &r, i, T, h->c, h->a_pi_PM, c);
      orc_updateAlpha3(d, &r, i, T, h->b, h->var_alpha3, c);
      orc_tilde_tau(K, M, c->delta + (size_t)K * M * i, tilde_tau);
      orc_updatePhi(d, &r, 1.0, i, T, tilde_tau, c);
      orc_updateDelta(d, &r, i, T, c);
      orc_updateA(d, &r, i, T, h, c);
      orc_updateGamma(d, &r, i, T, h->nu_1, c);
      orc_updateNu(d, &r, 1.0, i, T, c);
      orc_updateTau(d, &r, i, T, h->alpha_nu, h->beta_nu, c);
      orc_updateSigma(d, &r, 1.0, 0, i, T, h->alpha_0, h->beta_0, c);
      orc_updateChi(d, &r, 1.0, i, T, c);
      if (D > 0) {
        orc_updateEta(d, &r, 1.0, i, T, c);
        orc_updateTauEta(d, &r, i, T, h->alpha_eta, h->beta_eta, c);
        if (covariance_adj) {
          orc_tilde_tau_xi(K, M, D, c->delta_xi + (size_t)K * M * D * i, tilde_tau_xi);
          orc_updateXi(d, &r, 1.0, i, T, tilde_tau_xi, c);
          orc_updateDeltaXi(d, &r, i, T, c);
          orc_updateAXi(d, &r, i, T, h, c);
          orc_updateGammaXi(d, &r, i, T, h->nu_1, c);
        } else {
          carry(c->xi, (size_t)K * P * D * M, i, T);
        }
      }
    }
    /* BFMMM.h:1106 / :1291 / :1670 */
    c->loglik[i] = orc_calcLikelihood(d, i, c);
  }
  free(tilde_tau);
  free(tilde_tau_xi);
}

/* ==============================================================================================
 * Tempered transitions of BFMMM_MTT_warm_start (functional model, no covariates).
 * ============================================================================================== */

/* Geometric ladder, BFMMM.h:1452-1460.  (As in the reference the loop overwrites the last rung that was first
 * set to beta_N_t: beta_ladder(i) = geom_mult^i with geom_mult = beta_N_t^(1/N_t).) */
void orc_beta_ladder(int N_t, double beta_N_t, double* ladder) {
  for (int i = 0; i < N_t; ++i) ladder[i] = 1.0;
  ladder[N_t - 1] = beta_N_t;
  const double geom_mult = pow(beta_N_t, 1.0 / N_t);
  for (int i = 1; i < N_t; ++i) ladder[i] = ladder[i - 1] * geom_mult;
}

/* calculatePZeta, CalculateTTAcceptance.h:22-51 (state = slot `iter` of the TT arrays) */
double orc_calculatePZeta(const orc_data* d, double beta_i, int iter, const orc_chain* c) {
  if (d->D > 0) return orc_calculatePZetaCov(d, beta_i, iter, c);     /* CalculateTTAcceptance.h:195, :296 */
  const int n = d->n, K = d->K, P = d->P, M = d->M;
  const double* nu = c->nu + (size_t)K * P * iter;
  const double* Phi = c->Phi + (size_t)K * P * M * iter;
  const double* Z = c->Z + (size_t)n * K * iter;
  const double* chi = c->chi + (size_t)n * M * iter;
  const double sigma = c->sigma[iter];
  double logAcceptance = 0;
  if (d->mv) {   /* calculatePZetaMV, CalculateTTAcceptance.h:109-133 (row i of Y = the P observations of "curve" i) */
    double* mean = (double*)malloc(sizeof(double) * (size_t)P);
    for (int i = 0; i < n; ++i) {
      for (int p = 0; p < P; ++p) mean[p] = 0;
      for (int k = 0; k < K; ++k) {
        if (Z[i + (size_t)n * k] != 0) {
          for (int p = 0; p < P; ++p) mean[p] = mean[p] + Z[i + (size_t)n * k] * nu[k + (size_t)K * p];
          for (int m = 0; m < M; ++m)
            for (int p = 0; p < P; ++p)
              mean[p] = mean[p] + Z[i + (size_t)n * k] * chi[i + (size_t)n * m] * Phi[k + (size_t)K * (p + (size_t)P * m)];
        }
      }
      double ss = 0;
      for (int p = 0; p < P; ++p) { const double r = d->y[d->off[i] + p] - mean[p]; ss += r * r; }
      logAcceptance = logAcceptance + ((-(beta_i / 2) * log(sigma) * P) - (beta_i / (2 * sigma)) * ss);
    }
    free(mean);
    return logAcceptance;
  }
  for (int i = 0; i < n; ++i) {
    const int64_t o = d->off[i], ni = d->off[i + 1] - o;
    for (int64_t l = 0; l < ni; ++l) {
      const double* b = d->B + (size_t)(o + l) * P;
      double mean = 0;
      for (int k = 0; k < K; ++k) {
        if (Z[i + (size_t)n * k] != 0) {
          double dt = 0;
          for (int p = 0; p < P; ++p) dt += nu[k + (size_t)K * p] * b[p];
          mean = mean + Z[i + (size_t)n * k] * dt;
          for (int m = 0; m < M; ++m) {
            double dp = 0;
            for (int p = 0; p < P; ++p) dp += Phi[k + (size_t)K * (p + (size_t)P * m)] * b[p];
            mean = mean + Z[i + (size_t)n * k] * chi[i + (size_t)n * m] * dp;
          }
        }
      }
      const double res = d->y[o + l] - mean;
      logAcceptance = logAcceptance + ((-(beta_i / 2) * log(sigma)) - (beta_i / (2 * sigma)) * (res * res));
    }
  }
  return logAcceptance;
}

/* CalculateTTAcceptance, CalculateTTAcceptance.h:64-97: `tt` holds the 2 N_t + 1 tempered states */
double orc_CalculateTTAcceptance(const orc_data* d, int N_t, const double* beta, const orc_chain* tt) {
  double logAcceptance = 0;
  const int m = tt->T - 1;
  for (int i = 0; i < N_t - 1; ++i) {
    logAcceptance = logAcceptance + orc_calculatePZeta(d, beta[i + 1], i, tt);        /* heating up */
    logAcceptance = logAcceptance - orc_calculatePZeta(d, beta[i], i, tt);
    logAcceptance = logAcceptance - orc_calculatePZeta(d, beta[i + 1], m - i, tt);    /* cooling down */
    logAcceptance = logAcceptance + orc_calculatePZeta(d, beta[i], m - i, tt);
  }
  return logAcceptance;
}

static double* dalloc0(size_t n) { return (double*)calloc(n ? n : 1, sizeof(double)); }

/* The tempered-transition block of iteration i, BFMMM.h:1556-1657.  On return slot i of `c` holds the accepted or
 * the original state and slot i+1 is re-initialised from it (every block but gamma, as in the reference).
 * Keyed RNG: (seed, chain, iteration i, tt_step l) for the tempered sweeps, (.., tt_step 0, UPD_TT_ACC) for the test. */
void orc_tt_block(const orc_data* d, const orc_hyper* h, uint64_t seed, uint32_t chain, int i, int N_t,
                  double beta_N_t, orc_chain* c, double* logA_out, int* accepted_out) {
  orc_tt_block_cov(d, h, seed, chain, i, N_t, beta_N_t, 0, c, logA_out, accepted_out);
}

/* covariate-adjusted drivers: BFMMM.h:4313-4440 (MeanAdj), :4897-5084 (Mean_CovAdj); MV :5880-6050, :6420-6590.  The
 * tempered sweep adds eta, tau_eta (and, with covariance adjustment, Xi, delta_xi, A_xi, gamma_xi) after chi. */
void orc_tt_block_cov(const orc_data* d, const orc_hyper* h, uint64_t seed, uint32_t chain, int i, int N_t,
                      double beta_N_t, int covariance_adj, orc_chain* c, double* logA_out, int* accepted_out) {
  const int n = d->n, K = d->K, P = d->P, M = d->M, T = c->T, D = d->D;
  const int L = 2 * N_t + 1;
  double* ladder = dalloc0((size_t)N_t);
  orc_beta_ladder(N_t, beta_N_t, ladder);
  orc_chain tt;
  memset(&tt, 0, sizeof tt);
  tt.T = L;
  const size_t s_nu = (size_t)K * P, s_chi = (size_t)n * M, s_Z = (size_t)n * K, s_pi = K, s_dl = (size_t)K * M,
               s_A = (size_t)K * 2, s_g = (size_t)K * P * M;
  tt.nu = dalloc0(s_nu * L); tt.chi = dalloc0(s_chi * L); tt.Z = dalloc0(s_Z * L); tt.pi = dalloc0(s_pi * L);
  tt.alpha3 = dalloc0(L); tt.delta = dalloc0(s_dl * L); tt.A = dalloc0(s_A * L); tt.sigma = dalloc0(L);
  tt.tau = dalloc0((size_t)L * K); tt.gamma = dalloc0(s_g * L); tt.Phi = dalloc0(s_g * L); tt.loglik = dalloc0(L);
  const size_t s_eta = (size_t)P * D * K, s_te = (size_t)K * D, s_xi = (size_t)K * P * D * M, s_dx = (size_t)K * M * D,
               s_ax = (size_t)K * 2 * D;
  if (D > 0) {
    tt.eta = dalloc0(s_eta * L); tt.tau_eta = dalloc0(s_te * L); tt.xi = dalloc0(s_xi * L); tt.gamma_xi = dalloc0(s_xi * L);
    tt.delta_xi = dalloc0(s_dx * L); tt.A_xi = dalloc0(s_ax * L);
  }
  /* initialize placeholders: slots 0 and 1 <- slot i (BFMMM.h:1557-1581) */
  for (int s = 0; s < 2; ++s) {
    memcpy(tt.nu + s_nu * s, c->nu + s_nu * i, sizeof(double) * s_nu);
    memcpy(tt.chi + s_chi * s, c->chi + s_chi * i, sizeof(double) * s_chi);
    memcpy(tt.pi + s_pi * s, c->pi + s_pi * i, sizeof(double) * s_pi);
    tt.sigma[s] = c->sigma[i];
    memcpy(tt.Z + s_Z * s, c->Z + s_Z * i, sizeof(double) * s_Z);
    memcpy(tt.delta + s_dl * s, c->delta + s_dl * i, sizeof(double) * s_dl);
    memcpy(tt.gamma + s_g * s, c->gamma + s_g * i, sizeof(double) * s_g);
    memcpy(tt.Phi + s_g * s, c->Phi + s_g * i, sizeof(double) * s_g);
    memcpy(tt.A + s_A * s, c->A + s_A * i, sizeof(double) * s_A);
    for (int k = 0; k < K; ++k) tt.tau[s + (size_t)L * k] = c->tau[i + (size_t)T * k];
    tt.alpha3[s] = c->alpha3[i];
    if (D > 0) {              /* BFMMM.h:4912-4940 */
      memcpy(tt.eta + s_eta * s, c->eta + s_eta * i, sizeof(double) * s_eta);
      memcpy(tt.tau_eta + s_te * s, c->tau_eta + s_te * i, sizeof(double) * s_te);
      memcpy(tt.xi + s_xi * s, c->xi + s_xi * i, sizeof(double) * s_xi);
      memcpy(tt.gamma_xi + s_xi * s, c->gamma_xi + s_xi * i, sizeof(double) * s_xi);
      memcpy(tt.delta_xi + s_dx * s, c->delta_xi + s_dx * i, sizeof(double) * s_dx);
      memcpy(tt.A_xi + s_ax * s, c->A_xi + s_ax * i, sizeof(double) * s_ax);
    }
  }
  double* tilde_tau = dalloc0((size_t)K * M);
  double* tilde_tau_xi = dalloc0((size_t)K * M * (D > 0 ? D : 1));
  int temp_ind = 0;
  for (int l = 1; l < L; ++l) {              /* BFMMM.h:1586-1634 */
    orc_rng r = {seed, chain, (uint32_t)i, (uint32_t)l};
    const double bl = ladder[temp_ind];
    orc_updateZ_PM(d, &r, bl, l, L, h->a_Z_PM, &tt);
    orc_updatePi_PM(d, &r, l, L, h->c, h->a_pi_PM, &tt);
    orc_updateAlpha3(d, &r, l, L, h->b, h->var_alpha3, &tt);
    orc_tilde_tau(K, M, tt.delta + s_dl * l, tilde_tau);
    orc_updatePhi(d, &r, bl, l, L, tilde_tau, &tt);
    orc_updateDelta(d, &r, l, L, &tt);
    orc_updateA(d, &r, l, L, h, &tt);
    orc_updateGamma(d, &r, l, L, h->nu_1, &tt);
    orc_updateNu(d, &r, bl, l, L, &tt);
    orc_updateTau(d, &r, l, L, h->alpha_nu, h->beta_nu, &tt);
    orc_updateSigma(d, &r, bl, 1, l, L, h->alpha_0, h->beta_0, &tt);
    orc_updateChi(d, &r, bl, l, L, &tt);
    if (D > 0) {              /* BFMMM.h:4991-5020 */
      orc_updateEta(d, &r, bl, l, L, &tt);
      orc_updateTauEta(d, &r, l, L, h->alpha_eta, h->beta_eta, &tt);
      if (covariance_adj) {
        orc_tilde_tau_xi(K, M, D, tt.delta_xi + s_dx * l, tilde_tau_xi);
        orc_updateXi(d, &r, bl, l, L, tilde_tau_xi, &tt);
        orc_updateDeltaXi(d, &r, l, L, &tt);
        orc_updateAXi(d, &r, l, L, h, &tt);
        orc_updateGammaXi(d, &r, l, L, h->nu_1, &tt);
      } else if (l + 1 < L) {
        memcpy(tt.xi + s_xi * (l + 1), tt.xi + s_xi * l, sizeof(double) * s_xi);
      }
    }
    if (l < N_t) temp_ind = temp_ind + 1;
    if (l > N_t) temp_ind = temp_ind - 1;
  }
  const double logA = orc_CalculateTTAcceptance(d, N_t, ladder, &tt);
  orc_rng r0 = {seed, chain, (uint32_t)i, 0};
  const double logu = log(orc_runif(&r0, UPD_TT_ACC, 0));
  int accepted = 0;
  if (logu < logA) {                         /* BFMMM.h:1641-1657 */
    const int f = L - 1;
    accepted = 1;
    memcpy(c->nu + s_nu * i, tt.nu + s_nu * f, sizeof(double) * s_nu);
    memcpy(c->chi + s_chi * i, tt.chi + s_chi * f, sizeof(double) * s_chi);
    memcpy(c->pi + s_pi * i, tt.pi + s_pi * f, sizeof(double) * s_pi);
    c->sigma[i] = tt.sigma[f];
    memcpy(c->Z + s_Z * i, tt.Z + s_Z * f, sizeof(double) * s_Z);
    memcpy(c->delta + s_dl * i, tt.delta + s_dl * f, sizeof(double) * s_dl);
    memcpy(c->gamma + s_g * i, tt.gamma + s_g * f, sizeof(double) * s_g);
    memcpy(c->Phi + s_g * i, tt.Phi + s_g * f, sizeof(double) * s_g);
    memcpy(c->A + s_A * i, tt.A + s_A * f, sizeof(double) * s_A);
    for (int k = 0; k < K; ++k) c->tau[i + (size_t)T * k] = tt.tau[f + (size_t)L * k];
    c->alpha3[i] = tt.alpha3[f];
    if (D > 0) {              /* BFMMM.h:5043-5052 */
      memcpy(c->eta + s_eta * i, tt.eta + s_eta * f, sizeof(double) * s_eta);
      memcpy(c->tau_eta + s_te * i, tt.tau_eta + s_te * f, sizeof(double) * s_te);
      memcpy(c->delta_xi + s_dx * i, tt.delta_xi + s_dx * f, sizeof(double) * s_dx);
      memcpy(c->A_xi + s_ax * i, tt.A_xi + s_ax * f, sizeof(double) * s_ax);
      memcpy(c->xi + s_xi * i, tt.xi + s_xi * f, sizeof(double) * s_xi);
      memcpy(c->gamma_xi + s_xi * i, tt.gamma_xi + s_xi * f, sizeof(double) * s_xi);
    }
  }
  /* initialize next state (BFMMM.h:1660-1671): every block except gamma */
  if (i + 1 < T) {
    memcpy(c->nu + s_nu * (i + 1), c->nu + s_nu * i, sizeof(double) * s_nu);
    memcpy(c->chi + s_chi * (i + 1), c->chi + s_chi * i, sizeof(double) * s_chi);
    memcpy(c->pi + s_pi * (i + 1), c->pi + s_pi * i, sizeof(double) * s_pi);
    c->sigma[i + 1] = c->sigma[i];
    memcpy(c->Z + s_Z * (i + 1), c->Z + s_Z * i, sizeof(double) * s_Z);
    memcpy(c->delta + s_dl * (i + 1), c->delta + s_dl * i, sizeof(double) * s_dl);
    memcpy(c->A + s_A * (i + 1), c->A + s_A * i, sizeof(double) * s_A);
    for (int k = 0; k < K; ++k) c->tau[(i + 1) + (size_t)T * k] = c->tau[i + (size_t)T * k];
    memcpy(c->Phi + s_g * (i + 1), c->Phi + s_g * i, sizeof(double) * s_g);
    c->alpha3[i + 1] = c->alpha3[i];
    if (D > 0) {              /* BFMMM.h:5070-5079: here gamma_xi IS carried (gamma is not) */
      memcpy(c->eta + s_eta * (i + 1), c->eta + s_eta * i, sizeof(double) * s_eta);
      memcpy(c->tau_eta + s_te * (i + 1), c->tau_eta + s_te * i, sizeof(double) * s_te);
      memcpy(c->delta_xi + s_dx * (i + 1), c->delta_xi + s_dx * i, sizeof(double) * s_dx);
      memcpy(c->A_xi + s_ax * (i + 1), c->A_xi + s_ax * i, sizeof(double) * s_ax);
      memcpy(c->xi + s_xi * (i + 1), c->xi + s_xi * i, sizeof(double) * s_xi);
      memcpy(c->gamma_xi + s_xi * (i + 1), c->gamma_xi + s_xi * i, sizeof(double) * s_xi);
    }
  }
  if (logA_out) *logA_out = logA;
  if (accepted_out) *accepted_out = accepted;
  free(tilde_tau); free(tilde_tau_xi); free(ladder);
  free(tt.eta); free(tt.tau_eta); free(tt.xi); free(tt.gamma_xi); free(tt.delta_xi); free(tt.A_xi);
  free(tt.nu); free(tt.chi); free(tt.Z); free(tt.pi); free(tt.alpha3); free(tt.delta); free(tt.A); free(tt.sigma);
  free(tt.tau); free(tt.gamma); free(tt.Phi); free(tt.loglik);
}

/* BFMMM_MTT_warm_start with n_temp_trans > 0 (BFMMM.h:1502-1672): the untempered sweep of iteration i, then -- when
 * i % n_temp_trans == 0 and i > 0 -- the tempered-transition block, then loglik(i).  logA / accepted: one entry per
 * iteration (NaN / -1 where no block ran). */
void orc_run_warm_tt(const orc_data* d, const orc_hyper* h, uint64_t seed, uint32_t chain, int T, int first_iter,
                     int n_iter, int N_t, int n_temp_trans, double beta_N_t, orc_chain* c, double* logA, int* accepted) {
  orc_run_warm_tt_cov(d, h, seed, chain, T, first_iter, n_iter, N_t, n_temp_trans, beta_N_t, 0, c, logA, accepted);
}

void orc_run_warm_tt_cov(const orc_data* d, const orc_hyper* h, uint64_t seed, uint32_t chain, int T, int first_iter,
                         int n_iter, int N_t, int n_temp_trans, double beta_N_t, int covariance_adj, orc_chain* c,
                         double* logA, int* accepted) {
  for (int i = first_iter; i < first_iter + n_iter && i < T; ++i) {
    orc_run_sweeps(d, h, seed, chain, ORC_SWEEP_WARM, covariance_adj, T, i, 1, c);
    if (logA) logA[i] = NAN;
    if (accepted) accepted[i] = -1;
    if (n_temp_trans > 0 && (i % n_temp_trans) == 0 && i > 0) {
      double la; int acc;
      orc_tt_block_cov(d, h, seed, chain, i, N_t, beta_N_t, covariance_adj, c, &la, &acc);
      if (logA) logA[i] = la;
      if (accepted) accepted[i] = acc;
      c->loglik[i] = orc_calcLikelihood(d, i, c);       /* BFMMM.h:1670 (after the block) */
    }
  }
}
