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
