/*
 * oracle/gram.c -- TEST INFRASTRUCTURE (see oracle.h).
 *
 * The warm-start sweep of the functional model (BFMMM_MTT_warm_start, BFMMM.h:1502-1553 + loglik :1670) in
 * SUFFICIENT-STATISTICS ("Gram") form on the CPU: every data term of every full conditional is written through the
 * per-curve G_i = B_i'B_i, s_i = B_i'y_i, yy_i = y_i'y_i instead of the reference's per-observation
 * loops; priors, proposals, draws and their RNG keys are the reference-structure code of updates.c, unchanged.
 * Purpose: timed next to the reference-structure restatement it splits the GPU speed-up into its
 * algorithmic part (Gram form vs per-observation loops, both on one CPU thread) and its hardware part.
 * tests/test_oracle_gram.py checks that it reproduces the reference-structure sweep (reassociation only).
 * No covariates, no multivariate model.
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct orc_gram_s {
  int n, P;
  double* G;     /* n x P x P (row-major per curve, symmetric) */
  double* s;     /* n x P */
  double* yy;    /* n */
} orc_gram;

orc_gram* orc_gram_prepare(const orc_data* d) {
  const int n = d->n, P = d->P;
  orc_gram* g = (orc_gram*)malloc(sizeof(orc_gram));
  g->n = n; g->P = P;
  g->G = (double*)calloc((size_t)n * P * P, sizeof(double));
  g->s = (double*)calloc((size_t)n * P, sizeof(double));
  g->yy = (double*)calloc((size_t)n, sizeof(double));
  for (int i = 0; i < n; ++i) {
    double* Gi = g->G + (size_t)i * P * P;
    double* si = g->s + (size_t)i * P;
    for (int64_t l = d->off[i]; l < d->off[i + 1]; ++l) {
      const double* b = d->B + (size_t)l * P;
      const double y = d->y[l];
      for (int p = 0; p < P; ++p) {
        if (b[p] == 0) continue;
        for (int q = 0; q < P; ++q) Gi[p * P + q] += b[p] * b[q];
        si[p] += b[p] * y;
      }
      g->yy[i] += y * y;
    }
  }
  return g;
}

void orc_gram_free(orc_gram* g) {
  if (!g) return;
  free(g->G); free(g->s); free(g->yy); free(g);
}

#define GNU(c, t)   ((c)->nu + (size_t)K * P * (t))
#define GCHI(c, t)  ((c)->chi + (size_t)n * M * (t))
#define GZ(c, t)    ((c)->Z + (size_t)n * K * (t))
#define GPHI(c, t)  ((c)->Phi + (size_t)K * P * M * (t))

/* fitted coefficient of curve i: sum_k Z_ik (nu_k + sum_m chi_im phi_km) */
static void coef_of(const orc_data* d, const orc_chain* c, int iter, int i, double* out) {
  const int n = d->n, K = d->K, P = d->P, M = d->M;
  const double* nu = GNU(c, iter); const double* phi = GPHI(c, iter);
  const double* Z = GZ(c, iter); const double* chi = GCHI(c, iter);
  for (int p = 0; p < P; ++p) out[p] = 0.0;
  for (int k = 0; k < K; ++k) {
    const double z = Z[i + (size_t)n * k];
    if (z == 0) continue;
    for (int p = 0; p < P; ++p) {
      double u = nu[k + (size_t)K * p];
      for (int m = 0; m < M; ++m) u += chi[i + (size_t)n * m] * phi[k + (size_t)K * (p + (size_t)P * m)];
      out[p] += z * u;
    }
  }
}

static void matvec(int P, const double* Gi, const double* v, double* out) {
  for (int p = 0; p < P; ++p) {
    const double* row = Gi + (size_t)p * P;
    double s = 0.0;
    for (int q = 0; q < P; ++q) s += row[q] * v[q];
    out[p] = s;
  }
}

static double dotp(int P, const double* a, const double* b) {
  double s = 0.0;
  for (int p = 0; p < P; ++p) s += a[p] * b[p];
  return s;
}

static double rss_of(const orc_data* d, const orc_gram* g, const orc_chain* c, int iter) {
  const int n = d->n, P = d->P;
  double* cf = (double*)malloc(sizeof(double) * 2 * (size_t)P);
  double* gc = cf + P;
  double rss = 0.0;
  for (int i = 0; i < n; ++i) {
    coef_of(d, c, iter, i, cf);
    matvec(P, g->G + (size_t)i * P * P, cf, gc);
    rss += g->yy[i] - 2.0 * dotp(P, cf, g->s + (size_t)i * P) + dotp(P, cf, gc);
  }
  free(cf);
  return rss;
}

static double calc_lB(int K, const double* alpha) {
  double log_B = 0.0, acc = 0.0;
  for (int i = 0; i < K; ++i) { log_B = log_B + lgamma(alpha[i]); acc += alpha[i]; }
  return log_B - lgamma(acc);
}
static double Z_proposal_density(int K, const double* Z, const double* alpha) {
  double density = 0.0;
  for (int i = 0; i < K; ++i) density = density + (alpha[i] - 1) * log(Z[i]);
  return density - calc_lB(K, alpha);
}

/* updateZ_PM (UpdateMixedMembership.h:131-185): the residual sum of squares of a curve is a quadratic form in Z_i */
static void gram_updateZ(const orc_data* d, const orc_gram* g, const orc_rng* r, double beta_i, int iter, int T, double a_Z_PM,
                         orc_chain* c) {
  const int n = d->n, K = d->K, P = d->P, M = d->M;
  double* Z_t = GZ(c, iter);
  const double* nu = GNU(c, iter); const double* phi = GPHI(c, iter); const double* chi = GCHI(c, iter);
  const double* pi_t = c->pi + (size_t)K * iter;
  const double sigma_sq = c->sigma[iter], alpha_3 = c->alpha3[iter];
  double* u = (double*)malloc(sizeof(double) * 2 * (size_t)K * P);
  double* Gu = u + (size_t)K * P;
  double a[16], Q[16 * 16], Zold[16], Z_ph[16], a_old[16], a_new[16];
  for (int i = 0; i < n; ++i) {
    const double* Gi = g->G + (size_t)i * P * P;
    const double* si = g->s + (size_t)i * P;
    for (int k = 0; k < K; ++k) {
      for (int p = 0; p < P; ++p) {
        double v = nu[k + (size_t)K * p];
        for (int m = 0; m < M; ++m) v += chi[i + (size_t)n * m] * phi[k + (size_t)K * (p + (size_t)P * m)];
        u[k * P + p] = v;
      }
      matvec(P, Gi, u + k * P, Gu + k * P);
      a[k] = dotp(P, u + k * P, si);
    }
    for (int k = 0; k < K; ++k)
      for (int k2 = 0; k2 < K; ++k2) Q[k * 16 + k2] = dotp(P, u + k * P, Gu + k2 * P);
    for (int k = 0; k < K; ++k) { Zold[k] = Z_t[i + (size_t)n * k]; a_old[k] = a_Z_PM * Zold[k]; }
    orc_rdirichlet(r, UPD_Z_PROP, (uint32_t)(i * K), K, a_old, Z_ph);
    double lp[2];
    for (int w = 0; w < 2; ++w) {
      const double* Zr = w ? Z_ph : Zold;
      double lpdf = 0.0, rss = g->yy[i];
      for (int l = 0; l < K; ++l) lpdf = lpdf + ((alpha_3 * pi_t[l] - 1) * log(Zr[l]));
      for (int k = 0; k < K; ++k) {
        rss -= 2.0 * Zr[k] * a[k];
        for (int k2 = 0; k2 < K; ++k2) rss += Zr[k] * Zr[k2] * Q[k * 16 + k2];
      }
      lp[w] = lpdf - beta_i * (rss / (2 * sigma_sq));
    }
    for (int k = 0; k < K; ++k) a_new[k] = a_Z_PM * Z_ph[k];
    const double lpdf_propose_new = Z_proposal_density(K, Z_ph, a_old);
    const double lpdf_propose_old = Z_proposal_density(K, Zold, a_new);
    double acceptance_prob = lp[1] - lp[0] + lpdf_propose_old - lpdf_propose_new;
    const double rand_unif_var = orc_runif(r, UPD_Z_ACC, (uint32_t)i);
    for (int j = 0; j < K; ++j)
      if (Zold[j] <= 0) acceptance_prob = 1;
    if (log(rand_unif_var) < acceptance_prob)
      for (int k = 0; k < K; ++k) Z_t[i + (size_t)n * k] = Z_ph[k];
  }
  if (iter < (T - 1)) memcpy(GZ(c, iter + 1), Z_t, sizeof(double) * (size_t)n * K);
  free(u);
}

/* the weighted normal equations of one direction: M_1 = sum_i w_i^2 G_i,  m_1 = sum_i w_i (s_i - G_i (c_i - w_i theta)) */
static void normal_eq(const orc_data* d, const orc_gram* g, const orc_chain* c, int iter, int j, int mt, const double* theta,
                      double* m_1, double* M_1, double* cf, double* gv) {
  const int n = d->n, P = d->P, K = d->K, M = d->M;
  const double* Z = GZ(c, iter); const double* chi = GCHI(c, iter);
  memset(m_1, 0, sizeof(double) * (size_t)P);
  memset(M_1, 0, sizeof(double) * (size_t)P * P);
  for (int i = 0; i < n; ++i) {
    const double zij = Z[i + (size_t)n * j];
    if (zij == 0) continue;
    const double w = (mt == 0) ? zij : zij * chi[i + (size_t)n * (mt - 1)];
    const double* Gi = g->G + (size_t)i * P * P;
    coef_of(d, c, iter, i, cf);
    for (int p = 0; p < P; ++p) cf[p] -= w * theta[p];
    matvec(P, Gi, cf, gv);
    const double* si = g->s + (size_t)i * P;
    for (int p = 0; p < P; ++p) m_1[p] += w * (si[p] - gv[p]);
    const double w2 = w * w;
    for (int e = 0; e < P * P; ++e) M_1[e] += w2 * Gi[e];
  }
}

/* updatePhi, UpdatePhi.h:23-89 */
static void gram_updatePhi(const orc_data* d, const orc_gram* g, const orc_rng* r, double beta_i, int iter, int T,
                           const double* tilde_tau, orc_chain* c) {
  const int K = d->K, P = d->P, M = d->M;
  double* phi_t = GPHI(c, iter);
  const double* gamma_t = c->gamma + (size_t)K * P * M * iter;
  const double sigma_sq = c->sigma[iter];
  double* buf = (double*)malloc(sizeof(double) * ((size_t)P * P + 6 * (size_t)P));
  double* M_1 = buf; double* m_1 = M_1 + (size_t)P * P; double* mean = m_1 + P; double* draw = mean + P;
  double* th = draw + P; double* cf = th + P; double* gv = cf + P;
  for (int j = 0; j < K; ++j)
    for (int m = 0; m < M; ++m) {
      for (int p = 0; p < P; ++p) th[p] = phi_t[j + (size_t)K * (p + (size_t)P * m)];
      normal_eq(d, g, c, iter, j, m + 1, th, m_1, M_1, cf, gv);
      const double f = beta_i / sigma_sq;
      for (int p = 0; p < P; ++p) m_1[p] *= f;
      for (int q = 0; q < P * P; ++q) M_1[q] *= f;
      for (int k = 0; k < P; ++k)
        M_1[k + (size_t)P * k] += tilde_tau[j + (size_t)K * m] * gamma_t[j + (size_t)K * (k + (size_t)P * m)];
      orc_inv(P, M_1);
      for (int p = 0; p < P; ++p) {
        double s = 0.0;
        for (int q = 0; q < P; ++q) s += M_1[p + (size_t)P * q] * m_1[q];
        mean[p] = s;
      }
      orc_mvnrnd(r, UPD_PHI, (uint32_t)((j * M + m) * P), P, mean, M_1, draw);
      for (int p = 0; p < P; ++p) phi_t[j + (size_t)K * (p + (size_t)P * m)] = draw[p];
    }
  if (iter < (T - 1)) memcpy(GPHI(c, iter + 1), phi_t, sizeof(double) * (size_t)K * P * M);
  free(buf);
}

/* updateNu, UpdateNu.h:24-74 */
static void gram_updateNu(const orc_data* d, const orc_gram* g, const orc_rng* r, double beta_i, int iter, int T, orc_chain* c) {
  const int K = d->K, P = d->P;
  double* nu_t = GNU(c, iter);
  const double sigma = c->sigma[iter];
  double* buf = (double*)malloc(sizeof(double) * ((size_t)P * P + 6 * (size_t)P));
  double* B_1 = buf; double* b_1 = B_1 + (size_t)P * P; double* mean = b_1 + P; double* draw = mean + P;
  double* th = draw + P; double* cf = th + P; double* gv = cf + P;
  for (int j = 0; j < K; ++j) {
    for (int p = 0; p < P; ++p) th[p] = nu_t[j + (size_t)K * p];
    normal_eq(d, g, c, iter, j, 0, th, b_1, B_1, cf, gv);
    const double f = beta_i / sigma;
    for (int p = 0; p < P; ++p) b_1[p] *= f;
    for (int q = 0; q < P * P; ++q) B_1[q] *= f;
    const double tau_j = c->tau[iter + (size_t)T * j];
    for (int q = 0; q < P * P; ++q) B_1[q] += tau_j * d->Pmat[q];
    orc_pinv_sym(P, B_1);
    for (int q = 0; q < P; ++q)
      for (int p = 0; p < q; ++p) {
        const double s = (B_1[p + (size_t)P * q] + B_1[q + (size_t)P * p]) / 2;
        B_1[p + (size_t)P * q] = s;
        B_1[q + (size_t)P * p] = s;
      }
    for (int p = 0; p < P; ++p) {
      double s = 0.0;
      for (int q = 0; q < P; ++q) s += B_1[p + (size_t)P * q] * b_1[q];
      mean[p] = s;
    }
    orc_mvnrnd(r, UPD_NU, (uint32_t)(j * P), P, mean, B_1, draw);
    for (int p = 0; p < P; ++p) nu_t[j + (size_t)K * p] = draw[p];
  }
  if (iter < (T - 1)) memcpy(GNU(c, iter + 1), nu_t, sizeof(double) * (size_t)K * P);
  free(buf);
}

/* updateSigma, UpdateSigma.h:22-58 (integer division n_i / 2, :49) */
static void gram_updateSigma(const orc_data* d, const orc_gram* g, const orc_rng* r, int iter, int T, double alpha_0, double beta_0,
                             orc_chain* c) {
  double a = 0;
  for (int i = 0; i < d->n; ++i) a = a + ((int)(d->off[i + 1] - d->off[i]) / 2);
  const double b_1 = 0.5 * rss_of(d, g, c, iter) + beta_0;
  a = a + alpha_0;
  c->sigma[iter] = 1 / orc_rgamma(r, UPD_SIGMA, 0, a, 1 / b_1);
  if (iter < (T - 1)) c->sigma[iter + 1] = c->sigma[iter];
}

/* updateChi, UpdateChi.h:19-64 */
static void gram_updateChi(const orc_data* d, const orc_gram* g, const orc_rng* r, double beta_i, int iter, int T, orc_chain* c) {
  const int n = d->n, K = d->K, P = d->P, M = d->M;
  const double* phi = GPHI(c, iter); const double* Z = GZ(c, iter);
  double* chi_t = GCHI(c, iter);
  const double sigma = c->sigma[iter];
  double* buf = (double*)malloc(sizeof(double) * 4 * (size_t)P);
  double* um = buf; double* Gum = um + P; double* cf = Gum + P; double* gc = cf + P;
  for (int i = 0; i < n; ++i) {
    const double* Gi = g->G + (size_t)i * P * P;
    const double* si = g->s + (size_t)i * P;
    for (int m = 0; m < M; ++m) {
      for (int p = 0; p < P; ++p) {
        double v = 0.0;
        for (int k = 0; k < K; ++k) v += Z[i + (size_t)n * k] * phi[k + (size_t)K * (p + (size_t)P * m)];
        um[p] = v;
      }
      matvec(P, Gi, um, Gum);
      coef_of(d, c, iter, i, cf);
      const double chim = chi_t[i + (size_t)n * m];
      for (int p = 0; p < P; ++p) cf[p] -= chim * um[p];
      double w = dotp(P, um, si) - dotp(P, Gum, cf);
      double W = dotp(P, um, Gum);
      w = (w * beta_i) / sigma;
      W = 1 + ((W * beta_i) / sigma);
      W = 1 / W;
      chi_t[i + (size_t)n * m] = W * w + sqrt(W) * orc_rnorm(r, UPD_CHI, (uint32_t)(i * M + m));
    }
  }
  (void)gc;
  if (iter < (T - 1)) memcpy(GCHI(c, iter + 1), chi_t, sizeof(double) * (size_t)n * M);
  free(buf);
}

void orc_gram_run_warm(const orc_data* d, const orc_gram* g, const orc_hyper* h, uint64_t seed, uint32_t chain, int T,
                       int first_iter, int n_iter, orc_chain* c) {
  const int K = d->K, M = d->M;
  double* tilde_tau = (double*)malloc(sizeof(double) * (size_t)K * M);
  const int64_t N = d->off[d->n];
  for (int i = first_iter; i < first_iter + n_iter && i < T; ++i) {
    orc_rng r = {seed, chain, (uint32_t)i, 0};
    gram_updateZ(d, g, &r, 1.0, i, T, h->a_Z_PM, c);
    orc_updatePi_PM(d, &r, i, T, h->c, h->a_pi_PM, c);
    orc_updateAlpha3(d, &r, i, T, h->b, h->var_alpha3, c);
    orc_tilde_tau(K, M, c->delta + (size_t)K * M * i, tilde_tau);
    gram_updatePhi(d, g, &r, 1.0, i, T, tilde_tau, c);
    orc_updateDelta(d, &r, i, T, c);
    orc_updateA(d, &r, i, T, h, c);
    orc_updateGamma(d, &r, i, T, h->nu_1, c);
    gram_updateNu(d, g, &r, 1.0, i, T, c);
    orc_updateTau(d, &r, i, T, h->alpha_nu, h->beta_nu, c);
    gram_updateSigma(d, g, &r, i, T, h->alpha_0, h->beta_0, c);
    gram_updateChi(d, g, &r, 1.0, i, T, c);
    /* calcLikelihood, CalculateLikelihood.h:19-44: sum of dnorm(y; mean, sqrt(sigma), log) */
    const double sigma = c->sigma[i];
    c->loglik[i] = -(double)N * (0.91893853320467274178 + log(sqrt(sigma))) - rss_of(d, g, c, i) / (2.0 * sigma);
  }
  free(tilde_tau);
}

/* The Nu_Z sweep of BFMMM_Nu_Z (BFMMM.h:1073-1107 + loglik :1106) in the same sufficient-statistics form: Z, pi, alpha_3, nu,
 * tau, sigma^2; Phi, chi, gamma, delta, A are carried (with Phi = 0, chi = 0 in the multi-try entry point). */
static void gcarry(double* base, size_t len, int iter, int T) {
  if (iter < T - 1) memcpy(base + len * (size_t)(iter + 1), base + len * (size_t)iter, sizeof(double) * len);
}

void orc_gram_run_nu_z(const orc_data* d, const orc_gram* g, const orc_hyper* h, uint64_t seed, uint32_t chain, int T,
                       int first_iter, int n_iter, orc_chain* c) {
  const int n = d->n, K = d->K, P = d->P, M = d->M;
  const int64_t N = d->off[d->n];
  for (int i = first_iter; i < first_iter + n_iter && i < T; ++i) {
    orc_rng r = {seed, chain, (uint32_t)i, 0};
    gram_updateZ(d, g, &r, 1.0, i, T, h->a_Z_PM, c);
    orc_updatePi_PM(d, &r, i, T, h->c, h->a_pi_PM, c);
    orc_updateAlpha3(d, &r, i, T, h->b, h->var_alpha3, c);
    gram_updateNu(d, g, &r, 1.0, i, T, c);
    orc_updateTau(d, &r, i, T, h->alpha_nu, h->beta_nu, c);
    gram_updateSigma(d, g, &r, i, T, h->alpha_0, h->beta_0, c);
    gcarry(c->chi, (size_t)n * M, i, T);
    gcarry(c->Phi, (size_t)K * P * M, i, T);
    gcarry(c->gamma, (size_t)K * P * M, i, T);
    gcarry(c->delta, (size_t)K * M, i, T);
    gcarry(c->A, (size_t)K * 2, i, T);
    const double sigma = c->sigma[i];
    c->loglik[i] = -(double)N * (0.91893853320467274178 + log(sqrt(sigma))) - rss_of(d, g, c, i) / (2.0 * sigma);
  }
}
