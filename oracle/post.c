/* TEST INFRASTRUCTURE ONLY (see oracle.h): CPU restatement of the likelihood-based post-processing functions of the
 * functional model, over a chain whose T slots hold the saved draws (the concatenated on-disk batches).
 *
 *   orc_post_llik  FLLik  src/PostProcessing.cpp:5106-5111   LLik(i) = calcLikelihoodCovariateAdj(draw i)
 *   orc_post_dic   FDIC   :3835-3853 (no covariates), :3922-3944 (covariates)
 *   orc_post_aic   FAIC   :4142-4178, :4318-4364
 *   orc_post_bic   FBIC   :4558-4600, :4738-4792
 * and, for a multivariate data set (d->mv), MVLLik :6099, MVDIC :5789, MVAIC :5116, MVBIC :5452.
 *
 * Parity unpinned: the reference ships no expected values for these functions (its trace fixtures hold no Z / Chi
 * files); the restatement follows the reference's loops line by line and shares calcLikelihood / the fitted mean with the
 * sampler's oracle, which is pinned. */
#include <math.h>
#include <stdlib.h>

#include "oracle.h"

static double dnorm(double x, double mean, double sd, int lg) {      /* R::dnorm */
  const double z = (x - mean) / sd;
  if (lg) return -(0.91893853320467274178 + 0.5 * z * z + log(sd));
  return 0.39894228040143267794 * exp(-0.5 * z * z) / sd;
}

void orc_post_llik(const orc_data* d, const orc_chain* c, int T, double* out) {
  for (int i = 0; i < T; ++i) out[i] = orc_calcLikelihood(d, i, c);
}

double orc_post_dic(const orc_data* d, const orc_chain* c, int T, double burnin_prop) {
  const int kept = (int)round((1 - burnin_prop) * T);
  double expected_log_f = 0;
  for (int i = T - kept; i < T; ++i) expected_log_f = expected_log_f + orc_calcLikelihood(d, i, c);
  expected_log_f = expected_log_f / kept;
  double f_hat = 0;
  if (d->mv) {      /* MVDIC, PostProcessing.cpp:5871-5882: the joint density of a row (calcDIC2MV, CalculateLikelihood.h:172-194) */
    for (int i = 0; i < d->n; ++i) {
      double f_hat_i = 0;
      for (int n = T - kept; n < T; ++n) {
        double lik = 1;
        for (int j = 0; j < orc_ni(d, i); ++j) lik = lik * dnorm(orc_yobs(d, i, j), orc_fitted(d, c, n, i, j), sqrt(c->sigma[n]), 0);
        f_hat_i = f_hat_i + lik;
      }
      f_hat = f_hat + log(f_hat_i / kept);
    }
    return (2 * f_hat) - (4 * expected_log_f);
  }
  for (int i = 0; i < d->n; ++i) {
    for (int j = 0; j < orc_ni(d, i); ++j) {
      double f_hat_ij = 0;
      for (int n = T - kept; n < T; ++n)      /* calcDIC2 / calcDIC2CovariateAdj, CalculateLikelihood.h:59-125 */
        f_hat_ij = f_hat_ij + dnorm(orc_yobs(d, i, j), orc_fitted(d, c, n, i, j), sqrt(c->sigma[n]), 0);
      f_hat = f_hat + log(f_hat_ij / kept);
    }
  }
  return (2 * f_hat) - (4 * expected_log_f);
}

/* log-likelihood at the mean curve fits over the kept draws and the mean of all saved sigma^2 */
static double loglik_at_means(const orc_data* d, const orc_chain* c, int T, double burnin_prop) {
  const int kept = (int)round((1 - burnin_prop) * T);
  double mean_sigma = 0;
  for (int i = 0; i < T; ++i) mean_sigma += c->sigma[i];
  mean_sigma /= T;
  double log_lik = 0;
  for (int i = 0; i < d->n; ++i) {
    for (int j = 0; j < orc_ni(d, i); ++j) {
      double m = 0;
      for (int n = T - kept; n < T; ++n) m += orc_fitted(d, c, n, i, j);
      m /= kept;
      log_lik = log_lik + dnorm(orc_yobs(d, i, j), m, sqrt(mean_sigma), 1);
    }
  }
  return log_lik;
}

static double n_params(const orc_data* d, int has_x, int cov_adj) {
  const double n = d->n, K = d->K, P = d->P, M = d->M, D = d->D;
  double v = (n + P) * K + 2 * P * M * K + 2 + 4 * K + n * M + (M * K);
  if (has_x) {
    v += P * D * K + D * K;
    if (cov_adj) v += 2 * P * D * K * M + D * K * M + 2 * D * K;
  }
  return v;
}

double orc_post_aic(const orc_data* d, const orc_chain* c, int T, double burnin_prop, int has_x, int cov_adj) {
  return 2 * n_params(d, has_x, cov_adj) - (2 * loglik_at_means(d, c, T, burnin_prop));
}

double orc_post_bic(const orc_data* d, const orc_chain* c, int T, double burnin_prop, int has_x, int cov_adj) {
  double tilde_N = 0;
  for (int i = 0; i < d->n; ++i) tilde_N = tilde_N + orc_ni(d, i);
  if (d->mv) tilde_N = d->n;      /* MVBIC: log(Y.n_rows), PostProcessing.cpp:5560, :5717 */
  return (2 * loglik_at_means(d, c, T, burnin_prop)) - (log(tilde_N) * n_params(d, has_x, cov_adj));
}

/* ConditionalPredictiveOrdinates -> calcLikelihoodCPO, CalculateLikelihood.h:344-389: the dense n_i x n_i covariance of
 * every curve under every kept draw, log_det_sympd and inv_sympd through a Cholesky factorisation (what Armadillo's
 * sympd routines do), then the stabilised harmonic mean of :381-386.  Z entries are used as they are (no zero skip). */
void orc_post_cpo(const orc_data* d, const orc_chain* c, int T, double burnin_prop, double* out) {
  const int K = d->K, M = d->M;
  const int first = (int)floor(burnin_prop * T);
  const int L = (int)ceil((1 - burnin_prop) * T);
  double* logl = (double*)malloc(sizeof(double) * (size_t)L);
  for (int i = 0; i < d->n; ++i) {
    const int ni = orc_ni(d, i);
    double* mean = (double*)malloc(sizeof(double) * (size_t)ni);
    double* cov = (double*)malloc(sizeof(double) * (size_t)ni * ni);
    double* Lc = (double*)malloc(sizeof(double) * (size_t)ni * ni);
    double* u = (double*)malloc(sizeof(double) * (size_t)ni * K * M);
    double* w = (double*)malloc(sizeof(double) * (size_t)ni);
    const double* Z_i = c->Z + i;
    for (int l = first; l < T; ++l) {
      for (int a = 0; a < ni; ++a) mean[a] = 0;
      for (int a = 0; a < ni * ni; ++a) cov[a] = 0;
      for (int k = 0; k < K; ++k)
        for (int m = 0; m < M; ++m)
          for (int a = 0; a < ni; ++a) u[(k * M + m) * ni + a] = orc_row_dot(d, c, l, i, a, k, m + 1);
      for (int k = 0; k < K; ++k) {
        const double zk = Z_i[(size_t)d->n * (k + (size_t)K * l)];
        for (int a = 0; a < ni; ++a) mean[a] = mean[a] + zk * orc_row_dot(d, c, l, i, a, k, 0);
        for (int k1 = 0; k1 < K; ++k1) {
          const double zk1 = Z_i[(size_t)d->n * (k1 + (size_t)K * l)];
          for (int m = 0; m < M; ++m)
            for (int a = 0; a < ni; ++a)
              for (int b = 0; b < ni; ++b)
                cov[a + (size_t)ni * b] = cov[a + (size_t)ni * b] + zk * zk1 * (u[(k * M + m) * ni + a] * u[(k1 * M + m) * ni + b]);
        }
      }
      for (int a = 0; a < ni; ++a) cov[a + (size_t)ni * a] = cov[a + (size_t)ni * a] + c->sigma[l];
      orc_chol_lower(ni, cov, Lc);
      double logdet = 0, quad = 0;
      for (int a = 0; a < ni; ++a) {          /* L w = (y - mean):  (y - mean)' cov^-1 (y - mean) = w'w */
        double v = orc_yobs(d, i, a) - mean[a];
        for (int b = 0; b < a; ++b) v -= Lc[a + (size_t)ni * b] * w[b];
        w[a] = v / Lc[a + (size_t)ni * a];
        quad += w[a] * w[a];
        logdet += 2 * log(Lc[a + (size_t)ni * a]);
      }
      logl[l - first] = -(0.5 * ni) * log(2 * 3.14159265358979323846) - 0.5 * logdet - 0.5 * quad;
    }
    double mn = logl[0];
    for (int l = 1; l < L; ++l) if (logl[l] < mn) mn = logl[l];
    double ph = 0;
    for (int l = 0; l < L; ++l) ph = ph + exp(mn - logl[l]);
    out[i] = log((double)L) + mn - log(ph);
    free(mean); free(cov); free(Lc); free(u); free(w);
  }
  free(logl);
}
