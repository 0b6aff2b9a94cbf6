/*
 * bfmmm_post.h -- C ABI of the likelihood-based post-processing functions of the functional model:
 *
 *   bfmmm_FLLik  <-  FLLik  src/PostProcessing.cpp:4892-5114  (log-likelihood of every saved draw)
 *   bfmmm_FDIC   <-  FDIC   src/PostProcessing.cpp:3660-4039  (deviance information criterion)
 *   bfmmm_FAIC   <-  FAIC   src/PostProcessing.cpp:4041-4456
 *   bfmmm_FBIC   <-  FBIC   src/PostProcessing.cpp:4458-4801
 *   bfmmm_SigmaCI / bfmmm_ZCI / bfmmm_FMeanCI  <-  SigmaCI :3435, ZCI :3505, FMeanCI :99
 *   bfmmm_ConditionalPredictiveOrdinates  <-  ConditionalPredictiveOrdinates  src/PostProcessing.cpp:6339-6516
 *   bfmmm_MVLLik / bfmmm_MVDIC / bfmmm_MVAIC / bfmmm_MVBIC  <-  MVLLik :6099, MVDIC :5789, MVAIC :5116, MVBIC :5452
 *
 * All four evaluate the fitted value of every observation under every saved draw
 *   f_ij(t) = B_ij' sum_k Z_ik(t) [ nu_k(t) + eta_k(t) x_i + sum_m chi_im(t) (phi_km(t) + xi_km(t) x_i) ]
 * (calcLikelihoodCovariateAdj / calcDIC2CovariateAdj, inst/include/BayesFMMM/CalculateLikelihood.h:19-44, :59-125) and
 * differ only in how the residuals are reduced; bfmmm_post_pointwise is that evaluation on the MI355X (one pass over the
 * draws: per-draw log-likelihood, per-observation mean density and mean fitted value over the kept draws), the four
 * entry points read the on-disk batches (`<dir>Nu<q>.txt`, `Phi`, `Z`, `Chi`, `Sigma`, and `Eta`, `Xi` with covariates,
 * q < n_files; the files bfmmm_BFMMM_warm_start writes), call it and apply the reference's formulas.
 * Errors: non-zero return, message via bfmmm_entry_last_error() (bfmmm_entry.h), with the reference's wording.
 */
#ifndef BFMMM_POST_H
#define BFMMM_POST_H

#include <stdint.h>

#include "bfmmm_entry.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  int32_t n, K, P, M, D;           /* curves, clusters, basis functions, eigenfunctions, covariates (0: none) */
  const int64_t* offsets;          /* n + 1 */
  const double* y;                 /* offsets[n] observations */
  const double* B;                 /* offsets[n] x P basis rows, row-major */
  const double* X;                 /* n x D column-major, or NULL */
  int32_t T;                       /* draws, in the layouts of the reference's containers: */
  const double* nu;                /* K x P x T */
  const double* Phi;               /* T cubes K x P x M */
  const double* Z;                 /* n x K x T */
  const double* chi;               /* n x M x T */
  const double* sigma;             /* T */
  const double* eta;               /* T cubes P x D x K, or NULL (zero) */
  const double* xi;                /* T x K cubes P x D x M, cube (t, k) at offset (t * K + k) * P * D * M, or NULL (zero) */
  int32_t device;
  int32_t identity_basis;          /* multivariate model: B may be NULL, every row has P observations and observation j's basis row is e_j */
} bfmmm_post_input;

/* llik: T; mean_pdf, mean_fit: offsets[n] (means over the draws t >= first_kept); any of the three may be NULL */
int bfmmm_post_pointwise(const bfmmm_post_input* in, int32_t first_kept, double* llik, double* mean_pdf, double* mean_fit);
/* the same pass with, instead of the per-observation mean density, the mean over the kept draws of each curve's JOINT
 * density prod_j dnorm(y_ij; f_ij, sigma) (calcDIC2MV, CalculateLikelihood.h:172-194): mean_joint_pdf has n entries */
int bfmmm_post_pointwise_joint(const bfmmm_post_input* in, int32_t first_kept, double* llik, double* mean_joint_pdf, double* mean_fit);
/* log CPO_i of every curve over the draws t >= first_kept (calcLikelihoodCPO, CalculateLikelihood.h:344-389): the marginal
 * log-density of y_i under each kept draw (scores integrated out; the rank-M form of the reference's n_i x n_i
 * log_det_sympd / inv_sympd) and the harmonic mean in the reference's stabilised form.  log_cpo has n entries. */
int bfmmm_post_cpo(const bfmmm_post_input* in, int32_t first_kept, double* log_cpo);
/* device time (ms, HIP events) of the kernels of the last bfmmm_post_pointwise call of this process: measurement aid */
double bfmmm_post_last_kernel_ms(void);

typedef struct {
  const char* dir;                 /* as the reference: file = dir + "Nu" + q + ".txt" (end it with "/") */
  int32_t n_files;
  int32_t basis_degree, n_internal_knots;
  const double* boundary_knots;    /* 2 */
  const double* internal_knots;
  int32_t n_funct;                 /* CSR form of the R lists `time` and `Y` */
  const double* t;
  const double* y;
  const int64_t* offsets;
  double burnin_prop;              /* FDIC / FAIC / FBIC (default 0.1) */
  const double* X;                 /* n_funct x D column-major, or NULL */
  int32_t D;
  int32_t cov_adj;
  int32_t device;
  int32_t P;                       /* bfmmm_MV*: y is the n_funct x P data matrix, column-major; t, offsets, knots unused */
} bfmmm_post_args;

void bfmmm_post_defaults(bfmmm_post_args* a);
int bfmmm_FLLik(const bfmmm_post_args* a, bfmmm_result** out);      /* result element "value": one entry per saved draw */
int bfmmm_FDIC(const bfmmm_post_args* a, double* out);
int bfmmm_FAIC(const bfmmm_post_args* a, double* out);
int bfmmm_FBIC(const bfmmm_post_args* a, double* out);
/* ConditionalPredictiveOrdinates (src/PostProcessing.cpp:6339): result element "value", one entry per curve; log_CPO = 0
 * returns exp of it */
int bfmmm_ConditionalPredictiveOrdinates(const bfmmm_post_args* a, int32_t log_CPO, bfmmm_result** out);
/* FSamplePaths (src/PostProcessing.cpp:6599): posterior-predictive sample paths of every curve at its own time points (a->y is
 * not used and may be NULL) over the last round(T (1 - burnin_prop)) draws, with pointwise or simultaneous bands per curve.
 * Result elements, concatenated over the curves: "CI_Upper", "CI_50", "CI_Lower" (offsets[n] entries), "Path_trace" and
 * "Mean_only_Path_trace" (kept x offsets[n]: curve i is the kept x n_i block, draw fastest, at kept * offsets[i]).  The
 * predictive noise is drawn from the keyed generator (seed, draw index, observation index); the reference uses R's stream. */
int bfmmm_FSamplePaths(const bfmmm_post_args* a, double alpha, int32_t simultaneous, uint64_t seed, bfmmm_result** out);
/* the device pass behind it: fitted value + noise ("paths") and the mean-only value of every (kept draw, observation) */
int bfmmm_post_sample_paths(const bfmmm_post_input* in, int32_t first_kept, uint64_t seed, double* paths, double* mean_only);
/* bands of a T x ncol table (draw fastest) as is: pointwise (alpha / 2, 0.5, 1 - alpha / 2) quantiles of every column, or the
 * simultaneous band over all columns (src/PostProcessing.cpp:6829-6855) */
int bfmmm_post_table_bands(const double* V, int32_t T, int32_t ncol, double alpha, int32_t simultaneous, int32_t device,
                           double* upper, double* mid, double* lower);
/* multivariate model: MVLLik (src/PostProcessing.cpp:6099), MVDIC (:5789), MVAIC (:5116), MVBIC (:5452) */
int bfmmm_MVLLik(const bfmmm_post_args* a, bfmmm_result** out);
int bfmmm_MVDIC(const bfmmm_post_args* a, double* out);
int bfmmm_MVAIC(const bfmmm_post_args* a, double* out);
int bfmmm_MVBIC(const bfmmm_post_args* a, double* out);

/* ---- credible intervals over the saved draws (kernels_bands.hip) ---------------------------------------------------
 * bfmmm_post_col_quantiles: quantiles of every column of a T x ncol table (column-major, draw fastest) with Armadillo's
 * arma::quantile rule; out[q + nq * col].  bfmmm_post_bands: the table f[t][j] = B_j' coef_t (coef: T x P, one row per
 * draw; B: n_t x P row-major) and its pointwise (alpha / 2, 0.5, 1 - alpha / 2) or simultaneous bands
 * (src/PostProcessing.cpp:228-238, :284-302); trace (T x n_t, draw fastest) may be NULL.  Columns of more than 8192
 * draws are sorted through global memory (any length up to 2^24). */
int bfmmm_post_col_quantiles(const double* V, int32_t T, int32_t ncol, const double* probs, int32_t nq, int32_t device, double* out);
int bfmmm_post_bands(const double* coef, int32_t T, int32_t P, const double* B, int32_t n_t, double alpha, int32_t simultaneous,
                     int32_t device, double* upper, double* mid, double* lower, double* trace);

/* covariance surface between clusters l and m (FCovCI): coefL / coefM are (T M) x P, row t M + j = Phi.slice(j).row(l - 1) /
 * row(m - 1) of draw t; upper / mid / lower n1 x n2 column-major; trace T x (n1 n2), draw fastest, or NULL */
int bfmmm_post_cov_bands(const double* coefL, const double* coefM, int32_t T, int32_t M, int32_t P, const double* B1, int32_t n1,
                         const double* B2, int32_t n2, double alpha, int32_t simultaneous, int32_t device, double* upper, double* mid,
                         double* lower, double* trace);

typedef struct {
  const char* dir;
  int32_t n_files;
  const double* time;              /* FMeanCI: the n_time points of the band; FCovCI: time1 */
  int32_t n_time;
  int32_t basis_degree, n_internal_knots;
  const double* boundary_knots;
  const double* internal_knots;
  int32_t k;                       /* FMeanCI: cluster, 1-based as in R */
  double alpha;                    /* default 0.05 */
  int32_t rescale;                 /* default 1 (ignored with a message for K > 2, as the reference) */
  int32_t simultaneous;            /* default 0 */
  double burnin_prop;              /* default 0.1 */
  const double* X;                 /* FMeanCI: n_x x D covariate settings, column-major, or NULL */
  int32_t n_x, D;
  const double* trans_mats;        /* FMeanCI: (kept draws * K) x K, column-major, or NULL */
  int32_t device;
  const double* time2;             /* FCovCI: the second axis (n_time2 points) */
  int32_t n_time2;
  int32_t l, m;                    /* FCovCI: the two clusters, 1-based */
  /* HDFMeanCI (dim > 0): `time` is n_time x dim column-major, boundary_knots dim x 2 row-major, internal_knots the dimensions'
   * knots one after the other (as in bfmmm_entry_args); basis_degree / n_internal_knots are ignored */
  int32_t dim;
  const int32_t* basis_degree_hd;
  const int32_t* n_internal_hd;
} bfmmm_ci_args;

void bfmmm_ci_defaults(bfmmm_ci_args* a);
/* SigmaCI (src/PostProcessing.cpp:3435): "CI_Upper", "CI_50", "CI_Lower" -- the reference returns the MEDIAN as CI_Lower
 * (:3496 `CI_Lower = q(1)`), reproduced; ZCI (:3505): n x K matrices and "Z_trace"; FMeanCI (:99): vectors of n_time (n_x x
 * n_time matrices with X) and "mean_trace" (kept x n_time; with X a cube n_x x n_time x kept). */
int bfmmm_SigmaCI(const bfmmm_ci_args* a, bfmmm_result** out);
int bfmmm_ZCI(const bfmmm_ci_args* a, bfmmm_result** out);
int bfmmm_FMeanCI(const bfmmm_ci_args* a, bfmmm_result** out);
/* HDFMeanCI (src/PostProcessing.cpp:806): FMeanCI over the tensor-product basis (args.dim > 0).  The reference's function has no
 * trans_mats argument (RcppExports.cpp:40); the field is honoured here if set. */
int bfmmm_HDFMeanCI(const bfmmm_ci_args* a, bfmmm_result** out);
/* FCovCI (src/PostProcessing.cpp:1781): "CI_Upper", "CI_50", "CI_Lower" (n_time x n_time2) and "cov_trace" (n_time x n_time2 x
 * kept).  With X (n_x covariate settings; the covariance then depends on them through xi, :2199-2206) the bands are cubes
 * n_time x n_time2 x n_x and "cov_trace" is n_time x n_time2 x kept x n_x (the n_x cubes of the reference's field); trans_mats
 * is honoured only without X, as in the reference.  The reference allocates CI_Lower as n_time2 x n_time2 (:1879):
 * n_time > n_time2 is refused here.
 * HDFCovCI (:2468): the same over the tensor-product basis (args.dim > 0; `time`, `time2` n x dim column-major).  The reference
 * builds BOTH bases from time1 (:2570), so the surface is evaluated on time1 x time1 and n_time2 must equal n_time.
 * MVCovCI (:3097): the multivariate model (identity basis): P x P bands (x n_x with X), pointwise only, no trans_mats. */
int bfmmm_FCovCI(const bfmmm_ci_args* a, bfmmm_result** out);
int bfmmm_HDFCovCI(const bfmmm_ci_args* a, bfmmm_result** out);
int bfmmm_MVCovCI(const bfmmm_ci_args* a, bfmmm_result** out);
/* MVMeanCI (src/PostProcessing.cpp:1410): K x P matrices and "mean_trace" (K x P x kept); with X: K x P x n_x cubes and
 * "mean_trace" as K x P x (kept n_x), the n_x cubes of the reference's field one after the other */
int bfmmm_MVMeanCI(const bfmmm_ci_args* a, bfmmm_result** out);

#ifdef __cplusplus
}
#endif
#endif
