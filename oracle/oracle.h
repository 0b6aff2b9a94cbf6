/*
 * oracle/oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, no Armadillo / R) of the per-iteration Gibbs sweep of
 * ndmarco/BayesFMMM, written in "reference-structure mode": the same loop nests,
 * update order, skip rules and integer-division quirks as
 * inst/include/BayesFMMM/Update*.h and the drivers in inst/include/BayesFMMM/BFMMM.h.
 * Each function cites the reference file:line it follows.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library.  The product path (bayesfmmm_amd/, libbfmmm_hip.so) never links or calls it.
 *
 * PARITY PIN STATUS
 *   - B-spline basis / RW penalty: PINNED by the reference's golden files
 *     inst/test-data/Tensor_BSpline.txt and P_mat.txt (src/test-BSplines.cpp:66,81),
 *     committed as fixtures under tests/golden/.
 *   - Full-conditional updates: pinned only by ports of the reference's statistical
 *     recovery tests (src/test-*.cpp tolerances).  The reference itself cannot be built
 *     here (needs R, Rcpp, RcppArmadillo, splines2, RcppDist; none present), and it draws
 *     from R's sequential RNG stream, which a parallel sampler cannot reproduce.
 *     => draw-level parity with the R package is UNPINNED ("parity unpinned" for draws).
 *
 * RNG CONVENTION (shared *specification* with the HIP path, separate implementation):
 *   Philox4x32-10, key = 64-bit seed, counter = (idx, attempt | tt_step<<16, iter,
 *   chain<<8 | update_id).  One block yields two 52-bit uniforms in (0,1).
 *   normal  = AS241 inverse-CDF of U0 (R's default "Inversion" normal.kind uses the same
 *             quantile function, nmath/qnorm.c);
 *   gamma   = Marsaglia-Tsang (2000), one block per attempt (x = qnorm(U0), u = U1);
 *   truncated normal = plain / exponential-tilt rejection (Robert 1995), one block per attempt.
 */
#ifndef BFMMM_ORACLE_H
#define BFMMM_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- update ids used in the RNG counter (must match bayesfmmm_amd/csrc/rng.hpp) ---- */
enum {
  UPD_Z_PROP = 1, UPD_Z_ACC = 2, UPD_PI_PROP = 3, UPD_PI_ACC = 4, UPD_A3_PROP = 5, UPD_A3_ACC = 6,
  UPD_PHI = 7, UPD_DELTA = 8, UPD_A_PROP = 9, UPD_A_ACC = 10, UPD_GAMMA = 11, UPD_NU = 12,
  UPD_TAU = 13, UPD_SIGMA = 14, UPD_CHI = 15, UPD_ETA = 16, UPD_TAU_ETA = 17, UPD_XI = 18,
  UPD_DELTA_XI = 19, UPD_AXI_PROP = 20, UPD_AXI_ACC = 21, UPD_GAMMA_XI = 22,
  UPD_INIT_NU = 30, UPD_INIT_CHI = 31, UPD_INIT_PI = 32, UPD_INIT_Z = 33, UPD_INIT_PHI = 34,
  UPD_INIT_ETA = 35, UPD_INIT_XI = 36,
  UPD_TT_ACC = 40, UPD_SAMPLE_PATH = 41
};

typedef struct {
  uint64_t seed;
  uint32_t chain;
  uint32_t iter;
  uint32_t tt_step; /* 0 outside tempered transitions */
} orc_rng;

/* ---- rng.c ---- */
void   orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
void   orc_block(const orc_rng* r, uint32_t upd, uint32_t idx, uint32_t attempt, double* u0, double* u1);
double orc_qnorm(double p);
double orc_runif(const orc_rng* r, uint32_t upd, uint32_t idx);
double orc_rnorm(const orc_rng* r, uint32_t upd, uint32_t idx);
double orc_rgamma(const orc_rng* r, uint32_t upd, uint32_t idx, double shape, double scale);
double orc_rtruncnorm(const orc_rng* r, uint32_t upd, uint32_t idx, double mu, double sd, double lo, double hi);
double orc_dtruncnorm_log(double x, double mu, double sd, double lo, double hi);
double orc_pnorm(double x);
/* test hooks */
void   orc_rgamma_hook(int arm, uint32_t upd, const double* inject, double* rec_shape_scale, int cap);
double orc_rnorm_ms(const orc_rng* r, uint32_t upd, uint32_t idx, double mean, double sd);      /* mean + sd * N(0, 1), hookable */
void   orc_rnorm_hook(int arm, uint32_t upd, const double* inject, double* rec_mean_sd, int cap);
void   orc_test_fill(uint64_t seed, uint32_t chain, uint32_t iter, uint32_t upd, int kind,
                     double p1, double p2, int count, double* out);

/* ---- linalg.c ---- (column-major P x P) */
int  orc_chol_lower(int P, const double* A, double* L);              /* 0 ok, 1 not SPD */
int  orc_inv(int P, double* A);                                       /* arma::inv (in place) */
void orc_pinv_sym(int P, double* A);                                  /* arma::pinv of a symmetric matrix (in place) */
void orc_mvnrnd(const orc_rng* r, uint32_t upd, uint32_t idx0, int P,
                const double* mean, const double* C, double* out);   /* arma::mvnrnd(mean, C) */
/* rank-deficient precisions (pinv + eigen-decomposition route of the reference; specification in linalg.c) */
int  orc_prec_is_singular(int P, const double* Prec);
void orc_pinv_draw(int P, const double* Prec, const double* rhs, const double* z, double* out);

/* ---- bspline.c ---- */
/* splines2::BSpline(x, internal_knots, degree, boundary_knots).basis(true): n x P row-major out */
int  orc_bspline_basis(int n, const double* x, int n_internal, const double* internal_knots,
                       int degree, const double* boundary_knots, double* out_rowmajor);
void orc_pmat_rw1(int P, double* Pm);                                /* BFMMM.h:1027-1037 */
int  orc_tensor_bspline(int n_pts, int dim, const double* t_colmajor, const int* degree,
                        const double* boundary /* dim x 2 row-major */, const int* n_internal,
                        const double* const* internal_knots, double* out_colmajor);
int  orc_get_P(int dim, const int* degree, const int* n_internal, double* out_colmajor);

/* ---- model data ---- */
typedef struct {
  int n, K, P, M, D;          /* D = 0 when no covariates */
  const int64_t* off;         /* n+1 offsets into y / rows of B */
  const double* y;            /* concatenated observations */
  const double* B;            /* concatenated basis rows, row-major (row l of curve i at (off[i]+l)*P) */
  const double* X;            /* n x D column-major or NULL */
  const double* Pmat;         /* P x P */
  int mv;                     /* 1: multivariate model (B ignored, each curve has exactly P obs) */
} orc_data;

typedef struct {
  double c[16];               /* Dirichlet hyper-parameter for pi (K <= 16) */
  double b, nu_1;
  double alpha1l, alpha2l, beta1l, beta2l;
  double a_Z_PM, a_pi_PM, var_alpha3, var_epsilon1, var_epsilon2;
  double alpha_nu, beta_nu, alpha_eta, beta_eta, alpha_0, beta_0;
} orc_hyper;

/* One full set of chain arrays, laid out exactly as the reference returns them
 * (Armadillo column-major cubes; T slots). */
typedef struct {
  int T;
  double *nu;      /* K x P x T */
  double *chi;     /* n x M x T */
  double *Z;       /* n x K x T */
  double *pi;      /* K x T */
  double *alpha3;  /* T */
  double *delta;   /* K x M x T */
  double *A;       /* K x 2 x T */
  double *sigma;   /* T  (variance) */
  double *tau;     /* T x K (column-major: (t,k) at t + T*k) */
  double *gamma;   /* T slots of K x P x M */
  double *Phi;     /* T slots of K x P x M */
  double *loglik;  /* T */
  /* covariate-adjusted extras (NULL when D == 0) */
  double *eta;       /* T slots of P x D x K */
  double *tau_eta;   /* K x D x T */
  double *xi;        /* T slots of K cubes P x D x M : ((t*K + k) * P*D*M) */
  double *gamma_xi;  /* same shape as xi */
  double *delta_xi;  /* T slots of K x M x D */
  double *A_xi;      /* T slots of K x 2 x D */
} orc_chain;

/* ---- updates.c : functional model (reference-structure loops) ---- */
void orc_updateZ_PM(const orc_data* d, const orc_rng* r, double beta_i, int iter, int T, double a_Z_PM, orc_chain* c);
void orc_updatePi_PM(const orc_data* d, const orc_rng* r, int iter, int T, const double* cvec, double a_pi_PM, orc_chain* c);
void orc_updateAlpha3(const orc_data* d, const orc_rng* r, int iter, int T, double b, double var_alpha3, orc_chain* c);
void orc_updatePhi(const orc_data* d, const orc_rng* r, double beta_i, int iter, int T, const double* tilde_tau, orc_chain* c);
void orc_updateDelta(const orc_data* d, const orc_rng* r, int iter, int T, orc_chain* c);
void orc_updateA(const orc_data* d, const orc_rng* r, int iter, int T, const orc_hyper* h, orc_chain* c);
void orc_updateGamma(const orc_data* d, const orc_rng* r, int iter, int T, double nu_gamma, orc_chain* c);
void orc_updateNu(const orc_data* d, const orc_rng* r, double beta_i, int iter, int T, orc_chain* c);
void orc_updateTau(const orc_data* d, const orc_rng* r, int iter, int T, double alpha, double beta, orc_chain* c);
void orc_updateSigma(const orc_data* d, const orc_rng* r, double beta_i, int tempered, int iter, int T, double alpha_0, double beta_0, orc_chain* c);
void orc_updateChi(const orc_data* d, const orc_rng* r, double beta_i, int iter, int T, orc_chain* c);
double orc_calcLikelihood(const orc_data* d, int iter, const orc_chain* c);
double orc_fitted(const orc_data* d, const orc_chain* c, int iter, int i, int l);
double orc_yobs(const orc_data* d, int i, int l);
int orc_ni(const orc_data* d, int i);
double orc_row_dot(const orc_data* d, const orc_chain* c, int iter, int i, int l, int k, int mt);
void orc_post_cpo(const orc_data* d, const orc_chain* c, int T, double burnin_prop, double* out);

/* post.c: likelihood-based post-processing over a chain of T saved draws (src/PostProcessing.cpp) */
void orc_post_llik(const orc_data* d, const orc_chain* c, int T, double* out);
double orc_post_dic(const orc_data* d, const orc_chain* c, int T, double burnin_prop);
double orc_post_aic(const orc_data* d, const orc_chain* c, int T, double burnin_prop, int has_x, int cov_adj);
double orc_post_bic(const orc_data* d, const orc_chain* c, int T, double burnin_prop, int has_x, int cov_adj);
void orc_tilde_tau(int K, int M, const double* delta_slice, double* tilde_tau);
/* covariate-adjusted extras */
void orc_updateEta(const orc_data* d, const orc_rng* r, double beta_i, int iter, int T, orc_chain* c);
void orc_updateTauEta(const orc_data* d, const orc_rng* r, int iter, int T, double alpha, double beta, orc_chain* c);
void orc_updateXi(const orc_data* d, const orc_rng* r, double beta_i, int iter, int T, const double* tilde_tau_xi, orc_chain* c);
void orc_updateDeltaXi(const orc_data* d, const orc_rng* r, int iter, int T, orc_chain* c);
void orc_updateAXi(const orc_data* d, const orc_rng* r, int iter, int T, const orc_hyper* h, orc_chain* c);
void orc_updateGammaXi(const orc_data* d, const orc_rng* r, int iter, int T, double nu_gamma, orc_chain* c);
void orc_tilde_tau_xi(int K, int M, int D, const double* delta_xi_slice, double* tilde_tau_xi);

/* ---- drivers.c ---- */
/* sweep kinds */
enum { ORC_SWEEP_NU_Z = 0, ORC_SWEEP_THETA = 1, ORC_SWEEP_WARM = 2 };
/* Runs T iterations of the named sweep starting from slot 0 of `c` (which the caller has
 * initialised), following BFMMM.h:1073-1113 / 1253-1298 / 1500-1554 (+1670) or, when
 * d->D > 0, the covariate-adjusted orders (BFMMM.h:3741-3780 / 3944-4010 / 4809-4894).
 * covariance_adj selects the Xi block in the covariate-adjusted sweeps. */
/* tempered transitions (functional model, D == 0): BFMMM.h:1452-1460, :1556-1672; CalculateTTAcceptance.h:22-97 */
void orc_rdirichlet(const orc_rng* r, uint32_t upd, uint32_t idx0, int K, const double* alpha, double* out);   /* updates.c */

/* ---- gram.c : the warm-start sweep in sufficient-statistics form (algorithm vs hardware) ---- */
struct orc_gram_s;
struct orc_gram_s* orc_gram_prepare(const orc_data* d);
void orc_gram_free(struct orc_gram_s* g);
void orc_gram_run_warm(const orc_data* d, const struct orc_gram_s* g, const orc_hyper* h, uint64_t seed, uint32_t chain, int T,
                       int first_iter, int n_iter, orc_chain* c);
void orc_gram_run_nu_z(const orc_data* d, const struct orc_gram_s* g, const orc_hyper* h, uint64_t seed, uint32_t chain, int T,
                       int first_iter, int n_iter, orc_chain* c);

/* updates.c: loops over a basis row restricted to its non-zero window (bit-identical sums; full-size parity tests only) */
void orc_set_row_window(int on);

void orc_beta_ladder(int N_t, double beta_N_t, double* ladder);
double orc_calculatePZeta(const orc_data* d, double beta_i, int iter, const orc_chain* c);
double orc_calculatePZetaCov(const orc_data* d, double beta_i, int iter, const orc_chain* c);   /* updates.c */
double orc_CalculateTTAcceptance(const orc_data* d, int N_t, const double* beta, const orc_chain* tt);
void orc_tt_block(const orc_data* d, const orc_hyper* h, uint64_t seed, uint32_t chain, int i, int N_t,
                  double beta_N_t, orc_chain* c, double* logA_out, int* accepted_out);
void orc_run_warm_tt(const orc_data* d, const orc_hyper* h, uint64_t seed, uint32_t chain, int T, int first_iter,
                     int n_iter, int N_t, int n_temp_trans, double beta_N_t, orc_chain* c, double* logA, int* accepted);
void orc_tt_block_cov(const orc_data* d, const orc_hyper* h, uint64_t seed, uint32_t chain, int i, int N_t,
                      double beta_N_t, int covariance_adj, orc_chain* c, double* logA_out, int* accepted_out);
void orc_run_warm_tt_cov(const orc_data* d, const orc_hyper* h, uint64_t seed, uint32_t chain, int T, int first_iter,
                         int n_iter, int N_t, int n_temp_trans, double beta_N_t, int covariance_adj, orc_chain* c,
                         double* logA, int* accepted);
void orc_run_sweeps(const orc_data* d, const orc_hyper* h, uint64_t seed, uint32_t chain,
                    int sweep, int covariance_adj, int T, int first_iter, int n_iter, orc_chain* c);
/* initial state of BFMMM_Nu_Z (BFMMM.h:1039-1071) / BFMMM_Theta (1210-1250) */
void orc_init_nu_z(const orc_data* d, const orc_hyper* h, uint64_t seed, uint32_t chain, orc_chain* c);
void orc_init_theta(const orc_data* d, const orc_hyper* h, uint64_t seed, uint32_t chain,
                    const double* Z_est, const double* nu_est, const double* eta_est, orc_chain* c);

#ifdef __cplusplus
}
#endif
#endif
