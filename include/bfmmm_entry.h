/*
 * bfmmm_entry.h -- C ABI of the reference's user entry points for the functional model, built
 * on the sampler ABI of bfmmm.h.  Each function mirrors one exported C++ function of the
 * reference (argument names, defaults, validation messages, result names and layouts):
 *
 *   bfmmm_BFMMM_Nu_Z_multiple_try  <-  BFMMM_Nu_Z_multiple_try  src/UserFunctions.cpp:166-498
 *                                      (.Call symbol _BayesFMMM_BFMMM_Nu_Z_multiple_try, src/RcppExports.cpp:358)
 *   bfmmm_BFMMM_Theta_est          <-  BFMMM_Theta_est          src/UserFunctions.cpp:684-1114   (RcppExports.cpp:396)
 *   bfmmm_BFMMM_warm_start         <-  BFMMM_warm_start         src/UserFunctions.cpp:1341-2155  (RcppExports.cpp:438)
 *
 * R lists (Rcpp::List) become `bfmmm_result` objects: named, caller-readable arrays of doubles in
 * the reference's column-major layouts.  A result produced by one stage is passed to the next
 * stage exactly as the R lists `multiple_try` / `theta_est` are; a shim can also build one from an
 * R list with bfmmm_result_create / bfmmm_result_set.
 *
 * Differences from the reference that the ABI makes explicit:
 *   - `seed`: the reference draws from R's global RNG (Rcpp::RNGScope, RcppExports.cpp:361); here every
 *     chain uses the keyed generator (seed, chain index).
 *   - multi-try chains are independent (UserFunctions.cpp:302-325 runs 1 + n_try of them back to
 *     back); `chain_offset` / `chain_stride` let one process run the subset
 *     {chain_offset, chain_offset + chain_stride, ...} <= n_try of the chain indices so that the
 *     chains can be spread over GPUs; the result carries "best_chain" and "best_score" for the final
 *     selection.  With chain_offset = 0, chain_stride = 1 the call is the reference's.
 */
#ifndef BFMMM_ENTRY_H
#define BFMMM_ENTRY_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bfmmm_result bfmmm_result;

bfmmm_result* bfmmm_result_create(void);
void bfmmm_result_free(bfmmm_result* r);
/* copies `count` doubles; dims (n_dims entries) record the array's shape */
int bfmmm_result_set(bfmmm_result* r, const char* name, const double* data, int64_t count, const int64_t* dims, int n_dims);
/* borrowed pointers, valid until the result is freed or the name is set again */
int bfmmm_result_get(const bfmmm_result* r, const char* name, const double** data, int64_t* count,
                     const int64_t** dims, int* n_dims);
int bfmmm_result_count(const bfmmm_result* r);
const char* bfmmm_result_name(const bfmmm_result* r, int index);

typedef struct {
  /* data: CSR form of the R lists `Y` and `time` */
  int32_t n_funct;
  const double* y;
  const double* t;
  const int64_t* offsets;          /* n_funct + 1 */
  /* model */
  int32_t tot_mcmc_iters, n_try, K, basis_degree, n_eigen, n_internal_knots;
  const double* boundary_knots;    /* 2 */
  const double* internal_knots;    /* n_internal_knots */
  const double* c;                 /* K, or NULL for the default rep(10, K) (UserFunctions.cpp:272) */
  double burnin_prop;              /* Theta_est / warm_start */
  double b, nu_1, alpha1l, alpha2l, beta1l, beta2l, a_Z_PM, a_pi_PM, var_alpha3, var_epsilon1, var_epsilon2;
  double alpha_nu, beta_nu, alpha_eta, beta_eta, alpha_0, beta_0;
  /* warm_start only (UserFunctions.cpp:1353-1359) */
  double thinning_num, beta_N_t;
  int32_t N_t, n_temp_trans, r_stored_iters;
  /* execution */
  uint64_t seed;
  int32_t device;
  int32_t chain_offset, chain_stride;
  int32_t max_concurrent;          /* chains per sampler batch: that many multi-try chains advance in lockstep on a device (>= 1) */
  /* multivariate model (BMVMMM_*): model = 1, y = the n_funct x P column-major matrix `Y`
   * (UserFunctions.cpp:4582); t, offsets and the knot arguments are ignored */
  int32_t model;
  int32_t P;
  /* covariate adjustment: the `X` (n_funct x D, column-major) and `covariance_adj` arguments of the reference
   * (UserFunctions.cpp:176, :715); NULL = no covariates.  Results then also carry "eta" (P x D x K x T),
   * "tau_eta" (K x D x T) and, from Theta_est / warm_start, "xi" / "gamma_xi" (P x D x M x K x T), "delta_xi"
   * (K x M x D x T), "A_xi" (K x 2 x D x T). */
  const double* X;
  int32_t D;
  int32_t covariance_adj;
  /* warm_start only: the `dir` argument (UserFunctions.cpp:1354, :1515-1530), NULL = nothing is saved.  With a
   * directory and 0 < r_stored_iters <= tot_mcmc_iters only r_stored_iters draws are kept in memory: after every
   * r_stored_iters iterations the batch is thinned (every thinning_num-th draw) and saved as <dir>Nu<q>.txt, Chi, Pi,
   * alpha_3, A, Delta, Sigma, Tau, Z (arma_ascii), Gamma, Phi (binary fields of cubes) -- plus Eta, Xi, Gamma_Xi,
   * Delta_Xi, A_Xi, Tau_Eta with covariates -- exactly as BFMMM_MTT_warm_start does (BFMMM.h:1680-1746, :5086-5165);
   * the returned arrays then have r_stored_iters slots (the last batch in memory). */
  const char* dir;
  /* high-dimensional functional model (BHDFMMM_*, src/UserFunctions.cpp:2519, :3030, :3676): dim > 0.  Then `t` holds,
   * curve after curve, the n_i x dim matrix of time points (column-major, as R stores each element of the `time` list);
   * basis_degree_hd / n_internal_hd have dim entries (the arma::vec `basis_degree` and the lengths of the
   * `internal_knots` list); internal_knots holds the dimensions' knots one after the other; boundary_knots is dim x 2
   * row-major (lower, upper per dimension); basis_degree / n_internal_knots are ignored. */
  int32_t dim;
  const int32_t* basis_degree_hd;
  const int32_t* n_internal_hd;
  /* warm start: every `progress_every` iterations (0: never) progress_cb(last iteration, its log-likelihood, progress_user)
   * is called on the calling thread between device batches -- the counterpart of the reference's Rcpp::checkUserInterrupt() /
   * Rcout progress lines (BFMMM.h:1674-1678); a non-zero return aborts the run with an error */
  int32_t progress_every;
  int (*progress_cb)(int32_t iter, double loglik, void* user);
  void* progress_user;
  /* multi-GPU multi-try: with n_devices > 0 the chains of this call are dealt round-robin over devices[0 .. n_devices)
   * (one host thread and one sampler batch per device) and the best chain is selected by the RCCL gather of
   * bfmmm_gather_best of bfmmm.h: an all-gather of one score per device, then the winner's chain is sent over xGMI to
   * devices[0].  n_devices == 0: everything runs on `device`, no communicator is created. */
  const int32_t* devices;
  int32_t n_devices;
} bfmmm_entry_args;

/* fills in the reference defaults of the named entry point:
 *   0: BFMMM_Nu_Z_multiple_try (alpha1l = 1, alpha2l = 2, beta1l = beta2l = 1; UserFunctions.cpp:178-193)
 *   1: BFMMM_Theta_est, 2: BFMMM_warm_start (alpha1l = 2, alpha2l = 3, beta1l = beta2l = 2; :695-715, :1353-1378)
 *   3: BMVMMM_Nu_Z_multiple_try (alpha1l = 2, alpha2l = 3, beta1l = beta2l = 1; :4586-4589)
 *   4: BMVMMM_Theta_est, 5: BMVMMM_warm_start (alpha1l = 1, alpha2l = 2, beta1l = beta2l = 1; :5006-5009, :5556-5559) */
void bfmmm_entry_defaults(bfmmm_entry_args* a, int entry);

/* result names: "B" (all basis rows, row-major, see bfmmm_get_basis), "nu" K x P x T, "pi" K x T, "alpha_3" T,
 * "A" K x 2 x T, "delta" K x M x T, "sigma_sq" T, "tau" T x K, "Z" n x K x T, "loglik" T, "best_chain", "best_score" */
int bfmmm_BFMMM_Nu_Z_multiple_try(const bfmmm_entry_args* a, bfmmm_result** out);
/* multiple_try must hold "Z" and "nu".  result names: "B", "Z", "nu", "chi" n x M x T, "A", "delta", "sigma_sq",
 * "tau", "gamma" / "Phi" (T arrays K x P x M), "loglik", "best_chain", "best_score" */
int bfmmm_BFMMM_Theta_est(const bfmmm_entry_args* a, const bfmmm_result* multiple_try, bfmmm_result** out);
/* result names: "B_obs", "Z", "nu", "chi", "pi", "alpha_3", "A", "delta", "sigma_sq", "tau", "gamma", "Phi", "loglik";
 * every array has tot_mcmc_iters + 1 slots when r_stored_iters == 0 (UserFunctions.cpp:1510-1541, BFMMM.h:1414-1434) */
int bfmmm_BFMMM_warm_start(const bfmmm_entry_args* a, const bfmmm_result* multiple_try, const bfmmm_result* theta_est,
                           bfmmm_result** out);

/* Multivariate model: BMVMMM_Nu_Z_multiple_try (src/UserFunctions.cpp:4579, .Call symbol at src/RcppExports.cpp:680),
 * BMVMMM_Theta_est (:4995 / RcppExports.cpp:713), BMVMMM_warm_start (:5540 / RcppExports.cpp:750).  Same results as
 * the functional entry points minus "B" / "B_obs". */
int bfmmm_BMVMMM_Nu_Z_multiple_try(const bfmmm_entry_args* a, bfmmm_result** out);
int bfmmm_BMVMMM_Theta_est(const bfmmm_entry_args* a, const bfmmm_result* multiple_try, bfmmm_result** out);
int bfmmm_BMVMMM_warm_start(const bfmmm_entry_args* a, const bfmmm_result* multiple_try, const bfmmm_result* theta_est,
                            bfmmm_result** out);

/* Readers / writers of the on-disk batches (Armadillo arma_ascii and arma_binary field files), the counterparts of
 * ReadVec / ReadMat / ReadCube (src/UserFunctions.cpp:2158, :2205, :2253: result element "value", with its dims) and
 * ReadFieldCube / ReadFieldMat / ReadFieldVec (:2303, :2351, :2399: result elements "field_dims" = (n_rows, n_cols) and
 * "0" .. "n-1", the field's objects in column-major order).  bfmmm_arma_write_ascii writes a vector / matrix / cube
 * (n_dims = 1, 2, 3) as `.save(file, arma::arma_ascii)` does; bfmmm_arma_write_field writes the objects "0" .. of
 * `items` as a binary field (`field.save(file)`). */
int bfmmm_arma_read(const char* file, bfmmm_result** out);
int bfmmm_arma_read_field(const char* file, bfmmm_result** out);
int bfmmm_arma_write_ascii(const char* file, const double* data, const int64_t* dims, int n_dims);
int bfmmm_arma_write_field(const char* file, const bfmmm_result* items, int64_t n_rows, int64_t n_cols);

/* Set-up pieces of the high-dimensional functional model (the BHDFMMM_* entry points build them per call): the tensor-product B-spline basis of TensorBSpline (inst/include/BayesFMMM/BSplines.h:18-66) for n_pts
 * points in `dim` dimensions -- t is n_pts x dim column-major, boundary_knots dim x 2 row-major, internal_knots the
 * dimensions' knots one after the other, out n_pts x P column-major with the last dimension's index running fastest --
 * and the penalty matrix of GetP (BSplines.h:74-120), P x P. */
int bfmmm_tensor_bspline(int n_pts, int dim, const double* t, const int* degree, const double* boundary_knots,
                         const int* n_internal, const double* internal_knots, double* out);
int bfmmm_tensor_penalty(int dim, const int* degree, const int* n_internal, double* out);

/* High-dimensional functional model: BHDFMMM_Nu_Z_multiple_try (src/UserFunctions.cpp:2519), BHDFMMM_Theta_est (:3030),
 * BHDFMMM_warm_start (:3676); args.dim > 0.  Same results as the functional entry points ("B" / "B_obs": the rows of the
 * tensor-product basis). */
int bfmmm_BHDFMMM_Nu_Z_multiple_try(const bfmmm_entry_args* a, bfmmm_result** out);
int bfmmm_BHDFMMM_Theta_est(const bfmmm_entry_args* a, const bfmmm_result* multiple_try, bfmmm_result** out);
int bfmmm_BHDFMMM_warm_start(const bfmmm_entry_args* a, const bfmmm_result* multiple_try, const bfmmm_result* theta_est,
                             bfmmm_result** out);

/* message of the last failing bfmmm_result_* / bfmmm_BFMMM_* call on this thread */
const char* bfmmm_entry_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
