/*
 * bfmmm.h -- C ABI of the MI355X-native Gibbs sampler for the functional / multivariate
 * mixed-membership models of ndmarco/BayesFMMM.
 *
 * This is the drop-in boundary for the sampler's hot path.  The reference exposes that path to R
 * through Rcpp-generated `.Call` entry points
 *     _BayesFMMM_BFMMM_Nu_Z_multiple_try   src/RcppExports.cpp:358   (R/RcppExports.R:1604)
 *     _BayesFMMM_BFMMM_Theta_est           src/RcppExports.cpp:396   (R/RcppExports.R:1791)
 *     _BayesFMMM_BFMMM_warm_start          src/RcppExports.cpp:438   (R/RcppExports.R:2018)
 *     _BayesFMMM_BMVMMM_Nu_Z_multiple_try  src/RcppExports.cpp:680
 *     _BayesFMMM_BMVMMM_Theta_est          src/RcppExports.cpp:713
 *     _BayesFMMM_BMVMMM_warm_start         src/RcppExports.cpp:750
 * whose C++ bodies (src/UserFunctions.cpp:166, 684, 1341, 4579, 4995, 5540) build B-splines, run the
 * chain drivers of inst/include/BayesFMMM/BFMMM.h and return named lists of Armadillo arrays.
 * A `.Call` shim (shim/bfmmm_rcall.cpp) marshals SEXPs onto the plain-C entry
 * points below: ragged R lists become CSR arrays (values + offsets), matrices stay column-major,
 * results are copied into caller-allocated buffers laid out exactly like the reference's return
 * values.  No C++ types, no torch types, no exceptions cross this boundary.
 *
 * All functions return 0 on success and a non-zero code on failure; bfmmm_last_error() then gives
 * the message (the argument-validation messages are the reference's own, UserFunctions.cpp:198-286).
 * The library needs a HIP device: there is no CPU fallback.
 */
#ifndef BFMMM_H
#define BFMMM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bfmmm_handle bfmmm_handle;

enum { BFMMM_MODEL_FUNCTIONAL = 0, BFMMM_MODEL_MULTIVARIATE = 1 };

/* Update mask bits: every reference driver applies its updates in this relative order
 * (BFMMM.h:1073-1107 Nu_Z, :1253-1292 Theta, :1502-1553 warm start). */
enum {
  BFMMM_U_Z = 1 << 0, BFMMM_U_PI = 1 << 1, BFMMM_U_ALPHA3 = 1 << 2, BFMMM_U_PHI = 1 << 3,
  BFMMM_U_DELTA = 1 << 4, BFMMM_U_A = 1 << 5, BFMMM_U_GAMMA = 1 << 6, BFMMM_U_NU = 1 << 7,
  BFMMM_U_TAU = 1 << 8, BFMMM_U_SIGMA = 1 << 9, BFMMM_U_CHI = 1 << 10,
  BFMMM_U_ETA = 1 << 11, BFMMM_U_TAU_ETA = 1 << 12, BFMMM_U_XI = 1 << 13, BFMMM_U_DELTA_XI = 1 << 14,
  BFMMM_U_A_XI = 1 << 15, BFMMM_U_GAMMA_XI = 1 << 16, BFMMM_U_LOGLIK = 1 << 17
};
/* sweeps of the three stages */
#define BFMMM_SWEEP_NU_Z  (BFMMM_U_Z | BFMMM_U_PI | BFMMM_U_ALPHA3 | BFMMM_U_NU | BFMMM_U_TAU | BFMMM_U_SIGMA | BFMMM_U_LOGLIK)
#define BFMMM_SWEEP_THETA (BFMMM_U_PHI | BFMMM_U_DELTA | BFMMM_U_A | BFMMM_U_GAMMA | BFMMM_U_TAU | BFMMM_U_SIGMA | BFMMM_U_CHI | BFMMM_U_LOGLIK)
#define BFMMM_SWEEP_WARM  (BFMMM_SWEEP_NU_Z | BFMMM_SWEEP_THETA)
/* covariate-adjusted drivers (BFMMM.h:3741-3780, 3944-4010, 4248-4312 / 4809-4894): the eta / Xi blocks run after chi */
#define BFMMM_COV_MEAN (BFMMM_U_ETA | BFMMM_U_TAU_ETA)
#define BFMMM_COV_XI   (BFMMM_U_XI | BFMMM_U_DELTA_XI | BFMMM_U_A_XI | BFMMM_U_GAMMA_XI)

/* Hyper-parameters and sizes; field names follow the reference's argument names
 * (UserFunctions.cpp:166-193, 684-715, 1341-1378).  bfmmm_config_defaults() fills in the
 * reference defaults of the functional entry points. */
typedef struct {
  int32_t model;             /* BFMMM_MODEL_* */
  int32_t n_funct;           /* number of curves (rows of Y for the multivariate model) */
  int32_t K;                 /* clusters */
  int32_t n_eigen;           /* M */
  int32_t basis_degree;      /* functional model only */
  int32_t n_internal_knots;  /* functional model only; P = n_internal_knots + basis_degree + 1 */
  int32_t P;                 /* multivariate model: dimension of the observations */
  int32_t tot_mcmc_iters;    /* chain slots to allocate (r_stored_iters == tot_mcmc_iters) */
  double c[8];               /* Dirichlet hyper-parameter of pi (length K) */
  double b, nu_1;
  double alpha1l, alpha2l, beta1l, beta2l;
  double a_Z_PM, a_pi_PM, var_alpha3, var_epsilon1, var_epsilon2;
  double alpha_nu, beta_nu, alpha_eta, beta_eta, alpha_0, beta_0;
} bfmmm_config;

void bfmmm_config_defaults(bfmmm_config* cfg);

/* Creates a sampler for one data set and uploads it to the GPU `device`.
 *   functional:   y, t are the concatenated observations / time points of all curves,
 *                 offsets[n_funct+1] delimits curve i as [offsets[i], offsets[i+1]) -- the CSR form
 *                 of the R lists `Y` and `time` (arma::field<arma::vec>, UserFunctions.cpp:169-170);
 *                 internal_knots[n_internal_knots], boundary_knots[2] as in UserFunctions.cpp:174-175.
 *   multivariate: y is the n_funct x P column-major matrix `Y` (UserFunctions.cpp:4582);
 *                 t, offsets, knots are ignored (may be NULL).
 * The B-spline basis S(t_i) and the per-curve statistics S_i S_i', S_i y_i, y_i'y_i are computed on
 * the device (replacing BFMMM.h:1017-1025).  The caller's arrays are not retained. */
int bfmmm_create(const bfmmm_config* cfg, int device, const double* y, const double* t, const int64_t* offsets,
                 const double* internal_knots, const double* boundary_knots, bfmmm_handle** out);

/* The functional model over a basis supplied by the caller instead of univariate B-splines: B holds one row of P basis
 * values per observation (row-major, curve i's rows at offsets[i] .. offsets[i+1]), `band` is the half-bandwidth of
 * B_i'B_i (|p - q| > band => zero), Pmat (P x P, column-major) the penalty of the nu prior and pen_band its
 * half-bandwidth.  This is how the high-dimensional functional model enters (BHDFMMM_*: tensor-product basis and
 * penalty of inst/include/BayesFMMM/BSplines.h:18-120, bfmmm_tensor_bspline / bfmmm_tensor_penalty in bfmmm_entry.h);
 * the drivers of that model (BFMMM.h:2892, :3041, :3210) run the same updates.  band <= 31, P <= 64. */
int bfmmm_create_from_basis(const bfmmm_config* cfg, int device, const double* y, const double* B, const int64_t* offsets,
                            int P, int band, const double* Pmat, int pen_band, bfmmm_handle** out);
void bfmmm_destroy(bfmmm_handle* h);

/* Chain batches.  The multi-try entry points run 1 + n_try independent chains on the same data
 * (src/UserFunctions.cpp:302-325, sequentially in the reference).  bfmmm_create_batch / bfmmm_create_from_basis_batch make
 * ONE sampler that holds n_chains such chains over one copy of the per-curve statistics: bfmmm_run advances all of them
 * in lockstep (the chain index is a grid dimension of every kernel launch), chain q drawing from RNG chain id
 * `chain + q * stride` (bfmmm_set_chain_id_stride, default 1), so that chain q of a batch is bit-identical to a
 * stand-alone sampler run with that chain id.  bfmmm_select_chain picks the chain that bfmmm_set_state, bfmmm_get_state,
 * bfmmm_init_state, bfmmm_get_chain and bfmmm_debug_get address (default 0).  bfmmm_set_covariates applies to every chain.
 * bfmmm_tempered_transition needs a batch of one chain. */
int bfmmm_create_batch(const bfmmm_config* cfg, int device, const double* y, const double* t, const int64_t* offsets,
                       const double* internal_knots, const double* boundary_knots, int n_chains, bfmmm_handle** out);
int bfmmm_create_from_basis_batch(const bfmmm_config* cfg, int device, const double* y, const double* B, const int64_t* offsets,
                                  int P, int band, const double* Pmat, int pen_band, int n_chains, bfmmm_handle** out);
int bfmmm_select_chain(bfmmm_handle* h, int q);
int bfmmm_n_chains(const bfmmm_handle* h);
int bfmmm_set_chain_id_stride(bfmmm_handle* h, uint32_t stride);

/* Final gather of a multi-GPU multi-try (the reference keeps the best of its 1 + n_try chains, src/UserFunctions.cpp:302-325,
 * :861-885).  handles[g], g < n_handles, live on distinct devices of THIS process, were created with the same configuration
 * and have their best chain selected (bfmmm_select_chain); scores[g] / chain_ids[g] are that chain's score and chain index
 * (NaN score: the device holds no valid chain).  One RCCL communicator over the devices: ncclAllGather of the
 * (score, chain index) pairs, then the winner's chain (state and every chain slot) is sent over xGMI (ncclSend / ncclRecv)
 * into the selected chain of handles[0].  *winner = index of the winning handle (largest score, lowest chain index on
 * ties).  This is the only inter-GPU exchange of the library: nothing is communicated inside a chain. */
int bfmmm_gather_best(bfmmm_handle* const* handles, int n_handles, const double* scores, const int32_t* chain_ids, int* winner);

/* Covariate adjustment (the `X` argument of the reference's entry points, UserFunctions.cpp:176): X is the
 * n_funct x D column-major covariate matrix; covariance_adj != 0 enables the Xi block (BFMMM.h:4602 vs :4067).
 * Call once, right after bfmmm_create.  Adds the state / chain names "eta" (P x D x K), "xi" and "gamma_xi"
 * (K arrays P x D x M), "tau_eta" (K x D), "delta_xi" (K x M x D), "A_xi" (K x 2 x D). */
int bfmmm_set_covariates(bfmmm_handle* h, const double* X, int D, int covariance_adj);

/* Basis matrices "B" returned by the reference's entry points (UserFunctions.cpp:327): the rows of
 * all curves concatenated, each row P doubles (row-major: out[(offsets[i]+l)*P + p] = B_i(l,p)). */
int bfmmm_get_basis(bfmmm_handle* h, double* out, int64_t capacity);

/* Current state (the chain's working slot).  Names and layouts are the reference's
 * (column-major Armadillo objects):
 *   "nu" K x P, "Phi" K x P x M, "chi" n x M, "Z" n x K, "pi" K, "alpha_3" 1, "delta" K x M,
 *   "A" K x 2, "gamma" K x P x M, "tau" K, "sigma_sq" 1 (a variance, as everywhere in the reference). */
int bfmmm_set_state(bfmmm_handle* h, const char* name, const double* values, int64_t count);
int bfmmm_get_state(bfmmm_handle* h, const char* name, double* out, int64_t capacity);

/* Initial states of the chain drivers, drawn from the keyed RNG:
 *   stage 0: BFMMM_Nu_Z   (BFMMM.h:1039-1071)  nu ~ N(0,1), chi = 0, Phi = 0, pi ~ Dir(c), Z_i ~ Dir(100 pi), rest 1
 *   stage 1: BFMMM_Theta  (BFMMM.h:1210-1235)  as stage 0 but chi ~ N(0,1), Phi ~ N(0,1)
 * (the caller then pins Z / nu with bfmmm_set_state as BFMMM.h:1244-1250 does). */
int bfmmm_init_state(bfmmm_handle* h, int stage, uint64_t seed, uint32_t chain);

/* Runs n_iters Gibbs iterations with the updates selected by `mask`, starting at chain iteration
 * `first_iter` (which is also the chain slot written, and the RNG counter word).  `phi_chi_zero`
 * != 0 declares Phi = 0 and chi = 0 (stage 1 of the pipeline, BFMMM.h:1040,1063) so that their
 * directions are skipped.  beta = tempering temperature of the *Tempered kernels (1 = untempered). */
int bfmmm_run(bfmmm_handle* h, uint32_t mask, int first_iter, int n_iters, uint64_t seed, uint32_t chain,
              int phi_chi_zero, double beta);

/* Set-up half of bfmmm_run: captures and instantiates the HIP graphs a run with the same arguments replays (a caller that
 * times bfmmm_run, or needs its first call to return quickly, pays the capture here).  Every graph this call instantiates is
 * also launched a few times ("dry launch", about 10 ms; BFMMM_DRY_LAUNCH=0 / BFMMM_DRY_LAUNCH_MS=x) between a snapshot and a
 * restore of the chains' work state: the first launch of a graph costs the device 13 - 20 us more than later ones, and the
 * clocks of a device that idled through the capture take milliseconds to come up.  The state a later bfmmm_run starts from is
 * bit-identical with and without the call (tests/test_gpu_prepare.py); the chain SLOTS first_iter .. first_iter + 10 hold
 * scratch values until that run rewrites them. */
int bfmmm_prepare_run(bfmmm_handle* h, uint32_t mask, int first_iter, int n_iters, uint64_t seed, uint32_t chain,
                      int phi_chi_zero);

/* Chain iteration i is written to slot i - base (default 0).  The reference keeps r_stored_iters draws in memory and
 * reuses the slots for every on-disk batch (`i % r_stored_iters`, BFMMM.h:1500-1746): a driver that saves batches
 * moves the base to the first iteration of the next batch; the RNG counter word stays the iteration index, so a
 * batched run draws exactly what an unbatched one does. */
int bfmmm_set_slot_base(bfmmm_handle* h, int base);

/* Tempered-transition block of BFMMM_MTT_warm_start (inst/include/BayesFMMM/BFMMM.h:1556-1657, ladder :1452-1460,
 * acceptance CalculateTTAcceptance.h:22-97) for chain iteration `iter`, to be called right after bfmmm_run has
 * produced that iteration: 2 N_t tempered sweeps of the updates in `mask` (temperatures up and down the geometric
 * ladder ending at beta_N_t), then the Metropolis test.  Chain slot `iter` and the working state end up holding the
 * accepted or the original draw; *logA / *accepted report the test.  Functional model without covariates only. */
int bfmmm_tempered_transition(bfmmm_handle* h, uint32_t mask, int iter, int N_t, double beta_N_t, uint64_t seed,
                              uint32_t chain, double* logA, int* accepted);

/* Copies chain draws to the host: slots [0, n_slots) of `name`, laid out as the reference returns
 * them: "nu" K x P x T, "chi" n x M x T, "Z" n x K x T, "pi" K x T, "alpha_3" T, "A" K x 2 x T,
 * "delta" K x M x T, "sigma_sq" T, "tau" T x K, "gamma"/"Phi" T arrays of K x P x M, "loglik" T. */
int bfmmm_get_chain(bfmmm_handle* h, const char* name, int n_slots, double* out, int64_t capacity);

/* Diagnostics for the parity tests: "rec" (n x LREC per-curve statistics), "H", "tvec", "Cmat", "Lmat",
 * "dims" (as doubles).  Returns the number of doubles written through *count. */
int bfmmm_debug_get(bfmmm_handle* h, const char* name, double* out, int64_t capacity, int64_t* count);

/* Timing of the last bfmmm_run: milliseconds between HIP events recorded on the sampler's stream
 * around the whole run and, per kernel family, accumulated over iterations when `profile` was
 * enabled with bfmmm_set_profile (which disables graph replay).
 * names: "total", "curve_z", "pair_gram", "factor", "sweep", "curve_chi", "loglik". */
int bfmmm_set_profile(bfmmm_handle* h, int enable);
int bfmmm_get_timing(bfmmm_handle* h, const char* name, double* ms, int64_t* launches);

/* The per-curve and covariate kernels are also built in exact-shape instances (K, M, D compile-time: DESIGN.md section 5) that
 * the launchers pick when the model's shape is on the list.  0 makes every later launch (of samplers created afterwards) use the
 * general instances instead -- the parity tests run both and compare.  Process-wide; default 1. */
void bfmmm_set_exact_instances(int enable);

const char* bfmmm_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
