// .Call shim: the BayesFMMM R package's compiled entry points over the C ABI of this repository.
//
// Drop this file into src/ of the R package in place of the bodies behind the Rcpp-generated stubs and link with
// -lbfmmm_hip (PKG_LIBS).  It exports the SAME symbols with the SAME arity as src/RcppExports.cpp, so R/RcppExports.R and
// every R-level signature, default and returned list name stay as they are:
//
//   _BayesFMMM_BFMMM_Nu_Z_multiple_try (28 args, RcppExports.cpp:358)   -> bfmmm_BFMMM_Nu_Z_multiple_try
//   _BayesFMMM_BFMMM_Theta_est         (32,      :396)                  -> bfmmm_BFMMM_Theta_est
//   _BayesFMMM_BFMMM_warm_start        (38,      :438)                  -> bfmmm_BFMMM_warm_start
//   _BayesFMMM_BMVMMM_Nu_Z_multiple_try (23, :680), _BMVMMM_Theta_est (27, :713), _BMVMMM_warm_start (33, :750)
//   _BayesFMMM_BHDFMMM_Nu_Z_multiple_try (28, :552), _BHDFMMM_Theta_est (32, :590), _BHDFMMM_warm_start (38, :632)
//   _BayesFMMM_FDIC / FAIC / FBIC (10 args, :174, :194, :214), _BayesFMMM_FLLik (9, :234)  -> bfmmm_FDIC ... (bfmmm_post.h)
//   _BayesFMMM_ConditionalPredictiveOrdinates (11), MVLLik (5), MVDIC / MVAIC / MVBIC (6), SigmaCI (4), ZCI (5), FMeanCI (13),
//   FCovCI (15), MVMeanCI (6), HDFMeanCI (12), HDFCovCI (14), MVCovCI (8), FSamplePaths (11)      -> bfmmm_post.h
//   _BayesFMMM_ReadVec / ReadMat / ReadCube / ReadFieldCube / ReadFieldMat / ReadFieldVec (1 arg each, :486-541)
//
// All 33 entries of the reference's CallEntries[] (src/RcppExports.cpp:794-835) are registered here.
// It is pure marshalling (no arithmetic): R lists of numeric vectors become CSR arrays, Rcpp::List arguments become
// bfmmm_result objects, results become named R lists with the reference's shapes.  It needs <Rinternals.h> and is NOT
// built by this repository's Makefile (the image has no R); INTEGRATION.md explains the build line.
#include <R.h>
#include <Rinternals.h>
#include <R_ext/Random.h>
#include <R_ext/Rdynload.h>

#include <cstring>
#include <deque>
#include <stdexcept>
#include <string>
#include <vector>

#include "bfmmm_entry.h"
#include "bfmmm_post.h"

namespace {

// Errors: the library reports through bfmmm_entry_last_error(); the shim raises them as C++ exceptions inside a
// SHIM_BEGIN / SHIM_END pair, so that every C++ object of the entry point (the CSR copies, bfmmm_result handles held by
// Owned) is destroyed BEFORE Rf_error's longjmp -- the role of Rcpp's BEGIN_RCPP / END_RCPP (RcppExports.cpp:359, :392).
#define SHIM_BEGIN char* shim_msg_ = NULL; try {
#define SHIM_END } catch (const std::exception& e_) { const size_t n_ = std::strlen(e_.what()) + 1; shim_msg_ = R_alloc(n_, 1); std::memcpy(shim_msg_, e_.what(), n_); } \
  Rf_error("%s", shim_msg_ ? shim_msg_ : "unknown error"); return R_NilValue;
[[noreturn]] void lib_error() { throw std::runtime_error(bfmmm_entry_last_error()); }

struct Owned {                                          // a bfmmm_result freed on every path
  bfmmm_result* r = NULL;
  ~Owned() { if (r) bfmmm_result_free(r); }
};

// a numeric R vector / matrix as doubles: the reference's Rcpp::NumericMatrix / arma conversions coerce integer and logical
// input (e.g. a 0/1 covariate matrix), REAL() alone would raise an R error on it
struct Num {
  std::vector<double> own;
  const double* p = NULL;
  R_xlen_t len = 0;
  int nrow = 0, ncol = 0;
  explicit Num(SEXP x) {
    if (x == R_NilValue) return;
    len = Rf_xlength(x);
    nrow = Rf_isMatrix(x) ? Rf_nrows(x) : (int)len;
    ncol = Rf_isMatrix(x) ? Rf_ncols(x) : 1;
    if (TYPEOF(x) == REALSXP) { p = REAL(x); return; }
    if (TYPEOF(x) != INTSXP && TYPEOF(x) != LGLSXP) throw std::runtime_error("a numeric argument is required");
    SEXP c = PROTECT(Rf_coerceVector(x, REALSXP));
    own.assign(REAL(c), REAL(c) + len);
    UNPROTECT(1);
    p = own.data();
  }
};

// the coerced numeric arguments of one call (stable addresses)
struct Inputs {
  std::deque<Num> nums;
  const double* num(SEXP x) { nums.emplace_back(x); return nums.back().p; }
};

struct Ragged { std::vector<double> v; std::vector<int64_t> off; };

Ragged flatten(SEXP lst) {                              // R list of numeric vectors -> CSR
  Ragged r;
  r.off.push_back(0);
  if (TYPEOF(lst) != VECSXP) throw std::runtime_error("a list of numeric vectors is required");
  for (R_xlen_t i = 0; i < Rf_xlength(lst); ++i) {
    const Num e(VECTOR_ELT(lst, i));
    r.v.insert(r.v.end(), e.p, e.p + e.len);
    r.off.push_back((int64_t)r.v.size());
  }
  return r;
}

uint64_t seed_from_R() {                                // one pair of uniforms from R's stream: set.seed() stays meaningful
  GetRNGstate();
  const uint64_t hi = (uint64_t)(unif_rand() * 4294967296.0), lo = (uint64_t)(unif_rand() * 4294967296.0);
  PutRNGstate();
  return (hi << 32) | lo;
}

std::vector<int64_t> dims_of(SEXP x) {
  std::vector<int64_t> d;
  SEXP ds = Rf_getAttrib(x, R_DimSymbol);
  if (ds == R_NilValue) d.push_back((int64_t)Rf_xlength(x));
  else for (int k = 0; k < Rf_length(ds); ++k) d.push_back(INTEGER(ds)[k]);
  return d;
}

// Rcpp::List of a previous stage -> bfmmm_result.  Numeric arrays go in as they are; the list-valued elements of the
// reference ("gamma", "Phi", "eta", "delta_xi", "A_xi": T arrays; "xi", "gamma_xi": a T x K list-matrix of cubes) are
// stacked into one array with the slot last (xi / gamma_xi: slot t = the K cubes of that iteration, k fastest).
bfmmm_result* list_to_result(SEXP lst) {
  bfmmm_result* r = bfmmm_result_create();
  SEXP nm = Rf_getAttrib(lst, R_NamesSymbol);
  for (R_xlen_t i = 0; i < Rf_xlength(lst); ++i) {
    SEXP e = VECTOR_ELT(lst, i);
    const char* name = CHAR(STRING_ELT(nm, i));
    if (TYPEOF(e) == REALSXP) {
      std::vector<int64_t> d = dims_of(e);
      bfmmm_result_set(r, name, REAL(e), (int64_t)Rf_xlength(e), d.data(), (int)d.size());
    } else if (TYPEOF(e) == VECSXP && Rf_xlength(e) > 0 && TYPEOF(VECTOR_ELT(e, 0)) == REALSXP &&
               std::strcmp(name, "B") != 0 && std::strcmp(name, "B_obs") != 0) {
      SEXP fd = Rf_getAttrib(e, R_DimSymbol);            // T x K list-matrix (xi, gamma_xi) or a plain list of T arrays
      const int64_t T = (fd == R_NilValue) ? (int64_t)Rf_xlength(e) : INTEGER(fd)[0];
      const int64_t Kc = (fd == R_NilValue) ? 1 : INTEGER(fd)[1];
      SEXP first = VECTOR_ELT(e, 0);
      const int64_t len = (int64_t)Rf_xlength(first);
      std::vector<double> all((size_t)(len * T * Kc));
      for (int64_t t = 0; t < T; ++t)
        for (int64_t k = 0; k < Kc; ++k)
          std::memcpy(all.data() + (size_t)((t * Kc + k) * len), REAL(VECTOR_ELT(e, t + T * k)), sizeof(double) * (size_t)len);
      std::vector<int64_t> d = dims_of(first);
      if (Kc > 1) d.push_back(Kc);
      d.push_back(T);
      bfmmm_result_set(r, name, all.data(), (int64_t)all.size(), d.data(), (int)d.size());
    }
  }
  return r;
}

SEXP array_of(const double* data, const int64_t* dims, int nd) {
  SEXP ds = PROTECT(Rf_allocVector(INTSXP, nd));
  R_xlen_t len = 1;
  for (int k = 0; k < nd; ++k) { INTEGER(ds)[k] = (int)dims[k]; len *= (R_xlen_t)dims[k]; }
  SEXP a = PROTECT(Rf_allocVector(REALSXP, len));
  if (len > 0) std::memcpy(REAL(a), data, sizeof(double) * (size_t)len);
  if (nd > 1) Rf_setAttrib(a, R_DimSymbol, ds);
  UNPROTECT(2);
  return a;
}

bool is_slot_list(const std::string& n) {     // returned as a list of T arrays by the reference (arma::field<arma::cube>)
  return n == "gamma" || n == "Phi" || n == "eta" || n == "delta_xi" || n == "A_xi";
}

// bfmmm_result -> the named list the reference returns (UserFunctions.cpp:327-336, 887-897, 1094-1110)
SEXP result_to_list(const bfmmm_result* r, const std::vector<int64_t>* offsets) {
  const int n = bfmmm_result_count(r);
  SEXP out = PROTECT(Rf_allocVector(VECSXP, n));
  SEXP names = PROTECT(Rf_allocVector(STRSXP, n));
  for (int i = 0; i < n; ++i) {
    const char* name = bfmmm_result_name(r, i);
    const double* data; int64_t cnt; const int64_t* dims; int nd;
    bfmmm_result_get(r, name, &data, &cnt, &dims, &nd);
    const std::string s(name);
    SEXP el;
    if ((s == "B" || s == "B_obs") && offsets) {
      // all basis rows, row-major (row l of curve i at (off[i] + l) * P): list of n_i x P column-major matrices
      const int64_t P = dims[1], nf = (int64_t)offsets->size() - 1;
      el = PROTECT(Rf_allocVector(VECSXP, nf));
      for (int64_t c = 0; c < nf; ++c) {
        const int64_t ni = (*offsets)[c + 1] - (*offsets)[c];
        SEXP m = PROTECT(Rf_allocMatrix(REALSXP, (int)ni, (int)P));
        for (int64_t l = 0; l < ni; ++l)
          for (int64_t p = 0; p < P; ++p) REAL(m)[l + ni * p] = data[((*offsets)[c] + l) * P + p];
        SET_VECTOR_ELT(el, c, m);
        UNPROTECT(1);
      }
    } else if (s == "cov_trace" && nd == 4) {            // FCovCI / HDFCovCI / MVCovCI with X: the reference's list of n_x cubes
      const int64_t W = dims[3], len = cnt / W;
      el = PROTECT(Rf_allocVector(VECSXP, W));
      for (int64_t w = 0; w < W; ++w) SET_VECTOR_ELT(el, w, array_of(data + w * len, dims, 3));
    } else if (is_slot_list(s)) {
      const int64_t T = dims[nd - 1], len = cnt / T;
      el = PROTECT(Rf_allocVector(VECSXP, T));
      for (int64_t t = 0; t < T; ++t) SET_VECTOR_ELT(el, t, array_of(data + t * len, dims, nd - 1));
    } else if (s == "xi" || s == "gamma_xi") {           // P x D x M x K x T -> T x K list-matrix of P x D x M cubes
      const int64_t T = dims[nd - 1], Kc = dims[nd - 2], len = cnt / (T * Kc);
      el = PROTECT(Rf_allocVector(VECSXP, T * Kc));
      for (int64_t t = 0; t < T; ++t)
        for (int64_t k = 0; k < Kc; ++k) SET_VECTOR_ELT(el, t + T * k, array_of(data + (t * Kc + k) * len, dims, nd - 2));
      SEXP fd = PROTECT(Rf_allocVector(INTSXP, 2));
      INTEGER(fd)[0] = (int)T; INTEGER(fd)[1] = (int)Kc;
      Rf_setAttrib(el, R_DimSymbol, fd);
      UNPROTECT(1);
    } else {
      el = PROTECT(array_of(data, dims, nd));
    }
    SET_VECTOR_ELT(out, i, el);
    SET_STRING_ELT(names, i, Rf_mkChar(name));
    UNPROTECT(1);
  }
  Rf_setAttrib(out, R_NamesSymbol, names);
  UNPROTECT(2);
  return out;
}

// the hyper-parameters every entry point shares, in the order of the reference's signatures
struct Hyper {
  SEXP c, b, nu_1, alpha1l, alpha2l, beta1l, beta2l, a_Z_PM, a_pi_PM, var_alpha3, var_epsilon1, var_epsilon2, alpha_nu, beta_nu,
       alpha_eta, beta_eta, alpha_0, beta_0;
};

void set_hyper(bfmmm_entry_args& a, const Hyper& h, Inputs& in) {
  a.c = (h.c == R_NilValue) ? NULL : in.num(h.c);
  if (h.c != R_NilValue && Rf_length(h.c) != a.K) throw std::runtime_error("number of elements of the vector 'c' must be equal to K");
  a.b = Rf_asReal(h.b);
  if (h.nu_1 != R_NilValue) a.nu_1 = Rf_asReal(h.nu_1);
  a.alpha1l = Rf_asReal(h.alpha1l); a.alpha2l = Rf_asReal(h.alpha2l);
  a.beta1l = Rf_asReal(h.beta1l); a.beta2l = Rf_asReal(h.beta2l);
  a.a_Z_PM = Rf_asReal(h.a_Z_PM); a.a_pi_PM = Rf_asReal(h.a_pi_PM); a.var_alpha3 = Rf_asReal(h.var_alpha3);
  a.var_epsilon1 = Rf_asReal(h.var_epsilon1); a.var_epsilon2 = Rf_asReal(h.var_epsilon2);
  a.alpha_nu = Rf_asReal(h.alpha_nu); a.beta_nu = Rf_asReal(h.beta_nu);
  a.alpha_eta = Rf_asReal(h.alpha_eta); a.beta_eta = Rf_asReal(h.beta_eta);
  a.alpha_0 = Rf_asReal(h.alpha_0); a.beta_0 = Rf_asReal(h.beta_0);
}

void set_X(bfmmm_entry_args& a, SEXP X, SEXP covariance_adj, Inputs& in) {
  if (X != R_NilValue) {
    if (Rf_nrows(X) != a.n_funct) throw std::runtime_error("'X' must be have 'n_funct' number of rows");
    a.X = in.num(X); a.D = Rf_ncols(X);
  }
  if (covariance_adj != R_NilValue) a.covariance_adj = Rf_asLogical(covariance_adj) ? 1 : 0;
}

// (n_funct is checked against the lists: the library trusts offsets[n_funct])
void check_n_funct(const Ragged& Y, const Ragged& tm, int n_funct) {
  if ((int64_t)Y.off.size() - 1 != n_funct || (int64_t)tm.off.size() - 1 != n_funct)
    throw std::runtime_error("'Y' and 'time' must be lists of 'n_funct' elements");
  // one time point (high-dimensional model: one row of `dim` coordinates) per observation
  const int64_t fac = Y.v.empty() ? 1 : (int64_t)(tm.v.size() / Y.v.size());
  for (size_t i = 0; i + 1 < Y.off.size(); ++i)
    if (tm.off[i + 1] - tm.off[i] != fac * (Y.off[i + 1] - Y.off[i]))
      throw std::runtime_error("every element of 'time' must hold the time points of the matching element of 'Y'");
}

void set_functional(bfmmm_entry_args& a, const Ragged& Y, const Ragged& tm, SEXP n_funct, SEXP basis_degree, SEXP n_eigen,
                    SEXP boundary_knots, SEXP internal_knots, Inputs& in) {
  a.n_funct = Rf_asInteger(n_funct);
  check_n_funct(Y, tm, a.n_funct);
  if (Rf_length(boundary_knots) != 2) throw std::runtime_error("'boundary_knots' must have two elements");
  a.y = Y.v.data(); a.t = tm.v.data(); a.offsets = Y.off.data();
  a.basis_degree = Rf_asInteger(basis_degree); a.n_eigen = Rf_asInteger(n_eigen);
  a.n_internal_knots = Rf_length(internal_knots);
  a.boundary_knots = in.num(boundary_knots); a.internal_knots = in.num(internal_knots);
}

// high-dimensional model: `time` is a list of n_i x dim matrices (flattened as they are: column-major per curve),
// `basis_degree` an arma::vec, `boundary_knots` a dim x 2 matrix (column-major in R -> row-major for the ABI),
// `internal_knots` a list of vectors (UserFunctions.cpp:2519-2545)
struct HDArgs { std::vector<int32_t> deg, nint; std::vector<double> bk; Ragged ik; };

void set_hd(bfmmm_entry_args& a, HDArgs& h, const Ragged& Y, const Ragged& tm, SEXP n_funct, SEXP basis_degree, SEXP n_eigen,
            SEXP boundary_knots, SEXP internal_knots, Inputs&) {
  const int dim = Rf_length(basis_degree);
  a.n_funct = Rf_asInteger(n_funct); a.n_eigen = Rf_asInteger(n_eigen); a.dim = dim;
  check_n_funct(Y, tm, a.n_funct);
  a.y = Y.v.data(); a.t = tm.v.data(); a.offsets = Y.off.data();
  h.ik = flatten(internal_knots);
  const Num deg(basis_degree), bk(boundary_knots);
  if ((int)h.ik.off.size() - 1 != dim || bk.len != 2 * (R_xlen_t)dim) throw std::runtime_error("'basis_degree', 'boundary_knots' (dim x 2) and 'internal_knots' disagree on the dimension");
  for (int j = 0; j < dim; ++j) {
    h.deg.push_back((int32_t)deg.p[j]);
    h.nint.push_back((int32_t)(h.ik.off[j + 1] - h.ik.off[j]));
    h.bk.push_back(bk.p[j]);
    h.bk.push_back(bk.p[j + dim]);
  }
  a.basis_degree_hd = h.deg.data(); a.n_internal_hd = h.nint.data();
  a.boundary_knots = h.bk.data(); a.internal_knots = h.ik.v.data();
}

void set_multivariate(bfmmm_entry_args& a, SEXP Y, SEXP n_eigen, Inputs& in) {
  a.model = 1;                                           // BFMMM_MODEL_MULTIVARIATE
  if (!Rf_isMatrix(Y)) throw std::runtime_error("'Y' must be a matrix");
  a.y = in.num(Y); a.n_funct = Rf_nrows(Y); a.P = Rf_ncols(Y);
  a.n_eigen = Rf_asInteger(n_eigen);
}

// R's interrupt check between device batches (the reference calls Rcpp::checkUserInterrupt() in its loop)
static void check_interrupt_fn(void*) { R_CheckUserInterrupt(); }
static int progress_from_R(int32_t, double, void*) { return R_ToplevelExec(check_interrupt_fn, NULL) ? 0 : 1; }

void set_warm(bfmmm_entry_args& a, SEXP dir, SEXP thinning_num, SEXP beta_N_t, SEXP N_t, SEXP n_temp_trans, SEXP r_stored_iters) {
  a.dir = (dir == R_NilValue) ? NULL : CHAR(STRING_ELT(dir, 0));
  a.thinning_num = Rf_asReal(thinning_num); a.beta_N_t = Rf_asReal(beta_N_t);
  a.N_t = Rf_asInteger(N_t); a.n_temp_trans = Rf_asInteger(n_temp_trans); a.r_stored_iters = Rf_asInteger(r_stored_iters);
  a.progress_every = 100; a.progress_cb = progress_from_R;
}

// (p1 / p2: the previous-stage results the call was given; their owners are the callers' Owned objects)
SEXP finish(int rc, bfmmm_result* r, const std::vector<int64_t>* offsets, bfmmm_result* p1 = NULL, bfmmm_result* p2 = NULL) {
  (void)p1; (void)p2;
  Owned o3;
  o3.r = r;
  if (rc) lib_error();                                   // (r is NULL on failure)
  return result_to_list(r, offsets);
}

SEXP read_plain(SEXP file, bool as_vector) {
  SHIM_BEGIN
  Owned o;
  if (bfmmm_arma_read(CHAR(STRING_ELT(file, 0)), &o.r)) lib_error();
  const double* data; int64_t cnt; const int64_t* dims; int nd;
  bfmmm_result_get(o.r, "value", &data, &cnt, &dims, &nd);
  const int64_t one = cnt;
  return as_vector ? array_of(data, &one, 1) : array_of(data, dims, nd);
  SHIM_END
}

SEXP read_field(SEXP file, bool as_vectors) {             // arma::field -> list-matrix (n_rows x n_cols)
  SHIM_BEGIN
  Owned o;
  bfmmm_result*& r = o.r;
  if (bfmmm_arma_read_field(CHAR(STRING_ELT(file, 0)), &r)) lib_error();
  const double* fd; int64_t cnt; const int64_t* dims; int nd;
  bfmmm_result_get(r, "field_dims", &fd, &cnt, &dims, &nd);
  const int64_t nr = (int64_t)fd[0], nc = (int64_t)fd[1];
  SEXP out = PROTECT(Rf_allocVector(VECSXP, nr * nc));
  for (int64_t e = 0; e < nr * nc; ++e) {
    const double* data;
    bfmmm_result_get(r, std::to_string(e).c_str(), &data, &cnt, &dims, &nd);
    const int64_t one = cnt;
    SET_VECTOR_ELT(out, e, as_vectors ? array_of(data, &one, 1) : array_of(data, dims, nd));
  }
  SEXP d2 = PROTECT(Rf_allocVector(INTSXP, 2));
  INTEGER(d2)[0] = (int)nr; INTEGER(d2)[1] = (int)nc;
  Rf_setAttrib(out, R_DimSymbol, d2);
  UNPROTECT(2);
  return out;
  SHIM_END
}

}  // namespace

#define H18 SEXP c, SEXP b, SEXP alpha1l, SEXP alpha2l, SEXP beta1l, SEXP beta2l, SEXP a_Z_PM, SEXP a_pi_PM, SEXP var_alpha3, \
            SEXP var_epsilon1, SEXP var_epsilon2, SEXP alpha_nu, SEXP beta_nu, SEXP alpha_eta, SEXP beta_eta, SEXP alpha_0, SEXP beta_0
#define H19 SEXP c, SEXP b, SEXP nu_1, SEXP alpha1l, SEXP alpha2l, SEXP beta1l, SEXP beta2l, SEXP a_Z_PM, SEXP a_pi_PM, \
            SEXP var_alpha3, SEXP var_epsilon1, SEXP var_epsilon2, SEXP alpha_nu, SEXP beta_nu, SEXP alpha_eta, SEXP beta_eta, \
            SEXP alpha_0, SEXP beta_0
#define HY18 Hyper{c, b, R_NilValue, alpha1l, alpha2l, beta1l, beta2l, a_Z_PM, a_pi_PM, var_alpha3, var_epsilon1, var_epsilon2, \
                   alpha_nu, beta_nu, alpha_eta, beta_eta, alpha_0, beta_0}
#define HY19 Hyper{c, b, nu_1, alpha1l, alpha2l, beta1l, beta2l, a_Z_PM, a_pi_PM, var_alpha3, var_epsilon1, var_epsilon2, \
                   alpha_nu, beta_nu, alpha_eta, beta_eta, alpha_0, beta_0}

extern "C" {

// ---- functional model ----------------------------------------------------------------------------------------
SEXP _BayesFMMM_BFMMM_Nu_Z_multiple_try(SEXP tot_mcmc_iters, SEXP n_try, SEXP K, SEXP Y, SEXP time, SEXP n_funct, SEXP basis_degree,
                                        SEXP n_eigen, SEXP boundary_knots, SEXP internal_knots, SEXP X, H18) {
  SHIM_BEGIN
  Inputs in;
  const Ragged y = flatten(Y), t = flatten(time);
  bfmmm_entry_args a;
  bfmmm_entry_defaults(&a, 0);
  a.tot_mcmc_iters = Rf_asInteger(tot_mcmc_iters); a.n_try = Rf_asInteger(n_try); a.K = Rf_asInteger(K);
  set_functional(a, y, t, n_funct, basis_degree, n_eigen, boundary_knots, internal_knots, in);
  set_X(a, X, R_NilValue, in);
  set_hyper(a, HY18, in);
  a.seed = seed_from_R();
  bfmmm_result* r = NULL;
  return finish(bfmmm_BFMMM_Nu_Z_multiple_try(&a, &r), r, &y.off);
  SHIM_END
}

SEXP _BayesFMMM_BFMMM_Theta_est(SEXP tot_mcmc_iters, SEXP n_try, SEXP K, SEXP Y, SEXP time, SEXP n_funct, SEXP basis_degree,
                                SEXP n_eigen, SEXP boundary_knots, SEXP internal_knots, SEXP multiple_try, SEXP X,
                                SEXP burnin_prop, H19, SEXP covariance_adj) {
  SHIM_BEGIN
  Inputs in;
  const Ragged y = flatten(Y), t = flatten(time);
  bfmmm_entry_args a;
  bfmmm_entry_defaults(&a, 1);
  a.tot_mcmc_iters = Rf_asInteger(tot_mcmc_iters); a.n_try = Rf_asInteger(n_try); a.K = Rf_asInteger(K);
  a.burnin_prop = Rf_asReal(burnin_prop);
  set_functional(a, y, t, n_funct, basis_degree, n_eigen, boundary_knots, internal_knots, in);
  set_X(a, X, covariance_adj, in);
  set_hyper(a, HY19, in);
  a.seed = seed_from_R();
  Owned omt; omt.r = list_to_result(multiple_try);       // owned from the moment it exists (a later conversion may throw)
  bfmmm_result* mt = omt.r;
  bfmmm_result* r = NULL;
  return finish(bfmmm_BFMMM_Theta_est(&a, mt, &r), r, &y.off, mt);
  SHIM_END
}

SEXP _BayesFMMM_BFMMM_warm_start(SEXP tot_mcmc_iters, SEXP K, SEXP Y, SEXP time, SEXP n_funct, SEXP basis_degree, SEXP n_eigen,
                                 SEXP boundary_knots, SEXP internal_knots, SEXP multiple_try, SEXP theta_est, SEXP X,
                                 SEXP burnin_prop, SEXP dir, SEXP thinning_num, SEXP beta_N_t, SEXP N_t, SEXP n_temp_trans,
                                 SEXP r_stored_iters, H19, SEXP covariance_adj) {
  SHIM_BEGIN
  Inputs in;
  const Ragged y = flatten(Y), t = flatten(time);
  bfmmm_entry_args a;
  bfmmm_entry_defaults(&a, 2);
  a.tot_mcmc_iters = Rf_asInteger(tot_mcmc_iters); a.K = Rf_asInteger(K);
  a.burnin_prop = Rf_asReal(burnin_prop);
  set_functional(a, y, t, n_funct, basis_degree, n_eigen, boundary_knots, internal_knots, in);
  set_X(a, X, covariance_adj, in);
  set_warm(a, dir, thinning_num, beta_N_t, N_t, n_temp_trans, r_stored_iters);
  set_hyper(a, HY19, in);
  a.seed = seed_from_R();
  Owned omt; omt.r = list_to_result(multiple_try);       // owned from the moment it exists (a later conversion may throw)
  bfmmm_result* mt = omt.r;
  Owned ote; ote.r = list_to_result(theta_est);
  bfmmm_result* te = ote.r;
  bfmmm_result* r = NULL;
  return finish(bfmmm_BFMMM_warm_start(&a, mt, te, &r), r, &y.off, mt, te);
  SHIM_END
}

// ---- multivariate model ----------------------------------------------------------------------------------------
SEXP _BayesFMMM_BMVMMM_Nu_Z_multiple_try(SEXP tot_mcmc_iters, SEXP n_try, SEXP K, SEXP Y, SEXP n_eigen, SEXP X, H18) {
  SHIM_BEGIN
  Inputs in;
  bfmmm_entry_args a;
  bfmmm_entry_defaults(&a, 3);
  a.tot_mcmc_iters = Rf_asInteger(tot_mcmc_iters); a.n_try = Rf_asInteger(n_try); a.K = Rf_asInteger(K);
  set_multivariate(a, Y, n_eigen, in);
  set_X(a, X, R_NilValue, in);
  set_hyper(a, HY18, in);
  a.seed = seed_from_R();
  bfmmm_result* r = NULL;
  return finish(bfmmm_BMVMMM_Nu_Z_multiple_try(&a, &r), r, NULL);
  SHIM_END
}

SEXP _BayesFMMM_BMVMMM_Theta_est(SEXP tot_mcmc_iters, SEXP n_try, SEXP K, SEXP Y, SEXP n_eigen, SEXP multiple_try, SEXP X,
                                 SEXP burnin_prop, H19, SEXP covariance_adj) {
  SHIM_BEGIN
  Inputs in;
  bfmmm_entry_args a;
  bfmmm_entry_defaults(&a, 4);
  a.tot_mcmc_iters = Rf_asInteger(tot_mcmc_iters); a.n_try = Rf_asInteger(n_try); a.K = Rf_asInteger(K);
  a.burnin_prop = Rf_asReal(burnin_prop);
  set_multivariate(a, Y, n_eigen, in);
  set_X(a, X, covariance_adj, in);
  set_hyper(a, HY19, in);
  a.seed = seed_from_R();
  Owned omt; omt.r = list_to_result(multiple_try);       // owned from the moment it exists (a later conversion may throw)
  bfmmm_result* mt = omt.r;
  bfmmm_result* r = NULL;
  return finish(bfmmm_BMVMMM_Theta_est(&a, mt, &r), r, NULL, mt);
  SHIM_END
}

SEXP _BayesFMMM_BMVMMM_warm_start(SEXP tot_mcmc_iters, SEXP K, SEXP Y, SEXP n_eigen, SEXP multiple_try, SEXP theta_est, SEXP X,
                                  SEXP burnin_prop, SEXP dir, SEXP thinning_num, SEXP beta_N_t, SEXP N_t, SEXP n_temp_trans,
                                  SEXP r_stored_iters, H19, SEXP covariance_adj) {
  SHIM_BEGIN
  Inputs in;
  bfmmm_entry_args a;
  bfmmm_entry_defaults(&a, 5);
  a.tot_mcmc_iters = Rf_asInteger(tot_mcmc_iters); a.K = Rf_asInteger(K);
  a.burnin_prop = Rf_asReal(burnin_prop);
  set_multivariate(a, Y, n_eigen, in);
  set_X(a, X, covariance_adj, in);
  set_warm(a, dir, thinning_num, beta_N_t, N_t, n_temp_trans, r_stored_iters);
  set_hyper(a, HY19, in);
  a.seed = seed_from_R();
  Owned omt; omt.r = list_to_result(multiple_try);       // owned from the moment it exists (a later conversion may throw)
  bfmmm_result* mt = omt.r;
  Owned ote; ote.r = list_to_result(theta_est);
  bfmmm_result* te = ote.r;
  bfmmm_result* r = NULL;
  return finish(bfmmm_BMVMMM_warm_start(&a, mt, te, &r), r, NULL, mt, te);
  SHIM_END
}

// ---- high-dimensional functional model (RcppExports.cpp:552, :590, :632) -----------------------------------------
SEXP _BayesFMMM_BHDFMMM_Nu_Z_multiple_try(SEXP tot_mcmc_iters, SEXP n_try, SEXP K, SEXP Y, SEXP time, SEXP n_funct, SEXP basis_degree,
                                          SEXP n_eigen, SEXP boundary_knots, SEXP internal_knots, SEXP X, H18) {
  SHIM_BEGIN
  Inputs in;
  const Ragged y = flatten(Y), t = flatten(time);
  bfmmm_entry_args a;
  HDArgs h;
  bfmmm_entry_defaults(&a, 0);
  a.tot_mcmc_iters = Rf_asInteger(tot_mcmc_iters); a.n_try = Rf_asInteger(n_try); a.K = Rf_asInteger(K);
  set_hd(a, h, y, t, n_funct, basis_degree, n_eigen, boundary_knots, internal_knots, in);
  set_X(a, X, R_NilValue, in);
  set_hyper(a, HY18, in);
  a.seed = seed_from_R();
  bfmmm_result* r = NULL;
  return finish(bfmmm_BHDFMMM_Nu_Z_multiple_try(&a, &r), r, &y.off);
  SHIM_END
}

SEXP _BayesFMMM_BHDFMMM_Theta_est(SEXP tot_mcmc_iters, SEXP n_try, SEXP K, SEXP Y, SEXP time, SEXP n_funct, SEXP basis_degree,
                                  SEXP n_eigen, SEXP boundary_knots, SEXP internal_knots, SEXP multiple_try, SEXP X,
                                  SEXP burnin_prop, H19, SEXP covariance_adj) {
  SHIM_BEGIN
  Inputs in;
  const Ragged y = flatten(Y), t = flatten(time);
  bfmmm_entry_args a;
  HDArgs h;
  bfmmm_entry_defaults(&a, 1);
  a.tot_mcmc_iters = Rf_asInteger(tot_mcmc_iters); a.n_try = Rf_asInteger(n_try); a.K = Rf_asInteger(K);
  a.burnin_prop = Rf_asReal(burnin_prop);
  set_hd(a, h, y, t, n_funct, basis_degree, n_eigen, boundary_knots, internal_knots, in);
  set_X(a, X, covariance_adj, in);
  set_hyper(a, HY19, in);
  a.seed = seed_from_R();
  Owned omt; omt.r = list_to_result(multiple_try);       // owned from the moment it exists (a later conversion may throw)
  bfmmm_result* mt = omt.r;
  bfmmm_result* r = NULL;
  return finish(bfmmm_BHDFMMM_Theta_est(&a, mt, &r), r, &y.off, mt);
  SHIM_END
}

SEXP _BayesFMMM_BHDFMMM_warm_start(SEXP tot_mcmc_iters, SEXP K, SEXP Y, SEXP time, SEXP n_funct, SEXP basis_degree, SEXP n_eigen,
                                   SEXP boundary_knots, SEXP internal_knots, SEXP multiple_try, SEXP theta_est, SEXP X,
                                   SEXP burnin_prop, SEXP dir, SEXP thinning_num, SEXP beta_N_t, SEXP N_t, SEXP n_temp_trans,
                                   SEXP r_stored_iters, H19, SEXP covariance_adj) {
  SHIM_BEGIN
  Inputs in;
  const Ragged y = flatten(Y), t = flatten(time);
  bfmmm_entry_args a;
  HDArgs h;
  bfmmm_entry_defaults(&a, 2);
  a.tot_mcmc_iters = Rf_asInteger(tot_mcmc_iters); a.K = Rf_asInteger(K);
  a.burnin_prop = Rf_asReal(burnin_prop);
  set_hd(a, h, y, t, n_funct, basis_degree, n_eigen, boundary_knots, internal_knots, in);
  set_X(a, X, covariance_adj, in);
  set_warm(a, dir, thinning_num, beta_N_t, N_t, n_temp_trans, r_stored_iters);
  set_hyper(a, HY19, in);
  a.seed = seed_from_R();
  Owned omt; omt.r = list_to_result(multiple_try);       // owned from the moment it exists (a later conversion may throw)
  bfmmm_result* mt = omt.r;
  Owned ote; ote.r = list_to_result(theta_est);
  bfmmm_result* te = ote.r;
  bfmmm_result* r = NULL;
  return finish(bfmmm_BHDFMMM_warm_start(&a, mt, te, &r), r, &y.off, mt, te);
  SHIM_END
}

// ---- likelihood-based post-processing (RcppExports.cpp:174, :194, :214, :234) -------------------------------------
static void set_post(bfmmm_post_args& a, const Ragged& y, const Ragged& t, SEXP dir, SEXP n_files, SEXP basis_degree,
                     SEXP boundary_knots, SEXP internal_knots, SEXP X, SEXP cov_adj, Inputs& in) {
  bfmmm_post_defaults(&a);
  a.dir = CHAR(STRING_ELT(dir, 0)); a.n_files = Rf_asInteger(n_files); a.basis_degree = Rf_asInteger(basis_degree);
  if (Rf_length(boundary_knots) != 2) throw std::runtime_error("'boundary_knots' must have two elements");
  a.boundary_knots = in.num(boundary_knots); a.internal_knots = in.num(internal_knots); a.n_internal_knots = Rf_length(internal_knots);
  if (y.off.size() != t.off.size() || y.v.size() != t.v.size()) throw std::runtime_error("'Y' and 'time' must have the same shape");
  a.n_funct = (int32_t)(y.off.size() - 1); a.t = t.v.data(); a.y = y.v.data(); a.offsets = y.off.data();
  if (X != R_NilValue) { a.X = in.num(X); a.D = Rf_ncols(X); }      // (the library checks its rows / columns against the saved draws)
  a.cov_adj = Rf_asLogical(cov_adj) ? 1 : 0;
}

static SEXP post_scalar(int (*fn)(const bfmmm_post_args*, double*), SEXP dir, SEXP n_files, SEXP basis_degree, SEXP boundary_knots,
                        SEXP internal_knots, SEXP time, SEXP Y, SEXP burnin_prop, SEXP X, SEXP cov_adj) {
  SHIM_BEGIN
  Inputs in;
  const Ragged y = flatten(Y), t = flatten(time);
  bfmmm_post_args a;
  set_post(a, y, t, dir, n_files, basis_degree, boundary_knots, internal_knots, X, cov_adj, in);
  a.burnin_prop = Rf_asReal(burnin_prop);
  double v = 0.0;
  if (fn(&a, &v)) lib_error();
  return Rf_ScalarReal(v);
  SHIM_END
}

SEXP _BayesFMMM_FDIC(SEXP dir, SEXP n_files, SEXP basis_degree, SEXP boundary_knots, SEXP internal_knots, SEXP time, SEXP Y,
                     SEXP burnin_prop, SEXP X, SEXP cov_adj) {
  SHIM_BEGIN
  Inputs in;
  return post_scalar(bfmmm_FDIC, dir, n_files, basis_degree, boundary_knots, internal_knots, time, Y, burnin_prop, X, cov_adj);
  SHIM_END
}
SEXP _BayesFMMM_FAIC(SEXP dir, SEXP n_files, SEXP basis_degree, SEXP boundary_knots, SEXP internal_knots, SEXP time, SEXP Y,
                     SEXP burnin_prop, SEXP X, SEXP cov_adj) {
  SHIM_BEGIN
  Inputs in;
  return post_scalar(bfmmm_FAIC, dir, n_files, basis_degree, boundary_knots, internal_knots, time, Y, burnin_prop, X, cov_adj);
  SHIM_END
}
SEXP _BayesFMMM_FBIC(SEXP dir, SEXP n_files, SEXP basis_degree, SEXP boundary_knots, SEXP internal_knots, SEXP time, SEXP Y,
                     SEXP burnin_prop, SEXP X, SEXP cov_adj) {
  SHIM_BEGIN
  Inputs in;
  return post_scalar(bfmmm_FBIC, dir, n_files, basis_degree, boundary_knots, internal_knots, time, Y, burnin_prop, X, cov_adj);
  SHIM_END
}
SEXP _BayesFMMM_FLLik(SEXP dir, SEXP n_files, SEXP basis_degree, SEXP boundary_knots, SEXP internal_knots, SEXP time, SEXP Y,
                      SEXP X, SEXP cov_adj) {
  SHIM_BEGIN
  Inputs in;
  const Ragged y = flatten(Y), t = flatten(time);
  bfmmm_post_args a;
  set_post(a, y, t, dir, n_files, basis_degree, boundary_knots, internal_knots, X, cov_adj, in);
  bfmmm_result* r = NULL;
  if (bfmmm_FLLik(&a, &r)) lib_error();
  const double* d; int64_t cnt; const int64_t* dims; int nd;
  bfmmm_result_get(r, "value", &d, &cnt, &dims, &nd);
  SEXP out = PROTECT(Rf_allocVector(REALSXP, cnt));
  memcpy(REAL(out), d, sizeof(double) * (size_t)cnt);
  bfmmm_result_free(r);
  UNPROTECT(1);
  return out;
  SHIM_END
}

// ---- CPO, multivariate criteria, credible intervals (RcppExports.cpp:17, :62, :78, :145, :159, :253-316) ---------------
static SEXP value_vector(bfmmm_result* r) {
  SHIM_BEGIN
  Inputs in;
  const double* d; int64_t cnt; const int64_t* dims; int nd;
  bfmmm_result_get(r, "value", &d, &cnt, &dims, &nd);
  SEXP out = PROTECT(Rf_allocVector(REALSXP, cnt));
  memcpy(REAL(out), d, sizeof(double) * (size_t)cnt);
  bfmmm_result_free(r);
  UNPROTECT(1);
  return out;
  SHIM_END
}

SEXP _BayesFMMM_ConditionalPredictiveOrdinates(SEXP dir, SEXP n_files, SEXP basis_degree, SEXP boundary_knots, SEXP internal_knots,
                                               SEXP time, SEXP Y, SEXP burnin_prop, SEXP X, SEXP cov_adj, SEXP log_CPO) {
  SHIM_BEGIN
  Inputs in;
  const Ragged y = flatten(Y), t = flatten(time);
  bfmmm_post_args a;
  set_post(a, y, t, dir, n_files, basis_degree, boundary_knots, internal_knots, X, cov_adj, in);
  a.burnin_prop = Rf_asReal(burnin_prop);
  bfmmm_result* r = NULL;
  if (bfmmm_ConditionalPredictiveOrdinates(&a, Rf_asLogical(log_CPO) ? 1 : 0, &r)) lib_error();
  return value_vector(r);
  SHIM_END
}

static void set_post_mv(bfmmm_post_args& a, SEXP dir, SEXP n_files, SEXP Y, SEXP X, SEXP cov_adj, Inputs& in) {
  bfmmm_post_defaults(&a);
  a.dir = CHAR(STRING_ELT(dir, 0)); a.n_files = Rf_asInteger(n_files);
  if (!Rf_isMatrix(Y)) throw std::runtime_error("'Y' must be a matrix");
  a.y = in.num(Y); a.n_funct = Rf_nrows(Y); a.P = Rf_ncols(Y);
  if (X != R_NilValue) { a.X = in.num(X); a.D = Rf_ncols(X); }
  a.cov_adj = Rf_asLogical(cov_adj) ? 1 : 0;
}

SEXP _BayesFMMM_MVLLik(SEXP dir, SEXP n_files, SEXP Y, SEXP X, SEXP cov_adj) {
  SHIM_BEGIN
  Inputs in;
  bfmmm_post_args a;
  set_post_mv(a, dir, n_files, Y, X, cov_adj, in);
  bfmmm_result* r = NULL;
  if (bfmmm_MVLLik(&a, &r)) lib_error();
  return value_vector(r);
  SHIM_END
}

static SEXP mv_scalar(int (*fn)(const bfmmm_post_args*, double*), SEXP dir, SEXP n_files, SEXP Y, SEXP burnin_prop, SEXP X, SEXP cov_adj) {
  SHIM_BEGIN
  Inputs in;
  bfmmm_post_args a;
  set_post_mv(a, dir, n_files, Y, X, cov_adj, in);
  a.burnin_prop = Rf_asReal(burnin_prop);
  double v = 0.0;
  if (fn(&a, &v)) lib_error();
  return Rf_ScalarReal(v);
  SHIM_END
}
SEXP _BayesFMMM_MVDIC(SEXP dir, SEXP n_files, SEXP Y, SEXP burnin_prop, SEXP X, SEXP cov_adj) { return mv_scalar(bfmmm_MVDIC, dir, n_files, Y, burnin_prop, X, cov_adj); }
SEXP _BayesFMMM_MVAIC(SEXP dir, SEXP n_files, SEXP Y, SEXP burnin_prop, SEXP X, SEXP cov_adj) { return mv_scalar(bfmmm_MVAIC, dir, n_files, Y, burnin_prop, X, cov_adj); }
SEXP _BayesFMMM_MVBIC(SEXP dir, SEXP n_files, SEXP Y, SEXP burnin_prop, SEXP X, SEXP cov_adj) { return mv_scalar(bfmmm_MVBIC, dir, n_files, Y, burnin_prop, X, cov_adj); }

static void set_ci(bfmmm_ci_args& a, SEXP dir, SEXP n_files, SEXP alpha, SEXP burnin_prop) {
  bfmmm_ci_defaults(&a);
  a.dir = CHAR(STRING_ELT(dir, 0)); a.n_files = Rf_asInteger(n_files); a.alpha = Rf_asReal(alpha); a.burnin_prop = Rf_asReal(burnin_prop);
}
static void set_ci_x(bfmmm_ci_args& a, SEXP X, Inputs& in) {
  if (X != R_NilValue) { a.X = in.num(X); a.n_x = Rf_nrows(X); a.D = Rf_ncols(X); }
}
static void set_ci_basis(bfmmm_ci_args& a, SEXP basis_degree, SEXP boundary_knots, SEXP internal_knots, SEXP rescale, SEXP simultaneous,
                         SEXP X, SEXP trans_mats, Inputs& in) {
  if (Rf_length(boundary_knots) != 2) throw std::runtime_error("'boundary_knots' must have two elements");
  a.basis_degree = Rf_asInteger(basis_degree); a.boundary_knots = in.num(boundary_knots);
  a.internal_knots = in.num(internal_knots); a.n_internal_knots = Rf_length(internal_knots);
  a.rescale = Rf_asLogical(rescale) ? 1 : 0; a.simultaneous = Rf_asLogical(simultaneous) ? 1 : 0;
  set_ci_x(a, X, in);
  if (trans_mats != R_NilValue) a.trans_mats = in.num(trans_mats);
}
// tensor-product basis arguments of the HD functions: basis_degree a vector, boundary_knots dim x 2, internal_knots a list
struct HDCi { std::vector<int32_t> deg, nint; std::vector<double> bk; Ragged ik; };
static void set_ci_hd(bfmmm_ci_args& a, HDCi& h, SEXP basis_degree, SEXP boundary_knots, SEXP internal_knots) {
  const int dim = Rf_length(basis_degree);
  h.ik = flatten(internal_knots);
  const Num deg(basis_degree), bk(boundary_knots);
  if ((int)h.ik.off.size() - 1 != dim || bk.len != 2 * (R_xlen_t)dim) throw std::runtime_error("'basis_degree', 'boundary_knots' (dim x 2) and 'internal_knots' disagree on the dimension");
  for (int j = 0; j < dim; ++j) {
    h.deg.push_back((int32_t)deg.p[j]);
    h.nint.push_back((int32_t)(h.ik.off[j + 1] - h.ik.off[j]));
    h.bk.push_back(bk.p[j]); h.bk.push_back(bk.p[j + dim]);
  }
  a.dim = dim; a.basis_degree_hd = h.deg.data(); a.n_internal_hd = h.nint.data();
  a.boundary_knots = h.bk.data(); a.internal_knots = h.ik.v.data();
}

SEXP _BayesFMMM_SigmaCI(SEXP dir, SEXP n_files, SEXP alpha, SEXP burnin_prop) {
  SHIM_BEGIN
  Inputs in;
  bfmmm_ci_args a;
  set_ci(a, dir, n_files, alpha, burnin_prop);
  bfmmm_result* r = NULL;
  return finish(bfmmm_SigmaCI(&a, &r), r, NULL);
  SHIM_END
}
SEXP _BayesFMMM_ZCI(SEXP dir, SEXP n_files, SEXP alpha, SEXP rescale, SEXP burnin_prop) {
  SHIM_BEGIN
  Inputs in;
  bfmmm_ci_args a;
  set_ci(a, dir, n_files, alpha, burnin_prop);
  a.rescale = Rf_asLogical(rescale) ? 1 : 0;
  bfmmm_result* r = NULL;
  return finish(bfmmm_ZCI(&a, &r), r, NULL);
  SHIM_END
}
SEXP _BayesFMMM_FMeanCI(SEXP dir, SEXP n_files, SEXP time, SEXP basis_degree, SEXP boundary_knots, SEXP internal_knots, SEXP k, SEXP alpha,
                        SEXP rescale, SEXP simultaneous, SEXP burnin_prop, SEXP X, SEXP trans_mats) {
  SHIM_BEGIN
  Inputs in;
  bfmmm_ci_args a;
  set_ci(a, dir, n_files, alpha, burnin_prop);
  set_ci_basis(a, basis_degree, boundary_knots, internal_knots, rescale, simultaneous, X, trans_mats, in);
  a.time = in.num(time); a.n_time = Rf_length(time); a.k = Rf_asInteger(k);
  bfmmm_result* r = NULL;
  return finish(bfmmm_FMeanCI(&a, &r), r, NULL);
  SHIM_END
}
SEXP _BayesFMMM_FCovCI(SEXP dir, SEXP n_files, SEXP time1, SEXP time2, SEXP basis_degree, SEXP boundary_knots, SEXP internal_knots, SEXP l,
                       SEXP m, SEXP alpha, SEXP rescale, SEXP simultaneous, SEXP burnin_prop, SEXP X, SEXP trans_mats) {
  SHIM_BEGIN
  Inputs in;
  bfmmm_ci_args a;
  set_ci(a, dir, n_files, alpha, burnin_prop);
  set_ci_basis(a, basis_degree, boundary_knots, internal_knots, rescale, simultaneous, X, trans_mats, in);
  a.time = in.num(time1); a.n_time = Rf_length(time1); a.time2 = in.num(time2); a.n_time2 = Rf_length(time2);
  a.l = Rf_asInteger(l); a.m = Rf_asInteger(m);
  bfmmm_result* r = NULL;
  return finish(bfmmm_FCovCI(&a, &r), r, NULL);      // (with X: cov_trace is one n1 x n2 x kept x n_x array, see bfmmm_post.h)
  SHIM_END
}
SEXP _BayesFMMM_MVMeanCI(SEXP dir, SEXP n_files, SEXP alpha, SEXP rescale, SEXP burnin_prop, SEXP X) {
  SHIM_BEGIN
  Inputs in;
  bfmmm_ci_args a;
  set_ci(a, dir, n_files, alpha, burnin_prop);
  a.rescale = Rf_asLogical(rescale) ? 1 : 0;
  set_ci_x(a, X, in);
  bfmmm_result* r = NULL;
  SEXP out = PROTECT(finish(bfmmm_MVMeanCI(&a, &r), r, NULL));
  if (X != R_NilValue) {                                 // mean_trace: K x P x (kept n_x) -> the reference's list of n_x cubes K x P x kept
    SEXP nm = Rf_getAttrib(out, R_NamesSymbol);
    for (R_xlen_t e = 0; e < Rf_xlength(out); ++e)
      if (std::strcmp(CHAR(STRING_ELT(nm, e)), "mean_trace") == 0) {
        SEXP mt = VECTOR_ELT(out, e);
        const int* d = INTEGER(Rf_getAttrib(mt, R_DimSymbol));
        const int64_t nx = a.n_x, kept = d[2] / nx, len = (int64_t)d[0] * d[1] * kept;
        const int64_t d3[3] = {d[0], d[1], kept};
        SEXP lst = PROTECT(Rf_allocVector(VECSXP, nx));
        for (int64_t w = 0; w < nx; ++w) SET_VECTOR_ELT(lst, w, array_of(REAL(mt) + w * len, d3, 3));
        SET_VECTOR_ELT(out, e, lst);
        UNPROTECT(1);
      }
  }
  UNPROTECT(1);
  return out;
  SHIM_END
}

// HDFMeanCI (RcppExports.cpp:40, 12 arguments -- no trans_mats): time n_time x dim matrix, basis_degree a vector, boundary_knots dim x 2, internal_knots a list
SEXP _BayesFMMM_HDFMeanCI(SEXP dir, SEXP n_files, SEXP time, SEXP basis_degree, SEXP boundary_knots, SEXP internal_knots, SEXP k, SEXP alpha,
                          SEXP rescale, SEXP simultaneous, SEXP burnin_prop, SEXP X) {
  SHIM_BEGIN
  Inputs in;
  bfmmm_ci_args a;
  set_ci(a, dir, n_files, alpha, burnin_prop);
  HDCi h;
  set_ci_hd(a, h, basis_degree, boundary_knots, internal_knots);
  if (!Rf_isMatrix(time) || Rf_ncols(time) != a.dim) throw std::runtime_error("'time' must be a matrix with one column per dimension");
  a.time = in.num(time); a.n_time = Rf_nrows(time); a.k = Rf_asInteger(k);
  a.rescale = Rf_asLogical(rescale) ? 1 : 0; a.simultaneous = Rf_asLogical(simultaneous) ? 1 : 0;
  set_ci_x(a, X, in);
  bfmmm_result* r = NULL;
  return finish(bfmmm_HDFMeanCI(&a, &r), r, NULL);
  SHIM_END
}

// HDFCovCI (RcppExports.cpp:103, 14 arguments), MVCovCI (:127, 8), FSamplePaths (:337, 11)
SEXP _BayesFMMM_HDFCovCI(SEXP dir, SEXP n_files, SEXP time1, SEXP time2, SEXP basis_degree, SEXP boundary_knots, SEXP internal_knots, SEXP l,
                         SEXP m, SEXP alpha, SEXP rescale, SEXP simultaneous, SEXP burnin_prop, SEXP X) {
  SHIM_BEGIN
  Inputs in;
  bfmmm_ci_args a;
  set_ci(a, dir, n_files, alpha, burnin_prop);
  HDCi h;
  set_ci_hd(a, h, basis_degree, boundary_knots, internal_knots);
  if (!Rf_isMatrix(time1) || !Rf_isMatrix(time2) || Rf_ncols(time1) != a.dim || Rf_ncols(time2) != a.dim)
    throw std::runtime_error("'time1' and 'time2' must be matrices with one column per dimension");
  a.time = in.num(time1); a.n_time = Rf_nrows(time1); a.time2 = in.num(time2); a.n_time2 = Rf_nrows(time2);
  a.l = Rf_asInteger(l); a.m = Rf_asInteger(m);
  a.rescale = Rf_asLogical(rescale) ? 1 : 0; a.simultaneous = Rf_asLogical(simultaneous) ? 1 : 0;
  set_ci_x(a, X, in);
  bfmmm_result* r = NULL;
  return finish(bfmmm_HDFCovCI(&a, &r), r, NULL);
  SHIM_END
}
SEXP _BayesFMMM_MVCovCI(SEXP dir, SEXP n_files, SEXP l, SEXP m, SEXP alpha, SEXP rescale, SEXP burnin_prop, SEXP X) {
  SHIM_BEGIN
  Inputs in;
  bfmmm_ci_args a;
  set_ci(a, dir, n_files, alpha, burnin_prop);
  a.l = Rf_asInteger(l); a.m = Rf_asInteger(m); a.rescale = Rf_asLogical(rescale) ? 1 : 0;
  set_ci_x(a, X, in);
  bfmmm_result* r = NULL;
  return finish(bfmmm_MVCovCI(&a, &r), r, NULL);
  SHIM_END
}
// the reference returns lists over the curves (arma::field): CI_* vectors of n_i, the traces kept x n_i matrices
SEXP _BayesFMMM_FSamplePaths(SEXP dir, SEXP n_files, SEXP basis_degree, SEXP boundary_knots, SEXP internal_knots, SEXP time, SEXP alpha,
                             SEXP burnin_prop, SEXP simultaneous, SEXP X, SEXP cov_adj) {
  SHIM_BEGIN
  Inputs in;
  const Ragged t = flatten(time);
  Ragged y = t;                                          // FSamplePaths has no Y: only the shape is needed
  bfmmm_post_args a;
  set_post(a, y, t, dir, n_files, basis_degree, boundary_knots, internal_knots, X, cov_adj, in);
  a.y = NULL;
  a.burnin_prop = Rf_asReal(burnin_prop);
  Owned o;
  if (bfmmm_FSamplePaths(&a, Rf_asReal(alpha), Rf_asLogical(simultaneous) ? 1 : 0, seed_from_R(), &o.r)) lib_error();
  const char* names[5] = {"CI_Upper", "CI_50", "CI_Lower", "Path_trace", "Mean_only_Path_trace"};
  const int64_t nf = (int64_t)t.off.size() - 1;
  SEXP out = PROTECT(Rf_allocVector(VECSXP, 5));
  SEXP nms = PROTECT(Rf_allocVector(STRSXP, 5));
  for (int e = 0; e < 5; ++e) {
    const double* data; int64_t cnt; const int64_t* dims; int nd;
    bfmmm_result_get(o.r, names[e], &data, &cnt, &dims, &nd);
    const int64_t kept = (e < 3) ? 1 : dims[0];
    SEXP lst = PROTECT(Rf_allocVector(VECSXP, nf));
    for (int64_t c = 0; c < nf; ++c) {
      const int64_t ni = t.off[c + 1] - t.off[c];
      const int64_t d2[2] = {kept, ni};
      SET_VECTOR_ELT(lst, c, (e < 3) ? array_of(data + t.off[c], &ni, 1) : array_of(data + kept * t.off[c], d2, 2));
    }
    SET_VECTOR_ELT(out, e, lst);
    SET_STRING_ELT(nms, e, Rf_mkChar(names[e]));
    UNPROTECT(1);
  }
  Rf_setAttrib(out, R_NamesSymbol, nms);
  UNPROTECT(2);
  return out;
  SHIM_END
}

// ---- readers of the on-disk batches (UserFunctions.cpp:2158-2399) -------------------------------------------------
SEXP _BayesFMMM_ReadVec(SEXP file) { return read_plain(file, true); }
SEXP _BayesFMMM_ReadMat(SEXP file) { return read_plain(file, false); }
SEXP _BayesFMMM_ReadCube(SEXP file) { return read_plain(file, false); }
SEXP _BayesFMMM_ReadFieldCube(SEXP file) { return read_field(file, false); }
SEXP _BayesFMMM_ReadFieldMat(SEXP file) { return read_field(file, false); }
SEXP _BayesFMMM_ReadFieldVec(SEXP file) { return read_field(file, true); }

static const R_CallMethodDef CallEntries[] = {            // as src/RcppExports.cpp:794-835 (the entries this shim replaces)
    {"_BayesFMMM_BFMMM_Nu_Z_multiple_try", (DL_FUNC)&_BayesFMMM_BFMMM_Nu_Z_multiple_try, 28},
    {"_BayesFMMM_BFMMM_Theta_est", (DL_FUNC)&_BayesFMMM_BFMMM_Theta_est, 32},
    {"_BayesFMMM_BFMMM_warm_start", (DL_FUNC)&_BayesFMMM_BFMMM_warm_start, 38},
    {"_BayesFMMM_BMVMMM_Nu_Z_multiple_try", (DL_FUNC)&_BayesFMMM_BMVMMM_Nu_Z_multiple_try, 23},
    {"_BayesFMMM_BMVMMM_Theta_est", (DL_FUNC)&_BayesFMMM_BMVMMM_Theta_est, 27},
    {"_BayesFMMM_BMVMMM_warm_start", (DL_FUNC)&_BayesFMMM_BMVMMM_warm_start, 33},
    {"_BayesFMMM_BHDFMMM_Nu_Z_multiple_try", (DL_FUNC)&_BayesFMMM_BHDFMMM_Nu_Z_multiple_try, 28},
    {"_BayesFMMM_BHDFMMM_Theta_est", (DL_FUNC)&_BayesFMMM_BHDFMMM_Theta_est, 32},
    {"_BayesFMMM_BHDFMMM_warm_start", (DL_FUNC)&_BayesFMMM_BHDFMMM_warm_start, 38},
    {"_BayesFMMM_FDIC", (DL_FUNC)&_BayesFMMM_FDIC, 10},
    {"_BayesFMMM_FAIC", (DL_FUNC)&_BayesFMMM_FAIC, 10},
    {"_BayesFMMM_FBIC", (DL_FUNC)&_BayesFMMM_FBIC, 10},
    {"_BayesFMMM_FLLik", (DL_FUNC)&_BayesFMMM_FLLik, 9},
    {"_BayesFMMM_ConditionalPredictiveOrdinates", (DL_FUNC)&_BayesFMMM_ConditionalPredictiveOrdinates, 11},
    {"_BayesFMMM_MVLLik", (DL_FUNC)&_BayesFMMM_MVLLik, 5},
    {"_BayesFMMM_MVDIC", (DL_FUNC)&_BayesFMMM_MVDIC, 6},
    {"_BayesFMMM_MVAIC", (DL_FUNC)&_BayesFMMM_MVAIC, 6},
    {"_BayesFMMM_MVBIC", (DL_FUNC)&_BayesFMMM_MVBIC, 6},
    {"_BayesFMMM_SigmaCI", (DL_FUNC)&_BayesFMMM_SigmaCI, 4},
    {"_BayesFMMM_ZCI", (DL_FUNC)&_BayesFMMM_ZCI, 5},
    {"_BayesFMMM_FMeanCI", (DL_FUNC)&_BayesFMMM_FMeanCI, 13},
    {"_BayesFMMM_FCovCI", (DL_FUNC)&_BayesFMMM_FCovCI, 15},
    {"_BayesFMMM_MVMeanCI", (DL_FUNC)&_BayesFMMM_MVMeanCI, 6},
    {"_BayesFMMM_HDFMeanCI", (DL_FUNC)&_BayesFMMM_HDFMeanCI, 12},
    {"_BayesFMMM_HDFCovCI", (DL_FUNC)&_BayesFMMM_HDFCovCI, 14},
    {"_BayesFMMM_MVCovCI", (DL_FUNC)&_BayesFMMM_MVCovCI, 8},
    {"_BayesFMMM_FSamplePaths", (DL_FUNC)&_BayesFMMM_FSamplePaths, 11},
    {"_BayesFMMM_ReadVec", (DL_FUNC)&_BayesFMMM_ReadVec, 1},
    {"_BayesFMMM_ReadMat", (DL_FUNC)&_BayesFMMM_ReadMat, 1},
    {"_BayesFMMM_ReadCube", (DL_FUNC)&_BayesFMMM_ReadCube, 1},
    {"_BayesFMMM_ReadFieldCube", (DL_FUNC)&_BayesFMMM_ReadFieldCube, 1},
    {"_BayesFMMM_ReadFieldMat", (DL_FUNC)&_BayesFMMM_ReadFieldMat, 1},
    {"_BayesFMMM_ReadFieldVec", (DL_FUNC)&_BayesFMMM_ReadFieldVec, 1},
    {NULL, NULL, 0}};

void R_init_BayesFMMM(DllInfo* dll) {
  R_registerRoutines(dll, NULL, CallEntries, NULL, NULL);
  R_useDynamicSymbols(dll, FALSE);
}

}  // extern "C"
