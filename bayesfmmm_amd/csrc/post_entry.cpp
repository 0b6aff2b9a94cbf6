// FLLik / FDIC / FAIC / FBIC (include/bfmmm_post.h): read the on-disk batches as the reference does, evaluate every
// observation under every draw on the device (bfmmm_post_pointwise, kernels_post.hip) and apply the reference's formulas.
#include "../../include/bfmmm_post.h"

#include <cmath>
#include <string>
#include <vector>

int bfmmm_io_fail(const std::string& m);
int arma_load_obj(const std::string& path, std::vector<double>& v, int64_t dims[3]);
int arma_load_field(const std::string& path, std::vector<std::vector<double>>& objs, int64_t* n_rows, int64_t* n_cols, int64_t dims[3]);

namespace {

struct Draws {
  int K = 0, P = 0, M = 0, D = 0, n = 0, T = 0;
  std::vector<double> nu, Phi, Z, chi, sigma, eta, xi, B;
};

// concatenation of `<dir><name><q>.txt`, q < n_files, along the last (draw) dimension of a cube / vector
int load_cat(const std::string& dir, const char* name, int n_files, std::vector<double>& out, int64_t dims[3]) {
  out.clear();
  for (int q = 0; q < n_files; ++q) {
    std::vector<double> v;
    int64_t d[3];
    if (arma_load_obj(dir + name + std::to_string(q) + ".txt", v, d)) return 1;
    if (q == 0) { dims[0] = d[0]; dims[1] = d[1]; dims[2] = d[2]; }
    else if (d[0] != dims[0] || d[1] != dims[1] || d[2] != dims[2]) return bfmmm_io_fail(std::string("files '") + name + "<q>.txt' differ in size");
    out.insert(out.end(), v.begin(), v.end());
  }
  return 0;
}

// argument checks in the reference's order and wording (PostProcessing.cpp:3673-3700, :4902-4915)
int check(const bfmmm_post_args* a, bool with_burnin, bool mv = false) {
  if (mv) {
    if (!a || !a->dir || !a->y || a->P < 1 || a->n_funct < 1) return bfmmm_io_fail("null argument");
    if (with_burnin && (a->burnin_prop < 0 || a->burnin_prop >= 1)) return bfmmm_io_fail("'burnin_prop' must be between 0 and 1");
    if (a->n_files <= 0) return bfmmm_io_fail("'n_files' must be greater than 0");
    return 0;
  }
  if (!a || !a->dir || !a->t || !a->y || !a->offsets || !a->boundary_knots || (a->n_internal_knots > 0 && !a->internal_knots))
    return bfmmm_io_fail("null argument");
  if (a->basis_degree < 1) return bfmmm_io_fail("'basis_degree' must be an integer greater than or equal to 1");
  for (int i = 0; i < a->n_internal_knots; ++i) {
    if (a->boundary_knots[0] >= a->internal_knots[i])
      return bfmmm_io_fail("at least one element in 'internal_knots' is less than or equal to first boundary knot");
    if (a->boundary_knots[1] <= a->internal_knots[i])
      return bfmmm_io_fail("at least one element in 'internal_knots' is more than or equal to second boundary knot");
  }
  if (with_burnin && (a->burnin_prop < 0 || a->burnin_prop >= 1)) return bfmmm_io_fail("'burnin_prop' must be between 0 and 1");
  if (a->n_files <= 0) return bfmmm_io_fail("'n_files' must be greater than 0");
  return 0;
}

int load_draws(const bfmmm_post_args* a, Draws& dr, bool mv = false) {
  const std::string dir = a->dir;
  int64_t d[3], nr, nc;
  if (load_cat(dir, "Nu", a->n_files, dr.nu, d)) return 1;
  dr.K = (int)d[0]; dr.P = (int)d[1];
  const int per_file = (int)d[2];
  dr.T = per_file * a->n_files;
  if (load_cat(dir, "Z", a->n_files, dr.Z, d)) return 1;
  dr.n = (int)d[0];
  if (load_cat(dir, "Chi", a->n_files, dr.chi, d)) return 1;
  dr.M = (int)d[1];
  if (load_cat(dir, "Sigma", a->n_files, dr.sigma, d)) return 1;
  for (int q = 0; q < a->n_files; ++q) {
    std::vector<std::vector<double>> objs;
    if (arma_load_field(dir + "Phi" + std::to_string(q) + ".txt", objs, &nr, &nc, d)) return 1;
    if ((int)nr < per_file) return bfmmm_io_fail("'Phi<q>.txt' holds fewer draws than 'Nu<q>.txt'");
    for (int l = 0; l < per_file; ++l) dr.Phi.insert(dr.Phi.end(), objs[l].begin(), objs[l].end());
  }
  if (dr.n != a->n_funct) return bfmmm_io_fail("The number of functions in 'Y' must be equal to the number of rows of the saved 'Z' draws");
  if (a->X) {
    for (int q = 0; q < a->n_files; ++q) {
      std::vector<std::vector<double>> objs;
      if (arma_load_field(dir + "Eta" + std::to_string(q) + ".txt", objs, &nr, &nc, d)) return 1;
      if (q == 0) {
        dr.D = (int)d[1];
        if (a->D != dr.D) return bfmmm_io_fail("The number of columns in 'X' must be equal to the number of covariates in the model");
      }
      for (int l = 0; l < per_file; ++l) dr.eta.insert(dr.eta.end(), objs[l].begin(), objs[l].end());
    }
    if (a->cov_adj)
      for (int q = 0; q < a->n_files; ++q) {
        std::vector<std::vector<double>> objs;
        if (arma_load_field(dir + "Xi" + std::to_string(q) + ".txt", objs, &nr, &nc, d)) return 1;
        // field (draw l, cluster k) at l + nr * k  ->  draw-major
        for (int l = 0; l < per_file; ++l)
          for (int k = 0; k < dr.K; ++k) { const auto& c = objs[(size_t)l + (size_t)nr * k]; dr.xi.insert(dr.xi.end(), c.begin(), c.end()); }
      }
  }
  if (mv) {
    if (dr.P != a->P) return bfmmm_io_fail("the saved draws do not match the number of columns of 'Y'");
    return 0;
  }
  // basis rows at the observed time points (splines2::BSpline(time, internal_knots, degree, boundary_knots).basis(true))
  const int64_t n_obs = a->offsets[a->n_funct];
  if (dr.P != a->n_internal_knots + a->basis_degree + 1) return bfmmm_io_fail("the saved draws do not match the basis ('basis_degree', 'internal_knots')");
  std::vector<double> cm((size_t)n_obs * dr.P);
  const int32_t deg = a->basis_degree, nint = a->n_internal_knots;
  if (bfmmm_tensor_bspline((int)n_obs, 1, a->t, &deg, a->boundary_knots, &nint, a->internal_knots, cm.data())) return 1;
  dr.B.resize(cm.size());
  for (int64_t l = 0; l < n_obs; ++l)
    for (int p = 0; p < dr.P; ++p) dr.B[(size_t)l * dr.P + p] = cm[(size_t)l + (size_t)n_obs * p];
  return 0;
}

int run(const bfmmm_post_args* a, const Draws& dr, int first_kept, std::vector<double>* ll, std::vector<double>* pdf, std::vector<double>* fit) {
  bfmmm_post_input in{};
  in.n = dr.n; in.K = dr.K; in.P = dr.P; in.M = dr.M; in.D = dr.D;
  in.offsets = a->offsets; in.y = a->y; in.B = dr.B.data(); in.X = a->X;
  in.T = dr.T; in.nu = dr.nu.data(); in.Phi = dr.Phi.data(); in.Z = dr.Z.data(); in.chi = dr.chi.data(); in.sigma = dr.sigma.data();
  in.eta = dr.eta.empty() ? nullptr : dr.eta.data();
  in.xi = dr.xi.empty() ? nullptr : dr.xi.data();
  in.device = a->device;
  const int64_t n_obs = a->offsets[a->n_funct];
  if (ll) ll->assign((size_t)dr.T, 0.0);
  if (pdf) pdf->assign((size_t)n_obs, 0.0);
  if (fit) fit->assign((size_t)n_obs, 0.0);
  return bfmmm_post_pointwise(&in, first_kept, ll ? ll->data() : nullptr, pdf ? pdf->data() : nullptr, fit ? fit->data() : nullptr);
}

// number of parameters the reference charges (PostProcessing.cpp:4176-4178, :4356-4364; the same in FBIC)
double n_params(const Draws& dr, bool has_x, bool cov_adj) {
  const double n = dr.n, K = dr.K, P = dr.P, M = dr.M, D = dr.D;
  double v = (n + P) * K + 2 * P * M * K + 2 + 4 * K + n * M + M * K;
  if (has_x) {
    v += P * D * K + D * K;
    if (cov_adj) v += 2 * P * D * K * M + D * K * M + 2 * D * K;
  }
  return v;
}

// log-likelihood at the posterior-mean curve fits and the mean of ALL saved sigma^2 (PostProcessing.cpp:4158-4173)
int loglik_at_means(const bfmmm_post_args* a, const Draws& dr, double* out) {
  const int kept = (int)std::round((1 - a->burnin_prop) * dr.T);
  if (kept < 1) return bfmmm_io_fail("'burnin_prop' leaves no draws");
  std::vector<double> fit;
  if (run(a, dr, dr.T - kept, nullptr, nullptr, &fit)) return 1;
  double ms = 0.0;
  for (double s : dr.sigma) ms += s;
  ms /= (double)dr.sigma.size();
  const double sd = std::sqrt(ms);
  double ll = 0.0;
  for (size_t e = 0; e < fit.size(); ++e) {
    const double z = (a->y[e] - fit[e]) / sd;
    ll += -(0.91893853320467274178 + 0.5 * z * z + std::log(sd));
  }
  *out = ll;
  return 0;
}

}  // namespace

extern "C" void bfmmm_post_defaults(bfmmm_post_args* a) {
  *a = bfmmm_post_args{};
  a->burnin_prop = 0.1;
}

extern "C" int bfmmm_FLLik(const bfmmm_post_args* a, bfmmm_result** out) {
  if (!out) return bfmmm_io_fail("null argument");
  if (check(a, false)) return 1;
  Draws dr;
  if (load_draws(a, dr)) return 1;
  std::vector<double> ll;
  if (run(a, dr, 0, &ll, nullptr, nullptr)) return 1;
  bfmmm_result* r = bfmmm_result_create();
  const int64_t T = dr.T;
  bfmmm_result_set(r, "value", ll.data(), T, &T, 1);
  *out = r;
  return 0;
}

extern "C" int bfmmm_FDIC(const bfmmm_post_args* a, double* out) {
  if (!out) return bfmmm_io_fail("null argument");
  if (check(a, true)) return 1;
  Draws dr;
  if (load_draws(a, dr)) return 1;
  const int kept = (int)std::round((1 - a->burnin_prop) * dr.T);
  if (kept < 1) return bfmmm_io_fail("'burnin_prop' leaves no draws");
  std::vector<double> ll, pdf;
  if (run(a, dr, dr.T - kept, &ll, &pdf, nullptr)) return 1;
  double expected_log_f = 0.0;                               // PostProcessing.cpp:3835-3839 / :3922-3928
  for (int t = dr.T - kept; t < dr.T; ++t) expected_log_f += ll[t];
  expected_log_f /= kept;
  double f_hat = 0.0;                                        // :3841-3851 / :3930-3942
  for (double v : pdf) f_hat += std::log(v);
  *out = 2 * f_hat - 4 * expected_log_f;
  return 0;
}

extern "C" int bfmmm_FAIC(const bfmmm_post_args* a, double* out) {
  if (!out) return bfmmm_io_fail("null argument");
  if (check(a, true)) return 1;
  Draws dr;
  if (load_draws(a, dr)) return 1;
  double ll;
  if (loglik_at_means(a, dr, &ll)) return 1;
  *out = 2 * n_params(dr, a->X != nullptr, a->cov_adj != 0) - 2 * ll;
  return 0;
}

extern "C" int bfmmm_FBIC(const bfmmm_post_args* a, double* out) {
  if (!out) return bfmmm_io_fail("null argument");
  if (check(a, true)) return 1;
  Draws dr;
  if (load_draws(a, dr)) return 1;
  double ll;
  if (loglik_at_means(a, dr, &ll)) return 1;
  const double tilde_N = (double)a->offsets[a->n_funct];      // PostProcessing.cpp:4591-4594
  *out = 2 * ll - std::log(tilde_N) * n_params(dr, a->X != nullptr, a->cov_adj != 0);
  return 0;
}

extern "C" int bfmmm_ConditionalPredictiveOrdinates(const bfmmm_post_args* a, int32_t log_CPO, bfmmm_result** out) {
  if (!out) return bfmmm_io_fail("null argument");
  // argument checks in the reference's order (PostProcessing.cpp:6352-6375)
  if (a && a->n_files <= 0) return bfmmm_io_fail("'n_files' must be greater than 0");
  if (a && (a->burnin_prop < 0 || a->burnin_prop >= 1)) return bfmmm_io_fail("'burnin_prop' must be between 0 and 1");
  if (check(a, true)) return 1;
  Draws dr;
  if (load_draws(a, dr)) return 1;
  bfmmm_post_input in{};
  in.n = dr.n; in.K = dr.K; in.P = dr.P; in.M = dr.M; in.D = dr.D;
  in.offsets = a->offsets; in.y = a->y; in.B = dr.B.data(); in.X = a->X;
  in.T = dr.T; in.nu = dr.nu.data(); in.Phi = dr.Phi.data(); in.Z = dr.Z.data(); in.chi = dr.chi.data(); in.sigma = dr.sigma.data();
  in.eta = dr.eta.empty() ? nullptr : dr.eta.data();
  in.xi = dr.xi.empty() ? nullptr : dr.xi.data();
  in.device = a->device;
  std::vector<double> v((size_t)dr.n);
  const int first = (int)std::floor(a->burnin_prop * dr.T);      // CalculateLikelihood.h:361
  if (bfmmm_post_cpo(&in, first, v.data())) return 1;
  if (!log_CPO) for (double& x : v) x = std::exp(x);
  bfmmm_result* r = bfmmm_result_create();
  const int64_t nn = dr.n;
  bfmmm_result_set(r, "value", v.data(), nn, &nn, 1);
  *out = r;
  return 0;
}

// ---- multivariate model: the same pass with the identity basis (a row's P coordinates are its observations) ----------
namespace {

struct MVRun {
  std::vector<int64_t> off;
  std::vector<double> y;         // row-major: the observations of row i are contiguous
  std::vector<double> ll, joint, fit;
};

int run_mv(const bfmmm_post_args* a, const Draws& dr, int first_kept, bool want_joint, bool want_fit, MVRun& m) {
  const int n = a->n_funct, P = a->P;
  m.off.resize((size_t)n + 1);
  for (int i = 0; i <= n; ++i) m.off[(size_t)i] = (int64_t)i * P;
  m.y.resize((size_t)n * P);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < P; ++j) m.y[(size_t)i * P + j] = a->y[(size_t)i + (size_t)n * j];
  bfmmm_post_input in{};
  in.n = n; in.K = dr.K; in.P = P; in.M = dr.M; in.D = dr.D;
  in.offsets = m.off.data(); in.y = m.y.data(); in.B = nullptr; in.identity_basis = 1; in.X = a->X;
  in.T = dr.T; in.nu = dr.nu.data(); in.Phi = dr.Phi.data(); in.Z = dr.Z.data(); in.chi = dr.chi.data(); in.sigma = dr.sigma.data();
  in.eta = dr.eta.empty() ? nullptr : dr.eta.data();
  in.xi = dr.xi.empty() ? nullptr : dr.xi.data();
  in.device = a->device;
  m.ll.assign((size_t)dr.T, 0.0);
  if (want_joint) m.joint.assign((size_t)n, 0.0);
  if (want_fit) m.fit.assign((size_t)n * P, 0.0);
  if (bfmmm_post_pointwise_joint(&in, first_kept, m.ll.data(), want_joint ? m.joint.data() : nullptr, want_fit ? m.fit.data() : nullptr)) return 1;
  // calcLikelihoodMV charges (P / 2) log(2 pi sigma) per row with INTEGER division (CalculateLikelihood.h:155): for odd P
  // the reference's value lacks half a log(2 pi sigma) per row
  if (P % 2 == 1)
    for (int t = 0; t < dr.T; ++t) m.ll[(size_t)t] += n * 0.5 * std::log(2 * 3.14159265358979323846 * dr.sigma[(size_t)t]);
  return 0;
}

int mv_loglik_at_means(const bfmmm_post_args* a, const Draws& dr, double* out) {
  const int kept = (int)std::round((1 - a->burnin_prop) * dr.T);
  if (kept < 1) return bfmmm_io_fail("'burnin_prop' leaves no draws");
  MVRun m;
  if (run_mv(a, dr, dr.T - kept, false, true, m)) return 1;
  double ms = 0.0;
  for (double s : dr.sigma) ms += s;
  ms /= (double)dr.sigma.size();
  const double sd = std::sqrt(ms);
  double ll = 0.0;
  for (size_t e = 0; e < m.fit.size(); ++e) {
    const double z = (m.y[e] - m.fit[e]) / sd;
    ll += -(0.91893853320467274178 + 0.5 * z * z + std::log(sd));
  }
  *out = ll;
  return 0;
}

}  // namespace

extern "C" int bfmmm_MVLLik(const bfmmm_post_args* a, bfmmm_result** out) {
  if (!out) return bfmmm_io_fail("null argument");
  if (check(a, false, true)) return 1;
  Draws dr;
  if (load_draws(a, dr, true)) return 1;
  MVRun m;
  if (run_mv(a, dr, 0, false, false, m)) return 1;
  bfmmm_result* r = bfmmm_result_create();
  const int64_t T = dr.T;
  bfmmm_result_set(r, "value", m.ll.data(), T, &T, 1);
  *out = r;
  return 0;
}

extern "C" int bfmmm_MVDIC(const bfmmm_post_args* a, double* out) {
  if (!out) return bfmmm_io_fail("null argument");
  if (check(a, true, true)) return 1;
  Draws dr;
  if (load_draws(a, dr, true)) return 1;
  const int kept = (int)std::round((1 - a->burnin_prop) * dr.T);
  if (kept < 1) return bfmmm_io_fail("'burnin_prop' leaves no draws");
  MVRun m;
  if (run_mv(a, dr, dr.T - kept, true, false, m)) return 1;
  double expected_log_f = 0.0;                               // PostProcessing.cpp:5863-5869
  for (int t = dr.T - kept; t < dr.T; ++t) expected_log_f += m.ll[(size_t)t];
  expected_log_f /= kept;
  double f_hat = 0.0;                                        // :5871-5882 (joint density of a row, calcDIC2MV)
  for (double v : m.joint) f_hat += std::log(v);
  *out = 2 * f_hat - 4 * expected_log_f;
  return 0;
}

extern "C" int bfmmm_MVAIC(const bfmmm_post_args* a, double* out) {
  if (!out) return bfmmm_io_fail("null argument");
  if (check(a, true, true)) return 1;
  Draws dr;
  if (load_draws(a, dr, true)) return 1;
  double ll;
  if (mv_loglik_at_means(a, dr, &ll)) return 1;
  *out = 2 * n_params(dr, a->X != nullptr, a->cov_adj != 0) - 2 * ll;      // PostProcessing.cpp:5225-5227, :5421-5430
  return 0;
}

extern "C" int bfmmm_MVBIC(const bfmmm_post_args* a, double* out) {
  if (!out) return bfmmm_io_fail("null argument");
  if (check(a, true, true)) return 1;
  Draws dr;
  if (load_draws(a, dr, true)) return 1;
  double ll;
  if (mv_loglik_at_means(a, dr, &ll)) return 1;
  *out = 2 * ll - std::log((double)a->n_funct) * n_params(dr, a->X != nullptr, a->cov_adj != 0);      // :5560 (log(Y.n_rows))
  return 0;
}

// ---- credible intervals: SigmaCI, ZCI, FMeanCI -----------------------------------------------------------------------
namespace {

int ci_check(const bfmmm_ci_args* a, bool basis) {       // order and wording of PostProcessing.cpp:116-141, :3439-3453
  if (!a || !a->dir) return bfmmm_io_fail("null argument");
  if (a->n_files <= 0) return bfmmm_io_fail("'n_files' must be greater than 0");
  if (a->alpha < 0 || a->alpha >= 1) return bfmmm_io_fail("'alpha' must be between 0 and 1");
  if (a->burnin_prop < 0 || a->burnin_prop >= 1) return bfmmm_io_fail("'burnin_prop' must be between 0 and 1");
  if (!basis) return 0;
  if (!a->time || a->n_time < 1 || !a->boundary_knots) return bfmmm_io_fail("null argument");
  if (a->dim > 0) {      // HDFMeanCI, PostProcessing.cpp:836-852
    if (!a->basis_degree_hd || !a->n_internal_hd) return bfmmm_io_fail("null argument");
    size_t ko = 0;
    for (int j = 0; j < a->dim; ++j) {
      if (a->basis_degree_hd[j] < 1) return bfmmm_io_fail("'basis_degree' must be an integer greater than or equal to 1");
      for (int i = 0; i < a->n_internal_hd[j]; ++i) {
        if (a->boundary_knots[2 * j] >= a->internal_knots[ko + i])
          return bfmmm_io_fail("at least one element in 'internal_knots' is less than or equal to first boundary knot");
        if (a->boundary_knots[2 * j + 1] <= a->internal_knots[ko + i])
          return bfmmm_io_fail("at least one element in 'internal_knots' is more than or equal to second boundary knot");
      }
      ko += (size_t)a->n_internal_hd[j];
    }
    return 0;
  }
  if (a->basis_degree < 1) return bfmmm_io_fail("'basis_degree' must be an integer greater than or equal to 1");
  for (int i = 0; i < a->n_internal_knots; ++i) {
    if (a->boundary_knots[0] >= a->internal_knots[i])
      return bfmmm_io_fail("at least one element in 'internal_knots' is less than or equal to first boundary knot");
    if (a->boundary_knots[1] <= a->internal_knots[i])
      return bfmmm_io_fail("at least one element in 'internal_knots' is more than or equal to second boundary knot");
  }
  return 0;
}

// transform_mat of a draw (K x K, column-major): row i = the Z row of the observation with the largest Z(., i)  (:258-268)
void transform_of(const double* Zt, int n, int K, std::vector<double>& Tm) {
  Tm.assign((size_t)K * K, 0.0);
  for (int i = 0; i < K; ++i) {
    int mx = 0;
    for (int l = 1; l < n; ++l) if (Zt[l + (size_t)n * i] > Zt[mx + (size_t)n * i]) mx = l;      // arma::index_max: first maximum
    for (int c = 0; c < K; ++c) Tm[i + (size_t)K * c] = Zt[mx + (size_t)n * c];
  }
}

void put_mat(bfmmm_result* r, const char* name, const std::vector<double>& v, int64_t d0, int64_t d1, int64_t d2 = -1) {
  const int64_t dims[3] = {d0, d1, d2};
  bfmmm_result_set(r, name, v.data(), (int64_t)v.size(), dims, d2 < 0 ? (d1 < 0 ? 1 : 2) : 3);
}

}  // namespace

extern "C" void bfmmm_ci_defaults(bfmmm_ci_args* a) {
  *a = bfmmm_ci_args{};
  a->alpha = 0.05; a->rescale = 1; a->burnin_prop = 0.1; a->k = 1;
}

extern "C" int bfmmm_SigmaCI(const bfmmm_ci_args* a, bfmmm_result** out) {
  if (!out) return bfmmm_io_fail("null argument");
  if (ci_check(a, false)) return 1;
  std::vector<double> sig;
  int64_t d[3];
  if (load_cat(a->dir, "Sigma", a->n_files, sig, d)) return 1;
  const int T = (int)sig.size(), kept = (int)std::round(T * (1 - a->burnin_prop));
  if (kept < 1) return bfmmm_io_fail("'burnin_prop' leaves no draws");
  const double probs[3] = {a->alpha / 2, 0.5, 1 - a->alpha / 2};
  double q[3];
  if (bfmmm_post_col_quantiles(sig.data() + (T - kept), kept, 1, probs, 3, a->device, q)) return 1;
  bfmmm_result* r = bfmmm_result_create();
  const int64_t one = 1;
  bfmmm_result_set(r, "CI_Upper", &q[2], 1, &one, 1);
  bfmmm_result_set(r, "CI_50", &q[1], 1, &one, 1);
  bfmmm_result_set(r, "CI_Lower", &q[1], 1, &one, 1);      // PostProcessing.cpp:3496: the reference returns q(1) here
  *out = r;
  return 0;
}

extern "C" int bfmmm_ZCI(const bfmmm_ci_args* a, bfmmm_result** out) {
  if (!out || !a || !a->dir) return bfmmm_io_fail("null argument");      // (the reference checks nothing here)
  std::vector<double> Z;
  int64_t d[3];
  if (load_cat(a->dir, "Z", a->n_files, Z, d)) return 1;
  const int n = (int)d[0], K = (int)d[1], T = (int)d[2] * a->n_files;
  bool rescale = a->rescale != 0;
  if (rescale && K > 2) rescale = false;                     // "Rescale property cannot be used for K > 2"
  if (rescale) {      // Z_j <- solve(T', Z_j')' = Z_j T^-1 (:3527-3541), K = 2 (a 1-cluster model never gets here)
    std::vector<double> Tm;
    for (int j = 0; j < T; ++j) {
      double* Zt = Z.data() + (size_t)n * K * j;
      transform_of(Zt, n, K, Tm);
      if (K == 1) { for (int l = 0; l < n; ++l) Zt[l] /= Tm[0]; continue; }
      const double t00 = Tm[0], t10 = Tm[1], t01 = Tm[2], t11 = Tm[3], det = t00 * t11 - t01 * t10;
      for (int l = 0; l < n; ++l) {
        const double z0 = Zt[l], z1 = Zt[l + n];
        Zt[l] = (z0 * t11 - z1 * t10) / det;
        Zt[l + n] = (z1 * t00 - z0 * t01) / det;
      }
    }
  }
  const int kept = (int)std::round(T * (1 - a->burnin_prop));
  if (kept < 1) return bfmmm_io_fail("'burnin_prop' leaves no draws");
  std::vector<double> V((size_t)kept * n * K);               // column (i, j) = the kept draws of Z(i, j)
  for (int c = 0; c < n * K; ++c)
    for (int l = 0; l < kept; ++l) V[(size_t)l + (size_t)kept * c] = Z[(size_t)c + (size_t)n * K * (T - kept + l)];
  const double probs[3] = {a->alpha / 2, 0.5, 1 - a->alpha / 2};
  std::vector<double> q((size_t)3 * n * K);
  if (bfmmm_post_col_quantiles(V.data(), kept, n * K, probs, 3, a->device, q.data())) return 1;
  std::vector<double> up((size_t)n * K), md((size_t)n * K), lo((size_t)n * K);
  for (int c = 0; c < n * K; ++c) { lo[(size_t)c] = q[(size_t)3 * c]; md[(size_t)c] = q[(size_t)3 * c + 1]; up[(size_t)c] = q[(size_t)3 * c + 2]; }
  bfmmm_result* r = bfmmm_result_create();
  put_mat(r, "CI_Upper", up, n, K); put_mat(r, "CI_50", md, n, K); put_mat(r, "CI_Lower", lo, n, K);
  const int first = (int)std::round(T * a->burnin_prop);     // :3563: Z_trace starts at round(T * burnin_prop)
  std::vector<double> tr(Z.begin() + (size_t)n * K * first, Z.end());
  put_mat(r, "Z_trace", tr, n, K, T - first);
  *out = r;
  return 0;
}

extern "C" int bfmmm_FMeanCI(const bfmmm_ci_args* a, bfmmm_result** out) {
  if (!out) return bfmmm_io_fail("null argument");
  if (ci_check(a, true)) return 1;
  const std::string dir = a->dir;
  std::vector<double> nu;
  int64_t d[3], nr, nc;
  if (load_cat(dir, "Nu", a->n_files, nu, d)) return 1;
  const int K = (int)d[0], P = (int)d[1], per_file = (int)d[2], T = per_file * a->n_files;
  if (a->k <= 0) return bfmmm_io_fail("'k' must be positive");
  if (a->k > K) return bfmmm_io_fail("'k' must be less than or equal to the number of clusters in the model");
  int Pb = a->n_internal_knots + a->basis_degree + 1;
  if (a->dim > 0) { Pb = 1; for (int j = 0; j < a->dim; ++j) Pb *= a->n_internal_hd[j] + a->basis_degree_hd[j] + 1; }
  if (P != Pb) return bfmmm_io_fail("the saved draws do not match the basis ('basis_degree', 'internal_knots')");
  const int kept = (int)std::round(T * (1 - a->burnin_prop)), first = T - kept;
  if (kept < 2) return bfmmm_io_fail("'burnin_prop' leaves fewer than two draws");
  bool rescale = a->rescale != 0;
  if (rescale && K > 2) rescale = false;                     // "Rescale property cannot be used for K > 2" (:157-162)
  int D = 0;
  std::vector<double> eta;                                   // kept cubes P x D x K
  if (a->X) {
    for (int q = 0; q < a->n_files; ++q) {
      std::vector<std::vector<double>> objs;
      int64_t de[3];
      if (arma_load_field(dir + "Eta" + std::to_string(q) + ".txt", objs, &nr, &nc, de)) return 1;
      if (q == 0) {
        D = (int)de[1];
        if (a->D != D) return bfmmm_io_fail("The number of columns in 'X' must be equal to the number of covariates in the model");
      }
      for (int l = 0; l < per_file; ++l)
        if (q * per_file + l >= first) eta.insert(eta.end(), objs[(size_t)l].begin(), objs[(size_t)l].end());
    }
  }
  std::vector<double> nuk(nu.begin() + (size_t)K * P * first, nu.end());      // kept slices K x P
  if (rescale) {
    std::vector<double> Z, Tm, tmp((size_t)K * P), te((size_t)K);
    int64_t dz[3];
    if (load_cat(dir, "Z", a->n_files, Z, dz)) return 1;
    const int n = (int)dz[0];
    for (int j = 0; j < kept; ++j) {
      transform_of(Z.data() + (size_t)n * K * (first + j), n, K, Tm);
      double* nj = nuk.data() + (size_t)K * P * j;
      for (int p = 0; p < P; ++p)
        for (int i = 0; i < K; ++i) { double s_ = 0.0; for (int c = 0; c < K; ++c) s_ += Tm[i + (size_t)K * c] * nj[c + (size_t)K * p]; tmp[i + (size_t)K * p] = s_; }
      std::copy(tmp.begin(), tmp.end(), nj);
      for (int dd = 0; dd < D; ++dd)                          // eta_d <- T eta_d over the cluster index (:462-472)
        for (int p = 0; p < P; ++p) {
          double* ej = eta.data() + (size_t)P * D * K * j + p + (size_t)P * dd;
          for (int i = 0; i < K; ++i) { double s_ = 0.0; for (int c = 0; c < K; ++c) s_ += Tm[i + (size_t)K * c] * ej[(size_t)P * D * c]; te[(size_t)i] = s_; }
          for (int i = 0; i < K; ++i) ej[(size_t)P * D * i] = te[(size_t)i];
        }
    }
  } else if (a->trans_mats) {                                // (:271-279, with X :486-502)
    std::vector<double> tmp((size_t)K * P), te((size_t)K);
    const size_t ld = (size_t)kept * K;
    for (int j = 0; j < kept; ++j) {
      double* nj = nuk.data() + (size_t)K * P * j;
      auto tm = [&](int i, int c) { return a->trans_mats[(size_t)j * K + i + ld * c]; };
      for (int p = 0; p < P; ++p)
        for (int i = 0; i < K; ++i) { double s_ = 0.0; for (int c = 0; c < K; ++c) s_ += tm(i, c) * nj[c + (size_t)K * p]; tmp[i + (size_t)K * p] = s_; }
      std::copy(tmp.begin(), tmp.end(), nj);
      for (int dd = 0; dd < D; ++dd)
        for (int p = 0; p < P; ++p) {
          double* ej = eta.data() + (size_t)P * D * K * j + p + (size_t)P * dd;
          for (int i = 0; i < K; ++i) { double s_ = 0.0; for (int c = 0; c < K; ++c) s_ += tm(i, c) * ej[(size_t)P * D * c]; te[(size_t)i] = s_; }
          for (int i = 0; i < K; ++i) ej[(size_t)P * D * i] = te[(size_t)i];
        }
    }
  }
  // basis at the band's time points
  const int nt = a->n_time;
  std::vector<double> cm((size_t)nt * P), B((size_t)nt * P);
  const int32_t deg = a->basis_degree, nint = a->n_internal_knots;
  if (a->dim > 0 ? bfmmm_tensor_bspline(nt, a->dim, a->time, a->basis_degree_hd, a->boundary_knots, a->n_internal_hd, a->internal_knots, cm.data())
                 : bfmmm_tensor_bspline(nt, 1, a->time, &deg, a->boundary_knots, &nint, a->internal_knots, cm.data())) return 1;
  for (int l = 0; l < nt; ++l)
    for (int p = 0; p < P; ++p) B[(size_t)l * P + p] = cm[(size_t)l + (size_t)nt * p];
  const int k0 = a->k - 1, nx = a->X ? a->n_x : 1;
  std::vector<double> up((size_t)nx * nt), md((size_t)nx * nt), lo((size_t)nx * nt), trace((size_t)nx * nt * kept);
  std::vector<double> coef((size_t)kept * P), u1((size_t)nt), m1((size_t)nt), l1((size_t)nt), tr1((size_t)kept * nt);
  for (int x = 0; x < nx; ++x) {
    for (int j = 0; j < kept; ++j)
      for (int p = 0; p < P; ++p) {
        double v = nuk[(size_t)K * P * j + k0 + (size_t)K * p];
        for (int dd = 0; dd < D; ++dd) v += eta[(size_t)P * D * K * j + p + (size_t)P * (dd + (size_t)D * k0)] * a->X[x + (size_t)nx * dd];
        coef[(size_t)j * P + p] = v;
      }
    if (bfmmm_post_bands(coef.data(), kept, P, B.data(), nt, a->alpha, a->simultaneous, a->device, u1.data(), m1.data(), l1.data(), tr1.data())) return 1;
    for (int l = 0; l < nt; ++l) {
      up[x + (size_t)nx * l] = u1[(size_t)l]; md[x + (size_t)nx * l] = m1[(size_t)l]; lo[x + (size_t)nx * l] = l1[(size_t)l];
      for (int j = 0; j < kept; ++j)
        if (a->X) trace[x + (size_t)nx * (l + (size_t)nt * j)] = tr1[(size_t)j + (size_t)kept * l];
        else trace[(size_t)j + (size_t)kept * l] = tr1[(size_t)j + (size_t)kept * l];
    }
  }
  bfmmm_result* r = bfmmm_result_create();
  if (a->X) {
    put_mat(r, "CI_Upper", up, nx, nt); put_mat(r, "CI_50", md, nx, nt); put_mat(r, "CI_Lower", lo, nx, nt);
    put_mat(r, "mean_trace", trace, nx, nt, kept);
  } else {
    put_mat(r, "CI_Upper", up, nt, -1); put_mat(r, "CI_50", md, nt, -1); put_mat(r, "CI_Lower", lo, nt, -1);
    put_mat(r, "mean_trace", trace, kept, nt);
  }
  *out = r;
  return 0;
}

// FCovCI (src/PostProcessing.cpp:1781), HDFCovCI (:2468) and MVCovCI (:3097): credible bands of the covariance surface
// between clusters l and m, sum_j (B1 c_lj)(B2 c_mj)' over the kept draws, where c_kj = phi_kj without covariates and
// c_kj = phi_kj + xi_kj x_b for every row x_b of `X` (the covariate-dependent covariance of the reference's second branch:
// :2199-2206, :2894-2901, :3375-3382).  kind: 0 functional (B-splines on time1 / time2), 1 high-dimensional (tensor-product
// basis), 2 multivariate (identity basis, pointwise bands only).
// Quirks of the reference that are kept: trans_mats is honoured only in FCovCI's branch WITHOUT covariates (:1927, :1992);
// the rescale transform of the xi cubes is xi_k <- sum_b T(k, b) xi_b (:2175-2185); HDFCovCI evaluates BOTH bases at time1
// (:2570 passes time1field twice), so its surface is n_time x n_time and time2 only has to have as many rows; CI_Lower is
// allocated n_time2 x n_time2 (:1879): n_time > n_time2 is refused.
static int cov_ci_impl(const bfmmm_ci_args* a, int kind, bfmmm_result** out) {
  if (!out) return bfmmm_io_fail("null argument");
  if (ci_check(a, kind != 2)) return 1;
  if (kind == 1 && a->dim <= 0) return bfmmm_io_fail("bfmmm_HDFCovCI: args.dim must be positive");
  if (kind == 0 && a->dim > 0) return bfmmm_io_fail("bfmmm_FCovCI: args.dim must be 0 (bfmmm_HDFCovCI is the high-dimensional function)");
  if (kind != 2 && (!a->time2 || a->n_time2 < 1)) return bfmmm_io_fail("null argument");
  if (kind == 0 && a->n_time > a->n_time2) return bfmmm_io_fail("FCovCI: 'time1' longer than 'time2' overruns the reference's CI_Lower (PostProcessing.cpp:1879); not supported");
  if (kind == 1 && a->n_time != a->n_time2) return bfmmm_io_fail("HDFCovCI: 'time1' and 'time2' must have the same number of rows (the reference evaluates both bases at 'time1', PostProcessing.cpp:2570)");
  if (a->X && (a->n_x < 1 || a->D < 1)) return bfmmm_io_fail("null argument");
  const std::string dir = a->dir;
  std::vector<double> sig;
  int64_t d[3], nr, nc;
  if (load_cat(dir, "Sigma", a->n_files, sig, d)) return 1;
  const int T = (int)sig.size(), per_file = T / a->n_files;
  std::vector<std::vector<double>> phi;                      // T cubes K x P x M
  int K = 0, P = 0, M = 0;
  for (int q = 0; q < a->n_files; ++q) {
    std::vector<std::vector<double>> objs;
    if (arma_load_field(dir + "Phi" + std::to_string(q) + ".txt", objs, &nr, &nc, d)) return 1;
    if (q == 0) { K = (int)d[0]; P = (int)d[1]; M = (int)d[2]; }
    if ((int)nr < per_file) return bfmmm_io_fail("'Phi<q>.txt' holds fewer draws than 'Sigma<q>.txt'");
    for (int l = 0; l < per_file; ++l) phi.push_back(std::move(objs[(size_t)l]));
  }
  if (a->l <= 0) return bfmmm_io_fail("'l' must be positive");
  if (a->l > K) return bfmmm_io_fail("'l' must be less than or equal to the number of clusters in the model");
  if (a->m <= 0) return bfmmm_io_fail("'m' must be positive");
  if (a->m > K) return bfmmm_io_fail("'m' must be less than or equal to the number of clusters in the model");
  if (kind != 2) {
    int Pb = a->n_internal_knots + a->basis_degree + 1;
    if (a->dim > 0) { Pb = 1; for (int j = 0; j < a->dim; ++j) Pb *= a->n_internal_hd[j] + a->basis_degree_hd[j] + 1; }
    if (P != Pb) return bfmmm_io_fail("the saved draws do not match the basis ('basis_degree', 'internal_knots')");
  }
  const int kept = (int)std::round(T * (1 - a->burnin_prop)), first = T - kept;
  if (kept < 2) return bfmmm_io_fail("'burnin_prop' leaves fewer than two draws");
  bool rescale = a->rescale != 0;
  if (rescale && K > 2) rescale = false;
  // xi: T x K cubes P x D x M (field object (i, k) at i + n_rows * k)
  const int D = a->X ? a->D : 0, W = a->X ? a->n_x : 1;
  std::vector<std::vector<double>> xi;                       // [(draw) * K + k]
  if (a->X) {
    xi.resize((size_t)T * K);
    for (int q = 0; q < a->n_files; ++q) {
      std::vector<std::vector<double>> objs;
      int64_t dx[3];
      if (arma_load_field(dir + "Xi" + std::to_string(q) + ".txt", objs, &nr, &nc, dx)) return 1;
      if ((int)nc != K || (int)nr < per_file) return bfmmm_io_fail("'Xi<q>.txt' does not match 'Phi<q>.txt'");
      if (q == 0 && (int)dx[1] != D) return bfmmm_io_fail("The number of columns in 'X' must be equal to the number of covariates in the model");
      for (int l = 0; l < per_file; ++l)
        for (int k = 0; k < K; ++k) xi[(size_t)(q * per_file + l) * K + k] = std::move(objs[(size_t)l + (size_t)nr * k]);
    }
  }
  // rescale: Phi.slice(b) <- T Phi.slice(b); xi_k <- sum_b T(k, b) xi_b.  trans_mats (FCovCI without covariates): both apply
  std::vector<double> Z, Tm, tmp((size_t)K);
  int n = 0;
  if (rescale) {
    int64_t dz[3];
    if (load_cat(dir, "Z", a->n_files, Z, dz)) return 1;
    n = (int)dz[0];
  }
  const bool use_tm = a->trans_mats && kind == 0 && !a->X;
  const size_t ld = (size_t)kept * K;
  for (int j = 0; j < kept; ++j) {
    double* ph = phi[(size_t)(first + j)].data();
    for (int pass = 0; pass < 2; ++pass) {
      if (pass == 0) { if (!rescale) continue; transform_of(Z.data() + (size_t)n * K * (first + j), n, K, Tm); }
      else { if (!use_tm) continue; Tm.assign((size_t)K * K, 0.0); for (int i = 0; i < K; ++i) for (int c = 0; c < K; ++c) Tm[i + (size_t)K * c] = a->trans_mats[(size_t)j * K + i + ld * c]; }
      for (int b = 0; b < M; ++b)
        for (int p = 0; p < P; ++p) {
          double* col = ph + (size_t)K * (p + (size_t)P * b);
          for (int i = 0; i < K; ++i) { double s_ = 0.0; for (int c = 0; c < K; ++c) s_ += Tm[i + (size_t)K * c] * col[c]; tmp[(size_t)i] = s_; }
          for (int i = 0; i < K; ++i) col[i] = tmp[(size_t)i];
        }
      if (pass == 0 && a->X) {
        std::vector<std::vector<double>> old((size_t)K);
        for (int k = 0; k < K; ++k) old[(size_t)k] = xi[(size_t)(first + j) * K + k];
        for (int k = 0; k < K; ++k) {
          std::vector<double>& dst = xi[(size_t)(first + j) * K + k];
          for (size_t e = 0; e < dst.size(); ++e) {
            double s_ = old[0][e] * Tm[k + (size_t)K * 0];
            for (int b = 1; b < K; ++b) s_ = s_ + (old[(size_t)b][e] * Tm[k + (size_t)K * b]);
            dst[e] = s_;
          }
        }
      }
    }
  }
  // bases (row-major n x P)
  int n1 = a->n_time, n2 = a->n_time2;
  std::vector<double> B1, B2;
  if (kind == 2) {
    n1 = n2 = P;
    B1.assign((size_t)P * P, 0.0);
    for (int p = 0; p < P; ++p) B1[(size_t)p * P + p] = 1.0;
    B2 = B1;
  } else {
    const int32_t deg = a->basis_degree, nint = a->n_internal_knots;
    auto basis = [&](const double* t, int nt, std::vector<double>& B) {
      std::vector<double> cm((size_t)nt * P);
      if (a->dim > 0 ? bfmmm_tensor_bspline(nt, a->dim, t, a->basis_degree_hd, a->boundary_knots, a->n_internal_hd, a->internal_knots, cm.data())
                     : bfmmm_tensor_bspline(nt, 1, t, &deg, a->boundary_knots, &nint, a->internal_knots, cm.data())) return 1;
      B.resize((size_t)nt * P);
      for (int l = 0; l < nt; ++l)
        for (int p = 0; p < P; ++p) B[(size_t)l * P + p] = cm[(size_t)l + (size_t)nt * p];
      return 0;
    };
    if (basis(a->time, n1, B1)) return 1;
    if (kind == 1) B2 = B1;                                   // the reference's second basis is built from time1 as well (:2570)
    else if (basis(a->time2, n2, B2)) return 1;
  }
  const size_t cells = (size_t)n1 * n2;
  const int simultaneous = (kind == 2) ? 0 : a->simultaneous;
  std::vector<double> up(cells * W), md(cells * W), lo(cells * W), cube(cells * kept * W), tr(cells * kept);
  std::vector<double> cl((size_t)kept * M * P), cmv((size_t)kept * M * P);
  for (int w = 0; w < W; ++w) {
    for (int j = 0; j < kept; ++j) {
      const double* ph = phi[(size_t)(first + j)].data();
      for (int b = 0; b < M; ++b)
        for (int p = 0; p < P; ++p) {
          double vl = ph[(a->l - 1) + (size_t)K * (p + (size_t)P * b)], vm = ph[(a->m - 1) + (size_t)K * (p + (size_t)P * b)];
          if (a->X) {
            const double* xl = xi[(size_t)(first + j) * K + (a->l - 1)].data();
            const double* xm = xi[(size_t)(first + j) * K + (a->m - 1)].data();
            double sl = 0.0, sm = 0.0;
            for (int dd = 0; dd < D; ++dd) {
              const double x = a->X[(size_t)w + (size_t)a->n_x * dd];
              sl += xl[p + (size_t)P * (dd + (size_t)D * b)] * x;
              sm += xm[p + (size_t)P * (dd + (size_t)D * b)] * x;
            }
            vl += sl; vm += sm;
          }
          cl[((size_t)j * M + b) * P + p] = vl;
          cmv[((size_t)j * M + b) * P + p] = vm;
        }
    }
    if (bfmmm_post_cov_bands(cl.data(), cmv.data(), kept, M, P, B1.data(), n1, B2.data(), n2, a->alpha, simultaneous, a->device,
                             up.data() + cells * w, md.data() + cells * w, lo.data() + cells * w, tr.data())) return 1;
    double* cw = cube.data() + cells * kept * w;
    for (size_t c = 0; c < cells; ++c)
      for (int j = 0; j < kept; ++j) cw[c + cells * j] = tr[(size_t)j + (size_t)kept * c];
  }
  bfmmm_result* r = bfmmm_result_create();
  if (!a->X) {
    put_mat(r, "CI_Upper", up, n1, n2); put_mat(r, "CI_50", md, n1, n2); put_mat(r, "CI_Lower", lo, n1, n2);
    put_mat(r, "cov_trace", cube, n1, n2, kept);
  } else {      // cubes n1 x n2 x n_x; cov_trace: the n_x cubes of the reference's field, one after the other (n1 x n2 x kept x n_x)
    put_mat(r, "CI_Upper", up, n1, n2, W); put_mat(r, "CI_50", md, n1, n2, W); put_mat(r, "CI_Lower", lo, n1, n2, W);
    const int64_t dims[4] = {n1, n2, kept, W};
    bfmmm_result_set(r, "cov_trace", cube.data(), (int64_t)cube.size(), dims, 4);
  }
  *out = r;
  return 0;
}

extern "C" int bfmmm_FCovCI(const bfmmm_ci_args* a, bfmmm_result** out) { return cov_ci_impl(a, 0, out); }
extern "C" int bfmmm_HDFCovCI(const bfmmm_ci_args* a, bfmmm_result** out) { return cov_ci_impl(a, 1, out); }
extern "C" int bfmmm_MVCovCI(const bfmmm_ci_args* a, bfmmm_result** out) { return cov_ci_impl(a, 2, out); }

// FSamplePaths (src/PostProcessing.cpp:6599-6864): posterior-predictive sample paths of every curve at its own time points
// -- for each kept draw the fitted value plus N(0, sigma^2) noise (:6810; the noise comes from the keyed generator, the
// reference's from R's stream) -- with pointwise (:6817-6828) or simultaneous (:6829-6855) bands per curve, the paths
// ("Path_trace") and the mean-only paths (nu and eta terms, "Mean_only_Path_trace").  Without X the reference runs with
// X = 0 and eta = 0; cov_adj = 0 leaves xi = 0.  Result elements are concatenated over the curves: the bands have
// offsets[n] entries, the two traces kept x offsets[n] (curve i: a kept x n_i block, draw fastest, at kept * offsets[i]).
// (In the simultaneous branch the reference sizes its scratch vector by the number of CURVES (:6837) and so fails when a
// curve has more points than there are curves; here the maximum runs over the curve's own points.)
extern "C" int bfmmm_FSamplePaths(const bfmmm_post_args* a, double alpha, int32_t simultaneous, uint64_t seed, bfmmm_result** out) {
  if (!out || !a || !a->dir || !a->t || !a->offsets || !a->boundary_knots || (a->n_internal_knots > 0 && !a->internal_knots))
    return bfmmm_io_fail("null argument");
  if (a->n_files <= 0) return bfmmm_io_fail("'n_files' must be greater than 0");
  if (alpha < 0 || alpha >= 1) return bfmmm_io_fail("'alpha' must be between 0 and 1");
  if (a->burnin_prop < 0 || a->burnin_prop >= 1) return bfmmm_io_fail("'burnin_prop' must be between 0 and 1");
  if (a->basis_degree < 1) return bfmmm_io_fail("'basis_degree' must be an integer greater than or equal to 1");
  for (int i = 0; i < a->n_internal_knots; ++i) {
    if (a->boundary_knots[0] >= a->internal_knots[i])
      return bfmmm_io_fail("at least one element in 'internal_knots' is less than or equal to first boundary knot");
    if (a->boundary_knots[1] <= a->internal_knots[i])
      return bfmmm_io_fail("at least one element in 'internal_knots' is more than or equal to second boundary knot");
  }
  Draws dr;
  if (load_draws(a, dr)) return 1;
  const int n = dr.n, T = dr.T;
  const int kept = (int)std::round(T * (1 - a->burnin_prop)), first = T - kept;
  if (kept < 2) return bfmmm_io_fail("'burnin_prop' leaves fewer than two draws");
  const int64_t n_obs = a->offsets[n];
  bfmmm_post_input in{};
  in.n = n; in.K = dr.K; in.P = dr.P; in.M = dr.M; in.D = dr.D;
  in.offsets = a->offsets; in.y = nullptr; in.B = dr.B.data(); in.X = a->X;
  in.T = T; in.nu = dr.nu.data(); in.Phi = dr.Phi.data(); in.Z = dr.Z.data(); in.chi = dr.chi.data(); in.sigma = dr.sigma.data();
  in.eta = dr.eta.empty() ? nullptr : dr.eta.data();
  in.xi = dr.xi.empty() ? nullptr : dr.xi.data();
  in.device = a->device;
  std::vector<double> paths((size_t)kept * n_obs), mo((size_t)kept * n_obs), up((size_t)n_obs), md((size_t)n_obs), lo((size_t)n_obs);
  if (bfmmm_post_sample_paths(&in, first, seed, paths.data(), mo.data())) return 1;
  if (!simultaneous) {      // every column on its own: one call for all curves
    if (bfmmm_post_table_bands(paths.data(), kept, (int)n_obs, alpha, 0, a->device, up.data(), md.data(), lo.data())) return 1;
  } else {
    for (int i = 0; i < n; ++i) {
      const int64_t o = a->offsets[i], ni = a->offsets[i + 1] - o;
      if (ni < 1) continue;
      if (bfmmm_post_table_bands(paths.data() + (size_t)kept * o, kept, (int)ni, alpha, 1, a->device, up.data() + o, md.data() + o, lo.data() + o))
        return 1;
    }
  }
  bfmmm_result* r = bfmmm_result_create();
  put_mat(r, "CI_Upper", up, n_obs, -1); put_mat(r, "CI_50", md, n_obs, -1); put_mat(r, "CI_Lower", lo, n_obs, -1);
  put_mat(r, "Path_trace", paths, kept, n_obs); put_mat(r, "Mean_only_Path_trace", mo, kept, n_obs);
  *out = r;
  return 0;
}

extern "C" int bfmmm_MVMeanCI(const bfmmm_ci_args* a, bfmmm_result** out) {
  if (!out) return bfmmm_io_fail("null argument");
  if (ci_check(a, false)) return 1;
  const std::string dir = a->dir;
  std::vector<double> nu;
  int64_t d[3], nr, nc;
  if (load_cat(dir, "Nu", a->n_files, nu, d)) return 1;
  const int K = (int)d[0], P = (int)d[1], per_file = (int)d[2], T = per_file * a->n_files;
  const int kept = (int)std::round(T * (1 - a->burnin_prop)), first = T - kept;
  if (kept < 1) return bfmmm_io_fail("'burnin_prop' leaves no draws");
  bool rescale = a->rescale != 0;
  if (rescale && K > 2) rescale = false;
  int D = 0;
  std::vector<double> eta;
  if (a->X) {
    for (int q = 0; q < a->n_files; ++q) {
      std::vector<std::vector<double>> objs;
      int64_t de[3];
      if (arma_load_field(dir + "Eta" + std::to_string(q) + ".txt", objs, &nr, &nc, de)) return 1;
      if (q == 0) {
        D = (int)de[1];
        if (a->D != D) return bfmmm_io_fail("The number of columns in 'X' must be equal to the number of covariates in the model");
      }
      for (int l = 0; l < per_file; ++l)
        if (q * per_file + l >= first) eta.insert(eta.end(), objs[(size_t)l].begin(), objs[(size_t)l].end());
    }
  }
  std::vector<double> nuk(nu.begin() + (size_t)K * P * first, nu.end());
  if (rescale) {
    std::vector<double> Z, Tm, tmp((size_t)K);
    int64_t dz[3];
    if (load_cat(dir, "Z", a->n_files, Z, dz)) return 1;
    const int n = (int)dz[0];
    for (int j = 0; j < kept; ++j) {
      transform_of(Z.data() + (size_t)n * K * (first + j), n, K, Tm);
      for (int p = 0; p < P; ++p) {
        double* col = nuk.data() + (size_t)K * P * j + (size_t)K * p;
        for (int i = 0; i < K; ++i) { double s_ = 0.0; for (int c = 0; c < K; ++c) s_ += Tm[i + (size_t)K * c] * col[c]; tmp[(size_t)i] = s_; }
        for (int i = 0; i < K; ++i) col[i] = tmp[(size_t)i];
        for (int dd = 0; dd < D; ++dd) {
          double* ej = eta.data() + (size_t)P * D * K * j + p + (size_t)P * dd;
          for (int i = 0; i < K; ++i) { double s_ = 0.0; for (int c = 0; c < K; ++c) s_ += Tm[i + (size_t)K * c] * ej[(size_t)P * D * c]; tmp[(size_t)i] = s_; }
          for (int i = 0; i < K; ++i) ej[(size_t)P * D * i] = tmp[(size_t)i];
        }
      }
    }
  }
  const int nx = a->X ? a->n_x : 1, KP = K * P;
  std::vector<double> V((size_t)kept * KP * nx), trace((size_t)KP * kept * nx);
  for (int x = 0; x < nx; ++x)
    for (int j = 0; j < kept; ++j)
      for (int p = 0; p < P; ++p)
        for (int k = 0; k < K; ++k) {
          double v = nuk[(size_t)KP * j + k + (size_t)K * p];
          for (int dd = 0; dd < D; ++dd) v += eta[(size_t)P * D * K * j + p + (size_t)P * (dd + (size_t)D * k)] * a->X[x + (size_t)nx * dd];
          V[(size_t)j + (size_t)kept * (k + (size_t)K * p + (size_t)KP * x)] = v;
          trace[(size_t)k + (size_t)K * p + (size_t)KP * (j + (size_t)kept * x)] = v;
        }
  const double probs[3] = {a->alpha / 2, 0.5, 1 - a->alpha / 2};
  std::vector<double> q((size_t)3 * KP * nx);
  if (bfmmm_post_col_quantiles(V.data(), kept, KP * nx, probs, 3, a->device, q.data())) return 1;
  std::vector<double> up((size_t)KP * nx), md((size_t)KP * nx), lo((size_t)KP * nx);
  for (size_t c = 0; c < (size_t)KP * nx; ++c) { lo[c] = q[3 * c]; md[c] = q[3 * c + 1]; up[c] = q[3 * c + 2]; }
  bfmmm_result* r = bfmmm_result_create();
  if (a->X) { put_mat(r, "CI_Upper", up, K, P, nx); put_mat(r, "CI_50", md, K, P, nx); put_mat(r, "CI_Lower", lo, K, P, nx); }
  else { put_mat(r, "CI_Upper", up, K, P); put_mat(r, "CI_50", md, K, P); put_mat(r, "CI_Lower", lo, K, P); }
  put_mat(r, "mean_trace", trace, K, P, (int64_t)kept * nx);
  *out = r;
  return 0;
}

extern "C" int bfmmm_HDFMeanCI(const bfmmm_ci_args* a, bfmmm_result** out) {
  if (!a || a->dim <= 0) return bfmmm_io_fail("bfmmm_HDFMeanCI: args.dim must be positive");
  return bfmmm_FMeanCI(a, out);
}
