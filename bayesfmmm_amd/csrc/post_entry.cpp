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
