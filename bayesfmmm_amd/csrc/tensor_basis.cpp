// Tensor-product B-spline basis and its random-walk penalty for the high-dimensional functional model (BHDFMMM_*):
// host-side set-up code (it runs once per call; the per-curve statistics G_i, s_i, yy_i are built from its rows).
//
//   bfmmm_tensor_bspline  <-  TensorBSpline  inst/include/BayesFMMM/BSplines.h:18-66
//                             (splines2::BSpline(t.col(l), internal_knots(l), degree(l), boundary_knots.row(l)).basis(true)
//                              per dimension; column i of the tensor basis is the product over the dimensions of the
//                              univariate columns idx_l(i), the LAST dimension running fastest: BSplines.h:25-31, 56-60)
//   bfmmm_tensor_penalty  <-  GetP           BSplines.h:74-120: P = C'C with one constraint row e_i - e_j for every pair of
//                             basis functions whose multi-indices differ by +1 in exactly one dimension
//
// Pinned by the reference's own golden files (inst/test-data/Tensor_BSpline.txt, P_mat.txt; src/test-BSplines.cpp:9-52):
// tests/test_tensor_basis.py.  The sampler's wide-band per-curve kernels for this model are not built yet.
#include "../../include/bfmmm_entry.h"

#include <cmath>
#include <string>
#include <vector>

int bfmmm_io_fail(const std::string& m);      // entry_points.cpp: sets bfmmm_entry_last_error

namespace {

// clamped B-spline basis functions of `degree` at x (Cox-de Boor, the span's degree + 1 non-zero functions), complete
// basis with intercept, right boundary inclusive -- what splines2::BSpline(...).basis(true) returns for one point
void basis_row(double x, int degree, const std::vector<double>& knots, int P, double* out /* P */) {
  for (int p = 0; p < P; ++p) out[p] = 0.0;
  const int nk = (int)knots.size();
  const double b0 = knots[degree], b1 = knots[nk - 1 - degree];
  if (x < b0 || x > b1) return;                      // outside the boundary knots: all zero (splines2 behaviour)
  int span = degree;
  if (x >= b1) span = P - 1;                         // right boundary inclusive: the last function is 1 there
  else while (span + 1 < nk - 1 - degree && x >= knots[span + 1]) ++span;
  std::vector<double> N((size_t)degree + 1, 0.0), left((size_t)degree + 1), right((size_t)degree + 1);
  N[0] = 1.0;
  for (int j = 1; j <= degree; ++j) {
    left[j] = x - knots[span + 1 - j];
    right[j] = knots[span + j] - x;
    double saved = 0.0;
    for (int r = 0; r < j; ++r) {
      const double den = right[r + 1] + left[j - r];
      const double tmp = (den != 0.0) ? N[r] / den : 0.0;
      N[r] = saved + right[r + 1] * tmp;
      saved = left[j - r] * tmp;
    }
    N[j] = saved;
  }
  for (int j = 0; j <= degree; ++j) out[span - degree + j] = N[j];
}

}  // namespace

// t: n_pts x dim column-major; boundary_knots: dim x 2 row-major (lower, upper per dimension); internal_knots: the
// dimensions' internal knots one after the other (n_internal[l] each); out: n_pts x P column-major, P = prod_l
// (n_internal[l] + degree[l] + 1)
extern "C" int bfmmm_tensor_bspline(int n_pts, int dim, const double* t, const int* degree, const double* boundary_knots,
                                    const int* n_internal, const double* internal_knots, double* out) {
  if (n_pts < 0 || dim < 1 || !t || !degree || !boundary_knots || !n_internal || !out)
    return bfmmm_io_fail("bfmmm_tensor_bspline: bad arguments");
  std::vector<int> Pl((size_t)dim), stride((size_t)dim, 1);
  std::vector<std::vector<double>> knots((size_t)dim);
  size_t ko = 0;
  long long P = 1;
  for (int l = 0; l < dim; ++l) {
    if (degree[l] < 1 || n_internal[l] < 0) return bfmmm_io_fail("bfmmm_tensor_bspline: bad degree / knots");
    Pl[l] = n_internal[l] + degree[l] + 1;
    P *= Pl[l];
    const double b0 = boundary_knots[2 * l], b1 = boundary_knots[2 * l + 1];
    for (int q = 0; q <= degree[l]; ++q) knots[l].push_back(b0);
    for (int q = 0; q < n_internal[l]; ++q) knots[l].push_back(internal_knots[ko + q]);
    for (int q = 0; q <= degree[l]; ++q) knots[l].push_back(b1);
    ko += (size_t)n_internal[l];
  }
  for (int l = dim - 2; l >= 0; --l) stride[l] = stride[l + 1] * Pl[l + 1];       // dim_counter, BSplines.h:29-31
  std::vector<std::vector<double>> rows((size_t)dim);
  for (int l = 0; l < dim; ++l) rows[l].resize((size_t)Pl[l]);
  for (int k = 0; k < n_pts; ++k) {
    for (int l = 0; l < dim; ++l) basis_row(t[k + (size_t)n_pts * l], degree[l], knots[l], Pl[l], rows[l].data());
    for (long long i = 0; i < P; ++i) {
      double v = 1.0;
      for (int l = 0; l < dim; ++l) v = v * rows[l][(size_t)((i / stride[l]) % Pl[l])];      // B(k, i) *= B_l(k, counter(l))
      out[k + (size_t)n_pts * i] = v;
    }
  }
  return 0;
}

// out: P x P column-major
extern "C" int bfmmm_tensor_penalty(int dim, const int* degree, const int* n_internal, double* out) {
  if (dim < 1 || !degree || !n_internal || !out) return bfmmm_io_fail("bfmmm_tensor_penalty: bad arguments");
  std::vector<int> Pl((size_t)dim), stride((size_t)dim, 1);
  long long P = 1;
  for (int l = 0; l < dim; ++l) { Pl[l] = n_internal[l] + degree[l] + 1; P *= Pl[l]; }
  for (int l = dim - 2; l >= 0; --l) stride[l] = stride[l + 1] * Pl[l + 1];
  for (long long e = 0; e < P * P; ++e) out[e] = 0.0;
  // every pair (i, j = i + one step in one dimension) contributes (e_i - e_j)(e_i - e_j)'  (BSplines.h:100-117)
  for (long long i = 0; i < P; ++i)
    for (int l = 0; l < dim; ++l)
      if ((i / stride[l]) % Pl[l] + 1 < Pl[l]) {
        const long long j = i + stride[l];
        out[i + P * i] += 1.0; out[j + P * j] += 1.0;
        out[i + P * j] -= 1.0; out[j + P * i] -= 1.0;
      }
  return 0;
}
