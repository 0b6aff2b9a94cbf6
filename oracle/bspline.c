/*
 * oracle/bspline.c -- TEST INFRASTRUCTURE (see oracle.h).
 *
 * B-spline design matrices and random-walk penalties.
 *
 * The reference calls the third-party header library splines2 (>= 0.3.0, DESCRIPTION:13;
 * not vendored under /root/reference):
 *   splines2::BSpline(t, internal_knots, degree, boundary_knots).basis(true)
 * at BFMMM.h:1017-1025, 1188-1196, 1392-1400 and UserFunctions.cpp:289-299.  Its published
 * algorithm is the Cox-de Boor recursion on the clamped knot vector
 *   [b0 x (degree+1), internal knots, b1 x (degree+1)]
 * with the right boundary included in the last interval; `basis(true)` keeps the intercept
 * column, so P = n_internal + degree + 1 (BFMMM.h:1015).
 * PINNED by the reference's golden file inst/test-data/Tensor_BSpline.txt through
 * orc_tensor_bspline (src/test-BSplines.cpp:9-28,66; tolerance absdiff 1e-7).
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

int orc_bspline_basis(int n, const double* x, int n_internal, const double* internal_knots,
                      int degree, const double* boundary_knots, double* out) {
  const int order = degree + 1;
  const int P = n_internal + order;
  const int nk = n_internal + 2 * order;
  double* knots = (double*)malloc(sizeof(double) * (size_t)nk);
  double* N = (double*)malloc(sizeof(double) * (size_t)(order + 1));
  for (int i = 0; i < order; ++i) knots[i] = boundary_knots[0];
  for (int i = 0; i < n_internal; ++i) knots[order + i] = internal_knots[i];
  for (int i = 0; i < order; ++i) knots[order + n_internal + i] = boundary_knots[1];
  int rc = 0;
  for (int r = 0; r < n; ++r) {
    double* row = out + (size_t)r * P;
    memset(row, 0, sizeof(double) * (size_t)P);
    const double xv = x[r];
    if (!(xv >= boundary_knots[0] && xv <= boundary_knots[1])) { rc = 1; continue; }
    /* knot span: largest j in [degree, P-1] with knots[j] <= x (x == b1 -> last interval) */
    int j = degree;
    while (j < P - 1 && knots[j + 1] <= xv) ++j;
    /* de Boor's triangular scheme: non-zero basis functions N_{j-degree..j} */
    N[0] = 1.0;
    for (int d = 1; d <= degree; ++d) {
      double saved = 0.0;
      for (int q = 0; q < d; ++q) {
        const double right = knots[j + q + 1] - xv;
        const double left = xv - knots[j + 1 - d + q];
        const double denom = right + left;
        const double temp = (denom != 0.0) ? N[q] / denom : 0.0;
        N[q] = saved + right * temp;
        saved = left * temp;
      }
      N[d] = saved;
    }
    for (int q = 0; q <= degree; ++q) row[j - degree + q] = N[q];
  }
  free(knots);
  free(N);
  return rc;
}

/* RW1 penalty, BFMMM.h:1027-1037 (also :1198-1208, :1402-1412) */
void orc_pmat_rw1(int P, double* Pm) {
  memset(Pm, 0, sizeof(double) * (size_t)P * P);
  for (int j = 0; j < P; ++j) {
    Pm[0] = 1.0;
    if (j > 0) {
      Pm[j + (size_t)P * j] = 2.0;
      Pm[(j - 1) + (size_t)P * j] = -1.0;
      Pm[j + (size_t)P * (j - 1)] = -1.0;
    }
    Pm[(P - 1) + (size_t)P * (P - 1)] = 1.0;
  }
}

/* BSplines.h:18-62 TensorBSpline for ONE curve: t is n_pts x dim column-major;
 * column i of the output is the product over dimensions of the univariate basis columns
 * selected by the mixed-radix digits of i (last dimension fastest, BSplines.h:25-33,54-58). */
int orc_tensor_bspline(int n_pts, int dim, const double* t, const int* degree,
                       const double* boundary, const int* n_internal,
                       const double* const* internal_knots, double* out) {
  int P = 1, rc = 0;
  int* Pl = (int*)malloc(sizeof(int) * (size_t)dim);
  double** Bl = (double**)malloc(sizeof(double*) * (size_t)dim);
  for (int l = 0; l < dim; ++l) {
    Pl[l] = n_internal[l] + degree[l] + 1;
    P *= Pl[l];
    Bl[l] = (double*)malloc(sizeof(double) * (size_t)n_pts * Pl[l]);
    rc |= orc_bspline_basis(n_pts, t + (size_t)l * n_pts, n_internal[l], internal_knots[l],
                            degree[l], boundary + 2 * l, Bl[l]);
  }
  for (int i = 0; i < P; ++i) {
    int rem = i;
    for (int k = 0; k < n_pts; ++k) out[k + (size_t)n_pts * i] = 1.0;
    for (int l = dim - 1; l >= 0; --l) {
      const int digit = rem % Pl[l];
      rem /= Pl[l];
      for (int k = 0; k < n_pts; ++k) out[k + (size_t)n_pts * i] *= Bl[l][(size_t)k * Pl[l] + digit];
    }
  }
  for (int l = 0; l < dim; ++l) free(Bl[l]);
  free(Bl);
  free(Pl);
  return rc;
}

/* BSplines.h:70-120 GetP: Constraint' * Constraint where each constraint row is the first
 * difference between two tensor indices that differ by +1 in exactly one dimension. */
int orc_get_P(int dim, const int* degree, const int* n_internal, double* out) {
  int P = 1;
  int* Pl = (int*)malloc(sizeof(int) * (size_t)dim);
  for (int l = 0; l < dim; ++l) { Pl[l] = n_internal[l] + degree[l] + 1; P *= Pl[l]; }
  int* index = (int*)malloc(sizeof(int) * (size_t)P * dim);
  for (int i = 0; i < P; ++i) {
    int rem = i;
    for (int l = dim - 1; l >= 0; --l) { index[(size_t)i * dim + l] = rem % Pl[l]; rem /= Pl[l]; }
  }
  memset(out, 0, sizeof(double) * (size_t)P * P);
  for (int i = 0; i < P; ++i)
    for (int j = i; j < P; ++j) {
      int diff = 0, abs_diff = 0;
      for (int l = 0; l < dim; ++l) {
        int dl = index[(size_t)j * dim + l] - index[(size_t)i * dim + l];
        diff += dl;
        abs_diff += abs(dl);
      }
      if (diff == 1 && abs_diff == 1) {
        out[i + (size_t)P * i] += 1.0;
        out[j + (size_t)P * j] += 1.0;
        out[i + (size_t)P * j] -= 1.0;
        out[j + (size_t)P * i] -= 1.0;
      }
    }
  free(index);
  free(Pl);
  return P;
}
