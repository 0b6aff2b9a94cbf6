// Factorisation core shared by k_factor (nu / Phi directions) and k_cov_factor (eta / Xi directions).
//
// On entry S (PP x PP, column-major, stride PP) holds a symmetric positive definite precision
// matrix of half-bandwidth `bw` and X (PP x PP) is zero.  Computes
//     Prec = U U'   (U upper triangular; "reverse" Cholesky, processed from the last row up)
//     X    = U^-1   (row-major),   L = X' = chol_lower(Prec^-1),   C = L L' = Prec^-1,   Lz = L z
// which are exactly the covariance and the factor the reference's  arma::mvnrnd(C b, C)  uses
// (UpdateNu.h:67-69, UpdatePhi.h:79-82, UpdateEta.h:85-87, UpdateXi.h:80-83).
// 256 threads; returns true on the calling thread if a non-positive pivot was met.
#pragma once
#include "model.hpp"

namespace bfmmm {

template <int PP>
__device__ inline bool factor_core(double* S, double* X, const double* zv, int P, int bw, double* Cg, double* Lg,
                                   double* Lz_out, int tid) {
  bool bad = false;
  if (tid < 64) {
    // reverse Cholesky Prec = U U', one wave, LDS traffic is wave-ordered
    for (int k = P - 1; k >= 0; --k) {
      const int jhi = min(k + bw, P - 1);
      double dkk = S[k + PP * k];
      for (int jj = k + 1; jj <= jhi; ++jj) { const double u = S[k + PP * jj]; dkk -= u * u; }
      if (!(dkk > 0.0)) bad = true;
      const double ukk = sqrt(dkk);
      const int i = k - 1 - tid;
      double uik = 0.0;
      if (tid < bw && i >= 0) {
        double acc = S[i + PP * k];
        const int j2 = min(i + bw, P - 1);
        for (int jj = k + 1; jj <= j2; ++jj) acc -= S[i + PP * jj] * S[k + PP * jj];
        uik = acc / ukk;
      }
      __builtin_amdgcn_wave_barrier();
      if (tid == 0) S[k + PP * k] = ukk;
      if (tid < bw && i >= 0) S[i + PP * k] = uik;
      __builtin_amdgcn_wave_barrier();
    }
  }
  __syncthreads();
  if (tid < P) {
    // column c of X = U^-1 by back substitution (banded U)
    const int cc = tid;
    X[cc * PP + cc] = 1.0 / S[cc + PP * cc];
    for (int i = cc - 1; i >= 0; --i) {
      double acc = 0.0;
      const int j2 = min(i + bw, cc);
      for (int jj = i + 1; jj <= j2; ++jj) acc += S[i + PP * jj] * X[jj * PP + cc];
      X[i * PP + cc] = -acc / S[i + PP * i];
    }
  }
  __syncthreads();
  // L = X' (lower), C = L L', L z
  for (int e = tid; e < PP * PP; e += 256) {
    const int p = e & (PP - 1), q = e / PP;
    if (p < P && q < P) {
      const int kmax = min(p, q);
      double acc = 0.0;
      for (int k = 0; k <= kmax; ++k) acc += X[k * PP + p] * X[k * PP + q];
      Cg[p + (size_t)P * q] = acc;
      if (Lg) Lg[p + (size_t)P * q] = (q <= p) ? X[q * PP + p] : 0.0;
    }
  }
  if (tid < P) {
    double acc = 0.0;
    for (int q = 0; q <= tid; ++q) acc += X[q * PP + tid] * zv[q];
    Lz_out[tid] = acc;
  }
  return bad;
}

}  // namespace bfmmm
