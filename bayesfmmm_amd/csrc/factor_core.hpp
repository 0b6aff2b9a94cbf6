// Factorisation core shared by k_factor (nu / Phi directions) and k_cov_factor (eta / Xi directions).
//
// On entry S (PP x PP, column-major, stride PP) holds a symmetric positive definite precision
// matrix of half-bandwidth `bw` and X (PP x PP) is zero.  Computes
//     Prec = U U'   (U upper triangular; "reverse" Cholesky, processed from the last row up)
//     X    = U^-1   (row-major),   L = X' = chol_lower(Prec^-1),   C = L L' = Prec^-1,   Lz = L z
// which are exactly the covariance and the factor the reference's  arma::mvnrnd(C b, C)  uses
// (UpdateNu.h:67-69, UpdatePhi.h:79-82, UpdateEta.h:85-87, UpdateXi.h:80-83).
//
// The two triangular recursions are chains of P dependent pivots, so they run in the registers of
// ONE wave (lane i owns band row i of U, then column i of X; the pivot row travels by v_readlane)
// instead of through LDS; C = X'X is 16x16x4 fp64 MFMA tiles on all four waves.
// 256 threads.  Returns true -- on EVERY thread of the workgroup -- when a pivot fell to 1e-12 of the largest diagonal
// entry or below: the precision is singular to working accuracy, nothing has been written, and the caller takes the
// pseudo-inverse route (factor_pinv below), as the reference does through arma::pinv / mvnrnd's eigen fallback.
#pragma once
#include "model.hpp"

namespace bfmmm {

#ifdef BFMMM_TIMELINE
static __device__ unsigned long long g_fct[8];
#define FCT(i) do { if (blockIdx.x == 1 && threadIdx.x == 0) g_fct[i] = wall_clock64(); } while (0)      // (single-chain runs: workgroup 1 = direction 1)
#else
#define FCT(i) do { } while (0)
#endif

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

__device__ inline double readlane_f64(double v, int lane) {     // lane is wave-uniform
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// wave 0: reverse Cholesky, X = U^-1 (written to LDS), L z
// (always inlined: out of line the LDS pointers degrade to flat addresses and the recursions run 60 % slower)
template <int PP, int BWT>
__device__ __forceinline__ bool factor_wave(double* S, double* X, const double* zv, int P, double* Lz_out, int lane) {
  // (this wave carries the workgroup's chain of 2 P dependent steps while spare-job workgroups -- gamma rejection loops, lgamma
  //  series -- share its SIMD: it issues ahead of them)
  __builtin_amdgcn_s_setprio(3);
  // Root-free form of the recursion (round 4).  With v(i, k) = U(i, k) U(k, k) (the entries before their division by the pivot's
  // square root) the reverse Cholesky reads  d_k = Prec(k, k) - sum_m v(k, k + m)^2 / d_{k + m},
  // v(i, k) = Prec(i, k) - sum_m v(i, k + m) v(k, k + m) / d_{k + m}:  no square root on the chain of P dependent pivots, only
  // 1 / d_k (hardware estimate + two Newton steps = 4 dependent FMAs instead of rsq + 6), and every lane keeps its diagonal
  // residual d_i up to date incrementally (one FMA on the chain instead of BWT).  The square roots are taken afterwards, for all
  // rows at once: U(i, i) = sqrt(d_i), U(i, i + t) = v(i, i + t) / sqrt(d_{i + t}).
  double s[BWT + 1], u[BWT + 1];       // s[t] = Prec(i, i + t),  u[t] = U(i, i + t)
  double v[BWT + 1], w[BWT + 1];       // v[t] = v(i, i + t),  w[t] = v(i, i + t) / d_{i + t}
#pragma unroll
  for (int t = 0; t <= BWT; ++t) {
    const int j = lane + t;
    s[t] = (lane < P && j < P) ? S[lane + PP * j] : 0.0;
    u[t] = 0.0; v[t] = 0.0; w[t] = 0.0;
  }
  double rinv = 0.0;                   // 1 / U(i, i)
  bool bad = false;
  double dmax = s[0];                  // largest diagonal entry of Prec: the pivots are judged against it
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) dmax = fmax(dmax, __shfl_xor(dmax, o, 64));
  const double thr = 1e-12 * dmax;
  double dloc = s[0];                  // d_i so far: Prec(i, i) minus the terms of the columns already eliminated
  const int Pu = __builtin_amdgcn_readfirstlane(P);      // (a scalar loop counter: the pivot index feeds v_readlane directly)
  for (int k = Pu - 1; k >= 0; --k) {
    const double dk = readlane_f64(dloc, k);
    if (!(dk > thr)) bad = true;
    // everything that does not need 1 / d_k first: the rows' v(i, k) and which of them (if any) is this lane's
    double vk[BWT + 1];
#pragma unroll
    for (int m = 1; m < BWT; ++m) vk[m] = readlane_f64(v[m], k);      // v(k, k + m)
    double asel = 0.0;
#pragma unroll
    for (int t = BWT; t >= 1; --t) {   // row i = k - t:  v(i, k) = Prec(i, k) - sum_m [v(i, k + m) / d_{k + m}] v(k, k + m)
      double acc = s[t];
#pragma unroll
      for (int m = 1; m <= BWT - t; ++m) acc -= w[t + m] * vk[m];
      const bool mine = lane == k - t;
      v[t] = mine ? acc : v[t];
      asel = mine ? acc : asel;
    }
    // the chain of dependent pivots: d_k -> 1 / d_k (estimate + two Newton steps) -> w = v / d_k -> d_i -= v w -> d_{k - 1}
    double inv = __builtin_amdgcn_rcp(dk);
    inv = fma(fma(-dk, inv, 1.0), inv, inv);
    inv = fma(fma(-dk, inv, 1.0), inv, inv);
    const double wsel = asel * inv;
    dloc = fma(-asel, wsel, dloc);      // (asel = 0 on the lanes without a row of this column)
    __builtin_amdgcn_sched_barrier(0);  // (the selects below are independent of the chain: they fill the wait of the next v_readlane)
#pragma unroll
    for (int t = 1; t <= BWT; ++t) w[t] = (lane == k - t) ? wsel : w[t];
  }
  {
    // all rows at once: 1 / sqrt(d_i) (estimate + two Newton steps, as before), U(i, i + t) = v(i, i + t) / sqrt(d_{i + t})
    const double di = (lane < P) ? dloc : 1.0;
    double rk = __builtin_amdgcn_rsq(di);
    rk = rk * (1.5 - (0.5 * di) * (rk * rk));
    rk = rk * (1.5 - (0.5 * di) * (rk * rk));
    u[0] = di * rk; rinv = rk;
#pragma unroll
    for (int t = 1; t <= BWT; ++t) u[t] = v[t] * __shfl_down(rk, t, 64);      // (v is zero where i + t >= P)
  }
  FCT(0);
  // column `lane` of X = U^-1 by back substitution; xw[t] = X(i + t, lane).  Row i of U (its band, 1 / U(i,i)) and
  // z_i are needed by every lane: lane i parks them in LDS once (the band of Prec in S has been consumed) and the
  // loop reads them back with wave-uniform addresses, one row ahead, instead of 2 (BWT + 2) v_readlane per row.
  constexpr int BS = BWT + 2;          // doubles per parked row: 1/U(i,i), U(i,i+1..i+BWT), z_i
  double* bc = S;
  if (lane < P) {
    bc[lane * BS] = rinv;
#pragma unroll
    for (int t = 1; t <= BWT; ++t) bc[lane * BS + t] = u[t];
    bc[lane * BS + BWT + 1] = zv[lane];
  }
  __builtin_amdgcn_wave_barrier();
  double xw[BWT + 1];
#pragma unroll
  for (int t = 0; t <= BWT; ++t) xw[t] = 0.0;
  double lzacc = 0.0;
  double nxt[BS];
#pragma unroll
  for (int t = 0; t < BS; ++t) nxt[t] = bc[(P - 1) * BS + t];
  for (int i = P - 1; i >= 0; --i) {
    double cur[BS];
#pragma unroll
    for (int t = 0; t < BS; ++t) cur[t] = nxt[t];
    const int ip = max(i - 1, 0);
#pragma unroll
    for (int t = 0; t < BS; ++t) nxt[t] = bc[ip * BS + t];
    double acc = 0.0;
#pragma unroll
    for (int t = BWT; t >= 1; --t) acc += cur[t] * xw[t];       // (xw[1], the previous row's result, enters last: one FMA on the chain)
    const double xi = (lane == i) ? cur[0] : ((lane > i) ? -(acc * cur[0]) : 0.0);
    if (lane < PP) X[i * PP + lane] = xi;
    lzacc += xi * cur[BWT + 1];
#pragma unroll
    for (int t = BWT; t >= 2; --t) xw[t] = xw[t - 1];
    if constexpr (BWT >= 1) xw[1] = xi;
  }
  FCT(1);
  if (lane < P) Lz_out[lane] = lzacc;
  __builtin_amdgcn_s_setprio(0);
  return bad;
}

// Cl (optional): an LDS copy of C, Cl[q + PP * p] = C(p, q) -- may alias S, whose band has been consumed by then
template <int PP>
__device__ __forceinline__ bool factor_core(double* S, double* X, const double* zv, int P, int bw, double* Cg, double* Lg,
                                   double* Lz_out, int tid, double* Cl = nullptr) {
  bool bad = false;
  if (bw == 0) {
    // diagonal precision (multivariate model): C = diag(1 / s_pp), chol_lower(C) = diag(1 / sqrt(s_pp)), L z likewise --
    // no recursion at all.  (1 / sqrt by the same estimate + two Newton steps as the general path, so the factor is the
    // same function of the pivot.)
    for (int e = tid; e < P * P; e += 256) {
      const int p = e % P, q = e / P;
      double cv = 0.0, lv = 0.0;
      if (p == q) {
        const double dk = S[p + PP * p];
        double rk = __builtin_amdgcn_rsq(dk);
        rk = rk * (1.5 - (0.5 * dk) * (rk * rk));
        rk = rk * (1.5 - (0.5 * dk) * (rk * rk));
        if (!(dk > 0.0)) bad = true;
        lv = rk;
        cv = rk * rk;
        Lz_out[p] = rk * zv[p];
      }
      Cg[q + (size_t)P * p] = cv;
      if (Cl) Cl[q + PP * p] = cv;
      if (Lg) Lg[p + (size_t)P * q] = lv;
    }
    return __syncthreads_or(bad ? 1 : 0) != 0;       // every thread reports (the callers test thread 0)
  }
  if (bw > 5) {
    // wide band (user-supplied / tensor-product bases): plain dense reverse Cholesky in LDS.  S must be fully defined.
    // Prec = U U', U upper, columns from the last to the first: U(k,k) = sqrt(A(k,k)), U(i,k) = A(i,k) / U(k,k) (i < k),
    // then A(i,j) -= U(i,k) U(j,k) for i, j < k (right-looking).  U overwrites the upper triangle of S (S[i + PP*k]).
    double dmaxw = 0.0;
    for (int i = 0; i < P; ++i) dmaxw = fmax(dmaxw, S[i + PP * i]);
    const double thrw = 1e-12 * dmaxw;
    for (int k = P - 1; k >= 0; --k) {
      __syncthreads();                               // the previous column's update of the leading block is complete
      const double dk = S[k + PP * k];
      if (!(dk > thrw)) bad = true;
      const double rk = 1.0 / sqrt(dk);
      __syncthreads();                               // everyone has the pivot before column k is overwritten
      if (tid <= k) S[tid + PP * k] = (tid == k) ? dk * rk : S[tid + PP * k] * rk;
      __syncthreads();
      for (int e = tid; e < k * k; e += 256) {
        const int i = e % k, j = e / k;
        S[i + PP * j] -= S[i + PP * k] * S[j + PP * k];
      }
    }
    __syncthreads();
    // X = U^-1 (upper, row-major X[i*PP + c]): row i from the last up, column c >= i, 4 lanes per column
    for (int e = tid; e < PP * PP; e += 256) X[e] = 0.0;
    __syncthreads();
    const int cc = tid >> 2, ql = tid & 3;
    for (int i = P - 1; i >= 0; --i) {
      double acc = 0.0;
      if (cc < P && cc > i)
        for (int m = i + 1 + ql; m <= cc; m += 4) acc += S[i + PP * m] * X[m * PP + cc];
      acc = dpp_add<0xB1>(acc);
      acc = dpp_add<0x4E>(acc);
      __syncthreads();
      if (ql == 0 && cc < P && cc >= i) X[i * PP + cc] = ((cc == i) ? 1.0 - acc : -acc) / S[i + PP * i];
      __syncthreads();
    }
    if (tid < P) {
      double lz = 0.0;
      for (int i = 0; i <= tid; ++i) lz += X[i * PP + tid] * zv[i];
      Lz_out[tid] = lz;
    }
    bad = __syncthreads_or(bad ? 1 : 0) != 0;      // (also publishes X and Lz to every thread)
    if (bad) return true;
  } else if (tid < 64) {
    switch (bw) {
      case 0: bad = factor_wave<PP, 0>(S, X, zv, P, Lz_out, tid); break;
      case 1: bad = factor_wave<PP, 1>(S, X, zv, P, Lz_out, tid); break;
      case 2: bad = factor_wave<PP, 2>(S, X, zv, P, Lz_out, tid); break;
      case 3: bad = factor_wave<PP, 3>(S, X, zv, P, Lz_out, tid); break;
      case 4: bad = factor_wave<PP, 4>(S, X, zv, P, Lz_out, tid); break;
      default: bad = factor_wave<PP, 5>(S, X, zv, P, Lz_out, tid); break;
    }
  }
  if (bw <= 5) {
    // wave 0 ran the factorisation: its verdict reaches every thread through the last element of S (the band of Prec
    // was consumed at the start of factor_wave and its parking area ends far below)
    if (tid == 0) S[PP * PP - 1] = bad ? 1.0 : 0.0;
    __syncthreads();
    if (S[PP * PP - 1] != 0.0) return true;
  }
  FCT(2);
  // C = X' X on the matrix cores: tile (pt, qt) of 16 x 16, K = P rounded up to 4 (rows k > min(p, q) of X are zero)
  {
    constexpr int NTL = PP / 16;
    const int wave = tid >> 6, lane = tid & 63;
    const int kend = (P + 3) & ~3;
    for (int tt = wave; tt < NTL * NTL; tt += 4) {
      const int pt = tt / NTL, qt = tt - pt * NTL;
      double4_t acc = {0.0, 0.0, 0.0, 0.0};
      const double* xa = X + (lane >> 4) * PP + pt * 16 + (lane & 15);
      const double* xb = X + (lane >> 4) * PP + qt * 16 + (lane & 15);
      for (int k0 = 0; k0 < kend; k0 += 4)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[k0 * PP], xb[k0 * PP], acc, 0, 0, 0);
      const int q = qt * 16 + (lane & 15);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int p = pt * 16 + (lane >> 4) + 4 * r;
        if (p < P && q < P) Cg[q + (size_t)P * p] = acc[r];     // C is symmetric: the transposed store is coalesced
        if (Cl) Cl[q + PP * p] = acc[r];
      }
    }
  }
  FCT(3);
  if (Lg)
    for (int e = tid; e < PP * PP; e += 256) {
      const int p = e & (PP - 1), q = e / PP;
      if (p < P && q < P) Lg[p + (size_t)P * q] = (q <= p) ? X[q * PP + p] : 0.0;
    }
  return bad;
}

// ---------------------------------------------------------------------------------------------------------------------
// Pseudo-inverse route for a precision that is singular to working accuracy (factor_core returned true).  The reference
// takes arma::pinv for nu and eta (UpdateNu.h:67-68, UpdateEta.h:85-86): a cluster without members leaves Prec = tau P_mat
// of rank P - 1, pinv drops the null direction and arma::mvnrnd -- its Cholesky factorisation of the singular covariance
// failing -- draws through the symmetric eigen-decomposition.  Same law here, by a fixed specification (the tests' CPU
// restatement implements the same one): Prec = V diag(w) V' by Jacobi rotations, eigenpairs by ascending w, eigenvectors signed by
// the generic-weights rule, winv_k = 1 / w_k above Armadillo's tolerance P max|w| eps and 0 below,
//     C = V diag(winv) V',     L = V diag(sqrt(winv)),     L z with z in index order.
// S: the full symmetric PP x PP precision (zero beyond P), X: PP x PP scratch (V), wk: 4 PP + 2 doubles of scratch.
// A rare path: parallel cyclic Jacobi, PP / 2 disjoint rotations per round (round-robin pairing), three barriers a round.
// ---------------------------------------------------------------------------------------------------------------------
template <int PP>
__device__ __noinline__ void factor_pinv(double* S, double* V, const double* zv, int P, double* Cg, double* Lg, double* Lz_out,
                                         int tid, double* wk) {
  constexpr int NPR = PP / 2;
  double* cs = wk;                    // NPR
  double* sn = wk + NPR;              // NPR
  double* w = wk + PP;                // PP   eigenvalues, then winv
  int* pp = (int*)(wk + 2 * PP);      // NPR
  int* qq = pp + NPR;                 // NPR
  int* ord = qq + NPR;                // PP   ord[rank] = eigenpair
  int* flag = ord + PP;               // 1    a rotation was applied in this sweep
  double* sg = wk + 3 * PP + 1;       // PP   sign of every eigenvector
  for (int e = tid; e < PP * PP; e += 256) V[e] = ((e % PP) == (e / PP)) ? 1.0 : 0.0;
  __syncthreads();
  for (int sweep = 0; sweep < 30; ++sweep) {
    if (tid == 0) *flag = 0;
    __syncthreads();
    for (int r = 0; r < PP - 1; ++r) {
      if (tid < NPR) {
        const int p = (r + tid) % (PP - 1);
        const int q = (tid == 0) ? PP - 1 : (r - tid + (PP - 1)) % (PP - 1);
        const double apq = S[p + PP * q], app = S[p + PP * p], aqq = S[q + PP * q];
        double c = 1.0, sv = 0.0;
        if (fabs(apq) > 1e-20 * sqrt(fabs(app * aqq)) && apq != 0.0) {
          const double theta = (aqq - app) / (2.0 * apq);
          const double t = ((theta >= 0) ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
          c = 1.0 / sqrt(t * t + 1.0);
          sv = t * c;
          *flag = 1;
        }
        cs[tid] = c; sn[tid] = sv; pp[tid] = p; qq[tid] = q;
      }
      __syncthreads();
      for (int e = tid; e < PP * NPR; e += 256) {          // columns p, q of S and V
        const int k = e % PP, i = e / PP, p = pp[i], q = qq[i];
        const double c = cs[i], sv = sn[i];
        const double akp = S[k + PP * p], akq = S[k + PP * q];
        S[k + PP * p] = c * akp - sv * akq;
        S[k + PP * q] = sv * akp + c * akq;
        const double vkp = V[k + PP * p], vkq = V[k + PP * q];
        V[k + PP * p] = c * vkp - sv * vkq;
        V[k + PP * q] = sv * vkp + c * vkq;
      }
      __syncthreads();
      for (int e = tid; e < PP * NPR; e += 256) {          // rows p, q of S
        const int k = e % PP, i = e / PP, p = pp[i], q = qq[i];
        const double c = cs[i], sv = sn[i];
        const double apk = S[p + PP * k], aqk = S[q + PP * k];
        S[p + PP * k] = c * apk - sv * aqk;
        S[q + PP * k] = sv * apk + c * aqk;
      }
      __syncthreads();
    }
    if (*flag == 0) break;                                 // (uniform: read after the round's last barrier)
    __syncthreads();
  }
  // the real eigenpairs are columns 0 .. P-1 (the zero padding never mixes with them)
  if (tid < PP) w[tid] = (tid < P) ? S[tid + PP * tid] : 0.0;
  __syncthreads();
  if (tid < P) {
    int rank = 0;
    double wmax = 0.0;
    for (int j = 0; j < P; ++j) {
      if (w[j] < w[tid] || (w[j] == w[tid] && j < tid)) ++rank;
      wmax = fmax(wmax, fabs(w[j]));
    }
    ord[rank] = tid;
    double gs = 0.0;                  // sign rule of the specification: sum_i g_i v_i > 0, g_i = 1 / (i + 1.37)
    for (int i = 0; i < P; ++i) {
      gs += V[i + PP * tid] / ((double)i + 1.37);
    }
    sg[tid] = (gs < 0) ? -1.0 : 1.0;
    const double tol = (double)P * wmax * 2.220446049250313e-16;
    wk[tid] = (fabs(w[tid]) > tol) ? 1.0 / w[tid] : 0.0;   // winv takes over the (now free) rotation table wk[0 .. PP)
  }
  __syncthreads();
  const double* winv = wk;                                 // PP entries (cs and sn areas)
  for (int e = tid; e < P * P; e += 256) {
    const int i = e % P, j = e / P;
    double acc = 0.0;
    for (int k = 0; k < P; ++k) acc += V[i + PP * k] * winv[k] * V[j + PP * k];
    Cg[j + (size_t)P * i] = acc;
    if (Lg) {                                              // column j of the factor = eigenpair ord[j]
      const int k = ord[j];
      Lg[i + (size_t)P * j] = sg[k] * V[i + PP * k] * sqrt(fmax(winv[k], 0.0));
    }
  }
  if (tid < P) {
    double lz = 0.0;
    for (int r = 0; r < P; ++r) {
      const int k = ord[r];
      lz += sg[k] * V[tid + PP * k] * sqrt(fmax(winv[k], 0.0)) * zv[r];
    }
    Lz_out[tid] = lz;
  }
  __syncthreads();
}

}  // namespace bfmmm
