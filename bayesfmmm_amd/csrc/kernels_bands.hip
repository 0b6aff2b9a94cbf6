// Credible bands over saved draws on the device (include/bfmmm_post.h): column quantiles and simultaneous bands of a
// draws x columns table, and the table of mean-function draws f[t][j] = B_j' coef_t itself.
//
//   k_bands_fsamp      f[t + T j] = sum_p B[j][p] coef[t][p]                    (grid: columns x chunks of draws)
//   k_bands_quantiles  one workgroup per column: the column's T draws are sorted in LDS (bitonic network, padded with
//                      +inf to a power of two) and the requested quantiles read off with Armadillo's rule
//                      (arma::quantile: Hyndman & Fan definition 5 -- p_k = (k - 0.5) / N, linear in between, the extremes
//                      outside [0.5 / N, (N - 0.5) / N])
//   k_bands_moments    per column mean and standard deviation (N - 1), fixed-order tree
//   k_bands_maxdev     per draw C_t = max_j |f - mean_j| / sd_j                  (simultaneous bands, PostProcessing.cpp:287-296)
// Columns of up to 8192 draws are sorted in LDS (64 KiB); longer ones (the reference has no limit: arma::quantile sorts any
// length) by the same network over a padded copy of the column in global memory, k_bands_quantiles_big: the exchanges whose
// partner is at least LCH elements away run as passes over global memory, the rest of every merge step chunk by chunk in LDS.
#include <hip/hip_runtime.h>

#include <cmath>
#include <string>
#include <vector>

#include "../../include/bfmmm_post.h"

int bfmmm_io_fail(const std::string& m);

namespace {

constexpr int QT = 256;
constexpr int TMAX = 8192;         // draws per column sorted entirely in LDS
constexpr int TBIG = 1 << 24;      // draws per column, global-memory path
constexpr int BT = 1024;           // threads of the global-memory path
constexpr int LCH = 8192;          // elements of its LDS chunk

__global__ __launch_bounds__(QT) void k_bands_fsamp(const double* B, const double* coef, int T, int P, double* f) {
  const int j = blockIdx.x;
  const double* b = B + (size_t)j * P;
  for (int t = blockIdx.y * QT + threadIdx.x; t < T; t += gridDim.y * QT) {
    const double* c = coef + (size_t)t * P;
    double s = 0.0;
    for (int p = 0; p < P; ++p) s += b[p] * c[p];
    f[(size_t)t + (size_t)T * j] = s;
  }
}

__device__ inline void sort_lds(double* s, int NP, int tid) {
  for (int k = 2; k <= NP; k <<= 1)
    for (int jj = k >> 1; jj > 0; jj >>= 1) {
      __syncthreads();
      for (int e = tid; e < NP; e += QT) {
        const int partner = e ^ jj;
        if (partner > e) {
          const bool up = (e & k) == 0;
          const double a = s[e], b = s[partner];
          if ((a > b) == up) { s[e] = b; s[partner] = a; }
        }
      }
    }
  __syncthreads();
}

// out[q + nq * col]
__global__ __launch_bounds__(QT) void k_bands_quantiles(const double* V, int T, const double* probs, int nq, double* out) {
  extern __shared__ double s[];
  const int col = blockIdx.x, tid = threadIdx.x;
  int NP = 1;
  while (NP < T) NP <<= 1;
  for (int e = tid; e < NP; e += QT) s[e] = (e < T) ? V[(size_t)e + (size_t)T * col] : INFINITY;
  sort_lds(s, NP, tid);
  if (tid < nq) {
    const double N = (double)T, p = probs[tid];
    double v;
    if (p < 0.5 / N) v = (p < 0.0) ? -INFINITY : s[0];
    else if (p > (N - 0.5) / N) v = (p > 1.0) ? INFINITY : s[T - 1];
    else {
      const int k = (int)floor(N * p + 0.5);
      const double pk = ((double)k - 0.5) / N, w = (p - pk) * N;
      v = (1.0 - w) * s[k - 1] + w * s[min(k, T - 1)];
    }
    out[tid + (size_t)nq * col] = v;
  }
}


// The same bitonic network for a column that does not fit LDS: W is a scratch copy of the column padded with +inf to NP (a
// power of two > TMAX), one workgroup per column.  For the exchange distance jj >= LCH the pairs (e, e ^ jj) are exchanged in
// global memory (agent-scope relaxed accesses: the workgroup re-reads what its other waves wrote, and a wave's vector L1 is not
// refreshed by other waves' stores); once jj < LCH the remaining exchanges of the merge step stay inside aligned chunks of
// LCH elements, which are loaded into LDS, finished there and stored back.
__global__ __launch_bounds__(BT) void k_bands_quantiles_big(const double* V, int T, int NP, double* Wall, const double* probs, int nq, double* out) {
  extern __shared__ double s[];
  const int col = blockIdx.x, tid = threadIdx.x;
  double* W = Wall + (size_t)NP * col;
  auto ld = [&](int e) { return __hip_atomic_load(W + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  auto st = [&](int e, double v) { __hip_atomic_store(W + e, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  for (int e = tid; e < NP; e += BT) st(e, (e < T) ? V[(size_t)e + (size_t)T * col] : INFINITY);
  __syncthreads();
  for (int k = 2; k <= NP; k <<= 1) {
    int jj = k >> 1;
    for (; jj >= LCH; jj >>= 1) {
      for (int e = tid; e < NP; e += BT) {
        const int partner = e ^ jj;
        if (partner > e) {
          const bool up = (e & k) == 0;
          const double a = ld(e), b = ld(partner);
          if ((a > b) == up) { st(e, b); st(partner, a); }
        }
      }
      __syncthreads();
    }
    // jj < LCH: the rest of this merge step, chunk by chunk in LDS (the direction of a pair depends on its GLOBAL index)
    for (int c0 = 0; c0 < NP; c0 += LCH) {
      for (int e = tid; e < LCH; e += BT) s[e] = ld(c0 + e);
      for (int j2 = jj; j2 > 0; j2 >>= 1) {
        __syncthreads();
        for (int e = tid; e < LCH; e += BT) {
          const int partner = e ^ j2;
          if (partner > e) {
            const bool up = ((c0 + e) & k) == 0;
            const double a = s[e], b = s[partner];
            if ((a > b) == up) { s[e] = b; s[partner] = a; }
          }
        }
      }
      __syncthreads();
      for (int e = tid; e < LCH; e += BT) st(c0 + e, s[e]);
      __syncthreads();
    }
  }
  if (tid < nq) {
    const double N = (double)T, p = probs[tid];
    double v;
    if (p < 0.5 / N) v = (p < 0.0) ? -INFINITY : ld(0);
    else if (p > (N - 0.5) / N) v = (p > 1.0) ? INFINITY : ld(T - 1);
    else {
      const int k = (int)floor(N * p + 0.5);
      const double pk = ((double)k - 0.5) / N, w = (p - pk) * N;
      v = (1.0 - w) * ld(k - 1) + w * ld(min(k, T - 1));
    }
    out[tid + (size_t)nq * col] = v;
  }
}

// fixed-order pairwise sums of a column: mean, then sd with N - 1
__global__ __launch_bounds__(QT) void k_bands_moments(const double* V, int T, double* mean, double* sd) {
  __shared__ double red[QT];
  const int col = blockIdx.x, tid = threadIdx.x;
  const double* v = V + (size_t)T * col;
  double a = 0.0;
  for (int e = tid; e < T; e += QT) a += v[e];
  red[tid] = a;
  __syncthreads();
  for (int h = QT / 2; h > 0; h >>= 1) { if (tid < h) red[tid] += red[tid + h]; __syncthreads(); }
  const double m = red[0] / (double)T;
  __syncthreads();
  double q = 0.0;
  for (int e = tid; e < T; e += QT) { const double dlt = v[e] - m; q += dlt * dlt; }
  red[tid] = q;
  __syncthreads();
  for (int h = QT / 2; h > 0; h >>= 1) { if (tid < h) red[tid] += red[tid + h]; __syncthreads(); }
  if (tid == 0) { mean[col] = m; sd[col] = sqrt(red[0] / (double)(T - 1)); }
}

__global__ __launch_bounds__(QT) void k_bands_maxdev(const double* V, int T, int ncol, const double* mean, const double* sd, double* C) {
  const int t = blockIdx.x * QT + threadIdx.x;
  if (t >= T) return;
  double mx = -INFINITY;
  for (int j = 0; j < ncol; ++j) mx = fmax(mx, fabs((V[(size_t)t + (size_t)T * j] - mean[j]) / sd[j]));
  C[t] = mx;
}

struct Bufs {
  std::vector<void*> p;
  ~Bufs() { for (void* q : p) (void)hipFree(q); }
  bool put(double** out, const double* host, size_t count) {
    void* d = nullptr;
    if (hipMalloc(&d, std::max<size_t>(count, 1) * sizeof(double)) != hipSuccess) return false;
    p.push_back(d);
    if (host && count && hipMemcpy(d, host, count * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return false;
    *out = (double*)d;
    return true;
  }
};

int select_device(int device, const char* who) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return bfmmm_io_fail(std::string(who) + ": no HIP device (the MI355X library has no CPU path)");
  if (hipSetDevice(device) != hipSuccess) return bfmmm_io_fail(std::string(who) + ": cannot select the device");
  return 0;
}

size_t sort_lds_bytes(int T) { int NP = 1; while (NP < T) NP <<= 1; return (size_t)NP * sizeof(double); }


// quantiles of the columns of the device table dV (T x ncol): LDS sort, or the global-memory network for long columns
static int launch_quantiles(Bufs& b, const double* dV, int T, int ncol, const double* dprobs, int nq, double* dout) {
  if (T <= TMAX) {
    (void)hipFuncSetAttribute((const void*)k_bands_quantiles, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sort_lds_bytes(TMAX));
    hipLaunchKernelGGL(k_bands_quantiles, dim3(ncol), dim3(QT), sort_lds_bytes(T), 0, dV, T, dprobs, nq, dout);
    return 0;
  }
  int NP = 1;
  while (NP < T) NP <<= 1;
  double* dW;
  if (!b.put(&dW, nullptr, (size_t)NP * ncol)) return 1;
  (void)hipFuncSetAttribute((const void*)k_bands_quantiles_big, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LCH * sizeof(double)));
  hipLaunchKernelGGL(k_bands_quantiles_big, dim3(ncol), dim3(BT), LCH * sizeof(double), 0, dV, T, NP, dW, dprobs, nq, dout);
  return 0;
}

}  // namespace

extern "C" int bfmmm_post_col_quantiles(const double* V, int32_t T, int32_t ncol, const double* probs, int32_t nq, int32_t device, double* out) {
  if (!V || !probs || !out || T < 1 || ncol < 1 || nq < 1 || nq > QT) return bfmmm_io_fail("bfmmm_post_col_quantiles: bad arguments");
  if (T > TBIG) return bfmmm_io_fail("bfmmm_post_col_quantiles: at most 2^24 draws per column");
  if (select_device(device, "bfmmm_post_col_quantiles")) return 1;
  Bufs b;
  double *dV, *dp, *dout;
  if (!b.put(&dV, V, (size_t)T * ncol) || !b.put(&dp, probs, (size_t)nq) || !b.put(&dout, nullptr, (size_t)nq * ncol))
    return bfmmm_io_fail("bfmmm_post_col_quantiles: device allocation or copy failed");
  if (launch_quantiles(b, dV, T, ncol, dp, nq, dout)) return bfmmm_io_fail("bfmmm_post_col_quantiles: device allocation failed");
  if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess ||
      hipMemcpy(out, dout, sizeof(double) * nq * ncol, hipMemcpyDeviceToHost) != hipSuccess)
    return bfmmm_io_fail("bfmmm_post_col_quantiles: kernel launch or copy back failed");
  return 0;
}

// V[t + T (s1 + n1 s2)] = sum_j A[(t M + j) + T M s1] * Bm[(t M + j) + T M s2]   (covariance surface of a draw, FCovCI)
__global__ __launch_bounds__(QT) void k_bands_outer(const double* A, const double* Bm, int T, int M, int n1, double* V) {
  const int cell = blockIdx.x, s1 = cell % n1, s2 = cell / n1;
  const size_t TM = (size_t)T * M;
  for (int t = blockIdx.y * QT + threadIdx.x; t < T; t += gridDim.y * QT) {
    double s = 0.0;
    for (int j = 0; j < M; ++j) s += A[(size_t)t * M + j + TM * s1] * Bm[(size_t)t * M + j + TM * s2];
    V[(size_t)t + (size_t)T * cell] = s;
  }
}

// pointwise (alpha / 2, 0.5, 1 - alpha / 2) or simultaneous bands of the columns of a device table df (T x ncol)
static int table_bands(Bufs& b, const double* df, int T, int ncol, double alpha, int simultaneous, double* upper, double* mid, double* lower,
                       const char* who) {
  double *dp, *dq, *dm, *ds, *dC;
  const double probs[3] = {alpha / 2, 0.5, 1 - alpha / 2}, p1[1] = {1 - alpha};
  if (!b.put(&dp, simultaneous ? p1 : probs, simultaneous ? 1 : 3) || !b.put(&dq, nullptr, (size_t)3 * ncol) ||
      !b.put(&dm, nullptr, (size_t)ncol) || !b.put(&ds, nullptr, (size_t)ncol) || !b.put(&dC, nullptr, (size_t)T))
    return bfmmm_io_fail(std::string(who) + ": device allocation or copy failed");
  std::vector<double> q((size_t)3 * ncol), m((size_t)ncol), s((size_t)ncol);
  if (!simultaneous) {
    if (launch_quantiles(b, df, T, ncol, dp, 3, dq)) return bfmmm_io_fail(std::string(who) + ": device allocation failed");
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(q.data(), dq, sizeof(double) * 3 * ncol, hipMemcpyDeviceToHost) != hipSuccess)
      return bfmmm_io_fail(std::string(who) + ": kernel launch or copy back failed");
    for (int j = 0; j < ncol; ++j) { lower[j] = q[(size_t)3 * j]; mid[j] = q[(size_t)3 * j + 1]; upper[j] = q[(size_t)3 * j + 2]; }
  } else {
    hipLaunchKernelGGL(k_bands_moments, dim3(ncol), dim3(QT), 0, 0, df, T, dm, ds);
    hipLaunchKernelGGL(k_bands_maxdev, dim3((T + QT - 1) / QT), dim3(QT), 0, 0, df, T, ncol, dm, ds, dC);
    if (launch_quantiles(b, dC, T, 1, dp, 1, dq)) return bfmmm_io_fail(std::string(who) + ": device allocation failed");
    double qc = 0.0;
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(&qc, dq, sizeof(double), hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(m.data(), dm, sizeof(double) * ncol, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(s.data(), ds, sizeof(double) * ncol, hipMemcpyDeviceToHost) != hipSuccess)
      return bfmmm_io_fail(std::string(who) + ": kernel launch or copy back failed");
    for (int j = 0; j < ncol; ++j) { lower[j] = m[(size_t)j] - qc * s[(size_t)j]; mid[j] = m[(size_t)j]; upper[j] = m[(size_t)j] + qc * s[(size_t)j]; }
  }
  if (hipGetLastError() != hipSuccess) return bfmmm_io_fail(std::string(who) + ": kernel launch failed");
  return 0;
}

// coef: T x P (one row per draw); B: n_t x P row-major.  upper / mid / lower: n_t; trace: T x n_t column-major (draw fastest) or NULL
extern "C" int bfmmm_post_bands(const double* coef, int32_t T, int32_t P, const double* B, int32_t n_t, double alpha, int32_t simultaneous,
                                int32_t device, double* upper, double* mid, double* lower, double* trace) {
  if (!coef || !B || !upper || !mid || !lower || T < 2 || P < 1 || n_t < 1) return bfmmm_io_fail("bfmmm_post_bands: bad arguments");
  if (T > TBIG) return bfmmm_io_fail("bfmmm_post_bands: at most 2^24 kept draws");
  if (select_device(device, "bfmmm_post_bands")) return 1;
  Bufs b;
  double *dc, *dB, *df;
  if (!b.put(&dc, coef, (size_t)T * P) || !b.put(&dB, B, (size_t)n_t * P) || !b.put(&df, nullptr, (size_t)T * n_t))
    return bfmmm_io_fail("bfmmm_post_bands: device allocation or copy failed");
  hipLaunchKernelGGL(k_bands_fsamp, dim3(n_t, std::min((T + QT - 1) / QT, 64)), dim3(QT), 0, 0, dB, dc, T, P, df);
  if (table_bands(b, df, T, n_t, alpha, simultaneous, upper, mid, lower, "bfmmm_post_bands")) return 1;
  if (trace && hipMemcpy(trace, df, sizeof(double) * (size_t)T * n_t, hipMemcpyDeviceToHost) != hipSuccess)
    return bfmmm_io_fail("bfmmm_post_bands: copy back failed");
  return 0;
}

// coefL, coefM: (T M) x P, row t M + j = the transformed Phi.slice(j).row(l - 1) / row(m - 1) of draw t; B1: n1 x P, B2: n2 x P.
// upper / mid / lower: n1 x n2 column-major; trace: T x (n1 n2), draw fastest, or NULL
extern "C" int bfmmm_post_cov_bands(const double* coefL, const double* coefM, int32_t T, int32_t M, int32_t P, const double* B1, int32_t n1,
                                    const double* B2, int32_t n2, double alpha, int32_t simultaneous, int32_t device, double* upper, double* mid,
                                    double* lower, double* trace) {
  if (!coefL || !coefM || !B1 || !B2 || !upper || !mid || !lower || T < 2 || M < 1 || P < 1 || n1 < 1 || n2 < 1)
    return bfmmm_io_fail("bfmmm_post_cov_bands: bad arguments");
  if (T > TBIG) return bfmmm_io_fail("bfmmm_post_cov_bands: at most 2^24 kept draws");
  if (select_device(device, "bfmmm_post_cov_bands")) return 1;
  Bufs b;
  double *dl, *dm2, *dB1, *dB2, *dA, *dBm, *dV;
  const size_t TM = (size_t)T * M, nc = (size_t)n1 * n2;
  if (!b.put(&dl, coefL, TM * P) || !b.put(&dm2, coefM, TM * P) || !b.put(&dB1, B1, (size_t)n1 * P) || !b.put(&dB2, B2, (size_t)n2 * P) ||
      !b.put(&dA, nullptr, TM * n1) || !b.put(&dBm, nullptr, TM * n2) || !b.put(&dV, nullptr, (size_t)T * nc))
    return bfmmm_io_fail("bfmmm_post_cov_bands: device allocation or copy failed");
  const int gy = (int)std::min<size_t>((TM + QT - 1) / QT, 64);
  hipLaunchKernelGGL(k_bands_fsamp, dim3(n1, gy), dim3(QT), 0, 0, dB1, dl, (int)TM, P, dA);
  hipLaunchKernelGGL(k_bands_fsamp, dim3(n2, gy), dim3(QT), 0, 0, dB2, dm2, (int)TM, P, dBm);
  hipLaunchKernelGGL(k_bands_outer, dim3((unsigned)nc, std::min((T + QT - 1) / QT, 64)), dim3(QT), 0, 0, dA, dBm, T, M, n1, dV);
  if (table_bands(b, dV, T, (int)nc, alpha, simultaneous, upper, mid, lower, "bfmmm_post_cov_bands")) return 1;
  if (trace && hipMemcpy(trace, dV, sizeof(double) * (size_t)T * nc, hipMemcpyDeviceToHost) != hipSuccess)
    return bfmmm_io_fail("bfmmm_post_cov_bands: copy back failed");
  return 0;
}

// bands of a table given as is: V is T x ncol, draw fastest (FSamplePaths' Path_trace of one curve or of all curves)
extern "C" int bfmmm_post_table_bands(const double* V, int32_t T, int32_t ncol, double alpha, int32_t simultaneous, int32_t device,
                                      double* upper, double* mid, double* lower) {
  if (!V || !upper || !mid || !lower || T < 2 || ncol < 1) return bfmmm_io_fail("bfmmm_post_table_bands: bad arguments");
  if (T > TBIG) return bfmmm_io_fail("bfmmm_post_table_bands: at most 2^24 kept draws");
  if (select_device(device, "bfmmm_post_table_bands")) return 1;
  Bufs b;
  double* dV;
  if (!b.put(&dV, V, (size_t)T * ncol)) return bfmmm_io_fail("bfmmm_post_table_bands: device allocation or copy failed");
  return table_bands(b, dV, T, ncol, alpha, simultaneous, upper, mid, lower, "bfmmm_post_table_bands");
}
