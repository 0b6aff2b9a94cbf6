// Likelihood-based post-processing on the device (include/bfmmm_post.h): fitted values of every observation under every
// saved draw, reduced three ways in one pass -- per-draw log-likelihood (FLLik, the first DIC term), per-observation mean
// density (the second DIC term, calcDIC2) and per-observation mean fitted value (FAIC / FBIC).
//
// Grid (curve i, chunk of draws).  A workgroup walks its draws in tiles of G = min(256 / JW, 32) draws, JW = the curve's
// observation count rounded up to a power of two:
//   phase 0  Z_i.(t), chi_i.(t), sigma^2(t) of the tile                                  -> LDS
//   phase 1  c_p(t) = sum_k Z_ik [theta_{k,0} + sum_m chi_im theta_{k,m}](p), (g, p) pairs -> LDS   (skip Z_ik == 0 as the
//            reference does; covariates folded into the rows: theta + sum_d x_id thetaX_d)
//   phase 2  thread (j, g): f = B_ij' c(t_g), residual, log-density term -> LDS; density and f accumulate in registers
//   phase 3  thread g: fixed-order sum over j -> llpart[i][t]
// and leaves per-observation sums over its chunk in pdf_part / fit_part.  k_post_reduce sums curves (per draw) and chunks
// (per observation) in a fixed order: results do not depend on the launch geometry's scheduling.
//
// Bound: HBM.  Algorithmic bytes per draw = 8 [ n (K + M) + K (M + 1) P (1 + D) + 1 ] read + 8 n written (llpart); the
// basis rows of a curve (n_i x P) are staged in LDS once per workgroup.  Draw parameters arrive transposed to the
// sampler's row layout theta[t][r][p] (host, one pass) so that phase 1 reads are contiguous in p.
#include <hip/hip_runtime.h>

#include <cmath>
#include <string>
#include <vector>

#include "../../include/bfmmm_post.h"

int bfmmm_io_fail(const std::string& m);      // entry_points.cpp: sets bfmmm_entry_last_error

namespace {

constexpr int NJ = 4;            // observations per thread: n_i <= 1024
constexpr int GMAX = 32;         // draws per tile
constexpr int BL_MAX = 4096;     // doubles of basis rows staged per curve
constexpr int PMAXP = 65;        // row stride of the coefficient tile (P <= 64, odd)
constexpr int WMAX = 20;         // K + M + 2 <= 20

struct PostDev {
  int n, K, P, M, D, T, first_kept, tchunk;
  long long n_obs;
  const long long* off;
  const double *y, *B, *X, *theta, *thetaX, *Z, *chi, *sigma;
  double *llpart, *pdf_part, *fit_part;
};

__global__ __launch_bounds__(256) void k_post_pointwise(PostDev a) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int i = blockIdx.x, ch = blockIdx.y, tid = threadIdx.x;
  const long long o = a.off[i];
  const int ni = (int)(a.off[i + 1] - o);
  const int P = a.P, K = a.K, M = a.M, D = a.D, R = K * (M + 1), n = a.n, T = a.T;
  int JW = 1;
  while (JW < min(ni, 256)) JW <<= 1;
  const int G = min(256 / JW, GMAX);
  const int LW = JW * NJ;                   // row stride of the per-tile term table
  const int PS = P | 1;                     // odd row stride of the staged basis rows
  const bool staged = (size_t)ni * PS <= BL_MAX;
  double* sC = sm;                          // GMAX x PMAXP
  double* sW = sC + GMAX * PMAXP;           // GMAX x WMAX : Z (K), chi (M), sd, sigma^2
  double* sLL = sW + GMAX * WMAX;           // 2 x 1024: terms of the tile; the final reduction's scratch
  double* sB = sLL + 2 * 1024;              // BL_MAX
  double* sX = sB + BL_MAX;                 // 8
  if (staged)
    for (int e = tid; e < ni * P; e += 256) { const int j = e / P, p = e - j * P; sB[j * PS + p] = a.B[(size_t)(o + j) * P + p]; }
  if (tid < D) sX[tid] = a.X[i + (size_t)n * tid];
  const int jl = tid % JW, gl = tid / JW;
  double acc_pdf[NJ], acc_fit[NJ], yv[NJ];
#pragma unroll
  for (int jj = 0; jj < NJ; ++jj) {
    acc_pdf[jj] = 0.0; acc_fit[jj] = 0.0;
    const int j = jl + jj * JW;
    yv[jj] = (j < ni) ? a.y[o + j] : 0.0;
  }
  const int t_lo = ch * a.tchunk, t_hi = min(T, t_lo + a.tchunk);
  for (int tb = t_lo; tb < t_hi; tb += G) {
    __syncthreads();
    // ---- phase 0: the curve's membership and scores under the tile's draws ----
    for (int e = tid; e < G * (K + M + 1); e += 256) {
      const int g = e / (K + M + 1), w = e - g * (K + M + 1), t = min(tb + g, t_hi - 1);
      double v;
      if (w < K) v = a.Z[i + (size_t)n * (w + (size_t)K * t)];
      else if (w < K + M) v = a.chi[i + (size_t)n * ((w - K) + (size_t)M * t)];
      else { v = a.sigma[t]; sW[g * WMAX + K + M + 1] = v; v = sqrt(v); }
      sW[g * WMAX + w] = v;
    }
    __syncthreads();
    // ---- phase 1: coefficient vectors ----
    for (int e = tid; e < G * P; e += 256) {
      const int g = e / P, p = e - g * P, t = min(tb + g, t_hi - 1);
      const double* th = a.theta + (size_t)t * R * P + p;
      const double* thx = a.thetaX ? a.thetaX + (size_t)t * R * D * P + p : nullptr;
      const double* w = sW + g * WMAX;
      double c = 0.0;
      for (int k = 0; k < K; ++k) {
        const double zk = w[k];
        if (zk != 0.0) {                                           // CalculateLikelihood.h:29, :70
          double acc = 0.0;
          for (int mt = 0; mt <= M; ++mt) {
            const int r = k * (M + 1) + mt;
            double v = th[(size_t)r * P];
            for (int dd = 0; dd < D; ++dd) v += sX[dd] * thx[((size_t)r * D + dd) * P];
            acc += (mt == 0) ? v : w[K + mt - 1] * v;
          }
          c += zk * acc;
        }
      }
      sC[g * PMAXP + p] = c;
    }
    __syncthreads();
    // ---- phase 2: fitted values and density terms ----
    if (gl < G && tb + gl < t_hi) {
      const int t = tb + gl;
      const double sd = sW[gl * WMAX + K + M];
      const double lsd = log(sd);
      const double* cg = sC + gl * PMAXP;
      const bool kept = t >= a.first_kept;
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) {
        const int j = jl + jj * JW;
        if (j < ni) {
          double f = 0.0;
          if (staged) { const double* br = sB + j * PS; for (int p = 0; p < P; ++p) f += br[p] * cg[p]; }
          else { const double* br = a.B + (size_t)(o + j) * P; for (int p = 0; p < P; ++p) f += br[p] * cg[p]; }
          const double z = (yv[jj] - f) / sd;
          sLL[gl * LW + j] = -(0.91893853320467274178 + 0.5 * z * z + lsd);      // R::dnorm(., ., ., log = true)
          if (kept) { acc_pdf[jj] += exp(-0.5 * z * z) / (sd * 2.50662827463100050242); acc_fit[jj] += f; }
        }
      }
    }
    __syncthreads();
    // ---- phase 3: the curve's log-likelihood under each draw of the tile ----
    if (tid < G && tb + tid < t_hi) {
      double s = 0.0;
      for (int j = 0; j < ni; ++j) s += sLL[tid * LW + j];
      a.llpart[(size_t)i * T + tb + tid] = s;
    }
  }
  // ---- per-observation sums of this chunk: the draw lanes meet in LDS, fixed order ----
  for (int which = 0; which < 2; ++which) {
    __syncthreads();
    if (gl < G)
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) {
        const int j = jl + jj * JW;
        if (j < ni) sLL[gl * LW + j] = which ? acc_fit[jj] : acc_pdf[jj];
      }
    __syncthreads();
    for (int j = tid; j < ni; j += 256) {
      double s = 0.0;
      for (int g = 0; g < G; ++g) s += sLL[g * LW + j];
      (which ? a.fit_part : a.pdf_part)[(size_t)ch * a.n_obs + o + j] = s;
    }
  }
}

// llik[t] = sum_i llpart[i][t];  mean_pdf / mean_fit[obs] = sum_ch part[ch][obs] / kept
__global__ __launch_bounds__(256) void k_post_reduce(PostDev a, int NCH, double* llik, double* mean_pdf, double* mean_fit) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e < a.T) {
    double s = 0.0;
    for (int i = 0; i < a.n; ++i) s += a.llpart[(size_t)i * a.T + e];
    llik[e] = s;
  }
  if (e < a.n_obs) {
    const double kept = (double)(a.T - a.first_kept);
    double s1 = 0.0, s2 = 0.0;
    for (int c = 0; c < NCH; ++c) { s1 += a.pdf_part[(size_t)c * a.n_obs + e]; s2 += a.fit_part[(size_t)c * a.n_obs + e]; }
    mean_pdf[e] = s1 / kept;
    mean_fit[e] = s2 / kept;
  }
}

struct DevBufs {
  std::vector<void*> p;
  ~DevBufs() { for (void* q : p) (void)hipFree(q); }
  template <class Tp>
  bool put(Tp** out, const Tp* host, size_t count) {
    void* d = nullptr;
    if (hipMalloc(&d, std::max<size_t>(count, 1) * sizeof(Tp)) != hipSuccess) return false;
    p.push_back(d);
    if (host && count && hipMemcpy(d, host, count * sizeof(Tp), hipMemcpyHostToDevice) != hipSuccess) return false;
    *out = (Tp*)d;
    return true;
  }
};

}  // namespace

extern "C" int bfmmm_post_pointwise(const bfmmm_post_input* in, int32_t first_kept, double* llik, double* mean_pdf, double* mean_fit) {
  if (!in || !in->offsets || !in->y || !in->B || !in->nu || !in->Phi || !in->Z || !in->chi || !in->sigma)
    return bfmmm_io_fail("bfmmm_post_pointwise: null argument");
  const int n = in->n, K = in->K, P = in->P, M = in->M, D = in->X ? in->D : 0, T = in->T;
  if (n < 1 || K < 1 || P < 1 || M < 0 || T < 1 || first_kept < 0 || first_kept >= T)
    return bfmmm_io_fail("bfmmm_post_pointwise: bad dimensions");
  if (P > 64 || K + M + 2 > WMAX || D > 8) return bfmmm_io_fail("bfmmm_post_pointwise: P <= 64, K + M <= 18 and D <= 8 in this build");
  const long long n_obs = in->offsets[n];
  for (int i = 0; i < n; ++i)
    if (in->offsets[i + 1] - in->offsets[i] > 256 * NJ)
      return bfmmm_io_fail("bfmmm_post_pointwise: at most 1024 observations per curve in this build");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return bfmmm_io_fail("bfmmm_post_pointwise: no HIP device (the MI355X library has no CPU path)");
  if (hipSetDevice(in->device) != hipSuccess) return bfmmm_io_fail("bfmmm_post_pointwise: cannot select the device");
  // draws -> theta[t][r = k (M + 1) + mt][p] (+ thetaX[t][r][d][p])
  const int R = K * (M + 1);
  std::vector<double> theta((size_t)T * R * P), thetaX;
  for (int t = 0; t < T; ++t)
    for (int k = 0; k < K; ++k)
      for (int p = 0; p < P; ++p) {
        theta[((size_t)t * R + k * (M + 1)) * P + p] = in->nu[k + (size_t)K * (p + (size_t)P * t)];
        for (int m = 0; m < M; ++m)
          theta[((size_t)t * R + k * (M + 1) + m + 1) * P + p] = in->Phi[(size_t)t * K * P * M + k + (size_t)K * (p + (size_t)P * m)];
      }
  if (D > 0) {
    thetaX.assign((size_t)T * R * D * P, 0.0);
    for (int t = 0; t < T; ++t)
      for (int k = 0; k < K; ++k)
        for (int dd = 0; dd < D; ++dd)
          for (int p = 0; p < P; ++p) {
            if (in->eta)
              thetaX[(((size_t)t * R + k * (M + 1)) * D + dd) * P + p] = in->eta[(size_t)t * P * D * K + p + (size_t)P * (dd + (size_t)D * k)];
            if (in->xi)
              for (int m = 0; m < M; ++m)
                thetaX[(((size_t)t * R + k * (M + 1) + m + 1) * D + dd) * P + p] =
                    in->xi[((size_t)t * K + k) * P * D * M + p + (size_t)P * (dd + (size_t)D * m)];
          }
  }
  // chunks of draws: enough workgroups for the 256 CUs, tiles stay whole
  int tchunk = T;
  while ((long long)n * ((T + tchunk - 1) / tchunk) < 2048 && tchunk > GMAX) tchunk = (tchunk + 1) / 2;
  tchunk = (tchunk + GMAX - 1) / GMAX * GMAX;
  const int NCH = (T + tchunk - 1) / tchunk;
  DevBufs db;
  PostDev a{};
  a.n = n; a.K = K; a.P = P; a.M = M; a.D = D; a.T = T; a.first_kept = first_kept; a.tchunk = tchunk; a.n_obs = n_obs;
  std::vector<long long> off(in->offsets, in->offsets + n + 1);
  double *d_ll, *d_pdf, *d_fit;
  bool ok = db.put((long long**)&a.off, off.data(), off.size()) && db.put((double**)&a.y, in->y, (size_t)n_obs) &&
            db.put((double**)&a.B, in->B, (size_t)n_obs * P) && db.put((double**)&a.theta, theta.data(), theta.size()) &&
            db.put((double**)&a.Z, in->Z, (size_t)n * K * T) && db.put((double**)&a.chi, in->chi, (size_t)n * M * T) &&
            db.put((double**)&a.sigma, in->sigma, (size_t)T) && db.put(&a.llpart, (const double*)nullptr, (size_t)n * T) &&
            db.put(&a.pdf_part, (const double*)nullptr, (size_t)NCH * n_obs) && db.put(&a.fit_part, (const double*)nullptr, (size_t)NCH * n_obs) &&
            db.put(&d_ll, (const double*)nullptr, (size_t)T) && db.put(&d_pdf, (const double*)nullptr, (size_t)n_obs) &&
            db.put(&d_fit, (const double*)nullptr, (size_t)n_obs);
  if (ok && D > 0) ok = db.put((double**)&a.X, in->X, (size_t)n * D) && db.put((double**)&a.thetaX, thetaX.data(), thetaX.size());
  if (!ok) { (void)hipGetLastError(); return bfmmm_io_fail("bfmmm_post_pointwise: device allocation or copy failed"); }
  const size_t lds = ((size_t)GMAX * PMAXP + (size_t)GMAX * WMAX + 2 * 1024 + BL_MAX + 8) * sizeof(double);
  (void)hipFuncSetAttribute((const void*)k_post_pointwise, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(k_post_pointwise, dim3(n, NCH), dim3(256), lds, 0, a);
  const long long tot = std::max<long long>(T, n_obs);
  hipLaunchKernelGGL(k_post_reduce, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, 0, a, NCH, d_ll, d_pdf, d_fit);
  if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess)
    return bfmmm_io_fail("bfmmm_post_pointwise: kernel launch failed");
  if ((llik && hipMemcpy(llik, d_ll, sizeof(double) * T, hipMemcpyDeviceToHost) != hipSuccess) ||
      (mean_pdf && hipMemcpy(mean_pdf, d_pdf, sizeof(double) * n_obs, hipMemcpyDeviceToHost) != hipSuccess) ||
      (mean_fit && hipMemcpy(mean_fit, d_fit, sizeof(double) * n_obs, hipMemcpyDeviceToHost) != hipSuccess))
    return bfmmm_io_fail("bfmmm_post_pointwise: copy back failed");
  return 0;
}
