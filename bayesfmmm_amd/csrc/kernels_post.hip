// Likelihood-based post-processing on the device (include/bfmmm_post.h): fitted values of every observation under every
// saved draw, reduced three ways in one pass -- per-draw log-likelihood (FLLik, the first DIC term), per-observation mean
// density (the second DIC term, calcDIC2) and per-observation mean fitted value (FAIC / FBIC).
//
// Grid (curve i, chunk of draws).  A workgroup walks its draws in tiles of G = min(256 / JW, 32) draws, JW = the curve's
// observation count rounded up to a power of two:
//   phase 0  Z_i.(t), chi_i.(t), sigma^2(t) of the tile                                  -> LDS
//   phase 1  c_p(t) = sum_k Z_ik [theta_{k,0} + sum_m chi_im theta_{k,m}](p), (g, p) pairs -> LDS   (skip Z_ik == 0 as the
//            reference does; covariates folded into the rows: theta + sum_d x_id thetaX_d)
//   phase 2  thread (j, draw lane): f = B_ij' c(t) for NG draws at once over the row's non-zero window, residual,
//            log-density term; density and f accumulate in registers; the terms of a draw meet by a butterfly over the
//            observation lanes (fixed order) -> llpart[i][t]
// and leaves per-observation sums over its chunk in pdf_part / fit_part.  k_post_reduce sums curves (per draw) and chunks
// (per observation) in a fixed order: results do not depend on the launch geometry's scheduling.
//
// Algorithmic bytes per draw = 8 [ n (K + M) + K (M + 1) P (1 + D) + 1 ] read + 8 n written (llpart) -- but the pass is not
// HBM-bound as built: the draw's parameter rows are re-read from L2 by every curve's workgroup and each (observation, kept
// draw) costs one fp64 exp (DESIGN.md 7b has the measurement and the next step).  The
// non-zero windows of a curve's basis rows (n_i x W, W = degree + 1 for B-splines) are staged in LDS once per workgroup;
// the next tile's Z / chi / sigma^2 are requested while the current tile is evaluated.  Draw parameters arrive transposed to the
// sampler's row layout theta[t][r][p] (host, one pass) so that phase 1 reads are contiguous in p.
#include <hip/hip_runtime.h>

#include <cmath>
#include <string>
#include <vector>

#include "../../include/bfmmm_post.h"
#include "rng.hpp"

int bfmmm_io_fail(const std::string& m);      // entry_points.cpp: sets bfmmm_entry_last_error

namespace {

constexpr int NJ = 4;            // observations per thread: n_i <= 1024
constexpr int GMAX = 32;         // draws per tile
constexpr int BL_MAX = 1536;     // doubles of basis-row windows staged per curve
constexpr int WMAX = 26;         // K + M + 2 <= 26: the sampler's K <= 8, n_eigen <= 16
constexpr int NG = 8;            // draws per thread in the fitted-value phase
constexpr int KMAXP = 16;         // K <= 16 in the CPO tile tables
constexpr int NWR = 4;           // prefetch registers: GMAX (K + M + 1) <= 1024 items per tile

struct PostDev {
  int n, K, P, M, D, T, first_kept, tchunk;
  long long n_obs;
  const long long* off;
  const double *y, *Bc, *X, *theta, *thetaX, *Z, *chi, *sigma;      // Bc: the rows' non-zero windows, n_obs x W
  const int* bstart;                                                // first column of each row's window
  int W, need_pdf;
  double *llpart, *pdf_part, *fit_part;
};

// A basis row's non-zero window against a coefficient window: the row either sits in LDS (staged) or in global memory.  Two
// functions with differently typed pointers: a pointer SELECTED between the two would be a generic one, and every load of the
// inner loop a FLAT load.
typedef const __attribute__((address_space(3))) double* lds_cptr;
__device__ inline double window_dot_lds(const double* br_lds, const double* cg, int W) {
  lds_cptr br = (lds_cptr)br_lds;
  double f = 0.0;
  for (int w = 0; w < W; ++w) f += br[w] * cg[w];
  return f;
}
__device__ inline double window_dot(const double* br, const double* cg, int W) {
  double f = 0.0;
  for (int w = 0; w < W; ++w) f += br[w] * cg[w];
  return f;
}

__global__ __launch_bounds__(256, 3) void k_post_pointwise(PostDev a) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int i = blockIdx.x, ch = blockIdx.y, tid = threadIdx.x;
  const long long o = a.off[i];
  const int ni = (int)(a.off[i + 1] - o);
  const int P = a.P, K = a.K, M = a.M, D = a.D, R = K * (M + 1), n = a.n, T = a.T, W = a.W;
  int JW = 1;
  while (JW < min(ni, 256)) JW <<= 1;
  const int GT = min(256 / JW, GMAX / NG);  // draw lanes
  const int G = GT * NG;                    // draws per tile
  const int LW = JW * NJ;                   // row stride of the per-observation tables
  const int WS = W | 1;                     // odd row stride of the staged basis rows
  const int CS = P | 1;                     // odd row stride of the coefficient tile
  const bool staged = (size_t)ni * WS <= BL_MAX;
  const int NWI = G * (K + M + 1);          // membership / score / variance items of a tile
  double* sC = sm;                          // GMAX x CS
  double* sW = sC + GMAX * CS;              // GMAX x WMAX : Z (K), chi (M), 1 / sd, log sd
  double* sAccP = sW + GMAX * WMAX;         // 1024: per (draw lane, observation) sums of the density over the kept draws
  double* sAccF = sAccP + 1024;             // 1024: ... of the fitted value
  double* sRed = sAccF + 1024;              // 64: wave partials of a draw's log-likelihood
  double* sB = sRed + 64;                   // BL_MAX
  double* sX = sB + BL_MAX;                 // 8
  if (staged)
    for (int e = tid; e < ni * W; e += 256) { const int j = e / W, w = e - j * W; sB[j * WS + w] = a.Bc[(size_t)(o + j) * W + w]; }
  if (tid < D) sX[tid] = a.X[i + (size_t)n * tid];
  for (int e = tid; e < 2048; e += 256) sAccP[e] = 0.0;      // (sAccP and sAccF are adjacent)
  const int jl = tid % JW, gl = tid / JW;
  const double y0 = (jl < ni) ? a.y[o + jl] : 0.0;           // the first (usually the only) observation of this thread
  const int st0 = (jl < ni) ? a.bstart[o + jl] : 0;
  const int t_lo = ch * a.tchunk, t_hi = min(T, t_lo + a.tchunk);
  // the curve's membership, scores and the variance under a tile's draws are requested one tile ahead
  double wreg[NWR];
  auto request = [&](int tb) {
#pragma unroll
    for (int u = 0; u < NWR; ++u) {
      const int e = min(tid + 256 * u, NWI - 1);
      const int g = e / (K + M + 1), w = e - g * (K + M + 1), t = min(tb + g, t_hi - 1);
      const double* src = (w < K) ? a.Z + i + (size_t)n * (w + (size_t)K * t)
                        : (w < K + M) ? a.chi + i + (size_t)n * ((w - K) + (size_t)M * t) : a.sigma + t;
      wreg[u] = *src;
    }
  };
  request(t_lo);
  for (int tb = t_lo; tb < t_hi; tb += G) {
    __syncthreads();
#pragma unroll
    for (int u = 0; u < NWR; ++u) {
      const int e = tid + 256 * u;
      if (e < NWI) {
        const int g = e / (K + M + 1), w = e - g * (K + M + 1);
        if (w < K + M) sW[g * WMAX + w] = wreg[u];
        else { const double sd = sqrt(wreg[u]); sW[g * WMAX + K + M] = 1.0 / sd; sW[g * WMAX + K + M + 1] = log(sd); }
      }
    }
    if (tb + G < t_hi) request(tb + G);
    __syncthreads();
    // ---- phase 1: coefficient vectors.  The parameter rows of a draw are L2 hits shared by every curve; a thread reads
    //      them in batches of twelve independent loads (one dependent load per row made
    //      this phase the whole kernel) ----
    for (int e = tid; e < G * P; e += 256) {
      constexpr int NB = 12;
      const int g = e / P, p = e - g * P, t = min(tb + g, t_hi - 1);
      const unsigned base = (unsigned)((t * R) * P + p);           // element offsets from uniform bases: 32-bit address math
      const double* w = sW + g * WMAX;
      double c = 0.0;
      for (int r0 = 0; r0 < R; r0 += NB) {
        double v[NB];
#pragma unroll
        for (int u = 0; u < NB; ++u) v[u] = a.theta[base + (unsigned)(min(r0 + u, R - 1) * P)];
        for (int dd = 0; dd < D; ++dd) {
          double qv[NB];
#pragma unroll
          for (int u = 0; u < NB; ++u) qv[u] = a.thetaX[(base - (unsigned)p) * (unsigned)D + (unsigned)((min(r0 + u, R - 1) * D + dd) * P + p)];
          const double x = sX[dd];
#pragma unroll
          for (int u = 0; u < NB; ++u) v[u] += x * qv[u];
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
          const int r = r0 + u;
          if (r < R) {
            const int k = r / (M + 1), mt = r - k * (M + 1);
            const double zk = w[k];
            const double x = (zk != 0.0) ? v[u] : 0.0;                 // CalculateLikelihood.h:29, :70 (the Z_ik == 0 skip)
            c += zk * ((mt == 0) ? x : w[K + mt - 1] * x);
          }
        }
      }
      sC[g * CS + p] = c;
    }
    __syncthreads();
    // ---- phase 2: fitted values and density terms, one (observation, draw) pair at a time (the row's non-zero window is
    //      short; one copy of the exp / log code keeps the kernel at four waves per SIMD) ----
    if (gl < GT) {
#pragma unroll 1
      for (int q = 0; q < NG; ++q) {
        const int g = gl * NG + q, t = tb + g;
        const bool on = t < t_hi;
        const double isd = sW[g * WMAX + K + M], lsd = sW[g * WMAX + K + M + 1];
        const bool kept = on && t >= a.first_kept;
        double ll = 0.0;
#pragma unroll 1
        for (int j = jl; j < ni; j += JW) {
          const double yj = (j == jl) ? y0 : a.y[o + j];
          const int stj = (j == jl) ? st0 : a.bstart[o + j];
          const double* cg = sC + g * CS + stj;
          const double f = staged ? window_dot_lds(sB + j * WS, cg, W) : window_dot(a.Bc + (size_t)(o + j) * W, cg, W);
          const double z = (yj - f) * isd;
          const double lt = -(0.91893853320467274178 + 0.5 * z * z + lsd);     // R::dnorm(., ., ., log = true)
          if (on) ll += lt;
          if (kept) {
            if (a.need_pdf) sAccP[gl * LW + j] += exp(lt);
            sAccF[gl * LW + j] += f;
          }
        }
        // the curve's log-likelihood under draw t: butterfly over the observation lanes (fixed order)
        for (int off = min(JW, 64) >> 1; off > 0; off >>= 1) ll += __shfl_xor(ll, off);
        if (JW <= 64) { if (jl == 0 && on) a.llpart[(size_t)i * T + t] = ll; }
        else if ((tid & 63) == 0) sRed[(tid >> 6) * NG + q] = ll;
      }
    }
    if (JW > 64) {
      __syncthreads();
      if (tid < G && tb + tid < t_hi) {
        const int g2 = tid / NG, q = tid - g2 * NG, wpl = JW >> 6;      // waves per draw lane
        double s = 0.0;
        for (int wv = 0; wv < wpl; ++wv) s += sRed[(g2 * wpl + wv) * NG + q];
        a.llpart[(size_t)i * T + tb + tid] = s;
      }
    }
  }
  // ---- per-observation sums of this chunk: the draw lanes' tables, fixed order ----
  __syncthreads();
  for (int j = tid; j < ni; j += 256) {
    double s1 = 0.0, s2 = 0.0;
    for (int g = 0; g < GT; ++g) { s1 += sAccP[g * LW + j]; s2 += sAccF[g * LW + j]; }
    a.pdf_part[(size_t)ch * a.n_obs + o + j] = s1;
    a.fit_part[(size_t)ch * a.n_obs + o + j] = s2;
  }
}

// llik[t] = sum_i llpart[i][t];  mean_pdf / mean_fit[obs] = sum_ch part[ch][obs] / kept
__global__ __launch_bounds__(256) void k_post_reduce(PostDev a, int NCH, double* llik, double* mean_pdf, double* mean_fit, double* joint) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (joint && e < a.n) {       // mean over the kept draws of the curve's joint density (calcDIC2MV, CalculateLikelihood.h:172-194)
    double s = 0.0;
    for (int t = a.first_kept; t < a.T; ++t) s += exp(a.llpart[(size_t)e * a.T + t]);
    joint[e] = s / (double)(a.T - a.first_kept);
  }
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

// ---------------------------------------------------------------------------------------------------------------------
// Conditional predictive ordinates (calcLikelihoodCPO, CalculateLikelihood.h:344-389): per curve and kept draw the
// MARGINAL log-density of y_i (scores chi integrated out),
//   y_i ~ N( B_i c,  sigma^2 I + U U' ),   c = sum_k Z_ik (nu_k + eta_k x_i),   U = B_i [v_1 .. v_M],  v_m = sum_k Z_ik (phi_km + xi_km x_i)
// The reference forms the n_i x n_i covariance and calls log_det_sympd / inv_sympd; the rank-M structure gives the same
// numbers from an M x M system:  log det = (n_i - M) log sigma^2 + log det(sigma^2 I_M + U'U),
//   r' Cov^-1 r = ( r'r - (U'r)' (sigma^2 I_M + U'U)^-1 (U'r) ) / sigma^2.
// One workgroup per curve; tiles of G draws: (1) c, v_m -> LDS; (2) thread (j, draw): residual and the row of U -> LDS;
// (3) thread (draw, sum q, segment): the sums r'r, U'r, U'U over a quarter of the observations, fixed order; (4) thread
// (draw): M x M Cholesky, the draw's log-density -> cpo_ll[i][t].  k_post_cpo_reduce then takes the harmonic mean in the
// reference's stabilised form (:381-386).  No Z_ik == 0 skip here: the reference's loop has none.
constexpr int CPO_MREG = 8;        // M <= 8: the M x M system of a draw lives in registers (unrolled); 9 <= M <= 16: in LDS
constexpr int CPO_MMAX = 16;       // the sampler's n_eigen limit
constexpr int CPO_GMAX = 16;       // most draws of a tile
constexpr int CPO_LDS_BUDGET = 17000;      // doubles of the per-draw tiles (133 KB of the 160 KB)

// per-draw LDS doubles: coefficient tile (M + 1) x CS, U tile (M + 1) x NIP, segment sums NQ x 4, the LDS system of M > 8
__host__ __device__ inline int cpo_draw_doubles(int M, int CS, int NIP) {
  const int M1 = M + 1, NQ = M * (M + 1) / 2 + M + 1;
  return M1 * CS + M1 * NIP + NQ * 4 + ((M > CPO_MREG) ? M * M + M : 0);
}

// G: draws per tile, chosen on the host for the longest curve (NIPX = its padded length)
__global__ __launch_bounds__(256) void k_post_cpo(PostDev a, double* cpo_ll, int G, int NIPX) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int i = blockIdx.x, tid = threadIdx.x;
  const long long o = a.off[i];
  const int ni = (int)(a.off[i + 1] - o);
  const int P = a.P, K = a.K, M = a.M, D = a.D, R = K * (M + 1), n = a.n, T = a.T, W = a.W;
  const int M1 = M + 1, NQ = M * (M + 1) / 2 + M + 1;
  const int NIP = (ni + 3) & ~3;                       // observations padded to four segments
  const int CS = P | 1, WS = W | 1;
  const bool staged = (size_t)ni * WS <= BL_MAX;
  double* sV = sm;                          // G x M1 x CS
  double* sU = sV + G * M1 * CS;            // G x M1 x NIPX  (row 0: the residual)
  double* sP = sU + G * M1 * NIPX;          // G x NQ x 4 segment sums
  double* sA = sP + G * NQ * 4;             // G x (M x M + M): the system of a draw when M > 8
  double* sW = sA + ((M > CPO_MREG) ? G * (M * M + M) : 0);        // G x (K + 1): Z_i.(t), sigma^2(t)
  double* sB = sW + CPO_GMAX * (KMAXP + 1); // BL_MAX
  double* sX = sB + BL_MAX;                 // 8
  if (staged)
    for (int e = tid; e < ni * W; e += 256) { const int j = e / W, w = e - j * W; sB[j * WS + w] = a.Bc[(size_t)(o + j) * W + w]; }
  if (tid < D) sX[tid] = a.X[i + (size_t)n * tid];
  for (int tb = a.first_kept; tb < T; tb += G) {
    const int gn = min(G, T - tb);
    __syncthreads();
    for (int e = tid; e < gn * (K + 1); e += 256) {
      const int g = e / (K + 1), w = e - g * (K + 1), t = tb + g;
      sW[g * (KMAXP + 1) + w] = (w < K) ? a.Z[i + (size_t)n * (w + (size_t)K * t)] : a.sigma[t];
    }
    __syncthreads();
    // (1) c and v_m of every draw of the tile
    for (int e = tid; e < gn * M1 * P; e += 256) {
      const int g = e / (M1 * P), r2 = e - g * M1 * P, mt = r2 / P, p = r2 - mt * P, t = tb + g;
      double v = 0.0;
      for (int k = 0; k < K; ++k) {
        const int r = k * M1 + mt;
        double x = a.theta[((size_t)t * R + r) * P + p];
        for (int dd = 0; dd < D; ++dd) x += sX[dd] * a.thetaX[(((size_t)t * R + r) * D + dd) * P + p];
        v += sW[g * (KMAXP + 1) + k] * x;
      }
      sV[(g * M1 + mt) * CS + p] = v;
    }
    __syncthreads();
    // (2) residual and the row of U for every (observation, draw)
    for (int e = tid; e < gn * NIP; e += 256) {
      const int g = e / NIP, j = e - g * NIP;
      if (j < ni) {
        const int st = a.bstart[o + j];
        for (int mt = 0; mt < M1; ++mt) {
          const double* cg = sV + (g * M1 + mt) * CS + st;
          const double f = staged ? window_dot_lds(sB + j * WS, cg, W) : window_dot(a.Bc + (size_t)(o + j) * W, cg, W);
          sU[(g * M1 + mt) * NIPX + j] = (mt == 0) ? a.y[o + j] - f : f;
        }
      } else {
        for (int mt = 0; mt < M1; ++mt) sU[(g * M1 + mt) * NIPX + j] = 0.0;
      }
    }
    __syncthreads();
    // (3) r'r (q = 0), U'r (q = 1 .. M), U'U (upper triangle) over four segments of the observations
    for (int e = tid; e < gn * NQ * 4; e += 256) {
      const int g = e / (NQ * 4), r2 = e - g * NQ * 4, q = r2 >> 2, seg = r2 & 3;
      int ra = 0, rb = 0;
      if (q >= 1 && q <= M) ra = q;
      else if (q > M) { int m1 = 0, rem = q - M - 1; while (rem >= M - m1) { rem -= M - m1; ++m1; } ra = m1 + 1; rb = m1 + 1 + rem; }
      const double* ua = sU + (g * M1 + ra) * NIPX + seg * (NIP >> 2);
      const double* ub = sU + (g * M1 + rb) * NIPX + seg * (NIP >> 2);
      double s_ = 0.0;
      for (int j = 0; j < (NIP >> 2); ++j) s_ += ua[j] * ub[j];
      sP[(g * NQ + q) * 4 + seg] = s_;
    }
    __syncthreads();
    // (4) the draw's marginal log-density from the M x M system
    if (tid < gn) {
      const int g = tid, t = tb + g;
      const double sig = sW[g * (KMAXP + 1) + K];
      auto sum4 = [&](int q) { const double* p4 = sP + (g * NQ + q) * 4; return (p4[0] + p4[1]) + (p4[2] + p4[3]); };
      const double rr = sum4(0);
      double logdet = 0.0, ww = 0.0;
      if (M <= CPO_MREG) {
        double A[CPO_MREG][CPO_MREG], b[CPO_MREG];
        int q = M + 1;
#pragma unroll
        for (int m1 = 0; m1 < CPO_MREG; ++m1) {
          b[m1] = (m1 < M) ? sum4(1 + m1) : 0.0;
#pragma unroll
          for (int m2 = 0; m2 < CPO_MREG; ++m2)
            if (m2 >= m1) { A[m1][m2] = (m1 < M && m2 < M) ? sum4(q) + ((m1 == m2) ? sig : 0.0) : ((m1 == m2) ? 1.0 : 0.0); if (m1 < M && m2 < M) ++q; }
        }
        // Cholesky A = L L' (lower in A[m2][m1], m2 >= m1), forward solve L w = b: log det = 2 sum log L_mm, b' A^-1 b = w'w
#pragma unroll
        for (int c = 0; c < CPO_MREG; ++c) {
          double dg = A[c][c];
#pragma unroll
          for (int k2 = 0; k2 < CPO_MREG; ++k2) if (k2 < c) dg -= A[c][k2] * A[c][k2];
          const double l = sqrt(dg);
          A[c][c] = l;
          double wv = b[c];
#pragma unroll
          for (int k2 = 0; k2 < CPO_MREG; ++k2) if (k2 < c) wv -= A[c][k2] * b[k2];
          wv /= l;
          b[c] = wv;
          if (c < M) { logdet += 2.0 * log(l); ww += wv * wv; }
#pragma unroll
          for (int r3 = 0; r3 < CPO_MREG; ++r3)
            if (r3 > c) {
              double v = A[c][r3];                 // upper entry (c, r3) holds the symmetric value
#pragma unroll
              for (int k2 = 0; k2 < CPO_MREG; ++k2) if (k2 < c) v -= A[r3][k2] * A[c][k2];
              A[r3][c] = v / l;
            }
        }
      } else {
        // the same elimination, same order of operations, on a system kept in LDS (9 <= M <= 16: 256 doubles do not fit registers)
        double* A = sA + g * (M * M + M);     // A[r * M + c]
        double* b = A + M * M;
        int q = M + 1;
        for (int m1 = 0; m1 < M; ++m1) {
          b[m1] = sum4(1 + m1);
          for (int m2 = m1; m2 < M; ++m2) { A[m1 * M + m2] = sum4(q) + ((m1 == m2) ? sig : 0.0); ++q; }
        }
        for (int c = 0; c < M; ++c) {
          double dg = A[c * M + c];
          for (int k2 = 0; k2 < c; ++k2) dg -= A[c * M + k2] * A[c * M + k2];
          const double l = sqrt(dg);
          A[c * M + c] = l;
          double wv = b[c];
          for (int k2 = 0; k2 < c; ++k2) wv -= A[c * M + k2] * b[k2];
          wv /= l;
          b[c] = wv;
          logdet += 2.0 * log(l); ww += wv * wv;
          for (int r3 = c + 1; r3 < M; ++r3) {
            double v = A[c * M + r3];
            for (int k2 = 0; k2 < c; ++k2) v -= A[r3 * M + k2] * A[c * M + k2];
            A[r3 * M + c] = v / l;
          }
        }
      }
      const double ld = (double)(ni - M) * log(sig) + logdet;
      const double quad = (rr - ww) / sig;
      cpo_ll[(size_t)i * T + t] = -(0.5 * ni) * 1.83787706640934548356 - 0.5 * ld - 0.5 * quad;
    }
  }
}

// CPO(i) = log(L) + min_l logl - log sum_l exp(min - logl_l) over the L kept draws (CalculateLikelihood.h:381-386)
__global__ __launch_bounds__(256) void k_post_cpo_reduce(PostDev a, const double* cpo_ll, double* cpo) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n) return;
  const double* row = cpo_ll + (size_t)i * a.T;
  double mn = row[a.first_kept];
  for (int t = a.first_kept + 1; t < a.T; ++t) mn = fmin(mn, row[t]);
  double ph = 0.0;
  for (int t = a.first_kept; t < a.T; ++t) ph += exp(mn - row[t]);
  cpo[i] = log((double)(a.T - a.first_kept)) + mn - log(ph);
}

float g_last_kernel_ms = 0.f;

// ---- posterior-predictive sample paths (FSamplePaths, src/PostProcessing.cpp:6599-6864) ----------------------------------------
// For curve i, kept draw t and observation l: the fitted value f_il(t) (all terms; Z_ik == 0 clusters skipped as the
// reference does, :6797), the mean-only value (nu and eta terms, :6800-6801) and the predictive draw
// rnorm(f_il(t), sqrt(sigma^2(t))) (:6810) -- from the keyed generator, word (seed, chain 0, iteration = draw index,
// UPD_SAMPLE_PATH, observation index).  grid (curves, chunks of kept draws); a thread owns (observation, draw) pairs.
// Outputs: for curve i a kept x n_i block, draw fastest, at kept * off[i].
__global__ __launch_bounds__(256) void k_post_paths(PostDev a, unsigned long long seed, int kept, double* paths, double* mean_only) {
  const int i = blockIdx.x;
  const long long o = a.off[i];
  const int ni = (int)(a.off[i + 1] - o);
  const int K = a.K, P = a.P, M = a.M, D = a.D, W = a.W, R = K * (M + 1);
  const int t0 = a.first_kept + blockIdx.y * a.tchunk, t1 = min(t0 + a.tchunk, a.T);
  const long long npair = (long long)ni * (t1 - t0);
  for (long long e = threadIdx.x; e < npair; e += 256) {
    const int l = (int)(e % ni), t = t0 + (int)(e / ni);
    const double* bw = a.Bc + (size_t)(o + l) * W;
    const int st = a.bstart[o + l];
    const double* th = a.theta + (size_t)t * R * P;
    const double* thx = (D > 0) ? a.thetaX + (size_t)t * R * D * P : nullptr;
    double mean = 0.0, mo = 0.0;
    for (int k = 0; k < K; ++k) {
      const double z = a.Z[i + (size_t)a.n * (k + (size_t)K * t)];
      if (z != 0) {
        for (int mt = 0; mt <= M; ++mt) {
          const int r = k * (M + 1) + mt;
          double dot = 0.0;
          for (int w = 0; w < W; ++w) {
            double cf = th[(size_t)r * P + st + w];
            for (int dd = 0; dd < D; ++dd) cf += thx[((size_t)r * D + dd) * P + st + w] * a.X[i + (size_t)a.n * dd];
            dot += cf * bw[w];
          }
          if (mt == 0) { mean = mean + z * dot; mo = mo + z * dot; }
          else mean = mean + z * a.chi[i + (size_t)a.n * ((mt - 1) + (size_t)M * t)] * dot;
        }
      }
    }
    const size_t dst = (size_t)kept * o + (size_t)(t - a.first_kept) + (size_t)kept * l;
    mean_only[dst] = mo;
    const bfmmm::RngKey key = bfmmm::make_key(seed, 0u, (uint32_t)t, 0u);
    paths[dst] = mean + sqrt(a.sigma[t]) * bfmmm::rnorm(key, bfmmm::UPD_SAMPLE_PATH, (uint32_t)(o + l));
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

static int post_impl(const bfmmm_post_input* in, int32_t first_kept, double* llik, double* mean_pdf, double* mean_fit, double* mean_joint,
                     double* cpo = nullptr) {
  if (!in || !in->offsets || !in->y || (!in->B && !in->identity_basis) || !in->nu || !in->Phi || !in->Z || !in->chi || !in->sigma)
    return bfmmm_io_fail("bfmmm_post_pointwise: null argument");
  const int n = in->n, K = in->K, P = in->P, M = in->M, D = in->X ? in->D : 0, T = in->T;
  if (n < 1 || K < 1 || P < 1 || M < 0 || T < 1 || first_kept < 0 || first_kept >= T)
    return bfmmm_io_fail("bfmmm_post_pointwise: bad dimensions");
  if ((double)T * K * (M + 1) * P * std::max(D, 1) >= 5.0e8) return bfmmm_io_fail("bfmmm_post_pointwise: too many draws for one call (32-bit offsets into the parameter table): split the draws");
  if (P > 64 || K + M + 2 > WMAX || D > 8) return bfmmm_io_fail("bfmmm_post_pointwise: P <= 64, K + M <= 24 and D <= 8 in this build");
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
  // basis rows as windows of their non-zero columns (a B-spline row has degree + 1 of them): W = the widest window
  int W = 1;
  std::vector<int> first((size_t)n_obs, 0);
  if (in->identity_basis) {
    for (int i = 0; i < n; ++i) {
      if (in->offsets[i + 1] - in->offsets[i] != P) return bfmmm_io_fail("bfmmm_post_pointwise: identity basis needs P observations per row");
      for (int j = 0; j < P; ++j) first[(size_t)in->offsets[i] + j] = j;
    }
  }
  for (long long e = 0; e < n_obs && !in->identity_basis; ++e) {
    int f = -1, l = -1;
    for (int p = 0; p < P; ++p)
      if (in->B[(size_t)e * P + p] != 0.0) { if (f < 0) f = p; l = p; }
    first[(size_t)e] = std::max(f, 0);
    if (f >= 0) W = std::max(W, l - f + 1);
  }
  std::vector<double> Bc((size_t)n_obs * W);
  for (long long e = 0; e < n_obs; ++e) {
    const int st = std::min(first[(size_t)e], P - W);
    first[(size_t)e] = st;
    for (int w = 0; w < W; ++w) Bc[(size_t)e * W + w] = in->identity_basis ? 1.0 : in->B[(size_t)e * P + st + w];
  }
  // chunks of draws: enough workgroups for the 256 CUs, tiles stay whole
  int tchunk = T;
  while ((long long)n * ((T + tchunk - 1) / tchunk) < 2048 && tchunk > GMAX) tchunk = (tchunk + 1) / 2;
  tchunk = (tchunk + GMAX - 1) / GMAX * GMAX;
  const int NCH = (T + tchunk - 1) / tchunk;
  DevBufs db;
  PostDev a{};
  a.W = W; a.need_pdf = mean_pdf ? 1 : 0;
  a.n = n; a.K = K; a.P = P; a.M = M; a.D = D; a.T = T; a.first_kept = first_kept; a.tchunk = tchunk; a.n_obs = n_obs;
  std::vector<long long> off(in->offsets, in->offsets + n + 1);
  double *d_ll, *d_pdf, *d_fit, *d_joint;
  bool ok = db.put((long long**)&a.off, off.data(), off.size()) && db.put((double**)&a.y, in->y, (size_t)n_obs) &&
            db.put((double**)&a.Bc, Bc.data(), Bc.size()) && db.put((int**)&a.bstart, first.data(), first.size()) && db.put((double**)&a.theta, theta.data(), theta.size()) &&
            db.put((double**)&a.Z, in->Z, (size_t)n * K * T) && db.put((double**)&a.chi, in->chi, (size_t)n * M * T) &&
            db.put((double**)&a.sigma, in->sigma, (size_t)T) && db.put(&a.llpart, (const double*)nullptr, (size_t)n * T) &&
            db.put(&a.pdf_part, (const double*)nullptr, (size_t)NCH * n_obs) && db.put(&a.fit_part, (const double*)nullptr, (size_t)NCH * n_obs) &&
            db.put(&d_ll, (const double*)nullptr, (size_t)T) && db.put(&d_pdf, (const double*)nullptr, (size_t)n_obs) &&
            db.put(&d_fit, (const double*)nullptr, (size_t)n_obs) && db.put(&d_joint, (const double*)nullptr, (size_t)n);
  if (ok && D > 0) ok = db.put((double**)&a.X, in->X, (size_t)n * D) && db.put((double**)&a.thetaX, thetaX.data(), thetaX.size());
  if (!ok) { (void)hipGetLastError(); return bfmmm_io_fail("bfmmm_post_pointwise: device allocation or copy failed"); }
  if (cpo) {
    if (M < 1 || M > CPO_MMAX || K > KMAXP) return bfmmm_io_fail("bfmmm_post_cpo: 1 <= M <= 16 and K <= 16 in this build");
    long long ni_max = 1;
    for (int i = 0; i < n; ++i) ni_max = std::max<long long>(ni_max, in->offsets[i + 1] - in->offsets[i]);
    const int NIPX = (int)((ni_max + 3) & ~3LL), CSc = P | 1;
    const int per_draw = cpo_draw_doubles(M, CSc, NIPX);
    const int Gc = std::min(CPO_GMAX, CPO_LDS_BUDGET / per_draw);
    if (Gc < 1) return bfmmm_io_fail("bfmmm_post_cpo: (M + 1) x observations of a curve exceed the on-chip tile in this build");
    double *d_cll, *d_cpo;
    if (!db.put(&d_cll, (const double*)nullptr, (size_t)n * T) || !db.put(&d_cpo, (const double*)nullptr, (size_t)n))
      return bfmmm_io_fail("bfmmm_post_cpo: device allocation failed");
    const size_t lds_c = ((size_t)Gc * per_draw + CPO_GMAX * (KMAXP + 1) + BL_MAX + 8 + 8) * sizeof(double);
    (void)hipFuncSetAttribute((const void*)k_post_cpo, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_c);
    hipEvent_t c0, c1;
    (void)hipEventCreate(&c0); (void)hipEventCreate(&c1);
    (void)hipEventRecord(c0, 0);
    hipLaunchKernelGGL(k_post_cpo, dim3(n), dim3(256), lds_c, 0, a, d_cll, Gc, NIPX);
    hipLaunchKernelGGL(k_post_cpo_reduce, dim3((n + 255) / 256), dim3(256), 0, 0, a, d_cll, d_cpo);
    (void)hipEventRecord(c1, 0);
    const bool ran_c = hipDeviceSynchronize() == hipSuccess && hipGetLastError() == hipSuccess;
    if (ran_c) (void)hipEventElapsedTime(&g_last_kernel_ms, c0, c1);
    (void)hipEventDestroy(c0); (void)hipEventDestroy(c1);
    if (!ran_c || hipMemcpy(cpo, d_cpo, sizeof(double) * n, hipMemcpyDeviceToHost) != hipSuccess)
      return bfmmm_io_fail("bfmmm_post_cpo: kernel launch or copy back failed");
    return 0;
  }
  const size_t lds = ((size_t)GMAX * (P | 1) + (size_t)GMAX * WMAX + 2048 + 64 + BL_MAX + 8) * sizeof(double);
  (void)hipFuncSetAttribute((const void*)k_post_pointwise, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k_post_pointwise, dim3(n, NCH), dim3(256), lds, 0, a);
  const long long tot = std::max<long long>(std::max<long long>(T, n_obs), n);
  hipLaunchKernelGGL(k_post_reduce, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, 0, a, NCH, d_ll, d_pdf, d_fit, mean_joint ? d_joint : (double*)nullptr);
  (void)hipEventRecord(e1, 0);
  const bool ran = hipDeviceSynchronize() == hipSuccess && hipGetLastError() == hipSuccess;
  if (ran) (void)hipEventElapsedTime(&g_last_kernel_ms, e0, e1);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  if (!ran) return bfmmm_io_fail("bfmmm_post_pointwise: kernel launch failed");
  if ((llik && hipMemcpy(llik, d_ll, sizeof(double) * T, hipMemcpyDeviceToHost) != hipSuccess) ||
      (mean_pdf && hipMemcpy(mean_pdf, d_pdf, sizeof(double) * n_obs, hipMemcpyDeviceToHost) != hipSuccess) ||
      (mean_fit && hipMemcpy(mean_fit, d_fit, sizeof(double) * n_obs, hipMemcpyDeviceToHost) != hipSuccess) ||
      (mean_joint && hipMemcpy(mean_joint, d_joint, sizeof(double) * n, hipMemcpyDeviceToHost) != hipSuccess))
    return bfmmm_io_fail("bfmmm_post_pointwise: copy back failed");
  return 0;
}

extern "C" int bfmmm_post_pointwise(const bfmmm_post_input* in, int32_t first_kept, double* llik, double* mean_pdf, double* mean_fit) {
  return post_impl(in, first_kept, llik, mean_pdf, mean_fit, nullptr);
}

extern "C" int bfmmm_post_pointwise_joint(const bfmmm_post_input* in, int32_t first_kept, double* llik, double* mean_joint_pdf, double* mean_fit) {
  return post_impl(in, first_kept, llik, nullptr, mean_fit, mean_joint_pdf);
}

extern "C" int bfmmm_post_cpo(const bfmmm_post_input* in, int32_t first_kept, double* log_cpo) {
  if (!log_cpo) return bfmmm_io_fail("bfmmm_post_cpo: null argument");
  return post_impl(in, first_kept, nullptr, nullptr, nullptr, nullptr, log_cpo);
}

// device time of the last bfmmm_post_pointwise call's two kernels (HIP events on the launch stream), for measurement
extern "C" double bfmmm_post_last_kernel_ms(void) { return (double)g_last_kernel_ms; }

// bfmmm_post_sample_paths: paths / mean_only hold, for curve i, a kept x n_i block (draw fastest) at kept * offsets[i]
extern "C" int bfmmm_post_sample_paths(const bfmmm_post_input* in, int32_t first_kept, uint64_t seed, double* paths, double* mean_only) {
  if (!in || !in->offsets || !in->B || !in->nu || !in->Phi || !in->Z || !in->chi || !in->sigma || !paths || !mean_only)
    return bfmmm_io_fail("bfmmm_post_sample_paths: null argument");
  const int n = in->n, K = in->K, P = in->P, M = in->M, D = in->X ? in->D : 0, T = in->T;
  if (n < 1 || K < 1 || P < 1 || M < 0 || T < 1 || first_kept < 0 || first_kept >= T)
    return bfmmm_io_fail("bfmmm_post_sample_paths: bad dimensions");
  if ((double)T * K * (M + 1) * P * std::max(D, 1) >= 5.0e8) return bfmmm_io_fail("bfmmm_post_sample_paths: too many draws for one call: split the draws");
  const long long n_obs = in->offsets[n];
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return bfmmm_io_fail("bfmmm_post_sample_paths: no HIP device (the MI355X library has no CPU path)");
  if (hipSetDevice(in->device) != hipSuccess) return bfmmm_io_fail("bfmmm_post_sample_paths: cannot select the device");
  const int R = K * (M + 1), kept = T - first_kept;
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
            if (in->eta) thetaX[(((size_t)t * R + k * (M + 1)) * D + dd) * P + p] = in->eta[(size_t)t * P * D * K + p + (size_t)P * (dd + (size_t)D * k)];
            if (in->xi)
              for (int m = 0; m < M; ++m)
                thetaX[(((size_t)t * R + k * (M + 1) + m + 1) * D + dd) * P + p] = in->xi[((size_t)t * K + k) * P * D * M + p + (size_t)P * (dd + (size_t)D * m)];
          }
  }
  int W = 1;
  std::vector<int> first((size_t)n_obs, 0);
  for (long long e = 0; e < n_obs; ++e) {
    int f = -1, l = -1;
    for (int p = 0; p < P; ++p)
      if (in->B[(size_t)e * P + p] != 0.0) { if (f < 0) f = p; l = p; }
    first[(size_t)e] = std::max(f, 0);
    if (f >= 0) W = std::max(W, l - f + 1);
  }
  std::vector<double> Bc((size_t)n_obs * W);
  for (long long e = 0; e < n_obs; ++e) {
    const int st = std::min(first[(size_t)e], P - W);
    first[(size_t)e] = st;
    for (int w = 0; w < W; ++w) Bc[(size_t)e * W + w] = in->B[(size_t)e * P + st + w];
  }
  int tchunk = kept;
  while ((long long)n * ((kept + tchunk - 1) / tchunk) < 1024 && tchunk > 8) tchunk = (tchunk + 1) / 2;
  const int NCH = (kept + tchunk - 1) / tchunk;
  DevBufs db;
  PostDev a{};
  a.W = W; a.n = n; a.K = K; a.P = P; a.M = M; a.D = D; a.T = T; a.first_kept = first_kept; a.tchunk = tchunk; a.n_obs = n_obs;
  std::vector<long long> off(in->offsets, in->offsets + n + 1);
  double *d_paths, *d_mo;
  bool ok = db.put((long long**)&a.off, off.data(), off.size()) && db.put((double**)&a.Bc, Bc.data(), Bc.size()) &&
            db.put((int**)&a.bstart, first.data(), first.size()) && db.put((double**)&a.theta, theta.data(), theta.size()) &&
            db.put((double**)&a.Z, in->Z, (size_t)n * K * T) && db.put((double**)&a.chi, in->chi, (size_t)n * M * T) &&
            db.put((double**)&a.sigma, in->sigma, (size_t)T) && db.put(&d_paths, (const double*)nullptr, (size_t)kept * n_obs) &&
            db.put(&d_mo, (const double*)nullptr, (size_t)kept * n_obs);
  if (ok && D > 0) ok = db.put((double**)&a.X, in->X, (size_t)n * D) && db.put((double**)&a.thetaX, thetaX.data(), thetaX.size());
  if (!ok) { (void)hipGetLastError(); return bfmmm_io_fail("bfmmm_post_sample_paths: device allocation or copy failed"); }
  hipLaunchKernelGGL(k_post_paths, dim3(n, NCH), dim3(256), 0, 0, a, (unsigned long long)seed, kept, d_paths, d_mo);
  if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) return bfmmm_io_fail("bfmmm_post_sample_paths: kernel launch failed");
  if (hipMemcpy(paths, d_paths, sizeof(double) * (size_t)kept * n_obs, hipMemcpyDeviceToHost) != hipSuccess ||
      hipMemcpy(mean_only, d_mo, sizeof(double) * (size_t)kept * n_obs, hipMemcpyDeviceToHost) != hipSuccess)
    return bfmmm_io_fail("bfmmm_post_sample_paths: copy back failed");
  return 0;
}
