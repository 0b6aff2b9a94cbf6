// Counter-based random variates for the HIP sampler (host + device).
//
// The reference draws every variate from R's single sequential stream (R::rgamma / R::rnorm /
// R::runif at Distributions.h:34, UpdateChi.h:58, UpdateMixedMembership.h:168, ...; arma::mvnrnd
// through RcppArmadillo's R adapter), which cannot be consumed in parallel.  This sampler keys
// every variate by (seed, chain, iteration, update id, index, attempt) instead:
//   Philox4x32-10, key = 64-bit seed,
//   counter = (index, attempt | tt_step << 16, iteration, chain << 8 | update id).
// One block gives two 52-bit uniforms strictly inside (0,1).
//   normal    : AS241 inverse CDF of U0 (the quantile function behind R's default rnorm)
//   gamma     : Marsaglia-Tsang, one block per attempt
//   truncnorm : Robert (1995) rejection, one block per attempt
// Results are therefore independent of launch geometry and of the number of GPUs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

namespace bfmmm {

enum UpdId : uint32_t {
  UPD_Z_PROP = 1, UPD_Z_ACC = 2, UPD_PI_PROP = 3, UPD_PI_ACC = 4, UPD_A3_PROP = 5, UPD_A3_ACC = 6,
  UPD_PHI = 7, UPD_DELTA = 8, UPD_A_PROP = 9, UPD_A_ACC = 10, UPD_GAMMA = 11, UPD_NU = 12,
  UPD_TAU = 13, UPD_SIGMA = 14, UPD_CHI = 15, UPD_ETA = 16, UPD_TAU_ETA = 17, UPD_XI = 18,
  UPD_DELTA_XI = 19, UPD_AXI_PROP = 20, UPD_AXI_ACC = 21, UPD_GAMMA_XI = 22,
  UPD_INIT_NU = 30, UPD_INIT_CHI = 31, UPD_INIT_PI = 32, UPD_INIT_Z = 33, UPD_INIT_PHI = 34,
  UPD_INIT_ETA = 35, UPD_INIT_XI = 36, UPD_TT_ACC = 40,
  UPD_SAMPLE_PATH = 41      // posterior-predictive draws of FSamplePaths (kernels_post.hip)
};

struct RngKey {
  uint32_t k0, k1;    // seed
  uint32_t chain;
  uint32_t iter;
  uint32_t tt_step;
};

__host__ __device__ inline RngKey make_key(uint64_t seed, uint32_t chain, uint32_t iter, uint32_t tt_step = 0) {
  RngKey r;
  r.k0 = (uint32_t)(seed & 0xFFFFFFFFull);
  r.k1 = (uint32_t)(seed >> 32);
  r.chain = chain; r.iter = iter; r.tt_step = tt_step;
  return r;
}

__host__ __device__ inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
  for (int round = 0; round < 10; ++round) {
    if (round > 0) { k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__host__ __device__ inline double u52(uint32_t hi, uint32_t lo) {
  const double x = (double)(hi >> 6) * 67108864.0 + (double)(lo >> 6);
  return (x + 0.5) * (1.0 / 4503599627370496.0);
}

__host__ __device__ inline void rng_block(const RngKey& r, uint32_t upd, uint32_t idx, uint32_t attempt,
                                          double& u0, double& u1) {
  uint32_t o[4];
  philox4x32_10(idx, (attempt & 0xFFFFu) | (r.tt_step << 16), r.iter, (r.chain << 8) | (upd & 0xFFu),
                r.k0, r.k1, o);
  u0 = u52(o[0], o[1]);
  u1 = u52(o[2], o[3]);
}

// Wichura AS241 (PPND16)
__host__ __device__ inline double qnorm(double p) {
  const double q = p - 0.5;
  double r, val;
  if (fabs(q) <= 0.425) {
    r = 0.180625 - q * q;
    val = q * (((((((r * 2509.0809287301226727 + 33430.575583588128105) * r + 67265.770927008700853) * r
                   + 45921.953931549871457) * r + 13731.693765509461125) * r + 1971.5909503065514427) * r
                + 133.14166789178437745) * r + 3.387132872796366608)
          / (((((((r * 5226.495278852545925 + 28729.085735721942674) * r + 39307.89580009271061) * r
                 + 21213.794301586595867) * r + 5394.1960214247511077) * r + 687.1870074920579083) * r
              + 42.313330701600911252) * r + 1.0);
    return val;
  }
  r = (q < 0) ? p : (1.0 - p);
  r = sqrt(-log(r));
  if (r <= 5.0) {
    r -= 1.6;
    val = (((((((r * 7.7454501427834140764e-4 + 0.0227238449892691845833) * r + 0.24178072517745061177) * r
               + 1.27045825245236838258) * r + 3.64784832476320460504) * r + 5.7694972214606914055) * r
            + 4.6303378461565452959) * r + 1.42343711074968357734)
          / (((((((r * 1.05075007164441684324e-9 + 5.475938084995344946e-4) * r + 0.0151986665636164571966) * r
                 + 0.14810397642748007459) * r + 0.68976733498510000455) * r + 1.6763848301838038494) * r
              + 2.05319162663775882187) * r + 1.0);
  } else {
    r -= 5.0;
    val = (((((((r * 2.01033439929228813265e-7 + 2.71155556874348757815e-5) * r + 0.0012426609473880784386) * r
               + 0.026532189526576123093) * r + 0.29656057182850489123) * r + 1.7848265399172913358) * r
            + 5.4637849111641143699) * r + 6.6579046435011037772)
          / (((((((r * 2.04426310338993978564e-15 + 1.4215117583164458887e-7) * r + 1.8463183175100546818e-5) * r
                 + 7.868691311456132591e-4) * r + 0.0148753612908506148525) * r + 0.13692988092273580531) * r
              + 0.59983220655588793769) * r + 1.0);
  }
  return (q < 0.0) ? -val : val;
}

__host__ __device__ inline double pnorm_std(double x) { return 0.5 * erfc(-x * 0.70710678118654752440); }

__host__ __device__ inline double runif(const RngKey& r, uint32_t upd, uint32_t idx) {
  double u0, u1;
  rng_block(r, upd, idx, 0, u0, u1);
  return u0;
}

__host__ __device__ inline double rnorm(const RngKey& r, uint32_t upd, uint32_t idx) {
  double u0, u1;
  rng_block(r, upd, idx, 0, u0, u1);
  return qnorm(u0);
}

constexpr uint32_t kMaxAttempts = 256u;
constexpr uint32_t kBoostAttempt = 0xFFFFu;

// R::rgamma(shape, scale) in distribution (Marsaglia & Tsang 2000).  Attempt t of the rejection loop is a pure
// function of (key, update, index, t), so attempts can be evaluated in any order -- or side by side on idle
// lanes (k_curve_z) -- and the variate is the first accepted one.
struct GammaSetup { double d, c, boost; };

__host__ __device__ inline GammaSetup rgamma_setup(const RngKey& r, uint32_t upd, uint32_t idx, double shape) {
  double a = shape, boost = 1.0, u0, u1;
  if (a < 1.0) {
    rng_block(r, upd, idx, kBoostAttempt, u0, u1);
    boost = pow(u0, 1.0 / a);
    a += 1.0;
  }
  GammaSetup s;
  s.d = a - 1.0 / 3.0;
  s.c = 1.0 / sqrt(9.0 * s.d);
  s.boost = boost;
  return s;
}

// one attempt: returns true when it is accepted; g = d v either way (the loop's value after this attempt)
__host__ __device__ inline bool rgamma_attempt(const RngKey& r, uint32_t upd, uint32_t idx, uint32_t t, const GammaSetup& s,
                                               double& g) {
  double u0, u1;
  rng_block(r, upd, idx, t, u0, u1);
  const double x = qnorm(u0);
  double v = 1.0 + s.c * x;
  if (v <= 0.0) return false;            // (g keeps its previous value, as in the sequential loop)
  v = v * v * v;
  g = s.d * v;
  // Marsaglia-Tsang squeeze: a sufficient condition for the log test below, so the accept / reject
  // decisions (and therefore the variates) are unchanged; it only skips the two logarithms
  const double x2 = x * x;
  if (u1 < 1.0 - 0.0331 * (x2 * x2)) return true;
  return log(u1) < 0.5 * x2 + s.d - s.d * v + s.d * log(v);
}

__host__ __device__ inline double rgamma(const RngKey& r, uint32_t upd, uint32_t idx, double shape, double scale) {
  const GammaSetup s = rgamma_setup(r, upd, idx, shape);
  double g = s.d;
  for (uint32_t t = 0; t < kMaxAttempts; ++t)
    if (rgamma_attempt(r, upd, idx, t, s, g)) break;
  return g * s.boost * scale;
}

// log density of N(mu, sd) truncated to [lo, +inf): RcppDist d_truncnorm(x, mu, sd, lo, Inf, 1)
__host__ __device__ inline double dtruncnorm_lo_log(double x, double mu, double sd, double lo) {
  if (x < lo) return -INFINITY;
  const double z = (x - mu) / sd;
  return -0.91893853320467274178 - log(sd) - 0.5 * z * z - log(1.0 - pnorm_std((lo - mu) / sd));
}

// RcppDist r_truncnorm(mu, sd, lo, Inf)
__host__ __device__ inline double rtruncnorm_lo(const RngKey& r, uint32_t upd, uint32_t idx, double mu, double sd,
                                                double lo) {
  const double al = (lo - mu) / sd;
  double u0, u1, x = 0.0;
  if (al < 0.45) {
    for (uint32_t t = 0; t < kMaxAttempts; ++t) {
      rng_block(r, upd, idx, t, u0, u1);
      x = qnorm(u0);
      if (x >= al) break;
    }
    return mu + sd * x;
  }
  const double ainv = 1.0 / al;
  for (uint32_t t = 0; t < kMaxAttempts; ++t) {
    rng_block(r, upd, idx, t, u0, u1);
    x = -log(u0) * ainv;
    if (u1 <= exp(-0.5 * x * x)) break;
  }
  return mu + sd * (x + al);
}

// lgamma for x > 0 (x == 0 gives +inf): upward recurrence to x >= 8, then the Stirling series
//   lgamma(x) = (x - 1/2) log x - x + log(2 pi)/2 + sum_j B_2j / (2j (2j-1) x^(2j-1)),
// truncated after x^-13 (below 2e-15 at x = 8).  About 150 straight-line instructions against the few thousand,
// heavily divergent, of the device library's general-argument lgamma -- the Dirichlet proposal densities of the
// Z / pi updates evaluate it 2K + 2 times per curve and sweep.
__host__ __device__ inline double lgamma_pos(double x) {
  double prod = 1.0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const bool small_x = x < 8.0;
    prod = small_x ? prod * x : prod;
    x = small_x ? x + 1.0 : x;
  }
  const double r = 1.0 / x, r2 = r * r;
  double ser = 1.0 / 156.0;
  ser = ser * r2 - 691.0 / 360360.0;
  ser = ser * r2 + 1.0 / 1188.0;
  ser = ser * r2 - 1.0 / 1680.0;
  ser = ser * r2 + 1.0 / 1260.0;
  ser = ser * r2 - 1.0 / 360.0;
  ser = ser * r2 + 1.0 / 12.0;
  double lg = (x - 0.5) * log(x) - x + 0.91893853320467274178 + ser * r;
  if (prod != 1.0) lg -= log(prod);
  return lg;
}

// log of the multivariate Beta function, calc_lB of Distributions.h:51-60
__host__ __device__ inline double calc_lB(int K, const double* alpha) {
  double lb = 0.0, acc = 0.0;
  for (int k = 0; k < K; ++k) { lb += lgamma_pos(alpha[k]); acc += alpha[k]; }
  return lb - lgamma_pos(acc);
}

}  // namespace bfmmm
