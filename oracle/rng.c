/*
 * oracle/rng.c -- TEST INFRASTRUCTURE (see oracle.h).
 *
 * Counter-based random variates for the CPU restatement.  The reference draws from R's
 * sequential stream (R::rgamma / R::rnorm / R::runif, RcppDist r_truncnorm, arma::mvnrnd via
 * RcppArmadillo's R-RNG adapter), which a parallel sampler cannot
 * reproduce, so both this oracle and the HIP path use the keyed convention documented in
 * oracle.h.  Distributions are those of the reference; variate-to-variate identity with R is
 * UNPINNED.
 */
#include "oracle.h"
#include <math.h>

/* Philox4x32-10 (Salmon et al., SC'11); pinned by the Random123 known-answer vectors in
 * tests/test_oracle_rng.py. */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
  uint32_t k0 = key[0], k1 = key[1];
  for (int round = 0; round < 10; ++round) {
    if (round > 0) { k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    uint32_t n0 = hi1 ^ c1 ^ k0;
    uint32_t n1 = lo1;
    uint32_t n2 = hi0 ^ c3 ^ k1;
    uint32_t n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static double u52(uint32_t hi, uint32_t lo) {
  /* 26 + 26 bits, centred: strictly inside (0,1) and exactly representable */
  double x = (double)(hi >> 6) * 67108864.0 + (double)(lo >> 6);
  return (x + 0.5) * (1.0 / 4503599627370496.0);
}

void orc_block(const orc_rng* r, uint32_t upd, uint32_t idx, uint32_t attempt, double* u0, double* u1) {
  uint32_t ctr[4], key[2], out[4];
  ctr[0] = idx;
  ctr[1] = (attempt & 0xFFFFu) | (r->tt_step << 16);
  ctr[2] = r->iter;
  ctr[3] = (r->chain << 8) | (upd & 0xFFu);
  key[0] = (uint32_t)(r->seed & 0xFFFFFFFFu);
  key[1] = (uint32_t)(r->seed >> 32);
  orc_philox4x32_10(ctr, key, out);
  *u0 = u52(out[0], out[1]);
  *u1 = u52(out[2], out[3]);
}

/* Wichura (1988) AS241 PPND16; the quantile function behind R's default rnorm()
 * (nmath/qnorm.c, nmath/snorm.c INVERSION). */
double orc_qnorm(double p) {
  double q = p - 0.5, r, val;
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

double orc_pnorm(double x) { return 0.5 * erfc(-x * 0.70710678118654752440); }

double orc_runif(const orc_rng* r, uint32_t upd, uint32_t idx) {
  double u0, u1;
  orc_block(r, upd, idx, 0, &u0, &u1);
  return u0;
}

double orc_rnorm(const orc_rng* r, uint32_t upd, uint32_t idx) {
  double u0, u1;
  orc_block(r, upd, idx, 0, &u0, &u1);
  return orc_qnorm(u0);
}

/* R::rnorm(mean, sd) as the update functions draw it (mean + sd * standard normal), with a test hook of the same kind as
 * orc_rgamma_hook below (tests/test_oracle_pit_data_blocks.py): while armed for update id `upd`, a draw records the (mean, sd)
 * the restatement asked for and returns the injected value -- the reference package's own saved draw -- instead. */
static struct { int armed; uint32_t upd; const double* inject; double* rec; int cap; } g_norm_hook;
void orc_rnorm_hook(int arm, uint32_t upd, const double* inject, double* rec_mean_sd, int cap) {
  g_norm_hook.armed = arm; g_norm_hook.upd = upd; g_norm_hook.inject = inject; g_norm_hook.rec = rec_mean_sd; g_norm_hook.cap = cap;
}
double orc_rnorm_ms(const orc_rng* r, uint32_t upd, uint32_t idx, double mean, double sd) {
  if (g_norm_hook.armed && upd == g_norm_hook.upd && (int)idx < g_norm_hook.cap) {
    g_norm_hook.rec[2 * idx] = mean; g_norm_hook.rec[2 * idx + 1] = sd;
    return g_norm_hook.inject[idx];
  }
  return mean + sd * orc_rnorm(r, upd, idx);
}

#define ORC_MAX_ATTEMPTS 256u
#define ORC_BOOST_ATTEMPT 0xFFFFu

/* R::rgamma(shape, scale) -- distribution only; algorithm = Marsaglia & Tsang (2000).
 * Call sites: Distributions.h:34, UpdateSigma.h:53, UpdateTau.h:31, UpdateGamma.h:29,
 * UpdateDelta.h:42,57. */
/* Test hook (tests/test_oracle_pit_traces.py): while armed for update id `upd`, every gamma draw of that update records
 * the (shape, scale) it was asked for and returns the caller's value for that index instead of sampling.  Feeding an
 * update the reference package's own saved draws makes the restatement walk the reference's sequence of conditionals,
 * so each recorded pair is the conditional law the restatement assigns to the reference's draw. */
static struct { int armed; uint32_t upd; const double* inject; double* rec; int cap; } g_gamma_hook;
void orc_rgamma_hook(int arm, uint32_t upd, const double* inject, double* rec_shape_scale, int cap) {
  g_gamma_hook.armed = arm; g_gamma_hook.upd = upd; g_gamma_hook.inject = inject;
  g_gamma_hook.rec = rec_shape_scale; g_gamma_hook.cap = cap;
}

double orc_rgamma(const orc_rng* r, uint32_t upd, uint32_t idx, double shape, double scale) {
  if (g_gamma_hook.armed && upd == g_gamma_hook.upd && (int)idx < g_gamma_hook.cap) {
    g_gamma_hook.rec[2 * idx] = shape; g_gamma_hook.rec[2 * idx + 1] = scale;
    return g_gamma_hook.inject[idx];
  }
  double a = shape, boost = 1.0, u0, u1;
  if (a < 1.0) {
    orc_block(r, upd, idx, ORC_BOOST_ATTEMPT, &u0, &u1);
    boost = pow(u0, 1.0 / a);
    a += 1.0;
  }
  const double d = a - 1.0 / 3.0;
  const double c = 1.0 / sqrt(9.0 * d);
  double g = d;
  for (uint32_t t = 0; t < ORC_MAX_ATTEMPTS; ++t) {
    orc_block(r, upd, idx, t, &u0, &u1);
    double x = orc_qnorm(u0);
    double v = 1.0 + c * x;
    if (v <= 0.0) continue;
    v = v * v * v;
    g = d * v;
    if (log(u1) < 0.5 * x * x + d - d * v + d * log(v)) break;
  }
  return g * boost * scale;
}

/* RcppDist d_truncnorm(x, mu, sigma, a, b, log=1) (third-party, absent from /root/reference;
 * RcppDist is unversioned in DESCRIPTION:12): normal density renormalised on [a,b]. */
double orc_dtruncnorm_log(double x, double mu, double sd, double lo, double hi) {
  if (x < lo || x > hi) return -INFINITY;
  double phi_hi = isinf(hi) ? (hi > 0 ? 1.0 : 0.0) : orc_pnorm((hi - mu) / sd);
  double phi_lo = isinf(lo) ? (lo > 0 ? 1.0 : 0.0) : orc_pnorm((lo - mu) / sd);
  double z = (x - mu) / sd;
  return -0.91893853320467274178 - log(sd) - 0.5 * z * z - log(phi_hi - phi_lo);
}

/* RcppDist r_truncnorm(mu, sigma, a, b) (call sites UpdateA.h:79,98; UpdateAlpha3.h:45):
 * rejection samplers of Robert (1995) as in the `truncnorm` C sources RcppDist wraps:
 * plain normal rejection while the standardised bound is below 0.45, exponential-tilt
 * rejection beyond.  One Philox block per attempt. */
double orc_rtruncnorm(const orc_rng* r, uint32_t upd, uint32_t idx, double mu, double sd, double lo, double hi) {
  const double t4 = 0.45;
  double u0, u1, x = 0.0;
  const double al = (lo - mu) / sd, be = (hi - mu) / sd;
  if (isinf(hi) && hi > 0 && !(isinf(lo))) {
    if (al < t4) {
      for (uint32_t t = 0; t < ORC_MAX_ATTEMPTS; ++t) {
        orc_block(r, upd, idx, t, &u0, &u1);
        x = orc_qnorm(u0);
        if (x >= al) break;
      }
      return mu + sd * x;
    }
    const double ainv = 1.0 / al;
    for (uint32_t t = 0; t < ORC_MAX_ATTEMPTS; ++t) {
      orc_block(r, upd, idx, t, &u0, &u1);
      x = -log(u0) * ainv;
      if (u1 <= exp(-0.5 * x * x)) break;
    }
    return mu + sd * (x + al);
  }
  if (isinf(lo) && lo < 0 && !(isinf(hi))) {
    /* mirror image */
    return -orc_rtruncnorm(r, upd, idx, -mu, sd, -hi, INFINITY);
  }
  /* two-sided or unbounded: plain rejection (not used on the hot path) */
  for (uint32_t t = 0; t < ORC_MAX_ATTEMPTS; ++t) {
    orc_block(r, upd, idx, t, &u0, &u1);
    x = orc_qnorm(u0);
    if (x >= al && x <= be) break;
  }
  return mu + sd * x;
}

/* Test hook: fills out[i] with draw i of the given kind.
 * kind 0 uniform, 1 normal, 2 gamma(p1=shape,p2=scale), 3 truncnorm(p1=mu,p2=sd, [0,inf)) */
void orc_test_fill(uint64_t seed, uint32_t chain, uint32_t iter, uint32_t upd, int kind,
                   double p1, double p2, int count, double* out) {
  orc_rng r = {seed, chain, iter, 0};
  for (int i = 0; i < count; ++i) {
    switch (kind) {
      case 0: out[i] = orc_runif(&r, upd, (uint32_t)i); break;
      case 1: out[i] = orc_rnorm(&r, upd, (uint32_t)i); break;
      case 2: out[i] = orc_rgamma(&r, upd, (uint32_t)i, p1, p2); break;
      default: out[i] = orc_rtruncnorm(&r, upd, (uint32_t)i, p1, p2, 0.0, INFINITY); break;
    }
  }
}
