// Data-independent part of the mixed-membership update of ONE curve (updateZ_PM, UpdateMixedMembership.h:131-185;
// Z_proposal_density :102-113; the prior part of lpdf_z :20-50; rdirichlet / calc_lB, Distributions.h:22-60):
// the Dirichlet(a_Z_PM Z_old) proposal through K keyed gamma draws, both proposal densities, the prior terms and the
// acceptance uniform.  None of it depends on nu / Phi / chi / sigma^2, so it can be evaluated ahead of the sweep that
// produces those (job_z_prepare, in spare workgroups of k_factor of the PREVIOUS iteration) or in place (k_curve_z).
// A group of GW >= 2K + 1 lanes cooperates on a curve: lane t K + k evaluates rejection attempt t of component k,
// lanes 0 .. K+1 the lgamma terms, lanes 0 .. 2K the logs.  Every lane of the group returns the full result, and the
// result does not depend on GW (the variate is the first accepted attempt of the sequence however many are tried side by side).
#pragma once
#include "model.hpp"
#include "rng.hpp"

namespace bfmmm {

struct ZProposal {
  double Znew[KMAX];
  double lo[KMAX], ln[KMAX];      // log Z_old,k / log Z_new,k
  double pr_old, pr_new;          // sum_k (alpha_3 pi_k - 1) log Z_k   (lpdf_z without the likelihood)
  double lpn, lpo;                // log q(new | old), log q(old | new)
  double log_uu;                  // log of the acceptance uniform
};
constexpr int ZPREP_FIELDS_PER_K = 3, ZPREP_SCALARS = 5;      // layout of Ctx::zprep: field f of curve i at [f * n + i]

// with_prior = false (job_z_prepare): the prior terms pr_old / pr_new are left at zero and pi / alpha_3 are not read -- the
// consumer forms them from lo / ln when it loads the proposal (z_proposal_load), so that the job does not depend on this
// iteration's pi / alpha_3 update and can run beside it.
template <int GW>
__device__ inline void z_proposal(const Ctx& c, const RngKey& key, int i, int gl, const double (&Zold)[KMAX],
                                  double alpha3, const double* pi, ZProposal& out, bool with_prior = true) {
  const int K = c.d.K;
  double a_old[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) a_old[k] = c.h.a_Z_PM * Zold[k];
  // ---- K gamma variates: attempts 0 .. ZTRY-1 of every component's rejection loop side by side; the variate is the
  //      first accepted attempt, exactly as in the sequential loop, which only continues if all were rejected
  constexpr int ZTRY = 4;
  const int ztry = min(ZTRY, GW / K);
  const int lt = gl / K, lk = gl - lt * K;
  double a_lane = 1.0;
#pragma unroll
  for (int k = 0; k < KMAX; ++k) if (k < K && lk == k) a_lane = a_old[k];
  double g_try = 0.0;
  int ok_try = 0;
  GammaSetup gs_ = {1.0, 1.0, 1.0};
  const uint32_t gidx = (uint32_t)(i * K + lk);
  if (lt < ztry) {
    gs_ = rgamma_setup(key, UPD_Z_PROP, gidx, (a_lane <= 0) ? 10.0 : a_lane);      // Distributions.h:24-28
    g_try = gs_.d;
    ok_try = rgamma_attempt(key, UPD_Z_PROP, gidx, (uint32_t)lt, gs_, g_try) ? 1 : 0;
  }
  double g = gs_.d;
  int done = 0;
#pragma unroll
  for (int tq = 0; tq < ZTRY; ++tq) {
    const int src = min(tq * K + lk, GW - 1);
    const double gt = __shfl(g_try, src, GW);
    const int okt = __shfl(ok_try, src, GW);
    if (tq < ztry && !done) { g = gt; done = okt; }
  }
  double mygam = 0.0, mylg = 0.0;
  if (gl < K) {
    for (uint32_t tq = (uint32_t)ztry; !done && tq < kMaxAttempts; ++tq) done = rgamma_attempt(key, UPD_Z_PROP, gidx, tq, gs_, g) ? 1 : 0;
    mygam = g * gs_.boost;
    mylg = lgamma_pos(a_lane);
  }
  double a_new[KMAX];
  double gsum = 0.0, lB_old = 0.0, sa_old = 0.0;
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    out.Znew[k] = (k < K) ? __shfl(mygam, k, GW) : 0.0;
    if (k < K) { gsum += out.Znew[k]; lB_old += __shfl(mylg, k, GW); sa_old += a_old[k]; }
  }
#pragma unroll
  for (int k = 0; k < KMAX; ++k) { out.Znew[k] = out.Znew[k] / gsum; a_new[k] = c.h.a_Z_PM * out.Znew[k]; }
  // lane k: lgamma(a_new_k); lane K: lgamma(sum a_old); lane K+1: lgamma(sum a_new) -- one straight-line sequence
  double sa_new = 0.0;
#pragma unroll
  for (int k = 0; k < KMAX; ++k) if (k < K) sa_new += a_new[k];
  double lgarg = 1.0;
#pragma unroll
  for (int k = 0; k < KMAX; ++k) if (k < K && gl == k) lgarg = a_new[k];
  if (gl == K) lgarg = sa_old;
  if (gl == K + 1) lgarg = sa_new;
  const double lgv = lgamma_pos(lgarg);
  double lB_new = 0.0;
#pragma unroll
  for (int k = 0; k < KMAX; ++k) if (k < K) lB_new += __shfl(lgv, k, GW);
  lB_old -= __shfl(lgv, K, GW);
  lB_new -= __shfl(lgv, K + 1, GW);
  // log Z_old,k on lane k, log Z_new,k on lane K + k, log of the acceptance uniform on lane 2 K: one log sequence
  double larg = 1.0;
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    if (k < K && gl == k) larg = Zold[k];
    if (k < K && gl == K + k) larg = out.Znew[k];
  }
  if (gl == 2 * K) larg = runif(key, UPD_Z_ACC, (uint32_t)i);
  const double lgz = log(larg);
  out.log_uu = __shfl(lgz, 2 * K, GW);
  double pr_old = 0.0, pr_new = 0.0, dn = 0.0, dold = 0.0;
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    out.lo[k] = 0.0; out.ln[k] = 0.0;
    if (k < K) {
      const double lo = __shfl(lgz, k, GW), ln = __shfl(lgz, K + k, GW);
      out.lo[k] = lo; out.ln[k] = ln;
      if (with_prior) {
        pr_old += (alpha3 * pi[k] - 1.0) * lo;
        pr_new += (alpha3 * pi[k] - 1.0) * ln;
      }
      dn += (a_old[k] - 1.0) * ln;       // density of proposing new from old
      dold += (a_new[k] - 1.0) * lo;     // density of proposing old from new
    }
  }
  out.pr_old = pr_old; out.pr_new = pr_new;
  out.lpn = dn - lB_old;
  out.lpo = dold - lB_new;
}

// Spare workgroups of k_factor (iteration t): the proposals of iteration t + 1 for 256 / GW curves each (GW lanes per curve:
// 8 when 2K + 1 <= 8, 16 when 2K + 1 <= 16, else 32 -- zprep_lanes).
// Z is final for iteration t by then (k_curve_z ran before k_factor), and the keyed RNG makes the variates a function of
// (seed, chain, t + 1) alone; the prior terms, which need iteration t's pi / alpha_3, are added by the consumer
// (z_proposal_load).  k_curve_z checks the tag before using a prepared proposal.
__host__ __device__ inline int zprep_lanes(int K) { return (2 * K + 1 <= 8) ? 8 : (2 * K + 1 <= 16) ? 16 : 32; }

template <int GW>
__device__ inline void job_z_prepare_gw(const Ctx& c, int wg) {
  const Dims& d = c.d;
  const int n = d.n, K = d.K;
  Dyn* dyn = c.dyn;
  const int gl = threadIdx.x & (GW - 1), i = wg * (256 / GW) + (int)(threadIdx.x / GW);
  if (wg == 0 && threadIdx.x == 0) {
    dyn->zprep_iter = dyn->iter + 1u; dyn->zprep_tt = 0u; dyn->zprep_chain = c.chain; dyn->zprep_seed = c.seed;
    dyn->zprep_valid = 1u;
  }
  if (i >= n) return;
  const RngKey key = make_key(c.seed, c.chain, dyn->iter + 1u, 0u);
  double Zold[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) Zold[k] = (k < K) ? c.Z[i + (size_t)n * k] : 0.0;
  ZProposal zp;
  z_proposal<GW>(c, key, i, gl, Zold, 0.0, nullptr, zp, false);
  double* o = c.zprep + i;
#pragma unroll
  for (int k = 0; k < KMAX; ++k)
    if (k < K && gl == k) { o[(size_t)n * k] = zp.Znew[k]; o[(size_t)n * (K + k)] = zp.lo[k]; o[(size_t)n * (2 * K + k)] = zp.ln[k]; }
  if (gl == GW - 1) {
    double* s = o + (size_t)n * 3 * K;
    s[0] = zp.pr_old; s[n] = zp.pr_new; s[(size_t)2 * n] = zp.lpn; s[(size_t)3 * n] = zp.lpo; s[(size_t)4 * n] = zp.log_uu;
  }
}

__host__ __device__ inline int zprep_curves_per_wg(int K) { return 256 / zprep_lanes(K); }
__device__ inline void job_z_prepare(const Ctx& c, int wg) {
  if (zprep_lanes(c.d.K) == 8) job_z_prepare_gw<8>(c, wg);
  else if (zprep_lanes(c.d.K) == 16) job_z_prepare_gw<16>(c, wg);
  else job_z_prepare_gw<32>(c, wg);
}

// Spare workgroups of k_factor: the n x M standard normals of THIS iteration's chi update (UpdateChi.h:57-59), which
// depend on the keyed RNG alone.  256 per workgroup.
__device__ inline void job_chi_normals(const Ctx& c, int wg) {
  const Dims& d = c.d;
  Dyn* dyn = c.dyn;
  if (wg == 0 && threadIdx.x == 0) {
    dyn->znorm_iter = dyn->iter; dyn->znorm_tt = dyn->tt_step; dyn->zprep_chain = c.chain; dyn->zprep_seed = c.seed;
    dyn->znorm_valid = 1u;
  }
  const int e = wg * 256 + threadIdx.x;
  if (e >= d.n * d.M) return;
  const int i = e / d.M, m = e - i * d.M;
  c.chi_norm[i + (size_t)d.n * m] = rnorm(make_key(c.seed, c.chain, dyn->iter, dyn->tt_step), UPD_CHI, (uint32_t)e);
}

// reads a prepared proposal back (every lane of the curve's group gets all of it) and adds the prior terms
// sum_k (alpha_3 pi_k - 1) log Z_k of lpdf_z (UpdateMixedMembership.h:20-50) for the current pi / alpha_3
// the loads (issued with the kernel's other global loads) ...
__device__ inline void z_proposal_fetch(const Ctx& c, int i, ZProposal& zp) {
  const int n = c.d.n, K = c.d.K;
  const double* o = c.zprep + i;
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {          // (clamped address + mask, not a load behind a branch)
    const int kc = min(k, K - 1);
    const double a = o[(size_t)n * kc], b = o[(size_t)n * (K + kc)], e = o[(size_t)n * (2 * K + kc)];
    zp.Znew[k] = (k < K) ? a : 0.0;
    zp.lo[k] = (k < K) ? b : 0.0;
    zp.ln[k] = (k < K) ? e : 0.0;
  }
  const double* s = o + (size_t)n * 3 * K;
  zp.lpn = s[(size_t)2 * n]; zp.lpo = s[(size_t)3 * n]; zp.log_uu = s[(size_t)4 * n];
}
// ... and the prior terms, from this iteration's pi / alpha_3 (pi: global or LDS)
__device__ inline void z_proposal_prior(const Ctx& c, ZProposal& zp, double alpha3, const double* pi) {
  const int K = c.d.K;
  double piv[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) piv[k] = pi[min(k, K - 1)];
  double pr_old = 0.0, pr_new = 0.0;
#pragma unroll
  for (int k = 0; k < KMAX; ++k)
    if (k < K) {
      pr_old += (alpha3 * piv[k] - 1.0) * zp.lo[k];
      pr_new += (alpha3 * piv[k] - 1.0) * zp.ln[k];
    }
  zp.pr_old = pr_old; zp.pr_new = pr_new;
}
__device__ inline void z_proposal_load(const Ctx& c, int i, ZProposal& zp, double alpha3, const double* pi) {
  z_proposal_fetch(c, i, zp);
  z_proposal_prior(c, zp, alpha3, pi);
}

}  // namespace bfmmm
