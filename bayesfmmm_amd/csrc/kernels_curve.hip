// Per-curve kernels: Z (mixed-membership Metropolis-Hastings) and chi (scores), plus the
// per-curve residual sums that feed sigma^2 and the log-likelihood.
//
//   k_curve_z   <- updateZ_PM / lpdf_z / Z_proposal_density / rdirichlet
//                  (UpdateMixedMembership.h:20-50, 102-113, 131-185; Distributions.h:22-60)
//   k_curve_chi <- updateChi (UpdateChi.h:19-64) and the residual pass of calcLikelihood
//                  (CalculateLikelihood.h:19-44)
//
// Mapping: curves are independent, so a group of LPC lanes (32 or 64, one lane per basis
// function) owns one curve; a 256-thread workgroup holds 256/LPC curves.  The banded Gram matrix
// G_i sits in registers (2*BW+1 values per lane).  The handful of P-vectors a curve needs
// (u_k = nu_k + sum_m chi_im phi_km, G u_k, ...) are staged in a per-group LDS tile; banded
// matrix-vector products read neighbours from that tile and every quadratic form u'Gv is ONE
// lane's serial dot product over the tile (K + K(K+1)/2 <= 27 of them for Z, M(M+1)/2 + M + 2
// for chi), so there are no cross-lane reduction chains.  The chi update is Gauss-Seidel in m
// inside a curve; with A = U'GU and b = U'(s - G c) precomputed it becomes a scalar recursion.
// All sums have a fixed order: results do not depend on the launch geometry.
#include "model.hpp"
#include "rng.hpp"
#include "scalar_jobs.hpp"
#include "z_proposal.hpp"
#include "lds_dot.hpp"

#include <algorithm>

namespace bfmmm {

constexpr int MMAX = 16;   // eigenfunctions supported by the per-curve tiles

template <int BW, int LPC>
struct Curve {
  double g[BW + 1];    // g[d]  = G[p][p+d]
  double gl[BW + 1];   // gl[d] = G[p-d][p]
  double s, yy;
  // unconditional loads (clamped indices) so that the 2*BW+3 loads are issued back to back
  __device__ inline void load(const double* __restrict__ rec, int P, int LG, int p) {
    const bool act = p < P;
    const int pc = min(p, P - 1);
    double vg[BW + 1], vl[BW + 1];
#pragma unroll
    for (int d = 0; d <= BW; ++d) {
      vg[d] = rec[d * P + pc];                       // zero-padded where pc + d >= P
      vl[d] = rec[d * P + max(pc - d, 0)];
    }
    const double vs = rec[LG + pc];
    yy = rec[LG + P];
#pragma unroll
    for (int d = 0; d <= BW; ++d) {
      g[d] = act ? vg[d] : 0.0;
      gl[d] = (act && d > 0 && p - d >= 0) ? vl[d] : 0.0;
    }
    s = act ? vs : 0.0;
  }
  // (G u)[p] with u staged in an LDS row padded by BW zeros on both sides (row points at element 0)
  __device__ inline double matvec(const double* row, int p) const {
    double v = g[0] * lds_ld(row + p);             // (single ds_read_b64s: lds_dot.hpp)
#pragma unroll
    for (int d = 1; d <= BW; ++d) v += g[d] * lds_ld(row + p + d) + gl[d] * lds_ld(row + p - d);
    return v;
  }
};

template <int BW, int LPC>
struct Tile {
  // Rows of LPC entries with BW zeros on either side (the band products read p - BW .. p + BW), the pads SHARED between
  // neighbours: [BW zeros][row 0: LPC][gap >= BW zeros][row 1: LPC][gap] ...  The row stride is ODD: lane a of a group reads row a
  // in the dot products, and with an even stride the rows fall on the same LDS banks (stride 64 at BW = 0: every row on the same
  // banks).  Shared pads took the stride from LPC + 2 BW + 1 to LPC + BW: with the s row gone as well (k_curve_chi) the config-2
  // workgroup needs 40 KB instead of 46.8 -- a fourth workgroup per CU.
  static constexpr int STR = ((LPC + BW) % 2 == 0) ? LPC + BW + 1 : LPC + BW;
  __host__ __device__ static constexpr int doubles(int rows) { return BW + rows * STR; }
  double* base;        // the first pad
  __device__ inline double* row(int r) const { return base + BW + r * STR; }
  // (one store instruction per LPC pad entries: lane e takes entry e of the BW + rows (STR - LPC) pads.  A lane per gap column
  //  and a loop over the rows was `rows` store instructions executed by three lanes -- an LDS store is 6 cycles of the pipe
  //  whatever its mask, and the batches are bound by that pipe: 14 stores, 6 % of k_curve_chi's LDS time, now 2)
  __device__ inline void zero_pads(int rows, int lp) const {
    constexpr int GAP = STR - LPC;
    const int total = BW + rows * GAP;
    for (int e = lp; e < total; e += LPC) {
      const int g = e - BW, r = g / GAP, q = g - r * GAP;
      base[(e < BW) ? e : (BW + r * STR + LPC + q)] = 0.0;
    }
  }
};

// the value lane m of a curve group holds, to every lane of the group (LPC = 32: two groups to a wave -- both source lanes are
// read and each half of the wave keeps its own; v_readlane does not depend on EXEC)
template <int LPC>
__device__ inline double group_bcast(double v, int m) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const int lo0 = __builtin_amdgcn_readlane(lo, m), hi0 = __builtin_amdgcn_readlane(hi, m);
  if constexpr (LPC == 64) return __hiloint2double(hi0, lo0);
  const int lo1 = __builtin_amdgcn_readlane(lo, 32 + m), hi1 = __builtin_amdgcn_readlane(hi, 32 + m);
  const bool upper = (threadIdx.x & 32) != 0;
  return __hiloint2double(upper ? hi1 : hi0, upper ? lo1 : lo0);
}

__device__ inline double dotP(const double* a, const double* b, int P) {
  double s = 0.0;
  for (int p = 0; p < P; ++p) s += a[p] * b[p];
  return s;
}

// the same sum over a whole tile row (entries P .. LPC-1 of the rows are zero): a compile-time trip count, so the
// 2 LPC LDS reads are issued back to back instead of one dependent read per loop trip
template <int LPC>
__device__ inline double dotL(const double* a, const double* b) {
  double s = 0.0;
#pragma unroll
  for (int p = 0; p < LPC; ++p) s += a[p] * b[p];
  return s;
}

// The quadratic forms of the Z update into res: q < K: a_q = u_q's; then the pairs (k, k2), k <= k2: u_k'G u_k2 -- K + K (K + 1) / 2
// of them (9 at K = 3).  One lane per form left two thirds of a 32-lane group idle through a full 2 LPC-read dot, and the
// batches are bound by the LDS pipe: two lanes per form where they fit (lane 2 t + h takes entries [h LPC / 2, (h + 1) LPC / 2) of
// form t, the halves meet by DPP), one lane per form otherwise (K >= 5 at 32 lanes).  Shared by k_curve_z, its lean form and the
// fused update in k_curve_chi: the three produce the same bits.
template <int BW, int LPC>
__device__ inline void z_forms(const Tile<BW, LPC>& tU, const Tile<BW, LPC>& tG, const double* srow, int K, double* res, int lp) {
  const int ntask = K + K * (K + 1) / 2;
  auto rows = [&](int q, const double*& ra, const double*& rb) {
    ra = tU.row(min(q, K - 1));
    rb = srow;
    if (q >= K) {
      int a = 0, rem = q - K;
      while (rem >= K - a) { rem -= K - a; ++a; }
      ra = tU.row(a); rb = tG.row(a + rem);
    }
  };
  if (2 * ntask <= LPC) {
    const int t = lp >> 1, h = lp & 1;
    if (t < ntask) {
      const double *ra, *rb;
      rows(t, ra, rb);
      const double part = dot_half_lds<LPC>(ra + h * (LPC / 2), rb + h * (LPC / 2));
      const double s = dpp_add<0xB1>(part);            // + the other half (lane ^ 1)
      if (h == 0) res[t] = s;
    }
  } else {
    for (int q = lp; q < ntask; q += LPC) {
      const double *ra, *rb;
      rows(q, ra, rb);
      res[q] = dot_lds<LPC>(ra, rb);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Z update.  do_update == 0 only recomputes the partial sums of log Z (used when pi / alpha_3 are
// sampled with Z held fixed).
// LDS per group: U[K], GU[K], S (1 row), chi (MMAX), res (32)
// ------------------------------------------------------------------------------------------------
#ifdef BFMMM_TIMELINE
constexpr int ZTN = 8192;      // workgroups traced (index: chain * gridDim.x + blockIdx.x)
__device__ unsigned long long g_ztrace[3 * ZTN];
__device__ unsigned long long g_zphase[8 * ZTN];
void fetch_ztrace(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ztrace), sizeof(unsigned long long) * 3 * ZTN); }
void fetch_zphase(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_zphase), sizeof(unsigned long long) * 8 * ZTN); }
#endif

// KT: compile-time bound on K (4 or KMAX).  The per-cluster arrays below are unrolled to KT, not KMAX: with K <= 4 the
// quadratic-form registers (Q alone is KMAX^2 doubles) shrink enough for a third and fourth workgroup per CU.
// LEAN: the launch is one whose proposals were prepared by the previous iteration's k_factor for certain (bfmmm_capi.hip runs
// every Z update but the first of a run that way); the in-place evaluation of the proposal -- keyed gamma rejection loops,
// lgamma, logs -- is then not compiled in, which takes the kernel from 167 to 100 VGPRs and from three to four or five
// workgroups per CU.  Should the tag not match after all, the run fails loudly (status bit 2).
// A LEAN launch closes iteration t and opens t + 1: workgroup 0 runs iteration t's scalar job (delta, A, gamma, tau; it
// advances the iteration counters at its end), so the curve workgroups take the iteration and the chain slot from the
// snapshot the sweep left (iter_hyper + 1, slot_hyper + 1), as the fused update in k_curve_chi does; curve block b sits at
// grid index 8 + b (workgroups 1-7 idle), on the XCD of k_curve_chi's block b.
// KEX: K == KT exactly, a compile-time constant (see k_curve_chi's exact instances): 16.6 -> 14.4 us for the 8-chain batch of config 5.
template <int BW, int LPC, bool COV, int KT, bool LEAN = false, bool KEX = false>
__global__ __launch_bounds__(256, LEAN ? 4 : ((KT <= 4 && BW <= 5) ? 3 : 2)) void k_curve_z(Ctx c0, int do_update) {
  Ctx cx = chain_view(c0);           // chain blockIdx.z of the batch
  if constexpr (KEX) { cx.d.K = KT; cx.d.A = KT * cx.d.MD; }
  const Ctx& c = cx;
  TIMELINE(c, 0);
  if (LEAN && blockIdx.x < 8) {
    if (blockIdx.x == 0) job_hyper(c);
    return;
  }
  const int blk = LEAN ? (int)blockIdx.x - 8 : (int)blockIdx.x;
#ifdef BFMMM_TIMELINE
  if (threadIdx.x == 0 && blockIdx.x < 1024) {
    unsigned id, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    g_ztrace[3 * blockIdx.x] = wall_clock64();
    g_ztrace[3 * blockIdx.x + 1] = ((unsigned long long)(id & 0xf) << 32) | hw;
  }
  struct EndTrace { __device__ ~EndTrace() { if (threadIdx.x == 0 && blockIdx.x < 1024) g_ztrace[3 * blockIdx.x + 2] = wall_clock64(); } } et_;
#endif
#ifdef BFMMM_TIMELINE
  unsigned long long zt[10]; int zi = 0;
#define ZT() do { zt[zi++] = clock64(); } while (0)
#else
#define ZT() do { } while (0)
#endif
  ZT();
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int GPB = 256 / LPC;
  using T = Tile<BW, LPC>;
  const Dims& d = c.d;
  const int n = d.n, K = d.K, P = d.P, M = d.M, MD = d.MD;
  const int D = COV ? d.D : 0;   // the covariate code is compiled only into the COV instantiation
  const int nth = K * (M + 1) * P;
  double* sDyn = smem;                               // LEAN: the head of Dyn and pi (model.hpp: dyn_head_fetch)
  double* sTh = sDyn + DYN_LDS_DOUBLES;
  double* sThX = sTh + nth;                          // D > 0: thetaX, K*(M+1)*D*P
  double* sLog = sThX + (size_t)nth * D;             // GPB*KMAX
  double* sYp = sLog + GPB * KMAX;                   // GPB
  const int grp = threadIdx.x / LPC, lp = threadIdx.x % LPC;
  // multivariate model: G_i = I, so G u = u and the GU tile is the U tile (one tile less: a third workgroup per CU)
  const int TW = d.mv ? 1 : 2;
  const int per_group = T::doubles(TW * K + 3) + MMAX + 48;      // (48: the K + K (K + 1) / 2 <= 44 quadratic forms)
  double* gbase = sYp + GPB + (size_t)grp * per_group;
  T tU{gbase}, tG{gbase + (TW - 1) * K * T::STR}, tS{gbase + TW * K * T::STR};     // tS rows: 0 = s, 1 = o, 2 = G o
  double* sChi = gbase + T::doubles(TW * K + 3);
  double* sRes = sChi + MMAX;
  const int i = blk * GPB + grp;
  const bool valid = i < n;
  const bool act = lp < P;
  const Dyn* dyn = c.dyn;
  // (LEAN: one load per wave, handed round through LDS -- the head is needed only behind the barrier; otherwise one batch of
  //  loads, not a trip per field: model.hpp)
  uint32_t dyn_w = 0u;
  DynHead dh;
  if constexpr (LEAN) dyn_w = dyn_head_fetch(dyn);
  else dh = dyn_head(dyn);
  // ---- all global loads are requested up front.  Z first: the proposal phase below needs nothing else, and loads
  //      retire in issue order, so it can start while the record, theta and chi are still on their way ----
  double Zold[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) Zold[k] = (k < K) ? c.Z[min(i, n - 1) + (size_t)n * k] : 0.0;
  double thv[4];                                     // first 1024 entries of theta (the rest, if any, follows below)
#pragma unroll
  for (int u = 0; u < 4; ++u) thv[u] = c.theta[min((int)threadIdx.x + 256 * u, nth - 1)];
  Curve<BW, LPC> cv;
  cv.load(c.rec + (size_t)min(i, n - 1) * d.LREC, P, d.LG, lp);
  const double chi_l = (MD > 1 && lp < M) ? c.chi[min(i, n - 1) + (size_t)n * min(lp, M - 1)] : 0.0;
  // ---- proposal phase (UpdateMixedMembership.h:131-150): everything of the update that does not depend on the data
  //      (z_proposal.hpp).  Normally it was prepared during the previous iteration's k_factor; otherwise (first
  //      iteration of a run, tempered sweeps, changed state) it is evaluated here, while the loads are in flight ----
  ZProposal zp;
  if constexpr (LEAN) {
    z_proposal_fetch(c, min(i, n - 1), zp);          // (the tag is checked, the prior terms added behind the barrier)
    dyn_head_stage(sDyn, dyn_w);
  } else if (valid && do_update) {
    const bool pre = (dh.zprep_valid != 0u) & (dh.zprep_iter == dh.iter) & (dh.zprep_tt == dh.tt_step) &
                     (dh.zprep_chain == c.chain) & (dh.zprep_seed == c.seed);
    if (pre) z_proposal_load(c, i, zp, dh.alpha3, dyn->pi);
    else z_proposal<LPC>(c, make_key(c.seed, c.chain, dh.iter, dh.tt_step), i, lp, Zold, dh.alpha3, dyn->pi, zp);
  }
  // ---- now the staged data: theta to LDS, the curve's s and chi to its tile ----
#pragma unroll
  for (int u = 0; u < 4; ++u) { const int idx = (int)threadIdx.x + 256 * u; if (256 * u < nth) { if (idx < nth) sTh[idx] = thv[u]; } }      // (uniform test first: no store instruction for an empty trip)
  if (nth > 1024) copy_to_lds<4>(sTh + 1024, c.theta + 1024, nth - 1024, threadIdx.x, 256);
  if (D > 0) copy_to_lds<4>(sThX, c.thetaX, nth * D, threadIdx.x, 256);
  if (valid) {
    tU.zero_pads(TW * K + 3, lp);
    if (lp <= M) sChi[lp] = chi_l;                   // [M] = 0: pad of the 2-unrolled loops
    tS.row(0)[lp] = cv.s;
  }
  ZT();
  __syncthreads();
  ZT();
  if constexpr (LEAN) {
    dh = dyn_head_lds(sDyn);
    if (valid && do_update) {
      const bool pre = (dh.zprep_valid != 0u) & (dh.zprep_iter == dh.iter_hyper + 1u) & (dh.zprep_tt == dh.tt_step) &
                       (dh.zprep_chain == c.chain) & (dh.zprep_seed == c.seed);
      if (pre) z_proposal_prior(c, zp, dh.alpha3, dyn_pi_lds(sDyn));
      else if (lp == 0) atomicOr(&c.dyn->status, 2u);
    }
  }
  const uint32_t slot_cur = LEAN ? dh.slot_hyper + 1u : dh.slot;
  const double beta = dh.beta;
  const double inv_2s2 = 1.0 / (2.0 * dh.sigma2);
  // covariate adjustment: the curve sees theta_r + sum_d x_id thetaX_{r,d}.  The combination is folded into the sums
  // below (u_k = sum_r coef_r theta_r + sum_d x_d sum_r coef_r thetaX_{r,d}) instead of materialising the K(M+1)
  // effective rows per curve in LDS: that tile cost 50 KB per workgroup and left one workgroup per CU.
  double xv[8];
#pragma unroll
  for (int dd = 0; dd < 8; ++dd) xv[dd] = (valid && dd < D) ? c.X[i + (size_t)n * dd] : 0.0;
  double logz_mine = 0.0;
  if (valid) {
    double ucov[KMAX];                 // covariate part of u_k (D > 0)
    double uk[KMAX];
    {
      // u_k = nu_k + sum_m chi_im phi_km for all k at once: the m loop runs two eigenfunctions per trip with the
      // 2K + 2 LDS reads of a trip in flight together (sChi[M] = 0 pads an odd M; row indices stay inside direction k)
      double vb[KMAX];                 // the part without covariates
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        vb[k] = (k < K && act) ? lds_ld(sTh + (size_t)k * (M + 1) * P + lp) : 0.0;
        ucov[k] = 0.0;
        if (D > 0 && k < K && act) {
          const double* xb = sThX + (size_t)k * (M + 1) * D * P + lp;
#pragma unroll
          for (int dd = 0; dd < 8; ++dd) ucov[k] += xv[dd] * xb[min(dd, D - 1) * P];      // (no branch: x_d = 0 beyond D, the address is clamped)
        }
      }
      if (MD > 1 && act)
        for (int m = 0; m < M; m += 2) {
          const double c0 = sChi[m], c1 = sChi[m + 1];
          const int r0 = (m + 1), r1 = min(m + 2, M);
#pragma unroll
          for (int k = 0; k < KT; ++k)
            if (k < K) {
              const double* tb = sTh + (size_t)k * (M + 1) * P + lp;
              vb[k] += c0 * lds_ld(tb + r0 * P) + c1 * lds_ld(tb + r1 * P);
              if (D > 0) {
                const double* xb = sThX + (size_t)k * (M + 1) * D * P + lp;
                double e0 = 0.0, e1 = 0.0;
#pragma unroll
                for (int dd = 0; dd < 8; ++dd) {
                  const int dc = min(dd, D - 1);
                  e0 += xv[dd] * xb[(r0 * D + dc) * P]; e1 += xv[dd] * xb[(r1 * D + dc) * P];
                }
                ucov[k] += c0 * e0 + c1 * e1;
              }
            }
        }
#pragma unroll
      for (int k = 0; k < KT; ++k) uk[k] = vb[k] + ucov[k];
#pragma unroll
      for (int k = 0; k < KT; ++k)
        if (k < K) tU.row(k)[lp] = uk[k];
    }
    ZT();
    __builtin_amdgcn_wave_barrier();
    if (!d.mv)
      for (int k = 0; k < K; ++k) tG.row(k)[lp] = cv.matvec(tU.row(k), lp);
    ZT();
    __builtin_amdgcn_wave_barrier();
    z_forms<BW, LPC>(tU, tG, tS.row(0), K, sRes, lp);      // a_q = u_q's, Q_{k,k2} = u_k'G u_k2
    __builtin_amdgcn_wave_barrier();
    double av[KMAX], Q[KMAX][KMAX];
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      av[k] = (k < K) ? sRes[k] : 0.0;
#pragma unroll
      for (int k2 = 0; k2 < KT; ++k2)
        if (k2 >= k) Q[k][k2] = (k2 < K) ? sRes[K + tri_index(K, k, k2)] : 0.0;
    }
    ZT();
    double Zfin[KMAX];
#pragma unroll
    for (int k = 0; k < KT; ++k) Zfin[k] = Zold[k];
    bool took_new = false;
    if (do_update) {
      // quadratic form of the residual sum of squares in Z
      // q = yy + sum_k Z_k (-2 a_k + sum_k2 Z_k2 Q_k,k2): one short chain per k (a dependent double-precision operation issues
      // every ~16 clocks; the flat sum over (k, k2) was one chain of K (K + 1) of them).  The fused update in k_curve_chi forms
      // the same sums in the same order.
      double qo = 0.0, qn = 0.0;
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        if (k < K) {
          double to = -2.0 * av[k], tn = -2.0 * av[k];
#pragma unroll
          for (int k2 = 0; k2 < KT; ++k2) {
            if (k2 < K) {
              const double qq = (k2 >= k) ? Q[k][k2] : Q[k2][k];
              to += Zold[k2] * qq;
              tn += zp.Znew[k2] * qq;
            }
          }
          qo += Zold[k] * to;
          qn += zp.Znew[k] * tn;
        }
      }
      const double q_old = cv.yy + qo, q_new = cv.yy + qn;
      const double z_lpdf = zp.pr_old - beta * (q_old * inv_2s2);
      const double z_new_lpdf = zp.pr_new - beta * (q_new * inv_2s2);
      double acceptance = z_new_lpdf - z_lpdf + zp.lpo - zp.lpn;
#pragma unroll
      for (int k = 0; k < KT; ++k)
        if (k < K && Zold[k] <= 0) acceptance = 1;                        // UpdateMixedMembership.h:170-174
      if (zp.log_uu < acceptance) {
        took_new = true;
#pragma unroll
        for (int k = 0; k < KT; ++k) Zfin[k] = zp.Znew[k];
      }
      double* zslot = c.c_Z + (size_t)slot_cur * n * K;
#pragma unroll
      for (int k = 0; k < KT; ++k)
        if (k < K && lp == k) { c.Z[i + (size_t)n * k] = Zfin[k]; zslot[i + (size_t)n * k] = Zfin[k]; }
    }
    if (do_update) {                                      // log Z_ik of the kept state: both were needed by the proposal densities
#pragma unroll
      for (int k = 0; k < KT; ++k)
        if (k < K && lp == k) logz_mine = took_new ? zp.ln[k] : zp.lo[k];
    } else {
      double zarg = 1.0;
#pragma unroll
      for (int k = 0; k < KT; ++k)
        if (k < K && lp == k) zarg = Zfin[k];
      logz_mine = log(zarg);                              // one log sequence: lane k evaluates log Z_ik
    }
    if (D > 0) {
      // offset seen by the Phi / nu block: o = sum_k Z_k ucov_k;  s~ = s - G o;  yy~ = yy - 2 o's + o'G o
      double o = 0.0;
#pragma unroll
      for (int k = 0; k < KT; ++k)
        if (k < K) o += Zfin[k] * ucov[k];
      tS.row(1)[lp] = o;
      __builtin_amdgcn_wave_barrier();
      const double Go = cv.matvec(tS.row(1), lp);
      tS.row(2)[lp] = Go;
      if (act) c.stil[(size_t)i * P + lp] = cv.s - Go;
      __builtin_amdgcn_wave_barrier();
      if (lp == 0) sYp[grp] = cv.yy - 2.0 * dotP(tS.row(1), tS.row(0), P) + dotP(tS.row(1), tS.row(2), P);
    }
  } else if (lp == 0) {
    sYp[grp] = 0.0;
  }
  ZT();
#ifdef BFMMM_TIMELINE
  if (blockIdx.x == 7 && threadIdx.x == 0) for (int x = 0; x + 1 < zi; ++x) c.dyn->stamps[32 + x] = zt[x + 1] - zt[x];
  if (blockIdx.x < 512 && threadIdx.x == 0) for (int x = 0; x + 1 < zi && x < 8; ++x) g_zphase[8 * blockIdx.x + x] = zt[x + 1] - zt[x];
#endif
  // block partial of sum_i log Z_ik, fixed order over the groups of this block
  if (lp < KMAX) sLog[grp * KMAX + lp] = (lp < K) ? logz_mine : 0.0;
  __syncthreads();
  if (threadIdx.x < K) {
    double acc = 0.0;
    for (int g = 0; g < GPB; ++g) acc += sLog[g * KMAX + threadIdx.x];
    c.logz_part[(size_t)blk * K + threadIdx.x] = acc;
  }
  if (D > 0 && threadIdx.x == 0) {
    double acc = 0.0;
    for (int g = 0; g < GPB; ++g) acc += sYp[g];
    c.yyp_part[blk] = acc;
  }
}

// ------------------------------------------------------------------------------------------------
// chi update + per-curve residual sum of squares.  do_update == 0 computes only the residuals.
// LDS per group: U[M], GU[M], C0 (1), D = s - G c0 (1), chi (M), z (M), res (M(M+1)/2 + M + 2).  Sized by the
// actual M so that three workgroups fit a CU and the extra scalar-job workgroup never waits for a free slot.
// ------------------------------------------------------------------------------------------------
// SMALL: K <= 4 and M <= 8 -- the per-cluster and per-eigenfunction loops are unrolled to those bounds instead of KMAX = 6 and
// MMAX = 16 (the Gauss-Seidel recursion and the residual update are MMAX^2 guarded terms otherwise): 18.7 -> 17.4 us at config 2
// KX, MX > 0: an EXACT-shape instance -- K and M are compile-time constants (BFMMM_CHI_EXACT below lists the pairs built).  The
// kernel is bound by the latency of its dependent LDS reads, two waves to a SIMD; with run-time K and M every "k < K" / "m < M"
// guard inside the unrolled loops is a uniform branch that ends a batch of LDS reads (each batch then costs its own wait), and the
// packed-triangle indices of the Gauss-Seidel recursion are computed addresses.  With the two constants the loops unroll into
// straight-line code whose reads are issued together: 16.0 -> 12.7 us at config 2 (K 3, M 6); K alone or M alone gives 1.3 - 1.5 us.
template <int BW, int LPC, bool COV, bool SMALL, int KX = 0, int MX = 0>
__global__ __launch_bounds__(256) void k_curve_chi(Ctx c0, int mode) {
  static_assert((KX > 0) == (MX > 0) && KX <= KMAX && MX <= MMAX, "exact instances fix both K and M");
  constexpr int KT = KX ? KX : (SMALL ? 4 : KMAX), MT = MX ? MX : (SMALL ? 8 : MMAX);
  Ctx cx = chain_view(c0);           // chain blockIdx.z of the batch
  if constexpr (KX > 0) {            // (the launcher checked K == KX, M == MX; every use below, helpers included, sees constants)
    cx.d.K = KX; cx.d.M = MX; cx.d.MD = (cx.d.MD > 1) ? MX + 1 : 1; cx.d.A = KX * cx.d.MD;
  }
  const Ctx& c = cx;
  TIMELINE(c, 5);
  // mode 0: nothing per curve (only the scalar job), 1: residual sums only, 2: chi update + residual sums
  if (blockIdx.x == 0) {     // one extra workgroup (dispatched first): delta, A, gamma, tau -- hidden under the per-curve work
#ifdef BFMMM_ABLATE
    if ((mode >> 8) & 4) return;
#endif
    // (single chain: 513 workgroups on 512 slots -- the two curve workgroups that share this one's CU ended 1.5 us after all the
    //  others, and with them the kernel; with Ctx::defer_hyper the job itself rides the next k_pair_gram, whose grid has idle
    //  extra workgroups, and only the counters advance here)
    if (c.defer_hyper) {
      job_hyper_counters(c);
      if (threadIdx.x == 0) c.dyn->hyper_pending = 1u;
    } else {
      job_hyper(c);
    }
    TSTAMP(c, 15);
    return;
  }
  // bit 4 of mode: after its chi update / residual pass a curve group also runs the Z update of the NEXT iteration
  // (fused k_curve_z, D == 0 only): the record, theta and Z are already on chip, so the next iteration loses the
  // load phase of k_curve_z (60 % of that kernel) and one launch boundary.
  const bool fuse_z = (mode & 16) != 0;
#ifdef BFMMM_ABLATE
  // diagnostic build (tools/timeline.py build --extra -DBFMMM_ABLATE): BFMMM_ABLATE=<bits> cuts parts of the kernel out to time the
  // rest -- 1: the curve workgroups return at entry; 2: they return behind the first barrier (loads + staging only);
  // 4: no scalar job (workgroup 0 returns); 8: they return behind the quadratic forms; 16: behind the Gauss-Seidel phase
  const int abl = (mode >> 8) & 255;
  if ((abl & 1) && blockIdx.x >= 8) return;
#endif
  mode &= 15;
  if (mode == 0) return;
  // Workgroups 1-7 are idle: they keep curve block b at grid index 8 + b, i.e. on the XCD that runs block b of
  // k_curve_z (workgroups go to the XCDs round-robin), so the Z / chi / record lines the two kernels hand each other
  // iteration after iteration stay in that XCD's L2.
#ifdef BFMMM_TIMELINE
  // per-workgroup trace (start, XCC / HW id, end) of the curve workgroups and of the scalar-job workgroup, in k_curve_z's array
  // (traced: the launches that fuse the next Z update, i.e. all but the last iteration of a run)
  const int zti_ = fuse_z ? (int)(blockIdx.z * gridDim.x + blockIdx.x) : ZTN;
  if (threadIdx.x == 0 && zti_ < ZTN) {
    unsigned id, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    g_ztrace[3 * zti_] = wall_clock64();
    g_ztrace[3 * zti_ + 1] = ((unsigned long long)(id & 0xf) << 32) | hw;
    if (blockIdx.x == 0) c.dyn->stamps[39] = ((c.dyn->stamps[39] << 4) | (id & 0xf)) & 0xFFFFFFFFFULL;      // history of block 0's XCC, one nibble per launch
  }
  struct EndTraceC { int i; __device__ ~EndTraceC() { if (threadIdx.x == 0 && i < ZTN) g_ztrace[3 * i + 2] = wall_clock64(); } } etc_{zti_};
#endif
  if (blockIdx.x < 8) return;
  const int blk = blockIdx.x - 8;
  const int do_update = (mode == 2);
#ifdef BFMMM_TIMELINE
  unsigned long long ct_[10]; int ci_ = 0;
#define CT() do { ct_[ci_++] = clock64(); } while (0)
#else
#define CT() do { } while (0)
#endif
  CT();
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int GPB = 256 / LPC;
  using T = Tile<BW, LPC>;
  const Dims& d = c.d;
  const int n = d.n, K = d.K, P = d.P, M = d.M, MD = d.MD;
  const int D = COV ? d.D : 0;
  const int nth = K * (M + 1) * P;
  const int Mu = (do_update && MD > 1) ? M : 0;     // number of u_m vectors needed
  const int ntask = Mu * (Mu + 1) / 2 + Mu + 1;
  double* sDyn = smem;                              // the head of Dyn and pi (model.hpp: dyn_head_fetch)
  double* sTh = sDyn + DYN_LDS_DOUBLES;
  double* sThX = sTh + nth;
  double* sRss = sThX + (size_t)nth * D;            // GPB
  double* sLog = sRss + GPB;                        // GPB*KMAX (fused Z: block partial of sum_i log Z_ik)
  const int grp = threadIdx.x / LPC, lp = threadIdx.x % LPC;
  const int RT = max(M, K + 1);                      // rows of the U / GU tiles (the fused Z update needs K of them, and one of U's for s)
  const int nres = max(M * (M + 1) / 2 + M + 1, K + K * (K + 1) / 2);
  const int TW = d.mv ? 1 : 2;                       // multivariate model: G u = u, the GU tile is the U tile
  const int per_group = T::doubles(TW * RT + 2) + 3 * M + 1 + nres;
  double* gbase = sLog + GPB * KMAX + (size_t)grp * per_group;
  T tU{gbase}, tG{gbase + (TW - 1) * RT * T::STR}, tX{gbase + TW * RT * T::STR};   // tX rows: 0 = c0, 1 = s - G c0
  double* sChi = gbase + T::doubles(TW * RT + 2);     // M + 1 entries ([M] = 0: pad of the fused Z update's 2-unrolled loop)
  double* sWq = sChi + M + 1;                        // c1_m, then c3_m: the folded constants of the chi update
  double* sRes = sWq + 2 * M;
  const int i = blk * GPB + grp;
  const bool valid = i < n;
  const int ic = min(i, n - 1);
  const bool act = lp < P;
  const Dyn* dyn = c.dyn;
  // ---- ALL global loads of the workgroup are requested up front, into registers, before the first of them is waited for (as in
  //      k_curve_z), and WITHOUT branches (clamped addresses, values masked afterwards): a load behind a uniform branch makes the
  //      compiler wait for everything outstanding (vmcnt(0)) at the first use behind the join.  The small operands first and the
  //      record -- nine tenths of the bytes -- last: loads retire in issue order, and u_m / c0 below need only the small ones. ----
  const uint32_t dyn_w = dyn_head_fetch(dyn);        // (one load per wave, handed round through LDS: model.hpp)
  double thv[4];                                     // first 1024 entries of theta (the rest, if any, follows below)
#pragma unroll
  for (int u = 0; u < 4; ++u) thv[u] = c.theta[min((int)threadIdx.x + 256 * u, nth - 1)];
  double chi_l = c.chi[ic + (size_t)n * min(lp, M - 1)];
  if (!(MD > 1 && lp < M)) chi_l = 0.0;
  double Zi[KMAX];
#pragma unroll
  for (int k = 0; k < KT; ++k) { Zi[k] = c.Z[ic + (size_t)n * min(k, K - 1)]; if (k >= K) Zi[k] = 0.0; }
  double xv[8];                                      // covariates of the curve (adjustment folded into the sums below, see k_curve_z)
#pragma unroll
  for (int dd = 0; dd < 8; ++dd) xv[dd] = (dd < D) ? c.X[ic + (size_t)n * dd] : 0.0;
  Curve<BW, LPC> cv;
  cv.load(c.rec + (size_t)ic * d.LREC, P, d.LG, lp);
  double zn_pre = c.chi_norm[ic + (size_t)n * min(lp, M - 1)];       // (used only if its tag in Dyn matches)
  ZProposal zp;
  if (fuse_z) z_proposal_fetch(c, ic, zp);                               // likewise
  // ---- now the staged data: the Dyn head and theta to LDS, chi to the curve's tile ----
  dyn_head_stage(sDyn, dyn_w);
#ifdef BFMMM_TIMELINE
  CT();                                              // (the Dyn head has arrived)
#endif
#pragma unroll
  for (int u = 0; u < 4; ++u) { const int idx = (int)threadIdx.x + 256 * u; if (256 * u < nth) { if (idx < nth) sTh[idx] = thv[u]; } }      // (uniform test first: no store instruction for an empty trip)
#ifdef BFMMM_TIMELINE
  CT();                                              // (theta has arrived)
#endif
  if (nth > 1024) copy_to_lds<4>(sTh + 1024, c.theta + 1024, nth - 1024, threadIdx.x, 256);
  if (D > 0) copy_to_lds<4>(sThX, c.thetaX, nth * D, threadIdx.x, 256);
  if (valid) {
    tU.zero_pads(TW * RT + 2, lp);
    if (lp <= M) sChi[lp] = chi_l;
  }
  // (not __syncthreads(): its fence waits for every outstanding load -- the record included.  Only the LDS stores above are published.)
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  CT();
#ifdef BFMMM_ABLATE
  if (abl & 2) return;
#endif
  const DynHead dh = dyn_head_lds(sDyn);
  const double sigma2 = dh.sigma2, beta = dh.beta;
  // (reciprocals formed once: the Gauss-Seidel recursion below had a division -- ~10 dependent instructions -- on each of its M
  //  sequential steps)
  const double inv_s2 = 1.0 / sigma2, inv_2s2 = 1.0 / (2.0 * sigma2);
  if (fuse_z) z_proposal_prior(c, zp, dh.alpha3, dyn_pi_lds(sDyn));
  double rss = 0.0;
  double logz_mine = 0.0;
  if (valid) {
    // fused Z update: the proposal prepared by this iteration's k_factor (requested above) is used at the end
    const uint32_t it_next = dh.iter_hyper + 1u;
    bool zpre = false;
    if (fuse_z) {
      zpre = (dh.zprep_valid != 0u) & (dh.zprep_iter == it_next) & (dh.zprep_tt == dh.tt_step) &
             (dh.zprep_chain == c.chain) & (dh.zprep_seed == c.seed);
    }
    // u_m = sum_k Z_k phi_km ;  c0 = sum_k Z_k nu_k + sum_m chi_m u_m
    // row r = (k, mt) of the parameters as this curve sees it: theta_r + sum_d x_d thetaX_{r,d}
    // (no branch on k < K: the row index is clamped and Z_k = 0 there, so the LDS reads of a row are all in flight together)
    const int lpc = min(lp, P - 1);
    auto zrow = [&](int mt) {
      double v = 0.0;
#pragma unroll
      for (int k = 0; k < KT; ++k)
        {
          const int r = min(k, K - 1) * (M + 1) + mt;
          double e = lds_ld(sTh + r * P + lpc);
          if (D > 0) {
#pragma unroll
            for (int dd = 0; dd < 8; ++dd)
              e += xv[dd] * sThX[(r * D + min(dd, D - 1)) * P + lpc];      // (no branch: x_d = 0 beyond D)
          }
          v += Zi[k] * e;
        }
      return v;
    };
    double cf = zrow(0);
    if (!act) cf = 0.0;
    if (MD > 1) {
      for (int m = 0; m < M; ++m) {
        double um = zrow(m + 1);
        if (!act) um = 0.0;
        if (m < Mu) tU.row(m)[lp] = um;
        cf += sChi[m] * um;
      }
    }
    tX.row(0)[lp] = cf;
    // c0's: the one quadratic form against s of this kernel's chi part -- a sum over the lanes instead of a task over an s row
    // (the row is 6 % of the workgroup's LDS: the fourth workgroup per CU).  (first use of the record)
    const double c0s = group_sum<LPC>(cf * cv.s);
    __builtin_amdgcn_wave_barrier();
    CT();
    const double g0 = cv.matvec(tX.row(0), lp);
    tX.row(1)[lp] = cv.s - g0;
    if (!d.mv)
      for (int m = 0; m < Mu; ++m) tG.row(m)[lp] = cv.matvec(tU.row(m), lp);
    __builtin_amdgcn_wave_barrier();
    // tasks: [0, Mu(Mu+1)/2): A_{m,m2} = u_m' G u_m2 (m <= m2);  then Mu of b_m = u_m'(s - G c0);  then  c0'(s - G c0)
    const int nA = Mu * (Mu + 1) / 2;
    for (int q = lp; q < ntask; q += LPC) {
      const double* ra = tX.row(0);
      const double* rb = tX.row(1);
      if (q < nA) {
        int a = 0, rem = q;
        while (rem >= Mu - a) { rem -= Mu - a; ++a; }
        ra = tU.row(a); rb = tG.row(a + rem);
      } else if (q < nA + Mu) {
        ra = tU.row(q - nA);
      }
      sRes[q] = dot_lds<LPC>(ra, rb);
    }
    if (Mu > 0) {
      // the normals were drawn by spare workgroups of this iteration's k_factor (job_chi_normals); in place otherwise
      const bool pre = (dh.znorm_valid != 0u) & (dh.znorm_iter == dh.iter_hyper) & (dh.znorm_tt == dh.tt_step) &
                       (dh.zprep_chain == c.chain) & (dh.zprep_seed == c.seed);
      if (lp < M && !pre) zn_pre = rnorm(make_key(c.seed, c.chain, dh.iter_hyper, dh.tt_step), UPD_CHI, (uint32_t)(i * M + lp));
    }
    __builtin_amdgcn_wave_barrier();
    CT();
#ifdef BFMMM_ABLATE
    if (abl & 8) return;
#endif
    // rss at c0:  yy - 2 c0's + c0'G c0 = yy - c0's - c0'(s - G c0)
    rss = cv.yy - c0s - sRes[nA + Mu];
    if (Mu > 0) {
      // chi_m <- W_m w + sqrt(W_m) z_m,  W_m = 1 / (1 + A_mm beta / sigma^2),  w = (r_m + chi_m A_mm) beta / sigma^2,
      // r_m = b_m - sum_{m2 < m} A_{m2,m} dl_m2  (UpdateChi.h:40-59 in Gram form).  LANE m OWNS DIRECTION m: it holds b_m, column m
      // of A, chi_m, z_m and folds everything but r_m into two constants,  dl_m = chi_new - chi_old = c1_m r_m + c3_m  with
      // c1 = W beta / sigma^2,  c3 = c1 chi_old A_mm + sqrt(W) z - chi_old  (one division and one rsqrt sequence per wave, side by
      // side).  Step m of the recursion: every lane evaluates c1 r + c3 on its own r, lane m's value is the new dl_m and is handed
      // to the group through v_readlane (two groups to a wave: both lanes are read, each half keeps its own), and the lanes
      // behind m take A_{m,j} dl_m off their r_j.  The sums are those of the scalar recursion every lane used to run on broadcast
      // LDS reads, term for term; the wave issues M + 1 LDS reads for recursion and residual together where it issued
      // M (M + 1) / 2 + 4 M + M + 1 (the batches are bound by the LDS pipe), and two dependent operations per step remain.
      const int ml = min(lp, M - 1);
      const double bl = lds_ld(sRes + nA + ml);
      double col[MT];                                  // A_{m2, ml}
#pragma unroll
      for (int m2 = 0; m2 < MT; ++m2) {
        const int mc = min(m2, M - 1);
        const double v = lds_ld(sRes + tri_index(M, min(ml, mc), max(ml, mc)));
        col[m2] = (m2 < M) ? v : 0.0;
      }
      double W0l = 0.0;
#pragma unroll
      for (int m2 = 0; m2 < MT; ++m2) W0l = (ml == m2) ? col[m2] : W0l;
      const double den = 1.0 + ((W0l * beta) * inv_s2);
      const double Wl = 1.0 / den, sq = rsqrt(den);       // (both from den: the two sequences run side by side)
      const double c1 = Wl * (beta * inv_s2);
      const double c3 = (c1 * chi_l) * W0l + (sq * zn_pre - chi_l);
      double dl[MT];
      double* cslot = c.c_chi + (size_t)dh.slot_hyper * n * M;
      double r = bl, own = 0.0;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        dl[m] = 0.0;
        if (m < M) {
          const double cand = c1 * r + c3;
          own = (lp == m) ? cand : own;
          dl[m] = group_bcast<LPC>(cand, m);
          r -= col[m] * dl[m];
        }
      }
      const double mine = chi_l + own;
      if (lp < M) { c.chi[i + (size_t)n * lp] = mine; cslot[i + (size_t)n * lp] = mine; sChi[lp] = mine; }
      // rss(c0 + sum_m dl_m u_m) = rss0 + sum_m dl_m (-2 b_m + sum_m2 dl_m2 A_{m,m2}): lane m forms the bracket of its m from the
      // column it holds, the M products meet in a DPP sum over the 16-lane row; only lane 0's rss is used below
      {
        double tm = -2.0 * bl;
#pragma unroll
        for (int m2 = 0; m2 < MT; ++m2)
          if (m2 < M) tm += dl[m2] * col[m2];
        rss += row16_sum((lp < M) ? own * tm : 0.0);
      }
      if (D > 0 && act) {      // the eta / Xi steps start from the updated coefficient c_i and g_i = G_i c_i
        double cfin = tX.row(0)[lp], gfin = cv.s - tX.row(1)[lp];
#pragma unroll
        for (int m = 0; m < MT; ++m)
          if (m < M) { cfin += dl[m] * tU.row(m)[lp]; gfin += dl[m] * tG.row(m)[lp]; }
        c.cfull[(size_t)i * P + lp] = cfin;
        c.gfull[(size_t)i * P + lp] = gfin;
      }
    } else if (D > 0 && act) {
      c.cfull[(size_t)i * P + lp] = tX.row(0)[lp];
      c.gfull[(size_t)i * P + lp] = cv.s - tX.row(1)[lp];
    }
    CT();
#ifdef BFMMM_ABLATE
    if (abl & 16) return;
#endif
    if (fuse_z && !zpre && lp == 0) atomicOr(&c.dyn->status, 2u);      // cannot happen in a fused run (see below); reported by bfmmm_run
    if (fuse_z && zpre) {
      // ---- updateZ_PM of iteration it_next for this curve (UpdateMixedMembership.h:131-185), as in k_curve_z:
      //      theta (sTh), the record (cv), s (tX row 2) and the new chi (sChi) are on chip; the U / GU tiles are free ----
      // (the data-independent half of the update was prepared by this iteration's k_factor, job_z_prepare: it runs
      //  whenever Z is updated -- a superset of the conditions the host fuses under: U_Z, no covariates, untempered)
      __builtin_amdgcn_wave_barrier();
      double uk[KMAX];
#pragma unroll
      for (int k = 0; k < KT; ++k) uk[k] = (k < K && act) ? lds_ld(sTh + (size_t)k * (M + 1) * P + lp) : 0.0;
      if (MD > 1 && act)
        for (int m = 0; m < M; m += 2) {
          const double c0 = sChi[m], c1 = sChi[m + 1];
          const int r0 = (m + 1), r1 = min(m + 2, M);
#pragma unroll
          for (int k = 0; k < KT; ++k)
            if (k < K) {
              const double* th = sTh + (size_t)k * (M + 1) * P + lp;
              uk[k] += c0 * lds_ld(th + r0 * P) + c1 * lds_ld(th + r1 * P);
            }
        }
#pragma unroll
      for (int k = 0; k < KT; ++k)
        if (k < K) tU.row(k)[lp] = uk[k];
      tU.row(K)[lp] = cv.s;                            // (row K of the U tile is free now: RT >= K + 1)
      __builtin_amdgcn_wave_barrier();
      if (!d.mv)
        for (int k = 0; k < K; ++k) tG.row(k)[lp] = cv.matvec(tU.row(k), lp);
      __builtin_amdgcn_wave_barrier();
      z_forms<BW, LPC>(tU, tG, tU.row(K), K, sRes, lp);      // a_q = u_q's, Q_{k,k2} = u_k'G u_k2
      __builtin_amdgcn_wave_barrier();
      double qo = 0.0, qn = 0.0;                       // (as in k_curve_z: one short chain per k)
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        if (k < K) {
          const double avk = lds_ld(sRes + k);
          double to = -2.0 * avk, tn = -2.0 * avk;
#pragma unroll
          for (int k2 = 0; k2 < KT; ++k2) {
            if (k2 < K) {
              const double qq = lds_ld(sRes + K + tri_index(K, min(k, k2), max(k, k2)));
              to += Zi[k2] * qq;
              tn += zp.Znew[k2] * qq;
            }
          }
          qo += Zi[k] * to;
          qn += zp.Znew[k] * tn;
        }
      }
      const double q_old = cv.yy + qo, q_new = cv.yy + qn;
      const double z_lpdf = zp.pr_old - beta * (q_old * inv_2s2);
      const double z_new_lpdf = zp.pr_new - beta * (q_new * inv_2s2);
      double acceptance = z_new_lpdf - z_lpdf + zp.lpo - zp.lpn;
#pragma unroll
      for (int k = 0; k < KT; ++k)
        if (k < K && Zi[k] <= 0) acceptance = 1;                        // UpdateMixedMembership.h:170-174
      const bool took_new = zp.log_uu < acceptance;
      double* zslot = c.c_Z + (size_t)(dh.slot_hyper + 1u) * n * K;
#pragma unroll
      for (int k = 0; k < KT; ++k)
        if (k < K && lp == k) {
          const double zf = took_new ? zp.Znew[k] : Zi[k];
          c.Z[i + (size_t)n * k] = zf;
          zslot[i + (size_t)n * k] = zf;
          logz_mine = took_new ? zp.ln[k] : zp.lo[k];
        }
    }
  }
  CT();
#ifdef BFMMM_TIMELINE
  if (threadIdx.x == 0 && zti_ < ZTN) for (int x = 0; x + 1 < ci_ && x < 8; ++x) g_zphase[8 * zti_ + x] = ct_[x + 1] - ct_[x];
#endif
  if (lp == 0) sRss[grp] = rss;
  if (fuse_z && lp < KMAX) sLog[grp * KMAX + lp] = (lp < K) ? logz_mine : 0.0;
  __syncthreads();
  if (threadIdx.x == 0) {
    double acc = 0.0;
    for (int g = 0; g < GPB; ++g) acc += sRss[g];
    c.rss_part[blk] = acc;
  }
  if (fuse_z && threadIdx.x >= 64 && threadIdx.x < 64 + K) {      // block partial of sum_i log Z_ik, as k_curve_z leaves it
    const int k = threadIdx.x - 64;
    double acc = 0.0;
    for (int g = 0; g < GPB; ++g) acc += sLog[g * KMAX + k];
    c.logz_part[(size_t)blk * K + k] = acc;
  }
}

// ---- host launchers -------------------------------------------------------------------------
// The (K, M) pairs k_curve_chi is also built for exactly, for the cubic-spline functional model (BW 3; with covariates only at
// P <= 32) and the multivariate model (BW 0).  Any other shape runs the general instances.
#define BFMMM_CHI_EXACT(X)                                                                                     \
  X(2, 1) X(2, 2) X(2, 3) X(2, 4) X(2, 5) X(2, 6) X(2, 7) X(2, 8)                                               \
  X(3, 1) X(3, 2) X(3, 3) X(3, 4) X(3, 5) X(3, 6) X(3, 7) X(3, 8)                                               \
  X(4, 1) X(4, 2) X(4, 3) X(4, 4) X(4, 5) X(4, 6) X(4, 7) X(4, 8)
template <int BW, int L, bool CV>
constexpr bool chi_exact_built() { return (BW == 3 && (!CV || L == 32)) || (BW == 0 && !CV); }

// The 120 exact instances of k_curve_chi are compiled in a translation unit of their own (kernels_curve_exact.hip includes this
// file with BFMMM_CURVE_EXACT_TU defined and gets the kernels plus the two functions below; this file then holds everything else),
// so that the two halves build side by side.  The diagnostic timeline build keeps one unit: its trace arrays are device globals.
inline int ablate_bits() {
#ifdef BFMMM_ABLATE
  static const int v = getenv("BFMMM_ABLATE") ? (atoi(getenv("BFMMM_ABLATE")) << 8) : 0;
  return v;
#else
  return 0;
#endif
}
template <int BW, int L, bool CV>
bool launch_chi_exact(const Ctx& c, int nblk, size_t lds, hipStream_t st, int mode);      // (built combinations only: chi_exact_built)
template <int BW, int L, bool CV>
void prepare_chi_exact_kernels();

#if defined(BFMMM_CURVE_EXACT_TU) || defined(BFMMM_TIMELINE)
template <int BW, int L, bool CV>
bool launch_chi_exact(const Ctx& c, int nblk, size_t lds, hipStream_t st, int mode) {
  static_assert(chi_exact_built<BW, L, CV>(), "not on the list");
#define X(k, m)                                                                                                \
  if (c.d.K == k && c.d.M == m) {                                                                              \
    hipLaunchKernelGGL((k_curve_chi<BW, L, CV, true, k, m>), dim3(nblk + 8, 1, c.nch), dim3(256), lds, st, c, mode | ablate_bits());     \
    return true;                                                                                               \
  }
  BFMMM_CHI_EXACT(X)
#undef X
  return false;
}
template <int BW, int L, bool CV>
void prepare_chi_exact_kernels() {
#define X(k, m) set_max_lds((const void*)k_curve_chi<BW, L, CV, true, k, m>);
  BFMMM_CHI_EXACT(X)
#undef X
}
#define BFMMM_CHI_EXACT_COMBOS(Y) Y(3, 32, false) Y(3, 64, false) Y(3, 32, true) Y(0, 32, false) Y(0, 64, false)
#define Y(bw, l, cv)                                                                                           \
  static_assert(chi_exact_built<bw, l, cv>(), "BFMMM_CHI_EXACT_COMBOS lists what chi_exact_built admits");      \
  template bool launch_chi_exact<bw, l, cv>(const Ctx&, int, size_t, hipStream_t, int);                         \
  template void prepare_chi_exact_kernels<bw, l, cv>();
BFMMM_CHI_EXACT_COMBOS(Y)
#undef Y
#endif

#ifndef BFMMM_CURVE_EXACT_TU
template <int BW, int L, bool CV>
static bool try_chi_exact(const Ctx& c, int nblk, size_t lds, hipStream_t st, int mode) {
  if (!g_exact_instances) return false;
  if constexpr (chi_exact_built<BW, L, CV>()) return launch_chi_exact<BW, L, CV>(c, nblk, lds, st, mode);
  else return false;
}

// k_curve_z with K exact (2, 3, 4), same models
template <int BW, int L, bool CV>
static bool launch_z_exact(const Ctx& c, int nblk, size_t lds, hipStream_t st, int do_update) {
  if (!g_exact_instances) return false;
  if constexpr (chi_exact_built<BW, L, CV>()) {
    const bool lean = (do_update & 2) != 0 && !CV;       // (a lean launch is one without covariates: see LAUNCH_CURVE)
#define X(k)                                                                                                   \
    if (c.d.K == k) {                                                                                          \
      if constexpr (!CV) {                                                                                     \
        if (lean) { hipLaunchKernelGGL((k_curve_z<BW, L, false, k, true, true>), dim3(nblk + 8, 1, c.nch), dim3(256), lds, st, c, do_update & 1); return true; }  \
      }                                                                                                        \
      hipLaunchKernelGGL((k_curve_z<BW, L, CV, k, false, true>), dim3(nblk, 1, c.nch), dim3(256), lds, st, c, do_update & 1);          \
      return true;                                                                                             \
    }
    X(2) X(3) X(4)
#undef X
  }
  return false;
}

template <int BW, int L, bool CV>
static void prepare_chi_exact() {
  if constexpr (chi_exact_built<BW, L, CV>()) {
#define X(k) set_max_lds((const void*)k_curve_z<BW, L, CV, k, false, true>); if constexpr (!CV) set_max_lds((const void*)k_curve_z<BW, L, false, k, true, true>);
    X(2) X(3) X(4)
#undef X
  }
  if constexpr (chi_exact_built<BW, L, CV>()) prepare_chi_exact_kernels<BW, L, CV>();
}

template <int BW>
static void launch_curve_bw(const Ctx& c, int which, int do_update, hipStream_t st) {
  const int LPC = (c.d.P <= 32) ? 32 : 64;
  const int GPB = 256 / LPC;
  const int nblk = (c.d.n + GPB - 1) / GPB;
  const int K = c.d.K, M = c.d.M;
  const int STR = ((LPC + BW) % 2 == 0) ? LPC + BW + 1 : LPC + BW;      // Tile<BW, LPC>::STR
  const int D = c.d.D;
  const size_t nth = (size_t)K * (M + 1) * c.d.P;
  const size_t tileE = 0;      // (the covariate-adjusted rows are no longer materialised per curve)
  size_t lds;
  const int TW = c.d.mv ? 1 : 2;
  if (which == 0) lds = DYN_LDS_DOUBLES + nth * (1 + D) + GPB * KMAX + GPB + (size_t)GPB * (BW + (TW * K + 3) * STR + MMAX + 48 + tileE);
  else lds = DYN_LDS_DOUBLES + nth * (1 + D) + GPB + GPB * KMAX + (size_t)GPB * (BW + (TW * std::max(M, K + 1) + 2) * STR + 3 * M + 1 + std::max(M * (M + 1) / 2 + M + 1, K + K * (K + 1) / 2) + tileE);
  if (which == 1 || (do_update & 2)) lds = std::max(lds, (size_t)HYPER_LDS_DOUBLES);      // the scalar job's scratch (k_curve_chi, lean k_curve_z)
  lds = (lds + 8) * sizeof(double);
  const bool cov = D > 0;
#define LAUNCH_CURVE(L, CV)                                                                                   \
  do {                                                                                                        \
    if (which == 0) { if (launch_z_exact<BW, L, CV>(c, nblk, lds, st, do_update)) { }                          \
                      else if (K <= 4 && (do_update & 2) && !CV && BW <= 5) hipLaunchKernelGGL((k_curve_z<BW, L, false, 4, true>), dim3(nblk + 8, 1, c.nch), dim3(256), lds, st, c, do_update & 1);  \
                      else if (K <= 4) hipLaunchKernelGGL((k_curve_z<BW, L, CV, 4>), dim3(nblk, 1, c.nch), dim3(256), lds, st, c, do_update & 1);  \
                      else hipLaunchKernelGGL((k_curve_z<BW, L, CV, KMAX>), dim3(nblk, 1, c.nch), dim3(256), lds, st, c, do_update & 1); }  \
    else if (try_chi_exact<BW, L, CV>(c, nblk, lds, st, do_update)) { }                                        \
    else if (K <= 4 && M <= 8) hipLaunchKernelGGL((k_curve_chi<BW, L, CV, true>), dim3(nblk + 8, 1, c.nch), dim3(256), lds, st, c, do_update);      \
    else hipLaunchKernelGGL((k_curve_chi<BW, L, CV, false>), dim3(nblk + 8, 1, c.nch), dim3(256), lds, st, c, do_update);      \
  } while (0)
  if (LPC == 32) { if (cov) LAUNCH_CURVE(32, true); else LAUNCH_CURVE(32, false); }
  else { if (cov) LAUNCH_CURVE(64, true); else LAUNCH_CURVE(64, false); }
#undef LAUNCH_CURVE
}

template <int BW>
static void prepare_bw() {
  set_max_lds((const void*)k_curve_z<BW, 32, false, 4>);
  if constexpr (BW <= 5) { set_max_lds((const void*)k_curve_z<BW, 32, false, 4, true>); set_max_lds((const void*)k_curve_z<BW, 64, false, 4, true>); }
  set_max_lds((const void*)k_curve_z<BW, 32, false, KMAX>);
  set_max_lds((const void*)k_curve_z<BW, 64, false, 4>);
  set_max_lds((const void*)k_curve_z<BW, 64, false, KMAX>);
  set_max_lds((const void*)k_curve_z<BW, 32, true, 4>);
  set_max_lds((const void*)k_curve_z<BW, 32, true, KMAX>);
  set_max_lds((const void*)k_curve_z<BW, 64, true, 4>);
  set_max_lds((const void*)k_curve_z<BW, 64, true, KMAX>);
  set_max_lds((const void*)k_curve_chi<BW, 32, false, true>);
  set_max_lds((const void*)k_curve_chi<BW, 32, false, false>);
  set_max_lds((const void*)k_curve_chi<BW, 64, false, true>);
  set_max_lds((const void*)k_curve_chi<BW, 64, false, false>);
  set_max_lds((const void*)k_curve_chi<BW, 32, true, true>);
  set_max_lds((const void*)k_curve_chi<BW, 32, true, false>);
  set_max_lds((const void*)k_curve_chi<BW, 64, true, true>);
  set_max_lds((const void*)k_curve_chi<BW, 64, true, false>);
  prepare_chi_exact<BW, 32, false>(); prepare_chi_exact<BW, 64, false>(); prepare_chi_exact<BW, 32, true>(); prepare_chi_exact<BW, 64, true>();
}

void prepare_curve_kernels() {
  prepare_bw<0>(); prepare_bw<1>(); prepare_bw<2>(); prepare_bw<3>(); prepare_bw<4>(); prepare_bw<5>(); prepare_bw<BWMID>(); prepare_bw<BWWIDE>();
}

int launch_curve(const Ctx& c, int which, int do_update, hipStream_t st) {
  if (c.d.M > MMAX) return 2;
  switch (c.d.BW) {
    case 0: launch_curve_bw<0>(c, which, do_update, st); break;
    case 1: launch_curve_bw<1>(c, which, do_update, st); break;
    case 2: launch_curve_bw<2>(c, which, do_update, st); break;
    case 3: launch_curve_bw<3>(c, which, do_update, st); break;
    case 4: launch_curve_bw<4>(c, which, do_update, st); break;
    case 5: launch_curve_bw<5>(c, which, do_update, st); break;
    case BWMID: launch_curve_bw<BWMID>(c, which, do_update, st); break;
    case BWWIDE: launch_curve_bw<BWWIDE>(c, which, do_update, st); break;
    default: return 1;
  }
  return 0;
}

int curve_blocks(int n, int P) {
  const int LPC = (P <= 32) ? 32 : 64;
  const int GPB = 256 / LPC;
  return (n + GPB - 1) / GPB;
}
#endif  // !BFMMM_CURVE_EXACT_TU

}  // namespace bfmmm
