// Global-reduction side of the sweep: the pair-weighted Gram contraction (fp64 MFMA), the
// per-direction covariance factorisation and the sequential Gauss-Seidel block of the sweep.
//
// Every Gaussian full conditional of the reference (updateNu UpdateNu.h:24-74, updatePhi
// UpdatePhi.h:23-89 and their Tempered variants) has the form
//     Prec_a = (beta/sigma^2) H_aa + Prior_a
//     rhs_a  = (beta/sigma^2) ( t_a - sum_{b != a} H_ab theta_b )
//     theta_a ~ N( C_a rhs_a, C_a ),  C_a = Prec_a^-1,   drawn as  C_a rhs_a + chol_lower(C_a) z
// with  H_ab = sum_i w_ai w_bi G_i  and  t_a = sum_i w_ai s_i.  Z and chi do not change between the
// Phi and nu blocks of a sweep, so ONE pass over the per-curve records (k_pair_gram) yields every
// H_ab and t_a of the iteration; the K*M + K sequentially dependent draws then run on a few hundred
// KB of H inside one workgroup (k_sweep) instead of K*M + K passes over all curves.
//
//   k_pair_gram : [R pair weights x n] * [n x LG record columns]  -> v_mfma_f64_16x16x4_f64, split-K
//   k_pg_reduce : fixed-order sum of the split-K partial tiles -> H (R x LG), t (A x P)
//   k_factor    : one workgroup per direction: reverse Cholesky of Prec_a -> chol_lower(C_a), C_a, L_a z_a
//   k_sweep     : Phi sweep, nu sweep, sigma^2 (the critical chain)
//   (pi / alpha_3 ride as an extra workgroup of k_pair_gram, delta / A / gamma / tau as an extra
//    workgroup of k_curve_chi: scalar_jobs.hpp)
//   k_loglik    : calcLikelihood (CalculateLikelihood.h:19-44) from the per-curve residual sums
#include "model.hpp"
#include "rng.hpp"
#include "scalar_jobs.hpp"
#include "factor_core.hpp"
#include "z_proposal.hpp"

#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace bfmmm {



#ifdef BFMMM_TIMELINE
void fetch_fct(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fct), sizeof(unsigned long long) * 8); }
#endif

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the vector-memory
// counter (vmcnt(0)), which would serialise the global prefetches the sweep keeps in flight
// across its barriers.
__device__ inline void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---------------------------------------------------------------------------------------------
// pair-Gram
// ---------------------------------------------------------------------------------------------
// grid = (CTG + 2, NKS); block = 256 (4 waves).  Workgroup (ct, ks) owns the 16 record columns
// [16 ct, 16 ct + 16) of the G part for the KS curves of k-slice ks: it stages them once in LDS
// (coalesced 128-byte segments) and its four waves walk the RT row tiles of pair weights, each
// wave issuing one v_mfma_f64_16x16x4_f64 per 4 curves and row tile; the weights w_ai w_bi are
// rebuilt on the fly from Z and chi (also staged in LDS).  Workgroup (CTG, ks) does the same for the
// single-weight rows against the s part of the records (t_a = sum_i w_ai s_i).
//
// LDS per curve i:  raw row  sW = [ Z_i1 .. Z_iK | 1, chi_i1 .. chi_iM | 0 ]  and, for the G workgroups,
// the pair row  sP = [ Z_ij Z_ij' (j <= j') | chit_im chit_im' (m <= m') | 0 ].  Every MFMA weight is then the
// branch-free product of two LDS entries (padding rows point at the 0 slot), so the inner loop has
// no divergence and the TPW accumulators of a wave stay in flight together.
#ifdef BFMMM_TIMELINE
__device__ unsigned long long g_wgtrace[3 * 1024];
void fetch_wgtrace(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wgtrace), sizeof(unsigned long long) * 3 * 1024); }
#endif

constexpr int PG_THREADS = 512;   // 8 waves, two per SIMD: a wave's LDS reads and weight products issue while the other wave's MFMAs execute
                                  // (within one wave MFMA, VALU and LDS issue strictly in order: tools/ubench_mfma.hip)
// H2: the band blocks as k_factor and the sweep read them.  Block r holds, for every row p, the 2 BW + 2 entries
// e(p, k) = G(p, p + k - BW) (k = 2 BW + 1: a zero pad) PIECE-major: the 16-byte piece (e(p, 2 q), e(p, 2 q + 1)) of row p sits at
// v2d index q P + p of the block, so that threads owning consecutive rows read consecutive 16-byte pieces (a row-major block
// made every lane of a wave-wide load touch a cache line of its own: the loads of the sweep's row threads were bound by the
// number of lines per instruction, not by bytes).
__host__ __device__ inline int h2_index(int P, int p, int k) { return (((k >> 1) * P + p) << 1) + (k & 1); }

template <bool BATCH, bool GROUPS>
__global__ __launch_bounds__(PG_THREADS) void k_pair_gram(Ctx c0, int KS, int nks, int do_pg, int G) {
  // Chain batches (BATCH): the workgroup stages its record columns ONCE and walks the chains of the batch in groups of G
  // (the records are shared; Z / chi, the pair weights and the output tiles are per chain), so the grid has no chain
  // dimension.  The weights of a whole group are requested together (one memory round trip per group, not per chain) and
  // its (chain, row tile) items are dealt to the eight waves together: in the Nu_Z stage a chain has ONE row tile, and a
  // group of eight chains keeps all eight waves on the matrix cores.
  // The single-chain instantiation is the same code without the loop (and without the registers it keeps alive).
  const int nch = BATCH ? c0.nch : 1;
  TIMELINE(c0, 1);
#ifdef BFMMM_TIMELINE
  const int wgid = blockIdx.x + gridDim.x * blockIdx.y;
  if (threadIdx.x == 0 && wgid < 1024) {
    unsigned id, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    g_wgtrace[3 * wgid] = wall_clock64();
    g_wgtrace[3 * wgid + 1] = ((unsigned long long)(id & 0xf) << 32) | hw;
  }
  struct EndTrace { int w; __device__ ~EndTrace() { if (threadIdx.x == 0 && w < 1024) g_wgtrace[3 * w + 2] = wall_clock64(); } } et_{wgid};
#endif
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const Dims& d = c0.d;
  const int n = d.n, K = d.K, MD = d.MD;
  const int ks = blockIdx.y, ct = blockIdx.x;
  if (ct == d.CTG + 1) {            // extra workgroups: pi / alpha_3 of chain ks, hidden under the contraction
    if (ks < nch && threadIdx.x < 256) {      // the scalar jobs are written for 256 threads (waves 4-7 leave)
      const Ctx c = chain_ctx(c0, (unsigned)ks);
      job_pi_alpha(c);
      TSTAMP(c, 46);
    } else if (ks >= nch && ks < 2 * nch && threadIdx.x < 256) {
      // the previous iteration's scalar job (delta, A, gamma, tau), left pending by k_curve_chi (Ctx::defer_hyper): its results are
      // first read by k_factor, two kernels on
      const Ctx c = chain_ctx(c0, (unsigned)(ks - nch));
      if (c.dyn->hyper_pending) {
        job_hyper(c, false);
        __syncthreads();
        if (threadIdx.x == 0) c.dyn->hyper_pending = 0u;
      }
    }
    return;
  }
  if (ks >= nks) return;            // (the grid's y extent is max(k-slices, chains))
  if (!do_pg) return;
  const bool single = ct == d.CTG;
  const int ncol = single ? d.CTS * 16 : 16;
  const int col0 = single ? d.LG : ct * 16;
  const int colend = single ? d.LG + d.P : d.LG;
  const int i0 = ks * KS;
  const int RS = K + MD + 1, ONE = K;
  const int NP = d.NZZ + d.NCC, RP = NP + 1;
  // LDS: every quantity is stored per table row (row f of a table = the KS curves of the slice, stride KSP) with curve il of the
  // slice at POSITION pos(il) = (il & 3) KS/4 + (il >> 2): MFMA k-slot kq of step s reads position kq KS/4 + s, i.e. curve
  // 4 s + kq -- the four curves of a step are consecutive curves (the CANONICAL summation order, shared with
  // k_pair_gram_pack, whose chunks are runs of consecutive curves: a chain gives bit-identical H and t through either kernel),
  // and a lane's operands of two consecutive steps are adjacent, so one 16-byte LDS read feeds two MFMAs.
  const int KSP = KS + 2;                    // even (16-byte alignment of the rows) and 2 mod 8 (row starts spread over banks)
  const int KQ4 = KS >> 2;
  auto pos = [&](int il) { return (il & 3) * KQ4 + (il >> 2); };
  const int GG = GROUPS ? G : 1;             // chains staged together (GROUPS: its own instantiation, so that the plain chain loop keeps its registers)
  const int TB = RS + (single ? 0 : RP);     // table rows of a chain
  double* sB = smem;                         // ncol x KSP  record columns
  double* sW = sB + (size_t)ncol * KSP;      // chain g of the group at + g TB KSP:  RS x KSP  raw weights: Z_1..Z_K | 1, chi_1..chi_M | 0
  double* sP = sW + (size_t)RS * KSP;        //                                      RP x KSP  pair weights (G workgroups only)
  const int tid = threadIdx.x;
  constexpr int UW = 12, UB = 6;
  const int ncw = K + MD - 1;                // source columns: Z_1..Z_K, chi_1..chi_M
  const int nB = KS * ncol;
  // staging: a thread issues all its global loads (one curve's Z / chi entries, UB record entries)
  // before its first LDS store, so the workgroup pays about one memory round trip
  auto loadW = [&](const double* Zq, const double* chiq, int il0, int cb, double (&v)[UW]) {
    const int i = min(i0 + il0 + tid, n - 1);
#pragma unroll
    for (int u = 0; u < UW; ++u) {
      const int col = min(cb + u, ncw - 1);
      v[u] = (col < K) ? Zq[i + (size_t)n * col] : chiq[i + (size_t)n * (col - K)];
    }
  };
  auto storeW = [&](int il0, int cb, const double (&v)[UW]) {
    const int il = il0 + tid;
    if (il >= KS) return;
    const bool live = i0 + il < n;
#pragma unroll
    for (int u = 0; u < UW; ++u) {
      const int col = cb + u;
      if (col < ncw) sW[((col < K) ? col : col + 1) * KSP + pos(il)] = live ? v[u] : 0.0;
    }
    if (cb == 0) { sW[ONE * KSP + il] = 1.0; sW[(K + MD) * KSP + il] = 0.0; }      // (constant rows: any order)
  };
  auto loadB = [&](const double* stilq, int base, double (&v)[UB]) {        // s-part workgroups (ncol = CTS * 16)
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int q = min(base + tid + PG_THREADS * u, nB - 1);
      const int il = q / ncol, cc = q - il * ncol;
      const int i = min(i0 + il, n - 1), col = min(col0 + cc, d.LREC - 1);
      // covariate-adjusted models contract against s~_i = s_i - G_i o_i (k_curve_z, per chain) instead of s_i
      v[u] = (d.D > 0) ? stilq[(size_t)i * d.P + min(cc, d.P - 1)] : c0.rec[(size_t)i * d.LREC + col];
    }
  };
  auto storeB = [&](int base, const double (&v)[UB]) {
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int q = base + tid + PG_THREADS * u;
      if (q < nB) {
        const int il = q / ncol, cc = q - il * ncol;
        sB[cc * KSP + pos(il)] = (i0 + il < n && col0 + cc < colend) ? v[u] : 0.0;
      }
    }
  };
  // G workgroups (16 record columns): element (il, cc) = (tid / 16 + 32 u, tid % 16), so a load costs one
  // multiply-add and a store a constant LDS offset
  const int ccg = tid & 15, ilg = tid >> 4;
  const double* srcg = c0.rec + min(col0 + ccg, d.LREC - 1);
  const bool colok = col0 + ccg < colend;
  auto loadG = [&](int ub0, double (&v)[UB]) {
#pragma unroll
    for (int u = 0; u < UB; ++u) v[u] = srcg[(size_t)min(i0 + ilg + (PG_THREADS / 16) * (ub0 + u), n - 1) * d.LREC];
  };
  auto storeG = [&](int ub0, const double (&v)[UB]) {
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int il = ilg + (PG_THREADS / 16) * (ub0 + u);
      if (il < KS) sB[ccg * KSP + pos(il)] = (i0 + il < n && colok) ? v[u] : 0.0;
    }
  };
  // pair slot -> (a, b) table (packed upper triangles of Z x Z and chit x chit), decoded once
  int* ptab = (int*)(sW + (size_t)GG * TB * KSP);
  if (!single && tid < NP) {
    int e = tid, off = 0, dim = K;
    if (e >= d.NZZ) { e -= d.NZZ; off = K; dim = MD; }
    int a = 0;
    while (e >= dim - a) { e -= dim - a; ++a; }
    ptab[tid] = (off + a) | ((off + a + e) << 16);
  }
  const bool shared_cols = !(single && d.D > 0);     // the staged columns are the same for every chain
  for (int q = 0; q < nch; q += GG) {
    const int gc = GROUPS ? min(GG, nch - q) : 1;         // chains of this group
    // the per-chain operands (only these: a whole per-chain Ctx costs a few hundred scalar registers)
    const size_t off1 = (size_t)q * c0.chain_bytes;
    const double* Zq = ptr_shift(c0.Z, off1);
    const double* chiq = ptr_shift(c0.chi, off1);
    const double* stilq = ptr_shift(c0.stil, (size_t)q * c0.chain_bytes_cov);
#ifdef BFMMM_TIMELINE
    struct { Dyn* dyn; } c = {ptr_shift(c0.dyn, off1)};
#endif
    {
      TSTAMP0(c, 40);
      double vw[UW], vb[UB];
      const bool stage_cols = (q == 0) || !shared_cols;
      // BATCH: element e = tid + 512 u of the group's (chain, column, curve) items, curve fastest (gc ncw KS <= 512 UW)
      const int nitem = gc * ncw * KS;
      constexpr bool grp = GROUPS;
      if (grp) {
#pragma unroll
        for (int u = 0; u < UW; ++u) {
          const int e = min(tid + PG_THREADS * u, nitem - 1);
          const int ci = e / KS, il = e - ci * KS;
          const int g = ci / ncw, col = ci - g * ncw;
          const int i = min(i0 + il, n - 1);
          const size_t offg = (size_t)g * c0.chain_bytes;
          vw[u] = (col < K) ? ptr_shift(Zq, offg)[i + (size_t)n * col] : ptr_shift(chiq, offg)[i + (size_t)n * (col - K)];
        }
      } else {
        loadW(Zq, chiq, 0, 0, vw);
      }
      if (stage_cols) { if (single) loadB(stilq, 0, vb); else loadG(0, vb); }
      TSTAMP0(c, 47);
      if (grp) {
#pragma unroll
        for (int u = 0; u < UW; ++u) {
          const int e = tid + PG_THREADS * u;
          if (e < nitem) {
            const int ci = e / KS, il = e - ci * KS;
            const int g = ci / ncw, col = ci - g * ncw;
            sW[((size_t)g * TB + ((col < K) ? col : col + 1)) * KSP + pos(il)] = (i0 + il < n) ? vw[u] : 0.0;
          }
        }
        for (int x = tid; x < gc * KS; x += PG_THREADS) {
          const int g = x / KS, il = x - g * KS;
          sW[((size_t)g * TB + ONE) * KSP + il] = 1.0; sW[((size_t)g * TB + K + MD) * KSP + il] = 0.0;
        }
      } else {
        storeW(0, 0, vw);
      }
      TSTAMP0(c, 48);
      if (stage_cols) { if (single) storeB(0, vb); else storeG(0, vb); }
      if (!grp)
      for (int il0 = 0; il0 < KS; il0 += PG_THREADS)
        for (int cb = 0; cb < ncw; cb += UW) {
          if (il0 == 0 && cb == 0) continue;
          loadW(Zq, chiq, il0, cb, vw);
          storeW(il0, cb, vw);
        }
      if (stage_cols) {
        if (single) { for (int base = PG_THREADS * UB; base < nB; base += PG_THREADS * UB) { loadB(stilq, base, vb); storeB(base, vb); } }
        else { for (int ub0 = UB; (PG_THREADS / 16) * ub0 < KS; ub0 += UB) { loadG(ub0, vb); storeG(ub0, vb); } }
      }
      TSTAMP0(c, 41);
    }
    __syncthreads();
    TSTAMP0(c, 42);
    if (!single) {
      // pair rows: thread (tx, ty) = (tid % 32, tid / 32) fills pair slots ty, ty + 16, .. of curves tx, tx + 32, ..
      const int tx = tid & 31, ty = tid >> 5;
      for (int il0 = 0; il0 < KS; il0 += 256) {
        for (int e = ty; e < NP * gc; e += PG_THREADS / 32) {
          const int g = GROUPS ? e / NP : 0, ep = e - g * NP;
          const int pk = ptab[ep], ia = pk & 0xffff, ib = pk >> 16;
          double* sPg = sP + (size_t)g * TB * KSP;
          const double* sWg = sW + (size_t)g * TB * KSP;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int il = il0 + tx + 32 * j;
            if (il < KS) sPg[ep * KSP + il] = sWg[ia * KSP + il] * sWg[ib * KSP + il];
          }
        }
        if (ty < gc)
#pragma unroll
          for (int j = 0; j < 8; ++j) { const int il = il0 + tx + 32 * j; if (il < KS) sP[((size_t)ty * TB + NP) * KSP + il] = 0.0; }
      }
      __syncthreads();
    }
    TSTAMP0(c, 43);
    const int wave = tid >> 6, lane = tid & 63;
    const int lr = lane & 15, kq = lane >> 4;
    const int ntile = single ? d.AT * d.CTS : d.RT;
    const double* wsrc = single ? sW : sP;
    const int ZERO = single ? RS - 1 : RP - 1;
    const int KQ = KS / 4;                     // steps; k-slot kq of step s is curve kq * KQ + s  (KS is a multiple of 16)
    // each wave walks its tiles TPW at a time with independent accumulators
    constexpr int TPW = 2;
    constexpr int NW = PG_THREADS / 64;
    const int nitems = gc * ntile;             // (chain of the group, row tile)
    for (int t0 = wave; t0 < nitems; t0 += NW * TPW) {
      int tix[TPW], bcol[TPW], o1[TPW], o2[TPW], gch[TPW];
      bool tv[TPW];
#pragma unroll
      for (int qq = 0; qq < TPW; ++qq) {
        const int it = t0 + NW * qq;
        tv[qq] = it < nitems;
        gch[qq] = (GROUPS && tv[qq]) ? it / ntile : 0;
        const int tt = it - gch[qq] * ntile;
        o1[qq] = o2[qq] = ZERO; bcol[qq] = lr; tix[qq] = 0;
        if (tv[qq]) {
          if (!single) {
            const int row = tt * 16 + lr;
            tix[qq] = tt * d.CTG + ct;
            if (row < d.R) { const int zz = row / d.NCC; o1[qq] = zz; o2[qq] = d.NZZ + (row - zz * d.NCC); }
          } else {
            const int at = tt / d.CTS, cs = tt - at * d.CTS;
            const int row = at * 16 + lr;
            tix[qq] = d.RT * d.CTG + tt;
            bcol[qq] = cs * 16 + lr;
            if (row < d.A) { const int j = row / MD; o1[qq] = j; o2[qq] = K + (row - j * MD); }
          }
        }
      }
      double4_t acc[TPW];
#pragma unroll
      for (int qq = 0; qq < TPW; ++qq) acc[qq] = double4_t{0.0, 0.0, 0.0, 0.0};
      const v2d* pa[TPW]; const v2d* pb[TPW]; const v2d* pc[TPW];
#pragma unroll
      for (int qq = 0; qq < TPW; ++qq) {
        const double* wg = wsrc + (size_t)gch[qq] * TB * KSP;
        pa[qq] = (const v2d*)(wg + o1[qq] * KSP + kq * KQ);
        pb[qq] = (const v2d*)(wg + o2[qq] * KSP + kq * KQ);
        pc[qq] = (const v2d*)(sB + bcol[qq] * KSP + kq * KQ);
      }
      // The LDS pipe moves 1.5 KB per MFMA and wave -- three quarters of the time the matrix pipe needs for it -- so the
      // two must overlap: a trip is two pairs of k-steps (12 TPW MFMAs); the operands of trip t + 1 are read into the
      // other register set before the MFMAs of trip t are issued (the scheduling barriers keep the compiler from
      // moving the reads back next to their uses).
      struct OpSet { v2d wa[2][TPW], wb[2][TPW], bb[2][TPW]; };
      auto load_trip = [&](OpSet& o, int s2) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int qq = 0; qq < TPW; ++qq) { o.wa[u][qq] = pa[qq][s2 + u]; o.wb[u][qq] = pb[qq][s2 + u]; o.bb[u][qq] = pc[qq][s2 + u]; }
      };
      auto mfma_trip = [&](const OpSet& o, int npair) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
          if (u < npair)
#pragma unroll
            for (int qq = 0; qq < TPW; ++qq) {
              acc[qq] = __builtin_amdgcn_mfma_f64_16x16x4f64(o.wa[u][qq].x * o.wb[u][qq].x, o.bb[u][qq].x, acc[qq], 0, 0, 0);
              acc[qq] = __builtin_amdgcn_mfma_f64_16x16x4f64(o.wa[u][qq].y * o.wb[u][qq].y, o.bb[u][qq].y, acc[qq], 0, 0, 0);
            }
      };
      const int ntrip = KQ / 4;                // KS is a multiple of 16: trip t covers the step pairs 2t, 2t + 1
      OpSet s0, s1;
      load_trip(s0, 0);
      for (int t = 0; t < ntrip; t += 2) {
        if (t + 1 < ntrip) load_trip(s1, 2 * (t + 1));
        __builtin_amdgcn_sched_barrier(0);
        mfma_trip(s0, 2);
        __builtin_amdgcn_sched_barrier(0);
        if (t + 2 < ntrip) load_trip(s0, 2 * (t + 2));
        __builtin_amdgcn_sched_barrier(0);
        if (t + 1 < ntrip) mfma_trip(s1, 2);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int qq = 0; qq < TPW; ++qq)
        if (tv[qq]) {
          double* out = ptr_shift(c0.pg_part, (size_t)(q + gch[qq]) * c0.chain_bytes) + ((size_t)ks * d.NT + tix[qq]) * 256 + lane;
          // (streaming stores: the 4.5 MB of partial tiles are read next by k_pg_reduce on other XCDs, never again by this one; written
          //  through as they are produced they do not sit dirty in this XCD's L2 until the end-of-kernel write-back)
          // (single chain only: the 8-chain Nu_Z batch measured 2 % slower with them)
          if constexpr (!BATCH) {
            __builtin_nontemporal_store(acc[qq][0], out); __builtin_nontemporal_store(acc[qq][1], out + 64);
            __builtin_nontemporal_store(acc[qq][2], out + 128); __builtin_nontemporal_store(acc[qq][3], out + 192);
          } else {
            out[0] = acc[qq][0]; out[64] = acc[qq][1]; out[128] = acc[qq][2]; out[192] = acc[qq][3];
          }
        }
      TSTAMP0(c, 44);
    }
    if (q + GG < nch) __syncthreads();       // the next group overwrites sW / sP
  }
}

// four lanes per element of every output tile (lane g sums the k-slices g, g + 4, ..: the same four interleaved partial sums
// as ever, combined in the same fixed order), so that the 25 dependent-latency loads of an element shrink to 7
__global__ __launch_bounds__(256) void k_pg_reduce(Ctx c0, int NKS) {
  const Ctx c = chain_view(c0);      // chain blockIdx.z of the batch
  TIMELINE(c, 2);
  const Dims& d = c.d;
  const int gid4 = blockIdx.x * 256 + threadIdx.x;
  const int gid = gid4 >> 2, g = gid4 & 3;
  const bool live = gid < d.NT * 256;
  const int gc = live ? gid : 0;
  const int t = gc >> 8, q = gc & 255;
  const int r = q >> 6, lane = q & 63;
  const int rit = (lane >> 4) + 4 * r, cit = lane & 15;   // D layout of v_mfma_f64_16x16x4_f64
  const double* src = c.pg_part + (size_t)t * 256 + q;
  const size_t stride = (size_t)d.NT * 256;
  double sg = 0.0;
  const int nfull = NKS & ~3;                             // slices 0 .. nfull-1 go to the four interleaved sums
  // eight loads of a lane go out together (the slabs were written by the previous kernel: every load is a trip to memory,
  // and a plain accumulation loop pays one trip per term); the additions keep the sequential order
  for (int k0 = g; k0 < nfull; k0 += 32) {
    double v8[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v8[u] = src[(size_t)min(k0 + 4 * u, NKS - 1) * stride];
#pragma unroll
    for (int u = 0; u < 8; ++u) if (k0 + 4 * u < nfull) sg += v8[u];
  }
  {
    double vt[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) vt[u] = src[(size_t)min(nfull + u, NKS - 1) * stride];
#pragma unroll
    for (int u = 0; u < 3; ++u) if (g == 0 && nfull + u < NKS) sg += vt[u];
  }
  // (s0 + s1) + (s2 + s3), s_g on lane g of the quad
  const double s01 = sg + __shfl_xor(sg, 1, 4);
  const double s = s01 + __shfl_xor(s01, 2, 4);
  if (!live || g != 0) return;
  const int n_pair_tiles = d.RT * d.CTG;
  if (t < n_pair_tiles) {
    const int rt = t / d.CTG, ct = t - rt * d.CTG;
    const int row = rt * 16 + rit, col = ct * 16 + cit;
    if (d.mv) {
      // G_i = I: the block is s I (BW = 0: H2 rows are [G(p, p), 0])
      // (the 16 columns of the tile are the same column of ones: every lane of a row holds the same sum and takes its share
      //  of the P columns)
      if (row < d.R) {
        double* h2 = c.H2 + (size_t)row * d.P * 2;
        for (int p0 = cit; p0 < d.P; p0 += 16) { c.H[(size_t)row * d.LG + p0] = s; h2[2 * p0] = s; }
      }
    } else if (row < d.R && col < d.LG) {
      c.H[(size_t)row * d.LG + col] = s;
      // copy for k_factor / the sweep (h2_index): entry k of row p is G(p, p + k - BW)
      const int dd = col / d.P, p0 = col - dd * d.P, W = 2 * d.BW + 2;
      double* h2 = c.H2 + (size_t)row * d.P * W;
      h2[h2_index(d.P, p0, d.BW + dd)] = s;
      if (dd > 0 && p0 + dd < d.P) h2[h2_index(d.P, p0 + dd, d.BW - dd)] = s;
    }
  } else {
    const int t2 = t - n_pair_tiles;
    const int at = t2 / d.CTS, cs = t2 - at * d.CTS;
    const int row = at * 16 + rit, col = cs * 16 + cit;
    if (row < d.A && col < d.P) c.tvec[(size_t)row * d.P + col] = s;
  }
}

// ---------------------------------------------------------------------------------------------
// pair-Gram for chain BATCHES and LONG curve sets (round 4): row tiles packed across the chains, chunked k-loop
// ---------------------------------------------------------------------------------------------
// The batch is ONE contraction  [nch R pair rows] x [n curves] x [LG record columns]  (+ [nch A single rows] x n x P for t_a):
//  * the 16-row MFMA tiles run over the rows of ALL chains of the (sub-)batch back to back, so only the last tile is padded
//    (8 x 168 rows = 84 full tiles instead of 8 x 11 with 8 rows of padding each; the s part 8 x 21 = 168 rows = 11 tiles
//    instead of 16);
//  * a wave owns ONE packed row tile and ALL column tiles of it (2 NP2 accumulators), so an A operand -- the product of two pair
//    weights -- is formed once per k-step for 2 NP2 MFMAs and the LDS pipe moves 640 B per MFMA instead of 1.5 KB;
//  * the workgroup walks its k-slice in CHUNKS of 16 curves (4 k-steps), double-buffered: the next chunk's records and weights
//    are requested before the MFMAs of this one and stored after them, one barrier per chunk; LDS is ~50 KB and the kernel
//    holds <= 128 VGPRs, so two workgroups (16 waves) share a CU and one's staging hides behind the other's MFMAs.  The
//    accumulators persist across the chunks: partial tiles are written once per k-slice whatever its length.
//  * records are staged as they lie in memory ([curve][column], no transposition): the 16-byte B read of lane (n, kq) holds columns
//    32 tp + 2 n and 32 tp + 2 n + 1 of curve 4 s + kq and feeds TWO column tiles (the even and the odd columns of a 32-column
//    pair) -- any assignment of columns to tiles will do, k_pg_reduce_pack knows it.
// Summation order (canonical, shared with k_pair_gram): slice ks = curves [ks KS, ks KS + KS), one MFMA chain over its k-steps,
// step s = curves 4 s .. 4 s + 3; A = (Z_j Z_j') (chit_m chit_m') resp. Z_j chit_m; slices combined by k_pg_reduce_pack in
// k_pg_reduce's order.  A chain's H and t are therefore bit-identical to what k_pair_gram + k_pg_reduce give it alone.
// Limits (otherwise the launcher keeps k_pair_gram): K <= 4, M <= 8, LG <= 128, P <= 32, no covariates, functional model.
constexpr int PGP_CH = 16;                     // chunk: 16 curves = 4 k-steps
constexpr int PGP_RB = 66;                     // row stride of the staged chunk (doubles): 64 record columns + 2 (16-byte aligned rows)
constexpr int PGP_NWV = 3;                     // most weight values (Z_ik, chi_im) a thread stages per chunk

template <int PGP_WAVES>
__global__ __launch_bounds__(64 * PGP_WAVES, 4) void k_pair_gram_pack(Ctx c0, PgPack g, double* __restrict__ pack) {
  constexpr int PGP_THREADS = 64 * PGP_WAVES;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const Dims& d = c0.d;
  const int n = d.n, K = d.K, MD = d.MD, nch = c0.nch;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, kq = lane >> 4;
  const int ks = blockIdx.y, wg = blockIdx.x;
  const int nwg_g = g.NRG * g.NCG;
  if (ks >= g.NKS || wg >= nwg_g + g.NWG_S) return;
  // A workgroup is (row group, column group): PGP_WAVES packed row tiles (one per wave) x 64 record columns (two 32-column pairs,
  // four accumulators per wave).  The s part -- single-weight rows against the P columns of s_i -- runs through the SAME code: its
  // rows multiply Z_j chit_m by 1 x 1 (exact), its 32 columns fill the first pair and the second pair's accumulators are dropped.
  const bool single = wg >= nwg_g;
  const int rg = single ? wg - nwg_g : wg / g.NCG, cg = single ? 0 : wg - rg * g.NCG;
  const int i0 = ks * g.KS;
  const int nchunk = (min(g.KS, n - i0) + PGP_CH - 1) / PGP_CH;
  const int SL = g.SLS;                     // raw weight row of a (chain, curve): Z_1 .. Z_K | 1, chi_1 .. chi_M | 0 (| pad)
  const int RW = single ? d.A : d.R;        // rows per chain
  const int tile0 = rg * PGP_WAVES;
  const int ntile = single ? g.TS : g.TG;
  const int q0 = min((tile0 * 16) / RW, nch - 1);       // the chains this workgroup's rows belong to
  const int q1 = min(((min(tile0 + PGP_WAVES, ntile)) * 16 - 1) / RW, nch - 1);
  const int nq = q1 - q0 + 1;
  constexpr int nbuf_b = 16 * PGP_RB;
  double* sBb = smem;                                   // 2 x 16 x RB     record chunk, [curve][column]
  double* sWb = smem + 2 * nbuf_b;                      // 2 x nq x 16 x SL   weights, [chain][curve][slot]
  const int nbuf_w = nq * 16 * SL;
  // ---- staging roles ----
  // records: ONE 16-byte piece per thread: curve il = tid / 32 of the chunk, columns cb0 + 2 pc, + 1 (pc = tid % 32); the s part
  // starts at column LG, which need not be 16-byte aligned: two 8-byte loads there
  constexpr int NPB = 512 / PGP_THREADS;          // pieces per thread: piece e = tid + PGP_THREADS u -> curve e / 32, columns 2 (e % 32), + 1
  const int b_pc = tid & 31;
  const int cb0 = single ? d.LG : 64 * cg, cend = single ? d.LG + d.P : d.LG;      // (columns beyond the part are zero)
  const int bc = cb0 + 2 * b_pc;
  const bool ok0 = bc < cend, ok1 = bc + 1 < cend;
  const int bc0 = min(bc, d.LREC - 2);
  v2d vb[NPB];
  // weights: value e = tid + 512 u of the workgroup's (chain, column, curve) items, curve fastest: chain q0 + e / (16 NV), source
  // column (e / 16) % NV (Z_1 .. Z_K, chi_1 .. chi_M), curve e % 16 -- sixteen consecutive threads read a 128-byte segment
  const int NV = K + MD - 1, nval = nq * NV * 16;
  double wv[PGP_NWV];
  const double* wsrc[PGP_NWV];
  int wdst[PGP_NWV];
#pragma unroll
  for (int u = 0; u < PGP_NWV; ++u) {
    const int e = min(tid + PGP_THREADS * u, nval - 1);
    const int qq = e / (16 * NV), v = (e >> 4) - qq * NV;
    const size_t offq = (size_t)(q0 + qq) * c0.chain_bytes;
    wsrc[u] = (v < K) ? ptr_shift(c0.Z, offq) + (size_t)n * v : ptr_shift(c0.chi, offq) + (size_t)n * (v - K);
    wdst[u] = (qq * 16 + (e & 15)) * SL + ((v < K) ? v : v + 1);
    wv[u] = 0.0;
  }
  const int w_il = tid & 15;
  auto load_chunk = [&](int t) {
    const int ib = i0 + t * PGP_CH;
#pragma unroll
    for (int u = 0; u < NPB; ++u) {
      const int b_il = (tid + PGP_THREADS * u) >> 5;
      const double* src = c0.rec + (size_t)min(ib + b_il, n - 1) * d.LREC + bc0;
      if (!single) vb[u] = *(const v2d*)src;
      else { vb[u].x = src[0]; vb[u].y = src[1]; }
    }
    const int i = min(ib + w_il, n - 1);
#pragma unroll
    for (int u = 0; u < PGP_NWV; ++u)
      if (PGP_THREADS * u < nval) wv[u] = wsrc[u][i];
  };
  auto store_chunk = [&](int t, int buf) {
    const int ib = i0 + t * PGP_CH;
#pragma unroll
    for (int u = 0; u < NPB; ++u) {
      const int b_il = (tid + PGP_THREADS * u) >> 5;
      const bool liveb = ib + b_il < n;
      v2d o2;
      o2.x = (liveb && ok0) ? vb[u].x : 0.0; o2.y = (liveb && ok1) ? vb[u].y : 0.0;
      *(v2d*)(sBb + buf * nbuf_b + b_il * PGP_RB + 2 * b_pc) = o2;
    }
    const bool live = ib + w_il < n;
    double* sW = sWb + buf * nbuf_w;
#pragma unroll
    for (int u = 0; u < PGP_NWV; ++u)
      if (tid + PGP_THREADS * u < nval) sW[wdst[u]] = live ? wv[u] : 0.0;
  };
  // the constant slots of both weight buffers: chit_0 = 1 at K, the zero slot (rows of the tile padding) at K + MD
  for (int x = tid; x < 2 * nq * 16; x += PGP_THREADS) {
    double* w = sWb + (x / (nq * 16)) * nbuf_w + (x % (nq * 16)) * SL;
    w[K] = 1.0; w[K + MD] = 0.0;
  }
  // ---- MFMA role of the wave: packed row tile tile0 + wave, four accumulators (column pair 0 even / odd, pair 1 even / odd) ----
  double4_t acc[4];
#pragma unroll
  for (int x = 0; x < 4; ++x) acc[x] = double4_t{0.0, 0.0, 0.0, 0.0};
  // LDS offsets (doubles, within a weight buffer) of the four factors of this lane's row: A = (w[o0] w[o1]) (w[o2] w[o3]);
  // pair row (j, j', m, m'): Z_j Z_j' chit_m chit_m';  single row (j, m): Z_j chit_m 1 1;  padding rows: the zero slot
  const int ZERO = K + MD;
  int o0 = ZERO, o1 = ZERO, o2_ = ZERO, o3 = ZERO;
  const int tile = tile0 + wave;
  const bool has_tile = tile < ntile;
  if (has_tile) {
    const int grow = tile * 16 + lr;
    const int q = grow / RW, r = grow - q * RW;
    if (q < nch) {
      const int base = (q - q0) * 16 * SL;
      if (!single) {
        const int zz = r / d.NCC, cc = r - zz * d.NCC;
        int a = 0, e = zz;
        while (e >= K - a) { e -= K - a; ++a; }
        o0 = base + a; o1 = base + a + e;
        a = 0; e = cc;
        while (e >= MD - a) { e -= MD - a; ++a; }
        o2_ = base + K + a; o3 = base + K + a + e;
      } else {
        const int j = r / MD;
        o0 = base + j; o1 = base + K + (r - j * MD); o2_ = base + K; o3 = base + K;
      }
    }
  }
  auto mfma_chunk = [&](int buf) {
    const double* sB = sBb + buf * nbuf_b + 2 * lr;
    const double* sW = sWb + buf * nbuf_w;
#pragma unroll
    for (int s2 = 0; s2 < 4; s2 += 2) {
      double a[2];
      v2d b[2][2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int il = 4 * (s2 + u) + kq;
        const double* w = sW + il * SL;
        a[u] = (w[o0] * w[o1]) * (w[o2_] * w[o3]);
        b[u][0] = *(const v2d*)(sB + il * PGP_RB);
        b[u][1] = *(const v2d*)(sB + il * PGP_RB + 32);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u][0].x, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u][0].y, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u][1].x, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u][1].y, acc[3], 0, 0, 0);
      }
    }
  };
  // ---- the chunk pipeline: the next chunk is requested before the MFMAs of this one and stored behind them ----
  load_chunk(0);
  store_chunk(0, 0);
  __syncthreads();
  for (int t = 0; t < nchunk; ++t) {
    if (t + 1 < nchunk) load_chunk(t + 1);
    mfma_chunk(t & 1);
    if (t + 1 < nchunk) store_chunk(t + 1, (t + 1) & 1);
    __syncthreads();
  }
  // ---- partial tiles of this k-slice, accumulator layout (k_pg_reduce_pack): pair tile T, column pair tp, parity par at
  //      index (T NP2 + tp) 2 + par; single tile T, parity par at TG 2 NP2 + 2 T + par ----
  if (!has_tile) return;
  double* out0 = pack + (size_t)ks * g.NTP * 256 + lane;
#pragma unroll
  for (int x = 0; x < 4; ++x) {
    const int tp = 2 * cg + (x >> 1);
    const bool keep = single ? (x < 2) : (tp < g.NP2);
    if (keep) {
      const int idx = single ? g.TG * 2 * g.NP2 + tile * 2 + x : (tile * g.NP2 + tp) * 2 + (x & 1);
      double* out = out0 + (size_t)idx * 256;
      out[0] = acc[x][0]; out[64] = acc[x][1]; out[128] = acc[x][2]; out[192] = acc[x][3];
    }
  }
}

// pi / alpha_3 (+ the deferred log-likelihood) of every chain of the (sub-)batch: in k_pair_gram this job rides as an extra
// workgroup; inside k_pair_gram_pack it would cost that kernel its register budget (the job needs 177 VGPRs, the contraction 88),
// so it is a workgroup of the NEXT kernel, k_factor, whose register budget it fits; the spare jobs there that read pi / alpha_3
// wait for its flag (k_factor: pi_in_factor).  Tried first: as workgroups of the reduction kernel (its 177 VGPRs then set that
// memory-bound kernel's occupancy: 17.7 us instead of 12 for 8 chains, 59 instead of 40 for 32) and on a side stream forked and
// joined inside the captured graph (the two cross-stream edges cost the 8-chain batch 80 us per step).
// fixed-order sum of the k-slices of the packed partial tiles (the order of k_pg_reduce: four interleaved partial sums over the
// slices, then (s0 + s1) + (s2 + s3)) and scatter to the chains' H, H2, t
__global__ __launch_bounds__(256) void k_pg_reduce_pack(Ctx c0, PgPack g, const double* __restrict__ pack) {
  const Dims& d = c0.d;
  const int bx = blockIdx.x;
  const int gid4 = bx * 256 + threadIdx.x;
  const int gid = gid4 >> 2, gl = gid4 & 3;
  const bool live = gid < g.NTP * 256;
  const int gc = live ? gid : 0;
  const int t = gc >> 8, q = gc & 255;
  const int r = q >> 6, lane = q & 63;
  const int rit = (lane >> 4) + 4 * r, cit = lane & 15;   // D layout of v_mfma_f64_16x16x4_f64
  const double* src = pack + (size_t)t * 256 + q;
  const size_t stride = (size_t)g.NTP * 256;
  const int NKS = g.NKS;
  double sg = 0.0;
  const int nfull = NKS & ~3;
  for (int k0 = gl; k0 < nfull; k0 += 32) {
    double v8[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v8[u] = src[(size_t)min(k0 + 4 * u, NKS - 1) * stride];
#pragma unroll
    for (int u = 0; u < 8; ++u) if (k0 + 4 * u < nfull) sg += v8[u];
  }
  {
    double vt[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) vt[u] = src[(size_t)min(nfull + u, NKS - 1) * stride];
#pragma unroll
    for (int u = 0; u < 3; ++u) if (gl == 0 && nfull + u < NKS) sg += vt[u];
  }
  const double s01 = sg + __shfl_xor(sg, 1, 4);
  const double s = s01 + __shfl_xor(s01, 2, 4);
  if (!live || gl != 0) return;
  const int NACC = 2 * g.NP2;
  if (t < g.TG * NACC) {
    const int tile = t / NACC, x = t - tile * NACC;      // x = 2 tp + par
    const int grow = tile * 16 + rit;
    const int qc = grow / d.R, row = grow - qc * d.R;
    const int col = 32 * (x >> 1) + 2 * cit + (x & 1);
    if (qc < c0.nch && col < d.LG) {
      const size_t off = (size_t)qc * c0.chain_bytes;
      ptr_shift(c0.H, off)[(size_t)row * d.LG + col] = s;
      const int dd = col / d.P, p0 = col - dd * d.P, W = 2 * d.BW + 2;
      double* h2 = ptr_shift(c0.H2, off) + (size_t)row * d.P * W;
      h2[h2_index(d.P, p0, d.BW + dd)] = s;
      if (dd > 0 && p0 + dd < d.P) h2[h2_index(d.P, p0 + dd, d.BW - dd)] = s;
    }
  } else {
    const int t2 = t - g.TG * NACC;
    const int tile = t2 >> 1, par = t2 & 1;
    const int grow = tile * 16 + rit;
    const int qc = grow / d.A, a = grow - qc * d.A;
    const int p = 2 * cit + par;
    if (qc < c0.nch && p < d.P) ptr_shift(c0.tvec, (size_t)qc * c0.chain_bytes)[(size_t)a * d.P + p] = s;
  }
}

// ---------------------------------------------------------------------------------------------
// helpers shared by k_factor / k_sweep
// ---------------------------------------------------------------------------------------------
__device__ inline int hrow(const Dims& d, int a, int b) {   // a, b: active direction indices
  const int ja = a / d.MD, ma = a - ja * d.MD, jb = b / d.MD, mb = b - jb * d.MD;
  const int zz = tri_index(d.K, min(ja, jb), max(ja, jb));
  const int cc = tri_index(d.MD, min(ma, mb), max(ma, mb));
  return zz * d.NCC + cc;
}

__device__ inline int full_dir(const Dims& d, int a) {   // active direction -> row of c.theta
  const int j = a / d.MD, mt = a - j * d.MD;
  return j * (d.M + 1) + mt;
}

// (H_block * v)[p] for a band-packed symmetric block
__device__ inline double band_mv(const double* __restrict__ Hb, const double* v, int P, int BW, int p) {
  double s = Hb[p] * v[p];
  for (int dd = 1; dd <= BW; ++dd) {
    if (p + dd < P) s += Hb[dd * P + p] * v[p + dd];
    if (p - dd >= 0) s += Hb[dd * P + p - dd] * v[p - dd];
  }
  return s;
}

// ---------------------------------------------------------------------------------------------
__device__ inline int step_dir(const Dims& d, int s, int n_phi);

// k_factor: one workgroup per active direction a.
//   r_a  = t_a - sum_b H_ab theta_b,  hq_a = H_aa theta_a      (always: the sweep starts from these)
//   Prec = (beta/sigma^2) H_aa + Prior_a                        (banded: half-width BWP)
//   Prec = U U'  with U UPPER triangular ("reverse" Cholesky, processed from the last row up).
//   Then  C = Prec^-1 = U^-T U^-1  and, because U^-T is lower triangular with positive diagonal,
//   chol_lower(C) = U^-T  exactly -- the factor arma::mvnrnd(C b, C) multiplies z by
//   (UpdateNu.h:67-69, UpdatePhi.h:79-82).  So one banded factorisation + one triangular inverse
//   give both the reference's covariance C and its Cholesky factor L; no dense inverse is formed
//   by elimination.  The direction's normal variates z and L z are produced here as well, so the
//   sequential sweep only has to apply C.
// A pivot below 1e-12 of the largest diagonal entry sends the direction down the reference's arma::pinv / eigen-decomposition
// route instead (factor_pinv, factor_core.hpp).
// ---------------------------------------------------------------------------------------------
template <int PP, int BW>
// (three workgroups per CU: at four the 128-register cap spilled 62 registers of the factorisation path to scratch -- one chain
//  64.7 us per iteration against 65.6, 8 Nu_Z chains 103 k iterations/s against 100 k; the spare jobs of a batch still overlap)
__global__ __launch_bounds__(256, (BW <= 5) ? 3 : 1) void k_factor(Ctx c0) {
  // one-dimensional grid of chains x jobs with the chain index running FASTEST, so that the long factorisation workgroups
  // of every chain of a batch are dispatched before any of the short spare jobs (workgroups start in index order)
  const int nch_ = c0.nch;
  const Ctx c = chain_ctx(c0, blockIdx.x % nch_);
  const int bx = blockIdx.x / nch_;      // job index
  TIMELINE(c, 3);
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const Dims& d = c.d;
  const int P = d.P, MD = d.MD, K = d.K, A = d.A, M = d.M;
  constexpr int W = 2 * BW + 2;     // doubles per row of an H2 block: G(p, p - BW .. p + BW), 0
  const int tid = threadIdx.x;
  // Two workgroups per direction (round 4; not for the diagonal model, which has no factorisation): workgroup a < A runs the
  // factorisation of Prec_a -- it needs only H_aa and the prior -- and workgroup A + a forms r_a = t_a - sum_b H_ab theta_b and
  // H_aa theta_a, which the factorisation does not need: side by side on two CUs instead of one after the other (the r phase
  // was 2.2 us of the factorisation workgroup's 13).
  const bool split = !((BW == 0) && d.BWP == 0);
  const int nF0 = split ? 2 * A : A;       // first spare job
  if (c.pi_in_factor && bx == nF0) {
    // The iteration's pi / alpha_3 job (normally an extra workgroup of k_pair_gram; on the packed pair-Gram path of chain
    // batches it would cost k_pair_gram_pack or its reduction their register budget -- it needs 177 VGPRs, they 88 and 46).  The
    // one spare job that reads pi / alpha_3 of THIS iteration (job_pi_prepare, one workgroup) waits for it (wait_pi below); the
    // pi job is dispatched before it (lower workgroup index), so the wait cannot deadlock, and it is bounded anyway.
    job_pi_alpha(c);
    __syncthreads();
    if (tid == 0) {
      __threadfence();
      __hip_atomic_store(&c.dyn->pi_done, c.dyn->iter + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    return;
  }
  const int nF = nF0 + (c.pi_in_factor ? 1 : 0);
  auto wait_pi = [&]() {
    if (!c.pi_in_factor) return;
    if (tid == 0) {
      const uint32_t want = c.dyn->iter + 1u;
      int spins = 0;
      // (relaxed polls -- an acquire per poll invalidates the XCD's L2 every time, and four hundred waiting workgroups doing that
      //  made the kernel four times longer -- and ONE acquire fence once the flag is up)
      while (__hip_atomic_load(&c.dyn->pi_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != want) {
        __builtin_amdgcn_s_sleep(64);
        if (++spins > (1 << 20)) { atomicOr(&c.dyn->status, 8u); break; }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
  };
  if (bx >= nF) {       // spare workgroups: the state-independent variates of job_hyper, then next iteration's Z proposals
    const int ndraw = (hyper_gstd_count(d) + 1 + 8 * d.K + 255) / 256;
    const int zcw = zprep_curves_per_wg(d.K);
    const int nzp = (c.mask & U_Z) ? (d.n + zcw - 1) / zcw : 0;
    // Order: draws, then the ONE job that waits (next iteration's pi / alpha_3 tables: it needs this iteration's pi job, which has
    // a lower workgroup index and is therefore running by the time this one is dispatched -- no deadlock), then the short ones.  It
    // used to be the LAST workgroup of the grid: in a batch it then started when everything else had been dispatched (21 us into
    // the kernel for eight chains) and its 5.5 us were the kernel's tail.
    const int npi = (c.mask & (U_PI | U_ALPHA3)) ? 1 : 0;
    const int sb = bx - nF;
#ifdef BFMMM_TIMELINE
    // one workgroup of each kind of spare job: start / end stamps 56 .. 63
    const int kind_ = sb < ndraw ? 0 : sb < ndraw + npi ? 2 : sb < ndraw + npi + nzp ? 1 : 3;
    const bool first_ = (sb == 0) || kind_ == 2 || (sb == ndraw + npi && nzp > 0) || (kind_ == 3 && sb == ndraw + npi + nzp);
    if (first_ && threadIdx.x == 0) c.dyn->stamps[56 + 2 * kind_] = wall_clock64();
#endif
    if (sb < ndraw) job_hyper_draws(c, sb * 256);
    else if (sb < ndraw + npi) { wait_pi(); job_pi_prepare(c); }
    else if (sb < ndraw + npi + nzp) job_z_prepare(c, sb - ndraw - npi);      // (does not read pi / alpha_3: z_proposal.hpp)
    else job_chi_normals(c, sb - ndraw - npi - nzp);
#ifdef BFMMM_TIMELINE
    if (first_ && threadIdx.x == 0) c.dyn->stamps[57 + 2 * kind_] = wall_clock64();
#endif
    return;
  }
  const bool role_r = split && bx >= A;      // this workgroup forms r_a, H_aa theta_a (and, without the split, everything)
  const bool role_f = !role_r;               // this workgroup factorises
  const bool do_r = role_r || !split;
  const int a = role_r ? bx - A : bx;
  const int j = a / MD, mt = a - j * MD;
#ifdef BFMMM_TIMELINE
#define FST(i) do { if (bx == 1 && threadIdx.x == 0) c.dyn->stamps[48 + (i)] = wall_clock64(); } while (0)
#else
#define FST(i) do { } while (0)
#endif
  FST(0);
  const int AP = A * P, PS = P + 2 * BW + 1;
  // diagonal model (multivariate: G_i = I, prior (1 / tau) I or diag(gamma)): the precision is a diagonal, no P x P work
  // areas -- the launch then asks for 10 KB of LDS instead of 75, and the spare jobs of this kernel (which need none of it) fit
  // five to a CU instead of two
  const bool diag = (BW == 0) && d.BWP == 0;
  double* S = smem;                 // PP x PP : Prec (col-major, S[i + PP*k])
  double* X = S + PP * PP;          // PP x PP : U^-1, row-major X[i*PP + c]
  double* thp = diag ? smem : X + PP * PP;        // A x PS : theta_b with BW zero pads before and BW + 1 after
  double* part = thp + A * PS;      // A x P  : (H_ab theta_b)[p]
  double* zv = part + AP;           // PP
  double* hb2 = zv + PP;            // P x W  : rows of H_aa
  double* dsc = hb2 + P * W;        // 16     : delta(j, .)
  const bool upd_nu = (mt == 0) && (c.mask & U_NU);
  const bool upd_phi = (mt > 0) && (c.mask & U_PHI);
  const bool upd = upd_nu || upd_phi;
  const Dyn* dyn = c.dyn;
  // ---- everything this workgroup needs from global memory is requested up front, in one batch ----
  constexpr int MAXI = (BW > 5) ? 1 : 4;          // (b, p) items per thread and pass (wide band: a row is 64 doubles)
  v2d hreg[MAXI][BW + 1];
  double tval[MAXI];
  const bool upd_nu0 = (mt == 0) && (c.mask & U_NU), upd_phi0 = (mt > 0) && (c.mask & U_PHI);
  if (split && role_f && !(upd_nu0 || upd_phi0)) return;      // (a direction that is not sampled needs only its r workgroup)
  if (do_r) {
#pragma unroll
    for (int it = 0; it < MAXI; ++it) {
      const int e = min(tid + 256 * it, AP - 1);
      const int b = e / P, p = e - b * P;
      tval[it] = c.theta[(size_t)full_dir(d, b) * P + p];
      const v2d* blk = (const v2d*)(c.H2 + (size_t)hrow(d, a, b) * P * W);
#pragma unroll
      for (int k = 0; k <= BW; ++k) hreg[it][k] = blk[k * P + p];
    }
  } else {
    // factorisation workgroup: row p = tid of H_aa only
    const v2d* blk = (const v2d*)(c.H2 + (size_t)hrow(d, a, a) * P * W);
    const int p = min(tid, P - 1);
#pragma unroll
    for (int k = 0; k <= BW; ++k) hreg[0][k] = blk[k * P + p];
  }
  const double tv0 = do_r ? c.tvec[a * P + min(tid >> 3, P - 1)] : 0.0;      // t_a[p] of the r-reduction's first pass
  // prior entries of the band of Prec this thread will build: element (p, p + t), t <= BWP (two per thread at most)
  constexpr int NPRI = (BW > 5) ? ((BWWIDE + 1) * PP + 255) / 256 : 2;
  double pri[NPRI];
#pragma unroll
  for (int u = 0; u < NPRI; ++u) {
    const int e = tid + 256 * u, t = e / PP, p = e - t * PP, q = p + t;
    const bool in = t <= d.BWP && q < P;
    const int pc = min(p, P - 1), qc = min(q, P - 1);
    double v = 0.0;
    if (mt == 0) v = d.mv ? 0.0 : c.Pmat[pc + (size_t)P * qc];
    else v = c.gamma[j + (size_t)K * (pc + (size_t)P * (mt - 1))];
    pri[u] = (in && (mt == 0 || t == 0)) ? v : 0.0;
  }
  const double dlt = (mt > 0 && tid < M) ? c.delta[j + (size_t)K * tid] : 1.0;
  const double f = dyn->beta / dyn->sigma2;
  const double tau_j = dyn->tau[j];
  if (do_r) for (int x = tid; x < A * PS; x += 256) thp[x] = 0.0;
  if (role_f && upd && tid >= 64 && tid < 64 + P) {     // the direction's normal variates, while the loads are in flight
    const RngKey key = make_key(c.seed, c.chain, dyn->iter, dyn->tt_step);
    const uint32_t idx0 = (mt == 0) ? (uint32_t)(j * P) : (uint32_t)((j * M + (mt - 1)) * P);
    zv[tid - 64] = rnorm(key, (mt == 0) ? UPD_NU : UPD_PHI, idx0 + (uint32_t)(tid - 64));
  }
  FST(1);
  __syncthreads();
  FST(2);
  if (!do_r) {
    // factorisation workgroup of a split launch: the rows of H_aa straight to the precision's work area
    if (tid < P) {
#pragma unroll
      for (int k = 0; k <= BW; ++k) { hb2[tid * W + 2 * k] = hreg[0][k].x; hb2[tid * W + 2 * k + 1] = hreg[0][k].y; }
    }
    if (tid < 16) dsc[tid] = dlt;
    __syncthreads();
  }
  if (do_r) {
#pragma unroll
  for (int it = 0; it < MAXI; ++it) {
    const int e = tid + 256 * it;
    if (e < AP) { const int b = e / P, p = e - b * P; thp[b * PS + BW + p] = tval[it]; }
  }
  for (int e = tid + 256 * MAXI; e < AP; e += 256) {     // beyond the batched part (more than 1024 elements)
    const int b = e / P, p = e - b * P;
    thp[b * PS + BW + p] = c.theta[(size_t)full_dir(d, b) * P + p];
  }
  if (tid < 16) dsc[tid] = dlt;
  __syncthreads();
  // ---- (H_ab theta_b)[p] for every b; the rows of H_aa are kept for the precision matrix ----
  for (int base = 0; base < AP; base += 256 * MAXI) {
    if (base > 0) {                 // more than 1024 elements: further passes reload their rows (rare)
#pragma unroll
      for (int it = 0; it < MAXI; ++it) {
        const int e = min(base + tid + 256 * it, AP - 1);
        const int b = e / P, p = e - b * P;
        const v2d* blk = (const v2d*)(c.H2 + (size_t)hrow(d, a, b) * P * W);
#pragma unroll
        for (int k = 0; k <= BW; ++k) hreg[it][k] = blk[k * P + p];
      }
    }
#pragma unroll
    for (int it = 0; it < MAXI; ++it) {
      const int e = base + tid + 256 * it;
      if (e < AP) {
        const int b = e / P, p = e - b * P;
        const double* tb = thp + b * PS + p;        // tb[k] = theta_b[p + k - BW]
        double v = 0.0;
#pragma unroll
        for (int k = 0; k <= BW; ++k) v += hreg[it][k].x * tb[2 * k] + hreg[it][k].y * tb[2 * k + 1];
        part[e] = v;
        if (b == a) {
#pragma unroll
          for (int k = 0; k <= BW; ++k) { hb2[p * W + 2 * k] = hreg[it][k].x; hb2[p * W + 2 * k + 1] = hreg[it][k].y; }
        }
      }
    }
  }
  __syncthreads();
  FST(3);
  // ---- r_a = t_a - sum_b H_ab theta_b : 8 lanes per p, fixed summation order ----
  for (int p0 = 0; p0 < P; p0 += 32) {
    const int p = p0 + (tid >> 3), g = tid & 7;
    double acc = 0.0;
    if (p < P)
      for (int b = g; b < A; b += 8) acc += part[b * P + p];
    acc = dpp_add<0xB1>(acc);
    acc = dpp_add<0x4E>(acc);
    acc = dpp_add<0x141>(acc);
    const double tvp = (p0 == 0) ? tv0 : c.tvec[a * P + min(p, P - 1)];
    if (p < P && g == 0) {
      c.rvec[a * P + p] = tvp - acc;
      c.hq[a * P + p] = part[a * P + p];
    }
  }
  FST(4);
  }      // do_r
  if (role_r || !upd) return;
  // prior scale: tau_j (nu) or tilde_tau(j, m) = prod_{m' <= m} delta(j, m') (BFMMM.h:1514-1519)
  double tt = 1.0;
  for (int m2 = 0; m2 < mt; ++m2) tt *= dsc[m2];
  if (diag) {
    // C = diag(1 / d_p), chol_lower(C) = diag(1 / sqrt(d_p)), L z likewise (the same estimate + two Newton steps as
    // factor_core's diagonal branch, so the factor is the same function of the pivot).  d_p > 0 always: the prior term is.
    double* Cg = c.Cmat + (size_t)a * P * P;
    for (int e = tid; e < P * P; e += 256) {
      const int p = e % P, q = e / P;
      double cv = 0.0;
      if (p == q) {
        double dk = f * hb2[p * W + BW];
        if (mt == 0) dk += d.mv ? 1.0 / tau_j : tau_j * c.Pmat[p + (size_t)P * p];     // UpdateNu.h:197 (MV) / :66
        else dk += tt * c.gamma[j + (size_t)K * (p + (size_t)P * (mt - 1))];            // UpdatePhi.h:76-78
        double rk = __builtin_amdgcn_rsq(dk);
        rk = rk * (1.5 - (0.5 * dk) * (rk * rk));
        rk = rk * (1.5 - (0.5 * dk) * (rk * rk));
        if (!(dk > 0.0)) atomicOr(&c.dyn->status, 1u);
        cv = rk * rk;
        c.Lz[(size_t)a * P + p] = rk * zv[p];
      }
      Cg[q + (size_t)P * p] = cv;
    }
    FST(6);
    return;
  }
  // only the band of Prec is read by the factorisation (factor_core): (BWP + 1) x P entries
  auto build_prec = [&](bool full) {     // full: the whole symmetric matrix (pseudo-inverse route)
    for (int e = tid; e < PP * PP; e += 256) X[e] = 0.0;
    if (BW > 5 || full) {             // the dense factorisation / the Jacobi rotations read all of S
      for (int e = tid; e < PP * PP; e += 256) S[e] = (!full && e % PP == e / PP) ? 1.0 : 0.0;
      __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < NPRI; ++u) {
      const int e = tid + 256 * u, t = e / PP, p = e - t * PP, q = p + t;
      if (t <= d.BWP && q < P) {
        double v = (t <= BW) ? f * hb2[p * W + BW + t] : 0.0;
        if (mt == 0) v += d.mv ? ((t == 0) ? 1.0 / tau_j : 0.0) : tau_j * pri[u];    // UpdateNu.h:197 (MV) / :66
        else v += tt * pri[u];                                                       // UpdatePhi.h:76-78 (diagonal)
        S[p + PP * q] = v;
        if (BW > 5 || full) S[q + PP * p] = v;
      }
    }
    __syncthreads();
  };
  build_prec(false);
  FST(5);
  double* wkp = dsc + 16;             // 4 PP + 2 doubles: scratch of the pseudo-inverse route
  // (the factor L itself is not stored: the sweep needs only C_a and L_a z_a, and the store sat at the end of this kernel's
  //  critical path)
  if (factor_core<PP>(S, X, zv, P, d.BWP, c.Cmat + (size_t)a * P * P, nullptr, c.Lz + (size_t)a * P, tid, nullptr)) {
    // singular to working accuracy (e.g. a cluster without members: Prec = tau P_mat): arma::pinv + the eigen route of
    // arma::mvnrnd in the reference (UpdateNu.h:67-69), factor_pinv here
    build_prec(true);
    factor_pinv<PP>(S, X, zv, P, c.Cmat + (size_t)a * P * P, nullptr, c.Lz + (size_t)a * P, tid, wkp);
  }
  FST(6);
}

// ---------------------------------------------------------------------------------------------
// k_sweep: the sequentially dependent Gaussian block draws of one Gibbs iteration
// (updatePhi: j outer, m inner, UpdatePhi.h:40-84; then updateNu, UpdateNu.h:39-70) and sigma^2
// (UpdateSigma.h:22-58).  One workgroup of 1024 threads; two barriers per draw:
//   phase A  theta_a <- C_a rhs_a + (L z)_a                          (8 threads per row)
//   phase B  r_b -= H_ba (theta_a_new - theta_a_old) for every b     (one thread per element)
//            and, for the next direction a', rhs_a' = (beta/sigma^2)(r_a' + H_a'a' theta_a')
// The column blocks H_{.,a} and C_a of the NEXT step are fetched into registers while the
// current step runs and parked in the other half of an LDS double buffer, so no step waits on
// global memory.
// ---------------------------------------------------------------------------------------------
constexpr int SW_THREADS = 1024;
constexpr int NPF = 6;            // staged doubles per thread and step: A*LG + P*P <= 6144

__device__ inline int step_dir(const Dims& d, int s, int n_phi) {
  if (s < n_phi) {
    const int j = s / d.M, m = s - j * d.M;
    return j * d.MD + m + 1;
  }
  return (s - n_phi) * d.MD;
}

__global__ __launch_bounds__(SW_THREADS) void k_sweep(Ctx c0, int direct) {
  const Ctx c = chain_view(c0);      // chain blockIdx.z of the batch
  TIMELINE(c, 4);
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const Dims& d = c.d;
  const int P = d.P, A = d.A, K = d.K, M = d.M, MD = d.MD, BW = d.BW, LG = d.LG;
  const int tid = threadIdx.x;
  Dyn* dyn = c.dyn;
  const int AP = A * P;
  const int PS = P + 2 * BW;                 // padded vector stride
  double* th = smem;                         // A x PS (zero pads)
  double* tv = th + A * PS;                  // A x P
  double* r = tv + AP;                       // A x P
  double* hq = r + AP;                       // A x P   H_aa theta_a
  double* lz = hq + AP;                      // A x P   L_a z_a
  double* rhs = lz + AP;                     // PMAX
  double* dlp = rhs + PMAX;                  // PMAX + 2*BWWIDE (zero pads)
  double* red = dlp + PMAX + 2 * BWWIDE;     // 32
  // direct != 0: the column blocks and C_a do not fit the LDS double buffer (A*LG + P*P > 6144 doubles or the total
  // beyond 160 KB): no staging, every step reads them from L2 (slower per step, but no size limit)
  // diagonal model (multivariate: G_i = I and priors I/tau, diag(tilde_tau gamma)): every H block and every C_a is
  // diagonal, so only the diagonal of C_a is staged and phase A is one multiply per coordinate
  const bool diag = (BW == 0 && d.BWP == 0);
  const int csz = diag ? P : P * P;
  const int pf_len = direct ? 0 : A * LG + csz;
  double* pbuf0 = red + 32;
  double* pbuf1 = pbuf0 + pf_len;
  int* htab = (int*)(pbuf1 + pf_len);        // A x A : H row of block (b, a)
  const uint32_t slot = dyn->slot;
  const uint32_t mask = c.mask;
  const double beta = dyn->beta;
  const double f = beta / dyn->sigma2;
  if (tid == 0) { dyn->iter_hyper = dyn->iter; dyn->slot_hyper = slot; }
  const RngKey key = make_key(c.seed, c.chain, dyn->iter, dyn->tt_step);
  const int n_phi = ((mask & U_PHI) && MD > 1) ? K * M : 0;
  const int n_nu = (mask & U_NU) ? K : 0;
  const int n_steps = n_phi + n_nu;

  // the standard gamma variate of the sigma^2 draw was produced by a spare k_factor workgroup (job_hyper_draws)
  const double sig_g = (mask & U_SIGMA) ? c.gstd[hyper_gstd_count(d)] : 0.0;
  for (int e = tid; e < A * PS; e += SW_THREADS) {
    const int b = e / PS, pp = e - b * PS - BW;
    th[e] = (pp >= 0 && pp < P) ? c.theta[(size_t)full_dir(d, b) * P + pp] : 0.0;
  }
  for (int e = tid; e < AP; e += SW_THREADS) { tv[e] = c.tvec[e]; r[e] = c.rvec[e]; hq[e] = c.hq[e]; lz[e] = c.Lz[e]; }
  for (int e = tid; e < PMAX + 2 * BWWIDE; e += SW_THREADS) dlp[e] = 0.0;
  for (int e = tid; e < A * A; e += SW_THREADS) htab[e] = hrow(d, e / A, e % A);
  __syncthreads();
  // per-thread prefetch map: element e of [ H column blocks | C ] of a step
  int pf_b[NPF], pf_off[NPF];
#pragma unroll
  for (int k = 0; k < NPF; ++k) {
    const int e = tid + SW_THREADS * k;
    pf_b[k] = -2; pf_off[k] = 0;
    if (e < A * LG) { pf_b[k] = e / LG; pf_off[k] = e - pf_b[k] * LG; }
    else if (e < pf_len) { pf_b[k] = -1; pf_off[k] = e - A * LG; }
  }
  double preg[NPF];
  // branch-free: every lane always loads from a valid address (dummy lanes re-read H[0]) so that the
  // NPF loads of a step are issued back to back instead of one memory latency apart
  auto pf_load = [&](int a) {
#pragma unroll
    for (int k = 0; k < NPF; ++k) {
      const int hb = max(pf_b[k], 0);
      const size_t offH = (size_t)htab[hb * A + a] * LG + pf_off[k];
      const size_t offC = (size_t)a * P * P + (diag ? (size_t)pf_off[k] * (P + 1) : (size_t)pf_off[k]);
      const double* src = (pf_b[k] == -1) ? (c.Cmat + offC) : (c.H + ((pf_b[k] >= 0) ? offH : 0));
      preg[k] = *src;
    }
  };
  auto pf_store = [&](double* buf) {
#pragma unroll
    for (int k = 0; k < NPF; ++k)
      if (pf_b[k] >= -1) buf[tid + SW_THREADS * k] = preg[k];
  };
  if (n_steps > 0) {
    // H and C were written by other XCDs (k_pg_reduce, k_factor): a first touch costs a trip to memory, several times a
    // step of the chain.  Touch both once with fire-and-forget loads (one 4-byte load per 128-byte line) so that the
    // per-step prefetch only sees L2 hits.
    {
      int w0 = 0;
      auto touch = [&](const double* src, size_t count) {
        const size_t nl = (count * 8 + 127) / 128;
        for (size_t x = tid; x < nl; x += SW_THREADS) {
          const uint32_t o = (uint32_t)(x * 128);
          asm volatile("global_load_dword %0, %1, %2" : "+v"(w0) : "v"(o), "s"(src));
        }
      };
      touch(c.H, (size_t)d.R * LG);
      touch(c.Cmat, (size_t)A * P * P);
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(w0) :: "memory");
    }
    const int a0 = step_dir(d, 0, n_phi);
    if (!direct) { pf_load(a0); pf_store(pbuf0); }
    if (tid < P) rhs[tid] = f * (r[a0 * P + tid] + hq[a0 * P + tid]);
  }
  __syncthreads();
  for (int st = 0; st < n_steps; ++st) {
    const int a = step_dir(d, st, n_phi);
    const bool more = st + 1 < n_steps;
    const int an = more ? step_dir(d, st + 1, n_phi) : -1;
    const double* buf = (st & 1) ? pbuf1 : pbuf0;
    const double* Cg = direct ? c.Cmat + (size_t)a * P * P : buf + (size_t)A * LG;
    if (more && !direct) pf_load(an);
    // phase A: new = C rhs + L z
    if (diag) {
      if (tid < P) {
        const double cd = direct ? Cg[(size_t)tid * (P + 1)] : Cg[tid];
        const double nw = cd * rhs[tid] + lz[a * P + tid];
        dlp[BW + tid] = nw - th[a * PS + BW + tid];
        th[a * PS + BW + tid] = nw;
      }
    } else
    for (int p = tid >> 3; p < P; p += SW_THREADS / 8) {
      const int seg = tid & 7;
      double acc = 0.0;
      for (int q = seg; q < P; q += 8) acc += Cg[p + P * q] * rhs[q];
      acc += __shfl_xor(acc, 1, 8);
      acc += __shfl_xor(acc, 2, 8);
      acc += __shfl_xor(acc, 4, 8);
      if (seg == 0) {
        const double nw = acc + lz[a * P + p];
        dlp[BW + p] = nw - th[a * PS + BW + p];
        th[a * PS + BW + p] = nw;
      }
    }
    lds_barrier();
    // phase B: r_b -= H_ba dl ; hq_a ; next rhs
    for (int e = tid; e < AP; e += SW_THREADS) {
      const int b = e / P, p = e - b * P;
      const double* Hb = direct ? c.H + (size_t)htab[b * A + a] * LG : buf + (size_t)b * LG;
      const double* dl = dlp + BW + p;
      double v = Hb[p] * dl[0];
      for (int dd = 1; dd <= BW; ++dd) v += Hb[dd * P + p] * dl[dd] + Hb[dd * P + p - dd] * dl[-dd];
      const double rn = r[e] - v;
      r[e] = rn;
      if (b == a) hq[e] += v;                       // H_aa theta_a follows theta_a
      if (b == an) rhs[p] = f * (rn + hq[e]);
    }
    if (more && !direct) pf_store((st & 1) ? pbuf0 : pbuf1);
    lds_barrier();
  }
  // ---------------- sigma^2 (updateSigma, UpdateSigma.h:22-58) ---------------------------------
  if (mask & U_SIGMA) {
    // RSS = YY - sum_a theta_a'(t_a + r_a), fixed-order reduction
    // (covariate-adjusted: YY is replaced by sum_i yy_i - 2 o_i's_i + o_i'G_i o_i, block partials of k_curve_z)
    double acc = 0.0;
    for (int e = tid; e < AP; e += SW_THREADS) {
      const int b = e / P, p = e - b * P;
      acc += th[b * PS + BW + p] * (tv[e] + r[e]);
    }
    if (d.D > 0)
      for (int e = tid; e < c.nblk_curve; e += SW_THREADS) acc -= c.yyp_part[e];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == SW_THREADS - 1) {
      double q = 0.0;
      for (int w = 0; w < SW_THREADS / 64; ++w) q += red[w];
      const double rss = (d.D > 0) ? -q : (c.YY - q);
      const bool tempered = (dyn->tt_step != 0);
      const double b = (tempered ? (beta / 2) * rss : 0.5 * rss) + c.h.beta_0;
      const double s2 = 1.0 / (sig_g * (1.0 / b));
      dyn->sigma2 = s2;
      dyn->rss = rss;
      c.c_sigma[slot] = s2;
    }
  } else if (tid == 0) {
    c.c_sigma[slot] = dyn->sigma2;
  }
  // ---------------- publish theta and its chain slots -------------------------------------------
  double* s_nu = c.c_nu + (size_t)slot * K * P;
  double* s_phi = c.c_Phi + (size_t)slot * K * P * M;
  for (int e = tid; e < AP; e += SW_THREADS) {
    const int b = e / P, p = e - b * P;
    const int jj = b / MD, mt = b - jj * MD;
    const double v = th[b * PS + BW + p];
    c.theta[(size_t)(jj * (M + 1) + mt) * P + p] = v;
    if (mt == 0) s_nu[jj + (size_t)K * p] = v;
    else s_phi[jj + (size_t)K * (p + (size_t)P * (mt - 1))] = v;
  }
  if (MD == 1)
    for (int e = tid; e < K * P * M; e += SW_THREADS) {
      const int k = e % K, pm = e / K, p = pm % P, m = pm / P;
      s_phi[e] = c.theta[(size_t)(k * (M + 1) + m + 1) * P + p];
    }
}

// ---------------------------------------------------------------------------------------------
// k_sweep_diag: the sweep of the DIAGONAL model (multivariate: G_i = I, priors I/tau and diag(tilde_tau gamma)).
// Every H block and every C_a is diagonal, so the coordinates p are independent scalar Gauss-Seidel chains: 8 lanes
// per coordinate share its A directions (direction b lives on lane b % 8, slot b / 8) and keep r_b, H_bb theta_b, C_a,
// L z, theta in registers; a step is a handful of FMAs, an 8-lane DPP sum that broadcasts delta, and the H entries of
// the next step requested one step ahead.  No LDS hand-off and no barrier inside the sweep.  r follows the general
// kernel's definition (r_a = t_a - sum_b H_ab theta_b over all b).  One workgroup of 8 P threads; A <= 64.
// ---------------------------------------------------------------------------------------------
constexpr int DG_RPL_MAX = 8;      // directions per lane, at most (k_sweep_diag<DG_RPL>: DG_RPL = ceil(A / 8), the slots a lane really has)

// SCALAR_BLOCKS: a compile-time switch (the two sources of a step's H column must not meet in one load: a selected LDS-or-global
// pointer is a FLAT load)
template <int DG_RPL, bool SCALAR_BLOCKS>
__global__ __launch_bounds__(512) void k_sweep_diag(Ctx c0) {
  const Ctx c = chain_view(c0);      // chain blockIdx.z of the batch
  TIMELINE(c, 4);
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const Dims& d = c.d;
  const int P = d.P, A = d.A, K = d.K, M = d.M, MD = d.MD, LG = d.LG;
  const int tid = threadIdx.x, nthr = blockDim.x;
  Dyn* dyn = c.dyn;
  double* red = smem;                          // 16
  double* sH = red + 16;                       // A x A : multivariate model: the blocks are scalars, H_{b,a} = sH[b A + a] I
  int* htab = (int*)(sH + A * A);              // A x A : element offset of block (b, a) in H
  int* sdir = htab + A * A;
  const uint32_t slot = dyn->slot;
  const uint32_t mask = c.mask;
  const double beta = dyn->beta;
  const double f = beta / dyn->sigma2;
  if (tid == 0) { dyn->iter_hyper = dyn->iter; dyn->slot_hyper = slot; }
#ifdef DG_STAMPS
#define DST(k) do { if (tid == 0) dyn->stamps[40 + (k)] = clock64(); } while (0)
#else
#define DST(k) do { } while (0)
#endif
  DST(0);
  const int n_phi = ((mask & U_PHI) && MD > 1) ? K * M : 0;
  const int n_nu = (mask & U_NU) ? K : 0;
  const int n_steps = n_phi + n_nu;
  for (int x = tid; x < A * A; x += nthr) htab[x] = hrow(d, x / A, x % A) * LG;
  for (int x = tid; x < n_steps + 2; x += nthr) sdir[x] = step_dir(d, min(x, max(n_steps - 1, 0)), n_phi);
  const int p = min(tid >> 3, P - 1), g = tid & 7;
  const bool live = (tid >> 3) < P;
  double sig_g = (mask & U_SIGMA) ? c.gstd[hyper_gstd_count(d)] : 0.0;
  uint32_t tt_step0 = dyn->tt_step;
  // (both are needed by one lane at the very end: requested here, with the rest of the set-up loads, or they are two trips to
  //  memory after the last step)
  asm volatile("" : "+v"(sig_g), "+v"(tt_step0));
  double r[DG_RPL], hq[DG_RPL], cd[DG_RPL], lz[DG_RPL], th[DG_RPL], tv[DG_RPL];
#pragma unroll
  for (int j = 0; j < DG_RPL; ++j) {
    const int b = min(g + 8 * j, A - 1);
    const bool on = g + 8 * j < A;
    const int e = b * P + p;
    const double rv = c.rvec[e], hv = c.hq[e], cv = c.Cmat[(size_t)b * P * P + (size_t)p * (P + 1)], lv = c.Lz[e];
    const double t0 = c.theta[(size_t)full_dir(d, b) * P + p], t1 = c.tvec[e];
    r[j] = on ? rv : 0.0; hq[j] = on ? hv : 0.0; cd[j] = on ? cv : 0.0; lz[j] = on ? lv : 0.0; th[j] = on ? t0 : 0.0; tv[j] = on ? t1 : 0.0;
  }
  __syncthreads();
  DST(1);
  // Multivariate model (G_i = I): every block of H is a multiple of the identity, so the A^2 scalars are read ONCE into LDS
  // and a step's H_{b,a} is an LDS read instead of a request to L2 two steps ahead (the steps of this kernel were bound by
  // that latency: 36 steps took 39 us)
  constexpr bool scalar_blocks = SCALAR_BLOCKS;      // (the launcher passes d.mv)
  if (scalar_blocks) {
    for (int x = tid; x < A * A; x += nthr) sH[x] = c.H[(size_t)htab[x]];
    __syncthreads();
  }
  // H was written by another XCD (k_pg_reduce): touch it once (one 4-byte load per 128-byte line, fire and forget) so
  // that the per-step requests are L2 hits; they are issued two steps ahead (three register sets, compile-time indexed)
  if (n_steps > 0 && !scalar_blocks) {
    int w0 = 0;
    const size_t nl = ((size_t)d.R * LG * 8 + 127) / 128;
    for (size_t x = tid; x < nl; x += nthr) {
      const uint32_t o = (uint32_t)(x * 128);
      asm volatile("global_load_dword %0, %1, %2" : "+v"(w0) : "v"(o), "s"(c.H));
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(w0) :: "memory");
  }
  DST(2);
  double hs[3][DG_RPL];
  auto fetch = [&](auto which, int a) {       // H_{b,a}[p] for this lane's directions
    constexpr int Q = decltype(which)::value;
    if constexpr (scalar_blocks) {
#pragma unroll
      for (int j = 0; j < DG_RPL; ++j) hs[Q][j] = sH[min(g + 8 * j, A - 1) * A + a];
    } else {
#pragma unroll
      for (int j = 0; j < DG_RPL; ++j) hs[Q][j] = c.H[(size_t)htab[min(g + 8 * j, A - 1) * A + a] + p];
    }
  };
  // a: this step's direction, a2: the direction two steps ahead (its H column is requested now).  Branch-free: the draw is
  // evaluated for all of a lane's slots and the step's slot selected (a taken branch costs ~40 clk here, eight of them a step
  // were half of it).
  auto step = [&](int a, int a2, auto which) {
    constexpr int Q = decltype(which)::value;
    fetch(std::integral_constant<int, (Q + 2) % 3>{}, a2);
    const int owner = a & 7, js = a >> 3;
    const bool mine_lane = g == owner;
    double dsel = 0.0;
#pragma unroll
    for (int j = 0; j < DG_RPL; ++j) {
      const double nw = cd[j] * (f * (r[j] + hq[j])) + lz[j];
      const bool mine = mine_lane && j == js;
      dsel = mine ? nw - th[j] : dsel;
      th[j] = mine ? nw : th[j];
    }
    dsel = dpp_add<0xB1>(dsel);
    dsel = dpp_add<0x4E>(dsel);
    const double delta = dpp_add<0x141>(dsel);           // broadcast to the 8 lanes of the coordinate
#pragma unroll
    for (int j = 0; j < DG_RPL; ++j) {
      const double v = hs[Q][j] * delta;
      r[j] = (g + 8 * j < A) ? r[j] - v : r[j];
      hq[j] = (mine_lane && j == js) ? hq[j] + v : hq[j];
    }
  };
  if (n_steps > 0) {
    using Q0 = std::integral_constant<int, 0>;
    using Q1 = std::integral_constant<int, 1>;
    using Q2 = std::integral_constant<int, 2>;
    // the directions of the steps travel in scalar registers, read from LDS three steps ahead of their use
    auto dir_at = [&](int s) { return __builtin_amdgcn_readfirstlane(sdir[min(s, n_steps + 1)]); };
    int a0 = dir_at(0), a1 = dir_at(1), a2 = dir_at(2), a3 = dir_at(3), a4 = dir_at(4);
    fetch(Q0{}, a0);
    fetch(Q1{}, a1);
    int s = 0;
    for (; s + 2 < n_steps; s += 3) {
      const int b0 = dir_at(s + 5), b1 = dir_at(s + 6), b2 = dir_at(s + 7);
      step(a0, a2, Q0{}); step(a1, a3, Q1{}); step(a2, a4, Q2{});
      a0 = a3; a1 = a4; a2 = b0; a3 = b1; a4 = b2;
    }
    if (s < n_steps) { step(a0, a2, Q0{}); ++s; }
    if (s < n_steps) { step(a1, a3, Q1{}); ++s; }
  }
  DST(3);
  // ---------------- sigma^2 (updateSigma, UpdateSigma.h:127-165 MV) ---------------------------------
  if (mask & U_SIGMA) {
    double acc = 0.0;                                       // RSS = YY - sum_a theta_a'(t_a + r_a)
    if (live) {
#pragma unroll
      for (int j = 0; j < DG_RPL; ++j) acc += th[j] * (tv[j] + r[j]);
    }
    if (d.D > 0)
      for (int x = tid; x < c.nblk_curve; x += nthr) acc -= c.yyp_part[x];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) {
      double qs = 0.0;
      for (int w = 0; w < nthr / 64; ++w) qs += red[w];
      const double rss = (d.D > 0) ? -qs : (c.YY - qs);
      const bool tempered = (tt_step0 != 0);
      const double bb = (tempered ? (beta / 2) * rss : 0.5 * rss) + c.h.beta_0;
      const double s2 = 1.0 / (sig_g * (1.0 / bb));
      dyn->sigma2 = s2;
      dyn->rss = rss;
      c.c_sigma[slot] = s2;
    }
  } else if (tid == 0) {
    c.c_sigma[slot] = dyn->sigma2;
  }
  DST(4);
  // ---------------- publish theta and its chain slots -------------------------------------------
  double* s_nu = c.c_nu + (size_t)slot * K * P;
  double* s_phi = c.c_Phi + (size_t)slot * K * P * M;
  if (live) {
#pragma unroll
    for (int j = 0; j < DG_RPL; ++j) {
      const int b = g + 8 * j;
      if (b < A) {
        const int jj = b / MD, mt = b - jj * MD;
        c.theta[(size_t)(jj * (M + 1) + mt) * P + p] = th[j];
        if (mt == 0) s_nu[jj + (size_t)K * p] = th[j];
        else s_phi[jj + (size_t)K * (p + (size_t)P * (mt - 1))] = th[j];
      }
    }
  }
  if (MD == 1) {
    __syncthreads();
    for (int x = tid; x < K * P * M; x += nthr) {
      const int k = x % K, pm = x / K, pp = pm % P, m = pm / P;
      s_phi[x] = c.theta[(size_t)(k * (M + 1) + m + 1) * P + pp];
    }
  }
}

__device__ inline const double* ptr_off(const double* base, uint32_t byte_off) {   // uniform base + 32-bit lane offset
  return (const double*)((const char*)base + byte_off);
}

// Software-managed prefetch of the chain wave.  The loads below are inline assembly, so the compiler neither tracks them nor
// inserts s_waitcnt for them (its own placement drained the queue every step); swc_wait_* are the matching waits: "at most N
// younger loads may still be in flight" -- vmcnt retires in issue order, and the chain wave issues the same loads in the same
// order at every step.  The loaded registers pass through the wait as read-write operands, so no use can be scheduled above it.
// The destination of a load is a read-write operand too: the register stays allocated to the variable across the load (the
// compiler believes inline assembly completes synchronously; a write-only destination that is dead until its next definition
// could be handed out as a temporary while the load is still in flight).
template <int BW> struct SweepH { v2d h[BW + 1]; };

// ---------------------------------------------------------------------------------------------
// k_sweep_chain: the register-resident sweep with the dependent chain in ONE wave (round 3; replaces k_sweep_fast, whose
// step cost ~2000 clk: two cross-wave hand-offs through LDS flags and a 14-wave barrier sat on the chain of every step).
//
// Wave 0 is the chain.  Lane (p, h), p = lane >> 1, h = lane & 1.  Step st with direction a = a_st:
//   mat-vec   theta_a <- C_a rhs + L_a z_a      lane (p, h) holds C_a(p, 16 h .. 16 h + 15) in registers (prefetched two
//             steps ahead from L2), reads rhs(16 h ..) from LDS (broadcast), 16 FMAs on four accumulators, one DPP add
//             joins the halves; delta_st -> LDS
//   band dot  the h = 0 lanes hold r of the direction of step st + 1, the h = 1 lanes r of the direction of step st + 2:
//             r -= H_{., a} delta_st from prefetched rows of H; the h = 0 lanes publish the next rhs; then the h = 1 value
//             moves to the h = 0 lane (DPP) and the h = 1 lanes pick up the row of step st + 3
// so both hand-offs of a step (delta -> band rows, rhs -> mat-vec rows) are LDS write / read pairs of the SAME wave: LDS
// executes a wave's operations in order, there is no flag, no poll and no barrier on the chain.
// The other waves ("row threads", thread 64 + rk P + p owns element p of the direction updated at step rk) do what is
// off the chain: r_rk -= H delta_s for s <= rk - 3 (the chain applies the last two deltas itself), hand the row over, and
// the row's term of the residual sum of squares, delta'(H_aa delta - 2 r_a), one step late.
// Hand-offs between the chain and the row threads carry their own validity: every slot (delta of step s, row of rank rk,
// r_a before its step) is written ONCE per launch into a slot of its own that starts as a NaN with a payload no arithmetic
// produces (SW_SENT); a reader that finds the sentinel reads again (bounded: a spin that runs out sets status bit 2 and goes
// on, so the grid always drains).  No ordering between different LDS locations is assumed anywhere.
// ---------------------------------------------------------------------------------------------
constexpr unsigned long long SW_SENT = 0x7FF8DEADBEEF5A5AULL;
constexpr int SWC_SPIN_LIMIT = 1 << 15;
constexpr unsigned SWC_STATUS_SPIN = 4u;      // status bit of a hand-off spin that ran out (bfmmm_capi.hip::run_impl reports it by name)
constexpr int SWC_THREADS = 448;            // chain wave + A ceil(P / 2) <= 384 row threads (two rows each): two waves per SIMD, 256 VGPRs

__device__ inline double lds_ld(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ inline bool is_sent(double v) { return (unsigned long long)__double_as_longlong(v) == SW_SENT; }
__device__ inline double sw_sent() { return __longlong_as_double((long long)SW_SENT); }
template <int CTRL>
__device__ inline double dpp_get(double v) {      // the value of another lane of the quad
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}

struct SwcC { v2d v[8]; };
template <int N>
__device__ inline void swc_wait_c(SwcC& s) {      // at most N younger loads still in flight; the registers pass through
  asm volatile("s_waitcnt vmcnt(%8)" : "+v"(s.v[0]), "+v"(s.v[1]), "+v"(s.v[2]), "+v"(s.v[3]), "+v"(s.v[4]), "+v"(s.v[5]), "+v"(s.v[6]), "+v"(s.v[7]) : "n"(N));
}
template <int N, int BW>
__device__ inline void swc_wait_h(SweepH<BW>& s) {
  if constexpr (BW == 0) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(s.h[0]) : "n"(N));
  if constexpr (BW == 1) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(s.h[0]), "+v"(s.h[1]) : "n"(N));
  if constexpr (BW == 2) asm volatile("s_waitcnt vmcnt(%3)" : "+v"(s.h[0]), "+v"(s.h[1]), "+v"(s.h[2]) : "n"(N));
  if constexpr (BW == 3) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(s.h[0]), "+v"(s.h[1]), "+v"(s.h[2]), "+v"(s.h[3]) : "n"(N));
  if constexpr (BW == 4) asm volatile("s_waitcnt vmcnt(%5)" : "+v"(s.h[0]), "+v"(s.h[1]), "+v"(s.h[2]), "+v"(s.h[3]), "+v"(s.h[4]) : "n"(N));
  if constexpr (BW == 5) asm volatile("s_waitcnt vmcnt(%6)" : "+v"(s.h[0]), "+v"(s.h[1]), "+v"(s.h[2]), "+v"(s.h[3]), "+v"(s.h[4]), "+v"(s.h[5]) : "n"(N));
}
__device__ inline void sweep_ld16v(v2d& r, const double* sbase, uint32_t voff) {
  asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(r) : "v"(voff), "s"(sbase));
}

// Step tables of k_sweep_chain, built once per (MD, mask) of a run (bfmmm_capi.hip::run_impl), shared by the chains of a batch:
//   ent [2 (A + 2)] int4 : what lane half h of the chain wave needs for step st (entry 2 st + h):
//                          .x byte offset in H2 of block (a_{st + 1 + h}, a_st), .y byte offset of C_{a_st} in Cmat, .z a_st P
//   hstp[A x A]          : hstp[b A + s] = byte offset in H2 of block (b, a_s), a_s = direction of step s
// (their integer divisions cost the sweep kernel 1.6 us of its set-up when it built them itself)
size_t sweep_tab_ints(int A) { return (size_t)A * A + 8 * ((size_t)A + 2); }
__host__ __device__ inline int sweep_n_phi(const Dims& d, uint32_t mask) { return ((mask & U_PHI) && d.MD > 1) ? d.K * d.M : 0; }
__host__ __device__ inline int sweep_n_nu(const Dims& d, uint32_t mask) { return (mask & U_NU) ? d.K : 0; }

__global__ void k_sweep_tables(Ctx c, int* tab) {
  const Dims& d = c.d;
  const int A = d.A, P = d.P, W = 2 * d.BW + 2;
  const int n_phi = sweep_n_phi(d, c.mask), n_steps = n_phi + sweep_n_nu(d, c.mask);
  int4* ent = (int4*)tab;
  int* hstp = tab + 8 * (A + 2);
  for (int x = threadIdx.x; x < 2 * (A + 2); x += blockDim.x) {
    const int st = x >> 1, hh = x & 1;
    const int a = step_dir(d, min(st, max(n_steps - 1, 0)), n_phi), bb = step_dir(d, min(st + 1 + hh, max(n_steps - 1, 0)), n_phi);
    ent[x] = make_int4(hrow(d, bb, a) * P * W * 8, a * P * P * 8, a * P, 0);
  }
  for (int x = threadIdx.x; x < A * A; x += blockDim.x)
    hstp[x] = hrow(d, x / A, step_dir(d, min(x % A, max(n_steps - 1, 0)), n_phi)) * P * W * 8;
}
void launch_sweep_tables(const Ctx& c, hipStream_t st) { hipLaunchKernelGGL(k_sweep_tables, dim3(1), dim3(256), 0, st, c, (int*)c.sweep_tab); }

// direction that owns rank rk: ranks follow the order of the steps (Phi sweep j outer, m inner, then the nu sweep), the
// directions a sweep does not update come last
__device__ inline int rank_dir(const Dims& d, int rk, int n_phi, int n_nu) {
  const int n_steps = n_phi + n_nu;
  if (rk < n_steps) return step_dir(d, rk, n_phi);
  const int x = rk - n_steps;
  if (n_phi == 0 && n_nu == 0) return x;                       // nothing is updated: rank = direction
  if (n_nu == 0) return x * d.MD;                               // the nu directions of a Phi-only sweep
  if (d.MD == 1) return x;                                      // (no Phi directions at all)
  const int jx = x / (d.MD - 1);                                // the Phi directions of a nu-only sweep
  return jx * d.MD + 1 + (x - jx * (d.MD - 1));
}

template <int BW>
__global__ __launch_bounds__(SWC_THREADS) void k_sweep_chain(Ctx c0) {
  const Ctx c = chain_view(c0);      // chain blockIdx.z of the batch
  TIMELINE(c, 4);
#ifdef BFMMM_TIMELINE
  if (threadIdx.x == 0) { unsigned id; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id)); c.dyn->stamps[38] = ((c.dyn->stamps[38] << 4) | (id & 0xf)) & 0xFFFFFFFFFULL; }
#endif
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const Dims& d = c.d;
  const int P = d.P, A = d.A, K = d.K, M = d.M, MD = d.MD;
  constexpr int W = 2 * BW + 2;              // doubles per row of an H2 block
  constexpr int DLS = 32 + 2 * BW + 2;       // one delta slot: BW zero pads | 32 | BW + 2 zero pads
  constexpr int NC = 8, NH = BW + 1;         // loads per step of the chain wave: C pieces, H row pieces
  const int tid = threadIdx.x, nthr = blockDim.x;
  Dyn* dyn = c.dyn;
  const int AP = A * P, PH = (P + 1) >> 1;
  const int APe = (AP + 1) & ~1;             // (16-byte alignment of what follows)
  double* th = smem;                         // A x P   theta, direction-major (theta_0, then the new values)
  double* lzs = th + APe;                    // A x P   L_a z_a
  double* hqs = lzs + APe;                   // A x P   H_aa theta_a(0)
  double* pick = hqs + APe;                  // A x P   rank-major: r of rank rk with the deltas of the steps <= rk - 3 applied
  double* rbef = pick + APe;                 // A x P   rank-major: r of rank rk just before its step
  double* rhs = rbef + APe;                  // 32 (zero beyond P)
  double* dl = rhs + 32;                     // A slots of DLS: delta of step s
  double* red = dl + A * DLS;                // 16
  int4* ent = (int4*)(red + 16);             // 2 (A + 2): what lane half h of the chain needs for step st
  int* hstp = (int*)(ent + 2 * (A + 2));     // A x A : hstp[b A + s] = byte offset of block (b, a_s) in H2, a_s = direction of step s
  const uint32_t mask = c.mask;
  const int n_phi = sweep_n_phi(d, mask), n_nu = sweep_n_nu(d, mask);
  const int n_steps = n_phi + n_nu;
  const bool chainw = tid < 64;              // wave-uniform
#ifndef SWC_NOPRIO
  if (chainw) __builtin_amdgcn_s_setprio(3);
#endif
  // ---- ONE round of global loads: the run's step tables, the iteration's scalars, and every thread's rows ----
  const int4* gent = (const int4*)c.sweep_tab;
  const int* ghstp = c.sweep_tab + 8 * (A + 2);
  // row thread 64 + rk LRK + pp (pp < PH) owns rows p0 = 2 pp and p0 + 1 of the direction updated at step rk; a rank takes
  // LRK = 4, 8 or 16 lanes, so that the threads of a rank sit in ONE 16-lane DPP row (the band entries below the diagonal come
  // from the neighbouring threads' registers, see the row threads' loads)
  const int lrk_log = (PH <= 4) ? 2 : (PH <= 8) ? 3 : 4, LRK = 1 << lrk_log;
  const int e2i = max(tid - 64, 0);
  const int ppr = e2i & (LRK - 1);
  const bool isB = tid >= 64 && (e2i >> lrk_log) < A && ppr < PH;
  const int rk = min(e2i >> lrk_log, A - 1), p0 = 2 * min(ppr, PH - 1);
  const bool two = p0 + 1 < P;               // (odd P: the last thread of a rank owns one row)
  const int p1 = two ? p0 + 1 : p0;
  const int b = rank_dir(d, rk, n_phi, n_nu);
  const int fd = full_dir(d, b);
  const int eb = b * P + p0, er = rk * P + p0;     // element of the direction-major / rank-major vectors
  // the chain wave: its first table entries straight into registers, so that its first C / H loads are in flight during the set-up
  const int hch = (tid >> 1) & 1;
  int4 ge0 = make_int4(0, 0, 0, 0), ge1 = ge0, ge2 = ge0;
  if (chainw && n_steps > 0) { ge0 = gent[hch]; ge1 = gent[2 + hch]; ge2 = gent[2 * min(2, n_steps + 1) + hch]; }
  const uint32_t slot = dyn->slot;
  const double beta = dyn->beta;
  const double sig2 = dyn->sigma2;
  const double sig_g = (mask & U_SIGMA) ? c.gstd[hyper_gstd_count(d)] : 0.0;
  double r_0 = 0.0, r_1 = 0.0, rss_acc = 0.0;      // rss_acc: this thread's share of RSS - YY
  double hq_0 = 0.0, hq_1 = 0.0, tv_0 = 0.0, tv_1 = 0.0, t_0 = 0.0, t_1 = 0.0, l_0 = 0.0, l_1 = 0.0;
  if (!chainw) {
    r_0 = c.rvec[eb]; r_1 = c.rvec[b * P + p1];
    hq_0 = c.hq[eb]; hq_1 = c.hq[b * P + p1]; tv_0 = c.tvec[eb]; tv_1 = c.tvec[b * P + p1];
    t_0 = c.theta[(size_t)fd * P + p0]; t_1 = c.theta[(size_t)fd * P + p1]; l_0 = c.Lz[eb]; l_1 = c.Lz[b * P + p1];
  }
  // (the tables go through registers: two entries of each kind per thread cover A <= 21; the loops take the rest)
  const int n_ent = 2 * (A + 2), n_hs = A * A;
  int4 ge_a = make_int4(0, 0, 0, 0);
  int gh_a = 0, gh_b = 0;
  if (tid < n_ent) ge_a = gent[tid];
  if (tid < n_hs) gh_a = ghstp[tid];
  if (tid + nthr < n_hs) gh_b = ghstp[tid + nthr];
  if (tid == 0) { dyn->iter_hyper = dyn->iter; dyn->slot_hyper = slot; }
  const double f = beta / sig2;
  // ---- LDS initialisation while the loads are in flight
  if (tid < 32) rhs[tid] = 0.0;
  for (int x = tid; x < A * DLS; x += nthr) {
    const int k = x % DLS;
    dl[x] = (k >= BW && k < BW + P) ? sw_sent() : 0.0;
  }
  if (tid < n_ent) ent[tid] = ge_a;
  for (int x = tid + nthr; x < n_ent; x += nthr) ent[x] = gent[x];
  if (tid < n_hs) hstp[tid] = gh_a;
  if (tid + nthr < n_hs) hstp[tid + nthr] = gh_b;
  for (int x = tid + 2 * nthr; x < n_hs; x += nthr) hstp[x] = ghstp[x];
  if (isB) {
    th[eb] = t_0; lzs[eb] = l_0; hqs[eb] = hq_0;
    rss_acc = -(t_0 * (tv_0 + r_0));       // RSS(theta_0) = YY - theta_0'(t + r_0)
    // ranks 0 .. 2 (no delta to apply first) and the directions that are not updated are handed over at once
    const bool now = (rk <= 2 || rk >= n_steps);
    pick[er] = now ? r_0 : sw_sent();
    rbef[er] = (rk == 0) ? r_0 : sw_sent();
    if (two) {
      th[eb + 1] = t_1; lzs[eb + 1] = l_1; hqs[eb + 1] = hq_1;
      rss_acc -= t_1 * (tv_1 + r_1);
      pick[er + 1] = now ? r_1 : sw_sent();
      rbef[er + 1] = (rk == 0) ? r_1 : sw_sent();
    }
  }
  TSTAMP0(c, 28);
  // the chain wave's per-lane constants and its first loads (C and H of steps 0 and 1) before the barrier
  const int pp = tid >> 2, g = tid & 3, h = hch;
  const int pbl = 2 * pp + (tid & 1), pb = min(pbl, P - 1);
  const bool wr = (h == 0) && (pbl < P);
  const int ppc = min(2 * pp, max(P - 2, 0));           // first row of the pair, clamped (a pair never leaves its matrix row)
  uint32_t coff[NC];
#pragma unroll
  for (int u = 0; u < NC; ++u) coff[u] = (uint32_t)(ppc + P * min(8 * g + u, P - 1)) * 8u;     // C(q, p), C(q, p + 1) = C(p, q), C(p + 1, q)
  const bool sel_y = (tid & 1) || (2 * pp < P && ppc < 2 * pp);      // odd P: the last row sits in the SECOND slot of the clamped pair
  uint32_t hoffk[NH];
#pragma unroll
  for (int k = 0; k < NH; ++k) hoffk[k] = (uint32_t)(k * P + pb) * 16u;      // piece k of row pb (h2_index)
  auto issueC = [&](SwcC& s, int cbase) {
#ifndef SWC_NO_LOADS
#pragma unroll
    for (int u = 0; u < NC; ++u) sweep_ld16v(s.v[u], c.Cmat, (uint32_t)cbase + coff[u]);
#endif
  };
  auto issueH = [&](SweepH<BW>& s, int hbase) {
#ifndef SWC_NO_LOADS
#pragma unroll
    for (int k = 0; k < NH; ++k) sweep_ld16v(s.h[k], c.H2, (uint32_t)hbase + hoffk[k]);
#endif
  };
  SwcC c0s = {}, c1s = {};
  SweepH<BW> h0s = {}, h1s = {};
  if (chainw && n_steps > 0) {
    // (the compiler waits for ge0 / ge1 here; everything issued above is older and already on its way)
    // the issue order of the steady state: C(st), H(st), C(st + 1), H(st + 1)
    // (s_nop 13 / 14 and 11 / 12 below are MARKERS for tools/isa_check.py, which verifies in the shipped code object that no
    //  instruction touches a prefetch destination between its load and the counted wait: tests/test_isa_sweep_chain.py)
    asm volatile("s_nop 13");
    issueC(c0s, ge0.y); issueH(h0s, ge0.x); issueC(c1s, ge1.y); issueH(h1s, ge1.x);
    asm volatile("s_nop 14");
  }
  __syncthreads();
#ifdef BFMMM_TIMELINE
  if (tid == 0) dyn->stamps[24] = wall_clock64();
  unsigned long long n_spin = 0;
#endif
  if (n_steps > 0 && chainw) {
    // ================= the chain wave =================
    // mat-vec role of lane l: rows 2 pp, 2 pp + 1 (pp = l >> 2), columns q = 8 g .. 8 g + 7 (g = l & 3)
    // band role of lane l   : row pb = 2 pp + (l & 1), rank half h = (l >> 1) & 1
    asm volatile("s_nop 11");
    int4 e0 = ge0, e1 = ge1, e2 = ge2;
    // rank 0 is complete: its rhs; the h = 0 lanes then hold rank 1, the h = 1 lanes rank 2
    {
      const double r0 = pick[pb], hq0 = hqs[e0.z + pb];
      if (wr) rhs[pb] = f * (r0 + hq0);
    }
    double r = pick[min((1 + h) * P + pb, AP - 1)];
    double lz0 = lzs[e0.z + pb], th0 = th[e0.z + pb];
    double hq1 = hqs[e1.z + pb];                    // H_aa theta_a of the direction of step st + 1
    asm volatile("" ::: "memory");
    // (LAST: the odd last step behind the pair loop issues no further prefetch -- nothing follows it)
    auto step = [&](auto last_tag, int st, SwcC& cs, SweepH<BW>& hs) {
      constexpr bool LAST = decltype(last_tag)::value;
      // ---- mat-vec operands first: rhs was written at the end of the previous step
      const v2d* rv = (const v2d*)(rhs + 8 * g);
      v2d x[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) x[u] = rv[u];
      // ---- small reads for later in this step / the next ones (no address here depends on a read of this step)
      const int4 e3 = ent[2 * min(st + 3, n_steps + 1) + h];
      const double lz1 = lzs[e1.z + pb], th1 = th[e1.z + pb];
      const double hq2 = hqs[e2.z + pb];
      // the row the h = 1 lanes take over for this step's band dot: rank st + 2, which the row threads hand over with the
      // deltas of the steps <= st - 1 applied (published a whole step ago); the read is checked just before its use
#ifdef SWC_NO_HELPERS
      const bool want_pick = false;
#else
      const bool want_pick = (st >= 1) && (st + 2 < n_steps);
#endif
      const double* pk_ptr = pick + min((st + 2) * P + pb, AP - 1);
      double pk = lds_ld(pk_ptr);
      swc_wait_c<NH + NC + NH>(cs);
      // cs.v[u] = (C(p, q), C(p + 1, q)), q = 8 g + u: four chains
      v2d s0 = cs.v[0] * x[0].x, s1 = cs.v[1] * x[0].y;
#pragma unroll
      for (int u = 1; u < 4; ++u) { s0 += cs.v[2 * u] * x[u].x; s1 += cs.v[2 * u + 1] * x[u].y; }
      if constexpr (!LAST) issueC(cs, e2.y);
      const v2d ss = s0 + s1;
      // sum over the four column groups of the quad (every lane of the quad ends with the same two sums), then this lane's row
      double a_x = ss.x, a_y = ss.y;
      a_x += dpp_get<0xB1>(a_x); a_y += dpp_get<0xB1>(a_y);       // quad_perm [1, 0, 3, 2]
      a_x += dpp_get<0x4E>(a_x); a_y += dpp_get<0x4E>(a_y);       // quad_perm [2, 3, 0, 1]
      const double acc = sel_y ? a_y : a_x;
      const double nw = acc + lz0;
      const double dlt = nw - th0;
      double* dls = dl + st * DLS;
      if (wr) { dls[BW + pb] = dlt; th[e0.z + pb] = nw; }
      asm volatile("" ::: "memory");
      // ---- band dot: the rows of the next two directions take delta_st
      double dv[W];
#pragma unroll
      for (int k = 0; k < W; ++k) dv[k] = dls[pb + k];
      if (h == 1 && want_pick) {
        int spins = 0;
#pragma nounroll
        while (is_sent(pk)) {
          if (++spins > SWC_SPIN_LIMIT) {       // (the extra memory operation would shift the counted waits: drain)
            atomicOr(&dyn->status, SWC_STATUS_SPIN);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            break;
          }
          pk = lds_ld(pk_ptr);
#ifdef BFMMM_TIMELINE
          ++n_spin;
#endif
        }
        r = pk;
      }
      swc_wait_h<(LAST ? 0 : NC) + NH + NC, BW>(hs);       // younger loads in flight: C(st + 2) (not issued by the last step), H(st + 1), C(st + 1)
      double v0 = hs.h[0].x * dv[0], v1 = hs.h[0].y * dv[1];
#pragma unroll
      for (int k = 1; k <= BW; ++k) { v0 += hs.h[k].x * dv[2 * k]; v1 += hs.h[k].y * dv[2 * k + 1]; }      // last .y is the zero pad
      if constexpr (!LAST) issueH(hs, e2.x);
      r -= (v0 + v1);
      if (wr && st + 1 < n_steps) { rhs[pb] = f * (r + hq1); rbef[(st + 1) * P + pb] = r; }
      asm volatile("" ::: "memory");
      // ---- the h = 1 value (rank st + 2, delta_st applied) moves to the h = 0 lane; the h = 1 lanes take the next row over
      //      at the next step
      const double rsw = dpp_get<0x4E>(r);            // quad_perm [2, 3, 0, 1]: lane l ^ 2 (same row, other half)
      if (h == 0) r = rsw;
      e0 = e1; e1 = e2; e2 = e3; lz0 = lz1; th0 = th1; hq1 = hq2;
    };
    // (pairs of steps in the loop and an odd last step behind it -- not `if (st + 1 < n_steps)` inside the loop: the control-flow
    //  graph then has no path "first half, skipped second half, first half again", which never runs but which a static check of
    //  the counted waits would have to assume: tools/isa_check.py)
    int st = 0;
    for (; st + 1 < n_steps; st += 2) {
      step(std::false_type{}, st, c0s, h0s);
      step(std::false_type{}, st + 1, c1s, h1s);
    }
    if (st < n_steps) step(std::true_type{}, st, c0s, h0s);
    // drain the prefetches of the (clamped) tail before their registers are reused
    swc_wait_c<0>(c0s); swc_wait_c<0>(c1s);
    swc_wait_h<0, BW>(h0s); swc_wait_h<0, BW>(h1s);
    asm volatile("s_nop 12");
#ifdef BFMMM_TIMELINE
    if (tid == 2) { dyn->stamps[25] = wall_clock64(); dyn->stamps[26] = n_spin; }
#endif
  } else if (n_steps > 0) {
    // ================= the row threads =================
    const bool live = isB && rk < n_steps;
    int rk_hi = live ? rk : -1;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) rk_hi = max(rk_hi, __shfl_xor(rk_hi, o, 64));
    int rk_lo = live ? rk : (1 << 20);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) rk_lo = min(rk_lo, __shfl_xor(rk_lo, o, 64));
    const int* hs_b = hstp + b * A;                  // hs_b[s] = byte offset of block (b, a_s) in H2 (= twice its offset in H)
    // The rows of H_{b, a_s} come from the band-packed array H (G(p, p + d) at [d P + p]: half the bytes of the H2 rows; the sweep
    // is bound by the bytes one compute unit can pull through its L1): a thread loads the UPPER band entries of its two rows, one
    // 16-byte piece per diagonal, and takes the entries below the diagonal -- G(p, p - d) = G(p - d, p), which the thread owning
    // row p - d has loaded -- from its neighbours' registers on the DPP path (the threads of a rank share a 16-lane row).
    struct H2r { v2d u[BW + 1]; };                   // u[d] = (G(p0, p0 + d), G(p0 + 1, p0 + 1 + d))
    // (every lane loads at every step of its wave: predicating the loads on "the step is one of this lane's" made the kernel
    //  1.6 us slower -- the compiler then waits for each conditional group on its own)
    auto is_mine = [&](int s) { return live && (s <= rk - 3 || s == rk); };
    auto loadH = [&](H2r& s, int off) {
#ifndef SWC_HELPER_NOLOAD
      const char* blk = (const char*)c.H + ((uint32_t)off >> 1);
#pragma unroll
      for (int dd = 0; dd <= BW; ++dd) s.u[dd] = *(const v2d*)(blk + (size_t)(dd * P + p0) * 8);
#endif
    };
    // (va, vb) = (H_{b, a_s} delta)[p0], [p0 + 1] with dv[k] = delta[p0 - BW + k]
    auto band2 = [&](const H2r& hc, const double (&dv)[W + 1], double& va, double& vb) {
      double a0 = hc.u[0].x * dv[BW], a1 = 0.0, b0 = hc.u[0].y * dv[BW + 1], b1 = 0.0;
#pragma unroll
      for (int dd = 1; dd <= BW; ++dd) {
        // lower entries: row p0 takes G(p0 - dd, p0), row p0 + 1 takes G(p0 + 1 - dd, p0 + 1)
        double l0, l1;
        if (dd & 1) {
          constexpr int dummy = 0; (void)dummy;
          const int sh0 = (dd + 1) >> 1, sh1 = (dd - 1) >> 1;
          const double y = hc.u[dd].y, x = hc.u[dd].x;
          const double g0 = (sh0 == 1) ? dpp_get<0x111>(y) : (sh0 == 2) ? dpp_get<0x112>(y) : dpp_get<0x113>(y);
          const double g1 = (sh1 == 0) ? x : (sh1 == 1) ? dpp_get<0x111>(x) : dpp_get<0x112>(x);
          l0 = (ppr >= sh0) ? g0 : 0.0;
          l1 = (ppr >= sh1) ? g1 : 0.0;
        } else {
          const int sh = dd >> 1;
          const double y = hc.u[dd].y, x = hc.u[dd].x;
          const double g0 = (sh == 1) ? dpp_get<0x111>(x) : dpp_get<0x112>(x);
          const double g1 = (sh == 1) ? dpp_get<0x111>(y) : dpp_get<0x112>(y);
          l0 = (ppr >= sh) ? g0 : 0.0;
          l1 = (ppr >= sh) ? g1 : 0.0;
        }
        a0 += hc.u[dd].x * dv[BW + dd]; a1 += l0 * dv[BW - dd];
        b0 += hc.u[dd].y * dv[BW + 1 + dd]; b1 += l1 * dv[BW + 1 - dd];
      }
      va = a0 + a1; vb = b0 + b1;
    };
    // The steps the WAVE walks (wave-uniform): 0 .. rk_hi - 3 (some lane's lagging update), then rk_lo .. rk_hi (some lane's
    // own term).  Every lane loads the rows of H_{b, a_s} for every step of the wave, one step ahead, so that the two register
    // sets alternate without a copy; a lane uses the result when the step is one of its own: s <= rk - 3 or s == rk.
    auto next_w = [&](int s) { return (s + 1 <= rk_hi - 3 || s + 1 >= rk_lo) ? s + 1 : rk_lo; };
    auto process = [&](int s, const H2r& hc) {
      const bool mine = is_mine(s);
      const double* dls = dl + s * DLS;
      {
        int spins = 0;
#pragma nounroll
        while (is_sent(lds_ld(dls + BW))) {
#ifdef SWC_SLEEP
          __builtin_amdgcn_s_sleep(SWC_SLEEP);
#else
          __builtin_amdgcn_s_sleep(1);
#endif
          if (++spins > SWC_SPIN_LIMIT) { atomicOr(&dyn->status, SWC_STATUS_SPIN); break; }
        }
      }
      asm volatile("" ::: "memory");      // (compiler ordering only: an acquire fence would also wait for this wave's loads in flight)
      double dv[W + 1], va, vb;
      int tries = 0;
#pragma nounroll
      do {
#pragma unroll
        for (int k = 0; k <= W; ++k) dv[k] = lds_ld(dls + p0 + k);
        band2(hc, dv, va, vb);
        // a delta slot still holding the sentinel makes the sum a NaN: read again (the first element was seen, the rest of
        // the chain's one store instruction follows within cycles)
        bool bad = false;
        if (va != va || vb != vb) {
#pragma unroll
          for (int k = 0; k <= W; ++k) bad = bad || is_sent(dv[k]);
        }
        if (!__builtin_amdgcn_ballot_w64(mine && bad)) break;
      } while (++tries < SWC_SPIN_LIMIT);
      if (tries >= SWC_SPIN_LIMIT) atomicOr(&dyn->status, SWC_STATUS_SPIN);
      if (mine) {
        if (s == rk) {
          // RSS(theta + delta e_a) - RSS(theta) = delta'(H_aa delta - 2 r_a), r_a taken before the step
          double rb0 = lds_ld(rbef + er), rb1 = lds_ld(rbef + rk * P + p1);
          int spins = 0;
#pragma nounroll
          while (is_sent(rb0) || is_sent(rb1)) {
            if (++spins > SWC_SPIN_LIMIT) { atomicOr(&dyn->status, SWC_STATUS_SPIN); break; }
            rb0 = lds_ld(rbef + er); rb1 = lds_ld(rbef + rk * P + p1);
          }
          rss_acc += dv[BW] * (va - 2.0 * rb0);
          if (two) rss_acc += dv[BW + 1] * (vb - 2.0 * rb1);
        } else {
          r_0 -= va; r_1 -= vb;
          if (s == rk - 3) { pick[er] = r_0; if (two) pick[er + 1] = r_1; }
        }
      }
    };
#ifdef SWC_NO_HELPERS
    rk_hi = -1;
#endif
    if (rk_hi >= 0) {
      H2r hA = {}, hB = {};
      int s = (rk_hi >= 3) ? 0 : rk_lo;
      int s1 = next_w(s);
      int off1 = hs_b[min(s1, n_steps - 1)];           // offsets are read one step before the loads that use them
      loadH(hA, hs_b[s]);
      while (true) {
        int s2 = next_w(s1);
        int off2 = hs_b[min(s2, n_steps - 1)];
        loadH(hB, off1);
        process(s, hA);
        if (s1 > rk_hi) break;
        s = s2; s2 = next_w(s2);
        off1 = hs_b[min(s2, n_steps - 1)];
        loadH(hA, off2);
        process(s1, hB);
        if (s > rk_hi) break;
        s1 = s2;
      }
    }
  }
  __syncthreads();
  TSTAMP0(c, 30);
  // ---------------- sigma^2 (updateSigma, UpdateSigma.h:22-58) ---------------------------------
  if (mask & U_SIGMA) {
    // RSS = YY + sum of the threads' shares (RSS(theta_0) - YY and the steps' increments), fixed-order reduction
    // (covariate-adjusted: YY is replaced by sum_i yy_i - 2 o_i's_i + o_i'G_i o_i, block partials of k_curve_z)
    double acc = isB ? -rss_acc : 0.0;
    if (d.D > 0)
      for (int x = tid; x < c.nblk_curve; x += nthr) acc -= c.yyp_part[x];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) {
      double qs = 0.0;
      for (int w = 0; w < nthr / 64; ++w) qs += red[w];
      const double rss = (d.D > 0) ? -qs : (c.YY - qs);
      const bool tempered = (dyn->tt_step != 0);
      const double bb = (tempered ? (beta / 2) * rss : 0.5 * rss) + c.h.beta_0;
      const double s2 = 1.0 / (sig_g * (1.0 / bb));
      dyn->sigma2 = s2;
      dyn->rss = rss;
      c.c_sigma[slot] = s2;
    }
  } else if (tid == 0) {
    c.c_sigma[slot] = dyn->sigma2;
  }
  // ---------------- publish theta and its chain slots -------------------------------------------
  double* s_nu = c.c_nu + (size_t)slot * K * P;
  double* s_phi = c.c_Phi + (size_t)slot * K * P * M;
  if (isB) {
    const int jj = b / MD, mt = b - jj * MD;
    for (int u = 0; u < (two ? 2 : 1); ++u) {
      const int pu = p0 + u;
      const double th_e = th[b * P + pu];
      c.theta[(size_t)fd * P + pu] = th_e;
      if (mt == 0) s_nu[jj + (size_t)K * pu] = th_e;
      else s_phi[jj + (size_t)K * (pu + (size_t)P * (mt - 1))] = th_e;
    }
  }
  if (MD == 1)
    for (int x = tid; x < K * P * M; x += nthr) {
      const int k = x % K, pm = x / K, pp = pm % P, m = pm / P;
      s_phi[x] = c.theta[(size_t)(k * (M + 1) + m + 1) * P + pp];
    }
}

// ---------------------------------------------------------------------------------------------
// k_loglik: calcLikelihood = sum_il dnorm(y_il; mean_il, sqrt(sigma2), log)  and end-of-iteration
// bookkeeping (advance the iteration counter / slot for graph replay).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_loglik(Ctx c0, int use_rss_part, int r_stored) {
  const Ctx c = chain_view(c0);      // chain blockIdx.z of the batch
  TIMELINE(c, 6);
  __shared__ double red[256];
  Dyn* dyn = c.dyn;
  const int tid = threadIdx.x;
  double rss = dyn->rss;
  if (use_rss_part) {
    double acc = 0.0;
    const int per = (c.nblk_curve + 255) / 256;
    for (int b = tid * per; b < min(c.nblk_curve, (tid + 1) * per); ++b) acc += c.rss_part[b];
    rss = block_sum256(acc, red);
  }
  if (tid == 0) {
    const double s2 = dyn->sigma2;
    double ll;
    if (c.d.mv)   // calcLikelihoodMV: (y_obs.n_cols / 2) is an integer division, CalculateLikelihood.h:155
      ll = -(double)c.d.n * ((c.d.P / 2) * log(2 * 3.14159265358979323846 * s2)) - (1 / (s2 * 2)) * rss;
    else
      ll = -(double)c.d.n_obs_total * (0.91893853320467274178 + log(sqrt(s2))) - rss / (2.0 * s2);
    dyn->rss = rss;
    dyn->loglik = ll;
    if (c.mask & U_LOGLIK) c.c_loglik[dyn->slot] = ll;
    dyn->iter += 1;
    dyn->slot = (r_stored > 0) ? ((dyn->iter - dyn->slot_base) % (uint32_t)r_stored) : dyn->iter - dyn->slot_base;
  }
}

// reduces a still-pending log-likelihood (launched once at the end of a run)
// status_out (host-mapped pinned memory, one word per chain; may be null): the chain's status word for bfmmm_run -- written
// from the run's last kernel, it saves the device-to-host copy a run used to queue behind its kernels (about 10 us of a call)
__global__ __launch_bounds__(256) void k_loglik_flush(Ctx c0, uint32_t* status_out) {
  const Ctx c = chain_view(c0);      // chain blockIdx.z of the batch
  __shared__ double red[256];
  if (blockIdx.x == 1) {             // second workgroup: the last iteration's scalar job (Ctx::defer_hyper), beside the log-likelihood
    if (c.dyn->hyper_pending) {      // (dynamic LDS: HYPER_LDS_DOUBLES)
      job_hyper(c, false);
      __syncthreads();
      if (threadIdx.x == 0) c.dyn->hyper_pending = 0u;
    }
    return;
  }
  if (c.dyn->ll_pending) deferred_loglik(c, red);
  if (status_out && threadIdx.x == 0) {
    __hip_atomic_store(&status_out[blockIdx.z], c.dyn->status, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// broadcast the current value of blocks a sweep does not update into chain slots [s0, s1)
__global__ void k_fill_slots(double* chain, const double* cur, size_t len, int s0, int s1, size_t chain_bytes) {
  chain = ptr_shift(chain, blockIdx.z * chain_bytes);      // chain blockIdx.z of the batch
  cur = ptr_shift(cur, blockIdx.z * chain_bytes);
  const size_t total = len * (size_t)(s1 - s0);
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x)
    chain[(size_t)s0 * len + e] = cur[e % len];
}

// ---- host launchers -------------------------------------------------------------------------
void launch_pair_gram(const Ctx& c, int do_pg, int NKS, int KS, hipStream_t st) {
  const Dims& d = c.d;
  const int RS = d.K + d.MD + 1, RP = d.NZZ + d.NCC + 1, ncw = d.K + d.MD - 1;
  auto lds_doubles = [&](int G) {            // record columns + G chains' tables (G workgroups: 16 + G (RS + RP) rows; s workgroup: 16 CTS + G RS) + pair table
    return (size_t)(KS + 2) * std::max(16 + G * (RS + RP), d.CTS * 16 + G * RS) + 128;
  };
  // chains staged together (k_pair_gram): as many as 144 KB of LDS and 12 staged doubles per thread allow; covariate-adjusted
  // models stage s~_i per chain and keep one chain per group
  int G = 1;
  if (c.nch > 1 && d.D == 0 && d.RT < 8)      // (with eight or more row tiles per chain the waves are busy chain by chain)
    while (G < c.nch && lds_doubles(G + 1) * sizeof(double) <= 144 * 1024 && (size_t)(G + 1) * ncw * KS <= 12 * 512) ++G;
  const size_t lds = std::max(lds_doubles(G), (size_t)std::max(PI_ALPHA_LDS_DOUBLES, HYPER_LDS_DOUBLES)) * sizeof(double);
  if (c.nch > 1 && G > 1) hipLaunchKernelGGL((k_pair_gram<true, true>), dim3(d.CTG + 2, do_pg ? std::max(NKS, c.nch) : c.nch, 1), dim3(PG_THREADS), lds, st, c, KS, NKS, do_pg, G);
  else if (c.nch > 1) hipLaunchKernelGGL((k_pair_gram<true, false>), dim3(d.CTG + 2, do_pg ? std::max(NKS, c.nch) : c.nch, 1), dim3(PG_THREADS), lds, st, c, KS, NKS, do_pg, 1);
  else hipLaunchKernelGGL((k_pair_gram<false, false>), dim3(d.CTG + 2, do_pg ? NKS : 1, 1), dim3(PG_THREADS), lds, st, c, KS, NKS, do_pg, 1);
}

// geometry of k_pair_gram_pack for this (sub-)batch; returns false when the shape is outside its limits (the caller keeps k_pair_gram)
size_t pgp_lds_bytes(const PgPack& g) {
  const size_t lg = 2 * (16 * (size_t)PGP_RB + (size_t)g.NQG * 16 * g.SLS), ls = 2 * (16 * (size_t)PGP_RB + (size_t)g.NQS * 16 * g.SLS);
  return std::max(lg, ls) * sizeof(double);
}
bool pgp_geometry(const Dims& d, int nch, int KS, int NKS, PgPack& g) {
  if (d.mv || d.D > 0 || d.K > 4 || d.MD - 1 > 8 || d.P > 32 || (d.LREC & 1) || (KS & 15)) return false;
  g.KS = KS; g.NKS = NKS;
  g.NP2 = (d.LG + 31) / 32;
  g.TG = (nch * d.R + 15) / 16; g.TS = (nch * d.A + 15) / 16;
  const char* ew = getenv("BFMMM_PGP_WAVES");
  g.WPG = (ew && atoi(ew) == 8) ? 8 : (ew && atoi(ew) == 2) ? 2 : 4;
  const int PGP_WAVES = g.WPG, PGP_THREADS = 64 * g.WPG;
  g.NRG = (g.TG + PGP_WAVES - 1) / PGP_WAVES; g.NCG = (g.NP2 + 1) / 2; g.NWG_S = (g.TS + PGP_WAVES - 1) / PGP_WAVES;
  g.SLG = (d.NZZ + d.NCC + 1 + 1) & ~1; g.SLS = (d.K + d.MD + 1 + 1) & ~1;
  g.NQG = std::min(nch, (PGP_WAVES * 16 + d.R - 2) / d.R + 1); g.NQS = std::min(nch, (PGP_WAVES * 16 + d.A - 2) / d.A + 1);
  g.NTP = g.TG * 2 * g.NP2 + g.TS * 2;
  const int NV = d.K + d.MD - 1;
  if (std::max(g.NQG, g.NQS) * NV * 16 > PGP_NWV * PGP_THREADS) return false;      // weight values a thread stages per chunk
  return pgp_lds_bytes(g) <= 64 * 1024;
}
size_t pgp_pack_doubles(const PgPack& g) { return (size_t)g.NKS * g.NTP * 256; }

void launch_pair_gram_pack(const Ctx& c, const PgPack& g, double* pack, hipStream_t st) {
  const dim3 grid(g.NRG * g.NCG + g.NWG_S, g.NKS, 1);
  switch (g.WPG) {
    case 8: hipLaunchKernelGGL(k_pair_gram_pack<8>, grid, dim3(512), pgp_lds_bytes(g), st, c, g, pack); break;
    case 2: hipLaunchKernelGGL(k_pair_gram_pack<2>, grid, dim3(128), pgp_lds_bytes(g), st, c, g, pack); break;
    default: hipLaunchKernelGGL(k_pair_gram_pack<4>, grid, dim3(256), pgp_lds_bytes(g), st, c, g, pack); break;
  }
  const int nblk_red = (g.NTP * 256 * 4 + 255) / 256;
  hipLaunchKernelGGL(k_pg_reduce_pack, dim3(nblk_red), dim3(256), 0, st, c, g, pack);      // (the pi / alpha_3 job: a workgroup of k_factor, Ctx::pi_in_factor)
}

void launch_pg_reduce(const Ctx& c, int NKS, hipStream_t st) {
  const int nthreads = c.d.NT * 256 * 4;        // four lanes per element
  hipLaunchKernelGGL(k_pg_reduce, dim3((nthreads + 255) / 256, 1, c.nch), dim3(256), 0, st, c, NKS);
}

template <int PP>
static void launch_factor_pp(const Ctx& c, int grid, size_t lds, hipStream_t st) {
  switch (c.d.BW) {
    case 0: hipLaunchKernelGGL((k_factor<PP, 0>), dim3(c.nch * grid), dim3(256), lds, st, c); break;
    case 1: hipLaunchKernelGGL((k_factor<PP, 1>), dim3(c.nch * grid), dim3(256), lds, st, c); break;
    case 2: hipLaunchKernelGGL((k_factor<PP, 2>), dim3(c.nch * grid), dim3(256), lds, st, c); break;
    case 3: hipLaunchKernelGGL((k_factor<PP, 3>), dim3(c.nch * grid), dim3(256), lds, st, c); break;
    case 4: hipLaunchKernelGGL((k_factor<PP, 4>), dim3(c.nch * grid), dim3(256), lds, st, c); break;
    case 5: hipLaunchKernelGGL((k_factor<PP, 5>), dim3(c.nch * grid), dim3(256), lds, st, c); break;
    case BWMID: hipLaunchKernelGGL((k_factor<PP, BWMID>), dim3(c.nch * grid), dim3(256), lds, st, c); break;
    default: hipLaunchKernelGGL((k_factor<PP, BWWIDE>), dim3(c.nch * grid), dim3(256), lds, st, c); break;
  }
}

void launch_factor(const Ctx& c, hipStream_t st) {
  const int PP = (c.d.P <= 32) ? 32 : 64;
  const int W = 2 * c.d.BW + 2, PS = c.d.P + 2 * c.d.BW + 1;
  const bool diag = c.d.BW == 0 && c.d.BWP == 0;       // no P x P work areas (k_factor)
  const size_t lds = ((diag ? 0 : 2 * (size_t)PP * PP) + (size_t)c.d.A * PS + (size_t)c.d.A * c.d.P + PP + (size_t)c.d.P * W + 16 + 4 * PP + 2) * sizeof(double);
  const int n_draw = c.d.K * c.d.P * c.d.M + c.d.K * c.d.M + c.d.K + 4 * c.d.K + 1 + 8 * c.d.K;   // + sigma^2's gamma variate, A terms
  const int zcw = zprep_curves_per_wg(c.d.K);       // curves per workgroup of job_z_prepare (z_proposal.hpp)
  const int n_zprep = (c.mask & U_Z) ? (c.d.n + zcw - 1) / zcw : 0;      // (covariate-adjusted models too: the proposal does not see the data)
  const int n_znorm = ((c.mask & U_CHI) && c.d.MD > 1) ? (c.d.n * c.d.M + 255) / 256 : 0;
  const int n_pi = (c.mask & (U_PI | U_ALPHA3)) ? 1 : 0;          // next iteration's pi / alpha_3 tables (right behind the draws)
  const int grid = (diag ? 1 : 2) * c.d.A + (c.pi_in_factor ? 1 : 0) + (n_draw + 255) / 256 + n_zprep + n_znorm + n_pi;      // (k_factor: two workgroups per direction)
  if (PP == 32) launch_factor_pp<32>(c, grid, lds, st);
  else launch_factor_pp<64>(c, grid, lds, st);
}

int launch_sweep(const Ctx& c, hipStream_t st) {
  const Dims& d = c.d;
  if (d.BW == 0 && d.BWP == 0 && d.A <= 8 * DG_RPL_MAX && d.P <= 64) {        // diagonal model: independent scalar chains per coordinate
    const size_t lds = (16 + (size_t)d.A * d.A) * sizeof(double) + ((size_t)d.A * d.A + (size_t)d.K * (d.M + 1) + 8) * sizeof(int) + 16;
    const dim3 grid(1, 1, c.nch), block((8 * d.P + 63) / 64 * 64);
    switch ((d.A + 7) / 8) {
      case 1: if (d.mv) hipLaunchKernelGGL((k_sweep_diag<1, true>), grid, block, lds, st, c); else hipLaunchKernelGGL((k_sweep_diag<1, false>), grid, block, lds, st, c); break;
      case 2: if (d.mv) hipLaunchKernelGGL((k_sweep_diag<2, true>), grid, block, lds, st, c); else hipLaunchKernelGGL((k_sweep_diag<2, false>), grid, block, lds, st, c); break;
      case 3: if (d.mv) hipLaunchKernelGGL((k_sweep_diag<3, true>), grid, block, lds, st, c); else hipLaunchKernelGGL((k_sweep_diag<3, false>), grid, block, lds, st, c); break;
      case 4: if (d.mv) hipLaunchKernelGGL((k_sweep_diag<4, true>), grid, block, lds, st, c); else hipLaunchKernelGGL((k_sweep_diag<4, false>), grid, block, lds, st, c); break;
      case 5: if (d.mv) hipLaunchKernelGGL((k_sweep_diag<5, true>), grid, block, lds, st, c); else hipLaunchKernelGGL((k_sweep_diag<5, false>), grid, block, lds, st, c); break;
      case 6: if (d.mv) hipLaunchKernelGGL((k_sweep_diag<6, true>), grid, block, lds, st, c); else hipLaunchKernelGGL((k_sweep_diag<6, false>), grid, block, lds, st, c); break;
      case 7: if (d.mv) hipLaunchKernelGGL((k_sweep_diag<7, true>), grid, block, lds, st, c); else hipLaunchKernelGGL((k_sweep_diag<7, false>), grid, block, lds, st, c); break;
      default: if (d.mv) hipLaunchKernelGGL((k_sweep_diag<8, true>), grid, block, lds, st, c); else hipLaunchKernelGGL((k_sweep_diag<8, false>), grid, block, lds, st, c); break;
    }
    return 0;
  }
  const int swc_ph = (d.P + 1) / 2, swc_lrk = swc_ph <= 4 ? 4 : swc_ph <= 8 ? 8 : 16;      // lanes per rank of the row threads
  const size_t swc_lds = (5 * (((size_t)d.A * d.P + 1) & ~(size_t)1) + 32 + (size_t)d.A * (32 + 2 * d.BW + 2) + 16) * sizeof(double) + 2 * ((size_t)d.A + 2) * sizeof(int4) +
                         ((size_t)d.A * d.A + (size_t)d.A + 8) * sizeof(int) + 16;
  if (d.P <= 32 && d.A * swc_lrk <= SWC_THREADS - 64 && d.BW <= 5 && swc_lds <= 160 * 1024) {      // fast path: the chain in one wave
    const int nthr = 64 + (d.A * swc_lrk + 63) / 64 * 64;
    const size_t lds = swc_lds;       // beyond 64 KB at small P and many directions (P = 8, A >= 71): opted in by prepare_sweep_kernels
    switch (d.BW) {
      case 0: hipLaunchKernelGGL(k_sweep_chain<0>, dim3(1, 1, c.nch), dim3(nthr), lds, st, c); break;
      case 1: hipLaunchKernelGGL(k_sweep_chain<1>, dim3(1, 1, c.nch), dim3(nthr), lds, st, c); break;
      case 2: hipLaunchKernelGGL(k_sweep_chain<2>, dim3(1, 1, c.nch), dim3(nthr), lds, st, c); break;
      case 3: hipLaunchKernelGGL(k_sweep_chain<3>, dim3(1, 1, c.nch), dim3(nthr), lds, st, c); break;
      case 4: hipLaunchKernelGGL(k_sweep_chain<4>, dim3(1, 1, c.nch), dim3(nthr), lds, st, c); break;
      default: hipLaunchKernelGGL(k_sweep_chain<5>, dim3(1, 1, c.nch), dim3(nthr), lds, st, c); break;
    }
    return 0;
  }
  const bool diag = (d.BW == 0 && d.BWP == 0);
  size_t pf_len = (size_t)d.A * d.LG + (diag ? (size_t)d.P : (size_t)d.P * d.P);
  auto lds_for = [&](size_t pf) {
    const size_t doubles = (size_t)d.A * (d.P + 2 * d.BW) + 4 * (size_t)d.A * d.P + PMAX + PMAX + 2 * BWWIDE + 32 + 2 * pf;
    return doubles * sizeof(double) + (size_t)d.A * d.A * sizeof(int) + 16;
  };
  int direct = 0;
  if (pf_len > (size_t)NPF * SW_THREADS || lds_for(pf_len) > 160 * 1024) { direct = 1; pf_len = 0; }
  const size_t lds = lds_for(pf_len);
  if (lds > 160 * 1024) return 1;
  hipLaunchKernelGGL(k_sweep, dim3(1, 1, c.nch), dim3(SW_THREADS), lds, st, c, direct);
  return 0;
}

void launch_loglik_flush(const Ctx& c, hipStream_t st, uint32_t* status_out) {
  hipLaunchKernelGGL(k_loglik_flush, dim3(c.defer_hyper ? 2 : 1, 1, c.nch), dim3(256), (size_t)HYPER_LDS_DOUBLES * sizeof(double), st, c, status_out);
}

void launch_loglik(const Ctx& c, int use_rss_part, int r_stored, hipStream_t st) {
  hipLaunchKernelGGL(k_loglik, dim3(1, 1, c.nch), dim3(256), 0, st, c, use_rss_part, r_stored);
}

void prepare_sweep_kernels() {
  set_max_lds((const void*)k_sweep);
  set_max_lds((const void*)k_sweep_chain<0>); set_max_lds((const void*)k_sweep_chain<1>); set_max_lds((const void*)k_sweep_chain<2>);
  set_max_lds((const void*)k_sweep_chain<3>); set_max_lds((const void*)k_sweep_chain<4>); set_max_lds((const void*)k_sweep_chain<5>);
  set_max_lds((const void*)k_pair_gram<false, false>); set_max_lds((const void*)k_pair_gram<true, false>); set_max_lds((const void*)k_pair_gram<true, true>);
  set_max_lds((const void*)k_factor<32, 0>); set_max_lds((const void*)k_factor<64, 0>);
  set_max_lds((const void*)k_factor<32, 1>); set_max_lds((const void*)k_factor<64, 1>);
  set_max_lds((const void*)k_factor<32, 2>); set_max_lds((const void*)k_factor<64, 2>);
  set_max_lds((const void*)k_factor<32, 3>); set_max_lds((const void*)k_factor<64, 3>);
  set_max_lds((const void*)k_factor<32, 4>); set_max_lds((const void*)k_factor<64, 4>);
  set_max_lds((const void*)k_factor<32, 5>); set_max_lds((const void*)k_factor<64, 5>);
  set_max_lds((const void*)k_factor<32, BWMID>); set_max_lds((const void*)k_factor<64, BWMID>);
  set_max_lds((const void*)k_factor<32, BWWIDE>); set_max_lds((const void*)k_factor<64, BWWIDE>);
}

void launch_fill_slots(const Ctx& c, double* chain, const double* cur, size_t len, int s0, int s1, hipStream_t st) {
  if (s1 <= s0 || len == 0) return;
  hipLaunchKernelGGL(k_fill_slots, dim3(256, 1, c.nch), dim3(256), 0, st, chain, cur, len, s0, s1, c.chain_bytes);
}

}  // namespace bfmmm
