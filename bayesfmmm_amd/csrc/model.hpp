// Device-side data model of the MI355X Gibbs sampler (internal header, not part of the C ABI).
//
// Algebra: every data term of every full conditional of
// inst/include/BayesFMMM/Update*.h depends on curve i only through
//     G_i = B_i'B_i (P x P, banded with half-bandwidth = spline degree),  s_i = B_i'y_i,  yy_i = y_i'y_i.
// They are computed once (kernels_stats.hip) and kept resident in HBM as one record per curve:
//     rec_i = [ G_i band-packed (BW+1) x P, diagonal-major : rec[d*P + p] = G_i[p][p+d] | s_i (P) | yy_i | pad ]
// The sampled P-vectors are "directions" a = (j, mt): mt = 0 is nu_j, mt = m+1 is phi_{j,m};
// curve i sees direction a with weight  w_{a,i} = Z_ij * chit_{i,mt},  chit_{i,0} = 1, chit_{i,m+1} = chi_im,
// so that the fitted coefficient is  c_i = sum_a w_{a,i} theta_a.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bfmmm {

constexpr int KMAX = 8;     // clusters supported by the unrolled per-curve code (bfmmm_config.c[8] has room for as many)
constexpr int PMAX = 64;    // basis functions: one lane per basis function inside a curve group
constexpr int BWMAX = 5;    // spline degree (band half-width) instantiated
constexpr int BWWIDE = 31;  // the widest band instantiation (user-supplied / tensor-product bases, bfmmm_create_from_basis)
constexpr int BWMID = 15;   // bands 6 .. 15 (round 4: e.g. a 6 x 6 tensor-product basis of quadratic splines has band 14; padded to 31
                            // its records, pair-Gram columns and per-curve band registers were twice what the band needs)

// update mask bits, in the (fixed) order in which every reference driver applies them
// (BFMMM.h:1073-1107, 1253-1292, 1502-1553, 3741-3780, 3944-4010, 4809-4894)
enum : uint32_t {
  U_Z = 1u << 0, U_PI = 1u << 1, U_ALPHA3 = 1u << 2, U_PHI = 1u << 3, U_DELTA = 1u << 4, U_A = 1u << 5,
  U_GAMMA = 1u << 6, U_NU = 1u << 7, U_TAU = 1u << 8, U_SIGMA = 1u << 9, U_CHI = 1u << 10,
  U_ETA = 1u << 11, U_TAU_ETA = 1u << 12, U_XI = 1u << 13, U_DELTA_XI = 1u << 14, U_A_XI = 1u << 15,
  U_GAMMA_XI = 1u << 16, U_LOGLIK = 1u << 17
};

struct Hyper {
  double c[KMAX];
  double b, nu_1;
  double alpha1l, alpha2l, beta1l, beta2l;
  double a_Z_PM, a_pi_PM, var_alpha3, var_epsilon1, var_epsilon2;
  double alpha_nu, beta_nu, alpha_eta, beta_eta, alpha_0, beta_0;
};

// Scalars that change every iteration live in device memory so that a captured HIP graph
// can be replayed without re-recording kernel arguments.
struct Dyn {
  uint32_t iter;        // chain iteration index (RNG counter word) of the sweep being executed
  uint32_t slot;        // chain slot that sweep writes (iter - slot_base)
  uint32_t tt_step;     // tempered-transition sub-step (0 outside)
  uint32_t status;      // sticky error bits (1: precision matrix not positive definite)
  uint32_t iter_hyper;  // snapshot of iter / slot taken by the sweep kernel for k_curve_chi, whose scalar-job workgroup
  uint32_t slot_hyper;  //   advances iter / slot at its end while curve workgroups may still be running
  int32_t pend_dir;     // eta / Xi direction whose delta_cur has not been applied to c_i, g_i yet (-1: none)
  uint32_t ll_pending;  // a log-likelihood (slot ll_slot) is still to be reduced from the residual partial sums
  uint32_t ll_slot;
  uint32_t ll_use_part;
  uint32_t zprep_valid, zprep_iter, zprep_tt, zprep_chain;   // tag of the Z proposals prepared ahead (z_proposal.hpp)
  uint32_t znorm_valid, znorm_iter, znorm_tt;    // tag of the chi normals drawn ahead by spare k_factor workgroups
  uint32_t piprep_valid, piprep_iter;            // tag of the pi / alpha_3 tables prepared ahead
  uint32_t slot_base;   // slot = iter - slot_base: on-disk batches of r_stored_iters draws reuse the same slots (BFMMM.h:1680-1746)
  unsigned long long zprep_seed;
  double beta;          // temperature (1 = untempered)
  double sigma2;        // current sigma^2 (variance, as everywhere in the reference)
  double alpha3;
  double rss;           // residual sum of squares after the last sigma / chi pass
  double loglik;
  double pi[KMAX];
  double tau[KMAX];
  uint32_t pi_done;     // iteration + 1 whose pi / alpha_3 job has finished (the job as a workgroup of k_factor: Ctx::pi_in_factor)
  uint32_t hyper_pending;   // the scalar job (delta, A, gamma, tau) of the finished iteration is still to run (Ctx::defer_hyper)
  unsigned long long stamps[64];   // diagnostic kernel timeline (-DBFMMM_TIMELINE), 100 MHz wall clock
};

// The scalar head of Dyn, fetched in ONE batch of loads.  A kernel that reads its fields one by one through the pointer -- worse,
// inside a short-circuit "a && b == c && ..." -- makes a separate trip to L2 for each (0.3 - 0.4 us apiece: the tag test of the
// prepared Z proposal alone was five trips, 2 us of k_curve_chi's 16).  Same layout as Dyn up to alpha3 (checked below).
struct DynHead {
  uint32_t iter, slot, tt_step, status, iter_hyper, slot_hyper;
  int32_t pend_dir;
  uint32_t ll_pending, ll_slot, ll_use_part;
  uint32_t zprep_valid, zprep_iter, zprep_tt, zprep_chain;
  uint32_t znorm_valid, znorm_iter, znorm_tt;
  uint32_t piprep_valid, piprep_iter;
  uint32_t slot_base;
  unsigned long long zprep_seed;
  double beta, sigma2, alpha3;
};
static_assert(sizeof(DynHead) == 112, "DynHead is seven 16-byte loads");
static_assert(offsetof(Dyn, zprep_seed) == offsetof(DynHead, zprep_seed) && offsetof(Dyn, alpha3) == offsetof(DynHead, alpha3) &&
              offsetof(Dyn, slot_base) == offsetof(DynHead, slot_base), "DynHead mirrors the head of Dyn");

#if defined(__HIPCC__)
__device__ __forceinline__ DynHead dyn_head(const Dyn* dyn) {
  union U { DynHead h; int4 q[7]; __device__ U() {} } u;
  const int4* p = reinterpret_cast<const int4*>(dyn);
#pragma unroll
  for (int j = 0; j < 7; ++j) u.q[j] = p[j];
  return u.h;
}
#endif

// The same head (and pi) for the kernels whose workgroups all start together and all read it first (k_curve_z, k_curve_chi: 512
// workgroups x 4 waves x 7 + K uniform loads of ONE line -- 1800 requests per XCD for a line that seven of the eight L2s do not
// hold, and every other load of the wave queues behind them, in issue order): ONE load instruction per wave (lane j takes dword
// j of the first 192 bytes: head, rss, loglik, pi), staged in LDS before the workgroup's first barrier, read from there.
constexpr int DYN_LDS_DOUBLES = 24;
static_assert(offsetof(Dyn, pi) == 128 && sizeof(double) * KMAX == 64, "dwords 0 .. 47 of Dyn: head, rss, loglik, pi");
#if defined(__HIPCC__)
__device__ __forceinline__ uint32_t dyn_head_fetch(const Dyn* dyn) {
  return reinterpret_cast<const uint32_t*>(dyn)[min((int)(threadIdx.x & 63), 47)];
}
__device__ __forceinline__ void dyn_head_stage(double* sDyn, uint32_t w) {      // (every wave stores the same 48 words)
  reinterpret_cast<uint32_t*>(sDyn)[min((int)(threadIdx.x & 63), 47)] = w;
}
__device__ __forceinline__ DynHead dyn_head_lds(const double* sDyn) {           // sDyn: 16-byte aligned
  union U { DynHead h; int4 q[7]; __device__ U() {} } u;
  const int4* p = reinterpret_cast<const int4*>(sDyn);
#pragma unroll
  for (int j = 0; j < 7; ++j) u.q[j] = p[j];
  return u.h;
}
__device__ __forceinline__ const double* dyn_pi_lds(const double* sDyn) { return sDyn + 16; }
#endif

struct Dims {
  int n, K, P, M, D;
  int BW;               // band half-width of G_i (= spline degree; 0 for the multivariate model)
  int BWP;              // band half-width of the conditional precisions (max of BW and the penalty's)
  int LG;               // (BW+1)*P
  int LREC;             // record length in doubles (LG + P + 1, padded to even)
  int MD;               // active mt values: M+1, or 1 when Phi = chi = 0 (Nu_Z stage)
  int A;                // directions = K * MD
  int NZZ, NCC, R;      // pair rows: K(K+1)/2 * MD(MD+1)/2
  int RT, AT;           // 16-row tiles of pair rows / of single-weight rows
  int CTG, CTS;         // 16-col tiles over the G part (multivariate model: 1, the columns are identical) / the s part of a record
  int NT;               // total output tiles of the pair-Gram kernel = RT*CTG + AT*CTS
  int NWG;              // workgroups (= K-slices) of the pair-Gram kernel
  int mv;               // multivariate model flags (prior (1/tau) I, tau stored inverted, ...)
  int64_t n_obs_total;  // sum_i n_i
  int64_t half_sum;     // sum_i floor(n_i / 2)   (UpdateSigma.h:49 integer division)
};

// Everything a kernel needs, passed by value (kernarg).
struct Ctx {
  Dims d;
  Hyper h;
  uint64_t seed;
  uint32_t chain;
  uint32_t mask;
  int T;                        // chain slots allocated
  Dyn* dyn;
  // per-curve statistics
  const double* rec;            // n x LREC
  const int* ni;                // n
  // current state
  double* Z;                    // n x K col-major
  double* chi;                  // n x M col-major
  double* theta;                // A_full x P  (a = j*(M+1) + mt), direction-major
  double* delta;                // K x M
  double* Aa;                   // K x 2
  double* gamma;                // K x P x M
  // work buffers
  double* logz_part;            // nblk_curve x K   partial sums of log Z
  double* rss_part;             // nblk_curve       partial residual sums of squares
  double* pg_part;              // NWG x NT x 256   pair-Gram partial tiles
  double* H;                    // R x LG           pair-weighted Gram blocks (band-packed)
  double* H2;                   // R x P x (2BW+2)  the same blocks by row p: [G(p,p-BW) .. G(p,p+BW), 0], piece-major (h2_index, kernels_sweep.hip)
  double* tvec;                 // A x P            sum_i w_ai s_i
  double* rvec;                 // A x P            r_a = t_a - sum_b H_ab theta_b at the start of the sweep
  double* hq;                   // A x P            H_aa theta_a
  double* Lz;                   // A x P            chol_lower(C_a) z_a, z_a the direction's N(0,I) draw
  double* gstd;                 // state-independent variates of job_hyper (scalar_jobs.hpp)
  double* zprep;                // (3K + 5) x n  Z proposals prepared one iteration ahead (z_proposal.hpp)
  double* chi_norm;             // n x M  standard normals of this iteration's chi update (z_proposal.hpp)
  double* piprep;               // tables of the next iteration's pi / alpha_3 job (scalar_jobs.hpp)
  double* Cmat;                 // A x P x P        covariance of each direction's conditional
  const double* Pmat;           // P x P penalty
  const int* sweep_tab;         // step tables of k_sweep_chain for the run's (MD, mask): k_sweep_tables (shared by the chains of a batch)
  // ---- covariate adjustment (D > 0): eta_j[:,d] is direction (j, 0, d), xi_jm[:,d] is (j, m+1, d), weight
  //      w = Z_ij * chit_{i,mt} * X_id.  The Phi / nu block sees them as a per-curve offset o_i; they are
  //      sampled by the sequential "direct" steps of kernels_cov.hip.
  const double* X;              // n x D col-major covariates
  double* thetaX;               // [K*(M+1)*D][P], row ax = (j*(M+1) + mt)*D + d
  double* tau_eta;              // K x D
  double* gamma_xi;             // K cubes P x D x M (reference layout of gamma_xi(iter, k))
  double* delta_xi;             // K x M x D
  double* A_xi;                 // K x 2 x D
  double* stil;                 // n x P   s_i - G_i o_i  (o_i = covariate part of the fitted coefficient)
  double* yyp_part;             // nblk_curve  partial sums of yy_i - 2 o_i's_i + o_i'G_i o_i
  double* cfull;                // n x P   c_i (full fitted coefficient), maintained during the eta / Xi steps
  double* gfull;                // n x P   G_i c_i
  double* w2_part;              // NPAIR x NB2 x LG   partial sums of w_a w_b G_i over the in-group direction pairs
  double* H2aa;                 // NPAIR x LG         sum_i w_a w_b G_i (band-packed); pair (u <= v) of group g at g*NPG + v(v+1)/2 + u
  double* gstd2;                // K*D + K*M*D + K*D*P*M  standard gamma variates of tau_eta, delta_xi, gamma_xi (k_cov_prep)
  double* Wdir;                 // n x A2             w_{a,i} = Z_ij chit_{i,mt} X_id of this iteration's eta / Xi block
  double* C2;                   // A2 x P x P
  double* Lz2;                  // A2 x P
  double* step_part;            // 2 x NBS x D x P partial sums of w (s_i - g_i) of the current group of directions (two parities)
  double* thetaN;               // as thetaX: the values drawn in this iteration's eta / Xi steps (committed by k_cov_hyper)
  double* delta_cur;            // P + 1           theta_new - theta_old of the last step (pending on c_i, g_i)
  int pi_in_factor;             // the pi / alpha_3 job of the iteration is a workgroup of k_factor (batches on the packed pair-Gram path)
  int defer_loglik;             // the iteration has no k_loglik: bookkeeping in job_hyper, reduction in the next k_pair_gram
  int defer_hyper;              // k_curve_chi's scalar-job workgroup only advances the counters; the job itself runs as an extra
                                // workgroup of the NEXT k_pair_gram (or of the flush kernel at the end of a run)
  int ll_use_part;              // (deferred) log-likelihood from the per-curve residual partial sums
  int covariance_adj;           // Xi block on (BFMMM.h:4602 vs :4067)
  int A2;                       // eta / xi directions: K*D (+ K*M*D)
  int NB2, NBS;
  int NPG, NPAIR;               // in-group direction pairs: D(D+1)/2 per group of D consecutive directions, (A2/D)*NPG in all
  double *c_eta, *c_xi, *c_tau_eta, *c_gamma_xi, *c_delta_xi, *c_A_xi;
  double YY;                    // sum_i yy_i
  // ---- chain batch: a handle may hold several independent chains over ONE copy of the data (multi-try chains,
  //      UserFunctions.cpp:302-325).  Every per-chain buffer of chain q is the buffer of chain 0 moved by q * chain_bytes
  //      (covariate buffers: q * chain_bytes_cov); the kernels run chain blockIdx.z (chain_view below) with RNG chain id
  //      chain + q * chain_id_stride.  rec, ni, Pmat and X are shared.
  size_t chain_bytes, chain_bytes_cov;
  uint32_t chain_id_stride;
  int nch;                      // chains in the batch = gridDim.z of every launch
  // chain storage (slot-major, each slot laid out exactly as the reference returns it)
  double *c_nu, *c_chi, *c_Z, *c_pi, *c_alpha3, *c_delta, *c_A, *c_sigma, *c_tau, *c_gamma, *c_Phi, *c_loglik;
  int nblk_curve;
};

// geometry of k_pair_gram_pack (kernels_sweep.hip), the pair-Gram kernel of chain batches and long curve sets
struct PgPack {
  int KS, NKS;           // curves per k-slice (a multiple of 16), slices
  int TG, TS;            // packed row tiles: pair rows (nch R), single-weight rows (nch A)
  int NRG, NCG, NWG_S;   // G workgroups per slice: NRG row groups (WPG pair tiles each) x NCG column groups (2 column pairs
                         // = 64 record columns each); s workgroups per slice (WPG single tiles each)
  int SLG, SLS;          // doubles per (chain, curve) of the pair-weight table (NZZ + NCC + 1, even) / the raw table (K + MD + 1, even)
  int NQG, NQS;          // most chains a G / an s workgroup stages
  int WPG;               // waves (= packed row tiles) per workgroup
  int NTP;               // output tiles per slice: TG 2 NP2 + TS 2
  int NP2;               // 32-column pairs of the G part
};

// per-chain pointers of Ctx (everything but the shared data rec, ni, Pmat, X)
#define BFMMM_CHAIN_PTRS(X_)                                                                                         \
  X_(dyn) X_(Z) X_(chi) X_(theta) X_(delta) X_(Aa) X_(gamma) X_(logz_part) X_(rss_part) X_(pg_part) X_(H) X_(H2)       \
  X_(tvec) X_(rvec) X_(hq) X_(Lz) X_(gstd) X_(zprep) X_(chi_norm) X_(piprep) X_(Cmat)                        \
  X_(c_nu) X_(c_chi) X_(c_Z) X_(c_pi) X_(c_alpha3) X_(c_delta) X_(c_A) X_(c_sigma) X_(c_tau) X_(c_gamma) X_(c_Phi) X_(c_loglik)
#define BFMMM_CHAIN_PTRS_COV(X_)                                                                                     \
  X_(thetaX) X_(tau_eta) X_(gamma_xi) X_(delta_xi) X_(A_xi) X_(stil) X_(yyp_part) X_(cfull) X_(gfull) X_(w2_part)      \
  X_(H2aa) X_(gstd2) X_(Wdir) X_(C2) X_(Lz2) X_(step_part) X_(thetaN) X_(delta_cur)                                   \
  X_(c_eta) X_(c_xi) X_(c_tau_eta) X_(c_gamma_xi) X_(c_delta_xi) X_(c_A_xi)

template <typename T>
// (pointer arithmetic, not integer arithmetic: a pointer that has been through an integer loses its address space, and every
//  access through it becomes a FLAT instruction -- 64-bit address registers, and counted on the LDS counter as well, so that
//  each LDS wait of a kernel also waited for its global loads in flight)
__host__ __device__ inline T* ptr_shift(T* p, size_t bytes) { return (T*)((char*)p + bytes); }

// the context of chain q of the batch
__host__ __device__ inline Ctx chain_ctx(const Ctx& c0, unsigned q) {
  Ctx c = c0;
  const size_t o1 = (size_t)q * c0.chain_bytes, o2 = (size_t)q * c0.chain_bytes_cov;
#define X_(f) c.f = ptr_shift(c0.f, o1);
  BFMMM_CHAIN_PTRS(X_)
#undef X_
#define X_(f) c.f = ptr_shift(c0.f, o2);
  BFMMM_CHAIN_PTRS_COV(X_)
#undef X_
  c.chain = c0.chain + q * c0.chain_id_stride;
  return c;
}
#ifdef __HIPCC__
__device__ inline Ctx chain_view(const Ctx& c0) { return chain_ctx(c0, blockIdx.z); }
#endif

extern int g_exact_instances;      // bfmmm_set_exact_instances (bfmmm_capi.hip): 0 = the launchers use only the general instances

__host__ __device__ inline int tri_index(int n, int a, int b) {  // a <= b < n  -> index in packed upper triangle
  return a * n - (a * (a - 1)) / 2 + (b - a);
}

// Diagnostic timeline (compiled in only with -DBFMMM_TIMELINE): stamps[2k] = start of the first workgroup of
// kernel k, stamps[2k+1] = latest end over all its workgroups, on the 100 MHz wall clock.
#ifdef BFMMM_TIMELINE
struct Timeline {
  unsigned long long* s; int k;
  __device__ Timeline(const Ctx& c, int k_) : s(c.dyn->stamps), k(k_) {
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) s[2 * k] = wall_clock64();
    if (threadIdx.x == 0) atomicMax(&s[16 + k], (unsigned long long)wall_clock64());   // latest workgroup start
  }
  __device__ ~Timeline() { if (threadIdx.x == 0) atomicMax(&s[2 * k + 1], (unsigned long long)wall_clock64()); }
};
#define TIMELINE(c, k) Timeline tl_((c), (k))
#define TSTAMP(c, i) do { if (threadIdx.x == 0) atomicMax(&(c).dyn->stamps[i], (unsigned long long)wall_clock64()); } while (0)
#define TSTAMP0(c, i) do { if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) (c).dyn->stamps[i] = wall_clock64(); } while (0)
#else
#define TIMELINE(c, k) do { } while (0)
#define TSTAMP(c, i) do { } while (0)
#define TSTAMP0(c, i) do { } while (0)
#endif

template <int CTRL>
__device__ inline double dpp_add(double v) {   // v + dpp_permute<CTRL>(v) within a row of 16 lanes
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
  return v + __hiloint2double(hi, lo);
}
__device__ inline double row16_sum(double v) {   // every lane of a 16-lane row gets the row's sum
  v = dpp_add<0xB1>(v);     // quad_perm [1,0,3,2]
  v = dpp_add<0x4E>(v);     // quad_perm [2,3,0,1]
  v = dpp_add<0x141>(v);    // row_half_mirror
  v = dpp_add<0x140>(v);    // row_mirror
  return v;
}

// sum over the LPC (32 or 64) lanes of a curve group, every lane gets it; fixed order
template <int LPC>
__device__ inline double group_sum(double v) {
  v = row16_sum(v);
  v += __shfl_xor(v, 16, 64);
  if (LPC == 64) v += __shfl_xor(v, 32, 64);
  return v;
}

// Batched global -> LDS copy: every thread issues UN independent loads before the first store, so
// the copy costs one memory latency per UN*nthreads elements instead of one per nthreads.
template <int UN>
__device__ inline void copy_to_lds(double* dst, const double* __restrict__ src, int count, int tid, int nthreads) {
  for (int base = 0; base < count; base += nthreads * UN) {
    double v[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) v[u] = src[min(base + tid + nthreads * u, count - 1)];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int idx = base + tid + nthreads * u;
      if (idx < count) dst[idx] = v[u];
    }
  }
}

// allow a kernel to use all of the CU's 160 KiB of LDS for its dynamic region
inline void set_max_lds(const void* fn) {
  hipFuncAttributes at;
  if (hipFuncGetAttributes(&at, fn) != hipSuccess) { (void)hipGetLastError(); return; }
  const int room = 160 * 1024 - (int)at.sharedSizeBytes;
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, room) != hipSuccess) (void)hipGetLastError();
}

}  // namespace bfmmm
