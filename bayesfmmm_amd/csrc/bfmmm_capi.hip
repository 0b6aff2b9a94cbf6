// C ABI of the sampler (include/bfmmm.h): handle management, state marshalling between the
// reference's column-major layouts and the device layout, iteration driver with HIP-graph replay.
#include "../../include/bfmmm.h"
#include "model.hpp"
#include "rng.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <chrono>
#include <mutex>
#include <string>
#include <vector>

namespace bfmmm {
int launch_stats_functional(int degree, int n, int P, int LREC, const int64_t* off, const double* t, const double* y,
                            const double* knots, int n_knots, double* rec, int* ni, double* B_dense, int* err,
                            hipStream_t st);
void launch_stats_multivariate(int n, int P, int LREC, const double* Y, double* rec, int* ni, hipStream_t st);
void launch_stats_totals(int n, int LREC, int yy_off, const double* rec, const int* ni, double* out_yy,
                         long long* out_counts, hipStream_t st);
int launch_curve(const Ctx& c, int which, int do_update, hipStream_t st);
int curve_blocks(int n, int P);
void prepare_curve_kernels();
void prepare_sweep_kernels();
void launch_pair_gram(const Ctx& c, int do_pg, int NKS, int KS, hipStream_t st);
void launch_pg_reduce(const Ctx& c, int NKS, hipStream_t st);
bool pgp_geometry(const Dims& d, int nch, int KS, int NKS, PgPack& g);
size_t pgp_pack_doubles(const PgPack& g);
void launch_pair_gram_pack(const Ctx& c, const PgPack& g, double* pack, hipStream_t st);
void launch_factor(const Ctx& c, hipStream_t st);
int launch_sweep(const Ctx& c, hipStream_t st);
void launch_sweep_tables(const Ctx& c, hipStream_t st);
size_t sweep_tab_ints(int A);
void launch_loglik(const Ctx& c, int use_rss_part, int r_stored, hipStream_t st);
void launch_loglik_flush(const Ctx& c, hipStream_t st, uint32_t* status_out);
void launch_fill_slots(const Ctx& c, double* chain, const double* cur, size_t len, int s0, int s1, hipStream_t st);
void launch_cov_block(const Ctx& c, hipStream_t st);
int cov_step_blocks(int nblk_curve);
int cov_w2_chunks(int n);
void prepare_cov_kernels();
bool cov_block_fits(const Ctx& c);
#ifdef BFMMM_TIMELINE
void fetch_wgtrace(unsigned long long* out);
void fetch_ztrace(unsigned long long* out);
void fetch_zphase(unsigned long long* out);
void fetch_fct(unsigned long long* out);
#endif
}  // namespace bfmmm

using namespace bfmmm;

static thread_local std::string g_err;
static constexpr int GRAPH_UNROLL = 10;
static std::mutex g_capture_mutex;   // one stream capture at a time (samplers may run on several host threads)
static int fail(const std::string& msg) { g_err = msg; return 1; }

#define HIPCHK(x)                                                                                   \
  do {                                                                                              \
    hipError_t e_ = (x);                                                                            \
    if (e_ != hipSuccess) {                                                                         \
      (void)hipGetLastError(); /* (not sticky: a caller that retries with a smaller batch starts clean) */ \
      char buf_[512];                                                                               \
      snprintf(buf_, sizeof buf_, "HIP error %s at %s:%d (%s)", hipGetErrorString(e_), __FILE__, __LINE__, #x); \
      return fail(buf_);                                                                            \
    }                                                                                               \
  } while (0)

enum { FAM_TOTAL = 0, FAM_Z, FAM_PG, FAM_FACTOR, FAM_SWEEP, FAM_CHI, FAM_LOGLIK, FAM_REDUCE, FAM_COUNT };
static const char* kFamNames[FAM_COUNT] = {"total", "curve_z", "pair_gram", "factor", "sweep", "curve_chi", "loglik", "pg_reduce"};

struct bfmmm_handle {
  bfmmm_config cfg;
  int device = 0;
  hipStream_t st = nullptr, st2 = nullptr;
  hipEvent_t evA = nullptr, evB = nullptr, evC = nullptr, evD = nullptr;
  Ctx c;                       // template context (full MD); its per-chain pointers are those of chain 0 of the batch
  int nch = 1;                 // chains in the batch (bfmmm_create_batch), all advanced in lockstep by bfmmm_run
  int sel = 0;                 // the chain the state / chain accessors address (bfmmm_select_chain)
  int T = 0;
  int64_t n_obs = 0;
  // raw inputs kept on the device for bfmmm_get_basis
  double* d_t = nullptr; double* d_y = nullptr; int64_t* d_off = nullptr; double* d_knots = nullptr; int n_knots = 0;
  std::vector<void*> allocs;
  char* arena = nullptr;               // base of the per-chain arena (chain q at arena + q * c.chain_bytes)
  char* arena_cov = nullptr;           // the same for the covariate buffers (c.chain_bytes_cov)
  uint32_t* status_host = nullptr;     // pinned, host-mapped: the chains' status words after a run
  uint32_t* status_dev = nullptr;      // its device address (written by the run's last kernel)
  size_t pg_part_doubles = 0;
  // graph cache for the last (mask, md, seed, chain)
  // Captured graphs of a run, one set per SUB-BATCH (run_impl splits a chain batch over up to MAX_SUB streams):
  //   gN  GRAPH_UNROLL full iterations (amortises the fixed cost of a graph launch);
  //   fused runs (chi kernel of iteration i also runs the Z update of i + 1): gFN = GRAPH_UNROLL bodies [pair_gram .. chi + Z],
  //   gL = the closing iteration without the Z part;
  //   gR / gFR = the remainder of a run after the unrolled graphs, as ONE graph of rem / remF iterations
  //   (one per remainder length, kept: a warm-up run of another length between prepare_run and the run does not evict the run's graph)
  static constexpr int NREM = 10;        // = GRAPH_UNROLL
  struct GraphSet {
    hipGraphExec_t gN = nullptr, gFN = nullptr, gL = nullptr;
    hipGraphExec_t gR[NREM] = {}, gFR[NREM] = {};
    hipGraphExec_t gW = nullptr;      // a WHOLE short run (first Z, bodies, closing iteration, flush) of gW_n iterations
    int gW_n = 0;
    template <typename F> void each(F f) { f(&gN); f(&gFN); f(&gL); f(&gW); for (int r = 0; r < NREM; ++r) { f(&gR[r]); f(&gFR[r]); } }
  };
  static constexpr int MAX_SUB = 4;
  // packed partial tiles of k_pair_gram_pack, one buffer per sub-batch stream (+ one for the whole batch on one stream)
  // snapshot of the chains' work state for the dry launch of freshly captured graphs (bfmmm_prepare_run)
  char* dry_snap = nullptr;
  size_t dry_snap_bytes = 0;
  double* pg_pack[MAX_SUB + 1] = {};
  size_t pg_pack_doubles[MAX_SUB + 1] = {};
  GraphSet gs[MAX_SUB];
  hipStream_t sub_st[MAX_SUB] = {nullptr, nullptr, nullptr, nullptr};     // [0] = st
  hipEvent_t sub_ev[MAX_SUB] = {nullptr, nullptr, nullptr, nullptr};
  int g_nsub = 1;
  int g_pack_mode = 0;
  uint32_t g_mask = 0; int g_md = -1; uint64_t g_seed = 0; uint32_t g_chain = 0;
  int last_md = -1;
  int64_t tab_key = -1;                // (MD, mask) the step tables of k_sweep_chain were built for
  int launch_error = 0;
  bool g_valid = false;                // the captured graphs match (g_mask, g_md, g_seed, g_chain)
  int slot_base = 0;                   // chain slot of iteration i is i - slot_base (bfmmm_set_slot_base)
  double* tt_save = nullptr;            // state saved across a tempered-transition block
  std::vector<double> B_host;           // bfmmm_create_from_basis: the caller's basis rows (bfmmm_get_basis)
  bool state_dirty = true;             // the state was changed from the host: proposals prepared on the device are stale
  int profile = 0;
  double fam_ms[FAM_COUNT] = {0};
  int64_t fam_launches[FAM_COUNT] = {0};
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

// Synchronous copy on the sampler's own stream: the legacy (NULL) stream must not be touched while
// another host thread is capturing a graph.
static hipError_t copy_sync(bfmmm_handle* h, void* dst, const void* src, size_t bytes, hipMemcpyKind kind) {
  hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, h->st);
  if (e != hipSuccess) return e;
  return hipStreamSynchronize(h->st);
}

template <typename T>
static int dalloc(bfmmm_handle* h, T** p, size_t count) {
  void* q = nullptr;
  const size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
  HIPCHK(hipMalloc(&q, bytes));
  HIPCHK(hipMemsetAsync(q, 0, bytes, h->st));
  h->allocs.push_back(q);
  *p = (T*)q;
  return 0;
}

// Per-chain buffers come out of one arena per chain: the requests are collected first, then ONE allocation of
// nch * stride bytes is made and the pointers of chain 0 are handed out; chain q's copy of every buffer sits q * stride
// bytes further (Ctx::chain_bytes, chain_ctx in model.hpp).
static std::vector<hipGraphExec_t*> graph_slots(bfmmm_handle::GraphSet& g) {
  std::vector<hipGraphExec_t*> v;
  g.each([&](hipGraphExec_t* p) { v.push_back(p); });
  return v;
}

struct ArenaReq { void* slot; size_t bytes; };
template <typename T>
static void areq(std::vector<ArenaReq>& v, T** p, size_t count) { v.push_back({(void*)p, std::max<size_t>(count, 1) * sizeof(T)}); }
static int arena_commit(bfmmm_handle* h, const std::vector<ArenaReq>& v, int nch, size_t* stride_out, char** base_out) {
  size_t off = 0;
  std::vector<size_t> offs;
  for (const ArenaReq& r : v) { offs.push_back(off); off += (r.bytes + 255) & ~(size_t)255; }
  char* base = nullptr;
  HIPCHK(hipMalloc((void**)&base, off * (size_t)nch));
  h->allocs.push_back(base);
  HIPCHK(hipMemsetAsync(base, 0, off * (size_t)nch, h->st));
  for (size_t i = 0; i < v.size(); ++i) { void* q = base + offs[i]; memcpy(v[i].slot, &q, sizeof q); }
  *stride_out = off;
  *base_out = base;
  return 0;
}
// the context of the selected chain (host view)
static Ctx selc(const bfmmm_handle* h) { return chain_ctx(h->c, (unsigned)h->sel); }

extern "C" void bfmmm_config_defaults(bfmmm_config* cfg) {
  // defaults of BFMMM_Nu_Z_multiple_try / BFMMM_Theta_est / BFMMM_warm_start
  // (UserFunctions.cpp:178-193, 697-715, 1353-1378)
  memset(cfg, 0, sizeof *cfg);
  for (int k = 0; k < 8; ++k) cfg->c[k] = 10.0;
  cfg->b = 10; cfg->nu_1 = 3;
  cfg->alpha1l = 1; cfg->alpha2l = 2; cfg->beta1l = 1; cfg->beta2l = 1;
  cfg->a_Z_PM = 10000; cfg->a_pi_PM = 1000; cfg->var_alpha3 = 0.05; cfg->var_epsilon1 = 1; cfg->var_epsilon2 = 1;
  cfg->alpha_nu = 10; cfg->beta_nu = 1; cfg->alpha_eta = 10; cfg->beta_eta = 1; cfg->alpha_0 = 1; cfg->beta_0 = 1;
}

extern "C" const char* bfmmm_last_error(void) { return g_err.c_str(); }

// dims that depend on how many mt-directions are active
static void set_md(Dims& d, int MD) {
  d.MD = MD;
  d.A = d.K * MD;
  d.NZZ = d.K * (d.K + 1) / 2;
  d.NCC = MD * (MD + 1) / 2;
  d.R = d.NZZ * d.NCC;
  d.RT = (d.R + 15) / 16;
  d.AT = (d.A + 15) / 16;
  // multivariate model: G_i = I, every column of the G part of a record is the same column of ones, so the pair-Gram
  // contraction computes ONE column tile and the reduction writes the (scalar) block value to all P columns of H
  d.CTG = d.mv ? 1 : (d.LG + 15) / 16;
  d.CTS = (d.P + 15) / 16;
  d.NT = d.RT * d.CTG + d.AT * d.CTS;
}

static void pg_geometry(const Dims& d, int& NTG, int& NKS, int& KS) {
  // k-slices of the pair-Gram contraction.  LDS doubles per curve: the raw weight row, the record columns and,
  // for the G workgroups, the pair-weight row (k_pair_gram); a workgroup never stages more than 96 KB.
  const int row_g = (d.K + d.MD + 1) + 16 + (d.NZZ + d.NCC + 1);
  const int row_s = (d.K + d.MD + 1) + d.CTS * 16;
  const int ks_cap = std::max(16, (int)((96 * 1024) / (sizeof(double) * (size_t)std::max(row_g, row_s)) - 2) / 16 * 16);
  // one workgroup per CU: (CTG + 2) column groups x NKS k-slices <= 256 whenever the LDS cap allows, so that
  // every workgroup is resident at once (a 257th would wait a whole workgroup lifetime for a free CU)
  NKS = std::max(1, std::min(256 / (d.CTG + 2), d.n / 16));
  KS = (d.n + NKS - 1) / NKS;
  KS = (KS + 15) / 16 * 16;                  // MFMA k-slots are taken in trips of 4 steps per slot (two 16-byte LDS reads)
  KS = std::min(KS, ks_cap);
  NKS = (d.n + KS - 1) / KS;
  NTG = 1;
}

// a basis supplied by the caller (bfmmm_create_from_basis): rows of B, its band, the penalty of the nu prior
struct BasisSpec {
  int P, band, pen_band;
  const double* B;        // n_obs x P row-major
  const double* Pmat;     // P x P column-major
};

static int create_impl(const bfmmm_config* cfg, int device, const double* y, const double* t, const int64_t* offsets,
                       const double* internal_knots, const double* boundary_knots, const BasisSpec* bs, int n_chains,
                       bfmmm_handle** out);

extern "C" int bfmmm_create(const bfmmm_config* cfg, int device, const double* y, const double* t, const int64_t* offsets,
                            const double* internal_knots, const double* boundary_knots, bfmmm_handle** out) {
  return create_impl(cfg, device, y, t, offsets, internal_knots, boundary_knots, nullptr, 1, out);
}

extern "C" int bfmmm_create_batch(const bfmmm_config* cfg, int device, const double* y, const double* t, const int64_t* offsets,
                                  const double* internal_knots, const double* boundary_knots, int n_chains, bfmmm_handle** out) {
  return create_impl(cfg, device, y, t, offsets, internal_knots, boundary_knots, nullptr, n_chains, out);
}

extern "C" int bfmmm_create_from_basis(const bfmmm_config* cfg, int device, const double* y, const double* B, const int64_t* offsets,
                                       int P, int band, const double* Pmat, int pen_band, bfmmm_handle** out) {
  return bfmmm_create_from_basis_batch(cfg, device, y, B, offsets, P, band, Pmat, pen_band, 1, out);
}

extern "C" int bfmmm_create_from_basis_batch(const bfmmm_config* cfg, int device, const double* y, const double* B, const int64_t* offsets,
                                             int P, int band, const double* Pmat, int pen_band, int n_chains, bfmmm_handle** out) {
  if (!cfg || !y || !B || !offsets || !Pmat || !out) return fail("bfmmm_create_from_basis: null argument");
  if (cfg->model != BFMMM_MODEL_FUNCTIONAL) return fail("bfmmm_create_from_basis: functional model only");
  if (P < 1 || band < 0 || pen_band < 0) return fail("bfmmm_create_from_basis: bad dimensions");
  if (band > BWWIDE) return fail("bfmmm_create_from_basis: the band half-width of B'B must not exceed 31 in this build");
  BasisSpec bs = {P, band, pen_band, B, Pmat};
  return create_impl(cfg, device, y, nullptr, offsets, nullptr, nullptr, &bs, n_chains, out);
}

// frees the handle on every early return of create_impl / bfmmm_set_covariates' callers (released on success)
struct HandleGuard {
  bfmmm_handle* h;
  ~HandleGuard() { if (h) bfmmm_destroy(h); }
};

static int create_impl(const bfmmm_config* cfg, int device, const double* y, const double* t, const int64_t* offsets,
                       const double* internal_knots, const double* boundary_knots, const BasisSpec* bs, int n_chains,
                       bfmmm_handle** out) {
  if (!cfg || !out || !y) return fail("bfmmm_create: null argument");
  if (n_chains < 1 || n_chains > 4096) return fail("bfmmm_create_batch: n_chains must be between 1 and 4096");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail("bfmmm_create: no HIP device available (the sampler has no CPU fallback)");
  if (device < 0 || device >= ndev) return fail("bfmmm_create: invalid device index");
  const bool mv = cfg->model == BFMMM_MODEL_MULTIVARIATE;
  const int n = cfg->n_funct, K = cfg->K, M = cfg->n_eigen;
  if (n < 1) return fail("'n_funct' must be an integer greater than or equal to 1");
  if (K < 2) return fail("'K' must be an integer greater than or equal to 2");
  if (K > KMAX) return fail("K larger than 8 is not supported by this build");
  if (M < 1) return fail("'n_eigen' must be an integer greater than or equal to 1");
  if (cfg->tot_mcmc_iters < 1) return fail("'tot_mcmc_iters' must be positive");
  int P, BW;
  if (mv) {
    P = cfg->P; BW = 0;
    if (P < 1) return fail("multivariate model: P must be positive");
  } else if (bs) {
    P = bs->P;
    BW = (bs->band <= BWMAX) ? std::min(bs->band, std::max(P - 1, 0)) : (bs->band <= BWMID) ? BWMID : BWWIDE;     // wide bands share two instantiations (zero padded)
  } else {
    if (!t || !offsets || !boundary_knots || (cfg->n_internal_knots > 0 && !internal_knots))
      return fail("bfmmm_create: null argument");
    if (cfg->basis_degree < 1) return fail("'basis_degree' must be an integer greater than or equal to 1");
    if (cfg->basis_degree > BWMAX) return fail("basis_degree larger than 5 is not supported by this build");
    for (int i = 0; i < cfg->n_internal_knots; ++i) {
      if (boundary_knots[0] >= internal_knots[i])
        return fail("at least one element in 'internal_knots' is less than or equal to first boundary knot");
      if (boundary_knots[1] <= internal_knots[i])
        return fail("at least one element in 'internal_knots' is more than or equal to second boundary knot");
    }
    P = cfg->n_internal_knots + cfg->basis_degree + 1;
    BW = std::min(cfg->basis_degree, P - 1);
  }
  if (P > PMAX) return fail("P larger than 64 is not supported by this build");
  if (M > 16) return fail("n_eigen larger than 16 is not supported by this build");

  HIPCHK(hipSetDevice(device));
  prepare_curve_kernels();
  prepare_sweep_kernels();
  prepare_cov_kernels();
  { hipError_t e0 = hipGetLastError(); if (e0 != hipSuccess) fprintf(stderr, "[bfmmm] note: kernel attribute setup reported %s\n", hipGetErrorString(e0)); }
  bfmmm_handle* h = new bfmmm_handle();
  HandleGuard guard{h};        // every early return below frees the handle, its streams, events and device memory
  h->cfg = *cfg;
  h->device = device;
  h->T = cfg->tot_mcmc_iters;
  h->nch = n_chains;
  HIPCHK(hipStreamCreateWithFlags(&h->st, hipStreamNonBlocking));
  HIPCHK(hipEventCreate(&h->evA));
  HIPCHK(hipEventCreate(&h->evB));
  HIPCHK(hipEventCreate(&h->evC));
  HIPCHK(hipEventCreate(&h->evD));
  HIPCHK(hipEventCreate(&h->ev0));
  HIPCHK(hipEventCreate(&h->ev1));
  Ctx& c = h->c;
  memset(&c, 0, sizeof c);
  Dims& d = c.d;
  d.n = n; d.K = K; d.P = P; d.M = M; d.D = 0; d.BW = BW; d.LG = (BW + 1) * P;
  d.LREC = (d.LG + P + 1 + 1) / 2 * 2;
  d.mv = mv ? 1 : 0;
  d.BWP = mv ? 0 : std::max(BW, 1);   // RW1 penalty is tridiagonal
  if (bs) d.BWP = std::min(std::max(BW, bs->pen_band), (BW > BWMAX) ? BWWIDE : BWMAX);
  if (bs && BW == BWMID && bs->pen_band > BWMID) { BW = BWWIDE; d.BW = BW; d.LG = (BW + 1) * P; d.LREC = (d.LG + P + 1 + 1) / 2 * 2; d.BWP = std::min(std::max(BW, bs->pen_band), BWWIDE); }
  if (bs && BW <= BWMAX && bs->pen_band > BWMAX) return fail("bfmmm_create_from_basis: a penalty band wider than 5 needs a basis band wider than 5 in this build");
  set_md(d, M + 1);
  for (int k = 0; k < KMAX; ++k) c.h.c[k] = (k < 8) ? cfg->c[k] : 10.0;
  c.h.b = cfg->b; c.h.nu_1 = cfg->nu_1;
  c.h.alpha1l = cfg->alpha1l; c.h.alpha2l = cfg->alpha2l; c.h.beta1l = cfg->beta1l; c.h.beta2l = cfg->beta2l;
  c.h.a_Z_PM = cfg->a_Z_PM; c.h.a_pi_PM = cfg->a_pi_PM; c.h.var_alpha3 = cfg->var_alpha3;
  c.h.var_epsilon1 = cfg->var_epsilon1; c.h.var_epsilon2 = cfg->var_epsilon2;
  c.h.alpha_nu = cfg->alpha_nu; c.h.beta_nu = cfg->beta_nu; c.h.alpha_eta = cfg->alpha_eta; c.h.beta_eta = cfg->beta_eta;
  c.h.alpha_0 = cfg->alpha_0; c.h.beta_0 = cfg->beta_0;
  c.T = h->T;
  c.nblk_curve = curve_blocks(n, P);

  const int64_t n_obs = mv ? (int64_t)n * P : offsets[n];
  h->n_obs = n_obs;
  // ---- device buffers ----
  double* rec; int* ni;
  if (dalloc(h, &rec, (size_t)n * d.LREC) || dalloc(h, &ni, n)) return 1;
  c.rec = rec; c.ni = ni;
  int* stab;
  if (dalloc(h, &stab, sweep_tab_ints(K * (M + 1)))) return 1;
  c.sweep_tab = stab;
  // per-chain buffers (one arena per chain of the batch): state, work space, chain storage
  std::vector<ArenaReq> ar;
  areq(ar, &c.dyn, 1);
  areq(ar, &c.Z, (size_t)n * K); areq(ar, &c.chi, (size_t)n * M); areq(ar, &c.theta, (size_t)K * (M + 1) * P);
  areq(ar, &c.delta, (size_t)K * M); areq(ar, &c.Aa, (size_t)K * 2); areq(ar, &c.gamma, (size_t)K * P * M);
  int NTG, NKS, KS;
  pg_geometry(d, NTG, NKS, KS);
  h->pg_part_doubles = (size_t)NKS * d.NT * 256;
  areq(ar, &c.logz_part, (size_t)c.nblk_curve * K); areq(ar, &c.rss_part, c.nblk_curve);
  // (+ 2 doubles on H and Cmat: for odd P the last row thread of k_sweep_chain reads a 16-byte pair that starts at the last
  //  element -- the second half is masked, but the read must stay inside the request, whatever the arena's rounding)
  areq(ar, &c.pg_part, h->pg_part_doubles); areq(ar, &c.H, (size_t)d.R * d.LG + 2); areq(ar, &c.H2, (size_t)d.R * P * (2 * d.BW + 2));
  areq(ar, &c.tvec, (size_t)d.A * P); areq(ar, &c.rvec, (size_t)d.A * P); areq(ar, &c.hq, (size_t)d.A * P);
  areq(ar, &c.gstd, (size_t)K * P * M + (size_t)K * M + 13 * K + 8); areq(ar, &c.zprep, (size_t)(3 * K + 5) * n);
  areq(ar, &c.chi_norm, (size_t)n * M); areq(ar, &c.piprep, 9 * KMAX + 16); areq(ar, &c.Lz, (size_t)d.A * P);
  areq(ar, &c.Cmat, (size_t)d.A * P * P + 2);
  {
    const size_t T = (size_t)h->T;
    areq(ar, &c.c_nu, T * K * P); areq(ar, &c.c_chi, T * n * M); areq(ar, &c.c_Z, T * n * K); areq(ar, &c.c_pi, T * K);
    areq(ar, &c.c_alpha3, T); areq(ar, &c.c_delta, T * K * M); areq(ar, &c.c_A, T * K * 2); areq(ar, &c.c_sigma, T);
    areq(ar, &c.c_tau, T * K); areq(ar, &c.c_gamma, T * K * P * M); areq(ar, &c.c_Phi, T * K * P * M); areq(ar, &c.c_loglik, T);
  }
  if (arena_commit(h, ar, n_chains, &c.chain_bytes, &h->arena)) return 1;
  c.chain_bytes_cov = 0; c.chain_id_stride = 1; c.nch = n_chains;
  double* pm;
  if (dalloc(h, &pm, (size_t)P * P)) return 1;
  c.Pmat = pm;
  if (bs) {
    HIPCHK(copy_sync(h, pm, bs->Pmat, sizeof(double) * (size_t)P * P, hipMemcpyHostToDevice));
  } else {
    // RW1 penalty, BFMMM.h:1027-1037
    std::vector<double> Pm((size_t)P * P, 0.0);
    for (int j = 0; j < P; ++j) {
      Pm[0] = 1;
      if (j > 0) { Pm[j + (size_t)P * j] = 2; Pm[(j - 1) + (size_t)P * j] = -1; Pm[j + (size_t)P * (j - 1)] = -1; }
      Pm[(P - 1) + (size_t)P * (P - 1)] = 1;
    }
    HIPCHK(copy_sync(h, pm, Pm.data(), sizeof(double) * Pm.size(), hipMemcpyHostToDevice));
  }

  // ---- upload data, compute statistics on the device ----
  if (dalloc(h, &h->d_y, (size_t)n_obs)) return 1;
  HIPCHK(copy_sync(h, h->d_y, y, sizeof(double) * (size_t)n_obs, hipMemcpyHostToDevice));
  int* d_err;
  if (dalloc(h, &d_err, 1)) return 1;
  if (mv) {
    launch_stats_multivariate(n, P, d.LREC, h->d_y, rec, ni, h->st);
  } else if (bs) {
    // per-curve statistics from the supplied basis rows, on the host (a set-up step): band-packed G_i = B_i'B_i
    // ([dd * P + lo] = G(lo, lo + dd), dd <= BW), s_i = B_i'y_i, yy_i
    std::vector<double> hrec((size_t)n * d.LREC, 0.0);
    std::vector<int> hni((size_t)n);
    for (int i = 0; i < n; ++i) {
      double* r = hrec.data() + (size_t)i * d.LREC;
      hni[i] = (int)(offsets[i + 1] - offsets[i]);
      for (int64_t l = offsets[i]; l < offsets[i + 1]; ++l) {
        const double* b = bs->B + (size_t)l * P;
        for (int lo = 0; lo < P; ++lo) {
          if (b[lo] == 0.0) continue;
          for (int dd = 0; dd <= BW && lo + dd < P; ++dd) r[dd * P + lo] += b[lo] * b[lo + dd];
          r[d.LG + lo] += b[lo] * y[l];
        }
        r[d.LG + P] += y[l] * y[l];
      }
    }
    HIPCHK(copy_sync(h, rec, hrec.data(), sizeof(double) * hrec.size(), hipMemcpyHostToDevice));
    HIPCHK(copy_sync(h, ni, hni.data(), sizeof(int) * hni.size(), hipMemcpyHostToDevice));
    h->B_host.assign(bs->B, bs->B + (size_t)n_obs * P);
  } else {
    const int deg = cfg->basis_degree, nint = cfg->n_internal_knots;
    h->n_knots = nint + 2 * (deg + 1);
    std::vector<double> knots(h->n_knots);
    for (int i = 0; i <= deg; ++i) knots[i] = boundary_knots[0];
    for (int i = 0; i < nint; ++i) knots[deg + 1 + i] = internal_knots[i];
    for (int i = 0; i <= deg; ++i) knots[deg + 1 + nint + i] = boundary_knots[1];
    if (dalloc(h, &h->d_t, (size_t)n_obs) || dalloc(h, &h->d_off, (size_t)n + 1) || dalloc(h, &h->d_knots, h->n_knots))
      return 1;
    HIPCHK(copy_sync(h, h->d_t, t, sizeof(double) * (size_t)n_obs, hipMemcpyHostToDevice));
    HIPCHK(copy_sync(h, h->d_off, offsets, sizeof(int64_t) * ((size_t)n + 1), hipMemcpyHostToDevice));
    HIPCHK(copy_sync(h, h->d_knots, knots.data(), sizeof(double) * knots.size(), hipMemcpyHostToDevice));
    if (launch_stats_functional(deg, n, P, d.LREC, h->d_off, h->d_t, h->d_y, h->d_knots, h->n_knots, rec, ni, nullptr,
                                d_err, h->st))
      return fail("unsupported basis_degree");
  }
  double* d_yy; long long* d_cnt;
  if (dalloc(h, &d_yy, 1) || dalloc(h, &d_cnt, 2)) return 1;
  launch_stats_totals(n, d.LREC, d.LG + P, rec, ni, d_yy, d_cnt, h->st);
  HIPCHK(hipStreamSynchronize(h->st));
  HIPCHK(hipGetLastError());
  int herr = 0; long long cnt[2];
  HIPCHK(copy_sync(h, &herr, d_err, sizeof(int), hipMemcpyDeviceToHost));
  HIPCHK(copy_sync(h, &c.YY, d_yy, sizeof(double), hipMemcpyDeviceToHost));
  HIPCHK(copy_sync(h, cnt, d_cnt, sizeof cnt, hipMemcpyDeviceToHost));
  if (herr) return fail("at least one time point lies outside 'boundary_knots'");
  d.n_obs_total = cnt[0];
  d.half_sum = cnt[1];
  // neutral starting state (everything 1 / 0) so that a run before set_state is well defined
  Dyn dyn0;
  memset(&dyn0, 0, sizeof dyn0);
  dyn0.beta = 1; dyn0.sigma2 = 1; dyn0.alpha3 = 1;
  for (int k = 0; k < KMAX; ++k) { dyn0.pi[k] = 1.0 / K; dyn0.tau[k] = 1; }
  for (int q = 0; q < n_chains; ++q)
    HIPCHK(copy_sync(h, chain_ctx(c, (unsigned)q).dyn, &dyn0, sizeof dyn0, hipMemcpyHostToDevice));
  guard.h = nullptr;
  *out = h;
  return 0;
}

extern "C" int bfmmm_set_covariates(bfmmm_handle* h, const double* X, int D, int covariance_adj) {
  if (!h || !X) return fail("bfmmm_set_covariates: null argument");
  if (D < 1 || D > 8) return fail("bfmmm_set_covariates: the number of covariates must be between 1 and 8 in this build");
  if (h->c.d.D != 0) return fail("bfmmm_set_covariates: covariates are already set");
  HIPCHK(hipSetDevice(h->device));
  Ctx& c = h->c;
  Dims& d = c.d;
  const size_t n = d.n, K = d.K, P = d.P, M = d.M, T = (size_t)h->T;
  d.D = D;
  c.covariance_adj = covariance_adj ? 1 : 0;
  c.A2 = (int)(K * D + (covariance_adj ? K * M * D : 0));
  c.NB2 = cov_w2_chunks((int)n);
  c.NBS = cov_step_blocks(c.nblk_curve);
  c.NPG = D * (D + 1) / 2;
  c.NPAIR = (c.A2 / D) * c.NPG;
  if (!cov_block_fits(c)) {
    d.D = 0;
    return fail("bfmmm_set_covariates: this many covariates with this basis exceed the covariate kernels' on-chip staging (fewer covariates or a narrower band)");
  }
  double* Xd;
  if (dalloc(h, &Xd, n * D)) return 1;
  HIPCHK(copy_sync(h, Xd, X, sizeof(double) * n * D, hipMemcpyHostToDevice));
  c.X = Xd;
  // per-chain buffers of the covariate blocks: a second arena per chain (Ctx::chain_bytes_cov)
  std::vector<ArenaReq> ar;
  areq(ar, &c.thetaX, K * (M + 1) * D * P); areq(ar, &c.tau_eta, K * D); areq(ar, &c.gamma_xi, K * P * D * M);
  areq(ar, &c.delta_xi, K * M * D); areq(ar, &c.A_xi, K * 2 * D); areq(ar, &c.stil, n * P);
  areq(ar, &c.yyp_part, (size_t)c.nblk_curve); areq(ar, &c.cfull, n * P); areq(ar, &c.gfull, n * P);
  areq(ar, &c.w2_part, (size_t)c.NPAIR * c.NB2 * d.LG); areq(ar, &c.H2aa, (size_t)c.NPAIR * d.LG);
  areq(ar, &c.Wdir, n * (size_t)c.A2); areq(ar, &c.gstd2, K * D + K * M * D + K * D * P * M);
  areq(ar, &c.C2, (size_t)c.A2 * P * P + 2); areq(ar, &c.Lz2, (size_t)c.A2 * P);
  areq(ar, &c.step_part, 2 * (size_t)c.NBS * (D * P + 1)); areq(ar, &c.thetaN, K * (M + 1) * D * P); areq(ar, &c.delta_cur, P + 2);
  areq(ar, &c.c_eta, T * P * D * K); areq(ar, &c.c_xi, T * K * P * D * M); areq(ar, &c.c_tau_eta, T * K * D);
  areq(ar, &c.c_gamma_xi, T * K * P * D * M); areq(ar, &c.c_delta_xi, T * K * M * D); areq(ar, &c.c_A_xi, T * K * 2 * D);
  if (arena_commit(h, ar, h->nch, &c.chain_bytes_cov, &h->arena_cov)) { d.D = 0; return 1; }
  // neutral state: eta = xi = 0, tau_eta = gamma_xi = delta_xi = A_xi = 1 (BFMMM.h:3705-3722, 3896-3915)
  std::vector<double> ones(std::max({K * D, K * P * D * M, K * M * D, K * 2 * D}), 1.0);
  for (int q = 0; q < h->nch; ++q) {
    const Ctx cq = chain_ctx(c, (unsigned)q);
    HIPCHK(copy_sync(h, cq.tau_eta, ones.data(), sizeof(double) * K * D, hipMemcpyHostToDevice));
    HIPCHK(copy_sync(h, cq.gamma_xi, ones.data(), sizeof(double) * K * P * D * M, hipMemcpyHostToDevice));
    HIPCHK(copy_sync(h, cq.delta_xi, ones.data(), sizeof(double) * K * M * D, hipMemcpyHostToDevice));
    HIPCHK(copy_sync(h, cq.A_xi, ones.data(), sizeof(double) * K * 2 * D, hipMemcpyHostToDevice));
  }
  for (auto& g_ : h->gs)
    for (hipGraphExec_t* g : graph_slots(g_))
      if (*g) { (void)hipGraphExecDestroy(*g); *g = nullptr; }
  h->g_valid = false;
  return 0;
}

extern "C" void bfmmm_destroy(bfmmm_handle* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->st) (void)hipStreamSynchronize(h->st);
  for (auto& g_ : h->gs)
    for (hipGraphExec_t* gp_ : graph_slots(g_)) if (hipGraphExec_t g = *gp_)
      if (g) (void)hipGraphExecDestroy(g);
  for (int q = 1; q < bfmmm_handle::MAX_SUB; ++q) { if (h->sub_st[q]) (void)hipStreamDestroy(h->sub_st[q]); if (h->sub_ev[q]) (void)hipEventDestroy(h->sub_ev[q]); }
  for (void* p : h->allocs) (void)hipFree(p);
  if (h->status_host) (void)hipHostFree(h->status_host);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  if (h->evA) (void)hipEventDestroy(h->evA);
  if (h->evB) (void)hipEventDestroy(h->evB);
  if (h->evC) (void)hipEventDestroy(h->evC);
  if (h->evD) (void)hipEventDestroy(h->evD);
  if (h->st2) (void)hipStreamDestroy(h->st2);
  if (h->st) (void)hipStreamDestroy(h->st);
  delete h;
}

extern "C" int bfmmm_get_basis(bfmmm_handle* h, double* out, int64_t capacity) {
  if (!h || !out) return fail("bfmmm_get_basis: null argument");
  const Dims& d = h->c.d;
  if (d.mv) return fail("bfmmm_get_basis: the multivariate model has no basis");
  const int64_t need = h->n_obs * d.P;
  if (capacity < need) return fail("bfmmm_get_basis: buffer too small");
  if (!h->B_host.empty()) { std::copy(h->B_host.begin(), h->B_host.end(), out); return 0; }
  HIPCHK(hipSetDevice(h->device));
  double* dB = nullptr; double* rec_tmp = nullptr; int* ni_tmp = nullptr; int* err = nullptr;
  struct Tmp { void* p[4] = {nullptr, nullptr, nullptr, nullptr}; ~Tmp() { for (void* q : p) if (q) (void)hipFree(q); } } tmp;   // freed on every path
  HIPCHK(hipMalloc((void**)&dB, sizeof(double) * (size_t)need)); tmp.p[0] = dB;
  HIPCHK(hipMalloc((void**)&rec_tmp, sizeof(double) * (size_t)d.n * d.LREC)); tmp.p[1] = rec_tmp;
  HIPCHK(hipMalloc((void**)&ni_tmp, sizeof(int) * (size_t)d.n)); tmp.p[2] = ni_tmp;
  HIPCHK(hipMalloc((void**)&err, sizeof(int))); tmp.p[3] = err;
  HIPCHK(hipMemsetAsync(err, 0, sizeof(int), h->st));
  launch_stats_functional(h->cfg.basis_degree, d.n, d.P, d.LREC, h->d_off, h->d_t, h->d_y, h->d_knots, h->n_knots,
                          rec_tmp, ni_tmp, dB, err, h->st);
  HIPCHK(hipStreamSynchronize(h->st));
  HIPCHK(copy_sync(h, out, dB, sizeof(double) * (size_t)need, hipMemcpyDeviceToHost));
  return 0;
}

// ---- state marshalling ----------------------------------------------------------------------
// q < 0: the selected chain
static int dyn_get(bfmmm_handle* h, Dyn& dyn, int q = -1) {
  HIPCHK(hipStreamSynchronize(h->st));
  HIPCHK(copy_sync(h, &dyn, chain_ctx(h->c, (unsigned)(q < 0 ? h->sel : q)).dyn, sizeof dyn, hipMemcpyDeviceToHost));
  return 0;
}
static int dyn_put(bfmmm_handle* h, const Dyn& dyn, int q = -1) {
  HIPCHK(copy_sync(h, chain_ctx(h->c, (unsigned)(q < 0 ? h->sel : q)).dyn, &dyn, sizeof dyn, hipMemcpyHostToDevice));
  return 0;
}

extern "C" int bfmmm_select_chain(bfmmm_handle* h, int q) {
  if (!h) return fail("bfmmm_select_chain: null handle");
  if (q < 0 || q >= h->nch) return fail("bfmmm_select_chain: chain index outside the batch");
  h->sel = q;
  return 0;
}
extern "C" int bfmmm_n_chains(const bfmmm_handle* h) { return h ? h->nch : 0; }
extern "C" int bfmmm_set_chain_id_stride(bfmmm_handle* h, uint32_t stride) {
  if (!h || stride < 1) return fail("bfmmm_set_chain_id_stride: bad arguments");
  h->c.chain_id_stride = stride;
  h->g_valid = false;
  return 0;
}

extern "C" int bfmmm_set_state(bfmmm_handle* h, const char* name, const double* v, int64_t count) {
  if (!h || !name || !v) return fail("bfmmm_set_state: null argument");
  h->state_dirty = true;
  HIPCHK(hipSetDevice(h->device));
  const Ctx cs = selc(h);       // the selected chain of the batch
  const Dims& d = cs.d;
  const int n = d.n, K = d.K, P = d.P, M = d.M;
  const std::string s(name);
  auto need = [&](int64_t want) { return count == want ? 0 : fail("bfmmm_set_state(" + s + "): wrong element count"); };
  HIPCHK(hipStreamSynchronize(h->st));
  if (s == "nu" || s == "Phi") {
    std::vector<double> th((size_t)K * (M + 1) * P);
    HIPCHK(copy_sync(h, th.data(), cs.theta, sizeof(double) * th.size(), hipMemcpyDeviceToHost));
    if (s == "nu") {
      if (need((int64_t)K * P)) return 1;
      for (int j = 0; j < K; ++j)
        for (int p = 0; p < P; ++p) th[((size_t)j * (M + 1)) * P + p] = v[j + (size_t)K * p];
    } else {
      if (need((int64_t)K * P * M)) return 1;
      for (int j = 0; j < K; ++j)
        for (int m = 0; m < M; ++m)
          for (int p = 0; p < P; ++p) th[((size_t)j * (M + 1) + m + 1) * P + p] = v[j + (size_t)K * (p + (size_t)P * m)];
    }
    HIPCHK(copy_sync(h, cs.theta, th.data(), sizeof(double) * th.size(), hipMemcpyHostToDevice));
    return 0;
  }
  const int D = d.D;
  if ((s == "eta" || s == "xi") && D > 0) {
    std::vector<double> tx((size_t)K * (M + 1) * D * P);
    HIPCHK(copy_sync(h, tx.data(), cs.thetaX, sizeof(double) * tx.size(), hipMemcpyDeviceToHost));
    if (s == "eta") {        // P x D x K
      if (need((int64_t)P * D * K)) return 1;
      for (int k = 0; k < K; ++k)
        for (int dd = 0; dd < D; ++dd)
          for (int p = 0; p < P; ++p) tx[((size_t)(k * (M + 1)) * D + dd) * P + p] = v[p + (size_t)P * (dd + (size_t)D * k)];
    } else {                 // K arrays P x D x M
      if (need((int64_t)K * P * D * M)) return 1;
      for (int k = 0; k < K; ++k)
        for (int m = 0; m < M; ++m)
          for (int dd = 0; dd < D; ++dd)
            for (int p = 0; p < P; ++p)
              tx[((size_t)(k * (M + 1) + m + 1) * D + dd) * P + p] = v[(size_t)k * P * D * M + p + (size_t)P * (dd + (size_t)D * m)];
    }
    HIPCHK(copy_sync(h, cs.thetaX, tx.data(), sizeof(double) * tx.size(), hipMemcpyHostToDevice));
    return 0;
  }
  struct Arr { const char* nm; double* p; int64_t len; };
  const Arr arrs[] = {{"chi", cs.chi, (int64_t)n * M}, {"Z", cs.Z, (int64_t)n * K}, {"delta", cs.delta, (int64_t)K * M},
                      {"A", cs.Aa, (int64_t)K * 2}, {"gamma", cs.gamma, (int64_t)K * P * M},
                      {"tau_eta", cs.tau_eta, (int64_t)K * D}, {"gamma_xi", cs.gamma_xi, (int64_t)K * P * D * M},
                      {"delta_xi", cs.delta_xi, (int64_t)K * M * D}, {"A_xi", cs.A_xi, (int64_t)K * 2 * D}};
  for (const Arr& a : arrs)
    if (s == a.nm && a.p) {
      if (need(a.len)) return 1;
      HIPCHK(copy_sync(h, a.p, v, sizeof(double) * (size_t)a.len, hipMemcpyHostToDevice));
      return 0;
    }
  Dyn dyn;
  if (dyn_get(h, dyn)) return 1;
  if (s == "pi") { if (need(K)) return 1; for (int k = 0; k < K; ++k) dyn.pi[k] = v[k]; }
  else if (s == "tau") { if (need(K)) return 1; for (int k = 0; k < K; ++k) dyn.tau[k] = v[k]; }
  else if (s == "alpha_3") { if (need(1)) return 1; dyn.alpha3 = v[0]; }
  else if (s == "sigma_sq") { if (need(1)) return 1; dyn.sigma2 = v[0]; }
  else return fail("bfmmm_set_state: unknown name '" + s + "'");
  return dyn_put(h, dyn);
}

extern "C" int bfmmm_get_state(bfmmm_handle* h, const char* name, double* out, int64_t capacity) {
  if (!h || !name || !out) return fail("bfmmm_get_state: null argument");
  HIPCHK(hipSetDevice(h->device));
  const Ctx cs = selc(h);       // the selected chain of the batch
  const Dims& d = cs.d;
  const int n = d.n, K = d.K, P = d.P, M = d.M;
  const std::string s(name);
  auto need = [&](int64_t want) { return capacity >= want ? 0 : fail("bfmmm_get_state(" + s + "): buffer too small"); };
  HIPCHK(hipStreamSynchronize(h->st));
  if (s == "nu" || s == "Phi") {
    std::vector<double> th((size_t)K * (M + 1) * P);
    HIPCHK(copy_sync(h, th.data(), cs.theta, sizeof(double) * th.size(), hipMemcpyDeviceToHost));
    if (s == "nu") {
      if (need((int64_t)K * P)) return 1;
      for (int j = 0; j < K; ++j)
        for (int p = 0; p < P; ++p) out[j + (size_t)K * p] = th[((size_t)j * (M + 1)) * P + p];
    } else {
      if (need((int64_t)K * P * M)) return 1;
      for (int j = 0; j < K; ++j)
        for (int m = 0; m < M; ++m)
          for (int p = 0; p < P; ++p) out[j + (size_t)K * (p + (size_t)P * m)] = th[((size_t)j * (M + 1) + m + 1) * P + p];
    }
    return 0;
  }
  const int D = d.D;
  if ((s == "eta" || s == "xi") && D > 0) {
    std::vector<double> tx((size_t)K * (M + 1) * D * P);
    HIPCHK(copy_sync(h, tx.data(), cs.thetaX, sizeof(double) * tx.size(), hipMemcpyDeviceToHost));
    if (s == "eta") {
      if (need((int64_t)P * D * K)) return 1;
      for (int k = 0; k < K; ++k)
        for (int dd = 0; dd < D; ++dd)
          for (int p = 0; p < P; ++p) out[p + (size_t)P * (dd + (size_t)D * k)] = tx[((size_t)(k * (M + 1)) * D + dd) * P + p];
    } else {
      if (need((int64_t)K * P * D * M)) return 1;
      for (int k = 0; k < K; ++k)
        for (int m = 0; m < M; ++m)
          for (int dd = 0; dd < D; ++dd)
            for (int p = 0; p < P; ++p)
              out[(size_t)k * P * D * M + p + (size_t)P * (dd + (size_t)D * m)] = tx[((size_t)(k * (M + 1) + m + 1) * D + dd) * P + p];
    }
    return 0;
  }
  struct Arr { const char* nm; double* p; int64_t len; };
  const Arr arrs[] = {{"chi", cs.chi, (int64_t)n * M}, {"Z", cs.Z, (int64_t)n * K}, {"delta", cs.delta, (int64_t)K * M},
                      {"A", cs.Aa, (int64_t)K * 2}, {"gamma", cs.gamma, (int64_t)K * P * M},
                      {"tau_eta", cs.tau_eta, (int64_t)K * D}, {"gamma_xi", cs.gamma_xi, (int64_t)K * P * D * M},
                      {"delta_xi", cs.delta_xi, (int64_t)K * M * D}, {"A_xi", cs.A_xi, (int64_t)K * 2 * D}};
  for (const Arr& a : arrs)
    if (s == a.nm && a.p) {
      if (need(a.len)) return 1;
      HIPCHK(copy_sync(h, out, a.p, sizeof(double) * (size_t)a.len, hipMemcpyDeviceToHost));
      return 0;
    }
  Dyn dyn;
  if (dyn_get(h, dyn)) return 1;
  if (s == "pi") { if (need(K)) return 1; for (int k = 0; k < K; ++k) out[k] = dyn.pi[k]; }
  else if (s == "tau") { if (need(K)) return 1; for (int k = 0; k < K; ++k) out[k] = dyn.tau[k]; }
  else if (s == "alpha_3") { if (need(1)) return 1; out[0] = dyn.alpha3; }
  else if (s == "sigma_sq") { if (need(1)) return 1; out[0] = dyn.sigma2; }
  else if (s == "loglik") { if (need(1)) return 1; out[0] = dyn.loglik; }
  else if (s == "status") { if (need(1)) return 1; out[0] = (double)dyn.status; }
  else if (s == "stamps") { if (need(64)) return 1; for (int q = 0; q < 64; ++q) out[q] = (double)(dyn.stamps[q] % 100000000000ULL); }
#ifdef BFMMM_TIMELINE
  else if (s == "fct") { if (need(8)) return 1; unsigned long long w[8]; fetch_fct(w); for (int q = 0; q < 8; ++q) out[q] = (double)(w[q] % 100000000000ULL); }
  else if (s == "zphase") { const int N = 8 * 8192; if (need(N)) return 1; std::vector<unsigned long long> w(N); fetch_zphase(w.data()); for (int q = 0; q < N; ++q) out[q] = (double)w[q]; }
  else if (s == "ztrace") { const int N = 3 * 8192; if (need(N)) return 1; std::vector<unsigned long long> w(N); fetch_ztrace(w.data()); for (int q = 0; q < N; ++q) out[q] = (q % 3 == 1) ? (double)w[q] : (double)(w[q] % 100000000000ULL); }
  else if (s == "wgtrace") { if (need(3072)) return 1; std::vector<unsigned long long> w(3072); fetch_wgtrace(w.data()); for (int q = 0; q < 3072; ++q) out[q] = (q % 3 == 1) ? (double)w[q] : (double)(w[q] % 100000000000ULL); }
#endif
  else return fail("bfmmm_get_state: unknown name '" + s + "'");
  return 0;
}

// Initial states of BFMMM_Nu_Z (BFMMM.h:1039-1071) and BFMMM_Theta (:1210-1235)
extern "C" int bfmmm_init_state(bfmmm_handle* h, int stage, uint64_t seed, uint32_t chain) {
  if (h) h->state_dirty = true;
  if (!h) return fail("bfmmm_init_state: null handle");
  const Dims& d = h->c.d;
  const int n = d.n, K = d.K, P = d.P, M = d.M;
  const RngKey key = make_key(seed, chain, 0, 0);
  std::vector<double> nu((size_t)K * P), chi((size_t)n * M, 0.0), Phi((size_t)K * P * M, 0.0), Z((size_t)n * K), pi(K);
  for (int q = 0; q < K * P; ++q) nu[q] = rnorm(key, UPD_INIT_NU, (uint32_t)q);
  {
    double sum = 0;
    for (int k = 0; k < K; ++k) {
      const double a = h->c.h.c[k] <= 0 ? 10.0 : h->c.h.c[k];
      pi[k] = rgamma(key, UPD_INIT_PI, (uint32_t)k, a, 1.0);
      sum += pi[k];
    }
    for (int k = 0; k < K; ++k) pi[k] /= sum;
  }
  for (int i = 0; i < n; ++i) {
    double g[KMAX], sum = 0;
    for (int k = 0; k < K; ++k) {
      double a = pi[k] * 100;
      if (a <= 0) a = 10;
      g[k] = rgamma(key, UPD_INIT_Z, (uint32_t)(i * K + k), a, 1.0);
      sum += g[k];
    }
    for (int k = 0; k < K; ++k) Z[i + (size_t)n * k] = g[k] / sum;
  }
  if (stage == 1) {
    for (size_t q = 0; q < chi.size(); ++q) chi[q] = rnorm(key, UPD_INIT_CHI, (uint32_t)q);
    for (size_t q = 0; q < Phi.size(); ++q) Phi[q] = rnorm(key, UPD_INIT_PHI, (uint32_t)q);
  }
  std::vector<double> ones((size_t)K * P * M, 1.0);
  const double one = 1.0;
  if (bfmmm_set_state(h, "nu", nu.data(), (int64_t)nu.size()) || bfmmm_set_state(h, "chi", chi.data(), (int64_t)chi.size()) ||
      bfmmm_set_state(h, "Phi", Phi.data(), (int64_t)Phi.size()) || bfmmm_set_state(h, "Z", Z.data(), (int64_t)Z.size()) ||
      bfmmm_set_state(h, "pi", pi.data(), K) || bfmmm_set_state(h, "alpha_3", &one, 1) ||
      bfmmm_set_state(h, "sigma_sq", &one, 1) || bfmmm_set_state(h, "tau", ones.data(), K) ||
      bfmmm_set_state(h, "delta", ones.data(), (int64_t)K * M) || bfmmm_set_state(h, "A", ones.data(), (int64_t)K * 2) ||
      bfmmm_set_state(h, "gamma", ones.data(), (int64_t)K * P * M))
    return 1;
  return 0;
}

// The per-run fields of every chain's Dyn, set on the stream (no host round trip before a run).
__global__ void k_run_begin(Ctx c0, uint32_t first_iter, uint32_t slot_base, uint32_t tt_step, double beta, int state_dirty) {
  if (threadIdx.x != 0) return;
  Dyn* dyn = chain_ctx(c0, blockIdx.x).dyn;
  dyn->iter = first_iter; dyn->slot = first_iter - slot_base; dyn->slot_base = slot_base; dyn->tt_step = tt_step;
  dyn->beta = beta; dyn->status = 0;
  dyn->pend_dir = -1;
  dyn->ll_pending = 0;
  dyn->hyper_pending = 0;
  if (state_dirty) { dyn->zprep_valid = 0; dyn->piprep_valid = 0; }
  dyn->znorm_valid = 0;
  dyn->pi_done = 0;
}

// ---- iteration driver -------------------------------------------------------------------------
struct Plan {
  bool z, pg, factor, chi;
  int z_update, chi_update, use_rss_part;
};

static Plan make_plan(uint32_t mask, int MD) {
  Plan p;
  p.z = (mask & (U_Z | U_PI | U_ALPHA3)) != 0;   // (forced on by the caller for covariate-adjusted models: s~_i)
  p.z_update = (mask & U_Z) ? 1 : 0;
  p.pg = (mask & (U_PHI | U_NU | U_SIGMA)) != 0;
  p.factor = true;   // k_factor prepares r = t - H theta for the sweep and draws job_hyper's variates
  p.chi_update = ((mask & U_CHI) && MD > 1) ? 1 : 0;
  p.chi = p.chi_update || ((mask & U_LOGLIK) && !(mask & U_SIGMA));
  p.use_rss_part = p.chi ? 1 : 0;
  return p;
}

// One Gibbs iteration on the sampler's stream:
//   k_curve_z -> k_pair_gram (+1 workgroup: pi/alpha_3) -> k_pg_reduce -> k_factor -> k_sweep
//   -> k_curve_chi (+1 workgroup: delta/A/gamma/tau) -> k_loglik
// The scalar updates ride inside the wide kernels, so the replayed graph is a single chain of
// seven kernels with no cross-queue dependencies.
// skip_z: the Z update of this iteration was already done by the previous iteration's k_curve_chi (fuse_z there).
// trail_z: the iteration ends with the stand-alone Z update of the NEXT iteration, in its lean form (the proposals were prepared
// by this iteration's k_factor): sweeps without a chi pass cannot fuse the Z update into k_curve_chi, but they can still run
// it in the order "first Z of the run, then bodies [pair_gram .. chi, next Z]".
// pk / pack: the (sub-)batch runs its pair-Gram contraction through k_pair_gram_pack (chain batches, long curve sets)
static void launch_iteration(bfmmm_handle* h, const Ctx& c, const Plan& p, int NKS, int KS, hipStream_t st,
                             std::vector<hipEvent_t>* evs, bool skip_z = false, bool fuse_z = false, bool trail_z = false,
                             const PgPack* pk = nullptr, double* pack = nullptr, int side = -1) {
  auto mark = [&]() {
    if (evs) { hipEvent_t e; (void)hipEventCreate(&e); (void)hipEventRecord(e, st); evs->push_back(e); }
  };
  mark();
  if (p.z && !skip_z) launch_curve(c, 0, p.z_update, st);
  mark();
  const bool packed = p.pg && pk;
  Ctx cf = c;
  cf.pi_in_factor = packed ? 1 : 0;      // (the pi / alpha_3 job: an extra workgroup of k_pair_gram, or -- packed path -- of k_factor)
  if (packed) {
    launch_pair_gram_pack(c, *pk, pack, st);      // (contraction + reduction)
    mark();
  } else {
    launch_pair_gram(c, p.pg ? 1 : 0, NKS, KS, st);
    mark();
    if (p.pg) launch_pg_reduce(c, NKS, st);
  }
  mark();
  if (p.factor) launch_factor(cf, st);
  mark();
  if (launch_sweep(c, st)) h->launch_error = 1;
  mark();
  // (a trailing lean Z launch carries the scalar job of k_curve_chi as its first workgroup: sweeps without a chi pass -- the
  //  only ones that run deferred -- then need no k_curve_chi launch at all)
  if (!(trail_z && !p.chi)) launch_curve(c, 1, (p.chi ? (p.chi_update ? 2 : 1) : 0) | (fuse_z ? 16 : 0), st);
  if (trail_z) launch_curve(c, 0, p.z_update | 2, st);
  if (c.d.D > 0) launch_cov_block(c, st);      // eta, tau_eta, Xi, delta_xi, A_xi, gamma_xi (+ residual sums)
  mark();
  if (!c.defer_loglik) launch_loglik(c, p.use_rss_part, 0, st);      // otherwise: job_hyper + the next k_pair_gram (scalar_jobs.hpp)
  mark();
}

static int run_impl(bfmmm_handle* h, uint32_t mask, int first_iter, int n_iters, uint64_t seed, uint32_t chain,
                    int phi_chi_zero, double beta, uint32_t tt_step, bool prepare_only = false) {
  if (!h) return fail("bfmmm_run: null handle");
  // BFMMM_TRACE_RUN=1: host-side time stamps of the phases of a call (diagnostic: where the fixed cost of a short run goes)
  static const bool trace_run = getenv("BFMMM_TRACE_RUN") && atoi(getenv("BFMMM_TRACE_RUN")) != 0;
  const auto t_begin = std::chrono::steady_clock::now();
  auto tmark = [&](const char* what) {
    if (trace_run && !prepare_only)
      fprintf(stderr, "[bfmmm_run] %-28s %8.1f us\n", what, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_begin).count());
  };
  if (n_iters < 0 || first_iter < h->slot_base || first_iter - h->slot_base + n_iters > h->T)
    return fail("bfmmm_run: iterations exceed the allocated chain");
  HIPCHK(hipSetDevice(h->device));
  Ctx c = h->c;
  const int MD = phi_chi_zero ? 1 : (c.d.M + 1);
  set_md(c.d, MD);
  c.seed = seed; c.chain = chain; c.mask = mask;
  int NTG, NKS, KS;
  pg_geometry(c.d, NTG, NKS, KS);
  if ((size_t)NKS * c.d.NT * 256 > h->pg_part_doubles) return fail("bfmmm_run: internal workspace too small");
  // Long curve sets (beyond the cache-resident sizes): k_pair_gram's k-slices are capped by its LDS staging (192 curves), so at
  // n = 262144 it writes 1366 slabs of partial tiles -- as many bytes as the records themselves.  k_pair_gram_pack walks a slice of
  // ANY length in 16-curve chunks with persistent accumulators: about 128 slices whatever n, chosen from n alone so that a chain
  // of a batch and the same chain alone sum in the same order.
  bool long_set = false;
  {
    const char* e = getenv("BFMMM_PG_PACK");
    PgPack gt;
    const int KSb = ((c.d.n + 127) / 128 + 15) / 16 * 16, NKSb = (c.d.n + KSb - 1) / KSb;
    if (c.d.n > 16384 && !(e && atoi(e) == 0) && pgp_geometry(c.d, 1, KSb, NKSb, gt)) { long_set = true; KS = KSb; NKS = NKSb; }
  }
  Plan plan = make_plan(mask, MD);
  if (c.d.D > 0) { plan.z = true; plan.chi = true; plan.use_rss_part = 1; }
  // pair-Gram through k_pair_gram_pack: batches of four or more chains -- warm-start and Nu_Z sweeps alike (measured, chain-iterations/s
  // plain / packed: 4 warm chains 35.3 k / 40.4 k, 6: 44.9 / 46.9; 4 Nu_Z chains 65.0 / 68.3, 6: 84.4 / 92.6, 8: 105 / 114; two chains:
  // no gain) -- and long curve sets (BFMMM_PG_PACK=0 / 1 switches it off / forces it wherever its limits allow: both kernels sum
  // in the same order, the results are bit-identical)
  auto want_pack = [&](int cnt) {
    const char* e = getenv("BFMMM_PG_PACK");
    if (long_set) return true;
    if (e) return atoi(e) != 0;
    (void)cnt;
    return h->nch >= 4;
  };
  auto pack_for = [&](int slot, int cnt, PgPack& g, double** buf) -> int {      // 0: packed path ready, 1: not applicable, -1: error
    *buf = nullptr;
    if (!plan.pg || !want_pack(cnt) || !pgp_geometry(c.d, cnt, KS, NKS, g)) return 1;
    const size_t need = pgp_pack_doubles(g);
    if (h->pg_pack_doubles[slot] < need) {
      double* nb = nullptr;
      if (hipMalloc((void**)&nb, need * sizeof(double)) != hipSuccess) { (void)hipGetLastError(); return 1; }      // (no room: the plain kernel needs no extra buffer)
      h->allocs.push_back(nb);      // (an outgrown buffer stays until the handle is destroyed: graphs captured earlier may still hold it)
      h->pg_pack[slot] = nb; h->pg_pack_doubles[slot] = need;
    }
    *buf = h->pg_pack[slot];
    return 0;
  };
  // without covariates the iteration ends with k_curve_chi: its scalar-job workgroup advances the counters and the
  // log-likelihood is reduced by the next iteration's k_pair_gram job (one kernel boundary less per iteration)
  c.defer_loglik = (c.d.D == 0) ? 1 : 0;
  // single chain: the scalar job of k_curve_chi rides the next iteration's k_pair_gram instead (its grid has NKS - 1 idle extra
  // workgroups); the run's flush kernel runs the last one.  BFMMM_DEFER_HYPER=0 / 1 overrides (diagnostic).
  {
    static const int env = getenv("BFMMM_DEFER_HYPER") ? atoi(getenv("BFMMM_DEFER_HYPER")) : -1;
    const bool can = c.defer_loglik && plan.pg && plan.chi && h->nch == 1 && NKS >= 2 && !want_pack(1);
    c.defer_hyper = (can && (env < 0 || env != 0)) ? 1 : 0;
  }
  c.ll_use_part = plan.use_rss_part;
  h->last_md = MD;
  if (!prepare_only) {
    const int64_t tkey = ((int64_t)MD << 32) | (mask & (U_PHI | U_NU));
    if (h->tab_key != tkey) { launch_sweep_tables(c, h->st); h->tab_key = tkey; }
    // every chain of the batch starts the run at the same iteration
    hipLaunchKernelGGL(k_run_begin, dim3(h->nch), dim3(64), 0, h->st, h->c, (uint32_t)first_iter, (uint32_t)h->slot_base, tt_step, beta,
                       h->state_dirty ? 1 : 0);
    h->state_dirty = false;
    for (int f = 0; f < FAM_COUNT; ++f) { h->fam_ms[f] = 0; h->fam_launches[f] = 0; }
    HIPCHK(hipEventRecord(h->ev0, h->st));
    tmark("run_begin queued");
  }
  if (h->profile && prepare_only) return 0;
  bool whole_launched = false;      // the run went out as ONE graph that ends with the closing kernel (short runs, below)
  if (h->profile) {
    // the same kernels as the graph path (including the fused chi + next-Z launches), bracketed by events
    const bool pfuse = plan.z && plan.z_update && plan.chi && c.d.D == 0 && n_iters >= 2 && tt_step == 0;
    for (int it = 0; it < n_iters; ++it) {
      std::vector<hipEvent_t> evs;
      const bool skip_z = pfuse && it > 0, fuse_z = pfuse && it + 1 < n_iters;
      PgPack pkp; double* packp = nullptr;
      const bool use_p = pack_for(bfmmm_handle::MAX_SUB, h->nch, pkp, &packp) == 0;
      launch_iteration(h, c, plan, NKS, KS, h->st, &evs, skip_z, fuse_z, false, use_p ? &pkp : nullptr, packp, bfmmm_handle::MAX_SUB);
      HIPCHK(hipStreamSynchronize(h->st));
      const int fams[7] = {FAM_Z, FAM_PG, FAM_REDUCE, FAM_FACTOR, FAM_SWEEP, FAM_CHI, FAM_LOGLIK};
      const bool ran[7] = {plan.z && !skip_z, true, plan.pg, plan.factor, true, true, c.defer_loglik == 0};
      for (int q = 0; q < 7; ++q) {
        float ms = 0;
        (void)hipEventElapsedTime(&ms, evs[q], evs[q + 1]);
        if (ran[q]) { h->fam_ms[fams[q]] += ms; h->fam_launches[fams[q]] += 1; }
      }
      for (hipEvent_t e : evs) (void)hipEventDestroy(e);
    }
  } else if (n_iters > 0) {
    // A chain batch runs as SUB-BATCHES on separate streams: the kernels are the same (a sub-batch is a Ctx whose per-chain
    // pointers start at its first chain), but while one sub-batch is in its narrow kernels -- k_sweep_fast is one workgroup per
    // chain, k_pg_reduce and the factorisations a few dozen -- the others' wide per-curve kernels have the CUs.
    const char* env_split = getenv("BFMMM_BATCH_SPLIT");
    int nsub = (h->nch >= 4) ? 2 : 1;
    if (env_split) nsub = std::max(1, std::min({atoi(env_split), (int)bfmmm_handle::MAX_SUB, h->nch / 2}));
    // (the captured graphs bake in the kernel instances the launchers chose: the cache key carries the switches that choose them)
    const int pack_mode = (getenv("BFMMM_PG_PACK") ? 1 + (atoi(getenv("BFMMM_PG_PACK")) != 0) : 0) + 4 * (bfmmm::g_exact_instances ? 1 : 0);
    const bool reuse = h->g_valid && h->g_mask == mask && h->g_md == MD && h->g_seed == seed && h->g_chain == chain && h->g_nsub == nsub && h->g_pack_mode == pack_mode;
    if (!reuse) {
      for (auto& g_ : h->gs)
        for (hipGraphExec_t* g : graph_slots(g_))
          if (*g) { (void)hipGraphExecDestroy(*g); *g = nullptr; }
      h->g_mask = mask; h->g_md = MD; h->g_seed = seed; h->g_chain = chain; h->g_nsub = nsub; h->g_pack_mode = pack_mode; h->g_valid = true;
    }
    struct Sub { Ctx c; hipStream_t st; hipGraphExec_t *gN, *gFN, *gL, *gR, *gFR; PgPack pk; double* pack; bool use_pack; int slot; };
    Sub subs[bfmmm_handle::MAX_SUB];
    h->sub_st[0] = h->st;
    for (int q = 0, q0 = 0; q < nsub; ++q) {
      const int cnt = h->nch / nsub + (q < h->nch % nsub ? 1 : 0);
      if (!h->sub_st[q]) HIPCHK(hipStreamCreateWithFlags(&h->sub_st[q], hipStreamNonBlocking));
      if (!h->sub_ev[q]) HIPCHK(hipEventCreateWithFlags(&h->sub_ev[q], hipEventDisableTiming));
      bfmmm_handle::GraphSet& g_ = h->gs[q];
      subs[q] = Sub{chain_ctx(c, (unsigned)q0), h->sub_st[q], &g_.gN, &g_.gFN, &g_.gL, g_.gR, g_.gFR};
      subs[q].c.nch = cnt;
      subs[q].use_pack = pack_for(q, cnt, subs[q].pk, &subs[q].pack) == 0;
      subs[q].slot = q;
      q0 += cnt;
    }
    // graphs are captured on demand: kind 0 = full iterations, 1 = fused bodies (no Z in front, chi + next Z at the end),
    // 2 = the closing iteration of a fused run (no Z in front, plain chi)
    std::vector<std::pair<hipGraphExec_t, hipStream_t>> fresh;      // graphs instantiated by this call
    auto ensure = [&](const Sub& sb, hipGraphExec_t* g, int kind, int reps) -> int {
      if (*g) return 0;
      std::lock_guard<std::mutex> lock(g_capture_mutex);
      hipGraph_t graph = nullptr;
      HIPCHK(hipStreamBeginCapture(sb.st, hipStreamCaptureModeRelaxed));
      for (int r = 0; r < reps; ++r) launch_iteration(h, sb.c, plan, NKS, KS, sb.st, nullptr, kind != 0, kind == 1, kind == 3, sb.use_pack ? &sb.pk : nullptr, sb.pack, sb.slot);
      const hipError_t ec = hipStreamEndCapture(sb.st, &graph);      // always leaves capture mode, also after a failed launch
      if (ec != hipSuccess) { if (graph) (void)hipGraphDestroy(graph); HIPCHK(ec); }
      const hipError_t ei = hipGraphInstantiate(g, graph, nullptr, nullptr, 0);
      (void)hipGraphDestroy(graph);
      HIPCHK(ei);
      (void)hipGraphUpload(*g, sb.st);      // (set-up: the first launch of a graph otherwise pays for its upload)
      fresh.push_back({*g, sb.st});
      return 0;
    };
    // a run of nrep repetitions = nrep / GRAPH_UNROLL replays of the unrolled graph + ONE graph holding the remainder
    // (re-captured only when the remainder changes), so that a short run costs two or three graph launches, not one per iteration
    auto ensure_rem = [&](const Sub& sb, hipGraphExec_t* garr, int kind, int rem) -> int {
      if (rem <= 0) return 0;
      return ensure(sb, &garr[rem], kind, rem);
    };
    if (!h->status_host) {
      HIPCHK(hipHostMalloc((void**)&h->status_host, sizeof(uint32_t) * (size_t)h->nch, hipHostMallocMapped));
      if (hipHostGetDevicePointer((void**)&h->status_dev, h->status_host, 0) != hipSuccess) { (void)hipGetLastError(); h->status_dev = nullptr; }
    }
    const bool fuse = plan.z && plan.z_update && plan.chi && c.d.D == 0 && n_iters >= 2 && tt_step == 0;
    // sweeps whose Z update cannot ride in k_curve_chi (no chi pass: the Nu_Z stage) still run it at the END of the previous
    // iteration's body, as the lean stand-alone kernel (kind 3 bodies)
    const bool defer = !fuse && plan.z && plan.z_update && plan.factor && (mask & U_Z) && c.d.D == 0 && c.d.K <= 4 && c.d.BW <= 5 &&
                       n_iters >= 2 && tt_step == 0;
    const int body_kind = fuse ? 1 : 3;
    const bool bodies = fuse || defer;
    const int nrep = bodies ? n_iters - 1 : n_iters;                // fused / deferred run: n_iters - 1 bodies + the closing iteration
    const int nfull = nrep / GRAPH_UNROLL, rem = nrep % GRAPH_UNROLL;
    // A SHORT run on one stream is ONE graph: the first Z update, the bodies, the closing iteration and the closing kernel --
    // one graph launch instead of three and two kernel launches (a 20-iteration call: ~30 us of host time, 1.5 us per step).
    // Keyed by the number of iterations (re-captured when it changes); BFMMM_WHOLE_GRAPH=0 switches it off.
    static const int env_whole = getenv("BFMMM_WHOLE_GRAPH") ? atoi(getenv("BFMMM_WHOLE_GRAPH")) : 1;
    constexpr int WHOLE_MAX = 64;
    const bool whole = env_whole && bodies && nsub == 1 && n_iters <= WHOLE_MAX && c.defer_loglik && h->status_dev != nullptr;
    if (whole) {
      const Sub& sb = subs[0];
      bfmmm_handle::GraphSet& g_ = h->gs[0];
      if (g_.gW && g_.gW_n != n_iters) { (void)hipGraphExecDestroy(g_.gW); g_.gW = nullptr; }
      if (!g_.gW) {
        std::lock_guard<std::mutex> lock(g_capture_mutex);
        hipGraph_t graph = nullptr;
        HIPCHK(hipStreamBeginCapture(sb.st, hipStreamCaptureModeRelaxed));
        launch_curve(sb.c, 0, plan.z_update, sb.st);              // Z of the first iteration
        for (int r = 0; r < nrep; ++r) launch_iteration(h, sb.c, plan, NKS, KS, sb.st, nullptr, true, body_kind == 1, body_kind == 3, sb.use_pack ? &sb.pk : nullptr, sb.pack, sb.slot);
        launch_iteration(h, sb.c, plan, NKS, KS, sb.st, nullptr, true, false, false, sb.use_pack ? &sb.pk : nullptr, sb.pack, sb.slot);
        launch_loglik_flush(c, sb.st, h->status_dev);
        const hipError_t ec = hipStreamEndCapture(sb.st, &graph);
        if (ec != hipSuccess) { if (graph) (void)hipGraphDestroy(graph); HIPCHK(ec); }
        const hipError_t ei = hipGraphInstantiate(&g_.gW, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        HIPCHK(ei);
        (void)hipGraphUpload(g_.gW, sb.st);
        g_.gW_n = n_iters;
        fresh.push_back({g_.gW, sb.st});
      }
    }
    for (int s = 0; s < nsub && !whole; ++s) {
      const Sub& sb = subs[s];
      if (!bodies) {
        if ((nfull > 0 && ensure(sb, sb.gN, 0, GRAPH_UNROLL)) || ensure_rem(sb, sb.gR, 0, rem)) return 1;
      } else {
        if ((nfull > 0 && ensure(sb, sb.gFN, body_kind, GRAPH_UNROLL)) || ensure_rem(sb, sb.gFR, body_kind, rem) || ensure(sb, sb.gL, 2, 1)) return 1;
      }
    }
    if (prepare_only) {
      // DRY LAUNCH (set-up): the first launch of an instantiated graph costs the device 13 - 20 us more than every later one,
      // upload or not (measured: three fresh graphs = +40 us on a 20-iteration run, tools/gpu/trace_run.py).  So every graph this
      // call instantiated is launched once here, on the real state, between a snapshot and a restore of the chains' work state
      // (everything of the per-chain arenas but the chain storage; the slots a dry launch writes are the first slots of the
      // coming run, which rewrites them).  BFMMM_DRY_LAUNCH=0 switches it off.
      const char* ed = getenv("BFMMM_DRY_LAUNCH");
      if (!fresh.empty() && !(ed && atoi(ed) == 0)) {
        const size_t wb1 = (size_t)((char*)h->c.c_nu - (char*)h->c.dyn);
        const size_t wb2 = (c.d.D > 0 && h->arena_cov) ? (size_t)((char*)h->c.c_eta - (char*)h->c.thetaX) : 0;
        const size_t need = (size_t)h->nch * (wb1 + wb2);
        if (h->dry_snap_bytes < need) {
          char* nb = nullptr;
          if (hipMalloc((void**)&nb, need) != hipSuccess) { (void)hipGetLastError(); return 0; }      // (no room: the run's first launch pays)
          h->allocs.push_back(nb);
          h->dry_snap = nb; h->dry_snap_bytes = need;
        }
        const size_t cb1 = h->nch > 1 ? h->c.chain_bytes : wb1, cb2 = h->nch > 1 ? h->c.chain_bytes_cov : wb2;
        char* snap2 = h->dry_snap + (size_t)h->nch * wb1;
        for (int q = 0; q < nsub; ++q) HIPCHK(hipStreamSynchronize(subs[q].st));
        HIPCHK(hipMemcpy2DAsync(h->dry_snap, wb1, h->c.dyn, cb1, wb1, (size_t)h->nch, hipMemcpyDeviceToDevice, h->st));
        if (wb2) HIPCHK(hipMemcpy2DAsync(snap2, wb2, h->c.thetaX, cb2, wb2, (size_t)h->nch, hipMemcpyDeviceToDevice, h->st));
        const int64_t tkey = ((int64_t)MD << 32) | (mask & (U_PHI | U_NU));
        if (h->tab_key != tkey) { launch_sweep_tables(c, h->st); h->tab_key = tkey; }
        // (a few milliseconds of it: the device's clocks keep rising over the first ~5 ms of activity after an idle period -- a
        //  capture is one --, measured as 1240 -> 1219 -> 1212 -> 1205 us of device time for four consecutive 20-iteration runs;
        //  BFMMM_DRY_LAUNCH_MS sets the duration, default 10: us per step of the 20-step form 64.6 / 63.7 / 63.1 with one launch / 4 ms / 12 ms)
        const char* ems = getenv("BFMMM_DRY_LAUNCH_MS");
        const double want_ms = ems ? atof(ems) : 10.0;
        const auto t_dry = std::chrono::steady_clock::now();
        for (int round = 0; round < 64; ++round) {
          for (auto& fg : fresh) {
            hipLaunchKernelGGL(k_run_begin, dim3(h->nch), dim3(64), 0, h->st, h->c, (uint32_t)first_iter, (uint32_t)h->slot_base, tt_step, beta, 0);
            HIPCHK(hipStreamSynchronize(h->st));
            HIPCHK(hipGraphLaunch(fg.first, fg.second));
            HIPCHK(hipStreamSynchronize(fg.second));
          }
          if (std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_dry).count() >= want_ms) break;
        }
        HIPCHK(hipMemcpy2DAsync(h->c.dyn, cb1, h->dry_snap, wb1, wb1, (size_t)h->nch, hipMemcpyDeviceToDevice, h->st));
        if (wb2) HIPCHK(hipMemcpy2DAsync(h->c.thetaX, cb2, snap2, wb2, wb2, (size_t)h->nch, hipMemcpyDeviceToDevice, h->st));
        HIPCHK(hipStreamSynchronize(h->st));
        HIPCHK(hipGetLastError());
      }
      return 0;
    }
    HIPCHK(hipEventRecord(h->ev0, h->st));
    for (int q = 1; q < nsub; ++q) { HIPCHK(hipEventRecord(h->evA, h->st)); HIPCHK(hipStreamWaitEvent(subs[q].st, h->evA, 0)); }      // k_run_begin first
    if (whole) { HIPCHK(hipGraphLaunch(h->gs[0].gW, subs[0].st)); tmark("graph (whole run) queued"); whole_launched = true; }
    for (int s = 0; s < nsub && !whole; ++s) {
      const Sub& sb = subs[s];
      if (!bodies) {
        for (int q = 0; q < nfull; ++q) HIPCHK(hipGraphLaunch(*sb.gN, sb.st));
        if (rem > 0) HIPCHK(hipGraphLaunch(sb.gR[rem], sb.st));
      } else {
        launch_curve(sb.c, 0, plan.z_update, sb.st);              // Z of the first iteration
        tmark("first Z queued");
        for (int q = 0; q < nfull; ++q) { HIPCHK(hipGraphLaunch(*sb.gFN, sb.st)); tmark("graph (full) queued"); }
        if (rem > 0) HIPCHK(hipGraphLaunch(sb.gFR[rem], sb.st));
        HIPCHK(hipGraphLaunch(*sb.gL, sb.st));
      }
    }
    for (int q = 1; q < nsub; ++q) { HIPCHK(hipEventRecord(h->sub_ev[q], subs[q].st)); HIPCHK(hipStreamWaitEvent(h->st, h->sub_ev[q], 0)); }
  }
  if (prepare_only) return 0;
  if (!h->status_host) {
    HIPCHK(hipHostMalloc((void**)&h->status_host, sizeof(uint32_t) * (size_t)h->nch, hipHostMallocMapped));
    if (hipHostGetDevicePointer((void**)&h->status_dev, h->status_host, 0) != hipSuccess) { (void)hipGetLastError(); h->status_dev = nullptr; }
  }
  // the status words reach the host from the run's last kernel when there is one that can carry them (the deferred
  // log-likelihood's flush; the fill kernels behind it never touch a status word), by a queued copy otherwise
  const bool status_by_kernel = c.defer_loglik && n_iters > 0 && h->status_dev != nullptr;
  if (c.defer_loglik && n_iters > 0 && !whole_launched) launch_loglik_flush(c, h->st, status_by_kernel ? h->status_dev : nullptr);      // (a whole-run graph ends with it)
  // chain slots of blocks this sweep does not touch hold the (constant) current value
  if (!(mask & U_Z)) launch_fill_slots(c, c.c_Z, c.Z, (size_t)c.d.n * c.d.K, first_iter - h->slot_base, first_iter - h->slot_base + n_iters, h->st);
  if (!plan.chi_update) launch_fill_slots(c, c.c_chi, c.chi, (size_t)c.d.n * c.d.M, first_iter - h->slot_base, first_iter - h->slot_base + n_iters, h->st);
  HIPCHK(hipEventRecord(h->ev1, h->st));
  if (!status_by_kernel)      // the chains' status words: one strided copy queued behind the run
    HIPCHK(hipMemcpy2DAsync(h->status_host, sizeof(uint32_t), &h->c.dyn->status, h->nch > 1 ? h->c.chain_bytes : sizeof(uint32_t), sizeof(uint32_t),
                            (size_t)h->nch, hipMemcpyDeviceToHost, h->st));
  tmark("all queued");
  HIPCHK(hipStreamSynchronize(h->st));
  tmark("synchronised");
  if (h->launch_error) { h->launch_error = 0; return fail("bfmmm_run: problem size exceeds the sweep kernel's LDS (5 A P doubles + A^2 ints must fit 160 KB)"); }
  HIPCHK(hipGetLastError());
  float ms = 0;
  HIPCHK(hipEventElapsedTime(&ms, h->ev0, h->ev1));
  h->fam_ms[FAM_TOTAL] = ms;
  h->fam_launches[FAM_TOTAL] = n_iters;
  for (int q = 0; q < h->nch; ++q) {
    const uint32_t status = h->status_host[q];
    if (status & 8u) return fail("bfmmm_run: internal error (the pi / alpha_3 job of k_factor did not signal)");
    if (status & 4u) return fail("bfmmm_run: internal error (a hand-off inside the sweep kernel timed out)");
    if (status & 2u) return fail("bfmmm_run: internal error (fused Z update without prepared proposals)");
    if (status & 1u)
      return fail("bfmmm_run: a conditional precision matrix was not positive definite");
  }
  return 0;
}

extern "C" int bfmmm_set_slot_base(bfmmm_handle* h, int base) {
  if (!h || base < 0) return fail("bfmmm_set_slot_base: bad arguments");
  h->slot_base = base;
  return 0;
}

extern "C" int bfmmm_run(bfmmm_handle* h, uint32_t mask, int first_iter, int n_iters, uint64_t seed, uint32_t chain,
                         int phi_chi_zero, double beta) {
  return run_impl(h, mask, first_iter, n_iters, seed, chain, phi_chi_zero, beta, 0);
}

// Captures and instantiates the HIP graphs a bfmmm_run with the same arguments replays (set-up, launches nothing): a caller
// that times a run, or wants its first call to return quickly, pays the capture here instead.
extern "C" int bfmmm_prepare_run(bfmmm_handle* h, uint32_t mask, int first_iter, int n_iters, uint64_t seed, uint32_t chain,
                                 int phi_chi_zero) {
  return run_impl(h, mask, first_iter, n_iters, seed, chain, phi_chi_zero, 1.0, 0, true);
}

// Tempered-transition block of BFMMM_MTT_warm_start (BFMMM.h:1556-1657) for chain iteration `iter`, whose regular
// updates have already run (the sampler's working state is slot `iter`).  The 2 N_t tempered sweeps are ordinary
// iterations of the same kernels with beta from the ladder and the RNG counter word tt_step = l; they write chain
// slot `iter` directly.  CalculateTTAcceptance (CalculateTTAcceptance.h:22-97) depends on a tempered state only
// through (sigma^2, RSS), which every sweep's residual pass leaves in Dyn, so the acceptance costs nothing extra.
// Rejected: the saved state is restored and slot `iter` rewritten from it.  Accepted: as in the reference every block
// of the next iteration starts from the accepted state EXCEPT gamma, which BFMMM.h:1660-1671 does not re-copy.
extern "C" int bfmmm_tempered_transition(bfmmm_handle* h, uint32_t mask, int iter, int N_t, double beta_N_t,
                                         uint64_t seed, uint32_t chain, double* logA_out, int* accepted_out) {
  if (!h) return fail("bfmmm_tempered_transition: null handle");
  if (N_t < 1 || iter < h->slot_base || iter - h->slot_base >= h->T) return fail("bfmmm_tempered_transition: bad arguments");
  if (h->nch != 1) return fail("bfmmm_tempered_transition: a tempered transition accepts or rejects one chain: not available on a chain batch");
  HIPCHK(hipSetDevice(h->device));
  const Ctx& c = h->c;
  const Dims& d = c.d;
  const size_t n_th = (size_t)d.K * (d.M + 1) * d.P, n_chi = (size_t)d.n * d.M, n_Z = (size_t)d.n * d.K,
               n_dl = (size_t)d.K * d.M, n_A = (size_t)d.K * 2, n_g = (size_t)d.K * d.P * d.M;
  // covariate blocks (BFMMM.h:4912-4940): eta / xi rows, tau_eta, gamma_xi, delta_xi, A_xi
  const size_t Dc = (size_t)d.D;
  const size_t n_tx = (size_t)d.K * (d.M + 1) * Dc * d.P, n_te = (size_t)d.K * Dc, n_gx = (size_t)d.K * d.P * Dc * d.M,
               n_dx = (size_t)d.K * d.M * Dc, n_ax = (size_t)d.K * 2 * Dc;
  if (!h->tt_save) { if (dalloc(h, &h->tt_save, n_th + n_chi + n_Z + n_dl + n_A + n_g + n_tx + n_te + n_gx + n_dx + n_ax + 8)) return 1; }
  double* sv = h->tt_save;
  double* sv_th = sv; double* sv_chi = sv_th + n_th; double* sv_Z = sv_chi + n_chi; double* sv_dl = sv_Z + n_Z;
  double* sv_A = sv_dl + n_dl; double* sv_g = sv_A + n_A;
  double* sv_tx = sv_g + n_g; double* sv_te = sv_tx + n_tx; double* sv_gx = sv_te + n_te; double* sv_dx = sv_gx + n_gx;
  double* sv_ax = sv_dx + n_dx;
  auto d2d = [&](double* dst, const double* src, size_t cnt) {
    return cnt ? hipMemcpyAsync(dst, src, sizeof(double) * cnt, hipMemcpyDeviceToDevice, h->st) : hipSuccess;
  };
  HIPCHK(d2d(sv_th, c.theta, n_th)); HIPCHK(d2d(sv_chi, c.chi, n_chi)); HIPCHK(d2d(sv_Z, c.Z, n_Z));
  HIPCHK(d2d(sv_dl, c.delta, n_dl)); HIPCHK(d2d(sv_A, c.Aa, n_A)); HIPCHK(d2d(sv_g, c.gamma, n_g));
  if (Dc > 0) {
    HIPCHK(d2d(sv_tx, c.thetaX, n_tx)); HIPCHK(d2d(sv_te, c.tau_eta, n_te)); HIPCHK(d2d(sv_gx, c.gamma_xi, n_gx));
    HIPCHK(d2d(sv_dx, c.delta_xi, n_dx)); HIPCHK(d2d(sv_ax, c.A_xi, n_ax));
  }
  Dyn dyn0;
  if (dyn_get(h, dyn0)) return 1;
  // geometric ladder, BFMMM.h:1452-1460 (the loop overwrites the last rung: ladder[i] = geom_mult^i)
  std::vector<double> ladder((size_t)N_t, 1.0);
  ladder[N_t - 1] = beta_N_t;
  const double geom_mult = std::pow(beta_N_t, 1.0 / N_t);
  for (int i = 1; i < N_t; ++i) ladder[i] = ladder[i - 1] * geom_mult;
  const int L = 2 * N_t + 1;
  std::vector<double> sig((size_t)L), rss((size_t)L);
  sig[0] = dyn0.sigma2; rss[0] = dyn0.rss;
  int temp_ind = 0;
  for (int l = 1; l < L; ++l) {
    if (run_impl(h, mask | U_LOGLIK, iter, 1, seed, chain, 0, ladder[temp_ind], (uint32_t)l)) return 1;
    Dyn dl;
    if (dyn_get(h, dl)) return 1;
    sig[l] = dl.sigma2; rss[l] = dl.rss;
    if (l < N_t) temp_ind = temp_ind + 1;
    if (l > N_t) temp_ind = temp_ind - 1;
  }
  const double N = (double)d.n_obs_total;
  auto PZ = [&](double b, int l) { return (-(b / 2) * std::log(sig[l])) * N - (b / (2 * sig[l])) * rss[l]; };
  double logA = 0;
  const int m = L - 1;
  for (int i = 0; i < N_t - 1; ++i) {
    logA = logA + PZ(ladder[i + 1], i);
    logA = logA - PZ(ladder[i], i);
    logA = logA - PZ(ladder[i + 1], m - i);
    logA = logA + PZ(ladder[i], m - i);
  }
  const double logu = std::log(runif(make_key(seed, chain, (uint32_t)iter, 0), UPD_TT_ACC, 0));
  const int accepted = (logu < logA) ? 1 : 0;
  if (accepted) {
    HIPCHK(d2d(c.gamma, sv_g, n_g));                 // the next iteration starts from the pre-transition gamma
    HIPCHK(hipStreamSynchronize(h->st));
  } else {
    HIPCHK(d2d(c.theta, sv_th, n_th)); HIPCHK(d2d(c.chi, sv_chi, n_chi)); HIPCHK(d2d(c.Z, sv_Z, n_Z));
    HIPCHK(d2d(c.delta, sv_dl, n_dl)); HIPCHK(d2d(c.Aa, sv_A, n_A)); HIPCHK(d2d(c.gamma, sv_g, n_g));
    if (Dc > 0) {
      HIPCHK(d2d(c.thetaX, sv_tx, n_tx)); HIPCHK(d2d(c.tau_eta, sv_te, n_te)); HIPCHK(d2d(c.gamma_xi, sv_gx, n_gx));
      HIPCHK(d2d(c.delta_xi, sv_dx, n_dx)); HIPCHK(d2d(c.A_xi, sv_ax, n_ax));
    }
    dyn0.zprep_valid = 0;        // the prepared Z proposals were overwritten by the tempered sweeps
    dyn0.piprep_valid = 0;
    if (dyn_put(h, dyn0)) return 1;
    if (run_impl(h, U_LOGLIK, iter, 1, seed, chain, 0, 1.0, 0)) return 1;     // rewrites chain slot `iter` from the state
  }
  Dyn dn;
  if (dyn_get(h, dn)) return 1;
  dn.iter = (uint32_t)(iter + 1); dn.slot = (uint32_t)(iter + 1 - h->slot_base); dn.tt_step = 0; dn.beta = 1.0;
  if (dyn_put(h, dn)) return 1;
  if (logA_out) *logA_out = logA;
  if (accepted_out) *accepted_out = accepted;
  return 0;
}

extern "C" int bfmmm_get_chain(bfmmm_handle* h, const char* name, int n_slots, double* out, int64_t capacity) {
  if (!h || !name || !out) return fail("bfmmm_get_chain: null argument");
  if (n_slots < 0 || n_slots > h->T) return fail("bfmmm_get_chain: n_slots out of range");
  HIPCHK(hipSetDevice(h->device));
  const Ctx c = selc(h);        // the selected chain of the batch
  const Dims& d = c.d;
  const int64_t n = d.n, K = d.K, P = d.P, M = d.M;
  const std::string s(name);
  struct Arr { const char* nm; const double* p; int64_t len; };
  const Arr arrs[] = {{"nu", c.c_nu, K * P}, {"chi", c.c_chi, n * M}, {"Z", c.c_Z, n * K}, {"pi", c.c_pi, K},
                      {"alpha_3", c.c_alpha3, 1}, {"delta", c.c_delta, K * M}, {"A", c.c_A, K * 2},
                      {"sigma_sq", c.c_sigma, 1}, {"gamma", c.c_gamma, K * P * M}, {"Phi", c.c_Phi, K * P * M},
                      {"loglik", c.c_loglik, 1},
                      {"eta", c.c_eta, P * d.D * K}, {"xi", c.c_xi, K * P * d.D * M}, {"tau_eta", c.c_tau_eta, K * d.D},
                      {"gamma_xi", c.c_gamma_xi, K * P * d.D * M}, {"delta_xi", c.c_delta_xi, K * M * d.D},
                      {"A_xi", c.c_A_xi, K * 2 * d.D}};
  HIPCHK(hipStreamSynchronize(h->st));
  for (const Arr& a : arrs)
    if (s == a.nm && a.p) {
      const int64_t want = a.len * n_slots;
      if (capacity < want) return fail("bfmmm_get_chain(" + s + "): buffer too small");
      HIPCHK(copy_sync(h, out, a.p, sizeof(double) * (size_t)want, hipMemcpyDeviceToHost));
      return 0;
    }
  if (s == "tau") {   // stored T_alloc x K column-major on the device; returned n_slots x K column-major
    const int64_t want = (int64_t)n_slots * K;
    if (capacity < want) return fail("bfmmm_get_chain(tau): buffer too small");
    for (int k = 0; k < K; ++k)
      HIPCHK(copy_sync(h, out + (size_t)n_slots * k, c.c_tau + (size_t)h->T * k, sizeof(double) * (size_t)n_slots,
                       hipMemcpyDeviceToHost));
    return 0;
  }
  return fail("bfmmm_get_chain: unknown name '" + s + "'");
}

extern "C" int bfmmm_debug_get(bfmmm_handle* h, const char* name, double* out, int64_t capacity, int64_t* count) {
  if (!h || !name || !out || !count) return fail("bfmmm_debug_get: null argument");
  HIPCHK(hipSetDevice(h->device));
  const Ctx c = selc(h);        // the selected chain of the batch
  Dims d = c.d;
  if (h->last_md > 0) set_md(d, h->last_md);
  const std::string s(name);
  HIPCHK(hipStreamSynchronize(h->st));
  if (s == "dims") {
    const double v[] = {(double)d.n, (double)d.K, (double)d.P, (double)d.M, (double)d.BW, (double)d.LG, (double)d.LREC,
                        (double)d.MD, (double)d.A, (double)d.R, (double)d.NT, (double)d.n_obs_total, (double)d.half_sum, c.YY};
    const int64_t cnt = sizeof v / sizeof v[0];
    if (capacity < cnt) return fail("bfmmm_debug_get: buffer too small");
    memcpy(out, v, sizeof v);
    *count = cnt;
    return 0;
  }
  struct Arr { const char* nm; const double* p; int64_t len; };
  const Arr arrs[] = {{"rec", c.rec, (int64_t)d.n * d.LREC}, {"H", c.H, (int64_t)d.R * d.LG}, {"tvec", c.tvec, (int64_t)d.A * d.P},
                      {"Cmat", c.Cmat, (int64_t)d.A * d.P * d.P},
                      {"theta", c.theta, (int64_t)d.K * (d.M + 1) * d.P}};
  for (const Arr& a : arrs)
    if (s == a.nm) {
      if (capacity < a.len) return fail("bfmmm_debug_get(" + s + "): buffer too small");
      HIPCHK(copy_sync(h, out, a.p, sizeof(double) * (size_t)a.len, hipMemcpyDeviceToHost));
      *count = a.len;
      return 0;
    }
  return fail("bfmmm_debug_get: unknown name '" + s + "'");
}

namespace bfmmm { int g_exact_instances = 1; }
extern "C" void bfmmm_set_exact_instances(int enable) { bfmmm::g_exact_instances = enable ? 1 : 0; }

extern "C" int bfmmm_set_profile(bfmmm_handle* h, int enable) {
  if (!h) return fail("bfmmm_set_profile: null handle");
  h->profile = enable ? 1 : 0;
  return 0;
}

extern "C" int bfmmm_get_timing(bfmmm_handle* h, const char* name, double* ms, int64_t* launches) {
  if (!h || !name || !ms || !launches) return fail("bfmmm_get_timing: null argument");
  for (int f = 0; f < FAM_COUNT; ++f)
    if (!strcmp(name, kFamNames[f])) { *ms = h->fam_ms[f]; *launches = h->fam_launches[f]; return 0; }
  return fail("bfmmm_get_timing: unknown name");
}

// ---- final gather of a multi-GPU multi-try over RCCL ------------------------------------------------------------------
// BFMMM_Nu_Z_multiple_try / BFMMM_Theta_est keep the best of 1 + n_try independent chains (src/UserFunctions.cpp:302-325,
// :861-885).  With the chains dealt over several GPUs (one handle per device, all in this process) the only exchange of
// the whole computation is this one: an ncclAllGather of one (score, chain index) pair per device -- every rank then
// knows the winner: largest score, lowest chain index on ties (the reference's `<` at :320 keeps the earlier chain) --
// and the winner's chain arena (state + every chain slot) travels device to device over xGMI (ncclSend / ncclRecv) into
// the selected chain of handles[0], from where the entry point assembles its result.  No collective runs inside a chain.
#include <rccl/rccl.h>

#define NCCLCHK(x)                                                                                   \
  do {                                                                                              \
    ncclResult_t r_ = (x);                                                                          \
    if (r_ != ncclSuccess) {                                                                        \
      char buf_[512];                                                                               \
      snprintf(buf_, sizeof buf_, "RCCL error %s at %s:%d (%s)", ncclGetErrorString(r_), __FILE__, __LINE__, #x); \
      return fail(buf_);                                                                            \
    }                                                                                               \
  } while (0)

// what bfmmm_gather_best holds for the length of the call: the communicators and the per-device (score, id) buffers are
// released on EVERY return path
struct GatherGuard {
  std::vector<ncclComm_t> comms;
  std::vector<int> devs;
  std::vector<double*> bufs;
  ~GatherGuard() {
    for (ncclComm_t cm : comms) if (cm) (void)ncclCommDestroy(cm);
    for (size_t i = 0; i < bufs.size(); ++i)
      if (bufs[i]) { (void)hipSetDevice(devs[i / 2]); (void)hipFree(bufs[i]); }
  }
};

extern "C" int bfmmm_gather_best(bfmmm_handle* const* handles, int n_handles, const double* scores, const int32_t* chain_ids,
                                 int* winner_out) {
  if (!handles || !scores || !chain_ids || n_handles < 1) return fail("bfmmm_gather_best: bad arguments");
  const int G = n_handles;
  std::vector<int> devs(G);
  for (int g = 0; g < G; ++g) {
    if (!handles[g]) return fail("bfmmm_gather_best: null handle");
    devs[g] = handles[g]->device;
    if (handles[g]->c.chain_bytes != handles[0]->c.chain_bytes || handles[g]->c.chain_bytes_cov != handles[0]->c.chain_bytes_cov)
      return fail("bfmmm_gather_best: the handles were not created with the same configuration");
    for (int g2 = 0; g2 < g; ++g2)
      if (devs[g2] == devs[g]) return fail("bfmmm_gather_best: one handle per device");
  }
  GatherGuard gg;
  gg.devs = devs;
  gg.comms.assign(G, nullptr);
  gg.bufs.assign(2 * (size_t)G, nullptr);
  std::vector<ncclComm_t>& comms = gg.comms;
  NCCLCHK(ncclCommInitAll(comms.data(), G, devs.data()));
  // (score, chain index) pairs: send 2 doubles, receive 2 G
  std::vector<double*> sbuf(G, nullptr), rbuf(G, nullptr);
  for (int g = 0; g < G; ++g) {
    HIPCHK(hipSetDevice(devs[g]));
    HIPCHK(hipMalloc((void**)&gg.bufs[2 * g], sizeof(double) * 2));
    HIPCHK(hipMalloc((void**)&gg.bufs[2 * g + 1], sizeof(double) * 2 * (size_t)G));
    sbuf[g] = gg.bufs[2 * g];
    rbuf[g] = gg.bufs[2 * g + 1];
    const double pair[2] = {scores[g], (double)chain_ids[g]};
    HIPCHK(copy_sync(handles[g], sbuf[g], pair, sizeof pair, hipMemcpyHostToDevice));
  }
  NCCLCHK(ncclGroupStart());
  for (int g = 0; g < G; ++g) NCCLCHK(ncclAllGather(sbuf[g], rbuf[g], 2, ncclDouble, comms[g], handles[g]->st));
  NCCLCHK(ncclGroupEnd());
  std::vector<double> all(2 * (size_t)G);
  int winner = -1;
  for (int g = 0; g < G; ++g) {          // every rank holds the same table and takes the same decision
    HIPCHK(hipSetDevice(devs[g]));
    HIPCHK(hipStreamSynchronize(handles[g]->st));
    HIPCHK(copy_sync(handles[g], all.data(), rbuf[g], sizeof(double) * all.size(), hipMemcpyDeviceToHost));
    int w = -1;
    for (int r = 0; r < G; ++r) {
      if (!(all[2 * r] == all[2 * r])) continue;             // a rank without a valid chain reports NaN
      if (w < 0 || all[2 * r] > all[2 * w] || (all[2 * r] == all[2 * w] && all[2 * r + 1] < all[2 * w + 1])) w = r;
    }
    if (g == 0) winner = w;
    else if (w != winner) return fail("bfmmm_gather_best: ranks disagree on the winner");
  }
  if (winner < 0) return fail("bfmmm_gather_best: no rank holds a valid chain");
  if (winner != 0) {
    bfmmm_handle* hs = handles[winner], *hd = handles[0];
    const size_t cb = hd->c.chain_bytes, cbc = hd->c.chain_bytes_cov;
    if (!hs->arena || !hd->arena || (cbc && (!hs->arena_cov || !hd->arena_cov))) return fail("bfmmm_gather_best: a handle has no chain arena");
    NCCLCHK(ncclGroupStart());
    NCCLCHK(ncclSend(hs->arena + (size_t)hs->sel * cb, cb, ncclChar, 0, comms[winner], hs->st));
    NCCLCHK(ncclRecv(hd->arena + (size_t)hd->sel * cb, cb, ncclChar, winner, comms[0], hd->st));
    if (cbc) {
      NCCLCHK(ncclSend(hs->arena_cov + (size_t)hs->sel * cbc, cbc, ncclChar, 0, comms[winner], hs->st));
      NCCLCHK(ncclRecv(hd->arena_cov + (size_t)hd->sel * cbc, cbc, ncclChar, winner, comms[0], hd->st));
    }
    NCCLCHK(ncclGroupEnd());
    HIPCHK(hipSetDevice(devs[winner]));
    HIPCHK(hipStreamSynchronize(handles[winner]->st));
    HIPCHK(hipSetDevice(devs[0]));
    HIPCHK(hipStreamSynchronize(handles[0]->st));
    handles[0]->state_dirty = true;
  }
  if (winner_out) *winner_out = winner;
  return 0;
}
