// Host-side mirror of the reference's user entry points (include/bfmmm_entry.h) over the sampler ABI
// (include/bfmmm.h): argument validation with the reference's messages, posterior-median warm
// starts, multi-try chain selection and result assembly.  No device code here.
#include "../../include/bfmmm.h"
#include "../../include/bfmmm_entry.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

struct ResArr {
  std::vector<double> data;
  std::vector<int64_t> dims;
};

struct bfmmm_result {
  std::vector<std::string> names;
  std::map<std::string, ResArr> arrs;
};

static thread_local std::string g_entry_err;
extern "C" const char* bfmmm_last_error(void);
// errors raised here are reported through the same channel as the sampler's: stash them in a
// result-independent thread-local and let bfmmm_entry_last_error expose them
extern "C" const char* bfmmm_entry_last_error(void) { return g_entry_err.c_str(); }
static int efail(const std::string& m) { g_entry_err = m; return 1; }
static int efail_lib() { g_entry_err = bfmmm_last_error(); return 1; }
int bfmmm_io_fail(const std::string& m) { return efail(m); }      // arma_io.cpp reports through the same channel
int arma_save_ascii(const std::string& path, const double* d, int64_t r, int64_t c, int64_t s, bool cube);
int arma_save_field_cubes(const std::string& path, const std::vector<std::vector<double>>& cubes, int64_t n_rows, int64_t n_cols,
                          int64_t r, int64_t c, int64_t s);

extern "C" bfmmm_result* bfmmm_result_create(void) { return new bfmmm_result(); }
extern "C" void bfmmm_result_free(bfmmm_result* r) { delete r; }

extern "C" int bfmmm_result_set(bfmmm_result* r, const char* name, const double* data, int64_t count, const int64_t* dims,
                                int n_dims) {
  if (!r || !name || (!data && count > 0)) return efail("bfmmm_result_set: null argument");
  const std::string s(name);
  if (!r->arrs.count(s)) r->names.push_back(s);
  ResArr& a = r->arrs[s];
  a.data.assign(data, data + count);
  a.dims.assign(dims, dims + n_dims);
  return 0;
}

extern "C" int bfmmm_result_get(const bfmmm_result* r, const char* name, const double** data, int64_t* count,
                                const int64_t** dims, int* n_dims) {
  if (!r || !name) return efail("bfmmm_result_get: null argument");
  auto it = r->arrs.find(name);
  if (it == r->arrs.end()) return efail(std::string("result has no element named '") + name + "'");
  if (data) *data = it->second.data.data();
  if (count) *count = (int64_t)it->second.data.size();
  if (dims) *dims = it->second.dims.data();
  if (n_dims) *n_dims = (int)it->second.dims.size();
  return 0;
}

extern "C" int bfmmm_result_count(const bfmmm_result* r) { return r ? (int)r->names.size() : 0; }
extern "C" const char* bfmmm_result_name(const bfmmm_result* r, int i) {
  return (r && i >= 0 && i < (int)r->names.size()) ? r->names[i].c_str() : nullptr;
}

static void put(bfmmm_result* r, const char* name, std::vector<double>&& v, std::vector<int64_t> dims) {
  const std::string s(name);
  if (!r->arrs.count(s)) r->names.push_back(s);
  ResArr& a = r->arrs[s];
  a.data = std::move(v);
  a.dims = std::move(dims);
}

extern "C" void bfmmm_entry_defaults(bfmmm_entry_args* a, int entry) {
  memset(a, 0, sizeof *a);
  a->n_try = 1;
  a->burnin_prop = 0.8;
  a->b = 10; a->nu_1 = 3;
  if (entry == 0) { a->alpha1l = 1; a->alpha2l = 2; a->beta1l = 1; a->beta2l = 1; }        // UserFunctions.cpp:179-182
  else if (entry == 1 || entry == 2) { a->alpha1l = 2; a->alpha2l = 3; a->beta1l = 2; a->beta2l = 2; }   // :699-702, :1363-1366
  else if (entry == 3) { a->alpha1l = 2; a->alpha2l = 3; a->beta1l = 1; a->beta2l = 1; }   // :4586-4589
  else { a->alpha1l = 1; a->alpha2l = 2; a->beta1l = 1; a->beta2l = 1; }                    // :5006-5009, :5556-5559
  a->model = (entry >= 3) ? BFMMM_MODEL_MULTIVARIATE : BFMMM_MODEL_FUNCTIONAL;
  a->a_Z_PM = 10000; a->a_pi_PM = 1000; a->var_alpha3 = 0.05; a->var_epsilon1 = 1; a->var_epsilon2 = 1;
  a->alpha_nu = 10; a->beta_nu = 1; a->alpha_eta = 10; a->beta_eta = 1; a->alpha_0 = 1; a->beta_0 = 1;
  a->thinning_num = 1; a->beta_N_t = 1; a->N_t = 1; a->n_temp_trans = 0; a->r_stored_iters = 0;
  a->seed = 1; a->device = 0; a->chain_offset = 0; a->chain_stride = 1; a->max_concurrent = 8;
}

// argument checks in the reference's order and wording (UserFunctions.cpp:198-286, 727-818, 1394-1498)
static int validate(const bfmmm_entry_args* a, int entry) {
  if (!a || !a->y) return efail("null argument");
  const bool mv = a->model == BFMMM_MODEL_MULTIVARIATE;
  if (!mv && (!a->t || !a->offsets || !a->boundary_knots)) return efail("null argument");
  if (mv && a->P < 1) return efail("'Y' must have at least one column");
  if (a->tot_mcmc_iters < 100) return efail("'tot_mcmc_iters' must be an integer greater than or equal to 100");
  if (entry != 2 && a->n_try < 1) return efail("'n_try' must be an integer greater than or equal to 1");
  if (entry != 0) {
    if (a->burnin_prop < 0) return efail("'burnin_prop' must be between 0 and 1");
    if (a->burnin_prop >= 1) return efail("'burnin_prop' must be between 0 and 1");
  }
  if (a->K < 2) return efail("'K' must be an integer greater than or equal to 2");
  if (a->n_funct < 1) return efail("'n_funct' must be an integer greater than or equal to 1");
  const bool hd = a->dim > 0;
  if (hd) {      // UserFunctions.cpp:2563-2588
    if (mv) return efail("the high-dimensional model is a functional model");
    if (!a->basis_degree_hd || !a->n_internal_hd) return efail("null argument");
    size_t ko = 0;
    for (int j = 0; j < a->dim; ++j) {
      if (a->basis_degree_hd[j] < 1) return efail("'basis_degree' elements must be an integer greater than or equal to 1");
      for (int i = 0; i < a->n_internal_hd[j]; ++i) {
        if (a->boundary_knots[2 * j] >= a->internal_knots[ko + i])
          return efail("at least one element in 'internal_knots' is less than or equal to first boundary knot");
        if (a->boundary_knots[2 * j + 1] <= a->internal_knots[ko + i])
          return efail("at least one element in 'internal_knots' is more than or equal to second boundary knot");
      }
      ko += (size_t)a->n_internal_hd[j];
    }
  }
  if (!mv && !hd && a->basis_degree < 1) return efail("'basis_degree' must be an integer greater than or equal to 1");
  if (a->n_eigen < 1) return efail("'n_eigen' must be an integer greater than or equal to 1");
  for (int i = 0; !mv && !hd && i < a->n_internal_knots; ++i) {
    if (a->boundary_knots[0] >= a->internal_knots[i])
      return efail("at least one element in 'internal_knots' is less than or equal to first boundary knot");
    if (a->boundary_knots[1] <= a->internal_knots[i])
      return efail("at least one element in 'internal_knots' is more than or equal to second boundary knot");
  }
  if (a->b <= 0) return efail("'b' must be positive");
  if (entry != 0 && a->nu_1 <= 0) return efail("'nu_1' must be positive");
  if (a->alpha1l <= 0) return efail("'alpha1l' must be positive");
  if (a->beta1l <= 0) return efail("'beta1l' must be positive");
  if (a->alpha2l <= 0) return efail("'alpha2l' must be positive");
  if (a->beta2l <= 0) return efail("'beta2l' must be positive");
  if (a->a_Z_PM <= 0) return efail("'a_Z_PM' must be positive");
  if (a->a_pi_PM <= 0) return efail("'a_pi_PM' must be positive");
  if (a->var_alpha3 <= 0) return efail("'var_alpha3' must be positive");
  if (a->var_epsilon1 <= 0) return efail("'var_epsilon1' must be positive");
  if (a->var_epsilon2 <= 0) return efail("'var_epsilon2' must be positive");
  if (a->alpha_nu <= 0) return efail("'alpha_nu' must be positive");
  if (a->beta_nu <= 0) return efail("'beta_nu' must be positive");
  if (a->alpha_0 <= 0) return efail("'alpha_0' must be positive");
  if (a->beta_0 <= 0) return efail("'beta_0' must be positive");
  if (a->c)
    for (int i = 0; i < a->K; ++i)
      if (a->c[i] <= 0) return efail("all elements of 'c' must be positive");
  if (a->X && (a->D < 1 || a->D > 8)) return efail("the number of covariates (columns of 'X') must be between 1 and 8 in this build");
  if (a->X && a->alpha_eta <= 0) return efail("'alpha_eta' must be positive");
  if (a->X && a->beta_eta <= 0) return efail("'beta_eta' must be positive");
  if (a->K > 8) return efail("K larger than 8 is not supported by this build");
  if (a->chain_stride < 1 || a->chain_offset < 0) return efail("invalid chain_offset / chain_stride");
  return 0;
}

static void make_cfg(const bfmmm_entry_args* a, int T, bfmmm_config* cfg) {
  bfmmm_config_defaults(cfg);
  cfg->model = a->model;
  cfg->P = a->P;
  cfg->n_funct = a->n_funct; cfg->K = a->K; cfg->n_eigen = a->n_eigen;
  cfg->basis_degree = a->basis_degree; cfg->n_internal_knots = a->n_internal_knots;
  cfg->tot_mcmc_iters = T;
  for (int k = 0; k < a->K && k < 8; ++k) cfg->c[k] = a->c ? a->c[k] : 10.0;
  cfg->b = a->b; cfg->nu_1 = a->nu_1;
  cfg->alpha1l = a->alpha1l; cfg->alpha2l = a->alpha2l; cfg->beta1l = a->beta1l; cfg->beta2l = a->beta2l;
  cfg->a_Z_PM = a->a_Z_PM; cfg->a_pi_PM = a->a_pi_PM; cfg->var_alpha3 = a->var_alpha3;
  cfg->var_epsilon1 = a->var_epsilon1; cfg->var_epsilon2 = a->var_epsilon2;
  cfg->alpha_nu = a->alpha_nu; cfg->beta_nu = a->beta_nu; cfg->alpha_eta = a->alpha_eta; cfg->beta_eta = a->beta_eta;
  cfg->alpha_0 = a->alpha_0; cfg->beta_0 = a->beta_0;
}

// arma::median
static double median_of(std::vector<double>& v) {
  const size_t n = v.size();
  if (n == 0) return NAN;
  const size_t h = n / 2;
  std::nth_element(v.begin(), v.begin() + h, v.end());
  const double hi = v[h];
  if (n & 1) return hi;
  const double lo = *std::max_element(v.begin(), v.begin() + h);
  return 0.5 * (lo + hi);
}

// median over the trailing slices [burn, T) of element e of a (len x T) slot-major chain
static double tail_median(const double* chain, int64_t len, int64_t e, int64_t burn, int64_t T, std::vector<double>& buf) {
  buf.clear();
  for (int64_t l = burn; l < T; ++l) buf.push_back(chain[e + len * l]);
  return median_of(buf);
}

struct ChainRun {
  bfmmm_handle* h = nullptr;      // batch handle with the best chain selected
  double score = -INFINITY;
  int chain = -1;
};

// mean of the last 99 log-likelihood values, UserFunctions.cpp:309,320
static double tail_score(const std::vector<double>& ll, int T) {
  double s = 0.0;
  for (int i = T - 99; i <= T - 1; ++i) s += ll[i];
  return s / 99.0;
}

typedef int (*chain_setup_fn)(bfmmm_handle* h, int chain, const void* ctx);
static int make_handle(const bfmmm_entry_args* a, const bfmmm_config* cfg, int device, int n_chains, bfmmm_handle** h);
static int attach_cov(bfmmm_handle* h, const bfmmm_entry_args* a);

// the chains one device runs, and what it found
struct DevRun {
  int device = 0;
  std::vector<int> chains;        // ascending, equally spaced
  ChainRun best;
  int n_ok = 0;
  std::string err;
};

// Runs the device's chains in batches of at most max_concurrent (one batch = ONE sampler whose kernels carry the chain
// index as a grid dimension, bfmmm_create_batch) and keeps the batch that holds the device's best chain.
// Memory: a batch is one arena of n_chains x (state + workspace + all T chain slots) -- about 0.3 MB x T per chain at
// n_funct = 4096, K = 3, P = 30, M = 6 -- so when the arena cannot be allocated the batch size is halved and the same
// chains are retried in smaller batches.  Any other failure (a precision that is not positive definite, a failing
// kernel, a failing read-back) fails the whole call, as the reference does (an exception inside one of its chains
// aborts BFMMM_Nu_Z_multiple_try): r.err is set, nothing is kept, and run_multi_try returns the error.
static void run_device(const bfmmm_entry_args* a, const bfmmm_config& cfg, uint32_t mask, int phi_chi_zero,
                       chain_setup_fn setup, const void* ctx, DevRun& r) {
  const int T = cfg.tot_mcmc_iters;
  size_t cap = (size_t)std::max(1, a->max_concurrent);
  const uint32_t id_stride = r.chains.size() > 1 ? (uint32_t)(r.chains[1] - r.chains[0]) : 1u;
  std::vector<double> ll(T);
  auto fail_device = [&](bfmmm_handle* h) {
    if (r.err.empty()) r.err = bfmmm_last_error();
    if (r.err.empty()) r.err = "multi-try batch failed";
    if (h) bfmmm_destroy(h);
    if (r.best.h) bfmmm_destroy(r.best.h);
    r.best = ChainRun();
    r.n_ok = 0;
  };
  size_t base = 0;
  while (base < r.chains.size()) {
    const int nb = (int)std::min(r.chains.size() - base, cap);
    bfmmm_handle* h = nullptr;
    if (make_handle(a, &cfg, r.device, nb, &h)) {
      if (h) { bfmmm_destroy(h); h = nullptr; }
      if (nb > 1 && strstr(bfmmm_last_error(), "memory")) { cap = (size_t)(nb + 1) / 2; continue; }      // smaller batches
      return fail_device(nullptr);
    }
    if (attach_cov(h, a)) {      // (the covariate arena scales with the batch too: an out-of-memory here halves the batch as well)
      if (nb > 1 && strstr(bfmmm_last_error(), "memory")) { bfmmm_destroy(h); cap = (size_t)(nb + 1) / 2; continue; }
      return fail_device(h);
    }
    if (bfmmm_set_chain_id_stride(h, id_stride)) return fail_device(h);
    for (int q = 0; q < nb; ++q)
      if (bfmmm_select_chain(h, q) || setup(h, r.chains[base + q], ctx)) return fail_device(h);
    if (bfmmm_run(h, mask, 0, T, a->seed, (uint32_t)r.chains[base], phi_chi_zero, 1.0)) return fail_device(h);
    // score the batch into locals; r.best changes only after every chain of the batch has been read back
    int sel = -1, n_fin = 0;
    double sel_score = -INFINITY;
    for (int q = 0; q < nb; ++q) {
      if (bfmmm_select_chain(h, q) || bfmmm_get_chain(h, "loglik", T, ll.data(), T)) return fail_device(h);
      const double sc = tail_score(ll, T);
      if (!(sc == sc)) continue;                          // a chain whose log-likelihood is not a number does not compete
      n_fin += 1;
      // strictly larger score wins, the earlier chain on ties (the reference's `<` at UserFunctions.cpp:320)
      if (sel < 0 || sel_score < sc) { sel_score = sc; sel = q; }
    }
    r.n_ok += n_fin;
    if (sel >= 0 && (r.best.chain < 0 || r.best.score < sel_score)) {
      if (bfmmm_select_chain(h, sel)) return fail_device(h);
      if (r.best.h) bfmmm_destroy(r.best.h);
      r.best.h = h;
      r.best.score = sel_score;
      r.best.chain = r.chains[base + sel];
    } else {
      bfmmm_destroy(h);
    }
    base += (size_t)nb;
  }
}

// Runs the chains {offset, offset + stride, ...} <= n_try and returns the sampler holding the best one (selected).
// The chains are dealt round-robin over the devices of a->devices (default: the single a->device), one host thread per
// device; with more than one device the final selection is the RCCL gather of bfmmm_gather_best.
static int run_multi_try(const bfmmm_entry_args* a, const bfmmm_config& cfg, uint32_t mask, int phi_chi_zero,
                         chain_setup_fn setup, const void* ctx, ChainRun* best) {
  std::vector<int> chains;
  for (int c = a->chain_offset; c <= a->n_try; c += a->chain_stride) chains.push_back(c);
  if (chains.empty()) return efail("no chain index falls in this process's chain_offset / chain_stride range");
  std::vector<int> devs;
  if (a->n_devices > 0 && a->devices) devs.assign(a->devices, a->devices + a->n_devices);
  else devs.push_back(a->device);
  const bool use_rccl = a->n_devices > 0 && a->devices;       // an explicit device list asks for the RCCL selection
  for (size_t i = 0; i < devs.size(); ++i) {                   // checked before any chain runs
    if (devs[i] < 0) return efail("device indices must not be negative");
    for (size_t j = 0; j < i; ++j)
      if (devs[j] == devs[i]) return efail("'devices' names a device twice: one chain batch per device");
  }
  const size_t G = std::min(devs.size(), chains.size());
  std::vector<DevRun> runs(G);
  for (size_t g = 0; g < G; ++g) {
    runs[g].device = devs[g];
    for (size_t i = g; i < chains.size(); i += G) runs[g].chains.push_back(chains[i]);
  }
  if (G == 1) {
    run_device(a, cfg, mask, phi_chi_zero, setup, ctx, runs[0]);
  } else {
    std::vector<std::thread> th;
    for (size_t g = 0; g < G; ++g) th.emplace_back([&, g]() { run_device(a, cfg, mask, phi_chi_zero, setup, ctx, runs[g]); });
    for (auto& t : th) t.join();
  }
  auto cleanup = [&]() { for (DevRun& r : runs) if (r.best.h) { bfmmm_destroy(r.best.h); r.best.h = nullptr; } };
  int n_ok = 0;
  for (DevRun& r : runs) {
    if (!r.err.empty()) { const std::string e = r.err; cleanup(); return efail(e); }      // one failing batch fails the call
    n_ok += r.n_ok;
  }
  if (n_ok == 0) { cleanup(); return efail("no chain produced a finite log-likelihood"); }
  size_t w = 0;
  if (use_rccl) {
    // ranks that hold a sampler take part; the winner's chain ends up in the first of them
    std::vector<bfmmm_handle*> hs;
    std::vector<double> sc;
    std::vector<int32_t> ids;
    std::vector<size_t> idx;
    for (size_t g = 0; g < G; ++g)
      if (runs[g].best.h) { hs.push_back(runs[g].best.h); sc.push_back(runs[g].best.score); ids.push_back(runs[g].best.chain); idx.push_back(g); }
    int win = 0;
    if (bfmmm_gather_best(hs.data(), (int)hs.size(), sc.data(), ids.data(), &win)) { efail_lib(); cleanup(); return 1; }
    w = idx[0];
    runs[w].best.score = sc[win];
    runs[w].best.chain = ids[win];
  } else {
    bool have = false;
    for (size_t g = 0; g < G; ++g) {
      if (!runs[g].best.h) continue;
      const ChainRun& b = runs[g].best;
      if (!have || b.score > runs[w].best.score || (b.score == runs[w].best.score && b.chain < runs[w].best.chain)) { w = g; have = true; }
    }
  }
  *best = runs[w].best;
  runs[w].best.h = nullptr;
  cleanup();
  return 0;
}

static int fetch(bfmmm_handle* h, bfmmm_result* r, const char* chain_name, const char* out_name, int T, int64_t len,
                 std::vector<int64_t> dims, int extra_slot = 0) {
  std::vector<double> v((size_t)len * (T + extra_slot), 0.0);
  if (bfmmm_get_chain(h, chain_name, T, v.data(), (int64_t)len * T)) return efail_lib();
  if (extra_slot && std::string(chain_name) != "loglik")      // slot T carries slot T-1 (BFMMM.h:71-73 style carry)
    std::copy(v.begin() + (size_t)len * (T - 1), v.begin() + (size_t)len * T, v.begin() + (size_t)len * T);
  put(r, out_name, std::move(v), std::move(dims));
  return 0;
}

static int fetch_tau(bfmmm_handle* h, bfmmm_result* r, int T, int K, int extra_slot = 0) {
  std::vector<double> v((size_t)T * K), out((size_t)(T + extra_slot) * K, 0.0);
  if (bfmmm_get_chain(h, "tau", T, v.data(), (int64_t)T * K)) return efail_lib();
  const int TT = T + extra_slot;
  for (int k = 0; k < K; ++k) {
    for (int t = 0; t < T; ++t) out[t + (size_t)TT * k] = v[t + (size_t)T * k];
    if (extra_slot) out[T + (size_t)TT * k] = v[(T - 1) + (size_t)T * k];
  }
  put(r, "tau", std::move(out), {TT, K});
  return 0;
}

static int dimP(const bfmmm_entry_args* a) {
  if (a->dim > 0) {
    int P = 1;
    for (int j = 0; j < a->dim; ++j) P *= a->n_internal_hd[j] + a->basis_degree_hd[j] + 1;
    return P;
  }
  return (a->model == BFMMM_MODEL_MULTIVARIATE) ? a->P : a->n_internal_knots + a->basis_degree + 1;
}

// sampler handle of an entry point: univariate B-splines / multivariate data (bfmmm_create) or, for the high-dimensional
// model, the tensor-product basis and penalty of BSplines.h:18-120 built on the host (BFMMM.h:2923-2936 does the same
// per call) and handed to bfmmm_create_from_basis
static int make_handle(const bfmmm_entry_args* a, const bfmmm_config* cfg, int device, int n_chains, bfmmm_handle** h) {
  if (a->dim <= 0) return bfmmm_create_batch(cfg, device, a->y, a->t, a->offsets, a->internal_knots, a->boundary_knots, n_chains, h);
  const int dim = a->dim, P = dimP(a);
  const int64_t n_obs = a->offsets[a->n_funct];
  std::vector<double> B((size_t)n_obs * P), tmp;
  for (int i = 0; i < a->n_funct; ++i) {
    const int64_t o = a->offsets[i], ni = a->offsets[i + 1] - o;
    tmp.assign((size_t)ni * P, 0.0);
    if (bfmmm_tensor_bspline((int)ni, dim, a->t + (size_t)o * dim, a->basis_degree_hd, a->boundary_knots, a->n_internal_hd,
                             a->internal_knots, tmp.data()))
      return 1;
    for (int64_t l = 0; l < ni; ++l)
      for (int p = 0; p < P; ++p) B[(size_t)(o + l) * P + p] = tmp[(size_t)l + (size_t)ni * p];      // column-major -> rows
  }
  std::vector<double> Pm((size_t)P * P);
  if (bfmmm_tensor_penalty(dim, a->basis_degree_hd, a->n_internal_hd, Pm.data())) return 1;
  // band of B'B: basis functions whose multi-indices differ by more than the degree in some dimension never overlap
  int band = 0, pen_band = 1, stride = 1;
  for (int j = dim - 1; j >= 0; --j) {
    band += a->basis_degree_hd[j] * stride;
    pen_band = stride;
    stride *= a->n_internal_hd[j] + a->basis_degree_hd[j] + 1;
  }
  return bfmmm_create_from_basis_batch(cfg, device, a->y, B.data(), a->offsets, P, std::min(band, P - 1), Pm.data(),
                                       std::min(pen_band, P - 1), n_chains, h);
}

static int fetch_basis(bfmmm_handle* h, bfmmm_result* r, const bfmmm_entry_args* a, const char* name) {
  if (a->model == BFMMM_MODEL_MULTIVARIATE) return 0;      // no basis in the multivariate model
  const int P = dimP(a);
  const int64_t n_obs = a->offsets[a->n_funct];
  std::vector<double> B((size_t)n_obs * P);
  if (bfmmm_get_basis(h, B.data(), (int64_t)B.size())) return efail_lib();
  put(r, name, std::move(B), {n_obs, P});
  return 0;
}

static int attach_cov(bfmmm_handle* h, const bfmmm_entry_args* a) {      // every chain of the sampler
  if (!a->X) return 0;
  return bfmmm_set_covariates(h, a->X, a->D, a->covariance_adj);
}

// chain arrays of the covariate blocks, in the reference's shapes
static int fetch_cov(bfmmm_handle* h, bfmmm_result* r, const bfmmm_entry_args* a, int T, bool with_xi, int extra_slot = 0) {
  if (!a->X) return 0;
  const int64_t K = a->K, M = a->n_eigen, D = a->D, P = dimP(a);
  const int TT = T + extra_slot;
  if (fetch(h, r, "eta", "eta", T, P * D * K, {P, D, K, TT}, extra_slot) ||
      fetch(h, r, "tau_eta", "tau_eta", T, K * D, {K, D, TT}, extra_slot))
    return 1;
  if (!with_xi) return 0;
  return fetch(h, r, "xi", "xi", T, K * P * D * M, {P, D, M, K, TT}, extra_slot) ||
         fetch(h, r, "gamma_xi", "gamma_xi", T, K * P * D * M, {P, D, M, K, TT}, extra_slot) ||
         fetch(h, r, "delta_xi", "delta_xi", T, K * M * D, {K, M, D, TT}, extra_slot) ||
         fetch(h, r, "A_xi", "A_xi", T, K * 2 * D, {K, 2, D, TT}, extra_slot);
}

// ------------------------------------------------------------------------------------------------
static int setup_nu_z(bfmmm_handle* h, int chain, const void* ctx) {
  const bfmmm_entry_args* a = (const bfmmm_entry_args*)ctx;       // (covariates attached per sampler: eta = 0, tau_eta = 1, BFMMM.h:3705-3722)
  return bfmmm_init_state(h, 0, a->seed, (uint32_t)chain);      // BFMMM.h:1039-1071
}

extern "C" int bfmmm_BFMMM_Nu_Z_multiple_try(const bfmmm_entry_args* a, bfmmm_result** out) {
  if (!out) return efail("null argument");
  if (validate(a, 0)) return 1;
  const int T = a->tot_mcmc_iters, K = a->K, M = a->n_eigen, n = a->n_funct;
  const int P = dimP(a);
  bfmmm_config cfg;
  make_cfg(a, T, &cfg);
  ChainRun best;
  const uint32_t mask_nz = BFMMM_SWEEP_NU_Z | (a->X ? BFMMM_COV_MEAN : 0);      // BFMMM.h:3741-3780
  if (run_multi_try(a, cfg, mask_nz, 1, setup_nu_z, a, &best)) return 1;
  bfmmm_result* r = bfmmm_result_create();
  int rc = fetch_basis(best.h, r, a, "B") || fetch(best.h, r, "nu", "nu", T, (int64_t)K * P, {K, P, T}) ||
           fetch(best.h, r, "pi", "pi", T, K, {K, T}) || fetch(best.h, r, "alpha_3", "alpha_3", T, 1, {T}) ||
           fetch(best.h, r, "A", "A", T, (int64_t)K * 2, {K, 2, T}) ||
           fetch(best.h, r, "delta", "delta", T, (int64_t)K * M, {K, M, T}) ||
           fetch(best.h, r, "sigma_sq", "sigma_sq", T, 1, {T}) || fetch_tau(best.h, r, T, K) ||
           fetch(best.h, r, "Z", "Z", T, (int64_t)n * K, {n, K, T}) || fetch(best.h, r, "loglik", "loglik", T, 1, {T}) ||
           fetch_cov(best.h, r, a, T, false);
  bfmmm_destroy(best.h);
  if (rc) { bfmmm_result_free(r); return 1; }
  put(r, "best_chain", {(double)best.chain}, {1});
  put(r, "best_score", {best.score}, {1});
  *out = r;
  return 0;
}

// ------------------------------------------------------------------------------------------------
struct ThetaCtx {
  const bfmmm_entry_args* a;
  std::vector<double> Z_est, nu_est, eta_est;
};

static int setup_theta(bfmmm_handle* h, int chain, const void* ctx) {
  const ThetaCtx* t = (const ThetaCtx*)ctx;
  if (bfmmm_init_state(h, 1, t->a->seed, (uint32_t)chain)) return 1;             // BFMMM.h:1210-1235
  if (t->a->X && bfmmm_set_state(h, "eta", t->eta_est.data(), (int64_t)t->eta_est.size())) return 1;   // BFMMM.h:3936-3942
  if (bfmmm_set_state(h, "Z", t->Z_est.data(), (int64_t)t->Z_est.size())) return 1;   // BFMMM.h:1244-1250
  return bfmmm_set_state(h, "nu", t->nu_est.data(), (int64_t)t->nu_est.size());
}

// posterior medians of Z and nu over the trailing (1 - burnin_prop) of stage 1, rows of Z
// renormalised (UserFunctions.cpp:833-858)
static int median_Z_nu(const bfmmm_entry_args* a, const bfmmm_result* mt, std::vector<double>& Z_est,
                       std::vector<double>& nu_est, int64_t* n_nu_out) {
  const double *Zs, *nus;
  int64_t cz, cn;
  const int64_t* dz; const int64_t* dn;
  int ndz, ndn;
  if (bfmmm_result_get(mt, "Z", &Zs, &cz, &dz, &ndz) || bfmmm_result_get(mt, "nu", &nus, &cn, &dn, &ndn)) return 1;
  const int n = a->n_funct, K = a->K, P = dimP(a);
  if (cz % ((int64_t)n * K) != 0 || cn % ((int64_t)K * P) != 0) return efail("'multiple_try' arrays have the wrong shape");
  const int64_t n_nu = cn / ((int64_t)K * P);
  if (cz / ((int64_t)n * K) != n_nu) return efail("'multiple_try' arrays have the wrong shape");
  const int64_t burn = (int64_t)std::round(n_nu * a->burnin_prop);
  std::vector<double> buf;
  Z_est.assign((size_t)n * K, 0.0);
  nu_est.assign((size_t)K * P, 0.0);
  for (int64_t e = 0; e < (int64_t)n * K; ++e) Z_est[e] = tail_median(Zs, (int64_t)n * K, e, burn, n_nu, buf);
  for (int64_t e = 0; e < (int64_t)K * P; ++e) nu_est[e] = tail_median(nus, (int64_t)K * P, e, burn, n_nu, buf);
  for (int i = 0; i < n; ++i) {
    double s = 0.0;
    for (int k = 0; k < K; ++k) s += Z_est[i + (size_t)n * k];
    for (int k = 0; k < K; ++k) Z_est[i + (size_t)n * k] /= s;
  }
  if (n_nu_out) *n_nu_out = n_nu;
  return 0;
}

extern "C" int bfmmm_BFMMM_Theta_est(const bfmmm_entry_args* a, const bfmmm_result* multiple_try, bfmmm_result** out) {
  if (!out || !multiple_try) return efail("null argument");
  if (validate(a, 1)) return 1;
  const int T = a->tot_mcmc_iters, K = a->K, M = a->n_eigen, n = a->n_funct;
  const int P = dimP(a);
  ThetaCtx tc;
  tc.a = a;
  int64_t n_nu = 0;
  if (median_Z_nu(a, multiple_try, tc.Z_est, tc.nu_est, &n_nu)) return 1;
  if (a->X) {   // eta_est: median over the same trailing slices (UserFunctions.cpp:1049-1054)
    const double* es; int64_t ce;
    if (bfmmm_result_get(multiple_try, "eta", &es, &ce, nullptr, nullptr)) return 1;
    const int64_t len = (int64_t)P * a->D * K;
    if (ce != len * n_nu) return efail("'multiple_try$eta' has the wrong shape");
    const int64_t burn = (int64_t)std::round(n_nu * a->burnin_prop);
    std::vector<double> buf;
    tc.eta_est.resize((size_t)len);
    for (int64_t e = 0; e < len; ++e) tc.eta_est[e] = tail_median(es, len, e, burn, n_nu, buf);
  }
  bfmmm_config cfg;
  make_cfg(a, T, &cfg);
  ChainRun best;
  const uint32_t mask_th = BFMMM_SWEEP_THETA | (a->X ? (BFMMM_U_TAU_ETA | (a->covariance_adj ? BFMMM_COV_XI : 0)) : 0);   // BFMMM.h:3944-4010
  if (run_multi_try(a, cfg, mask_th, 0, setup_theta, &tc, &best)) return 1;
  bfmmm_result* r = bfmmm_result_create();
  int rc = fetch_basis(best.h, r, a, "B") || fetch(best.h, r, "Z", "Z", T, (int64_t)n * K, {n, K, T}) ||
           fetch(best.h, r, "nu", "nu", T, (int64_t)K * P, {K, P, T}) ||
           fetch(best.h, r, "chi", "chi", T, (int64_t)n * M, {n, M, T}) ||
           fetch(best.h, r, "A", "A", T, (int64_t)K * 2, {K, 2, T}) ||
           fetch(best.h, r, "delta", "delta", T, (int64_t)K * M, {K, M, T}) ||
           fetch(best.h, r, "sigma_sq", "sigma_sq", T, 1, {T}) || fetch_tau(best.h, r, T, K) ||
           fetch(best.h, r, "gamma", "gamma", T, (int64_t)K * P * M, {K, P, M, T}) ||
           fetch(best.h, r, "Phi", "Phi", T, (int64_t)K * P * M, {K, P, M, T}) ||
           fetch(best.h, r, "loglik", "loglik", T, 1, {T}) || fetch_cov(best.h, r, a, T, true);
  bfmmm_destroy(best.h);
  if (rc) { bfmmm_result_free(r); return 1; }
  put(r, "best_chain", {(double)best.chain}, {1});
  put(r, "best_score", {best.score}, {1});
  *out = r;
  return 0;
}

// ---- one on-disk batch of BFMMM_MTT_warm_start (BFMMM.h:1680-1730; covariate-adjusted drivers :4455-4545, :5086-5165) ----
// The batch's r_stored_iters slots are thinned to cnt = r_stored_iters / thinning_num draws: draw 0 is slot 0, draw p > 0
// is slot thinning_num * p - 1.  Two quirks of the reference are kept: alpha_31(0) is never assigned (the file starts
// with 0), and the covariate blocks are saved in containers of r_stored_iters (not cnt) entries whose tail stays
// unassigned (empty cubes in the fields, ones in Tau_Eta).
static int save_batch(bfmmm_handle* h, const bfmmm_entry_args* a, const std::string& dir, int q, int rs) {
  const int64_t K = a->K, M = a->n_eigen, n = a->n_funct, P = dimP(a), D = a->D;
  const int cnt = (int)(rs / a->thinning_num);
  auto slot_of = [&](int p) { return p == 0 ? 0 : (int)(a->thinning_num * p - 1); };
  auto get = [&](const char* name, int64_t len, std::vector<double>& v) {
    v.assign((size_t)len * rs, 0.0);
    return bfmmm_get_chain(h, name, rs, v.data(), (int64_t)v.size()) ? efail_lib() : 0;
  };
  auto thin = [&](const std::vector<double>& v, int64_t len) {
    std::vector<double> o((size_t)len * cnt);
    for (int p = 0; p < cnt; ++p) std::copy(v.begin() + len * slot_of(p), v.begin() + len * (slot_of(p) + 1), o.begin() + len * p);
    return o;
  };
  const std::string sq = std::to_string(q) + ".txt";
  std::vector<double> v;
  auto cube_txt = [&](const char* chain, const char* file, int64_t r, int64_t c) {
    if (get(chain, r * c, v)) return 1;
    return arma_save_ascii(dir + file + sq, thin(v, r * c).data(), r, c, cnt, true);
  };
  auto field_bin = [&](const char* chain, const char* file, int64_t r, int64_t c, int64_t s3, int64_t cols, int rows) {
    // chain slot = `cols` consecutive cubes r x c x s3; field(rows, cols), object (p, k) at p + rows * k
    if (get(chain, r * c * s3 * cols, v)) return 1;
    std::vector<std::vector<double>> cubes((size_t)rows * cols);
    const int64_t len = r * c * s3;
    for (int p = 0; p < cnt; ++p)
      for (int64_t k = 0; k < cols; ++k) {
        const double* src = v.data() + (size_t)(len * cols) * slot_of(p) + len * k;
        cubes[(size_t)p + (size_t)rows * k].assign(src, src + len);
      }
    return arma_save_field_cubes(dir + file + sq, cubes, rows, cols, r, c, s3);
  };
  if (cube_txt("nu", "Nu", K, P) || cube_txt("chi", "Chi", n, M)) return 1;
  if (get("pi", K, v) || arma_save_ascii(dir + "Pi" + sq, thin(v, K).data(), K, cnt, 1, false)) return 1;
  if (get("alpha_3", 1, v)) return 1;
  { std::vector<double> t = thin(v, 1); if (cnt > 0) t[0] = 0.0; if (arma_save_ascii(dir + "alpha_3" + sq, t.data(), cnt, 1, 1, false)) return 1; }
  if (cube_txt("A", "A", K, 2) || cube_txt("delta", "Delta", K, M)) return 1;
  if (get("sigma_sq", 1, v) || arma_save_ascii(dir + "Sigma" + sq, thin(v, 1).data(), cnt, 1, 1, false)) return 1;
  {
    if (get("tau", K, v)) return 1;           // rs x K column-major
    std::vector<double> t((size_t)cnt * K);
    for (int64_t k = 0; k < K; ++k)
      for (int p = 0; p < cnt; ++p) t[(size_t)p + (size_t)cnt * k] = v[(size_t)slot_of(p) + (size_t)rs * k];
    if (arma_save_ascii(dir + "Tau" + sq, t.data(), cnt, K, 1, false)) return 1;
  }
  if (field_bin("gamma", "Gamma", K, P, M, 1, cnt) || field_bin("Phi", "Phi", K, P, M, 1, cnt)) return 1;
  if (cube_txt("Z", "Z", n, K)) return 1;
  if (a->X) {
    if (a->covariance_adj) {
      if (field_bin("xi", "Xi", P, D, M, K, rs) || field_bin("gamma_xi", "Gamma_Xi", P, D, M, K, rs) ||
          field_bin("delta_xi", "Delta_Xi", K, M, D, 1, rs) || field_bin("A_xi", "A_Xi", K, 2, D, 1, rs))
        return 1;
    }
    if (field_bin("eta", "Eta", P, D, K, 1, rs)) return 1;
    if (get("tau_eta", K * D, v)) return 1;
    std::vector<double> t((size_t)K * D * rs, 1.0);
    for (int p = 0; p < cnt; ++p) std::copy(v.begin() + K * D * slot_of(p), v.begin() + K * D * (slot_of(p) + 1), t.begin() + K * D * p);
    if (arma_save_ascii(dir + "Tau_Eta" + sq, t.data(), K, D, rs, true)) return 1;
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------
extern "C" int bfmmm_BFMMM_warm_start(const bfmmm_entry_args* a, const bfmmm_result* mt, const bfmmm_result* te,
                                      bfmmm_result** out) {
  if (!out || !mt || !te) return efail("null argument");
  if (validate(a, 2)) return 1;
  // UserFunctions.cpp:1475-1489
  if (a->beta_N_t <= 0 || a->beta_N_t > 1) return efail("'beta_N_t' must be between 0 and 1");
  if (a->N_t < 1) return efail("'N_t' must be a positive integer");
  if (a->n_temp_trans < 0) return efail("'n_temp_trans' must be a non-negative integer");
  if (a->thinning_num <= 0) return efail("'thinning_num' must be a positive integer");            // UserFunctions.cpp:1472-1474
  if (a->r_stored_iters < 0) return efail("'r_stored_iters' must be a non-negative integer");      // :1484-1486
  const int T = a->tot_mcmc_iters, K = a->K, M = a->n_eigen, n = a->n_funct;
  const int P = dimP(a);
  // UserFunctions.cpp:1508-1541: r_stored_iters = 0 keeps everything in memory (tot_mcmc_iters + 1 slots)
  const bool have_dir = a->dir && a->dir[0];
  int rs = a->r_stored_iters == 0 ? T + 1 : a->r_stored_iters;
  if (!have_dir && rs <= T)
    return efail("'r_stored_iters' <= 'tot_mcmc_iters' with no 'dir' specified. Either specify 'dir' or increase 'r_stored_iters'");
  if (rs > T + 1) rs = T + 1;
  const bool batched = rs <= T;          // otherwise (i + 1) % r_stored_iters is never 0 inside the loop: nothing is saved
  if (batched && a->thinning_num != std::floor(a->thinning_num)) return efail("'thinning_num' must be a positive integer");
  // ---- posterior medians of every block (UserFunctions.cpp:1557-1647) ----
  std::vector<double> Z_est, nu_est, buf;
  int64_t n_nu = 0;
  if (median_Z_nu(a, mt, Z_est, nu_est, &n_nu)) return 1;
  auto get = [&](const bfmmm_result* r, const char* name, const double** p, int64_t want_per_slot, int64_t* slots) {
    int64_t cnt;
    if (bfmmm_result_get(r, name, p, &cnt, nullptr, nullptr)) return 1;
    if (want_per_slot <= 0 || cnt % want_per_slot != 0) return efail(std::string("'") + name + "' has the wrong shape");
    *slots = cnt / want_per_slot;
    return 0;
  };
  const double *pis, *a3s, *taus, *deltas, *gammas, *Phis, *As, *sigmas, *chis;
  int64_t s_pi, s_a3, s_tau, s_delta, s_gamma, s_Phi, s_A, s_sigma, s_chi;
  if (get(mt, "pi", &pis, K, &s_pi) || get(mt, "alpha_3", &a3s, 1, &s_a3) || get(mt, "tau", &taus, K, &s_tau) ||
      get(te, "delta", &deltas, (int64_t)K * M, &s_delta) || get(te, "gamma", &gammas, (int64_t)K * P * M, &s_gamma) ||
      get(te, "Phi", &Phis, (int64_t)K * P * M, &s_Phi) || get(te, "A", &As, (int64_t)K * 2, &s_A) ||
      get(te, "sigma_sq", &sigmas, 1, &s_sigma) || get(te, "chi", &chis, (int64_t)n * M, &s_chi))
    return 1;
  if (s_a3 != n_nu || s_pi != n_nu || s_tau != n_nu) return efail("'multiple_try' arrays have inconsistent lengths");
  const int64_t n_Phi = s_sigma;
  if (s_delta != n_Phi || s_gamma != n_Phi || s_Phi != n_Phi || s_A != n_Phi || s_chi != n_Phi)
    return efail("'theta_est' arrays have inconsistent lengths");
  const int64_t burn_nu = (int64_t)std::round(n_nu * a->burnin_prop);
  const int64_t burn_Phi = (int64_t)std::round(n_Phi * a->burnin_prop);
  const double alpha_3_est = tail_median(a3s, 1, 0, burn_nu, n_nu, buf);
  std::vector<double> pi_est(K), tau_est(K);
  double pis_sum = 0.0;
  for (int k = 0; k < K; ++k) { pi_est[k] = tail_median(pis, K, k, burn_nu, n_nu, buf); pis_sum += pi_est[k]; }
  for (int k = 0; k < K; ++k) pi_est[k] /= pis_sum;
  for (int k = 0; k < K; ++k) {   // tau is T x K column-major: (l, k) at l + n_nu * k
    buf.clear();
    for (int64_t l = burn_nu; l < n_nu; ++l) buf.push_back(taus[l + n_nu * k]);
    tau_est[k] = median_of(buf);
  }
  const double sigma_est = tail_median(sigmas, 1, 0, burn_Phi, n_Phi, buf);
  auto med_block = [&](const double* chain, int64_t len) {
    std::vector<double> v((size_t)len);
    for (int64_t e = 0; e < len; ++e) v[e] = tail_median(chain, len, e, burn_Phi, n_Phi, buf);
    return v;
  };
  std::vector<double> delta_est = med_block(deltas, (int64_t)K * M), gamma_est = med_block(gammas, (int64_t)K * P * M),
                      Phi_est = med_block(Phis, (int64_t)K * P * M), A_est = med_block(As, (int64_t)K * 2),
                      chi_est = med_block(chis, (int64_t)n * M);
  // covariate blocks: eta, tau_eta from stage 1; xi, gamma_xi, delta_xi, A_xi from stage 2 (UserFunctions.cpp:1893-1899, 1951-1960)
  std::vector<double> eta_est, tau_eta_est, xi_est, gamma_xi_est, delta_xi_est, A_xi_est;
  if (a->X) {
    const int64_t D = a->D;
    const double *es, *tes;
    int64_t s_e, s_te;
    if (get(mt, "eta", &es, (int64_t)P * D * K, &s_e) || get(mt, "tau_eta", &tes, (int64_t)K * D, &s_te)) return 1;
    if (s_e != n_nu || s_te != n_nu) return efail("'multiple_try' arrays have inconsistent lengths");
    auto med_nu = [&](const double* chain, int64_t len) {
      std::vector<double> v((size_t)len);
      for (int64_t e = 0; e < len; ++e) v[e] = tail_median(chain, len, e, burn_nu, n_nu, buf);
      return v;
    };
    eta_est = med_nu(es, (int64_t)P * D * K);
    tau_eta_est = med_nu(tes, (int64_t)K * D);
    if (a->covariance_adj) {
      const double *xs, *gxs, *dxs, *axs;
      int64_t s1, s2, s3, s4;
      if (get(te, "xi", &xs, (int64_t)K * P * D * M, &s1) || get(te, "gamma_xi", &gxs, (int64_t)K * P * D * M, &s2) ||
          get(te, "delta_xi", &dxs, (int64_t)K * M * D, &s3) || get(te, "A_xi", &axs, (int64_t)K * 2 * D, &s4))
        return 1;
      if (s1 != n_Phi || s2 != n_Phi || s3 != n_Phi || s4 != n_Phi) return efail("'theta_est' arrays have inconsistent lengths");
      xi_est = med_block(xs, (int64_t)K * P * D * M);
      gamma_xi_est = med_block(gxs, (int64_t)K * P * D * M);
      delta_xi_est = med_block(dxs, (int64_t)K * M * D);
      A_xi_est = med_block(axs, (int64_t)K * 2 * D);
    }
  }
  // ---- one chain of the full sweep from the medians (BFMMM.h:1486-1498, 1500-1554) ----
  int tt_blocks = 0, tt_accepted = 0;
  bfmmm_config cfg;
  make_cfg(a, batched ? rs : T, &cfg);          // chain slots in HBM
  bfmmm_handle* h = nullptr;
  if (make_handle(a, &cfg, a->device, 1, &h)) return efail_lib();
  const uint32_t mask_ws = BFMMM_SWEEP_WARM | (a->X ? (BFMMM_COV_MEAN | (a->covariance_adj ? BFMMM_COV_XI : 0)) : 0);   // BFMMM.h:4248-4312 / 4809-4894
  int rc = attach_cov(h, a) ||
           (a->X && (bfmmm_set_state(h, "eta", eta_est.data(), (int64_t)eta_est.size()) ||
                     bfmmm_set_state(h, "tau_eta", tau_eta_est.data(), (int64_t)tau_eta_est.size()))) ||
           (a->X && a->covariance_adj &&
            (bfmmm_set_state(h, "xi", xi_est.data(), (int64_t)xi_est.size()) ||
             bfmmm_set_state(h, "gamma_xi", gamma_xi_est.data(), (int64_t)gamma_xi_est.size()) ||
             bfmmm_set_state(h, "delta_xi", delta_xi_est.data(), (int64_t)delta_xi_est.size()) ||
             bfmmm_set_state(h, "A_xi", A_xi_est.data(), (int64_t)A_xi_est.size()))) ||
           bfmmm_set_state(h, "Z", Z_est.data(), (int64_t)Z_est.size()) || bfmmm_set_state(h, "pi", pi_est.data(), K) ||
           bfmmm_set_state(h, "alpha_3", &alpha_3_est, 1) || bfmmm_set_state(h, "delta", delta_est.data(), (int64_t)K * M) ||
           bfmmm_set_state(h, "gamma", gamma_est.data(), (int64_t)K * P * M) ||
           bfmmm_set_state(h, "Phi", Phi_est.data(), (int64_t)K * P * M) || bfmmm_set_state(h, "A", A_est.data(), (int64_t)K * 2) ||
           bfmmm_set_state(h, "nu", nu_est.data(), (int64_t)K * P) || bfmmm_set_state(h, "tau", tau_est.data(), K) ||
           bfmmm_set_state(h, "sigma_sq", &sigma_est, 1) || bfmmm_set_state(h, "chi", chi_est.data(), (int64_t)n * M);
  // iterations [i_begin, i_end) of BFMMM.h:1500-1672: the sweep of iteration i, then -- every n_temp_trans iterations --
  // the tempered-transition block
  auto run_range = [&](int i_begin, int i_end) -> int {
    if (a->n_temp_trans == 0) return bfmmm_run(h, mask_ws, i_begin, i_end - i_begin, a->seed, (uint32_t)a->chain_offset, 0, 1.0);
    int i0 = i_begin, r2 = 0;
    for (int i = i_begin; i < i_end && !r2; ++i)
      if (i > 0 && (i % a->n_temp_trans) == 0) {
        double logA; int acc;
        r2 = bfmmm_run(h, mask_ws, i0, i + 1 - i0, a->seed, (uint32_t)a->chain_offset, 0, 1.0) ||
             bfmmm_tempered_transition(h, mask_ws, i, a->N_t, a->beta_N_t, a->seed, (uint32_t)a->chain_offset, &logA, &acc);
        i0 = i + 1; tt_blocks += 1; tt_accepted += acc;
      }
    if (!r2 && i0 < i_end) r2 = bfmmm_run(h, mask_ws, i0, i_end - i0, a->seed, (uint32_t)a->chain_offset, 0, 1.0);
    return r2;
  };
  bool io_failed = false, interrupted = false;
  // run_range in pieces of progress_every iterations, the caller's callback in between (slot of iteration i: i - base)
  auto run_reporting = [&](int i_begin, int i_end, int base) -> int {
    if (!a->progress_cb || a->progress_every <= 0) return run_range(i_begin, i_end);
    std::vector<double> ll((size_t)(i_end - base));
    for (int b = i_begin; b < i_end; b += a->progress_every) {
      const int e = std::min(b + a->progress_every, i_end);
      if (run_range(b, e)) return 1;
      if (bfmmm_get_chain(h, "loglik", e - base, ll.data(), (int64_t)ll.size())) return 1;
      if (a->progress_cb(e - 1, ll[(size_t)(e - 1 - base)], a->progress_user)) { interrupted = true; return 1; }
    }
    return 0;
  };
  if (!rc) {
    if (!batched) {
      rc = run_reporting(0, T, 0);
    } else {
      // BFMMM.h:1680-1746: iteration i lives in slot i % r_stored_iters; whenever a batch is full (and i > 1) it is
      // thinned and saved, and the next batch reuses the slots
      const std::string dir(a->dir);
      int q = 0;
      for (int b0 = 0; b0 < T && !rc; b0 += rs) {
        const int b1 = std::min(b0 + rs, T);
        rc = bfmmm_set_slot_base(h, b0) || run_reporting(b0, b1, b0);
        if (!rc && b1 - b0 == rs && b1 - 1 > 1) {
          if (save_batch(h, a, dir, q, rs)) { io_failed = true; rc = 1; }
          q += 1;
        }
      }
    }
  }
  if (rc && io_failed) { bfmmm_destroy(h); return 1; }
  if (rc && interrupted) { bfmmm_destroy(h); return efail("the run was interrupted by the progress callback"); }
  if (rc) { efail_lib(); bfmmm_destroy(h); return 1; }
  bfmmm_result* r = bfmmm_result_create();
  if (!batched) {
    // r_stored_iters defaults to tot_mcmc_iters + 1 slots (UserFunctions.cpp:1510-1541): slot T repeats slot T-1
    const int TT = T + 1;
    rc = fetch_basis(h, r, a, "B_obs") || fetch(h, r, "Z", "Z", T, (int64_t)n * K, {n, K, TT}, 1) ||
         fetch(h, r, "nu", "nu", T, (int64_t)K * P, {K, P, TT}, 1) || fetch(h, r, "chi", "chi", T, (int64_t)n * M, {n, M, TT}, 1) ||
         fetch(h, r, "pi", "pi", T, K, {K, TT}, 1) || fetch(h, r, "alpha_3", "alpha_3", T, 1, {TT}, 1) ||
         fetch(h, r, "A", "A", T, (int64_t)K * 2, {K, 2, TT}, 1) || fetch(h, r, "delta", "delta", T, (int64_t)K * M, {K, M, TT}, 1) ||
         fetch(h, r, "sigma_sq", "sigma_sq", T, 1, {TT}, 1) || fetch_tau(h, r, T, K, 1) ||
         fetch(h, r, "gamma", "gamma", T, (int64_t)K * P * M, {K, P, M, TT}, 1) ||
         fetch(h, r, "Phi", "Phi", T, (int64_t)K * P * M, {K, P, M, TT}, 1) || fetch(h, r, "loglik", "loglik", T, 1, {TT}, 1) ||
         fetch_cov(h, r, a, T, a->covariance_adj != 0, 1);
  } else {
    // the r_stored_iters slots of the last batch in memory; when the run ended on a full batch the reference has
    // already copied the final state into slot 0 for the iteration that never comes ("reset all parameters",
    // BFMMM.h:1732-1743; loglik is not part of that reset)
    rc = fetch_basis(h, r, a, "B_obs") || fetch(h, r, "Z", "Z", rs, (int64_t)n * K, {n, K, rs}) ||
         fetch(h, r, "nu", "nu", rs, (int64_t)K * P, {K, P, rs}) || fetch(h, r, "chi", "chi", rs, (int64_t)n * M, {n, M, rs}) ||
         fetch(h, r, "pi", "pi", rs, K, {K, rs}) || fetch(h, r, "alpha_3", "alpha_3", rs, 1, {rs}) ||
         fetch(h, r, "A", "A", rs, (int64_t)K * 2, {K, 2, rs}) || fetch(h, r, "delta", "delta", rs, (int64_t)K * M, {K, M, rs}) ||
         fetch(h, r, "sigma_sq", "sigma_sq", rs, 1, {rs}) || fetch_tau(h, r, rs, K) ||
         fetch(h, r, "gamma", "gamma", rs, (int64_t)K * P * M, {K, P, M, rs}) ||
         fetch(h, r, "Phi", "Phi", rs, (int64_t)K * P * M, {K, P, M, rs}) || fetch(h, r, "loglik", "loglik", rs, 1, {rs}) ||
         fetch_cov(h, r, a, rs, a->covariance_adj != 0);
    if (!rc && T % rs == 0 && T - 1 > 1)
      for (const std::string& nm : r->names) {
        if (nm == "loglik" || nm == "B_obs") continue;
        ResArr& arr = r->arrs[nm];
        if (nm == "tau") { for (int k = 0; k < K; ++k) arr.data[(size_t)rs * k] = arr.data[(size_t)rs * k + rs - 1]; continue; }
        const size_t len = arr.data.size() / (size_t)rs;
        std::copy(arr.data.begin() + len * (rs - 1), arr.data.begin() + len * rs, arr.data.begin());
      }
  }
  bfmmm_destroy(h);
  if (rc) { bfmmm_result_free(r); return 1; }
  if (a->n_temp_trans > 0) {     // (the reference only prints its running acceptance rate, BFMMM.h:1675; exposed here)
    put(r, "tt_blocks", {(double)tt_blocks}, {1});
    put(r, "tt_accepted", {(double)tt_accepted}, {1});
  }
  *out = r;
  return 0;
}

// ---- multivariate aliases (BMVMMM_*): the same drivers on G_i = I records ------------------------
static int need_mv(const bfmmm_entry_args* a) {
  if (!a || a->model != BFMMM_MODEL_MULTIVARIATE) return efail("bfmmm_BMVMMM_*: args.model must be BFMMM_MODEL_MULTIVARIATE");
  return 0;
}
extern "C" int bfmmm_BMVMMM_Nu_Z_multiple_try(const bfmmm_entry_args* a, bfmmm_result** out) {
  return need_mv(a) || bfmmm_BFMMM_Nu_Z_multiple_try(a, out);
}
extern "C" int bfmmm_BMVMMM_Theta_est(const bfmmm_entry_args* a, const bfmmm_result* mt, bfmmm_result** out) {
  return need_mv(a) || bfmmm_BFMMM_Theta_est(a, mt, out);
}
extern "C" int bfmmm_BMVMMM_warm_start(const bfmmm_entry_args* a, const bfmmm_result* mt, const bfmmm_result* te,
                                       bfmmm_result** out) {
  return need_mv(a) || bfmmm_BFMMM_warm_start(a, mt, te, out);
}

// ---- high-dimensional functional model (BHDFMMM_*): the same drivers over the tensor-product basis ------------------
static int need_hd(const bfmmm_entry_args* a) {
  if (!a || a->dim <= 0) return efail("bfmmm_BHDFMMM_*: args.dim must be positive");
  return 0;
}
extern "C" int bfmmm_BHDFMMM_Nu_Z_multiple_try(const bfmmm_entry_args* a, bfmmm_result** out) {
  return need_hd(a) || bfmmm_BFMMM_Nu_Z_multiple_try(a, out);
}
extern "C" int bfmmm_BHDFMMM_Theta_est(const bfmmm_entry_args* a, const bfmmm_result* mt, bfmmm_result** out) {
  return need_hd(a) || bfmmm_BFMMM_Theta_est(a, mt, out);
}
extern "C" int bfmmm_BHDFMMM_warm_start(const bfmmm_entry_args* a, const bfmmm_result* mt, const bfmmm_result* te,
                                        bfmmm_result** out) {
  return need_hd(a) || bfmmm_BFMMM_warm_start(a, mt, te, out);
}
