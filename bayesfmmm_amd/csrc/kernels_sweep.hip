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
//   k_factor    : one workgroup per direction: C_a (Gauss-Jordan inverse), chol_lower(C_a)
//   k_sweep     : pi, alpha_3, Phi sweep, delta, A, gamma, nu sweep, tau, sigma^2 in reference order
//   k_loglik    : calcLikelihood (CalculateLikelihood.h:19-44) from the per-curve residual sums
#include "model.hpp"
#include "rng.hpp"

namespace bfmmm {

typedef double double4_t __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------
// pair-Gram
// ---------------------------------------------------------------------------------------------
// grid = (NTG tile groups, NKS k-slices); block = 256 (4 waves).  Tile t of the NT output tiles
// belongs to group t % NTG and, inside the group, to wave (t / NTG) % 4.
__global__ __launch_bounds__(256) void k_pair_gram(Ctx c, int NTG, int KS) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const Dims& d = c.d;
  const int n = d.n, K = d.K, M = d.M, MD = d.MD;
  const int ks = blockIdx.y, tg = blockIdx.x;
  const int i0 = ks * KS;
  double* sZ = smem;                 // KS x K
  double* sC = smem + (size_t)KS * K;  // KS x MD   (chit: 1, chi_1..chi_M)
  for (int q = threadIdx.x; q < KS * K; q += 256) {
    const int il = q / K, k = q - il * K, i = i0 + il;
    sZ[q] = (i < n) ? c.Z[i + (size_t)n * k] : 0.0;
  }
  for (int q = threadIdx.x; q < KS * MD; q += 256) {
    const int il = q / MD, mt = q - il * MD, i = i0 + il;
    sC[q] = (i < n) ? ((mt == 0) ? 1.0 : c.chi[i + (size_t)n * (mt - 1)]) : 0.0;
  }
  __syncthreads();
  (void)M;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int lr = lane & 15, kq = lane >> 4;
  const int n_pair_tiles = d.RT * d.CTG;
  for (int t = tg + NTG * wave; t < d.NT; t += NTG * 4) {
    int row, col;
    bool rvalid, cvalid;
    int j1 = 0, j2 = 0, m1 = 0, m2 = 0;
    if (t < n_pair_tiles) {
      const int rt = t / d.CTG, ct = t - rt * d.CTG;
      row = rt * 16 + lr;
      col = ct * 16 + lr;
      rvalid = row < d.R;
      cvalid = col < d.LG;
      if (rvalid) {
        const int zz = row / d.NCC, cc = row - zz * d.NCC;
        // invert the packed-upper-triangle indices
        int a = 0, rem = zz;
        while (rem >= K - a) { rem -= K - a; ++a; }
        j1 = a; j2 = a + rem;
        a = 0; rem = cc;
        while (rem >= MD - a) { rem -= MD - a; ++a; }
        m1 = a; m2 = a + rem;
      }
    } else {
      const int t2 = t - n_pair_tiles;
      const int at = t2 / d.CTS, cs = t2 - at * d.CTS;
      row = at * 16 + lr;
      col = d.LG + cs * 16 + lr;
      rvalid = row < d.A;
      cvalid = col < d.LG + d.P;
      if (rvalid) { j1 = row / MD; m1 = row - j1 * MD; }
    }
    const bool single = t >= n_pair_tiles;
    double4_t acc = {0.0, 0.0, 0.0, 0.0};
    const double* recp = c.rec + (size_t)(i0 + kq) * d.LREC + col;
    for (int kk = 0; kk < KS; kk += 4) {
      const int il = kk + kq;
      double a = 0.0, b = 0.0;
      if (rvalid) {
        a = sZ[il * K + j1] * sC[il * MD + m1];
        if (!single) a *= sZ[il * K + j2] * sC[il * MD + m2];
      }
      if (cvalid && (i0 + il) < n) b = recp[(size_t)kk * d.LREC];
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    double* out = c.pg_part + ((size_t)ks * d.NT + t) * 256 + lane;
    out[0] = acc[0]; out[64] = acc[1]; out[128] = acc[2]; out[192] = acc[3];
  }
}

// one thread per element of every output tile; fixed summation order over the k-slices
__global__ __launch_bounds__(256) void k_pg_reduce(Ctx c, int NKS) {
  const Dims& d = c.d;
  const int gid = blockIdx.x * 256 + threadIdx.x;
  if (gid >= d.NT * 256) return;
  const int t = gid >> 8, q = gid & 255;
  const int r = q >> 6, lane = q & 63;
  const int rit = (lane >> 4) + 4 * r, cit = lane & 15;   // D layout of v_mfma_f64_16x16x4_f64
  double s = 0.0;
  for (int ks = 0; ks < NKS; ++ks) s += c.pg_part[((size_t)ks * d.NT + t) * 256 + q];
  const int n_pair_tiles = d.RT * d.CTG;
  if (t < n_pair_tiles) {
    const int rt = t / d.CTG, ct = t - rt * d.CTG;
    const int row = rt * 16 + rit, col = ct * 16 + cit;
    if (row < d.R && col < d.LG) c.H[(size_t)row * d.LG + col] = s;
  } else {
    const int t2 = t - n_pair_tiles;
    const int at = t2 / d.CTS, cs = t2 - at * d.CTS;
    const int row = at * 16 + rit, col = cs * 16 + cit;
    if (row < d.A && col < d.P) c.tvec[(size_t)row * d.P + col] = s;
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

// (H_block * v)[p] for a band-packed symmetric block
__device__ inline double band_mv(const double* __restrict__ Hb, const double* v, int P, int BW, int p) {
  double s = Hb[p] * v[p];
  for (int dd = 1; dd <= BW; ++dd) {
    if (p + dd < P) s += Hb[dd * P + p] * v[p + dd];
    if (p - dd >= 0) s += Hb[dd * P + p - dd] * v[p - dd];
  }
  return s;
}

// deterministic tree reduction over blockDim.x == 256 values held in LDS scratch
__device__ inline double block_sum256(double v, double* scratch) {
  __syncthreads();
  scratch[threadIdx.x] = v;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) scratch[threadIdx.x] += scratch[threadIdx.x + o];
    __syncthreads();
  }
  const double r = scratch[0];
  __syncthreads();
  return r;
}

// ---------------------------------------------------------------------------------------------
// k_factor: one workgroup per active direction.
//   Prec = (beta/sigma^2) H_aa + Prior;  C = Prec^-1 (in-place Gauss-Jordan; SPD so no pivoting);
//   C <- (C + C')/2 (UpdateNu.h:68);  L = chol_lower(C) (what arma::mvnrnd factors, UpdateNu.h:69).
// Non-positive pivots set dyn->status bit 0: the reference would fall back to pinv / eig there.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_factor(Ctx c) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const Dims& d = c.d;
  const int P = d.P, MD = d.MD, K = d.K;
  const int a = blockIdx.x;
  const int j = a / MD, mt = a - j * MD;
  if (mt == 0 && !(c.mask & U_NU)) return;
  if (mt > 0 && !(c.mask & U_PHI)) return;
  double* S = smem;               // P x P column-major
  double* rowk = smem + (size_t)P * P;
  double* colk = rowk + P;
  const Dyn* dyn = c.dyn;
  const double f = dyn->beta / dyn->sigma2;
  const double* Hb = c.H + (size_t)hrow(d, a, a) * d.LG;
  // prior scale: tau_j (nu) or tilde_tau(j, m) = prod_{m' <= m} delta(j, m') (BFMMM.h:1514-1519)
  double tt = 1.0;
  if (mt > 0)
    for (int m2 = 0; m2 < mt; ++m2) tt *= c.delta[j + (size_t)K * m2];
  const double tau_j = dyn->tau[j];
  for (int e = threadIdx.x; e < P * P; e += 256) {
    const int p = e % P, q = e / P;
    const int lo = min(p, q), dd = max(p, q) - lo;
    double v = (dd <= d.BW) ? f * Hb[dd * P + lo] : 0.0;
    if (mt == 0) {
      if (d.mv) { if (p == q) v += 1.0 / tau_j; }               // UpdateNu.h:197 (MV)
      else v += tau_j * c.Pmat[p + (size_t)P * q];                // UpdateNu.h:66
    } else if (p == q) {
      v += tt * c.gamma[j + (size_t)K * (p + (size_t)P * (mt - 1))];   // UpdatePhi.h:76-78
    }
    S[e] = v;
  }
  __syncthreads();
  bool bad = false;
  // in-place Gauss-Jordan inversion
  for (int k = 0; k < P; ++k) {
    if ((int)threadIdx.x < P) { rowk[threadIdx.x] = S[k + (size_t)P * threadIdx.x]; colk[threadIdx.x] = S[threadIdx.x + (size_t)P * k]; }
    __syncthreads();
    const double piv = rowk[k];
    if (!(piv > 0.0)) bad = true;
    const double inv = 1.0 / piv;
    for (int e = threadIdx.x; e < P * P; e += 256) {
      const int p = e % P, q = e / P;
      double v;
      if (p == k && q == k) v = inv;
      else if (p == k) v = rowk[q] * inv;
      else if (q == k) v = -colk[p] * inv;
      else v = S[e] - colk[p] * rowk[q] * inv;
      S[e] = v;
    }
    __syncthreads();
  }
  // symmetrise and store C
  double* Cg = c.Cmat + (size_t)a * P * P;
  for (int e = threadIdx.x; e < P * P; e += 256) {
    const int p = e % P, q = e / P;
    Cg[e] = 0.5 * (S[p + (size_t)P * q] + S[q + (size_t)P * p]);
  }
  __syncthreads();
  for (int e = threadIdx.x; e < P * P; e += 256) S[e] = Cg[e];
  __syncthreads();
  // right-looking Cholesky, lower triangle in place
  for (int k = 0; k < P; ++k) {
    const double piv = S[k + (size_t)P * k];
    if (!(piv > 0.0)) bad = true;
    __syncthreads();
    const double lkk = sqrt(piv);
    if ((int)threadIdx.x < P) {
      const int p = threadIdx.x;
      if (p == k) S[k + (size_t)P * k] = lkk;
      else if (p > k) S[p + (size_t)P * k] = S[p + (size_t)P * k] / lkk;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < P * P; e += 256) {
      const int p = e % P, q = e / P;
      if (q > k && p >= q) S[e] -= S[p + (size_t)P * k] * S[q + (size_t)P * k];
    }
    __syncthreads();
  }
  double* Lg = c.Lmat + (size_t)a * P * P;
  for (int e = threadIdx.x; e < P * P; e += 256) {
    const int p = e % P, q = e / P;
    Lg[e] = (p >= q) ? S[e] : 0.0;
  }
  if (bad && threadIdx.x == 0) atomicOr(&c.dyn->status, 1u);
}

// ---------------------------------------------------------------------------------------------
// k_sweep: the sequentially dependent part of one Gibbs iteration, one workgroup of 256 threads.
// ---------------------------------------------------------------------------------------------
struct SweepLds {
  double* th;     // A x P   current theta of the active directions
  double* tv;     // A x P   t_a
  double* r;      // A x P   r_a = t_a - sum_b H_ab theta_b
  double* rhs;    // P
  double* z;      // P
  double* nw;     // P
  double* dl;     // P
  double* red;    // 256
  double* sm;     // small scratch (64)
};

__device__ inline int full_dir(const Dims& d, int a) {   // active direction -> row of c.theta
  const int j = a / d.MD, mt = a - j * d.MD;
  return j * (d.M + 1) + mt;
}

// one Gaussian block draw for direction a (UpdateNu.h:64-69 / UpdatePhi.h:72-82)
__device__ inline void gauss_step(const Ctx& c, const SweepLds& L, int a, uint32_t upd, uint32_t idx0,
                                  const RngKey& key, double f) {
  const Dims& d = c.d;
  const int P = d.P, A = d.A, tid = threadIdx.x;
  const double* Haa = c.H + (size_t)hrow(d, a, a) * d.LG;
  if (tid < P) {
    L.rhs[tid] = f * (L.r[a * P + tid] + band_mv(Haa, L.th + a * P, P, d.BW, tid));
  } else if (tid >= 64 && tid < 64 + P) {
    L.z[tid - 64] = rnorm(key, upd, idx0 + (uint32_t)(tid - 64));
  }
  __syncthreads();
  // new = C rhs + L z : 8 threads per row, fixed-order combine
  const double* Cg = c.Cmat + (size_t)a * P * P;
  const double* Lg = c.Lmat + (size_t)a * P * P;
  for (int p = tid >> 3; p < P; p += 32) {
    const int seg = tid & 7;
    double acc = 0.0;
    for (int q = seg; q < P; q += 8) {
      acc += Cg[p + (size_t)P * q] * L.rhs[q];
      if (q <= p) acc += Lg[p + (size_t)P * q] * L.z[q];
    }
    // combine the 8 segment partials in a fixed order (lanes tid..tid+7 share a wave)
    acc += __shfl_xor(acc, 1, 8);
    acc += __shfl_xor(acc, 2, 8);
    acc += __shfl_xor(acc, 4, 8);
    if (seg == 0) { L.nw[p] = acc; L.dl[p] = acc - L.th[a * P + p]; }
  }
  __syncthreads();
  if (tid < P) L.th[a * P + tid] = L.nw[tid];
  // r_b -= H_ba (theta_new - theta_old) for every direction b
  for (int e = tid; e < A * P; e += 256) {
    const int b = e / P, p = e - b * P;
    const double* Hb = c.H + (size_t)hrow(d, b, a) * d.LG;
    L.r[e] -= band_mv(Hb, L.dl, P, d.BW, p);
  }
  __syncthreads();
}

__device__ inline double logGamma_ref(double x) { return log(tgamma(x)); }   // Distributions.h:13-15

__device__ inline double lpdf_a1(const Hyper& h, double a, double delta) {    // UpdateA.h:17-24
  return -logGamma_ref(a) + (a - 1) * log(delta) + (h.alpha1l - 1) * log(a) - (a * h.beta1l);
}
__device__ inline double lpdf_a2(const Hyper& h, double a, int M, const double* delta_row, int stride) {  // :33-44
  const double x = M - 1;
  double lpdf = -x * logGamma_ref(a) + (h.alpha2l - 1) * log(a) - (a * h.beta2l);
  for (int i = 1; i < M; ++i) lpdf = lpdf + (a - 1) * log(delta_row[(size_t)i * stride]);
  return lpdf;
}

__global__ __launch_bounds__(256) void k_sweep(Ctx c) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const Dims& d = c.d;
  const int P = d.P, A = d.A, K = d.K, M = d.M, MD = d.MD, n = d.n;
  const int tid = threadIdx.x;
  Dyn* dyn = c.dyn;
  SweepLds L;
  L.th = smem; L.tv = L.th + A * P; L.r = L.tv + A * P; L.rhs = L.r + A * P;
  L.z = L.rhs + PMAX; L.nw = L.z + PMAX; L.dl = L.nw + PMAX; L.red = L.dl + PMAX; L.sm = L.red + 256;
  const uint32_t slot = dyn->slot;
  const RngKey key = make_key(c.seed, c.chain, dyn->iter, dyn->tt_step);
  const double beta = dyn->beta;
  const uint32_t mask = c.mask;

  // ---------------- pi (updatePi_PM, UpdatePi.h:84-116) and alpha_3 (UpdateAlpha3.h:36-63) -------
  if (mask & (U_PI | U_ALPHA3)) {
    // S_k = sum_i log Z_ik from the block partials of k_curve_z, fixed order
    for (int k = 0; k < K; ++k) {
      double acc = 0.0;
      const int per = (c.nblk_curve + 255) / 256;
      for (int b = tid * per; b < min(c.nblk_curve, (tid + 1) * per); ++b) acc += c.logz_part[(size_t)b * K + k];
      const double s = block_sum256(acc, L.red);
      if (tid == 0) L.sm[k] = s;
    }
    __syncthreads();
    if (tid == 0) {
      double pi[KMAX], pi_ph[KMAX], a_old[KMAX], a_new[KMAX], ap[KMAX];
      double alpha3 = dyn->alpha3;
      for (int k = 0; k < K; ++k) pi[k] = dyn->pi[k];
      if (mask & U_PI) {
        double sum = 0.0;
        for (int k = 0; k < K; ++k) {
          a_old[k] = c.h.a_pi_PM * pi[k];
          const double aa = (a_old[k] <= 0) ? 10.0 : a_old[k];
          pi_ph[k] = rgamma(key, UPD_PI_PROP, (uint32_t)k, aa, 1.0);
          sum += pi_ph[k];
        }
        for (int k = 0; k < K; ++k) { pi_ph[k] /= sum; a_new[k] = c.h.a_pi_PM * pi_ph[k]; }
        double lpdf_new = 0.0, lpdf_old = 0.0, pn = 0.0, po = 0.0;
        for (int k = 0; k < K; ++k) {
          lpdf_new += (c.h.c[k] - 1) * log(pi_ph[k]) + ((alpha3 * pi_ph[k]) - 1) * L.sm[k];
          lpdf_old += (c.h.c[k] - 1) * log(pi[k]) + ((alpha3 * pi[k]) - 1) * L.sm[k];
          pn += (a_old[k] - 1) * log(pi_ph[k]);
          po += (a_new[k] - 1) * log(pi[k]);
        }
        for (int k = 0; k < K; ++k) ap[k] = alpha3 * pi_ph[k];
        lpdf_new -= n * calc_lB(K, ap);
        for (int k = 0; k < K; ++k) ap[k] = alpha3 * pi[k];
        lpdf_old -= n * calc_lB(K, ap);
        const double lpn = pn - calc_lB(K, a_old);
        const double lpo = po - calc_lB(K, a_new);
        const double acc = lpdf_new - lpdf_old + lpo - lpn;
        const double u = runif(key, UPD_PI_ACC, 0);
        if (log(u) < acc)
          for (int k = 0; k < K; ++k) pi[k] = pi_ph[k];
        for (int k = 0; k < K; ++k) dyn->pi[k] = pi[k];
      }
      if (mask & U_ALPHA3) {
        const double sd = c.h.var_alpha3;
        const double ph = rtruncnorm_lo(key, UPD_A3_PROP, 0, alpha3, sd, 0.0);
        double l_old = (-c.h.b) * alpha3, l_new = (-c.h.b) * ph;
        for (int k = 0; k < K; ++k) {
          l_old += ((alpha3 * pi[k]) - 1) * L.sm[k];
          l_new += ((ph * pi[k]) - 1) * L.sm[k];
        }
        for (int k = 0; k < K; ++k) ap[k] = alpha3 * pi[k];
        l_old -= n * calc_lB(K, ap);
        for (int k = 0; k < K; ++k) ap[k] = ph * pi[k];
        l_new -= n * calc_lB(K, ap);
        // d_truncnorm(x, x, sd, 0, Inf, log) with x = the *other* state (UpdateAlpha3.h:23-24)
        l_old += dtruncnorm_lo_log(ph, ph, sd, 0.0);
        l_new += dtruncnorm_lo_log(alpha3, alpha3, sd, 0.0);
        const double u = runif(key, UPD_A3_ACC, 0);
        if (log(u) < l_new - l_old) alpha3 = ph;
        dyn->alpha3 = alpha3;
      }
    }
    __syncthreads();
  }

  const bool need_gauss = (mask & (U_PHI | U_NU | U_SIGMA)) != 0;
  // ---------------- load theta (always: delta / gamma / tau read it) --------------------------
  for (int e = tid; e < A * P; e += 256) {
    const int a = e / P, p = e - a * P;
    L.th[e] = c.theta[(size_t)full_dir(d, a) * P + p];
  }
  __syncthreads();
  if (need_gauss) {
    // ---------------- t; r = t - H theta ------------------------------------------------------
    for (int e = tid; e < A * P; e += 256) L.tv[e] = c.tvec[e];
    __syncthreads();
    for (int e = tid; e < A * P; e += 256) {
      const int a = e / P, p = e - a * P;
      double acc = L.tv[e];
      for (int b = 0; b < A; ++b) acc -= band_mv(c.H + (size_t)hrow(d, a, b) * d.LG, L.th + b * P, P, d.BW, p);
      L.r[e] = acc;
    }
    __syncthreads();
  }
  const double f = beta / dyn->sigma2;

  // ---------------- Phi (updatePhi: j outer, m inner) ----------------------------------------
  if ((mask & U_PHI) && MD > 1) {
    for (int j = 0; j < K; ++j)
      for (int m = 0; m < M; ++m)
        gauss_step(c, L, j * MD + m + 1, UPD_PHI, (uint32_t)((j * M + m) * P), key, f);
    // publish Phi so that delta / gamma below read the new values
    for (int e = tid; e < K * M * P; e += 256) {
      const int jm = e / P, p = e - jm * P, j = jm / M, m = jm - j * M;
      c.theta[(size_t)(j * (M + 1) + m + 1) * P + p] = L.th[(j * MD + m + 1) * P + p];
    }
    __syncthreads();
  }

  // ---------------- delta (updateDelta, UpdateDelta.h:17-64) ----------------------------------
  if ((mask & U_DELTA) && MD > 1) {
    // S_km = sum_p gamma(k,p,m) phi(k,p,m)^2
    if (tid < K * M) {
      const int k = tid / M, m = tid - k * M;
      double acc = 0.0;
      for (int p = 0; p < P; ++p) {
        const double ph = L.th[(k * MD + m + 1) * P + p];
        acc += c.gamma[k + (size_t)K * (p + (size_t)P * m)] * (ph * ph);
      }
      L.red[tid] = acc;
    }
    __syncthreads();
    if (tid < K) {
      const int k = tid;
      for (int i = 0; i < M; ++i) {
        double param1, param2 = 1.0;
        if (i == 0) {
          param1 = c.Aa[k] + ((P * M) / 2.0);
          param2 += 0.5 * L.red[k * M + 0];
          for (int m = 1; m < M; ++m) {
            double tt = 1.0;
            for (int nn = 1; nn <= m; ++nn) tt *= c.delta[k + (size_t)K * nn];
            param2 += 0.5 * tt * L.red[k * M + m];
          }
        } else {
          param1 = c.Aa[k + (size_t)K] + ((P * (M - i)) / 2.0);
          for (int m = i; m < M; ++m) {
            double tt = 1.0;
            for (int nn = 0; nn <= m; ++nn)
              if (nn != i) tt *= c.delta[k + (size_t)K * nn];
            param2 += 0.5 * tt * L.red[k * M + m];
          }
        }
        c.delta[k + (size_t)K * i] = rgamma(key, UPD_DELTA, (uint32_t)(k * M + i), param1, 1.0 / param2);
      }
    }
    __syncthreads();
  }

  // ---------------- A (updateA, UpdateA.h:58-123) ---------------------------------------------
  if ((mask & U_A) && MD > 1) {
    if (tid < K * 2) {
      const int j = tid / 2, i = tid - 2 * j;
      const bool first = (i == 0);
      const double sd = first ? (c.h.var_epsilon1 / c.h.beta1l) : (c.h.var_epsilon2 / c.h.beta2l);
      const double cur = c.Aa[j + (size_t)K * i];
      const double na = rtruncnorm_lo(key, UPD_A_PROP, (uint32_t)(j * 2 + i), cur, sd, 0.0);
      double l0, l1;
      if (first) {
        l0 = lpdf_a1(c.h, cur, c.delta[j]);
        l1 = lpdf_a1(c.h, na, c.delta[j]);
      } else {
        l0 = lpdf_a2(c.h, cur, M, c.delta + j, K);
        l1 = lpdf_a2(c.h, na, M, c.delta + j, K);
      }
      const double acc = (l1 + dtruncnorm_lo_log(cur, na, sd, 0.0)) - l0 - dtruncnorm_lo_log(na, cur, sd, 0.0);
      const double u = runif(key, UPD_A_ACC, (uint32_t)(j * 2 + i));
      if (log(u) < acc) c.Aa[j + (size_t)K * i] = na;
    }
    __syncthreads();
  }

  // ---------------- gamma (updateGamma, UpdateGamma.h:17-37) -----------------------------------
  if ((mask & U_GAMMA) && MD > 1) {
    for (int e = tid; e < K * P * M; e += 256) {
      // e = (i*P + l)*M + j  (reference loop order i, l, j)
      const int jj = e % M, il = e / M, l = il % P, i = il / P;
      double ph = 1.0;
      for (int j2 = 0; j2 <= jj; ++j2) ph *= c.delta[i + (size_t)K * j2];
      const double phi = L.th[(i * MD + jj + 1) * P + l];
      c.gamma[i + (size_t)K * (l + (size_t)P * jj)] =
          rgamma(key, UPD_GAMMA, (uint32_t)e, (c.h.nu_1 + 1) / 2, 2 / (c.h.nu_1 + ph * (phi * phi)));
    }
    __syncthreads();
  }

  // ---------------- nu (updateNu) -------------------------------------------------------------
  if (mask & U_NU) {
    for (int j = 0; j < K; ++j) gauss_step(c, L, j * MD, UPD_NU, (uint32_t)(j * P), key, f);
    for (int e = tid; e < K * P; e += 256) {
      const int j = e / P, p = e - j * P;
      c.theta[(size_t)(j * (M + 1)) * P + p] = L.th[(j * MD) * P + p];
    }
    __syncthreads();
  }

  // ---------------- tau (updateTau, UpdateTau.h:18-36; MV :47-63) ------------------------------
  if (mask & U_TAU) {
    for (int k = 0; k < K; ++k) {
      double acc = 0.0;
      if (tid < P) {
        const double* nu = L.th + (k * MD) * P;
        const double vp = nu[tid];
        double s = 0.0;
        if (d.mv) s = vp;
        else
          for (int q = 0; q < P; ++q) s += c.Pmat[tid + (size_t)P * q] * nu[q];
        acc = vp * s;
      }
      const double qf = block_sum256(acc, L.red);
      if (tid == 0) {
        const double a = c.h.alpha_nu + (P / 2);                   // integer division, UpdateTau.h:29
        const double b = c.h.beta_nu + (0.5 * qf);
        const double g = rgamma(key, UPD_TAU, (uint32_t)k, a, 1.0 / b);
        dyn->tau[k] = d.mv ? (1.0 / g) : g;
      }
    }
    __syncthreads();
  }

  // ---------------- sigma^2 (updateSigma, UpdateSigma.h:22-58) ---------------------------------
  if (mask & U_SIGMA) {
    // RSS = YY - sum_a theta_a'(t_a + r_a)
    double acc = 0.0;
    for (int e = tid; e < A * P; e += 256) acc += L.th[e] * (L.tv[e] + L.r[e]);
    const double q = block_sum256(acc, L.red);
    if (tid == 0) {
      const double rss = c.YY - q;
      double a, b;
      const bool tempered = (dyn->tt_step != 0);
      if (tempered) {
        a = (d.mv ? ((beta * (double)d.n_obs_total) / 2) : (beta * (double)d.n_obs_total) / 2);
        b = (beta / 2) * rss;
      } else {
        a = d.mv ? (double)(d.n_obs_total / 2) : (double)d.half_sum;    // UpdateSigma.h:49 / :150
        b = 0.5 * rss;
      }
      b += c.h.beta_0;
      a += c.h.alpha_0;
      dyn->sigma2 = 1.0 / rgamma(key, UPD_SIGMA, 0, a, 1.0 / b);
      dyn->rss = rss;
    }
    __syncthreads();
  }

  // ---------------- chain slots ---------------------------------------------------------------
  {
    double* s_nu = c.c_nu + (size_t)slot * K * P;
    double* s_phi = c.c_Phi + (size_t)slot * K * P * M;
    double* s_gam = c.c_gamma + (size_t)slot * K * P * M;
    for (int e = tid; e < K * P; e += 256) {
      const int p = e / K, k = e - p * K;
      s_nu[e] = c.theta[(size_t)(k * (M + 1)) * P + p];
    }
    for (int e = tid; e < K * P * M; e += 256) {
      const int k = e % K, pm = e / K, p = pm % P, m = pm / P;
      s_phi[e] = c.theta[(size_t)(k * (M + 1) + m + 1) * P + p];
      s_gam[e] = c.gamma[e];
    }
    if (tid < K * M) c.c_delta[(size_t)slot * K * M + tid] = c.delta[tid];
    if (tid < K * 2) c.c_A[(size_t)slot * K * 2 + tid] = c.Aa[tid];
    if (tid < K) {
      c.c_pi[(size_t)slot * K + tid] = dyn->pi[tid];
      c.c_tau[slot + (size_t)c.T * tid] = dyn->tau[tid];
    }
    if (tid == 0) {
      c.c_alpha3[slot] = dyn->alpha3;
      c.c_sigma[slot] = dyn->sigma2;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// k_loglik: calcLikelihood = sum_il dnorm(y_il; mean_il, sqrt(sigma2), log)  and end-of-iteration
// bookkeeping (advance the iteration counter / slot for graph replay).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_loglik(Ctx c, int use_rss_part, int r_stored) {
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
    const double ll = -(double)c.d.n_obs_total * (0.91893853320467274178 + log(sqrt(s2))) - rss / (2.0 * s2);
    dyn->rss = rss;
    dyn->loglik = ll;
    if (c.mask & U_LOGLIK) c.c_loglik[dyn->slot] = ll;
    dyn->iter += 1;
    dyn->slot = (r_stored > 0) ? (dyn->iter % (uint32_t)r_stored) : dyn->iter;
  }
}

// broadcast the current value of blocks a sweep does not update into chain slots [s0, s1)
__global__ void k_fill_slots(double* chain, const double* cur, size_t len, int s0, int s1) {
  const size_t total = len * (size_t)(s1 - s0);
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x)
    chain[(size_t)s0 * len + e] = cur[e % len];
}

// ---- host launchers -------------------------------------------------------------------------
void launch_pair_gram(const Ctx& c, int NTG, int NKS, int KS, hipStream_t st) {
  const size_t lds = (size_t)KS * (c.d.K + c.d.MD) * sizeof(double);
  hipLaunchKernelGGL(k_pair_gram, dim3(NTG, NKS), dim3(256), lds, st, c, NTG, KS);
  const int nthreads = c.d.NT * 256;
  hipLaunchKernelGGL(k_pg_reduce, dim3((nthreads + 255) / 256), dim3(256), 0, st, c, NKS);
}

void launch_factor(const Ctx& c, hipStream_t st) {
  const size_t lds = ((size_t)c.d.P * c.d.P + 2 * c.d.P) * sizeof(double);
  hipLaunchKernelGGL(k_factor, dim3(c.d.A), dim3(256), lds, st, c);
}

void launch_sweep(const Ctx& c, hipStream_t st) {
  const size_t lds = ((size_t)3 * c.d.A * c.d.P + 4 * PMAX + 256 + 64) * sizeof(double);
  hipLaunchKernelGGL(k_sweep, dim3(1), dim3(256), lds, st, c);
}

void launch_loglik(const Ctx& c, int use_rss_part, int r_stored, hipStream_t st) {
  hipLaunchKernelGGL(k_loglik, dim3(1), dim3(256), 0, st, c, use_rss_part, r_stored);
}

void launch_fill_slots(double* chain, const double* cur, size_t len, int s0, int s1, hipStream_t st) {
  if (s1 <= s0 || len == 0) return;
  hipLaunchKernelGGL(k_fill_slots, dim3(256), dim3(256), 0, st, chain, cur, len, s0, s1);
}

}  // namespace bfmmm
