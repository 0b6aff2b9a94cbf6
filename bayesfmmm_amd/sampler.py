"""Thin object wrapper over the C ABI (include/bfmmm.h): one `Sampler` = one `bfmmm_handle`.

All arrays cross the boundary in the reference's layouts (column-major, R/Armadillo order);
numpy arrays returned here are Fortran-ordered with the reference's shapes.
"""
import ctypes as C

import numpy as np

from . import _lib

U_Z, U_PI, U_ALPHA3, U_PHI, U_DELTA, U_A, U_GAMMA, U_NU, U_TAU, U_SIGMA, U_CHI = (1 << i for i in range(11))
U_ETA, U_TAU_ETA, U_XI, U_DELTA_XI, U_A_XI, U_GAMMA_XI = (1 << i for i in range(11, 17))
U_LOGLIK = 1 << 17
COV_MEAN = U_ETA | U_TAU_ETA
COV_XI = U_XI | U_DELTA_XI | U_A_XI | U_GAMMA_XI
SWEEP_NU_Z = U_Z | U_PI | U_ALPHA3 | U_NU | U_TAU | U_SIGMA | U_LOGLIK
SWEEP_THETA = U_PHI | U_DELTA | U_A | U_GAMMA | U_TAU | U_SIGMA | U_CHI | U_LOGLIK
SWEEP_WARM = SWEEP_NU_Z | SWEEP_THETA

MODEL_FUNCTIONAL, MODEL_MULTIVARIATE = 0, 1


def _dp(a):
    return a.ctypes.data_as(_lib.c_double_p)


def default_config(**kw):
    cfg = _lib.BfmmmConfig()
    _lib.load().bfmmm_config_defaults(C.byref(cfg))
    for k, v in kw.items():
        if k == "c":
            for i, x in enumerate(v):
                cfg.c[i] = float(x)
        else:
            setattr(cfg, k, v)
    return cfg


class Sampler:
    def __init__(self, cfg, Y, time=None, internal_knots=None, boundary_knots=None, device=0, basis=None, band=None,
                 penalty=None, penalty_band=None, n_chains=1):
        """Functional model: Y, time are lists of 1-D arrays (one per curve).
        Multivariate model: Y is an (n, P) matrix.
        Functional model over a caller-supplied basis (bfmmm_create_from_basis; the high-dimensional model's tensor-product
        basis): `basis` is a list of n_i x P matrices, `band` the half-bandwidth of B'B, `penalty` the P x P penalty of the
        nu prior and `penalty_band` its half-bandwidth.
        n_chains > 1: a chain batch (bfmmm_create_batch) -- `run` advances all chains in lockstep, `select_chain` picks the
        chain the state / chain accessors address."""
        self.lib = _lib.load()
        self.cfg = cfg
        self.h = C.c_void_p()
        if basis is not None:
            self.offsets = np.zeros(len(Y) + 1, dtype=np.int64)
            self.offsets[1:] = np.cumsum([len(y) for y in Y])
            y = np.ascontiguousarray(np.concatenate([np.asarray(v, dtype=np.float64) for v in Y]))
            B = np.ascontiguousarray(np.concatenate([np.asarray(b, dtype=np.float64) for b in basis], axis=0))
            Pm = np.asfortranarray(penalty, dtype=np.float64)
            cfg.n_funct = len(Y)
            self.P = B.shape[1]
            _lib.check(self.lib.bfmmm_create_from_basis_batch(C.byref(cfg), device, _dp(y), _dp(B),
                                                              self.offsets.ctypes.data_as(_lib.c_int64_p), self.P, int(band),
                                                              _dp(Pm), int(penalty_band), int(n_chains), C.byref(self.h)))
        elif cfg.model == MODEL_FUNCTIONAL:
            self.offsets = np.zeros(len(Y) + 1, dtype=np.int64)
            self.offsets[1:] = np.cumsum([len(y) for y in Y])
            y = np.ascontiguousarray(np.concatenate([np.asarray(v, dtype=np.float64) for v in Y]))
            t = np.ascontiguousarray(np.concatenate([np.asarray(v, dtype=np.float64) for v in time]))
            ik = np.ascontiguousarray(internal_knots, dtype=np.float64)
            bk = np.ascontiguousarray(boundary_knots, dtype=np.float64)
            cfg.n_funct = len(Y)
            cfg.n_internal_knots = len(ik)
            self.P = len(ik) + cfg.basis_degree + 1
            _lib.check(self.lib.bfmmm_create_batch(C.byref(cfg), device, _dp(y), _dp(t),
                                                   self.offsets.ctypes.data_as(_lib.c_int64_p), _dp(ik), _dp(bk),
                                                   int(n_chains), C.byref(self.h)))
        else:
            Ym = np.asfortranarray(Y, dtype=np.float64)
            cfg.n_funct, cfg.P = Ym.shape
            self.P = cfg.P
            self.offsets = None
            _lib.check(self.lib.bfmmm_create_batch(C.byref(cfg), device, _dp(Ym), None, None, None, None, int(n_chains),
                                                   C.byref(self.h)))
        self.n, self.K, self.M, self.T = cfg.n_funct, cfg.K, cfg.n_eigen, cfg.tot_mcmc_iters
        self.D = 0
        self.n_chains = int(n_chains)

    def select_chain(self, q):
        """Chain of the batch that set_state / get_state / init_state / get_chain / debug address."""
        _lib.check(self.lib.bfmmm_select_chain(self.h, int(q)))

    def set_chain_id_stride(self, stride):
        """Chain q of the batch draws from RNG chain id `chain + q * stride` (default 1)."""
        _lib.check(self.lib.bfmmm_set_chain_id_stride(self.h, int(stride)))

    def set_covariates(self, X, covariance_adj=False):
        """X: (n, D) covariate matrix (the `X` argument of the reference's entry points)."""
        Xm = np.asfortranarray(X, dtype=np.float64)
        assert Xm.shape[0] == self.n
        _lib.check(self.lib.bfmmm_set_covariates(self.h, _dp(Xm), Xm.shape[1], int(covariance_adj)))
        self.D = Xm.shape[1]

    def close(self):
        if self.h:
            self.lib.bfmmm_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- shapes of the reference's objects ----
    def _state_shape(self, name):
        n, K, P, M = self.n, self.K, self.P, self.M
        return {"nu": (K, P), "Phi": (K, P, M), "chi": (n, M), "Z": (n, K), "pi": (K,), "alpha_3": (1,),
                "delta": (K, M), "A": (K, 2), "gamma": (K, P, M), "tau": (K,), "sigma_sq": (1,),
                "loglik": (1,), "status": (1,), "stamps": (64,), "wgtrace": (3072,), "ztrace": (3 * 8192,), "zphase": (8 * 8192,), "fct": (8,),
                "eta": (P, self.D, K), "xi": (P, self.D, M, K), "gamma_xi": (P, self.D, M, K),
                "tau_eta": (K, self.D), "delta_xi": (K, M, self.D), "A_xi": (K, 2, self.D)}[name]

    def set_state(self, **kw):
        for name, v in kw.items():
            a = np.asfortranarray(np.asarray(v, dtype=np.float64).reshape(self._state_shape(name), order="F"))
            _lib.check(self.lib.bfmmm_set_state(self.h, name.encode(), _dp(a), a.size))

    def get_state(self, name):
        out = np.zeros(self._state_shape(name), order="F")
        _lib.check(self.lib.bfmmm_get_state(self.h, name.encode(), _dp(out), out.size))
        return out

    def init_state(self, stage, seed, chain=0):
        _lib.check(self.lib.bfmmm_init_state(self.h, stage, seed, chain))

    def run(self, mask, n_iters, first_iter=0, seed=1, chain=0, phi_chi_zero=False, beta=1.0):
        _lib.check(self.lib.bfmmm_run(self.h, mask, first_iter, n_iters, seed, chain, int(phi_chi_zero), beta))

    def prepare_run(self, mask, n_iters, first_iter=0, seed=1, chain=0, phi_chi_zero=False):
        """Captures the HIP graphs `run` with the same arguments replays (set-up only, launches nothing)."""
        _lib.check(self.lib.bfmmm_prepare_run(self.h, mask, first_iter, n_iters, seed, chain, int(phi_chi_zero)))

    def set_slot_base(self, base):
        """Chain iteration i is written to slot i - base (on-disk batches reuse the slots, include/bfmmm.h)."""
        _lib.check(self.lib.bfmmm_set_slot_base(self.h, int(base)))

    def tempered_transition(self, mask, iteration, N_t, beta_N_t, seed=1, chain=0):
        """Tempered-transition block of BFMMM_warm_start (BFMMM.h:1556-1657) for the chain iteration that `run` has just
        produced; returns (log acceptance probability, accepted)."""
        import ctypes as C
        la = C.c_double(0.0)
        acc = C.c_int(0)
        _lib.check(self.lib.bfmmm_tempered_transition(self.h, mask, iteration, N_t, beta_N_t, seed, chain,
                                                      C.byref(la), C.byref(acc)))
        return la.value, bool(acc.value)

    def get_chain(self, name, n_slots=None):
        T = self.T if n_slots is None else n_slots
        shp = {"nu": (self.K, self.P, T), "chi": (self.n, self.M, T), "Z": (self.n, self.K, T), "pi": (self.K, T),
               "alpha_3": (T,), "delta": (self.K, self.M, T), "A": (self.K, 2, T), "sigma_sq": (T,),
               "tau": (T, self.K), "gamma": (self.K, self.P, self.M, T), "Phi": (self.K, self.P, self.M, T),
               "loglik": (T,), "eta": (self.P, self.D, self.K, T), "xi": (self.P, self.D, self.M, self.K, T),
               "gamma_xi": (self.P, self.D, self.M, self.K, T), "tau_eta": (self.K, self.D, T),
               "delta_xi": (self.K, self.M, self.D, T), "A_xi": (self.K, 2, self.D, T)}[name]
        out = np.zeros(shp, order="F")
        _lib.check(self.lib.bfmmm_get_chain(self.h, name.encode(), T, _dp(out), out.size))
        return out

    def get_basis(self):
        n_obs = int(self.offsets[-1])
        out = np.zeros((n_obs, self.P))
        _lib.check(self.lib.bfmmm_get_basis(self.h, _dp(out), out.size))
        return [out[self.offsets[i]:self.offsets[i + 1]] for i in range(self.n)]

    def debug(self, name, capacity=1 << 24):
        out = np.zeros(capacity)
        cnt = C.c_int64()
        _lib.check(self.lib.bfmmm_debug_get(self.h, name.encode(), _dp(out), capacity, C.byref(cnt)))
        return out[:cnt.value].copy()

    def dims(self):
        v = self.debug("dims", 64)
        names = ["n", "K", "P", "M", "BW", "LG", "LREC", "MD", "A", "R", "NT", "n_obs_total", "half_sum"]
        d = {k: int(x) for k, x in zip(names, v)}
        d["YY"] = float(v[13])
        return d

    def set_profile(self, enable):
        _lib.check(self.lib.bfmmm_set_profile(self.h, int(enable)))

    def timing(self, name="total"):
        ms, cnt = C.c_double(), C.c_int64()
        _lib.check(self.lib.bfmmm_get_timing(self.h, name.encode(), C.byref(ms), C.byref(cnt)))
        return ms.value, cnt.value
