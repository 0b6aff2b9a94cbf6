"""ctypes binding of the CPU oracle (oracle/liboracle.so) -- test infrastructure only.

The product package (bayesfmmm_amd) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB = None

c_double_p = C.POINTER(C.c_double)


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", _ORACLE_DIR, "liboracle.so"])
    return os.path.join(_ORACLE_DIR, "liboracle.so")


class OrcRng(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("chain", C.c_uint32), ("iter", C.c_uint32), ("tt_step", C.c_uint32)]


class OrcData(C.Structure):
    _fields_ = [("n", C.c_int), ("K", C.c_int), ("P", C.c_int), ("M", C.c_int), ("D", C.c_int),
                ("off", C.POINTER(C.c_int64)), ("y", c_double_p), ("B", c_double_p), ("X", c_double_p),
                ("Pmat", c_double_p), ("mv", C.c_int)]


class OrcHyper(C.Structure):
    _fields_ = [("c", C.c_double * 16), ("b", C.c_double), ("nu_1", C.c_double),
                ("alpha1l", C.c_double), ("alpha2l", C.c_double), ("beta1l", C.c_double), ("beta2l", C.c_double),
                ("a_Z_PM", C.c_double), ("a_pi_PM", C.c_double), ("var_alpha3", C.c_double),
                ("var_epsilon1", C.c_double), ("var_epsilon2", C.c_double),
                ("alpha_nu", C.c_double), ("beta_nu", C.c_double), ("alpha_eta", C.c_double),
                ("beta_eta", C.c_double), ("alpha_0", C.c_double), ("beta_0", C.c_double)]


_CHAIN_FIELDS = ["nu", "chi", "Z", "pi", "alpha3", "delta", "A", "sigma", "tau", "gamma", "Phi", "loglik",
                 "eta", "tau_eta", "xi", "gamma_xi", "delta_xi", "A_xi"]


class OrcChain(C.Structure):
    _fields_ = [("T", C.c_int)] + [(f, c_double_p) for f in _CHAIN_FIELDS]


def lib():
    global _LIB
    if _LIB is None:
        path = build_oracle()
        L = C.CDLL(path)
        L.orc_qnorm.restype = C.c_double
        L.orc_qnorm.argtypes = [C.c_double]
        L.orc_pnorm.restype = C.c_double
        L.orc_pnorm.argtypes = [C.c_double]
        L.orc_dtruncnorm_log.restype = C.c_double
        L.orc_dtruncnorm_log.argtypes = [C.c_double] * 5
        L.orc_calcLikelihood.restype = C.c_double
        L.orc_fitted.restype = C.c_double
        L.orc_test_fill.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_double,
                                    C.c_double, C.c_int, c_double_p]
        L.orc_run_sweeps.argtypes = [C.POINTER(OrcData), C.POINTER(OrcHyper), C.c_uint64, C.c_uint32, C.c_int,
                                     C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(OrcChain)]
        L.orc_init_nu_z.argtypes = [C.POINTER(OrcData), C.POINTER(OrcHyper), C.c_uint64, C.c_uint32,
                                    C.POINTER(OrcChain)]
        L.orc_init_theta.argtypes = [C.POINTER(OrcData), C.POINTER(OrcHyper), C.c_uint64, C.c_uint32,
                                     c_double_p, c_double_p, c_double_p, C.POINTER(OrcChain)]
        _LIB = L
    return _LIB


def dp(a):
    return a.ctypes.data_as(c_double_p)


def philox(ctr, key):
    out = (C.c_uint32 * 4)()
    lib().orc_philox4x32_10((C.c_uint32 * 4)(*ctr), (C.c_uint32 * 2)(*key), out)
    return list(out)


def fill(kind, count, seed=1, chain=0, it=0, upd=1, p1=0.0, p2=0.0):
    out = np.empty(count)
    lib().orc_test_fill(seed, chain, it, upd, kind, p1, p2, count, dp(out))
    return out


def bspline_basis(x, internal_knots, degree, boundary_knots):
    """splines2::BSpline(x, internal_knots, degree, boundary_knots).basis(true) -> (n, P) array"""
    x = np.ascontiguousarray(x, dtype=np.float64)
    ik = np.ascontiguousarray(internal_knots, dtype=np.float64)
    bk = np.ascontiguousarray(boundary_knots, dtype=np.float64)
    P = len(ik) + degree + 1
    out = np.zeros((len(x), P))
    rc = lib().orc_bspline_basis(len(x), dp(x), len(ik), dp(ik), degree, dp(bk), dp(out))
    if rc != 0:
        raise ValueError("x outside boundary knots")
    return out


def bspline_df(x, df, degree=3):
    """splines2::BSpline(x, df) as used by the reference's unit tests (src/test-Nu.cpp:15):
    degree 3, df - degree - 1 internal knots at quantiles of x, boundary = range(x)."""
    x = np.asarray(x, dtype=np.float64)
    n_int = df - degree - 1
    probs = np.arange(1, n_int + 1) / (n_int + 1)
    ik = np.quantile(x, probs)
    return bspline_basis(x, ik, degree, [x.min(), x.max()])


def pmat_rw1(P):
    out = np.zeros((P, P))
    lib().orc_pmat_rw1(P, dp(out))
    return out


def tensor_bspline(t, degrees, boundary, internal_knots):
    t = np.asfortranarray(t, dtype=np.float64)
    n_pts, dim = t.shape
    deg = (C.c_int * dim)(*degrees)
    nint = (C.c_int * dim)(*[len(k) for k in internal_knots])
    bk = np.ascontiguousarray(boundary, dtype=np.float64)
    iks = [np.ascontiguousarray(k, dtype=np.float64) for k in internal_knots]
    ptrs = (c_double_p * dim)(*[dp(k) for k in iks])
    P = int(np.prod([len(k) + d + 1 for k, d in zip(internal_knots, degrees)]))
    out = np.zeros((n_pts, P), order="F")
    rc = lib().orc_tensor_bspline(n_pts, dim, dp(t), deg, dp(bk), nint, ptrs, dp(out))
    assert rc == 0
    return out


def get_P(degrees, n_internal):
    dim = len(degrees)
    P = int(np.prod([n + d + 1 for n, d in zip(n_internal, degrees)]))
    out = np.zeros((P, P), order="F")
    lib().orc_get_P(dim, (C.c_int * dim)(*degrees), (C.c_int * dim)(*n_internal), dp(out))
    return out


HYPER_DEFAULTS = dict(b=10.0, nu_1=3.0, alpha1l=1.0, alpha2l=2.0, beta1l=1.0, beta2l=1.0, a_Z_PM=10000.0,
                      a_pi_PM=1000.0, var_alpha3=0.05, var_epsilon1=1.0, var_epsilon2=1.0, alpha_nu=10.0,
                      beta_nu=1.0, alpha_eta=10.0, beta_eta=1.0, alpha_0=1.0, beta_0=1.0)


def make_hyper(K, c=None, **kw):
    h = OrcHyper()
    cv = np.full(K, 10.0) if c is None else np.asarray(c, dtype=np.float64)
    for k in range(K):
        h.c[k] = cv[k]
    vals = dict(HYPER_DEFAULTS)
    vals.update(kw)
    for k, v in vals.items():
        setattr(h, k, float(v))
    return h


class Model:
    """Owns the numpy buffers behind an orc_data."""

    def __init__(self, y_list, B_list, K, M, X=None, mv=False, Pmat=None):
        self.n = len(y_list)
        self.K, self.M = K, M
        self.P = B_list[0].shape[1]
        self.off = np.zeros(self.n + 1, dtype=np.int64)
        self.off[1:] = np.cumsum([len(y) for y in y_list])
        self.y = np.ascontiguousarray(np.concatenate(y_list), dtype=np.float64)
        self.B = np.ascontiguousarray(np.concatenate(B_list, axis=0), dtype=np.float64)
        self.X = None if X is None else np.asfortranarray(X, dtype=np.float64)
        self.D = 0 if X is None else self.X.shape[1]
        self.Pmat = np.asfortranarray(pmat_rw1(self.P) if Pmat is None else np.asarray(Pmat, dtype=np.float64))
        self.mv = mv
        self.data = OrcData(self.n, K, self.P, M, self.D, self.off.ctypes.data_as(C.POINTER(C.c_int64)),
                            dp(self.y), dp(self.B), dp(self.X) if self.X is not None else None,
                            dp(self.Pmat), int(mv))


class Chain:
    """Owns chain arrays shaped like the reference's return values (Fortran order)."""

    def __init__(self, model, T):
        n, K, P, M, D = model.n, model.K, model.P, model.M, model.D
        self.T = T
        self.model = model
        F = dict(order="F")
        self.nu = np.zeros((K, P, T), **F)
        self.chi = np.zeros((n, M, T), **F)
        self.Z = np.zeros((n, K, T), **F)
        self.pi = np.zeros((K, T), **F)
        self.alpha3 = np.zeros(T)
        self.delta = np.zeros((K, M, T), **F)
        self.A = np.zeros((K, 2, T), **F)
        self.sigma = np.zeros(T)
        self.tau = np.zeros((T, K), **F)
        self.gamma = np.zeros((K, P, M, T), **F)
        self.Phi = np.zeros((K, P, M, T), **F)
        self.loglik = np.zeros(T)
        if D > 0:
            self.eta = np.zeros((P, D, K, T), **F)
            self.tau_eta = np.zeros((K, D, T), **F)
            self.xi = np.zeros((P, D, M, K, T), **F)
            self.gamma_xi = np.zeros((P, D, M, K, T), **F)
            self.delta_xi = np.zeros((K, M, D, T), **F)
            self.A_xi = np.zeros((K, 2, D, T), **F)
        self.c = OrcChain()
        self.c.T = T
        for f in _CHAIN_FIELDS:
            arr = getattr(self, f, None)
            setattr(self.c, f, dp(arr) if arr is not None else None)

    def set_slot0(self, **kw):
        for k, v in kw.items():
            arr = getattr(self, k)
            if arr.ndim == 1:
                arr[0] = v
            elif k == "tau":
                arr[0, :] = v
            else:
                arr[..., 0] = v


def set_row_window(on):
    """oracle/updates.c: loops over a basis row restricted to its non-zero window (bit-identical; full-size parity tests only)"""
    lib().orc_set_row_window(int(bool(on)))


def run_sweeps(model, hyper, chain, sweep, n_iter=None, seed=1, chain_id=0, covariance_adj=False, first_iter=0):
    n_iter = chain.T if n_iter is None else n_iter
    lib().orc_run_sweeps(C.byref(model.data), C.byref(hyper), seed, chain_id, sweep, int(covariance_adj),
                         chain.T, first_iter, n_iter, C.byref(chain.c))


SWEEP_NU_Z, SWEEP_THETA, SWEEP_WARM = 0, 1, 2


def run_warm_gram(model, hyper, chain, n_iter=None, seed=1, chain_id=0, first_iter=0, sweep=SWEEP_WARM, prepared=None):
    """The warm-start sweep (or, sweep=SWEEP_NU_Z, the Nu_Z sweep) in sufficient-statistics form on the CPU (oracle/gram.c);
    returns the seconds spent in the sweeps (the one-off G_i, s_i, yy_i pass is not included).  `prepared`: a handle from
    gram_prepare(model) to share that pass between calls (the caller frees it with gram_free)."""
    import time
    n_iter = chain.T if n_iter is None else n_iter
    L = lib()
    _gram_sigs(L)
    g = L.orc_gram_prepare(C.addressof(model.data)) if prepared is None else prepared
    fn = L.orc_gram_run_warm if sweep == SWEEP_WARM else L.orc_gram_run_nu_z
    t0 = time.perf_counter()
    fn(C.addressof(model.data), g, C.addressof(hyper), seed, chain_id, chain.T, first_iter, n_iter, C.addressof(chain.c))
    dt = time.perf_counter() - t0
    if prepared is None:
        L.orc_gram_free(g)
    return dt


def _gram_sigs(L):
    L.orc_gram_prepare.restype = C.c_void_p
    L.orc_gram_prepare.argtypes = [C.c_void_p]
    L.orc_gram_free.argtypes = [C.c_void_p]
    for fn in (L.orc_gram_run_warm, L.orc_gram_run_nu_z):
        fn.restype = None
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_void_p]


def gram_prepare(model):
    L = lib()
    _gram_sigs(L)
    return L.orc_gram_prepare(C.addressof(model.data))


def gram_free(g):
    L = lib()
    _gram_sigs(L)
    L.orc_gram_free(g)


def run_warm_tt(model, hyper, chain, N_t, n_temp_trans, beta_N_t, n_iter=None, seed=1, chain_id=0, first_iter=0,
                covariance_adj=False):
    """BFMMM_MTT_warm_start with tempered transitions (BFMMM.h:1502-1672); returns (logA, accepted) per iteration."""
    n_iter = chain.T if n_iter is None else n_iter
    logA = np.full(chain.T, np.nan)
    acc = np.full(chain.T, -1, dtype=np.int32)
    f = lib().orc_run_warm_tt_cov
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
                  C.c_int, C.c_void_p, np.ctypeslib.ndpointer(np.float64), np.ctypeslib.ndpointer(np.int32)]
    f(C.addressof(model.data), C.addressof(hyper), seed, chain_id, chain.T, first_iter, n_iter, N_t, n_temp_trans,
      float(beta_N_t), int(covariance_adj), C.addressof(chain.c), logA, acc)
    return logA, acc


def beta_ladder(N_t, beta_N_t):
    out = np.zeros(N_t)
    f = lib().orc_beta_ladder
    f.restype = None
    f.argtypes = [C.c_int, C.c_double, np.ctypeslib.ndpointer(np.float64)]
    f(N_t, float(beta_N_t), out)
    return out


# ---- single-update wrappers (names follow the reference's Update*.h functions) ----
def _rng(seed, chain_id, it, tt_step=0):
    return OrcRng(seed, chain_id, it, tt_step)


def updateZ_PM(model, ch, it, a_Z_PM, beta_i=1.0, seed=1, chain_id=0):
    r = _rng(seed, chain_id, it)
    lib().orc_updateZ_PM(C.byref(model.data), C.byref(r), C.c_double(beta_i), it, ch.T, C.c_double(a_Z_PM),
                         C.byref(ch.c))


def updatePi_PM(model, ch, it, c, a_pi_PM, seed=1, chain_id=0):
    r = _rng(seed, chain_id, it)
    cv = np.ascontiguousarray(c, dtype=np.float64)
    lib().orc_updatePi_PM(C.byref(model.data), C.byref(r), it, ch.T, dp(cv), C.c_double(a_pi_PM), C.byref(ch.c))


def updateAlpha3(model, ch, it, b, var_alpha3, seed=1, chain_id=0):
    r = _rng(seed, chain_id, it)
    lib().orc_updateAlpha3(C.byref(model.data), C.byref(r), it, ch.T, C.c_double(b), C.c_double(var_alpha3),
                           C.byref(ch.c))


def updatePhi(model, ch, it, tilde_tau, beta_i=1.0, seed=1, chain_id=0):
    r = _rng(seed, chain_id, it)
    tt = np.asfortranarray(tilde_tau, dtype=np.float64)
    lib().orc_updatePhi(C.byref(model.data), C.byref(r), C.c_double(beta_i), it, ch.T, dp(tt), C.byref(ch.c))


def updateDelta(model, ch, it, seed=1, chain_id=0):
    r = _rng(seed, chain_id, it)
    lib().orc_updateDelta(C.byref(model.data), C.byref(r), it, ch.T, C.byref(ch.c))


def updateA(model, ch, it, hyper, seed=1, chain_id=0):
    r = _rng(seed, chain_id, it)
    lib().orc_updateA(C.byref(model.data), C.byref(r), it, ch.T, C.byref(hyper), C.byref(ch.c))


def updateGamma(model, ch, it, nu_gamma, seed=1, chain_id=0):
    r = _rng(seed, chain_id, it)
    lib().orc_updateGamma(C.byref(model.data), C.byref(r), it, ch.T, C.c_double(nu_gamma), C.byref(ch.c))


def updateNu(model, ch, it, beta_i=1.0, seed=1, chain_id=0):
    r = _rng(seed, chain_id, it)
    lib().orc_updateNu(C.byref(model.data), C.byref(r), C.c_double(beta_i), it, ch.T, C.byref(ch.c))


def updateTau(model, ch, it, alpha, beta, seed=1, chain_id=0):
    r = _rng(seed, chain_id, it)
    lib().orc_updateTau(C.byref(model.data), C.byref(r), it, ch.T, C.c_double(alpha), C.c_double(beta),
                        C.byref(ch.c))


def updateSigma(model, ch, it, alpha_0, beta_0, beta_i=1.0, tempered=False, seed=1, chain_id=0):
    r = _rng(seed, chain_id, it)
    lib().orc_updateSigma(C.byref(model.data), C.byref(r), C.c_double(beta_i), int(tempered), it, ch.T,
                          C.c_double(alpha_0), C.c_double(beta_0), C.byref(ch.c))


def updateChi(model, ch, it, beta_i=1.0, seed=1, chain_id=0):
    r = _rng(seed, chain_id, it)
    lib().orc_updateChi(C.byref(model.data), C.byref(r), C.c_double(beta_i), it, ch.T, C.byref(ch.c))


def updateEta(model, ch, it, beta_i=1.0, seed=1, chain_id=0):
    r = _rng(seed, chain_id, it)
    lib().orc_updateEta(C.byref(model.data), C.byref(r), C.c_double(beta_i), it, ch.T, C.byref(ch.c))


def updateTauEta(model, ch, it, alpha, beta, seed=1, chain_id=0):
    r = _rng(seed, chain_id, it)
    lib().orc_updateTauEta(C.byref(model.data), C.byref(r), it, ch.T, C.c_double(alpha), C.c_double(beta),
                           C.byref(ch.c))


def updateXi(model, ch, it, tilde_tau_xi, beta_i=1.0, seed=1, chain_id=0):
    r = _rng(seed, chain_id, it)
    tt = np.asfortranarray(tilde_tau_xi, dtype=np.float64)
    lib().orc_updateXi(C.byref(model.data), C.byref(r), C.c_double(beta_i), it, ch.T, dp(tt), C.byref(ch.c))


def updateDeltaXi(model, ch, it, seed=1, chain_id=0):
    r = _rng(seed, chain_id, it)
    lib().orc_updateDeltaXi(C.byref(model.data), C.byref(r), it, ch.T, C.byref(ch.c))


def updateAXi(model, ch, it, hyper, seed=1, chain_id=0):
    r = _rng(seed, chain_id, it)
    lib().orc_updateAXi(C.byref(model.data), C.byref(r), it, ch.T, C.byref(hyper), C.byref(ch.c))


def updateGammaXi(model, ch, it, nu_gamma, seed=1, chain_id=0):
    r = _rng(seed, chain_id, it)
    lib().orc_updateGammaXi(C.byref(model.data), C.byref(r), it, ch.T, C.c_double(nu_gamma), C.byref(ch.c))


def calcLikelihood(model, ch, it):
    return lib().orc_calcLikelihood(C.byref(model.data), it, C.byref(ch.c))


# ---- post-processing (oracle/post.c) over a chain whose slots are the saved draws ----
def post_llik(model, ch):
    L = lib()
    L.orc_post_llik.restype = None
    L.orc_post_llik.argtypes = [C.c_void_p, C.c_void_p, C.c_int, c_double_p]
    out = np.zeros(ch.T)
    L.orc_post_llik(C.addressof(model.data), C.addressof(ch.c), ch.T, dp(out))
    return out


def post_dic(model, ch, burnin_prop):
    L = lib()
    L.orc_post_dic.restype = C.c_double
    L.orc_post_dic.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double]
    return L.orc_post_dic(C.addressof(model.data), C.addressof(ch.c), ch.T, burnin_prop)


def post_aic_bic(model, ch, burnin_prop, has_x, cov_adj):
    L = lib()
    for f in (L.orc_post_aic, L.orc_post_bic):
        f.restype = C.c_double
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_int]
    a = (C.addressof(model.data), C.addressof(ch.c), ch.T, burnin_prop, int(has_x), int(cov_adj))
    return L.orc_post_aic(*a), L.orc_post_bic(*a)


def post_cpo(model, ch, burnin_prop):
    L = lib()
    L.orc_post_cpo.restype = None
    L.orc_post_cpo.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, c_double_p]
    out = np.zeros(model.n)
    L.orc_post_cpo(C.addressof(model.data), C.addressof(ch.c), ch.T, burnin_prop, dp(out))
    return out
