"""The reference's R entry points for the functional model, over the C ABI of include/bfmmm_entry.h.

Function names, argument names, defaults and returned names follow
`BFMMM_Nu_Z_multiple_try`, `BFMMM_Theta_est` and `BFMMM_warm_start` of src/UserFunctions.cpp
(:166, :684, :1341; R wrappers R/RcppExports.R:1604, 1791, 2018).  Results are dicts of numpy arrays in
the reference's shapes (Fortran order); lists of per-curve vectors stay Python lists.

Extra keyword arguments that the reference does not have: `seed` (the reference uses R's global
RNG), `device`, `max_concurrent` (multi-try chains run concurrently on one GPU) and, for the
multi-GPU multi-try, `group` (a torch.distributed process group: chains are dealt round-robin to
the ranks, scores are all-gathered and the winning chain is broadcast; see parallel.py).
"""
import ctypes as C

import numpy as np

from . import _lib

c_double_p = _lib.c_double_p
c_int64_p = _lib.c_int64_p


class EntryArgs(C.Structure):
    """Mirror of `bfmmm_entry_args` (include/bfmmm_entry.h)."""
    _fields_ = [("n_funct", C.c_int32), ("y", c_double_p), ("t", c_double_p), ("offsets", c_int64_p),
                ("tot_mcmc_iters", C.c_int32), ("n_try", C.c_int32), ("K", C.c_int32), ("basis_degree", C.c_int32),
                ("n_eigen", C.c_int32), ("n_internal_knots", C.c_int32), ("boundary_knots", c_double_p),
                ("internal_knots", c_double_p), ("c", c_double_p), ("burnin_prop", C.c_double),
                ("b", C.c_double), ("nu_1", C.c_double), ("alpha1l", C.c_double), ("alpha2l", C.c_double),
                ("beta1l", C.c_double), ("beta2l", C.c_double), ("a_Z_PM", C.c_double), ("a_pi_PM", C.c_double),
                ("var_alpha3", C.c_double), ("var_epsilon1", C.c_double), ("var_epsilon2", C.c_double),
                ("alpha_nu", C.c_double), ("beta_nu", C.c_double), ("alpha_eta", C.c_double), ("beta_eta", C.c_double),
                ("alpha_0", C.c_double), ("beta_0", C.c_double), ("thinning_num", C.c_double), ("beta_N_t", C.c_double),
                ("N_t", C.c_int32), ("n_temp_trans", C.c_int32), ("r_stored_iters", C.c_int32),
                ("seed", C.c_uint64), ("device", C.c_int32), ("chain_offset", C.c_int32), ("chain_stride", C.c_int32),
                ("max_concurrent", C.c_int32), ("model", C.c_int32), ("P", C.c_int32),
                ("X", c_double_p), ("D", C.c_int32), ("covariance_adj", C.c_int32), ("dir", C.c_char_p),
                ("dim", C.c_int32), ("basis_degree_hd", C.POINTER(C.c_int32)), ("n_internal_hd", C.POINTER(C.c_int32)),
                ("progress_every", C.c_int32), ("progress_cb", C.c_void_p), ("progress_user", C.c_void_p),
                ("devices", C.POINTER(C.c_int32)), ("n_devices", C.c_int32)]


ENTRY_SYMBOLS = {
    "bfmmm_result_create": (C.c_void_p, []),
    "bfmmm_result_free": (None, [C.c_void_p]),
    "bfmmm_result_set": (C.c_int, [C.c_void_p, C.c_char_p, c_double_p, C.c_int64, c_int64_p, C.c_int]),
    "bfmmm_result_get": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(c_double_p), c_int64_p, C.POINTER(c_int64_p),
                                   C.POINTER(C.c_int)]),
    "bfmmm_result_count": (C.c_int, [C.c_void_p]),
    "bfmmm_result_name": (C.c_char_p, [C.c_void_p, C.c_int]),
    "bfmmm_entry_defaults": (None, [C.POINTER(EntryArgs), C.c_int]),
    "bfmmm_BFMMM_Nu_Z_multiple_try": (C.c_int, [C.POINTER(EntryArgs), C.POINTER(C.c_void_p)]),
    "bfmmm_BFMMM_Theta_est": (C.c_int, [C.POINTER(EntryArgs), C.c_void_p, C.POINTER(C.c_void_p)]),
    "bfmmm_BFMMM_warm_start": (C.c_int, [C.POINTER(EntryArgs), C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "bfmmm_BMVMMM_Nu_Z_multiple_try": (C.c_int, [C.POINTER(EntryArgs), C.POINTER(C.c_void_p)]),
    "bfmmm_BMVMMM_Theta_est": (C.c_int, [C.POINTER(EntryArgs), C.c_void_p, C.POINTER(C.c_void_p)]),
    "bfmmm_BMVMMM_warm_start": (C.c_int, [C.POINTER(EntryArgs), C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "bfmmm_BHDFMMM_Nu_Z_multiple_try": (C.c_int, [C.POINTER(EntryArgs), C.POINTER(C.c_void_p)]),
    "bfmmm_BHDFMMM_Theta_est": (C.c_int, [C.POINTER(EntryArgs), C.c_void_p, C.POINTER(C.c_void_p)]),
    "bfmmm_BHDFMMM_warm_start": (C.c_int, [C.POINTER(EntryArgs), C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "bfmmm_arma_read": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p)]),
    "bfmmm_arma_read_field": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p)]),
    "bfmmm_arma_write_ascii": (C.c_int, [C.c_char_p, c_double_p, c_int64_p, C.c_int]),
    "bfmmm_arma_write_field": (C.c_int, [C.c_char_p, C.c_void_p, C.c_int64, C.c_int64]),
    "bfmmm_tensor_bspline": (C.c_int, [C.c_int, C.c_int, c_double_p, C.POINTER(C.c_int), c_double_p, C.POINTER(C.c_int),
                                       c_double_p, c_double_p]),
    "bfmmm_tensor_penalty": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), c_double_p]),
    "bfmmm_entry_last_error": (C.c_char_p, []),
}

_bound = False


class PostInput(C.Structure):
    """bfmmm_post_input of include/bfmmm_post.h."""
    _fields_ = [("n", C.c_int32), ("K", C.c_int32), ("P", C.c_int32), ("M", C.c_int32), ("D", C.c_int32),
                ("offsets", c_int64_p), ("y", c_double_p), ("B", c_double_p), ("X", c_double_p), ("T", C.c_int32),
                ("nu", c_double_p), ("Phi", c_double_p), ("Z", c_double_p), ("chi", c_double_p), ("sigma", c_double_p),
                ("eta", c_double_p), ("xi", c_double_p), ("device", C.c_int32), ("identity_basis", C.c_int32)]


class PostArgs(C.Structure):
    """bfmmm_post_args of include/bfmmm_post.h."""
    _fields_ = [("dir", C.c_char_p), ("n_files", C.c_int32), ("basis_degree", C.c_int32), ("n_internal_knots", C.c_int32),
                ("boundary_knots", c_double_p), ("internal_knots", c_double_p), ("n_funct", C.c_int32),
                ("t", c_double_p), ("y", c_double_p), ("offsets", c_int64_p), ("burnin_prop", C.c_double),
                ("X", c_double_p), ("D", C.c_int32), ("cov_adj", C.c_int32), ("device", C.c_int32), ("P", C.c_int32)]


class CiArgs(C.Structure):
    """bfmmm_ci_args of include/bfmmm_post.h."""
    _fields_ = [("dir", C.c_char_p), ("n_files", C.c_int32), ("time", c_double_p), ("n_time", C.c_int32),
                ("basis_degree", C.c_int32), ("n_internal_knots", C.c_int32), ("boundary_knots", c_double_p),
                ("internal_knots", c_double_p), ("k", C.c_int32), ("alpha", C.c_double), ("rescale", C.c_int32),
                ("simultaneous", C.c_int32), ("burnin_prop", C.c_double), ("X", c_double_p), ("n_x", C.c_int32), ("D", C.c_int32),
                ("trans_mats", c_double_p), ("device", C.c_int32), ("time2", c_double_p), ("n_time2", C.c_int32),
                ("l", C.c_int32), ("m", C.c_int32), ("dim", C.c_int32), ("basis_degree_hd", C.POINTER(C.c_int32)),
                ("n_internal_hd", C.POINTER(C.c_int32))]


POST_SYMBOLS = {
    "bfmmm_post_col_quantiles": (C.c_int, [c_double_p, C.c_int32, C.c_int32, c_double_p, C.c_int32, C.c_int32, c_double_p]),
    "bfmmm_post_bands": (C.c_int, [c_double_p, C.c_int32, C.c_int32, c_double_p, C.c_int32, C.c_double, C.c_int32, C.c_int32,
                                   c_double_p, c_double_p, c_double_p, c_double_p]),
    "bfmmm_post_cov_bands": (C.c_int, [c_double_p, c_double_p, C.c_int32, C.c_int32, C.c_int32, c_double_p, C.c_int32, c_double_p, C.c_int32,
                                       C.c_double, C.c_int32, C.c_int32, c_double_p, c_double_p, c_double_p, c_double_p]),
    "bfmmm_FCovCI": (C.c_int, [C.POINTER(CiArgs), C.POINTER(C.c_void_p)]),
    "bfmmm_HDFCovCI": (C.c_int, [C.POINTER(CiArgs), C.POINTER(C.c_void_p)]),
    "bfmmm_MVCovCI": (C.c_int, [C.POINTER(CiArgs), C.POINTER(C.c_void_p)]),
    "bfmmm_MVMeanCI": (C.c_int, [C.POINTER(CiArgs), C.POINTER(C.c_void_p)]),
    "bfmmm_HDFMeanCI": (C.c_int, [C.POINTER(CiArgs), C.POINTER(C.c_void_p)]),
    "bfmmm_ci_defaults": (None, [C.POINTER(CiArgs)]),
    "bfmmm_SigmaCI": (C.c_int, [C.POINTER(CiArgs), C.POINTER(C.c_void_p)]),
    "bfmmm_ZCI": (C.c_int, [C.POINTER(CiArgs), C.POINTER(C.c_void_p)]),
    "bfmmm_FMeanCI": (C.c_int, [C.POINTER(CiArgs), C.POINTER(C.c_void_p)]),
    "bfmmm_post_pointwise": (C.c_int, [C.POINTER(PostInput), C.c_int32, c_double_p, c_double_p, c_double_p]),
    "bfmmm_post_pointwise_joint": (C.c_int, [C.POINTER(PostInput), C.c_int32, c_double_p, c_double_p, c_double_p]),
    "bfmmm_post_cpo": (C.c_int, [C.POINTER(PostInput), C.c_int32, c_double_p]),
    "bfmmm_ConditionalPredictiveOrdinates": (C.c_int, [C.POINTER(PostArgs), C.c_int32, C.POINTER(C.c_void_p)]),
    "bfmmm_post_last_kernel_ms": (C.c_double, []),
    "bfmmm_FSamplePaths": (C.c_int, [C.POINTER(PostArgs), C.c_double, C.c_int32, C.c_uint64, C.POINTER(C.c_void_p)]),
    "bfmmm_post_sample_paths": (C.c_int, [C.POINTER(PostInput), C.c_int32, C.c_uint64, c_double_p, c_double_p]),
    "bfmmm_post_table_bands": (C.c_int, [c_double_p, C.c_int32, C.c_int32, C.c_double, C.c_int32, C.c_int32, c_double_p, c_double_p, c_double_p]),
    "bfmmm_MVLLik": (C.c_int, [C.POINTER(PostArgs), C.POINTER(C.c_void_p)]),
    "bfmmm_MVDIC": (C.c_int, [C.POINTER(PostArgs), c_double_p]),
    "bfmmm_MVAIC": (C.c_int, [C.POINTER(PostArgs), c_double_p]),
    "bfmmm_MVBIC": (C.c_int, [C.POINTER(PostArgs), c_double_p]),
    "bfmmm_post_defaults": (None, [C.POINTER(PostArgs)]),
    "bfmmm_FLLik": (C.c_int, [C.POINTER(PostArgs), C.POINTER(C.c_void_p)]),
    "bfmmm_FDIC": (C.c_int, [C.POINTER(PostArgs), c_double_p]),
    "bfmmm_FAIC": (C.c_int, [C.POINTER(PostArgs), c_double_p]),
    "bfmmm_FBIC": (C.c_int, [C.POINTER(PostArgs), c_double_p]),
}


def _lib_entry():
    global _bound
    lib = _lib.load()
    if not _bound:
        for name, (res, args) in list(ENTRY_SYMBOLS.items()) + list(POST_SYMBOLS.items()):
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _bound = True
    return lib


def _check(rc):
    if rc != 0:
        raise _lib.BfmmmError(_lib_entry().bfmmm_entry_last_error().decode())


def _set_kw(owner, a, kw):
    """remaining keyword arguments are fields of bfmmm_entry_args; `devices` (a list of GPU indices) selects the
    multi-GPU multi-try of include/bfmmm_entry.h"""
    devices = kw.pop("devices", None)
    if devices is not None:
        owner.devices = (C.c_int32 * len(devices))(*[int(x) for x in devices])
        a.devices, a.n_devices = owner.devices, len(devices)
    for k, v in kw.items():
        if not hasattr(a, k):
            raise TypeError(f"unexpected argument '{k}'")
        setattr(a, k, v)


class _ArgsMV:
    """bfmmm_entry_args of the multivariate entry points: Y is the n x P matrix."""

    def __init__(self, entry, tot_mcmc_iters, K, Y, n_eigen, X, kw):
        lib = _lib_entry()
        self.a = EntryArgs()
        lib.bfmmm_entry_defaults(C.byref(self.a), entry)
        self.Y = np.asfortranarray(Y, dtype=np.float64)
        n, P = self.Y.shape
        a = self.a
        a.n_funct, a.P, a.tot_mcmc_iters, a.K, a.n_eigen = n, P, tot_mcmc_iters, K, n_eigen
        a.y = self.Y.ctypes.data_as(c_double_p)
        self.P, self.offsets = P, None
        if X is not None:        # UserFunctions.cpp:4582 / :5000 / :5545 (`X`, `covariance_adj`)
            self.X = np.asfortranarray(X, dtype=np.float64)
            if self.X.shape[0] != n:
                raise _lib.BfmmmError("'X' must be have 'n_funct' number of rows")
            a.X, a.D = self.X.ctypes.data_as(c_double_p), self.X.shape[1]
            a.covariance_adj = int(bool(kw.pop("covariance_adj", False)))
        else:
            kw.pop("covariance_adj", None)
        c = kw.pop("c", None)
        if c is not None:
            self.c = np.ascontiguousarray(c, dtype=np.float64)
            if self.c.size != K:
                raise _lib.BfmmmError("number of elements of the vector 'c' must be equal to K")
            a.c = self.c.ctypes.data_as(c_double_p)
        _set_kw(self, a, kw)


def _result_to_dict(lib, res, offsets, P):
    out = {}
    for i in range(lib.bfmmm_result_count(res)):
        name = lib.bfmmm_result_name(res, i)
        data, cnt, dims, nd = c_double_p(), C.c_int64(), c_int64_p(), C.c_int()
        _check(lib.bfmmm_result_get(res, name, C.byref(data), C.byref(cnt), C.byref(dims), C.byref(nd)))
        shape = tuple(dims[k] for k in range(nd.value))
        arr = np.ctypeslib.as_array(data, shape=(cnt.value,)).copy() if cnt.value > 0 else np.zeros(0)
        key = name.decode()
        if key in ("B", "B_obs") and offsets is not None:
            rows = arr.reshape(-1, P)
            out[key] = [rows[offsets[j]:offsets[j + 1]].copy() for j in range(len(offsets) - 1)]
        elif key in ("best_chain", "best_score"):
            out[key] = float(arr[0])
        else:
            out[key] = arr.reshape(shape, order="F")
    return out


def _dict_to_result(lib, d):
    res = C.c_void_p(lib.bfmmm_result_create())
    for key, val in d.items():
        if key in ("B", "B_obs", "best_chain", "best_score"):
            continue
        a = np.asfortranarray(np.asarray(val, dtype=np.float64))
        flat = np.ascontiguousarray(a.reshape(-1, order="F"))
        dims = (C.c_int64 * a.ndim)(*a.shape)
        _check(lib.bfmmm_result_set(res, key.encode(), flat.ctypes.data_as(c_double_p), flat.size, dims, a.ndim))
    return res


class _Args:
    """Owns the numpy buffers behind a bfmmm_entry_args."""

    def __init__(self, entry, tot_mcmc_iters, K, Y, time, n_funct, basis_degree, n_eigen, boundary_knots,
                 internal_knots, X, kw):
        if len(Y) != n_funct or len(time) != n_funct:
            raise ValueError("'Y' and 'time' must have 'n_funct' elements")
        lib = _lib_entry()
        self.a = EntryArgs()
        lib.bfmmm_entry_defaults(C.byref(self.a), entry)
        self.offsets = np.zeros(n_funct + 1, dtype=np.int64)
        self.offsets[1:] = np.cumsum([len(v) for v in Y])
        self.y = np.ascontiguousarray(np.concatenate([np.asarray(v, dtype=np.float64) for v in Y]))
        self.t = np.ascontiguousarray(np.concatenate([np.asarray(v, dtype=np.float64) for v in time]))
        self.bk = np.ascontiguousarray(boundary_knots, dtype=np.float64)
        self.ik = np.ascontiguousarray(internal_knots, dtype=np.float64)
        a = self.a
        a.n_funct, a.tot_mcmc_iters, a.K, a.basis_degree, a.n_eigen = n_funct, tot_mcmc_iters, K, basis_degree, n_eigen
        a.n_internal_knots = len(self.ik)
        a.y, a.t = self.y.ctypes.data_as(c_double_p), self.t.ctypes.data_as(c_double_p)
        a.offsets = self.offsets.ctypes.data_as(c_int64_p)
        a.boundary_knots, a.internal_knots = self.bk.ctypes.data_as(c_double_p), self.ik.ctypes.data_as(c_double_p)
        self.P = len(self.ik) + basis_degree + 1
        if X is not None:
            self.X = np.asfortranarray(X, dtype=np.float64)
            if self.X.shape[0] != n_funct:
                raise _lib.BfmmmError("'X' must be have 'n_funct' number of rows")
            a.X, a.D = self.X.ctypes.data_as(c_double_p), self.X.shape[1]
            a.covariance_adj = int(bool(kw.pop("covariance_adj", False)))
        else:
            kw.pop("covariance_adj", None)
        c = kw.pop("c", None)
        if c is not None:
            self.c = np.ascontiguousarray(c, dtype=np.float64)
            if self.c.size != K:
                raise _lib.BfmmmError("number of elements of the vector 'c' must be equal to K")
            a.c = self.c.ctypes.data_as(c_double_p)
        _set_kw(self, a, kw)


def BFMMM_Nu_Z_multiple_try(tot_mcmc_iters, n_try, K, Y, time, n_funct, basis_degree, n_eigen, boundary_knots,
                            internal_knots, X=None, group=None, **kw):
    """src/UserFunctions.cpp:166.  Runs 1 + n_try independent chains of the (Z, pi, alpha_3, nu, tau, sigma^2)
    sweep and returns the chain with the largest mean log-likelihood over its last 99 draws."""
    lib = _lib_entry()
    args = _Args(0, tot_mcmc_iters, K, Y, time, n_funct, basis_degree, n_eigen, boundary_knots, internal_knots, X, kw)
    args.a.n_try = n_try
    if group is not None:
        from . import parallel
        return parallel.multi_try(lambda: _call1(lib.bfmmm_BFMMM_Nu_Z_multiple_try, args), args, group)
    return _call1(lib.bfmmm_BFMMM_Nu_Z_multiple_try, args)


PROGRESS_CB = C.CFUNCTYPE(C.c_int, C.c_int32, C.c_double, C.c_void_p)


def set_progress(args, every, fn):
    """Attach a progress callback fn(iteration, loglik) -> truthy to abort (warm start; include/bfmmm_entry.h)."""
    args._cb = PROGRESS_CB(lambda it, ll, user: 1 if fn(it, ll) else 0)
    args.a.progress_every = int(every)
    args.a.progress_cb = C.cast(args._cb, C.c_void_p)


def _call1(fn, args, *prev):
    lib = _lib_entry()
    res = C.c_void_p()
    handles = [_dict_to_result(lib, p) for p in prev]
    try:
        _check(fn(C.byref(args.a), *handles, C.byref(res)))
        return _result_to_dict(lib, res, args.offsets, args.P)
    finally:
        for h in handles:
            lib.bfmmm_result_free(h)
        if res:
            lib.bfmmm_result_free(res)


def BFMMM_Theta_est(tot_mcmc_iters, n_try, K, Y, time, n_funct, basis_degree, n_eigen, boundary_knots, internal_knots,
                    multiple_try, X=None, burnin_prop=0.8, group=None, **kw):
    """src/UserFunctions.cpp:684.  Conditions on the posterior medians of Z and nu from stage 1 and samples
    (Phi, delta, A, gamma, tau, sigma^2, chi); best of 1 + n_try chains."""
    lib = _lib_entry()
    args = _Args(1, tot_mcmc_iters, K, Y, time, n_funct, basis_degree, n_eigen, boundary_knots, internal_knots, X, kw)
    args.a.n_try, args.a.burnin_prop = n_try, burnin_prop
    mt = {k: multiple_try[k] for k in ("Z", "nu") + (("eta",) if X is not None else ())}
    if group is not None:
        from . import parallel
        return parallel.multi_try(lambda: _call1(lib.bfmmm_BFMMM_Theta_est, args, mt), args, group)
    return _call1(lib.bfmmm_BFMMM_Theta_est, args, mt)


def BFMMM_warm_start(tot_mcmc_iters, K, Y, time, n_funct, basis_degree, n_eigen, boundary_knots, internal_knots,
                     multiple_try, theta_est, X=None, burnin_prop=0.8, dir=None, **kw):
    """src/UserFunctions.cpp:1341.  Full sampler started at the posterior medians of stages 1 and 2."""
    lib = _lib_entry()
    progress = kw.pop("progress", None)          # (every, fn(iteration, loglik) -> truthy to abort)
    args = _Args(2, tot_mcmc_iters, K, Y, time, n_funct, basis_degree, n_eigen, boundary_knots, internal_knots, X, kw)
    if progress is not None:
        set_progress(args, *progress)
    args.a.burnin_prop = burnin_prop
    if dir is not None:      # <dir>Nu<q>.txt, ... are written as the reference does (string concatenation: end it with "/")
        args.dir = str(dir).encode()
        args.a.dir = args.dir
    mt = {k: multiple_try[k] for k in ("Z", "nu", "pi", "alpha_3", "tau") + (("eta", "tau_eta") if X is not None else ())}
    te_names = ("delta", "gamma", "Phi", "A", "sigma_sq", "chi")
    if X is not None and args.a.covariance_adj:
        te_names += ("xi", "gamma_xi", "delta_xi", "A_xi")
    te = {k: theta_est[k] for k in te_names}
    return _call1(lib.bfmmm_BFMMM_warm_start, args, mt, te)


def BMVMMM_Nu_Z_multiple_try(tot_mcmc_iters, n_try, K, Y, n_eigen, X=None, group=None, **kw):
    """src/UserFunctions.cpp:4579 (multivariate model: Y is an n x P matrix, no basis)."""
    lib = _lib_entry()
    args = _ArgsMV(3, tot_mcmc_iters, K, Y, n_eigen, X, kw)
    args.a.n_try = n_try
    if group is not None:
        from . import parallel
        return parallel.multi_try(lambda: _call1(lib.bfmmm_BMVMMM_Nu_Z_multiple_try, args), args, group)
    return _call1(lib.bfmmm_BMVMMM_Nu_Z_multiple_try, args)


def BMVMMM_Theta_est(tot_mcmc_iters, n_try, K, Y, n_eigen, multiple_try, X=None, burnin_prop=0.8, group=None, **kw):
    """src/UserFunctions.cpp:4995."""
    lib = _lib_entry()
    args = _ArgsMV(4, tot_mcmc_iters, K, Y, n_eigen, X, kw)
    args.a.n_try, args.a.burnin_prop = n_try, burnin_prop
    mt = {k: multiple_try[k] for k in ("Z", "nu") + (("eta",) if X is not None else ())}
    if group is not None:
        from . import parallel
        return parallel.multi_try(lambda: _call1(lib.bfmmm_BMVMMM_Theta_est, args, mt), args, group)
    return _call1(lib.bfmmm_BMVMMM_Theta_est, args, mt)


def BMVMMM_warm_start(tot_mcmc_iters, K, Y, n_eigen, multiple_try, theta_est, X=None, burnin_prop=0.8, dir=None, **kw):
    """src/UserFunctions.cpp:5540."""
    lib = _lib_entry()
    args = _ArgsMV(5, tot_mcmc_iters, K, Y, n_eigen, X, kw)
    args.a.burnin_prop = burnin_prop
    if dir is not None:
        args.dir = str(dir).encode()
        args.a.dir = args.dir
    mt = {k: multiple_try[k] for k in ("Z", "nu", "pi", "alpha_3", "tau") + (("eta", "tau_eta") if X is not None else ())}
    te_names = ("delta", "gamma", "Phi", "A", "sigma_sq", "chi")
    if X is not None and args.a.covariance_adj:
        te_names += ("xi", "gamma_xi", "delta_xi", "A_xi")
    te = {k: theta_est[k] for k in te_names}
    return _call1(lib.bfmmm_BMVMMM_warm_start, args, mt, te)


# ---- readers / writers of the on-disk chain batches (src/UserFunctions.cpp:2157-2357) ---------------------------
def _read(fn, file):
    lib = _lib_entry()
    res = C.c_void_p()
    try:
        _check(fn(str(file).encode(), C.byref(res)))
        return _result_to_dict(lib, res, None, 0)
    finally:
        if res:
            lib.bfmmm_result_free(res)


def ReadVec(file):
    """src/UserFunctions.cpp:2158 (arma::vec::load): 1-d array."""
    return _read(_lib_entry().bfmmm_arma_read, file)["value"].reshape(-1, order="F")


def ReadMat(file):
    """src/UserFunctions.cpp:2205."""
    return _read(_lib_entry().bfmmm_arma_read, file)["value"]


def ReadCube(file):
    """src/UserFunctions.cpp:2253."""
    return _read(_lib_entry().bfmmm_arma_read, file)["value"]


def _read_field(file):
    d = _read(_lib_entry().bfmmm_arma_read_field, file)
    nr, nc = (int(x) for x in d["field_dims"])
    out = np.empty((nr, nc), dtype=object)
    for e in range(nr * nc):
        out[e % nr, e // nr] = d[str(e)]
    return out


def ReadFieldCube(file):
    """src/UserFunctions.cpp:2303: object array (n_rows, n_cols) of cubes."""
    return _read_field(file)


def ReadFieldMat(file):
    """src/UserFunctions.cpp:2351."""
    return _read_field(file)


def ReadFieldVec(file):
    """src/UserFunctions.cpp:2399: object array of 1-d arrays."""
    f = _read_field(file)
    for idx in np.ndindex(f.shape):
        f[idx] = f[idx].reshape(-1, order="F")
    return f


def write_arma_ascii(file, x):
    """`x.save(file, arma::arma_ascii)` for a vector (saved as an n x 1 matrix), matrix or cube."""
    a = np.asfortranarray(np.asarray(x, dtype=np.float64))
    if a.ndim == 1:
        a = a.reshape(-1, 1, order="F")
    flat = np.ascontiguousarray(a.reshape(-1, order="F"))
    dims = (C.c_int64 * a.ndim)(*a.shape)
    _check(_lib_entry().bfmmm_arma_write_ascii(str(file).encode(), flat.ctypes.data_as(c_double_p), dims, a.ndim))


def write_arma_field(file, field):
    """`field.save(file)` (arma_binary) for an object array (n_rows, n_cols) of matrices / cubes."""
    lib = _lib_entry()
    field = np.asarray(field, dtype=object)
    if field.ndim == 1:
        field = field.reshape(-1, 1)
    nr, nc = field.shape
    items = {}
    for e in range(nr * nc):
        v = np.asarray(field[e % nr, e // nr], dtype=np.float64)
        items[str(e)] = v.reshape(-1, 1) if v.ndim == 1 else v
    res = _dict_to_result(lib, items)
    try:
        _check(lib.bfmmm_arma_write_field(str(file).encode(), res, nr, nc))
    finally:
        lib.bfmmm_result_free(res)


# ---- set-up pieces of the high-dimensional functional model (BSplines.h:18-120) -----------------------------------
def TensorBSpline(t, basis_degree, boundary_knots, internal_knots):
    """Tensor-product B-spline basis of the points t (n_pts x dim): n_pts x prod_l (len(internal_knots[l]) + degree[l] + 1)."""
    t = np.asfortranarray(np.asarray(t, dtype=np.float64))
    n_pts, dim = t.shape
    deg = (C.c_int * dim)(*[int(x) for x in basis_degree])
    nint = (C.c_int * dim)(*[len(k) for k in internal_knots])
    bk = np.ascontiguousarray(np.asarray(boundary_knots, dtype=np.float64).reshape(dim, 2))
    ik = np.ascontiguousarray(np.concatenate([np.asarray(k, dtype=np.float64).reshape(-1) for k in internal_knots]) if dim else [])
    P = int(np.prod([len(k) + int(g) + 1 for k, g in zip(internal_knots, basis_degree)]))
    out = np.zeros((n_pts, P), order="F")
    _check(_lib_entry().bfmmm_tensor_bspline(n_pts, dim, t.ctypes.data_as(c_double_p), deg, bk.ctypes.data_as(c_double_p), nint,
                                             ik.ctypes.data_as(c_double_p), out.ctypes.data_as(c_double_p)))
    return out


def GetP(basis_degree, n_internal_knots):
    """Penalty matrix of the tensor-product basis (BSplines.h:74-120)."""
    dim = len(basis_degree)
    deg = (C.c_int * dim)(*[int(x) for x in basis_degree])
    nint = (C.c_int * dim)(*[int(x) for x in n_internal_knots])
    P = int(np.prod([int(k) + int(g) + 1 for k, g in zip(n_internal_knots, basis_degree)]))
    out = np.zeros((P, P), order="F")
    _check(_lib_entry().bfmmm_tensor_penalty(dim, deg, nint, out.ctypes.data_as(c_double_p)))
    return out


# ---- high-dimensional functional model (src/UserFunctions.cpp:2519, :3030, :3676) --------------------------------
class _ArgsHD:
    """bfmmm_entry_args of the BHDFMMM_* entry points: `time` is a list of n_i x dim matrices, `basis_degree` a vector,
    `boundary_knots` a dim x 2 matrix, `internal_knots` a list of vectors."""

    def __init__(self, entry, tot_mcmc_iters, K, Y, time, n_funct, basis_degree, n_eigen, boundary_knots, internal_knots, X, kw):
        if len(Y) != n_funct or len(time) != n_funct:
            raise ValueError("'Y' and 'time' must have 'n_funct' elements")
        lib = _lib_entry()
        self.a = EntryArgs()
        lib.bfmmm_entry_defaults(C.byref(self.a), entry)
        dim = len(basis_degree)
        tm = [np.asfortranarray(np.asarray(t, dtype=np.float64).reshape(len(y), -1)) for t, y in zip(time, Y)]
        if tm[0].shape[1] != dim:
            raise _lib.BfmmmError("number of elemnts in 'basis_degree' does not match number of columns in time matrix")
        self.offsets = np.zeros(n_funct + 1, dtype=np.int64)
        self.offsets[1:] = np.cumsum([len(v) for v in Y])
        self.y = np.ascontiguousarray(np.concatenate([np.asarray(v, dtype=np.float64) for v in Y]))
        self.t = np.ascontiguousarray(np.concatenate([t.reshape(-1, order="F") for t in tm]))
        self.bk = np.ascontiguousarray(np.asarray(boundary_knots, dtype=np.float64).reshape(dim, 2))
        self.ik = np.ascontiguousarray(np.concatenate([np.asarray(k, dtype=np.float64).reshape(-1) for k in internal_knots]))
        self.deg = (C.c_int32 * dim)(*[int(x) for x in basis_degree])
        self.nint = (C.c_int32 * dim)(*[len(k) for k in internal_knots])
        a = self.a
        a.n_funct, a.tot_mcmc_iters, a.K, a.n_eigen, a.dim = n_funct, tot_mcmc_iters, K, n_eigen, dim
        a.basis_degree = max(int(x) for x in basis_degree)
        a.y, a.t = self.y.ctypes.data_as(c_double_p), self.t.ctypes.data_as(c_double_p)
        a.offsets = self.offsets.ctypes.data_as(c_int64_p)
        a.boundary_knots, a.internal_knots = self.bk.ctypes.data_as(c_double_p), self.ik.ctypes.data_as(c_double_p)
        a.basis_degree_hd, a.n_internal_hd = self.deg, self.nint
        self.P = int(np.prod([len(k) + int(g) + 1 for k, g in zip(internal_knots, basis_degree)]))
        if X is not None:        # UserFunctions.cpp:2531 / :3043 / :3689 (`X`, `covariance_adj`)
            self.X = np.asfortranarray(X, dtype=np.float64)
            if self.X.shape[0] != n_funct:
                raise _lib.BfmmmError("'X' must be have 'n_funct' number of rows")
            a.X, a.D = self.X.ctypes.data_as(c_double_p), self.X.shape[1]
            a.covariance_adj = int(bool(kw.pop("covariance_adj", False)))
        else:
            kw.pop("covariance_adj", None)
        c = kw.pop("c", None)
        if c is not None:
            self.c = np.ascontiguousarray(c, dtype=np.float64)
            if self.c.size != K:
                raise _lib.BfmmmError("number of elements of the vector 'c' must be equal to K")
            a.c = self.c.ctypes.data_as(c_double_p)
        _set_kw(self, a, kw)


def BHDFMMM_Nu_Z_multiple_try(tot_mcmc_iters, n_try, K, Y, time, n_funct, basis_degree, n_eigen, boundary_knots,
                              internal_knots, X=None, **kw):
    """src/UserFunctions.cpp:2519."""
    lib = _lib_entry()
    args = _ArgsHD(0, tot_mcmc_iters, K, Y, time, n_funct, basis_degree, n_eigen, boundary_knots, internal_knots, X, kw)
    args.a.n_try = n_try
    return _call1(lib.bfmmm_BHDFMMM_Nu_Z_multiple_try, args)


def BHDFMMM_Theta_est(tot_mcmc_iters, n_try, K, Y, time, n_funct, basis_degree, n_eigen, boundary_knots, internal_knots,
                      multiple_try, X=None, burnin_prop=0.8, **kw):
    """src/UserFunctions.cpp:3030."""
    lib = _lib_entry()
    args = _ArgsHD(1, tot_mcmc_iters, K, Y, time, n_funct, basis_degree, n_eigen, boundary_knots, internal_knots, X, kw)
    args.a.n_try, args.a.burnin_prop = n_try, burnin_prop
    mt = {k: multiple_try[k] for k in ("Z", "nu") + (("eta",) if X is not None else ())}
    return _call1(lib.bfmmm_BHDFMMM_Theta_est, args, mt)


def BHDFMMM_warm_start(tot_mcmc_iters, K, Y, time, n_funct, basis_degree, n_eigen, boundary_knots, internal_knots,
                       multiple_try, theta_est, X=None, burnin_prop=0.8, dir=None, **kw):
    """src/UserFunctions.cpp:3676."""
    lib = _lib_entry()
    args = _ArgsHD(2, tot_mcmc_iters, K, Y, time, n_funct, basis_degree, n_eigen, boundary_knots, internal_knots, X, kw)
    args.a.burnin_prop = burnin_prop
    if dir is not None:
        args.dir = str(dir).encode()
        args.a.dir = args.dir
    mt = {k: multiple_try[k] for k in ("Z", "nu", "pi", "alpha_3", "tau") + (("eta", "tau_eta") if X is not None else ())}
    te_names = ("delta", "gamma", "Phi", "A", "sigma_sq", "chi")
    if X is not None and args.a.covariance_adj:
        te_names += ("xi", "gamma_xi", "delta_xi", "A_xi")
    return _call1(lib.bfmmm_BHDFMMM_warm_start, args, mt, {k: theta_est[k] for k in te_names})


# ---- likelihood-based post-processing (src/PostProcessing.cpp:3660-5114; include/bfmmm_post.h) ----------------------
def _post_input(Y, B, nu, Phi, Z, chi, sigma, X=None, eta=None, xi=None, device=0):
    """bfmmm_post_input over in-memory draws in the reference's shapes; returns (struct, arrays to keep alive)."""
    n = len(Y)
    off = np.zeros(n + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(v) for v in Y])
    y = np.ascontiguousarray(np.concatenate([np.asarray(v, dtype=np.float64) for v in Y]))
    Bm = np.ascontiguousarray(np.concatenate([np.asarray(b, dtype=np.float64) for b in B], axis=0))
    K, P, T = nu.shape
    M = chi.shape[1]
    keep = [off, y, Bm]

    def fa(x):
        a = np.asfortranarray(x, dtype=np.float64)
        keep.append(a)
        return a.ctypes.data_as(c_double_p)

    inp = PostInput()
    inp.n, inp.K, inp.P, inp.M, inp.T, inp.device = n, K, P, M, T, device
    inp.offsets, inp.y, inp.B = off.ctypes.data_as(c_int64_p), y.ctypes.data_as(c_double_p), Bm.ctypes.data_as(c_double_p)
    inp.nu, inp.Phi, inp.Z, inp.chi, inp.sigma = fa(nu), fa(Phi), fa(Z), fa(chi), fa(sigma)
    if X is not None:
        inp.X, inp.D = fa(X), np.asarray(X).shape[1]
        if eta is not None:
            inp.eta = fa(eta)
        if xi is not None:      # (P, D, M, K, T) in Fortran order is T x K cubes P x D x M
            inp.xi = fa(xi)
    return inp, keep


def post_pointwise(Y, B, nu, Phi, Z, chi, sigma, first_kept=0, X=None, eta=None, xi=None, device=0):
    """bfmmm_post_pointwise on in-memory draws in the reference's shapes: nu (K, P, T), Phi (K, P, M, T), Z (n, K, T),
    chi (n, M, T), sigma (T,), eta (P, D, K, T), xi (P, D, M, K, T).  Returns (llik (T,), mean_pdf, mean_fit) with the
    per-observation arrays as lists shaped like Y."""
    lib = _lib_entry()
    inp, keep = _post_input(Y, B, nu, Phi, Z, chi, sigma, X, eta, xi, device)
    off, y = keep[0], keep[1]
    n, T = len(Y), nu.shape[2]
    ll, pdf, fit = np.zeros(T), np.zeros(len(y)), np.zeros(len(y))
    _check(lib.bfmmm_post_pointwise(C.byref(inp), int(first_kept), ll.ctypes.data_as(c_double_p), pdf.ctypes.data_as(c_double_p),
                                    fit.ctypes.data_as(c_double_p)))
    split = lambda v: [v[off[i]:off[i + 1]].copy() for i in range(n)]
    return ll, split(pdf), split(fit)


class _PostArgs:
    def __init__(self, dir, n_files, basis_degree, boundary_knots, internal_knots, time, Y, burnin_prop, X, cov_adj, device=0):
        lib = _lib_entry()
        self.a = PostArgs()
        lib.bfmmm_post_defaults(C.byref(self.a))
        n = len(Y)
        self.off = np.zeros(n + 1, dtype=np.int64)
        self.off[1:] = np.cumsum([len(v) for v in Y])
        self.y = np.ascontiguousarray(np.concatenate([np.asarray(v, dtype=np.float64).reshape(-1) for v in Y]))
        self.t = np.ascontiguousarray(np.concatenate([np.asarray(v, dtype=np.float64).reshape(-1) for v in time]))
        self.bk = np.ascontiguousarray(boundary_knots, dtype=np.float64)
        self.ik = np.ascontiguousarray(internal_knots, dtype=np.float64)
        self.dir = str(dir).encode()
        a = self.a
        a.dir, a.n_files, a.basis_degree, a.n_internal_knots = self.dir, n_files, basis_degree, len(self.ik)
        a.boundary_knots, a.internal_knots = self.bk.ctypes.data_as(c_double_p), self.ik.ctypes.data_as(c_double_p)
        a.n_funct, a.t, a.y, a.offsets = n, self.t.ctypes.data_as(c_double_p), self.y.ctypes.data_as(c_double_p), self.off.ctypes.data_as(c_int64_p)
        if burnin_prop is not None:
            a.burnin_prop = burnin_prop
        if X is not None:
            self.X = np.asfortranarray(X, dtype=np.float64)
            a.X, a.D = self.X.ctypes.data_as(c_double_p), self.X.shape[1]
        a.cov_adj, a.device = int(bool(cov_adj)), device


def FLLik(dir, n_files, basis_degree, boundary_knots, internal_knots, time, Y, X=None, cov_adj=False):
    """src/PostProcessing.cpp:4892: the log-likelihood of every saved draw."""
    lib = _lib_entry()
    args = _PostArgs(dir, n_files, basis_degree, boundary_knots, internal_knots, time, Y, None, X, cov_adj)
    res = C.c_void_p()
    _check(lib.bfmmm_FLLik(C.byref(args.a), C.byref(res)))
    try:
        return _result_to_dict(lib, res, None, 0)["value"]
    finally:
        lib.bfmmm_result_free(res)


def FSamplePaths(dir, n_files, basis_degree, boundary_knots, internal_knots, time, alpha=0.05, burnin_prop=0.1, simultaneous=False,
                 X=None, cov_adj=False, seed=1):
    """src/PostProcessing.cpp:6599: posterior-predictive sample paths.  Returns lists over the curves: CI_Upper / CI_50 /
    CI_Lower (vectors of n_i), Path_trace / Mean_only_Path_trace (kept x n_i matrices).  `seed`: the keyed generator of the
    predictive noise (the reference draws it from R's stream)."""
    lib = _lib_entry()
    args = _PostArgs(dir, n_files, basis_degree, boundary_knots, internal_knots, time, [np.zeros(len(v)) for v in time],
                     burnin_prop, X, cov_adj)
    res = C.c_void_p()
    _check(lib.bfmmm_FSamplePaths(C.byref(args.a), float(alpha), int(bool(simultaneous)), int(seed), C.byref(res)))
    try:
        d = _result_to_dict(lib, res, None, 0)
    finally:
        lib.bfmmm_result_free(res)
    off = args.off
    out = {}
    for nm in ("CI_Upper", "CI_50", "CI_Lower"):
        v = d[nm].reshape(-1)
        out[nm] = [v[off[i]:off[i + 1]].copy() for i in range(len(off) - 1)]
    for nm in ("Path_trace", "Mean_only_Path_trace"):
        m = d[nm]
        out[nm] = [m[:, off[i]:off[i + 1]].copy() for i in range(len(off) - 1)]
    return out


def _post_scalar(name, dir, n_files, basis_degree, boundary_knots, internal_knots, time, Y, burnin_prop, X, cov_adj):
    lib = _lib_entry()
    args = _PostArgs(dir, n_files, basis_degree, boundary_knots, internal_knots, time, Y, burnin_prop, X, cov_adj)
    out = C.c_double()
    _check(getattr(lib, name)(C.byref(args.a), C.cast(C.byref(out), c_double_p)))
    return out.value


def FDIC(dir, n_files, basis_degree, boundary_knots, internal_knots, time, Y, burnin_prop=0.1, X=None, cov_adj=False):
    """src/PostProcessing.cpp:3660."""
    return _post_scalar("bfmmm_FDIC", dir, n_files, basis_degree, boundary_knots, internal_knots, time, Y, burnin_prop, X, cov_adj)


def FAIC(dir, n_files, basis_degree, boundary_knots, internal_knots, time, Y, burnin_prop=0.1, X=None, cov_adj=False):
    """src/PostProcessing.cpp:4041."""
    return _post_scalar("bfmmm_FAIC", dir, n_files, basis_degree, boundary_knots, internal_knots, time, Y, burnin_prop, X, cov_adj)


def FBIC(dir, n_files, basis_degree, boundary_knots, internal_knots, time, Y, burnin_prop=0.1, X=None, cov_adj=False):
    """src/PostProcessing.cpp:4458."""
    return _post_scalar("bfmmm_FBIC", dir, n_files, basis_degree, boundary_knots, internal_knots, time, Y, burnin_prop, X, cov_adj)


class _PostArgsMV:
    def __init__(self, dir, n_files, Y, burnin_prop, X, cov_adj, device=0):
        lib = _lib_entry()
        self.a = PostArgs()
        lib.bfmmm_post_defaults(C.byref(self.a))
        self.Y = np.asfortranarray(Y, dtype=np.float64)
        self.dir = str(dir).encode()
        a = self.a
        a.dir, a.n_files, a.n_funct, a.P = self.dir, n_files, self.Y.shape[0], self.Y.shape[1]
        a.y = self.Y.ctypes.data_as(c_double_p)
        if burnin_prop is not None:
            a.burnin_prop = burnin_prop
        if X is not None:
            self.X = np.asfortranarray(X, dtype=np.float64)
            a.X, a.D = self.X.ctypes.data_as(c_double_p), self.X.shape[1]
        a.cov_adj, a.device = int(bool(cov_adj)), device


def MVLLik(dir, n_files, Y, X=None, cov_adj=False):
    """src/PostProcessing.cpp:6099."""
    lib = _lib_entry()
    args = _PostArgsMV(dir, n_files, Y, None, X, cov_adj)
    res = C.c_void_p()
    _check(lib.bfmmm_MVLLik(C.byref(args.a), C.byref(res)))
    try:
        return _result_to_dict(lib, res, None, 0)["value"]
    finally:
        lib.bfmmm_result_free(res)


def _post_scalar_mv(name, dir, n_files, Y, burnin_prop, X, cov_adj):
    lib = _lib_entry()
    args = _PostArgsMV(dir, n_files, Y, burnin_prop, X, cov_adj)
    out = C.c_double()
    _check(getattr(lib, name)(C.byref(args.a), C.cast(C.byref(out), c_double_p)))
    return out.value


def MVDIC(dir, n_files, Y, burnin_prop=0.1, X=None, cov_adj=False):
    """src/PostProcessing.cpp:5789."""
    return _post_scalar_mv("bfmmm_MVDIC", dir, n_files, Y, burnin_prop, X, cov_adj)


def MVAIC(dir, n_files, Y, burnin_prop=0.1, X=None, cov_adj=False):
    """src/PostProcessing.cpp:5116."""
    return _post_scalar_mv("bfmmm_MVAIC", dir, n_files, Y, burnin_prop, X, cov_adj)


def MVBIC(dir, n_files, Y, burnin_prop=0.1, X=None, cov_adj=False):
    """src/PostProcessing.cpp:5452."""
    return _post_scalar_mv("bfmmm_MVBIC", dir, n_files, Y, burnin_prop, X, cov_adj)


def ConditionalPredictiveOrdinates(dir, n_files, basis_degree, boundary_knots, internal_knots, time, Y, burnin_prop=0.1, X=None,
                                   cov_adj=False, log_CPO=True):
    """src/PostProcessing.cpp:6339: (log) conditional predictive ordinate of every curve."""
    lib = _lib_entry()
    args = _PostArgs(dir, n_files, basis_degree, boundary_knots, internal_knots, time, Y, burnin_prop, X, cov_adj)
    res = C.c_void_p()
    _check(lib.bfmmm_ConditionalPredictiveOrdinates(C.byref(args.a), int(bool(log_CPO)), C.byref(res)))
    try:
        return _result_to_dict(lib, res, None, 0)["value"]
    finally:
        lib.bfmmm_result_free(res)


# ---- credible intervals (src/PostProcessing.cpp:99, :3435, :3505) -------------------------------------------------------
def _ci_call(fn, a, keep):
    lib = _lib_entry()
    res = C.c_void_p()
    _check(fn(C.byref(a), C.byref(res)))
    try:
        return _result_to_dict(lib, res, None, 0)
    finally:
        lib.bfmmm_result_free(res)


def _ci_args(dir, n_files, alpha, burnin_prop):
    lib = _lib_entry()
    a = CiArgs()
    lib.bfmmm_ci_defaults(C.byref(a))
    keep = [str(dir).encode()]
    a.dir, a.n_files, a.alpha, a.burnin_prop = keep[0], n_files, alpha, burnin_prop
    return a, keep


def SigmaCI(dir, n_files, alpha=0.05, burnin_prop=0.1):
    """src/PostProcessing.cpp:3435 (CI_Lower is the median, as in the reference)."""
    a, keep = _ci_args(dir, n_files, alpha, burnin_prop)
    d = _ci_call(_lib_entry().bfmmm_SigmaCI, a, keep)
    return {k: float(v.reshape(-1)[0]) for k, v in d.items()}


def ZCI(dir, n_files, alpha=0.05, rescale=True, burnin_prop=0.1):
    """src/PostProcessing.cpp:3505."""
    a, keep = _ci_args(dir, n_files, alpha, burnin_prop)
    a.rescale = int(bool(rescale))
    return _ci_call(_lib_entry().bfmmm_ZCI, a, keep)


def FMeanCI(dir, n_files, time, basis_degree, boundary_knots, internal_knots, k, alpha=0.05, rescale=True, simultaneous=False,
            burnin_prop=0.1, X=None, trans_mats=None):
    """src/PostProcessing.cpp:99."""
    a, keep = _ci_args(dir, n_files, alpha, burnin_prop)
    t = np.ascontiguousarray(time, dtype=np.float64).reshape(-1)
    bk = np.ascontiguousarray(boundary_knots, dtype=np.float64)
    ik = np.ascontiguousarray(internal_knots, dtype=np.float64)
    keep += [t, bk, ik]
    a.time, a.n_time, a.basis_degree, a.n_internal_knots = t.ctypes.data_as(c_double_p), len(t), basis_degree, len(ik)
    a.boundary_knots, a.internal_knots = bk.ctypes.data_as(c_double_p), ik.ctypes.data_as(c_double_p)
    a.k, a.rescale, a.simultaneous = k, int(bool(rescale)), int(bool(simultaneous))
    if X is not None:
        Xf = np.asfortranarray(X, dtype=np.float64)
        keep.append(Xf)
        a.X, a.n_x, a.D = Xf.ctypes.data_as(c_double_p), Xf.shape[0], Xf.shape[1]
    if trans_mats is not None:
        tm = np.asfortranarray(trans_mats, dtype=np.float64)
        keep.append(tm)
        a.trans_mats = tm.ctypes.data_as(c_double_p)
    d = _ci_call(_lib_entry().bfmmm_FMeanCI, a, keep)
    if X is None:
        for nm in ("CI_Upper", "CI_50", "CI_Lower"):
            d[nm] = d[nm].reshape(-1)
    return d


def _cov_x(a, keep, X):
    if X is not None:
        Xf = np.asfortranarray(X, dtype=np.float64)
        keep.append(Xf)
        a.X, a.n_x, a.D = Xf.ctypes.data_as(c_double_p), Xf.shape[0], Xf.shape[1]


def FCovCI(dir, n_files, time1, time2, basis_degree, boundary_knots, internal_knots, l, m, alpha=0.05, rescale=True,
           simultaneous=False, burnin_prop=0.1, X=None, trans_mats=None):
    """src/PostProcessing.cpp:1781.  With X (n_x covariate settings: the covariate-dependent covariance) the bands are
    (n_time1, n_time2, n_x) and cov_trace (n_time1, n_time2, kept, n_x) -- the reference's list of n_x cubes."""
    a, keep = _ci_args(dir, n_files, alpha, burnin_prop)
    t1 = np.ascontiguousarray(time1, dtype=np.float64).reshape(-1)
    t2 = np.ascontiguousarray(time2, dtype=np.float64).reshape(-1)
    bk = np.ascontiguousarray(boundary_knots, dtype=np.float64)
    ik = np.ascontiguousarray(internal_knots, dtype=np.float64)
    keep += [t1, t2, bk, ik]
    a.time, a.n_time, a.time2, a.n_time2 = t1.ctypes.data_as(c_double_p), len(t1), t2.ctypes.data_as(c_double_p), len(t2)
    a.basis_degree, a.n_internal_knots = basis_degree, len(ik)
    a.boundary_knots, a.internal_knots = bk.ctypes.data_as(c_double_p), ik.ctypes.data_as(c_double_p)
    a.l, a.m, a.rescale, a.simultaneous = l, m, int(bool(rescale)), int(bool(simultaneous))
    _cov_x(a, keep, X)
    if trans_mats is not None:
        tm = np.asfortranarray(trans_mats, dtype=np.float64)
        keep.append(tm)
        a.trans_mats = tm.ctypes.data_as(c_double_p)
    return _ci_call(_lib_entry().bfmmm_FCovCI, a, keep)


def HDFCovCI(dir, n_files, time1, time2, basis_degree, boundary_knots, internal_knots, l, m, alpha=0.05, rescale=True,
             simultaneous=False, burnin_prop=0.1, X=None):
    """src/PostProcessing.cpp:2468: `time1`, `time2` n x dim matrices, `basis_degree` a vector, `boundary_knots` dim x 2,
    `internal_knots` a list.  (The reference evaluates both bases at time1: the surface is time1 x time1.)"""
    a, keep = _ci_args(dir, n_files, alpha, burnin_prop)
    dim = len(basis_degree)
    t1 = np.asfortranarray(np.asarray(time1, dtype=np.float64).reshape(-1, dim))
    t2 = np.asfortranarray(np.asarray(time2, dtype=np.float64).reshape(-1, dim))
    bk = np.ascontiguousarray(np.asarray(boundary_knots, dtype=np.float64).reshape(dim, 2))
    ik = np.ascontiguousarray(np.concatenate([np.asarray(v, dtype=np.float64).reshape(-1) for v in internal_knots]))
    deg = (C.c_int32 * dim)(*[int(x) for x in basis_degree])
    nint = (C.c_int32 * dim)(*[len(v) for v in internal_knots])
    keep += [t1, t2, bk, ik, deg, nint]
    a.time, a.n_time, a.time2, a.n_time2 = t1.ctypes.data_as(c_double_p), t1.shape[0], t2.ctypes.data_as(c_double_p), t2.shape[0]
    a.dim, a.basis_degree_hd, a.n_internal_hd = dim, deg, nint
    a.boundary_knots, a.internal_knots = bk.ctypes.data_as(c_double_p), ik.ctypes.data_as(c_double_p)
    a.l, a.m, a.rescale, a.simultaneous = l, m, int(bool(rescale)), int(bool(simultaneous))
    _cov_x(a, keep, X)
    return _ci_call(_lib_entry().bfmmm_HDFCovCI, a, keep)


def MVCovCI(dir, n_files, l, m, alpha=0.05, rescale=True, burnin_prop=0.1, X=None):
    """src/PostProcessing.cpp:3097 (multivariate model: P x P bands, pointwise)."""
    a, keep = _ci_args(dir, n_files, alpha, burnin_prop)
    a.l, a.m, a.rescale = l, m, int(bool(rescale))
    _cov_x(a, keep, X)
    return _ci_call(_lib_entry().bfmmm_MVCovCI, a, keep)


def MVMeanCI(dir, n_files, alpha=0.05, rescale=True, burnin_prop=0.1, X=None):
    """src/PostProcessing.cpp:1410.  With X the mean_trace is returned as (K, P, kept, n_x)."""
    a, keep = _ci_args(dir, n_files, alpha, burnin_prop)
    a.rescale = int(bool(rescale))
    if X is not None:
        Xf = np.asfortranarray(X, dtype=np.float64)
        keep.append(Xf)
        a.X, a.n_x, a.D = Xf.ctypes.data_as(c_double_p), Xf.shape[0], Xf.shape[1]
    d = _ci_call(_lib_entry().bfmmm_MVMeanCI, a, keep)
    if X is not None:
        K, P, tot = d["mean_trace"].shape
        d["mean_trace"] = d["mean_trace"].reshape((K, P, tot // Xf.shape[0], Xf.shape[0]), order="F")
    return d


def HDFMeanCI(dir, n_files, time, basis_degree, boundary_knots, internal_knots, k, alpha=0.05, rescale=True, simultaneous=False,
              burnin_prop=0.1, X=None, trans_mats=None):
    """src/PostProcessing.cpp:806: `time` n_time x dim, `basis_degree` a vector, `boundary_knots` dim x 2, `internal_knots` a list."""
    a, keep = _ci_args(dir, n_files, alpha, burnin_prop)
    dim = len(basis_degree)
    t = np.asfortranarray(np.asarray(time, dtype=np.float64).reshape(-1, dim))
    bk = np.ascontiguousarray(np.asarray(boundary_knots, dtype=np.float64).reshape(dim, 2))
    ik = np.ascontiguousarray(np.concatenate([np.asarray(v, dtype=np.float64).reshape(-1) for v in internal_knots]))
    deg = (C.c_int32 * dim)(*[int(x) for x in basis_degree])
    nint = (C.c_int32 * dim)(*[len(v) for v in internal_knots])
    keep += [t, bk, ik, deg, nint]
    a.time, a.n_time, a.dim, a.basis_degree_hd, a.n_internal_hd = t.ctypes.data_as(c_double_p), t.shape[0], dim, deg, nint
    a.boundary_knots, a.internal_knots = bk.ctypes.data_as(c_double_p), ik.ctypes.data_as(c_double_p)
    a.k, a.rescale, a.simultaneous = k, int(bool(rescale)), int(bool(simultaneous))
    if X is not None:
        Xf = np.asfortranarray(X, dtype=np.float64)
        keep.append(Xf)
        a.X, a.n_x, a.D = Xf.ctypes.data_as(c_double_p), Xf.shape[0], Xf.shape[1]
    if trans_mats is not None:
        tm = np.asfortranarray(trans_mats, dtype=np.float64)
        keep.append(tm)
        a.trans_mats = tm.ctypes.data_as(c_double_p)
    d = _ci_call(_lib_entry().bfmmm_HDFMeanCI, a, keep)
    if X is None:
        for nm in ("CI_Upper", "CI_50", "CI_Lower"):
            d[nm] = d[nm].reshape(-1)
    return d
