"""ctypes loader of the C-ABI sampler library (bayesfmmm_amd/libbfmmm_hip.so, declared in include/bfmmm.h).

There is no CPU fallback: if the HIP library is missing, or no GPU is visible when a sampler is
created, the call fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BFMMM_LIB_PATH", os.path.join(_HERE, "libbfmmm_hip.so"))      # (override: diagnostic builds, tools/timeline.py)

c_double_p = C.POINTER(C.c_double)
c_int64_p = C.POINTER(C.c_int64)


class BfmmmConfig(C.Structure):
    """Mirror of `bfmmm_config` (include/bfmmm.h)."""
    _fields_ = [("model", C.c_int32), ("n_funct", C.c_int32), ("K", C.c_int32), ("n_eigen", C.c_int32),
                ("basis_degree", C.c_int32), ("n_internal_knots", C.c_int32), ("P", C.c_int32),
                ("tot_mcmc_iters", C.c_int32), ("c", C.c_double * 8), ("b", C.c_double), ("nu_1", C.c_double),
                ("alpha1l", C.c_double), ("alpha2l", C.c_double), ("beta1l", C.c_double), ("beta2l", C.c_double),
                ("a_Z_PM", C.c_double), ("a_pi_PM", C.c_double), ("var_alpha3", C.c_double),
                ("var_epsilon1", C.c_double), ("var_epsilon2", C.c_double), ("alpha_nu", C.c_double),
                ("beta_nu", C.c_double), ("alpha_eta", C.c_double), ("beta_eta", C.c_double),
                ("alpha_0", C.c_double), ("beta_0", C.c_double)]


# every symbol include/bfmmm.h declares: (restype, argtypes)
SYMBOLS = {
    "bfmmm_config_defaults": (None, [C.POINTER(BfmmmConfig)]),
    "bfmmm_create": (C.c_int, [C.POINTER(BfmmmConfig), C.c_int, c_double_p, c_double_p, c_int64_p, c_double_p,
                               c_double_p, C.POINTER(C.c_void_p)]),
    "bfmmm_destroy": (None, [C.c_void_p]),
    "bfmmm_create_batch": (C.c_int, [C.POINTER(BfmmmConfig), C.c_int, c_double_p, c_double_p, c_int64_p, c_double_p,
                                     c_double_p, C.c_int, C.POINTER(C.c_void_p)]),
    "bfmmm_create_from_basis_batch": (C.c_int, [C.POINTER(BfmmmConfig), C.c_int, c_double_p, c_double_p, c_int64_p, C.c_int,
                                                C.c_int, c_double_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "bfmmm_select_chain": (C.c_int, [C.c_void_p, C.c_int]),
    "bfmmm_n_chains": (C.c_int, [C.c_void_p]),
    "bfmmm_set_chain_id_stride": (C.c_int, [C.c_void_p, C.c_uint32]),
    "bfmmm_gather_best": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, c_double_p, C.POINTER(C.c_int32), C.POINTER(C.c_int)]),
    "bfmmm_set_covariates": (C.c_int, [C.c_void_p, c_double_p, C.c_int, C.c_int]),
    "bfmmm_get_basis": (C.c_int, [C.c_void_p, c_double_p, C.c_int64]),
    "bfmmm_set_state": (C.c_int, [C.c_void_p, C.c_char_p, c_double_p, C.c_int64]),
    "bfmmm_get_state": (C.c_int, [C.c_void_p, C.c_char_p, c_double_p, C.c_int64]),
    "bfmmm_init_state": (C.c_int, [C.c_void_p, C.c_int, C.c_uint64, C.c_uint32]),
    "bfmmm_run": (C.c_int, [C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_uint64, C.c_uint32, C.c_int, C.c_double]),
    "bfmmm_prepare_run": (C.c_int, [C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_uint64, C.c_uint32, C.c_int]),
    "bfmmm_set_slot_base": (C.c_int, [C.c_void_p, C.c_int]),
    "bfmmm_create_from_basis": (C.c_int, [C.POINTER(BfmmmConfig), C.c_int, c_double_p, c_double_p, c_int64_p, C.c_int, C.c_int,
                                          c_double_p, C.c_int, C.POINTER(C.c_void_p)]),
    "bfmmm_tempered_transition": (C.c_int, [C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_double, C.c_uint64, C.c_uint32,
                                            c_double_p, C.POINTER(C.c_int)]),
    "bfmmm_get_chain": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, c_double_p, C.c_int64]),
    "bfmmm_debug_get": (C.c_int, [C.c_void_p, C.c_char_p, c_double_p, C.c_int64, c_int64_p]),
    "bfmmm_set_profile": (C.c_int, [C.c_void_p, C.c_int]),
    "bfmmm_get_timing": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_double), c_int64_p]),
    "bfmmm_set_exact_instances": (None, [C.c_int]),
    "bfmmm_last_error": (C.c_char_p, []),
}

_LIB = None


def load():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). bayesfmmm_amd has no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _LIB = lib
    return _LIB


class BfmmmError(RuntimeError):
    pass


def check(rc):
    if rc != 0:
        raise BfmmmError(load().bfmmm_last_error().decode())
