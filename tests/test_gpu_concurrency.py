"""A chain must give bit-identical draws whatever else the GPU is doing.  A second handle keeps the device busy on its own
stream while a stand-alone chain is run repeatedly; every run must equal the run made on an idle device.  (This is the
situation of the multi-try entry points, which run batches on several host threads, and of a chain batch whose two halves
share the device on two streams.  It caught a prefetch wait of k_sweep_fast that was one group short: with the device to
itself the data had always arrived, under load one run in five differed in sigma^2.)"""
import threading

import numpy as np
import pytest

from gpu_parity import make_sampler, oracle_slot, random_state, STATE_NAMES
from simdata import simulate_functional, truth_chain

pytestmark = pytest.mark.gpu

CHAIN_NAMES = ["nu", "Phi", "chi", "Z", "pi", "alpha_3", "delta", "A", "gamma", "tau", "sigma_sq", "loglik"]


def _state(sim, seed):
    model, ch = truth_chain(sim, 2)
    random_state(sim, ch, seed)
    return {nm: oracle_slot(ch, nm, 0) for nm in STATE_NAMES}


@pytest.mark.parametrize("sweep", ["warm", "nu_z", "cov"])
def test_chain_is_bit_identical_under_concurrent_load(sweep):
    import bayesfmmm_amd as bf
    S = bf.sampler
    sim = simulate_functional(n=203, M=3, sigma_sq=0.01, seed=21)
    big = simulate_functional(n=3000, M=3, sigma_sq=0.01, seed=5)
    T = 23
    st, st_big = _state(sim, 100), _state(big, 101)
    pcz = sweep == "nu_z"
    mask = S.SWEEP_NU_Z if pcz else S.SWEEP_WARM
    X = None
    if sweep == "cov":        # the covariate-adjusted sweep: eta / Xi block, k_loglik
        mask = S.SWEEP_WARM | S.COV_MEAN | S.COV_XI
        X = np.random.default_rng(2).standard_normal((sim["n"], 2))
    if pcz:
        for s_ in (st, st_big):
            s_["Phi"] = np.zeros_like(s_["Phi"]); s_["chi"] = np.zeros_like(s_["chi"])

    def run_once():
        a = make_sampler(sim, T)
        if X is not None:
            a.set_covariates(X, covariance_adj=True)
        a.set_state(**st)
        a.run(mask, 9, first_iter=0, seed=5, chain=2, phi_chi_zero=pcz)
        a.run(mask, T - 9, first_iter=9, seed=5, chain=2, phi_chi_zero=pcz)
        out = {nm: a.get_chain(nm) for nm in CHAIN_NAMES}
        a.close()
        return out

    ref = run_once()
    stop = threading.Event()
    err = []

    def load():
        try:
            b = make_sampler(big, 200)
            b.set_state(**st_big)
            while not stop.is_set():
                b.run(S.SWEEP_WARM, 150, first_iter=0, seed=1, chain=0)
            b.close()
        except Exception as e:        # pragma: no cover
            err.append(e)

    th = threading.Thread(target=load)
    th.start()
    try:
        for trial in range(30 if X is None else 15):
            o = run_once()
            for nm in CHAIN_NAMES:
                np.testing.assert_array_equal(o[nm], ref[nm], err_msg=f"{sweep}: run {trial} under load differs in {nm}")
    finally:
        stop.set()
        th.join()
    assert not err, err


def test_multivariate_chain_is_bit_identical_under_concurrent_load():
    import bayesfmmm_amd as bf
    S = bf.sampler
    rng = np.random.default_rng(6)
    n, P, K, M, T = 300, 12, 3, 2, 12
    Y = rng.standard_normal((n, P))
    big = simulate_functional(n=3000, M=3, sigma_sq=0.01, seed=5)
    st_big = _state(big, 101)

    def run_once():
        a = bf.Sampler(bf.default_config(model=bf.MODEL_MULTIVARIATE, K=K, n_eigen=M, tot_mcmc_iters=T), Y)
        a.init_state(1, 17, chain=0)
        a.run(S.SWEEP_WARM, T, seed=17, chain=0)
        out = {nm: a.get_chain(nm) for nm in CHAIN_NAMES}
        a.close()
        return out

    ref = run_once()
    stop = threading.Event()

    def load():
        b = make_sampler(big, 200)
        b.set_state(**st_big)
        while not stop.is_set():
            b.run(S.SWEEP_WARM, 150, first_iter=0, seed=1, chain=0)
        b.close()

    th = threading.Thread(target=load)
    th.start()
    try:
        for trial in range(20):
            o = run_once()
            for nm in CHAIN_NAMES:
                np.testing.assert_array_equal(o[nm], ref[nm], err_msg=f"multivariate: run {trial} under load differs in {nm}")
    finally:
        stop.set()
        th.join()


def _under_load(run_once, names, trials, label):
    """run_once() on an idle device, then `trials` times while a second handle keeps the GPU busy: all bit-identical"""
    import bayesfmmm_amd as bf
    S = bf.sampler
    big = simulate_functional(n=3000, M=3, sigma_sq=0.01, seed=5)
    st_big = _state(big, 101)
    ref = run_once()
    stop = threading.Event()

    def load():
        b = make_sampler(big, 200)
        b.set_state(**st_big)
        while not stop.is_set():
            b.run(S.SWEEP_WARM, 150, first_iter=0, seed=1, chain=0)
        b.close()

    th = threading.Thread(target=load)
    th.start()
    try:
        for trial in range(trials):
            o = run_once()
            for nm in names:
                np.testing.assert_array_equal(o[nm], ref[nm], err_msg=f"{label}: run {trial} under load differs in {nm}")
    finally:
        stop.set()
        th.join()


def test_chain_batch_on_two_streams_is_bit_identical_under_concurrent_load():
    """six Nu_Z chains as one batch: two sub-batches on two streams, the lean trailing Z kernel, grouped pair-Gram staging"""
    import bayesfmmm_amd as bf
    S = bf.sampler
    sim = simulate_functional(n=500, M=3, sigma_sq=0.01, seed=3)
    T, NCH = 14, 6

    def run_once():
        cfg = bf.default_config(model=bf.MODEL_FUNCTIONAL, K=sim["K"], n_eigen=sim["M"], basis_degree=3, tot_mcmc_iters=T)
        s = bf.Sampler(cfg, sim["y"], sim["t"], sim["internal_knots"], sim["boundary_knots"], n_chains=NCH)
        for q in range(NCH):
            s.select_chain(q)
            s.init_state(0, 7, chain=q)
        s.run(S.SWEEP_NU_Z, T, seed=7, phi_chi_zero=True)
        out = {}
        for q in range(NCH):
            s.select_chain(q)
            for nm in ["nu", "Z", "pi", "tau", "sigma_sq", "loglik"]:
                out[f"{nm}{q}"] = s.get_chain(nm)
        s.close()
        return out

    names = [f"{nm}{q}" for q in range(NCH) for nm in ["nu", "Z", "pi", "tau", "sigma_sq", "loglik"]]
    _under_load(run_once, names, 15, "Nu_Z batch")


def test_general_sweep_kernel_is_bit_identical_under_concurrent_load():
    """K = 5, M = 9, P = 30: A P = 1500 elements, beyond the register-resident sweep -- the general k_sweep and the KMAX / MMAX
    instantiations of the per-curve kernels"""
    import bayesfmmm_amd as bf
    from test_gpu_shapes import simulate
    S = bf.sampler
    sim = simulate(64, 5, 9, 3, 26, seed=159)

    def run_once():
        cfg = bf.default_config(model=bf.MODEL_FUNCTIONAL, K=5, n_eigen=9, basis_degree=3, tot_mcmc_iters=8)
        s = bf.Sampler(cfg, sim["y"], sim["t"], sim["internal_knots"], sim["boundary_knots"])
        s.init_state(1, 5, chain=0)
        s.run(S.SWEEP_WARM, 8, seed=5)
        out = {nm: s.get_chain(nm) for nm in CHAIN_NAMES}
        s.close()
        return out

    _under_load(run_once, CHAIN_NAMES, 15, "general sweep")
