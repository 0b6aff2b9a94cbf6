"""bfmmm_prepare_run captures the HIP graphs a run replays AND launches every freshly instantiated graph once (the first launch of
a graph costs the device 13 - 20 us more than later ones), between a snapshot and a restore of the chains' work state
(bfmmm_capi.hip, "DRY LAUNCH").  A prepared run must therefore be BIT-identical to the same run without the preparation: single
chain, a batch on two streams, the Nu_Z stage (trailing lean Z update), covariates (second arena), the multivariate model, and a
run prepared in the middle of a chain (prepared proposals / deferred log-likelihood pending in device state)."""
import numpy as np
import pytest

from gpu_parity import make_sampler
from simdata import simulate_functional

pytestmark = pytest.mark.gpu

NAMES = ["nu", "Phi", "chi", "Z", "pi", "alpha_3", "delta", "A", "gamma", "tau", "sigma_sq", "loglik"]


def _state(sim, rng):
    n, K, P, M = sim["n"], sim["K"], sim["P"], sim["M"]
    return dict(nu=sim["nu"] + 0.2 * rng.standard_normal((K, P)), Phi=sim["Phi"] + 0.05 * rng.standard_normal((K, P, M)),
                chi=sim["chi"] + 0.1 * rng.standard_normal((n, M)), Z=rng.dirichlet(np.full(K, 2.0), size=n),
                pi=rng.dirichlet(np.full(K, 5.0)), alpha_3=[3.5], delta=rng.gamma(2.0, 1.0, size=(K, M)),
                A=rng.gamma(2.0, 1.0, size=(K, 2)), gamma=rng.gamma(2.0, 0.7, size=(K, P, M)), tau=rng.gamma(3.0, 0.5, size=K),
                sigma_sq=[0.02])


@pytest.mark.parametrize("case", ["warm", "warm_batch5", "nu_z_batch4", "cov", "mid_chain"])
def test_prepared_run_is_bit_identical(case):
    import bayesfmmm_amd as bf
    S = bf.sampler
    D = 2 if case == "cov" else 0
    sim = simulate_functional(n=77, M=2, sigma_sq=0.01, seed=17, D=max(D, 1))
    nch = {"warm_batch5": 5, "nu_z_batch4": 4}.get(case, 1)
    T, first = 37, (12 if case == "mid_chain" else 0)
    mask = S.SWEEP_NU_Z if case == "nu_z_batch4" else S.SWEEP_WARM
    if D:
        mask |= S.COV_MEAN | S.COV_XI
    pcz = case == "nu_z_batch4"
    outs = []
    for prepared in (False, True):
        cfg = bf.default_config(model=bf.MODEL_FUNCTIONAL, K=sim["K"], n_eigen=sim["M"], basis_degree=3, tot_mcmc_iters=T)
        smp = bf.Sampler(cfg, sim["y"], sim["t"], sim["internal_knots"], sim["boundary_knots"], n_chains=nch)
        if D:
            smp.set_covariates(sim["X"][:, :D], True)
        for q in range(nch):
            smp.select_chain(q)
            st = _state(sim, np.random.default_rng(50 + q))
            if pcz:
                st["Phi"] = np.zeros_like(st["Phi"]); st["chi"] = np.zeros_like(st["chi"])
            smp.set_state(**st)
        if first:
            smp.run(mask, first, first_iter=0, seed=4, phi_chi_zero=pcz)
        if prepared:
            smp.prepare_run(mask, T - first, first_iter=first, seed=4, phi_chi_zero=pcz)
            smp.prepare_run(mask, T - first, first_iter=first, seed=4, phi_chi_zero=pcz)      # (a second call finds the graphs: no dry launch)
        smp.run(mask, T - first, first_iter=first, seed=4, phi_chi_zero=pcz)
        got = []
        for q in range(nch):
            smp.select_chain(q)
            got.append({nm: np.array(smp.get_chain(nm)) for nm in NAMES})
            got[-1].update({"state_" + nm: np.array(smp.get_state(nm)) for nm in ("nu", "Z", "chi", "sigma_sq")})
            if D:
                got[-1].update({nm: np.array(smp.get_chain(nm)) for nm in ("eta", "xi", "tau_eta")})
        outs.append(got)
        smp.close()
    for q in range(nch):
        for nm in outs[0][q]:
            np.testing.assert_array_equal(outs[0][q][nm], outs[1][q][nm], err_msg=f"{case} chain {q} {nm}")
    assert np.isfinite(outs[0][0]["loglik"]).all()


def test_prepared_multivariate_run_is_bit_identical():
    import bayesfmmm_amd as bf
    rng = np.random.default_rng(3)
    n, P, K, M, T = 120, 9, 2, 2, 25
    Y = rng.standard_normal((n, P))
    outs = []
    for prepared in (False, True):
        cfg = bf.default_config(model=bf.MODEL_MULTIVARIATE, K=K, n_eigen=M, tot_mcmc_iters=T)
        smp = bf.Sampler(cfg, Y)
        r2 = np.random.default_rng(9)
        smp.set_state(nu=r2.standard_normal((K, P)), Phi=0.3 * r2.standard_normal((K, P, M)), chi=r2.standard_normal((n, M)),
                      Z=r2.dirichlet(np.ones(K), size=n), pi=np.full(K, 1.0 / K), alpha_3=[5.0], delta=np.ones((K, M)),
                      A=np.ones((K, 2)), gamma=np.ones((K, P, M)), tau=np.ones(K), sigma_sq=[0.5])
        if prepared:
            smp.prepare_run(bf.sampler.SWEEP_WARM, T, first_iter=0, seed=2)
        smp.run(bf.sampler.SWEEP_WARM, T, first_iter=0, seed=2)
        outs.append({nm: np.array(smp.get_chain(nm)) for nm in NAMES})
        smp.close()
    for nm in NAMES:
        np.testing.assert_array_equal(outs[0][nm], outs[1][nm], err_msg=nm)


def test_run_in_pieces_is_bit_identical_and_state_is_current_between_calls():
    """A single chain leaves the scalar job (delta, A, gamma, tau) of an iteration to an idle workgroup of the NEXT iteration's
    k_pair_gram (Ctx::defer_hyper); the run's closing kernel runs the last one.  So: (a) the chain of one 37-iteration call equals
    that of calls of 12 + 1 + 24 iterations, bit for bit; (b) the state read between the calls is the state of the last stored
    slot -- nothing is left pending when bfmmm_run returns."""
    import bayesfmmm_amd as bf
    S = bf.sampler
    sim = simulate_functional(n=77, M=2, sigma_sq=0.01, seed=17, D=1)
    T = 37
    outs = []
    for pieces in ((T,), (12, 1, 24)):
        cfg = bf.default_config(model=bf.MODEL_FUNCTIONAL, K=sim["K"], n_eigen=sim["M"], basis_degree=3, tot_mcmc_iters=T)
        smp = bf.Sampler(cfg, sim["y"], sim["t"], sim["internal_knots"], sim["boundary_knots"], n_chains=1)
        smp.set_state(**_state(sim, np.random.default_rng(50)))
        done = 0
        for cnt in pieces:
            smp.run(S.SWEEP_WARM, cnt, first_iter=done, seed=4)
            done += cnt
            for nm in ("delta", "A", "gamma", "tau"):
                ch = np.array(smp.get_chain(nm))
                last = ch[..., done - 1] if ch.shape[-1] == T else ch[done - 1]
                np.testing.assert_array_equal(np.array(smp.get_state(nm)).reshape(-1), np.asarray(last).reshape(-1),
                                              err_msg=f"{nm} after {done} iterations")
        outs.append({nm: np.array(smp.get_chain(nm)) for nm in NAMES})
        smp.close()
    for nm in NAMES:
        np.testing.assert_array_equal(outs[0][nm], outs[1][nm], err_msg=nm)
