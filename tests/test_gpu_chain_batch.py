"""Chain batches (bfmmm_create_batch): chain q of a batch must be BIT-identical to a stand-alone sampler run with RNG
chain id `chain + q * stride` -- the chain index is only a grid dimension of the same kernels, and every reduction has a
fixed order.  Multi-try chains (src/UserFunctions.cpp:302-325) are what the batches are for."""
import numpy as np
import pytest

import oracle_lib as O
from gpu_parity import make_sampler, oracle_slot, push_state, random_state, rel_err, STATE_NAMES
from simdata import simulate_functional, truth_chain

pytestmark = pytest.mark.gpu

CHAIN_NAMES = ["nu", "Phi", "chi", "Z", "pi", "alpha_3", "delta", "A", "gamma", "tau", "sigma_sq", "loglik"]


def _states(sim, n_chains):
    """a different generic start state per chain"""
    out = []
    for q in range(n_chains):
        model, ch = truth_chain(sim, 2)
        random_state(sim, ch, 100 + q)
        out.append({nm: oracle_slot(ch, nm, 0) for nm in STATE_NAMES})
    return out


@pytest.mark.parametrize("sweep", ["warm", "nu_z", "theta"])
def test_batched_chains_equal_standalone_chains_bitwise(sweep):
    import bayesfmmm_amd as bf
    S = bf.sampler
    sim = simulate_functional(n=203, M=3, sigma_sq=0.01, seed=21)       # 203: a ragged last curve block
    T, NCH, stride, chain0 = 23, 5, 3, 2
    states = _states(sim, NCH)
    mask = {"warm": S.SWEEP_WARM, "nu_z": S.SWEEP_NU_Z, "theta": S.SWEEP_THETA}[sweep]
    pcz = sweep == "nu_z"
    if pcz:
        for st in states:
            st["Phi"] = np.zeros_like(st["Phi"])
            st["chi"] = np.zeros_like(st["chi"])
    batch = make_sampler_batch(sim, T, NCH)
    batch.set_chain_id_stride(stride)
    for q in range(NCH):
        batch.select_chain(q)
        batch.set_state(**states[q])
    # two calls: the second continues from device state (prepared proposals, deferred log-likelihood) of the first
    batch.run(mask, 9, first_iter=0, seed=5, chain=chain0, phi_chi_zero=pcz)
    batch.run(mask, T - 9, first_iter=9, seed=5, chain=chain0, phi_chi_zero=pcz)
    for q in range(NCH):
        solo = make_sampler(sim, T)
        solo.set_state(**states[q])
        solo.run(mask, 9, first_iter=0, seed=5, chain=chain0 + q * stride, phi_chi_zero=pcz)
        solo.run(mask, T - 9, first_iter=9, seed=5, chain=chain0 + q * stride, phi_chi_zero=pcz)
        batch.select_chain(q)
        for nm in CHAIN_NAMES:
            np.testing.assert_array_equal(batch.get_chain(nm), solo.get_chain(nm), err_msg=f"{sweep} chain {q} {nm}")
        for nm in STATE_NAMES:
            np.testing.assert_array_equal(batch.get_state(nm), solo.get_state(nm), err_msg=f"{sweep} chain {q} state {nm}")
        solo.close()
    # the chains are different chains
    batch.select_chain(0)
    a = batch.get_chain("nu")
    batch.select_chain(1)
    assert rel_err(a, batch.get_chain("nu")) > 1e-3
    batch.close()


def make_sampler_batch(sim, T, n_chains, **cfg_kw):
    import bayesfmmm_amd as bf
    cfg = bf.default_config(model=bf.MODEL_FUNCTIONAL, K=sim["K"], n_eigen=sim["M"], basis_degree=3,
                            tot_mcmc_iters=T, **cfg_kw)
    return bf.Sampler(cfg, sim["y"], sim["t"], sim["internal_knots"], sim["boundary_knots"], n_chains=n_chains)


def test_batched_chain_matches_the_oracle():
    """chain 2 of a batch against the CPU oracle run with the same chain id (the stand-alone parity tests cover chain 0)"""
    import bayesfmmm_amd as bf
    sim = simulate_functional(n=41, M=2, sigma_sq=0.01, seed=9)
    T, NCH = 6, 3
    model, ch = truth_chain(sim, T)
    random_state(sim, ch, 7)
    batch = make_sampler_batch(sim, T, NCH)
    for q in range(NCH):
        batch.select_chain(q)
        push_state(batch, ch)
    O.run_sweeps(model, O.make_hyper(sim["K"]), ch, O.SWEEP_WARM, n_iter=T, seed=13, chain_id=2)
    batch.run(bf.SWEEP_WARM, T, seed=13, chain=0)
    batch.select_chain(2)
    for nm in ["nu", "Phi", "chi", "Z", "sigma_sq", "loglik", "tau", "gamma"]:
        ref = getattr(ch, {"sigma_sq": "sigma"}.get(nm, nm))
        assert rel_err(batch.get_chain(nm), ref) < 2e-6, nm
    batch.close()


def test_batched_chains_with_covariates_equal_standalone_bitwise():
    import bayesfmmm_amd as bf
    S = bf.sampler
    sim = simulate_functional(n=90, M=2, sigma_sq=0.01, seed=33)
    rng = np.random.default_rng(2)
    X = rng.standard_normal((sim["n"], 2))
    T, NCH = 7, 5          # (five chains: the batch runs as two sub-batches on two streams, the second over moved covariate buffers)
    states = _states(sim, NCH)
    mask = S.SWEEP_WARM | S.COV_MEAN | S.COV_XI
    batch = make_sampler_batch(sim, T, NCH)
    batch.set_covariates(X, covariance_adj=True)
    for q in range(NCH):
        batch.select_chain(q)
        batch.set_state(**states[q])
    batch.run(mask, T, seed=3, chain=10)
    names = CHAIN_NAMES + ["eta", "xi", "tau_eta", "gamma_xi", "delta_xi", "A_xi"]
    for q in range(NCH):
        solo = make_sampler(sim, T)
        solo.set_covariates(X, covariance_adj=True)
        solo.set_state(**states[q])
        solo.run(mask, T, seed=3, chain=10 + q)
        batch.select_chain(q)
        for nm in names:
            np.testing.assert_array_equal(batch.get_chain(nm), solo.get_chain(nm), err_msg=f"chain {q} {nm}")
        solo.close()
    batch.close()


def test_batched_multivariate_chains_equal_standalone_bitwise():
    import bayesfmmm_amd as bf
    S = bf.sampler
    rng = np.random.default_rng(6)
    n, P, K, M, T, NCH = 300, 12, 3, 2, 8, 4
    Y = rng.standard_normal((n, P))
    cfg = bf.default_config(model=bf.MODEL_MULTIVARIATE, K=K, n_eigen=M, tot_mcmc_iters=T)
    batch = bf.Sampler(cfg, Y, n_chains=NCH)
    for q in range(NCH):
        batch.select_chain(q)
        batch.init_state(1, 17, chain=q)
    batch.run(S.SWEEP_WARM, T, seed=17, chain=0)
    for q in range(NCH):
        cfg2 = bf.default_config(model=bf.MODEL_MULTIVARIATE, K=K, n_eigen=M, tot_mcmc_iters=T)
        solo = bf.Sampler(cfg2, Y)
        solo.init_state(1, 17, chain=q)
        solo.run(S.SWEEP_WARM, T, seed=17, chain=q)
        batch.select_chain(q)
        for nm in CHAIN_NAMES:
            np.testing.assert_array_equal(batch.get_chain(nm), solo.get_chain(nm), err_msg=f"chain {q} {nm}")
        solo.close()
    batch.close()


def test_batch_argument_checks():
    import bayesfmmm_amd as bf
    from bayesfmmm_amd import _lib
    sim = simulate_functional(n=24, M=2, sigma_sq=0.01, seed=1)
    b = make_sampler_batch(sim, 4, 2)
    with pytest.raises(_lib.BfmmmError, match="outside the batch"):
        b.select_chain(2)
    for q in range(2):
        b.select_chain(q)
        b.init_state(1, 1, chain=q)
    b.run(bf.SWEEP_WARM, 2, seed=1)
    with pytest.raises(_lib.BfmmmError, match="chain batch"):
        b.tempered_transition(bf.SWEEP_WARM, 1, 2, 0.5)
    b.close()
    with pytest.raises(_lib.BfmmmError, match="n_chains"):
        make_sampler_batch(sim, 4, 0)
