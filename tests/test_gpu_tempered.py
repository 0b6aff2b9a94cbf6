"""Tempered transitions: the device block against the oracle's line-faithful restatement of
BFMMM.h:1556-1672 / CalculateTTAcceptance.h under the same keyed variates."""
import numpy as np
import pytest

import oracle_lib as O
from gpu_parity import ORC_FIELD, make_sampler, push_state, random_state, rel_err
from simdata import simulate_functional, truth_chain

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def bf():
    import bayesfmmm_amd as bf
    return bf


def _setup(seed, T, n=41):
    sim = simulate_functional(n=n, M=3, sigma_sq=0.01, seed=seed, ragged=True)
    model, ch = truth_chain(sim, T)
    random_state(sim, ch, seed + 100)
    smp = make_sampler(sim, T)
    return sim, model, ch, smp


@pytest.mark.parametrize("N_t,beta_N_t,n_temp_trans", [(3, 0.6, 2), (2, 0.9, 3), (4, 0.3, 4)])
def test_tempered_transitions_match_oracle(bf, N_t, beta_N_t, n_temp_trans):
    S = bf.sampler
    T = 9
    sim, model, ch, smp = _setup(seed=11 + N_t, T=T)
    h = O.make_hyper(sim["K"])
    push_state(smp, ch)
    logA_ref, acc_ref = O.run_warm_tt(model, h, ch, N_t, n_temp_trans, beta_N_t, seed=5)
    logA, acc = np.full(T, np.nan), np.full(T, -1)
    i0 = 0
    for i in range(T):
        if i > 0 and i % n_temp_trans == 0:
            smp.run(S.SWEEP_WARM, i + 1 - i0, first_iter=i0, seed=5)
            la, a = smp.tempered_transition(S.SWEEP_WARM, i, N_t, beta_N_t, seed=5)
            logA[i], acc[i] = la, int(a)
            i0 = i + 1
    if i0 < T:
        smp.run(S.SWEEP_WARM, T - i0, first_iter=i0, seed=5)
    blocks = [i for i in range(T) if i > 0 and i % n_temp_trans == 0]
    assert len(blocks) >= 2
    for i in blocks:
        assert acc[i] == acc_ref[i], (i, logA[i], logA_ref[i])
        assert abs(logA[i] - logA_ref[i]) < 1e-6 * max(1.0, abs(logA_ref[i])), (i, logA[i], logA_ref[i])
    for nm in ["nu", "Phi", "chi", "Z", "pi", "alpha_3", "delta", "A", "gamma", "tau", "sigma_sq", "loglik"]:
        got = smp.get_chain(nm)
        ref = getattr(ch, ORC_FIELD.get(nm, nm))
        assert rel_err(got, ref) < 2e-6, (nm, rel_err(got, ref))


def test_ladder_and_argument_checks(bf):
    lad = O.beta_ladder(4, 0.5)
    g = 0.5 ** 0.25
    np.testing.assert_allclose(lad, [1.0, g, g * g, g ** 3], rtol=1e-15)      # the reference's ladder never reaches beta_N_t
    sim, model, ch, smp = _setup(seed=3, T=4)
    with pytest.raises(bf._lib.BfmmmError):
        smp.tempered_transition(bf.sampler.SWEEP_WARM, 7, 2, 0.5)


@pytest.mark.parametrize("covariance_adj,mv", [(False, False), (True, False), (True, True)])
def test_tempered_transitions_with_covariates_match_oracle(bf, covariance_adj, mv):
    """The tempered block of the covariate-adjusted drivers (BFMMM.h:4313-4440 MeanAdj, :4897-5084 Mean_CovAdj; MV
    :5880-6050, :6420-6590; CalculateTTAcceptance.h:195-290, :296-385): eta, tau_eta (and Xi, delta_xi, A_xi, gamma_xi)
    are tempered, saved and restored with the rest of the state."""
    S = bf.sampler
    T, N_t, ntt, bN = 7, 3, 3, 0.6
    if mv:
        from test_gpu_multivariate import setup_mv_cov
        dims, model, ch, smp = setup_mv_cov(seed=31, T=T, covariance_adj=covariance_adj)
        K = dims["K"]
    else:
        from test_gpu_covariates import setup as setup_cov
        sim, model, ch, smp = setup_cov(seed=23, T=T, covariance_adj=covariance_adj)
        K = sim["K"]
    h = O.make_hyper(K)
    logA_ref, acc_ref = O.run_warm_tt(model, h, ch, N_t, ntt, bN, seed=8, covariance_adj=covariance_adj)
    mask = S.SWEEP_WARM | S.COV_MEAN | (S.COV_XI if covariance_adj else 0)
    smp.run(mask, 4, first_iter=0, seed=8)
    la3, a3 = smp.tempered_transition(mask, 3, N_t, bN, seed=8)
    smp.run(mask, 3, first_iter=4, seed=8)
    la6, a6 = smp.tempered_transition(mask, 6, N_t, bN, seed=8)
    for la, a, i in ((la3, a3, 3), (la6, a6, 6)):
        assert int(a) == acc_ref[i] and abs(la - logA_ref[i]) < 1e-6 * max(1.0, abs(logA_ref[i])), (i, la, logA_ref[i])
    names = ["nu", "Phi", "chi", "Z", "pi", "alpha_3", "delta", "A", "gamma", "tau", "sigma_sq", "eta", "tau_eta", "loglik"]
    names += ["xi", "delta_xi", "A_xi", "gamma_xi"] if covariance_adj else []
    for nm in names:
        assert rel_err(smp.get_chain(nm), getattr(ch, ORC_FIELD.get(nm, nm))) < 2e-6, nm
