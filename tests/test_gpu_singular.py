"""Rank-deficient conditional precisions.  The reference takes arma::pinv for nu and eta (UpdateNu.h:67-68,
UpdateEta.h:85-86): a cluster without members (Z[:, k] == 0) leaves the precision tau * P_mat of rank P - 1, which the
reference accepts (pinv drops the null direction, arma::mvnrnd draws through the eigen-decomposition).  The device takes
the same route (factor_pinv, csrc/factor_core.hpp) by the specification it shares with the oracle (oracle/linalg.c):
parity to the single-update tolerance, and no run aborts on such a state -- including a sampler whose state was never
set (all-zero Z) and multi-try batches."""
import numpy as np
import pytest

import oracle_lib as O
from gpu_parity import make_sampler, oracle_slot, push_state, random_state, rel_err
from simdata import simulate_functional, truth_chain

pytestmark = pytest.mark.gpu


def _setup(n=37, M=2, T=6, seed=4, empty=1, **kw):
    sim = simulate_functional(n=n, M=M, sigma_sq=0.01, seed=seed, **kw)
    model, ch = truth_chain(sim, T)
    random_state(sim, ch, seed + 50)
    Z = ch.Z[:, :, 0].copy()
    Z[:, empty] = 0.0                       # nobody belongs to cluster `empty`
    Z /= Z.sum(axis=1, keepdims=True)
    ch.Z[:, :, 0] = Z
    smp = make_sampler(sim, T)
    push_state(smp, ch)
    return sim, model, ch, smp


def test_nu_update_with_an_empty_cluster_matches_oracle():
    import bayesfmmm_amd as bf
    sim, model, ch, smp = _setup()
    O.updateNu(model, ch, 0, seed=31)
    smp.run(bf.sampler.U_NU, 1, first_iter=0, seed=31)
    got, ref = smp.get_state("nu"), oracle_slot(ch, "nu", 0)
    assert np.isfinite(got).all()
    assert rel_err(got, ref) < 1e-8, rel_err(got, ref)
    # the empty cluster's draw lives in the prior's range space: no component along the null vector (the constant)
    assert abs(got[1].sum()) < 1e-9 * np.abs(got[1]).sum()


def test_sweeps_from_a_state_with_an_empty_cluster_match_oracle():
    import bayesfmmm_amd as bf
    S = bf.sampler
    sim, model, ch, smp = _setup(T=6)
    h = O.make_hyper(sim["K"])
    # Z held fixed: cluster 1 stays empty through all six sweeps (nu_1, tau_1 are drawn from the singular conditional)
    mask = S.U_PHI | S.U_DELTA | S.U_A | S.U_GAMMA | S.U_NU | S.U_TAU | S.U_SIGMA | S.U_CHI | S.U_LOGLIK
    for it in range(6):
        if it > 0:        # the blocks this sweep does not update stay where they are (the oracle's updates carry their own)
            ch.Z[:, :, it], ch.pi[:, it], ch.alpha3[it] = ch.Z[:, :, it - 1], ch.pi[:, it - 1], ch.alpha3[it - 1]
        tilde_tau = np.cumprod(ch.delta[:, :, it], axis=1)
        O.updatePhi(model, ch, it, tilde_tau, seed=8)
        O.updateDelta(model, ch, it, seed=8)
        O.updateA(model, ch, it, h, seed=8)
        O.updateGamma(model, ch, it, h.nu_1, seed=8)
        O.updateNu(model, ch, it, seed=8)
        O.updateTau(model, ch, it, h.alpha_nu, h.beta_nu, seed=8)
        O.updateSigma(model, ch, it, h.alpha_0, h.beta_0, seed=8)
        O.updateChi(model, ch, it, seed=8)
        ch.loglik[it] = O.calcLikelihood(model, ch, it)
    smp.run(mask, 6, seed=8)
    for nm in ["nu", "Phi", "tau", "sigma_sq", "chi", "loglik"]:
        ref = getattr(ch, {"sigma_sq": "sigma"}.get(nm, nm))
        assert rel_err(smp.get_chain(nm), ref) < 2e-6, (nm, rel_err(smp.get_chain(nm), ref))
    # and the full warm-start sweep (Z moves away from the empty column through the alpha <= 0 -> 10 rule, Distributions.h:24-28)
    sim, model, ch, smp = _setup(T=5, seed=6)
    O.run_sweeps(model, h, ch, O.SWEEP_WARM, seed=5)
    smp.run(S.SWEEP_WARM, 5, seed=5)
    for nm in ["nu", "Z", "sigma_sq", "loglik"]:
        ref = getattr(ch, {"sigma_sq": "sigma"}.get(nm, nm))
        assert rel_err(smp.get_chain(nm), ref) < 2e-6, nm


def test_eta_update_with_an_empty_cluster_matches_oracle():
    import bayesfmmm_amd as bf
    S = bf.sampler
    sim = simulate_functional(n=40, M=2, sigma_sq=0.01, seed=12)
    X = np.random.default_rng(3).standard_normal((sim["n"], 2))
    T = 3
    model = O.Model(sim["y"], sim["B"], sim["K"], sim["M"], X=X)
    ch = O.Chain(model, T)
    random_state(sim, ch, 9)
    Z = ch.Z[:, :, 0].copy()
    Z[:, 0] = 0.0
    ch.Z[:, :, 0] = Z / Z.sum(axis=1, keepdims=True)
    ch.eta[..., 0] = 0.2 * np.random.default_rng(4).standard_normal(ch.eta[..., 0].shape)
    ch.tau_eta[..., 0] = 1.3
    smp = make_sampler(sim, T)
    smp.set_covariates(X, covariance_adj=False)
    push_state(smp, ch)
    smp.set_state(eta=ch.eta[..., 0], tau_eta=ch.tau_eta[..., 0])
    O.updateEta(model, ch, 0, seed=21)
    smp.run(S.U_ETA, 1, seed=21)
    got = smp.get_state("eta")
    assert np.isfinite(got).all() and rel_err(got, ch.eta[..., 0]) < 1e-8, rel_err(got, ch.eta[..., 0])


def test_runs_do_not_abort_on_degenerate_states():
    import bayesfmmm_amd as bf
    S = bf.sampler
    sim = simulate_functional(n=24, M=2, sigma_sq=0.01, seed=1)
    # a sampler whose state was never set: Z = 0 everywhere, every nu precision is tau P_mat
    smp = make_sampler(sim, 8)
    smp.run(S.SWEEP_WARM, 8, seed=2)
    assert np.isfinite(smp.get_chain("nu")).all() and np.isfinite(smp.get_chain("loglik")).all()
    # a multi-try batch in which ONE chain has an empty cluster: every chain finishes, all are finite
    b = bf.Sampler(bf.default_config(model=bf.MODEL_FUNCTIONAL, K=sim["K"], n_eigen=sim["M"], basis_degree=3, tot_mcmc_iters=8),
                   sim["y"], sim["t"], sim["internal_knots"], sim["boundary_knots"], n_chains=3)
    for q in range(3):
        b.select_chain(q)
        b.init_state(1, 3, chain=q)
    b.select_chain(1)
    Z = b.get_state("Z")
    Z[:, 2] = 0.0
    b.set_state(Z=Z / Z.sum(axis=1, keepdims=True))
    b.run(S.U_PHI | S.U_NU | S.U_TAU | S.U_SIGMA | S.U_CHI | S.U_LOGLIK, 8, seed=3)
    for q in range(3):
        b.select_chain(q)
        assert np.isfinite(b.get_chain("nu")).all() and np.isfinite(b.get_chain("loglik")).all(), q
