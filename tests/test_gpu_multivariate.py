"""Multivariate model (BMVMMM_*): the HIP path (band width 0 records, G_i = I) against the CPU
oracle's MV variants (UpdateNu.h:160, UpdatePhi.h:190, UpdateChi.h:138, UpdateSigma.h:127,
UpdateTau.h:47, CalculateLikelihood.h:137), and the package's multivariate example pipeline
(man/BMVMMM_warm_start.Rd: MVSim_data.RDS, K=2, n_eigen=2, 150 iterations)."""
import os

import numpy as np
import pytest

import oracle_lib as O
from gpu_parity import ORC_FIELD, push_state, rel_err
from rds_reader import read_rds

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def simulate_mv(n, P, K, M, sigma_sq, seed):
    # src/test-Nu.cpp:213-254 style: Y = Z (nu + sum_m chi_m Phi_m) + N(0, sigma_sq)
    rng = np.random.default_rng(seed)
    nu = rng.standard_normal((K, P)) * 2
    Phi = np.stack([(M - m) * 0.3 * rng.standard_normal((K, P)) for m in range(M)], axis=2)
    chi = rng.standard_normal((n, M))
    Z = rng.dirichlet(np.full(K, 2.0), size=n)
    Y = Z @ nu + np.einsum("ik,im,kpm->ip", Z, chi, Phi) + np.sqrt(sigma_sq) * rng.standard_normal((n, P))
    return dict(Y=Y, nu=nu, Phi=Phi, chi=chi, Z=Z, n=n, P=P, K=K, M=M, sigma_sq=sigma_sq)


def setup(seed, n=45, P=11, K=3, M=2, T=6):
    import bayesfmmm_amd as bf
    sim = simulate_mv(n, P, K, M, 0.05, seed)
    model = O.Model([sim["Y"][i] for i in range(n)], [np.eye(P)] * n, K, M, mv=True)
    ch = O.Chain(model, T)
    rng = np.random.default_rng(seed + 1)
    ch.nu[:, :, 0] = sim["nu"] + 0.2 * rng.standard_normal((K, P))
    ch.Phi[..., 0] = sim["Phi"] + 0.1 * rng.standard_normal((K, P, M))
    ch.chi[:, :, 0] = sim["chi"] + 0.2 * rng.standard_normal((n, M))
    ch.Z[:, :, 0] = rng.dirichlet(np.full(K, 2.0), size=n)
    ch.pi[:, 0] = rng.dirichlet(np.full(K, 5.0))
    ch.alpha3[0] = 4.0
    ch.delta[:, :, 0] = rng.gamma(2.0, 1.0, size=(K, M))
    ch.A[:, :, 0] = rng.gamma(2.0, 1.0, size=(K, 2))
    ch.gamma[..., 0] = rng.gamma(2.0, 0.7, size=(K, P, M))
    ch.tau[0, :] = rng.gamma(3.0, 0.5, size=K)
    ch.sigma[0] = 0.07
    cfg = bf.default_config(model=bf.MODEL_MULTIVARIATE, K=K, n_eigen=M, tot_mcmc_iters=T)
    smp = bf.Sampler(cfg, sim["Y"])
    push_state(smp, ch)
    return sim, model, ch, smp


@pytest.mark.parametrize("sweep", ["nu_z", "theta", "warm"])
def test_mv_trajectory_matches_oracle(sweep):
    import bayesfmmm_amd as bf
    S = bf.sampler
    T = 6
    sim, model, ch, smp = setup(seed=21, T=T)
    h = O.make_hyper(sim["K"])
    if sweep == "nu_z":
        ch.chi[:] = 0.0
        ch.Phi[:] = 0.0
        push_state(smp, ch)
        O.run_sweeps(model, h, ch, O.SWEEP_NU_Z, seed=5)
        smp.run(S.SWEEP_NU_Z, T, seed=5, phi_chi_zero=True)
        names = ["nu", "Z", "pi", "alpha_3", "tau", "sigma_sq", "loglik"]
    elif sweep == "theta":
        O.run_sweeps(model, h, ch, O.SWEEP_THETA, seed=5)
        smp.run(S.SWEEP_THETA, T, seed=5)
        names = ["Phi", "chi", "delta", "A", "gamma", "tau", "sigma_sq", "loglik"]
    else:
        O.run_sweeps(model, h, ch, O.SWEEP_WARM, seed=5)
        smp.run(S.SWEEP_WARM, T, seed=5)
        names = ["nu", "Phi", "chi", "Z", "pi", "alpha_3", "delta", "A", "gamma", "tau", "sigma_sq", "loglik"]
    for nm in names:
        got = smp.get_chain(nm)
        ref = getattr(ch, ORC_FIELD.get(nm, nm))
        assert rel_err(got, ref) < 1e-6, (sweep, nm, rel_err(got, ref))


def test_mv_quirks_integer_divisions():
    # odd P: loglik uses floor(P/2) log(2 pi sigma^2) per row (CalculateLikelihood.h:155) and sigma's shape uses
    # floor(n*P/2) (UpdateSigma.h:150); tau is stored inverted (UpdateTau.h:58)
    import bayesfmmm_amd as bf
    S = bf.sampler
    sim, model, ch, smp = setup(seed=33, n=7, P=5, T=3)
    smp.run(S.U_LOGLIK, 1, seed=1)
    ll = smp.get_chain("loglik", 1)[0]
    Y = sim["Y"]
    coef = ch.Z[:, :, 0] @ ch.nu[:, :, 0] + np.einsum("ik,im,kpm->ip", ch.Z[:, :, 0], ch.chi[:, :, 0], ch.Phi[..., 0])
    rss = ((Y - coef) ** 2).sum()
    s2 = ch.sigma[0]
    expect = -7 * (5 // 2) * np.log(2 * np.pi * s2) - rss / (2 * s2)
    assert abs(ll - expect) < 1e-9 * abs(expect)
    assert abs(O.calcLikelihood(model, ch, 0) - expect) < 1e-9 * abs(expect)


def test_mv_example_pipeline():
    from bayesfmmm_amd import api
    Y = np.asarray(read_rds(os.path.join(GOLD, "MVSim_data.RDS")))
    T, K, M = 150, 2, 2
    n, P = Y.shape
    est1 = api.BMVMMM_Nu_Z_multiple_try(T, 1, K, Y, M, seed=2)
    assert "B" not in est1 and est1["nu"].shape == (K, P, T) and est1["Z"].shape == (n, K, T)
    est2 = api.BMVMMM_Theta_est(T, 1, K, Y, M, est1, seed=3)
    assert est2["Phi"].shape == (K, P, M, T)
    mcmc = api.BMVMMM_warm_start(T, K, Y, M, est1, est2, seed=4)
    assert mcmc["chi"].shape == (n, M, T + 1) and np.isfinite(mcmc["loglik"][:T]).all()
    assert np.allclose(mcmc["Z"][:, :, :T].sum(axis=1), 1.0)
    with pytest.raises(Exception, match="'K' must be an integer greater than or equal to 2"):
        api.BMVMMM_Nu_Z_multiple_try(T, 1, 1, Y, M)


def test_mv_tempered_transitions_match_oracle():
    """BMVMMM warm start with tempered transitions (BFMMM.h:2677-2760, CalculateTTAcceptanceMV) against the oracle."""
    import bayesfmmm_amd as bf
    S = bf.sampler
    T, N_t, ntt, bN = 7, 3, 3, 0.5
    sim, model, ch, smp = setup(seed=21, T=T)
    h = O.make_hyper(sim["K"])
    push_state(smp, ch)
    logA_ref, acc_ref = O.run_warm_tt(model, h, ch, N_t, ntt, bN, seed=8)
    smp.run(S.SWEEP_WARM, 4, first_iter=0, seed=8)
    la3, a3 = smp.tempered_transition(S.SWEEP_WARM, 3, N_t, bN, seed=8)
    smp.run(S.SWEEP_WARM, 3, first_iter=4, seed=8)
    la6, a6 = smp.tempered_transition(S.SWEEP_WARM, 6, N_t, bN, seed=8)
    for la, a, i in ((la3, a3, 3), (la6, a6, 6)):
        assert int(a) == acc_ref[i] and abs(la - logA_ref[i]) < 1e-6 * max(1.0, abs(logA_ref[i])), (i, la, logA_ref[i])
    for nm in ["nu", "Phi", "chi", "Z", "pi", "alpha_3", "delta", "A", "gamma", "tau", "sigma_sq", "loglik"]:
        assert rel_err(smp.get_chain(nm), getattr(ch, ORC_FIELD.get(nm, nm))) < 2e-6, nm


# ---- multivariate model with covariates (BFMMM.h:5249, :5404, :5614, :6122; the MVCovariateAdj variants of
#      UpdateNu.h:443, UpdatePhi.h:540, UpdateChi.h:370, UpdateSigma.h:346, UpdateMixedMembership.h:866,
#      UpdateEta.h:203, UpdateXi.h:201, UpdateTau.h:106, CalculateLikelihood.h:300) --------------------------------
def setup_mv_cov(seed, n=37, P=9, K=3, M=2, D=2, T=5, covariance_adj=True):
    import bayesfmmm_amd as bf
    rng = np.random.default_rng(seed)
    sim = simulate_mv(n, P, K, M, 0.05, seed)
    X = rng.standard_normal((n, D))
    eta = 0.5 * rng.standard_normal((P, D, K))
    xi = 0.2 * rng.standard_normal((P, D, M, K)) * (1.0 if covariance_adj else 0.0)
    Y = sim["Y"].copy()
    for k in range(K):
        u = X @ eta[:, :, k].T
        for m in range(M):
            u = u + sim["chi"][:, m:m + 1] * (X @ xi[:, :, m, k].T)
        Y += sim["Z"][:, k:k + 1] * u
    model = O.Model([Y[i] for i in range(n)], [np.eye(P)] * n, K, M, X=X, mv=True)
    ch = O.Chain(model, T)
    ch.nu[:, :, 0] = sim["nu"] + 0.2 * rng.standard_normal((K, P))
    ch.Phi[..., 0] = sim["Phi"] + 0.1 * rng.standard_normal((K, P, M))
    ch.chi[:, :, 0] = sim["chi"] + 0.2 * rng.standard_normal((n, M))
    ch.Z[:, :, 0] = rng.dirichlet(np.full(K, 2.0), size=n)
    ch.pi[:, 0] = rng.dirichlet(np.full(K, 5.0))
    ch.alpha3[0] = 4.0
    ch.delta[:, :, 0] = rng.gamma(2.0, 1.0, size=(K, M))
    ch.A[:, :, 0] = rng.gamma(2.0, 1.0, size=(K, 2))
    ch.gamma[..., 0] = rng.gamma(2.0, 0.7, size=(K, P, M))
    ch.tau[0, :] = rng.gamma(3.0, 0.5, size=K)
    ch.sigma[0] = 0.07
    ch.eta[..., 0] = eta + 0.1 * rng.standard_normal((P, D, K))
    ch.xi[..., 0] = (xi + 0.05 * rng.standard_normal((P, D, M, K))) if covariance_adj else 0.0
    ch.tau_eta[..., 0] = rng.gamma(3.0, 0.5, size=(K, D))
    ch.gamma_xi[..., 0] = rng.gamma(2.0, 0.7, size=(P, D, M, K))
    ch.delta_xi[..., 0] = rng.gamma(2.0, 1.0, size=(K, M, D))
    ch.A_xi[..., 0] = rng.gamma(2.0, 1.0, size=(K, 2, D))
    cfg = bf.default_config(model=bf.MODEL_MULTIVARIATE, K=K, n_eigen=M, tot_mcmc_iters=T)
    smp = bf.Sampler(cfg, Y)
    smp.set_covariates(X, covariance_adj)
    push_state(smp, ch)
    from gpu_parity import oracle_slot
    smp.set_state(**{nm: oracle_slot(ch, nm, 0) for nm in ["eta", "xi", "tau_eta", "gamma_xi", "delta_xi", "A_xi"]})
    return dict(K=K, M=M, D=D, P=P, n=n), model, ch, smp


@pytest.mark.parametrize("which", ["Eta", "Xi", "TauEta", "Nu", "Phi", "Chi", "Z", "Sigma"])
def test_mv_single_update_with_covariates(which):
    import bayesfmmm_amd as bf
    from gpu_parity import oracle_slot
    S = bf.sampler
    dims, model, ch, smp = setup_mv_cov(seed=5)
    h = O.make_hyper(dims["K"])
    it, seed = 0, 17
    tilde_tau = np.cumprod(ch.delta[:, :, 0], axis=1)
    tilde_tau_xi = np.cumprod(ch.delta_xi[..., 0], axis=1)
    calls = {
        "Eta": (S.U_ETA, lambda: O.updateEta(model, ch, it, seed=seed), "eta"),
        "Xi": (S.U_XI, lambda: O.updateXi(model, ch, it, tilde_tau_xi, seed=seed), "xi"),
        "TauEta": (S.U_TAU_ETA, lambda: O.updateTauEta(model, ch, it, h.alpha_eta, h.beta_eta, seed=seed), "tau_eta"),
        "Nu": (S.U_NU, lambda: O.updateNu(model, ch, it, seed=seed), "nu"),
        "Phi": (S.U_PHI, lambda: O.updatePhi(model, ch, it, tilde_tau, seed=seed), "Phi"),
        "Chi": (S.U_CHI, lambda: O.updateChi(model, ch, it, seed=seed), "chi"),
        "Z": (S.U_Z, lambda: O.updateZ_PM(model, ch, it, h.a_Z_PM, seed=seed), "Z"),
        "Sigma": (S.U_SIGMA, lambda: O.updateSigma(model, ch, it, h.alpha_0, h.beta_0, seed=seed), "sigma_sq"),
    }
    mask, call, nm = calls[which]
    call()
    smp.run(mask, 1, seed=seed)
    got = smp.get_state(nm).reshape(-1, order="F")
    ref = oracle_slot(ch, nm, 0).reshape(-1, order="F")
    assert rel_err(got, ref) < 1e-8, (which, rel_err(got, ref))


@pytest.mark.parametrize("cov_adj", [False, True])
def test_mv_warm_trajectory_with_covariates(cov_adj):
    import bayesfmmm_amd as bf
    S = bf.sampler
    T = 5
    dims, model, ch, smp = setup_mv_cov(seed=12, T=T, covariance_adj=cov_adj)
    h = O.make_hyper(dims["K"])
    O.run_sweeps(model, h, ch, O.SWEEP_WARM, seed=6, covariance_adj=cov_adj)
    smp.run(S.SWEEP_WARM | S.COV_MEAN | (S.COV_XI if cov_adj else 0), T, seed=6)
    names = ["nu", "Phi", "chi", "Z", "pi", "alpha_3", "delta", "A", "gamma", "tau", "sigma_sq", "eta", "tau_eta", "loglik"]
    names += ["xi", "delta_xi", "A_xi", "gamma_xi"] if cov_adj else []
    for nm in names:
        got = smp.get_chain(nm)
        ref = getattr(ch, ORC_FIELD.get(nm, nm))
        assert rel_err(got, ref) < 1e-6, (cov_adj, nm, rel_err(got, ref))


@pytest.mark.parametrize("covariance_adj", [False, True])
def test_mv_pipeline_covariate_adjusted(covariance_adj, tmp_path):
    # man/BMVMMM_warm_start.Rd "Covariate Adj" example shape: MVSim_data.RDS with X = matrix(rnorm(n), n, 1)
    from bayesfmmm_amd import api
    Y = np.asarray(read_rds(os.path.join(GOLD, "MVSim_data.RDS")))
    T, K, M, D = 150, 2, 2, 1
    n, P = Y.shape
    X = np.random.default_rng(3).standard_normal((n, D))
    est1 = api.BMVMMM_Nu_Z_multiple_try(T, 1, K, Y, M, X=X, seed=2)
    assert est1["eta"].shape == (P, D, K, T) and np.abs(est1["eta"][..., -1]).max() > 0
    est2 = api.BMVMMM_Theta_est(T, 1, K, Y, M, est1, X=X, covariance_adj=covariance_adj, seed=3)
    assert est2["xi"].shape == (P, D, M, K, T) and (np.abs(est2["xi"]).max() > 0) == covariance_adj
    d = str(tmp_path) + "/"
    mcmc = api.BMVMMM_warm_start(T, K, Y, M, est1, est2, X=X, covariance_adj=covariance_adj, seed=4, dir=d, r_stored_iters=50)
    assert mcmc["eta"].shape == (P, D, K, 50) and np.isfinite(mcmc["loglik"]).all()
    eta = api.ReadFieldCube(d + "Eta2.txt")
    np.testing.assert_array_equal(eta[49, 0], mcmc["eta"][..., 48])     # saved draw p > 0 is slot thinning * p - 1
    np.testing.assert_array_equal(eta[0, 0], eta[1, 0])                  # ... so with thinning 1 slot 0 is saved twice
    np.testing.assert_array_equal(mcmc["eta"][..., 0], mcmc["eta"][..., 49])      # slot 0 was reset to the final state
    assert os.path.exists(d + "Xi0.txt") == covariance_adj
    with pytest.raises(Exception, match="'X' must be have 'n_funct' number of rows"):
        api.BMVMMM_Nu_Z_multiple_try(T, 1, K, Y, M, X=X[:5])


def test_mv_large_dim_with_covariates_matches_oracle():
    """dim = 40 (> 32: 64-lane groups, k_sweep_diag with the covariate-adjusted residual term, the eta / Xi block at band
    width 0) against the oracle."""
    import bayesfmmm_amd as bf
    S = bf.sampler
    T = 3
    dims, model, ch, smp = setup_mv_cov(seed=44, n=53, P=40, K=2, M=3, D=2, T=T, covariance_adj=True)
    h = O.make_hyper(dims["K"])
    O.run_sweeps(model, h, ch, O.SWEEP_WARM, seed=6, covariance_adj=True)
    smp.run(S.SWEEP_WARM | S.COV_MEAN | S.COV_XI, T, seed=6)
    for nm in ["nu", "Phi", "chi", "Z", "pi", "alpha_3", "delta", "A", "gamma", "tau", "sigma_sq", "eta", "tau_eta", "xi",
               "delta_xi", "A_xi", "gamma_xi", "loglik"]:
        assert rel_err(smp.get_chain(nm), getattr(ch, ORC_FIELD.get(nm, nm))) < 1e-6, nm
