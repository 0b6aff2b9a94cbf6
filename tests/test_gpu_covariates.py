"""Covariate-adjusted functional model (eta: UpdateEta.h:28-94, Xi: UpdateXi.h:26-93 and their
hyper-parameters; *CovariateAdj variants of nu, Phi, chi, Z, sigma^2, loglik): HIP path against the
CPU oracle, single updates and short trajectories of the three covariate drivers
(BFMMM.h:3741-3780, 3944-4010, 4809-4894)."""
import numpy as np
import pytest

import oracle_lib as O
from gpu_parity import ORC_FIELD, STATE_NAMES, make_sampler, oracle_slot, push_state, random_state, rel_err
from simdata import simulate_functional, truth_chain

pytestmark = pytest.mark.gpu
COV_NAMES = ["eta", "xi", "tau_eta", "gamma_xi", "delta_xi", "A_xi"]


def setup(seed, D=2, n=29, M=2, T=5, covariance_adj=True):
    sim = simulate_functional(n=n, M=M, sigma_sq=0.01, seed=seed, ragged=True, D=D)
    model, ch = truth_chain(sim, T)
    random_state(sim, ch, seed + 50)
    rng = np.random.default_rng(seed + 7)
    K, P = sim["K"], sim["P"]
    ch.eta[..., 0] = sim["eta"] + 0.2 * rng.standard_normal((P, D, K))
    ch.xi[..., 0] = (sim["xi"] + 0.05 * rng.standard_normal((P, D, M, K))) if covariance_adj else 0.0
    ch.tau_eta[..., 0] = rng.gamma(3.0, 0.5, size=(K, D))
    ch.gamma_xi[..., 0] = rng.gamma(2.0, 0.7, size=(P, D, M, K))
    ch.delta_xi[..., 0] = rng.gamma(2.0, 1.0, size=(K, M, D))
    ch.A_xi[..., 0] = rng.gamma(2.0, 1.0, size=(K, 2, D))
    smp = make_sampler(sim, T)
    smp.set_covariates(sim["X"], covariance_adj)
    push_state(smp, ch)
    smp.set_state(**{nm: oracle_slot(ch, nm, 0) for nm in COV_NAMES})
    return sim, model, ch, smp


@pytest.mark.parametrize("which", ["Eta", "Xi", "TauEta", "DeltaXi", "AXi", "GammaXi", "Nu", "Phi", "Chi", "Z", "Sigma"])
def test_single_update_cov(which):
    import bayesfmmm_amd as bf
    S = bf.sampler
    sim, model, ch, smp = setup(seed=3)
    h = O.make_hyper(sim["K"])
    it, seed = 0, 41
    K, M, D = sim["K"], sim["M"], sim["D"]
    tilde_tau = np.cumprod(ch.delta[:, :, 0], axis=1)
    tilde_tau_xi = np.cumprod(ch.delta_xi[..., 0], axis=1)
    calls = {
        "Eta": (S.U_ETA, lambda: O.updateEta(model, ch, it, seed=seed), ["eta"]),
        "Xi": (S.U_XI, lambda: O.updateXi(model, ch, it, tilde_tau_xi, seed=seed), ["xi"]),
        "TauEta": (S.U_TAU_ETA, lambda: O.updateTauEta(model, ch, it, h.alpha_eta, h.beta_eta, seed=seed), ["tau_eta"]),
        "DeltaXi": (S.U_DELTA_XI, lambda: O.updateDeltaXi(model, ch, it, seed=seed), ["delta_xi"]),
        "AXi": (S.U_A_XI, lambda: O.updateAXi(model, ch, it, h, seed=seed), ["A_xi"]),
        "GammaXi": (S.U_GAMMA_XI, lambda: O.updateGammaXi(model, ch, it, h.nu_1, seed=seed), ["gamma_xi"]),
        "Nu": (S.U_NU, lambda: O.updateNu(model, ch, it, seed=seed), ["nu"]),
        "Phi": (S.U_PHI, lambda: O.updatePhi(model, ch, it, tilde_tau, seed=seed), ["Phi"]),
        "Chi": (S.U_CHI, lambda: O.updateChi(model, ch, it, seed=seed), ["chi"]),
        "Z": (S.U_Z, lambda: O.updateZ_PM(model, ch, it, h.a_Z_PM, seed=seed), ["Z"]),
        "Sigma": (S.U_SIGMA, lambda: O.updateSigma(model, ch, it, h.alpha_0, h.beta_0, seed=seed), ["sigma_sq"]),
    }
    mask, call, names = calls[which]
    call()
    smp.run(mask, 1, seed=seed)
    for nm in names:
        got = smp.get_state(nm).reshape(-1, order="F")
        ref = oracle_slot(ch, nm, 0).reshape(-1, order="F")
        assert rel_err(got, ref) < 1e-8, (which, nm, rel_err(got, ref))


@pytest.mark.parametrize("sweep,cov_adj", [("nu_z", False), ("theta", True), ("warm", False), ("warm", True)])
def test_cov_trajectory(sweep, cov_adj):
    import bayesfmmm_amd as bf
    S = bf.sampler
    T = 4
    sim, model, ch, smp = setup(seed=9, T=T, covariance_adj=cov_adj)
    h = O.make_hyper(sim["K"])
    xi_mask = S.COV_XI if cov_adj else 0
    if sweep == "nu_z":
        ch.chi[:] = 0.0
        ch.Phi[:] = 0.0
        push_state(smp, ch)
        O.run_sweeps(model, h, ch, O.SWEEP_NU_Z, seed=6, covariance_adj=cov_adj)
        smp.run(S.SWEEP_NU_Z | S.COV_MEAN, T, seed=6, phi_chi_zero=True)
        names = ["nu", "Z", "pi", "alpha_3", "tau", "sigma_sq", "eta", "tau_eta", "loglik"]
    elif sweep == "theta":
        O.run_sweeps(model, h, ch, O.SWEEP_THETA, seed=6, covariance_adj=cov_adj)
        smp.run(S.SWEEP_THETA | S.U_TAU_ETA | xi_mask, T, seed=6)
        names = ["Phi", "chi", "delta", "A", "gamma", "tau", "sigma_sq", "tau_eta", "xi", "delta_xi", "A_xi", "gamma_xi", "loglik"]
    else:
        O.run_sweeps(model, h, ch, O.SWEEP_WARM, seed=6, covariance_adj=cov_adj)
        smp.run(S.SWEEP_WARM | S.COV_MEAN | xi_mask, T, seed=6)
        names = STATE_NAMES + ["eta", "tau_eta", "loglik"] + (["xi", "delta_xi", "A_xi", "gamma_xi"] if cov_adj else [])
    for nm in names:
        got = smp.get_chain(nm)
        ref = getattr(ch, ORC_FIELD.get(nm, nm))
        assert got.shape == ref.shape, (nm, got.shape, ref.shape)
        assert rel_err(got, ref) < 1e-6, (sweep, cov_adj, nm, rel_err(got, ref))
