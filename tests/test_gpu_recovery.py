"""The reference's statistical recovery tests (src/test-Nu.cpp, test-Phi.cpp, test-Chi.cpp,
test-Sigma.cpp, test-PartialMembership.cpp, test-Eta.cpp, test-Xi.cpp) run on the DEVICE path with the reference's own
tolerances: data simulated from known parameters, one update repeated with everything else held at the truth, posterior
median against the truth.  (tests/test_oracle_recovery.py runs the same checks on the CPU oracle.)"""
import numpy as np
import pytest

from gpu_parity import make_sampler, push_state
from simdata import simulate_functional, truth_chain

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def S():
    import bayesfmmm_amd as bf
    return bf.sampler


def test_updateNu_recovers_truth(S):
    # src/test-Nu.cpp:9-101, tolerance 0.3 (:863)
    sim = simulate_functional(n=20, M=5, sigma_sq=0.01, seed=1)
    T = 500
    model, ch = truth_chain(sim, T)
    ch.tau[:] = 0.1
    ch.nu[:, :, 0] = np.random.default_rng(2).standard_normal(ch.nu.shape[:2])
    smp = make_sampler(sim, T)
    push_state(smp, ch)
    smp.run(S.U_NU, T, seed=3)
    est = np.median(smp.get_chain("nu")[:, :, 300:], axis=2)
    assert np.abs(est - sim["nu"]).max() <= 0.3


def test_updateNu_tempered_recovers_truth(S):
    # src/test-Nu.cpp TestUpdateNuTempered: beta = 0.6, tolerance 0.6 (:881)
    sim = simulate_functional(n=20, M=5, sigma_sq=0.01, seed=3)
    T = 500
    model, ch = truth_chain(sim, T)
    ch.tau[:] = 0.1
    smp = make_sampler(sim, T)
    push_state(smp, ch)
    smp.run(S.U_NU, T, seed=3, beta=0.6)
    est = np.median(smp.get_chain("nu")[:, :, 300:], axis=2)
    assert np.abs(est - sim["nu"]).max() <= 0.6


def test_updatePhi_recovers_truth(S):
    # src/test-Phi.cpp:8-100: sigma_sq = 0.001, 250 iterations, median of 100-249, tolerance 0.3
    sim = simulate_functional(n=40, M=2, sigma_sq=0.001, seed=4, phi_scale=1.0)
    T = 250
    model, ch = truth_chain(sim, T)
    ch.Phi[..., 0] = np.random.default_rng(5).standard_normal(ch.Phi.shape[:3])
    ch.delta[:] = 1.0                                    # tilde_tau = 1
    smp = make_sampler(sim, T)
    push_state(smp, ch)
    smp.run(S.U_PHI, T, seed=3)
    est = np.median(smp.get_chain("Phi")[..., 100:], axis=3)
    assert np.abs(est - sim["Phi"]).max() <= 0.3


def test_updateChi_recovers_truth(S):
    # src/test-Chi.cpp:8-86 (fixed data instead of re-simulating it every iteration), tolerance 0.2 (:717)
    rng = np.random.default_rng(6)
    sim = simulate_functional(n=40, M=3, sigma_sq=1e-4, seed=6)
    K, P, M, n = sim["K"], sim["P"], sim["M"], sim["n"]
    Phi = np.stack([(M - m) * rng.standard_normal((K, P)) for m in range(M)], axis=2)
    Z = rng.dirichlet(np.full(K, 10.0), size=n)
    B = sim["B"][0]
    coef = np.einsum("ik,kp->ip", Z, sim["nu"]) + np.einsum("ik,im,kpm->ip", Z, sim["chi"], Phi)
    Y = coef @ B.T + 0.01 * rng.standard_normal((n, B.shape[0]))
    sim["Phi"], sim["Z"], sim["y"] = Phi, Z, [Y[i] for i in range(n)]
    T = 300
    model, ch = truth_chain(sim, T)
    ch.chi[:, :, 0] = rng.standard_normal((n, M))
    smp = make_sampler(sim, T)
    push_state(smp, ch)
    smp.run(S.U_CHI, T, seed=3)
    est = np.median(smp.get_chain("chi")[:, :, 100:], axis=2)
    assert np.abs(est - sim["chi"]).max() <= 0.2


def test_updateSigma_recovers_truth(S):
    # src/test-Sigma.cpp:8-81: n = 100, M = 5, sigma_sq = 0.5, median of all draws, tolerance 0.05 (:664)
    sim = simulate_functional(n=100, M=5, sigma_sq=0.5, seed=7)
    T = 300
    model, ch = truth_chain(sim, T)
    ch.sigma[:] = 1.0
    smp = make_sampler(sim, T, alpha_0=1.0, beta_0=1.0)
    push_state(smp, ch)
    smp.run(S.U_SIGMA, T, seed=3)
    assert abs(np.median(smp.get_chain("sigma_sq")) - 0.5) <= 0.05


def test_updateZ_recovers_truth(S):
    # src/test-PartialMembership.cpp:8-100: Z ~ Dir(10,10,10), sigma_sq = 1e-4, pi = (10,10,10), alpha_3 = 1,
    # a_Z_PM = 2000, 500 iterations, median of 200-499 renormalised, tolerance 0.02 (:923)
    rng = np.random.default_rng(8)
    sim = simulate_functional(n=20, M=5, sigma_sq=1e-4, seed=8, phi_scale=0.2, alpha_dir=1e9)
    K, n = sim["K"], sim["n"]
    Z = rng.dirichlet(np.full(K, 10.0), size=n)
    coef_old = np.einsum("ik,kp->ip", sim["Z"], sim["nu"]) + np.einsum("ik,im,kpm->ip", sim["Z"], sim["chi"], sim["Phi"])
    coef_new = np.einsum("ik,kp->ip", Z, sim["nu"]) + np.einsum("ik,im,kpm->ip", Z, sim["chi"], sim["Phi"])
    B = sim["B"][0]
    for i in range(n):                       # move the noiseless part of y to the new Z, keep the noise
        sim["y"][i] = sim["y"][i] + B @ (coef_new[i] - coef_old[i])
    sim["Z"] = Z
    T = 500
    model, ch = truth_chain(sim, T)
    ch.pi[:] = 10.0
    ch.alpha3[:] = 1.0
    ch.Z[:, :, 0] = rng.dirichlet(np.full(K, 10.0), size=n)
    smp = make_sampler(sim, T, a_Z_PM=2000.0)
    push_state(smp, ch)
    smp.run(S.U_Z, T, seed=3)
    est = np.median(smp.get_chain("Z")[:, :, 200:], axis=2)
    est /= est.sum(axis=1, keepdims=True)
    assert np.abs(est - sim["Z"]).max() <= 0.02


def test_updateEta_and_Xi_recover_truth(S):
    # src/test-Eta.cpp:9-130 / test-Xi.cpp: covariate effects with everything else at the truth, tolerances 0.3 / 0.4
    sim = simulate_functional(n=60, M=2, sigma_sq=0.001, seed=11, D=2)
    T = 300
    model, ch = truth_chain(sim, T)
    rng = np.random.default_rng(12)
    ch.eta[..., 0] = rng.standard_normal(ch.eta.shape[:3])
    ch.xi[..., 0] = sim["xi"]
    ch.tau_eta[:] = 0.1
    ch.gamma_xi[:] = 1.0
    ch.delta_xi[:] = 1.0
    ch.A_xi[:] = 1.0
    from gpu_parity import oracle_slot
    cov = ["eta", "xi", "tau_eta", "gamma_xi", "delta_xi", "A_xi"]
    smp = make_sampler(sim, T)
    smp.set_covariates(sim["X"], True)
    push_state(smp, ch)
    smp.set_state(**{nm: oracle_slot(ch, nm, 0) for nm in cov})
    smp.run(S.U_ETA, T, seed=3)
    est = np.median(smp.get_chain("eta")[..., 100:], axis=3)
    assert np.abs(est - sim["eta"]).max() <= 0.3
    ch.eta[..., 0] = sim["eta"]
    ch.xi[..., 0] = rng.standard_normal(ch.xi.shape[:4]) * 0.5
    push_state(smp, ch)
    smp.set_state(**{nm: oracle_slot(ch, nm, 0) for nm in cov})
    smp.run(S.U_XI, T, seed=4)
    est = np.median(smp.get_chain("xi")[..., 100:], axis=4)
    assert np.abs(est - sim["xi"]).max() <= 0.4
