"""Ports of the reference's statistical recovery tests, run against the
CPU oracle with the reference's own tolerances.  They pin the *logic* of each update: every
test simulates from known parameters, runs ONE update with everything else at truth and
compares the posterior median with the truth."""
import numpy as np

import oracle_lib as O
from simdata import simulate_functional, truth_chain


def test_updateNu_recovers_truth():
    # src/test-Nu.cpp:9-101, tolerance 0.3 (:863)
    sim = simulate_functional(n=20, M=5, sigma_sq=0.01, seed=1)
    T = 500
    model, ch = truth_chain(sim, T)
    ch.tau[:] = 0.1
    ch.nu[:] = np.random.default_rng(2).standard_normal(ch.nu.shape)
    for it in range(T):
        O.updateNu(model, ch, it)
    est = np.median(ch.nu[:, :, 300:], axis=2)
    assert np.abs(est - sim["nu"]).max() <= 0.3


def test_updateNu_tempered_recovers_truth():
    # src/test-Nu.cpp TestUpdateNuTempered: beta = 0.6, tolerance 0.6 (:881)
    sim = simulate_functional(n=20, M=5, sigma_sq=0.01, seed=3)
    T = 500
    model, ch = truth_chain(sim, T)
    ch.tau[:] = 0.1
    for it in range(T):
        O.updateNu(model, ch, it, beta_i=0.6)
    est = np.median(ch.nu[:, :, 300:], axis=2)
    assert np.abs(est - sim["nu"]).max() <= 0.6


def test_updatePhi_recovers_truth():
    # src/test-Phi.cpp:8-100: n=40, M=2 hmm-free version, sigma_sq=0.001, 250 its, median of 100-249, tol 0.3
    sim = simulate_functional(n=40, M=2, sigma_sq=0.001, seed=4, phi_scale=1.0)
    T = 250
    model, ch = truth_chain(sim, T)
    ch.Phi[:] = np.random.default_rng(5).standard_normal(ch.Phi.shape)
    tilde_tau = np.full((sim["K"], sim["M"]), 1.0)
    for it in range(T):
        O.updatePhi(model, ch, it, tilde_tau)
    est = np.median(ch.Phi[..., 100:], axis=3)
    assert np.abs(est - sim["Phi"]).max() <= 0.3


def test_updateChi_recovers_truth():
    # src/test-Chi.cpp:8-86: n=40, M=3, Phi_m = (3-m) randn, Z ~ Dir(10,10,10), sigma_sq=1e-4,
    # data re-simulated every iteration (:57-72), median of the last draws, tolerance 0.2 (:717)
    rng = np.random.default_rng(6)
    sim = simulate_functional(n=40, M=3, sigma_sq=1e-4, seed=6)
    K, P, M, n = sim["K"], sim["P"], sim["M"], sim["n"]
    Phi = np.stack([(M - m) * rng.standard_normal((K, P)) for m in range(M)], axis=2)
    Z = rng.dirichlet(np.full(K, 10.0), size=n)
    sim["Phi"], sim["Z"] = Phi, Z
    T = 300
    model, ch = truth_chain(sim, T)
    ch.chi[:, :, 0] = rng.standard_normal((n, M))
    B = sim["B"][0]
    coef = np.einsum("ik,kp->ip", Z, sim["nu"]) + np.einsum("ik,im,kpm->ip", Z, sim["chi"], Phi)
    for it in range(T):
        y = coef @ B.T + 0.01 * rng.standard_normal((n, B.shape[0]))
        model.y[:] = y.reshape(-1)
        O.updateChi(model, ch, it)
    est = np.median(ch.chi[:, :, 100:], axis=2)
    assert np.abs(est - sim["chi"]).max() <= 0.2


def test_updateSigma_recovers_truth():
    # src/test-Sigma.cpp:8-81: n=100, M=5, sigma_sq=0.5, 1000 its, median of all, tol 0.05 (:664)
    sim = simulate_functional(n=100, M=5, sigma_sq=0.5, seed=7)
    T = 300
    model, ch = truth_chain(sim, T)
    ch.sigma[:] = 1.0
    for it in range(T):
        O.updateSigma(model, ch, it, 1.0, 1.0)
    assert abs(np.median(ch.sigma) - 0.5) <= 0.05


def test_updateZ_recovers_truth():
    # src/test-PartialMembership.cpp:8-100: n=20, M=5, Phi_m = (5-m) 0.2 U(0,1), Z ~ Dir(10,10,10),
    # sigma_sq=1e-4, pi = (10,10,10), alpha_3 = 1, a_Z_PM=2000, 500 its, median of 200-499
    # renormalised, tolerance 0.02 (:923)
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
    for it in range(T):
        O.updateZ_PM(model, ch, it, 2000.0)
    est = np.median(ch.Z[:, :, 200:], axis=2)
    est /= est.sum(axis=1, keepdims=True)
    assert np.abs(est - sim["Z"]).max() <= 0.02


def test_updateTau_sigma_int_division_quirks():
    # UpdateTau.h:29 (nu.n_cols / 2) and UpdateSigma.h:49 (n_elem / 2) are integer divisions:
    # with P = 7 the shape is alpha + 3, not alpha + 3.5.  Check through the sampled mean.
    rng = np.random.default_rng(0)
    P, K, n = 7, 2, 3
    t = np.arange(0, 1000, 10.0)
    B = O.bspline_basis(t, [250, 500, 750], 3, [0, 990])
    ys = [rng.standard_normal(100) for _ in range(n)]
    model = O.Model(ys, [B] * n, K, 1)
    T = 4000
    ch = O.Chain(model, T)
    ch.nu[:] = rng.standard_normal((K, P))[:, :, None]
    ch.tau[:] = 1.0
    for it in range(T):
        O.updateTau(model, ch, it, 10.0, 1.0)
    Pm = O.pmat_rw1(P)
    for k in range(K):
        b = 1.0 + 0.5 * ch.nu[k, :, 0] @ Pm @ ch.nu[k, :, 0]
        assert abs(ch.tau[:, k].mean() - 13.0 / b) < 4 * np.sqrt(13.0) / b / np.sqrt(T)
        assert abs(ch.tau[:, k].mean() - 13.5 / b) > abs(ch.tau[:, k].mean() - 13.0 / b)


def test_hyper_updates_run_and_recover():
    # src/test-Phi.cpp updateGamma/updateDelta/updateA (tolerances 0.5 / 2 / 0.5 on medians of
    # draws around the generating values) -- here: the conjugate gamma posteriors have the
    # closed-form means the reference's formulas imply.
    sim = simulate_functional(n=10, M=3, sigma_sq=0.01, seed=11)
    T = 3000
    model, ch = truth_chain(sim, T)
    K, P, M = sim["K"], sim["P"], sim["M"]
    for it in range(T):
        O.updateGamma(model, ch, it, 3.0)
    phi = sim["Phi"]
    # gamma(i,l,j) ~ Gamma((nu+1)/2, scale 2/(nu + tilde_tau * phi^2)), delta = 1 -> tilde_tau = 1
    expect = ((3.0 + 1) / 2) * 2 / (3.0 + phi ** 2)
    got = ch.gamma.mean(axis=3)
    assert np.abs(got / expect - 1).max() < 0.08
    # delta: first column shape a1 + P*M/2, others a2 + P*(M-i)/2; all means finite and positive
    for it in range(T):
        O.updateDelta(model, ch, it)
    assert (ch.delta > 0).all() and np.isfinite(ch.delta).all()
    h = O.make_hyper(K)
    for it in range(T):
        O.updateA(model, ch, it, h)
    assert (ch.A > 0).all()
    # the MH chain for A moves (acceptance neither 0 nor 1)
    acc = np.mean(np.diff(ch.A[0, 0, :]) != 0)
    assert 0.05 < acc < 0.99
