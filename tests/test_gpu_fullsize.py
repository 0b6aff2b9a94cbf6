"""Full-size runs (BASELINE.json configs 2 and 4) checked through size-independent properties:
simplex rows, bit-reproducibility, the log-likelihood recomputed from the state on the host, and
agreement of the two independent residual-sum paths on the device (the quadratic form inside
k_sweep vs the per-curve pass of k_curve_chi)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def host_loglik_functional(w, nu, Phi, chi, Z, s2):
    B = w["B"][0]
    Y = np.stack(w["y"])
    coef = Z @ nu + np.einsum("ik,im,kpm->ip", Z, chi, Phi)
    rss = ((Y - coef @ B.T) ** 2).sum()
    N = Y.size
    return -N * (0.9189385332046727 + 0.5 * np.log(s2)) - rss / (2 * s2), rss


def test_config2_properties():
    import bayesfmmm_amd as bf
    from bench import make_config2
    S = bf.sampler
    w = make_config2()
    T = 12
    cfg = bf.default_config(model=bf.MODEL_FUNCTIONAL, K=3, n_eigen=6, basis_degree=3, tot_mcmc_iters=T)
    smp = bf.Sampler(cfg, w["y"], w["t"], w["internal_knots"], w["boundary_knots"])
    smp.set_state(**w["state"])
    smp.run(S.SWEEP_WARM, T, seed=7)
    Z = smp.get_chain("Z"); nu = smp.get_chain("nu"); Phi = smp.get_chain("Phi"); chi = smp.get_chain("chi")
    s2 = smp.get_chain("sigma_sq"); ll = smp.get_chain("loglik")
    assert np.abs(Z.sum(axis=1) - 1).max() < 1e-12 and (Z > 0).all()
    for t in (0, T - 1):
        ref, _ = host_loglik_functional(w, nu[:, :, t], Phi[..., t], chi[:, :, t], Z[:, :, t], s2[t])
        assert abs(ll[t] - ref) < 1e-8 * abs(ref), (t, ll[t], ref)
    assert 0.005 < s2[-1] < 0.02            # chain started at the generating values stays there
    # bit-reproducible
    smp.set_state(**w["state"])
    smp.run(S.SWEEP_WARM, T, seed=7)
    np.testing.assert_array_equal(smp.get_chain("nu"), nu)
    np.testing.assert_array_equal(smp.get_chain("chi"), chi)
    # two residual paths: sigma-only sweep sets rss through the H quadratic form; loglik-only recomputes per curve
    smp.set_state(**w["state"])
    smp.run(S.U_SIGMA | S.U_LOGLIK, 1, seed=3)
    ll_quad = smp.get_chain("loglik", 1)[0]
    s2_new = smp.get_state("sigma_sq")[0]
    smp.run(S.U_LOGLIK, 1, first_iter=1, seed=3)
    ll_curve = smp.get_chain("loglik", 2)[1]
    assert abs(ll_quad - ll_curve) < 1e-9 * abs(ll_curve)
    ref, _ = host_loglik_functional(w, w["state"]["nu"], w["state"]["Phi"], w["state"]["chi"], w["state"]["Z"], s2_new)
    assert abs(ll_curve - ref) < 1e-8 * abs(ref)


def test_config4_multivariate_properties():
    import bayesfmmm_amd as bf
    S = bf.sampler
    rng = np.random.default_rng(4)
    n, P, K, M = 8192, 50, 4, 8
    nu = rng.standard_normal((K, P)) * 2
    Phi = np.stack([(M - m) / M * 0.5 * rng.standard_normal((K, P)) for m in range(M)], axis=2)
    chi = rng.standard_normal((n, M))
    Z = rng.dirichlet(np.ones(K), size=n)
    Z = np.clip(Z, 1e-10, None); Z /= Z.sum(axis=1, keepdims=True)
    Y = Z @ nu + np.einsum("ik,im,kpm->ip", Z, chi, Phi) + np.sqrt(0.001) * rng.standard_normal((n, P))
    T = 8
    cfg = bf.default_config(model=bf.MODEL_MULTIVARIATE, K=K, n_eigen=M, tot_mcmc_iters=T)
    smp = bf.Sampler(cfg, Y)
    state = dict(nu=nu, Phi=Phi, chi=chi, Z=Z, pi=np.full(K, 1.0 / K), alpha_3=[10.0], delta=np.ones((K, M)),
                 A=np.ones((K, 2)), gamma=np.ones((K, P, M)), tau=np.ones(K), sigma_sq=[0.001])
    smp.set_state(**state)
    smp.run(S.SWEEP_WARM, T, seed=2)
    Zc = smp.get_chain("Z"); ll = smp.get_chain("loglik"); s2 = smp.get_chain("sigma_sq")
    assert np.abs(Zc.sum(axis=1) - 1).max() < 1e-12
    t = T - 1
    coef = Zc[:, :, t] @ smp.get_chain("nu")[:, :, t] + np.einsum("ik,im,kpm->ip", Zc[:, :, t], smp.get_chain("chi")[:, :, t],
                                                               smp.get_chain("Phi")[..., t])
    rss = ((Y - coef) ** 2).sum()
    ref = -n * (P // 2) * np.log(2 * np.pi * s2[t]) - rss / (2 * s2[t])
    assert abs(ll[t] - ref) < 1e-8 * abs(ref)
    assert 0.0005 < s2[t] < 0.002


def test_config3_covariate_adjusted_properties():
    # BASELINE.json configs[2]: config 2 + D = 5 covariates, mean and covariance adjustment (19-update sweep)
    import time
    import bayesfmmm_amd as bf
    from bench import make_config2
    S = bf.sampler
    w = make_config2()
    rng = np.random.default_rng(8)
    n, K, P, M, D = w["n"], w["K"], w["P"], w["M"], 5
    X = rng.standard_normal((n, D))
    eta = 0.3 * rng.standard_normal((P, D, K))
    xi = np.stack([0.1 * (M - m) / M * rng.standard_normal((P, D, K)) for m in range(M)], axis=2)   # P x D x M x K
    B = w["B"][0]
    st = w["state"]
    coef = np.zeros((n, P))
    for k in range(K):
        u = st["nu"][k][None, :] + X @ eta[:, :, k].T
        for m in range(M):
            u = u + st["chi"][:, m:m + 1] * (st["Phi"][k, :, m][None, :] + X @ xi[:, :, m, k].T)
        coef += st["Z"][:, k:k + 1] * u
    Y = coef @ B.T + 0.1 * rng.standard_normal((n, B.shape[0]))
    T = 6
    cfg = bf.default_config(model=bf.MODEL_FUNCTIONAL, K=K, n_eigen=M, basis_degree=3, tot_mcmc_iters=T)
    smp = bf.Sampler(cfg, [Y[i] for i in range(n)], w["t"], w["internal_knots"], w["boundary_knots"])
    smp.set_covariates(X, True)
    smp.set_state(**st)
    smp.set_state(eta=eta, xi=xi)
    mask = S.SWEEP_WARM | S.COV_MEAN | S.COV_XI
    smp.run(mask, 2, seed=2)
    t0 = time.perf_counter()
    smp.run(mask, T - 2, first_iter=2, seed=2)
    dt = (time.perf_counter() - t0) / (T - 2)
    print(f"config 3: {dt * 1e3:.2f} ms per 19-update sweep")
    t = T - 1
    Z = smp.get_chain("Z")[:, :, t]; nu = smp.get_chain("nu")[:, :, t]; Phi = smp.get_chain("Phi")[..., t]
    chi = smp.get_chain("chi")[:, :, t]; e_t = smp.get_chain("eta")[..., t]; x_t = smp.get_chain("xi")[..., t]
    s2 = smp.get_chain("sigma_sq")[t]
    coef = np.zeros((n, P))
    for k in range(K):
        u = nu[k][None, :] + X @ e_t[:, :, k].T
        for m in range(M):
            u = u + chi[:, m:m + 1] * (Phi[k, :, m][None, :] + X @ x_t[:, :, m, k].T)
        coef += Z[:, k:k + 1] * u
    rss = ((Y - coef @ B.T) ** 2).sum()
    ref = -Y.size * (0.9189385332046727 + 0.5 * np.log(s2)) - rss / (2 * s2)
    ll = smp.get_chain("loglik")[t]
    assert abs(ll - ref) < 1e-8 * abs(ref)
    assert 0.005 < s2 < 0.02


def test_beyond_the_cache_resident_size_properties():
    """n_funct = 65536 (16 x BASELINE configs[1]; 80 MB of records): the geometry no benchmark reaches -- 342 k-slices of k_pair_gram
    and their fixed-order reduction, 8192 curve workgroups per launch, 64-bit offsets of the chain slots -- checked through the same
    size-independent properties as test_config2_properties: simplex rows, the log-likelihood recomputed from the state on the host,
    sigma^2 staying at the generating value, bit-reproducibility (BFMMM.h:1502-1553, CalculateLikelihood.h:19-44)."""
    import bayesfmmm_amd as bf
    from bench import make_config2
    S = bf.sampler
    w = make_config2(n=65536, n_i=24)
    T = 5
    cfg = bf.default_config(model=bf.MODEL_FUNCTIONAL, K=3, n_eigen=6, basis_degree=3, tot_mcmc_iters=T)
    smp = bf.Sampler(cfg, w["y"], w["t"], w["internal_knots"], w["boundary_knots"])
    smp.set_state(**w["state"])
    smp.run(S.SWEEP_WARM, T, seed=7)
    Z = smp.get_chain("Z"); nu = smp.get_chain("nu"); Phi = smp.get_chain("Phi"); chi = smp.get_chain("chi")
    s2 = smp.get_chain("sigma_sq"); ll = smp.get_chain("loglik")
    assert np.abs(Z.sum(axis=1) - 1).max() < 1e-12 and (Z > 0).all()
    t = T - 1
    ref, _ = host_loglik_functional(w, nu[:, :, t], Phi[..., t], chi[:, :, t], Z[:, :, t], s2[t])
    assert abs(ll[t] - ref) < 1e-8 * abs(ref), (ll[t], ref)
    assert 0.008 < s2[-1] < 0.0125
    smp.set_state(**w["state"])
    smp.run(S.SWEEP_WARM, T, seed=7)
    np.testing.assert_array_equal(smp.get_chain("nu"), nu)
    np.testing.assert_array_equal(smp.get_chain("sigma_sq"), s2)
    smp.close()
