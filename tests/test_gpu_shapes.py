"""Shape sweep: the HIP path against the oracle for problem shapes other than the benchmark's (number of clusters,
eigenfunctions, spline degree, knots, ragged curves) -- exercises every band-width instantiation, both sweep kernels
(register-resident and general) and the odd-size tails of the tiled kernels."""
import numpy as np
import pytest

import oracle_lib as O
from gpu_parity import ORC_FIELD, push_state, rel_err

pytestmark = pytest.mark.gpu


def simulate(n, K, M, degree, n_internal, seed, n_pts=60):
    rng = np.random.default_rng(seed)
    t_full = np.linspace(0.0, 100.0, n_pts)
    ik = np.linspace(0.0, 100.0, n_internal + 2)[1:-1]
    bk = np.array([0.0, 100.0])
    B_full = O.bspline_basis(t_full, ik, degree, bk)
    P = B_full.shape[1]
    nu = np.cumsum(rng.standard_normal((K, P)), axis=1)
    Phi = np.stack([(M - m) / M * 0.4 * rng.standard_normal((K, P)) for m in range(M)], axis=2)
    chi = rng.standard_normal((n, M))
    Z = rng.dirichlet(np.full(K, 1.5), size=n)
    ts, Bs, ys = [], [], []
    for i in range(n):
        keep = np.sort(rng.choice(n_pts, size=rng.integers(n_pts // 2, n_pts + 1), replace=False))
        B = B_full[keep]
        c = Z[i] @ (nu + np.einsum("m,kpm->kp", chi[i], Phi))
        ts.append(t_full[keep]); Bs.append(B); ys.append(B @ c + 0.1 * rng.standard_normal(len(keep)))
    return dict(t=ts, B=Bs, y=ys, nu=nu, Phi=Phi, chi=chi, Z=Z, K=K, M=M, P=P, n=n, internal_knots=ik, boundary_knots=bk,
                degree=degree)


@pytest.mark.parametrize("K,M,degree,n_internal,n", [
    (2, 1, 1, 3, 33),      # band width 1, a single eigenfunction
    (4, 5, 2, 6, 70),      # band width 2, A*P = 216
    (6, 2, 3, 8, 90),      # K = 6, P = 12
    (7, 1, 3, 4, 80),      # K = 7: 2 K + 1 = 15 lanes per curve in the Z-proposal job
    (8, 2, 2, 5, 96),      # the largest K (2 K + 1 = 17: 32 lanes per curve), A * P = 192
    (3, 7, 4, 10, 50),     # band width 4, odd M, P = 15
    (2, 3, 5, 20, 40),     # band width 5, P = 26
    (5, 9, 3, 26, 64),     # A*P = 1500 > 704: the general sweep kernel, P = 30
    (8, 8, 3, 4, 64),      # P = 8, A = 72: the one-wave sweep kernel with more than 64 KB of dynamic LDS (opt-in attribute)
])
def test_warm_trajectory_matches_oracle_across_shapes(K, M, degree, n_internal, n):
    import bayesfmmm_amd as bf
    T = 4
    sim = simulate(n, K, M, degree, n_internal, seed=100 + K * 10 + M)
    model = O.Model(sim["y"], sim["B"], K, M)
    ch = O.Chain(model, T)
    rng = np.random.default_rng(7)
    P = sim["P"]
    ch.nu[:, :, 0] = sim["nu"] + 0.2 * rng.standard_normal((K, P))
    ch.Phi[..., 0] = sim["Phi"] + 0.1 * rng.standard_normal((K, P, M))
    ch.chi[:, :, 0] = sim["chi"] + 0.2 * rng.standard_normal((n, M))
    ch.Z[:, :, 0] = rng.dirichlet(np.full(K, 2.0), size=n)
    ch.pi[:, 0] = rng.dirichlet(np.full(K, 5.0))
    ch.alpha3[0] = 3.5
    ch.delta[:, :, 0] = rng.gamma(2.0, 1.0, size=(K, M))
    ch.A[:, :, 0] = rng.gamma(2.0, 1.0, size=(K, 2))
    ch.gamma[..., 0] = rng.gamma(2.0, 0.7, size=(K, P, M))
    ch.tau[0, :] = rng.gamma(3.0, 0.5, size=K)
    ch.sigma[0] = 0.02
    cfg = bf.default_config(model=bf.MODEL_FUNCTIONAL, K=K, n_eigen=M, basis_degree=degree, tot_mcmc_iters=T)
    smp = bf.Sampler(cfg, sim["y"], sim["t"], sim["internal_knots"], sim["boundary_knots"])
    push_state(smp, ch)
    h = O.make_hyper(K)
    O.run_sweeps(model, h, ch, O.SWEEP_WARM, seed=3)
    smp.run(bf.sampler.SWEEP_WARM, T, seed=3)
    for nm in ["nu", "Phi", "chi", "Z", "pi", "alpha_3", "delta", "A", "gamma", "tau", "sigma_sq", "loglik"]:
        err = rel_err(smp.get_chain(nm), getattr(ch, ORC_FIELD.get(nm, nm)))
        assert err < 1e-6, (nm, err)


@pytest.mark.parametrize("K,M,degree,n_internal,n,D,cov_adj", [
    (2, 2, 3, 31, 70, 3, True),     # P = 35 (odd, > 32): 64-lane groups, C_a not 16-byte aligned for odd directions
    (3, 3, 2, 37, 45, 2, True),     # P = 40, band width 2
    (2, 1, 1, 6, 130, 8, True),     # the largest D, band width 1, more than one k_cov_group workgroup
    (4, 2, 5, 10, 37, 1, False),    # band width 5, mean adjustment only
    (3, 5, 4, 20, 33, 4, True),     # band width 4, P = 25, odd M
    (7, 2, 3, 6, 60, 2, True),      # K = 7 with covariates
])
def test_covariate_adjusted_trajectory_across_shapes(K, M, degree, n_internal, n, D, cov_adj):
    """The eta / Xi block (k_cov_prep, k_cov_w2, k_cov_factor, k_cov_group, k_cov_hyper) and the covariate variants of the
    per-curve kernels for shapes other than the benchmark's."""
    import bayesfmmm_amd as bf
    from gpu_parity import oracle_slot
    S = bf.sampler
    T = 3
    sim = simulate(n, K, M, degree, n_internal, seed=200 + K * 10 + M + D)
    rng = np.random.default_rng(17 + D)
    P = sim["P"]
    X = rng.standard_normal((n, D))
    eta = 0.5 * rng.standard_normal((P, D, K))
    xi = 0.2 * rng.standard_normal((P, D, M, K)) * (1.0 if cov_adj else 0.0)
    for i in range(n):          # add the covariate part of the mean to the simulated curves
        c = np.zeros(P)
        for k in range(K):
            u = eta[:, :, k] @ X[i]
            for m in range(M):
                u = u + sim["chi"][i, m] * (xi[:, :, m, k] @ X[i])
            c += sim["Z"][i, k] * u
        sim["y"][i] = sim["y"][i] + sim["B"][i] @ c
    model = O.Model(sim["y"], sim["B"], K, M, X=X)
    ch = O.Chain(model, T)
    ch.nu[:, :, 0] = sim["nu"] + 0.2 * rng.standard_normal((K, P))
    ch.Phi[..., 0] = sim["Phi"] + 0.1 * rng.standard_normal((K, P, M))
    ch.chi[:, :, 0] = sim["chi"] + 0.2 * rng.standard_normal((n, M))
    ch.Z[:, :, 0] = rng.dirichlet(np.full(K, 2.0), size=n)
    ch.pi[:, 0] = rng.dirichlet(np.full(K, 5.0))
    ch.alpha3[0] = 3.5
    ch.delta[:, :, 0] = rng.gamma(2.0, 1.0, size=(K, M))
    ch.A[:, :, 0] = rng.gamma(2.0, 1.0, size=(K, 2))
    ch.gamma[..., 0] = rng.gamma(2.0, 0.7, size=(K, P, M))
    ch.tau[0, :] = rng.gamma(3.0, 0.5, size=K)
    ch.sigma[0] = 0.02
    ch.eta[..., 0] = eta + 0.1 * rng.standard_normal((P, D, K))
    ch.xi[..., 0] = (xi + 0.05 * rng.standard_normal((P, D, M, K))) if cov_adj else 0.0
    ch.tau_eta[..., 0] = rng.gamma(3.0, 0.5, size=(K, D))
    ch.gamma_xi[..., 0] = rng.gamma(2.0, 0.7, size=(P, D, M, K))
    ch.delta_xi[..., 0] = rng.gamma(2.0, 1.0, size=(K, M, D))
    ch.A_xi[..., 0] = rng.gamma(2.0, 1.0, size=(K, 2, D))
    cfg = bf.default_config(model=bf.MODEL_FUNCTIONAL, K=K, n_eigen=M, basis_degree=degree, tot_mcmc_iters=T)
    smp = bf.Sampler(cfg, sim["y"], sim["t"], sim["internal_knots"], sim["boundary_knots"])
    smp.set_covariates(X, cov_adj)
    push_state(smp, ch)
    smp.set_state(**{nm: oracle_slot(ch, nm, 0) for nm in ["eta", "xi", "tau_eta", "gamma_xi", "delta_xi", "A_xi"]})
    h = O.make_hyper(K)
    O.run_sweeps(model, h, ch, O.SWEEP_WARM, seed=3, covariance_adj=cov_adj)
    smp.run(S.SWEEP_WARM | S.COV_MEAN | (S.COV_XI if cov_adj else 0), T, seed=3)
    names = ["nu", "Phi", "chi", "Z", "pi", "alpha_3", "delta", "A", "gamma", "tau", "sigma_sq", "eta", "tau_eta", "loglik"]
    names += ["xi", "delta_xi", "A_xi", "gamma_xi"] if cov_adj else []
    for nm in names:
        err = rel_err(smp.get_chain(nm), getattr(ch, ORC_FIELD.get(nm, nm)))
        assert err < 1e-6, (nm, err)


def test_build_limits_fail_loudly():
    """The limits of this build that the reference does not have (K <= 8, P <= 64, n_eigen <= 16, degree <= 5, D <= 8) are
    refused at set-up with a message that names the limit -- never truncated, never run on another path."""
    import bayesfmmm_amd as bf
    from bayesfmmm_amd import _lib
    sim = simulate(24, 2, 2, 3, 4, seed=3)

    def make(**kw):
        args = dict(model=bf.MODEL_FUNCTIONAL, K=2, n_eigen=2, basis_degree=3, tot_mcmc_iters=4)
        args.update(kw)
        ik = kw.pop("_ik", sim["internal_knots"])
        cfg = bf.default_config(**{k: v for k, v in args.items() if not k.startswith("_")})
        return bf.Sampler(cfg, sim["y"], sim["t"], ik, sim["boundary_knots"])

    with pytest.raises(_lib.BfmmmError, match="K larger than 8"):
        make(K=9)
    with pytest.raises(_lib.BfmmmError, match="n_eigen larger than 16"):
        make(n_eigen=17)
    with pytest.raises(_lib.BfmmmError, match="basis_degree larger than 5"):
        make(basis_degree=6)
    with pytest.raises(_lib.BfmmmError, match="P larger than 64"):
        make(_ik=np.linspace(0.0, 100.0, 64)[1:-1])        # 62 internal knots + degree 3 + 1 = 66 basis functions
    rng = np.random.default_rng(1)
    with pytest.raises(_lib.BfmmmError, match="P larger than 64"):
        bf.Sampler(bf.default_config(model=bf.MODEL_MULTIVARIATE, K=2, n_eigen=2, tot_mcmc_iters=4), rng.standard_normal((30, 65)))
    s = make()
    with pytest.raises(_lib.BfmmmError, match="between 1 and 8"):
        s.set_covariates(rng.standard_normal((sim["n"], 9)), covariance_adj=False)
    s.close()
