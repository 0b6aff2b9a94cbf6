"""The high-dimensional functional model's statistics on the device: the functional sampler over a
caller-supplied basis (bfmmm_create_from_basis) with the tensor-product B-spline basis and penalty of
inst/include/BayesFMMM/BSplines.h:18-120 -- band half-widths beyond the spline-degree instantiations go through the
wide-band (BW = 31) kernels, the dense factorisation fallback and the general sweep kernel -- against the oracle, whose
reference-structure loops take the basis rows and the penalty as data (drivers BFMMM.h:2892, :3041, :3210 run the same
updates as the univariate ones)."""
import numpy as np
import pytest

import oracle_lib as O
from gpu_parity import ORC_FIELD, push_state, rel_err

pytestmark = pytest.mark.gpu


def simulate_tensor(n, K, M, degs, n_int, seed, n_pts=45):
    from bayesfmmm_amd import api
    rng = np.random.default_rng(seed)
    dim = len(degs)
    iks = [np.linspace(0.0, 1.0, k + 2)[1:-1] for k in n_int]
    bks = [[0.0, 1.0]] * dim
    Pl = [k + g + 1 for k, g in zip(n_int, degs)]
    P = int(np.prod(Pl))
    strides = [int(np.prod(Pl[l + 1:])) for l in range(dim)]
    band = sum(g * s for g, s in zip(degs, strides))
    nu = rng.standard_normal((K, P))
    Phi = np.stack([(M - m) / M * 0.4 * rng.standard_normal((K, P)) for m in range(M)], axis=2)
    chi = rng.standard_normal((n, M))
    Z = rng.dirichlet(np.full(K, 1.5), size=n)
    ts, Bs, ys = [], [], []
    for i in range(n):
        t = rng.uniform(0.0, 1.0, size=(int(rng.integers(n_pts // 2, n_pts + 1)), dim))
        B = api.TensorBSpline(t, degs, bks, iks)
        c = Z[i] @ (nu + np.einsum("m,kpm->kp", chi[i], Phi))
        ts.append(t); Bs.append(np.ascontiguousarray(B)); ys.append(B @ c + 0.1 * rng.standard_normal(len(t)))
    Pm = api.GetP(degs, n_int)
    return dict(t=ts, B=Bs, y=ys, nu=nu, Phi=Phi, chi=chi, Z=Z, K=K, M=M, P=P, n=n, band=band, Pmat=Pm, pen_band=max(strides))


@pytest.mark.parametrize("K,M,degs,n_int,n", [
    (2, 2, [2, 2], [2, 2], 31),        # 5 x 5 = 25 basis functions, band 12: 32-lane groups, wide band
    (3, 2, [3, 3], [3, 3], 40),        # 7 x 7 = 49 (the reference's golden basis), band 24: 64-lane groups
    (2, 1, [1, 2, 1], [1, 1, 1], 37),  # three dimensions: 3 x 4 x 3 = 36, band 19
])
def test_tensor_model_warm_trajectory_matches_oracle(K, M, degs, n_int, n):
    import bayesfmmm_amd as bf
    T = 3
    sim = simulate_tensor(n, K, M, degs, n_int, seed=300 + K + len(degs))
    P = sim["P"]
    model = O.Model(sim["y"], sim["B"], K, M, Pmat=sim["Pmat"])
    ch = O.Chain(model, T)
    rng = np.random.default_rng(7)
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
    cfg = bf.default_config(model=bf.MODEL_FUNCTIONAL, K=K, n_eigen=M, basis_degree=max(degs), tot_mcmc_iters=T)
    smp = bf.Sampler(cfg, sim["y"], basis=sim["B"], band=sim["band"], penalty=sim["Pmat"], penalty_band=sim["pen_band"])
    push_state(smp, ch)
    h = O.make_hyper(K)
    O.run_sweeps(model, h, ch, O.SWEEP_WARM, seed=3)
    smp.run(bf.sampler.SWEEP_WARM, T, seed=3)
    for nm in ["nu", "Phi", "chi", "Z", "pi", "alpha_3", "delta", "A", "gamma", "tau", "sigma_sq", "loglik"]:
        err = rel_err(smp.get_chain(nm), getattr(ch, ORC_FIELD.get(nm, nm)))
        assert err < 1e-6, (nm, err)


@pytest.mark.parametrize("K,M,degs,n_int,n,D,cov_adj", [
    (2, 2, [2, 2], [2, 2], 31, 2, True),         # 25 basis functions, band 12, two covariates, mean + covariance adjusted
    (2, 2, [2, 2], [3, 3], 37, 1, False),        # 36 (the package's HD example basis), band 14: 64-lane groups, mean adjusted
    (2, 1, [3, 3], [3, 3], 33, 2, True),         # 49, band 24
])
def test_tensor_model_with_covariates_matches_oracle(K, M, degs, n_int, n, D, cov_adj):
    """Covariate-adjusted high-dimensional drivers (BFMMM.h:6750, :6913, :7137 mean-adjusted; :7655 mean + covariance
    adjusted): the eta / Xi block over wide-band statistics (k_cov_group<31, .>, dense factorisation of the penalised
    pair blocks)."""
    import bayesfmmm_amd as bf
    from gpu_parity import STATE_NAMES, oracle_slot
    S = bf.sampler
    T = 3
    sim = simulate_tensor(n, K, M, degs, n_int, seed=500 + D + len(n_int) + n_int[0])
    P = sim["P"]
    rng = np.random.default_rng(17)
    X = rng.standard_normal((n, D))
    model = O.Model(sim["y"], sim["B"], K, M, X=X, Pmat=sim["Pmat"])
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
    ch.eta[..., 0] = 0.2 * rng.standard_normal((P, D, K))
    ch.xi[..., 0] = 0.05 * rng.standard_normal((P, D, M, K)) if cov_adj else 0.0
    ch.tau_eta[..., 0] = rng.gamma(3.0, 0.5, size=(K, D))
    ch.gamma_xi[..., 0] = rng.gamma(2.0, 0.7, size=(P, D, M, K))
    ch.delta_xi[..., 0] = rng.gamma(2.0, 1.0, size=(K, M, D))
    ch.A_xi[..., 0] = rng.gamma(2.0, 1.0, size=(K, 2, D))
    cfg = bf.default_config(model=bf.MODEL_FUNCTIONAL, K=K, n_eigen=M, basis_degree=max(degs), tot_mcmc_iters=T)
    smp = bf.Sampler(cfg, sim["y"], basis=sim["B"], band=sim["band"], penalty=sim["Pmat"], penalty_band=sim["pen_band"])
    smp.set_covariates(X, cov_adj)
    push_state(smp, ch)
    smp.set_state(**{nm: oracle_slot(ch, nm, 0) for nm in ["eta", "xi", "tau_eta", "gamma_xi", "delta_xi", "A_xi"]})
    h = O.make_hyper(K)
    O.run_sweeps(model, h, ch, O.SWEEP_WARM, seed=4, covariance_adj=cov_adj)
    smp.run(S.SWEEP_WARM | S.COV_MEAN | (S.COV_XI if cov_adj else 0), T, seed=4)
    names = STATE_NAMES + ["eta", "tau_eta", "loglik"] + (["xi", "delta_xi", "A_xi", "gamma_xi"] if cov_adj else [])
    for nm in names:
        err = rel_err(smp.get_chain(nm), getattr(ch, ORC_FIELD.get(nm, nm)))
        assert err < 1e-6, (nm, err)
