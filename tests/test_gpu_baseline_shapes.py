"""Oracle parity AT THE KERNEL INSTANTIATIONS THE BENCHMARKS RUN (BASELINE.json configs 1-5): the same
(K, M, P, degree, D) as the benchmark shapes -- only n_funct is small enough for the CPU oracle to finish in seconds -- so that
the template instantiations the headline numbers come from (k_sweep_fast at A*P = 630 with 14 waves, the 168-pair-row
k_pair_gram, the D = 5 group kernels, k_sweep_diag<5, true> with 450 pair rows, the grouped Nu_Z batch contraction) are checked
against the restatement of the reference's loops, not only through size-independent properties (tests/test_gpu_fullsize.py).

Reference loops: BFMMM.h:1502-1553 (warm-start sweep), :4809-4894 (Mean_CovAdj sweep), :2597-2650 (multivariate warm start),
:1073-1113 (Nu_Z sweep), :1250-1300 (Theta sweep).  Tolerances as everywhere: 1e-8 single updates, 1e-6 four-sweep trajectories."""
import numpy as np
import pytest

import oracle_lib as O
from gpu_parity import ORC_FIELD, STATE_NAMES, oracle_slot, push_state, rel_err
from test_gpu_shapes import simulate

pytestmark = pytest.mark.gpu

WARM_NAMES = ["nu", "Phi", "chi", "Z", "pi", "alpha_3", "delta", "A", "gamma", "tau", "sigma_sq", "loglik"]
COV_NAMES = ["eta", "xi", "tau_eta", "gamma_xi", "delta_xi", "A_xi"]


def generic_state(ch, sim, rng, mv=False):
    n, K, M, P = sim["n"], sim["K"], sim["M"], sim["P"]
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


def config2_setup(T, n=64, seed=402):
    """BASELINE configs[1] except n_funct: K = 3, M = 6, cubic splines with 26 internal knots (P = 30)"""
    import bayesfmmm_amd as bf
    K, M = 3, 6
    sim = simulate(n, K, M, 3, 26, seed=seed, n_pts=100)
    assert sim["P"] == 30
    model = O.Model(sim["y"], sim["B"], K, M)
    ch = O.Chain(model, T)
    generic_state(ch, sim, np.random.default_rng(seed + 1))
    cfg = bf.default_config(model=bf.MODEL_FUNCTIONAL, K=K, n_eigen=M, basis_degree=3, tot_mcmc_iters=T)
    smp = bf.Sampler(cfg, sim["y"], sim["t"], sim["internal_knots"], sim["boundary_knots"])
    push_state(smp, ch)
    return sim, model, ch, smp


def test_config2_shape_warm_trajectory():
    """k_sweep_fast at A*P = 630 (14 waves), k_pair_gram with 168 pair rows, k_factor at P = 30, k_curve_chi at K = 3, M = 6"""
    import bayesfmmm_amd as bf
    T = 4
    sim, model, ch, smp = config2_setup(T)
    d = smp.dims()
    assert d["P"] == 30 and d["A"] == 21 and d["A"] * d["P"] == 630
    O.run_sweeps(model, O.make_hyper(3), ch, O.SWEEP_WARM, seed=3)
    smp.run(bf.sampler.SWEEP_WARM, T, seed=3)
    for nm in WARM_NAMES:
        err = rel_err(smp.get_chain(nm), getattr(ch, ORC_FIELD.get(nm, nm)))
        assert err < 1e-6, (nm, err)
    smp.close()


@pytest.mark.parametrize("which", ["Nu", "Phi", "Chi", "Z", "Sigma"])
def test_config2_shape_single_updates(which):
    """the data-dependent blocks one at a time at K = 3, M = 6, P = 30 (UpdateNu.h:39-70, UpdatePhi.h:40-84, UpdateChi.h:19-64,
    UpdateMixedMembership.h:131-185, UpdateSigma.h:22-58)"""
    import bayesfmmm_amd as bf
    S = bf.sampler
    sim, model, ch, smp = config2_setup(2, n=48, seed=411)
    h = O.make_hyper(3)
    it, seed = 0, 77
    tilde_tau = np.cumprod(ch.delta[:, :, 0], axis=1)
    calls = {
        "Nu": (S.U_NU, lambda: O.updateNu(model, ch, it, seed=seed), "nu"),
        "Phi": (S.U_PHI, lambda: O.updatePhi(model, ch, it, tilde_tau, seed=seed), "Phi"),
        "Chi": (S.U_CHI, lambda: O.updateChi(model, ch, it, seed=seed), "chi"),
        "Z": (S.U_Z, lambda: O.updateZ_PM(model, ch, it, h.a_Z_PM, seed=seed), "Z"),
        "Sigma": (S.U_SIGMA, lambda: O.updateSigma(model, ch, it, h.alpha_0, h.beta_0, seed=seed), "sigma_sq"),
    }
    mask, orc_call, nm = calls[which]
    orc_call()
    smp.run(mask, 1, first_iter=0, seed=seed)
    err = rel_err(smp.get_state(nm).reshape(-1), oracle_slot(ch, nm, 0).reshape(-1))
    assert err < 1e-8, (which, err)
    smp.close()


def test_config3_shape_covariate_adjusted_trajectory():
    """BASELINE configs[2] except n_funct: config 2 + D = 5 covariates, covariance_adj: the 19-update Mean_CovAdj sweep
    (BFMMM.h:4809-4894); k_cov_group<.., D = 5> with 21 groups, k_cov_w2 with 15 in-group pairs, k_cov_factor on 105 directions"""
    import bayesfmmm_amd as bf
    S = bf.sampler
    T, n, K, M, D = 3, 64, 3, 6, 5
    sim = simulate(n, K, M, 3, 26, seed=403, n_pts=100)
    P = sim["P"]
    rng = np.random.default_rng(31)
    X = rng.standard_normal((n, D))
    eta = 0.5 * rng.standard_normal((P, D, K))
    xi = np.stack([0.1 * (M - m) / M * rng.standard_normal((P, D, K)) for m in range(M)], axis=2)      # P x D x M x K
    for i in range(n):
        c = np.zeros(P)
        for k in range(K):
            u = eta[:, :, k] @ X[i]
            for m in range(M):
                u = u + sim["chi"][i, m] * (xi[:, :, m, k] @ X[i])
            c += sim["Z"][i, k] * u
        sim["y"][i] = sim["y"][i] + sim["B"][i] @ c
    model = O.Model(sim["y"], sim["B"], K, M, X=X)
    ch = O.Chain(model, T)
    generic_state(ch, sim, rng)
    ch.eta[..., 0] = eta + 0.1 * rng.standard_normal((P, D, K))
    ch.xi[..., 0] = xi + 0.05 * rng.standard_normal((P, D, M, K))
    ch.tau_eta[..., 0] = rng.gamma(3.0, 0.5, size=(K, D))
    ch.gamma_xi[..., 0] = rng.gamma(2.0, 0.7, size=(P, D, M, K))
    ch.delta_xi[..., 0] = rng.gamma(2.0, 1.0, size=(K, M, D))
    ch.A_xi[..., 0] = rng.gamma(2.0, 1.0, size=(K, 2, D))
    cfg = bf.default_config(model=bf.MODEL_FUNCTIONAL, K=K, n_eigen=M, basis_degree=3, tot_mcmc_iters=T)
    smp = bf.Sampler(cfg, sim["y"], sim["t"], sim["internal_knots"], sim["boundary_knots"])
    smp.set_covariates(X, True)
    push_state(smp, ch)
    smp.set_state(**{nm: oracle_slot(ch, nm, 0) for nm in COV_NAMES})
    O.run_sweeps(model, O.make_hyper(K), ch, O.SWEEP_WARM, seed=3, covariance_adj=True)
    smp.run(S.SWEEP_WARM | S.COV_MEAN | S.COV_XI, T, seed=3)
    for nm in WARM_NAMES + COV_NAMES:
        err = rel_err(smp.get_chain(nm), getattr(ch, ORC_FIELD.get(nm, nm)))
        assert err < 1e-6, (nm, err)
    smp.close()


def test_config4_shape_multivariate_warm_trajectory():
    """BASELINE configs[3] except N: dim = 50, K = 4, M = 8 (BFMMM.h:2597-2650): k_sweep_diag<5, true> (A = 36), 450 pair rows in
    k_pair_gram's one-column-tile contraction, the K / M = 4 / 8 instantiation of the per-row kernels with 64-lane groups"""
    import bayesfmmm_amd as bf
    from test_gpu_multivariate import simulate_mv
    T, n, P, K, M = 4, 96, 50, 4, 8
    sim = simulate_mv(n, P, K, M, 0.05, seed=404)
    model = O.Model([sim["Y"][i] for i in range(n)], [np.eye(P)] * n, K, M, mv=True)
    ch = O.Chain(model, T)
    generic_state(ch, sim, np.random.default_rng(405))
    ch.sigma[0] = 0.07
    cfg = bf.default_config(model=bf.MODEL_MULTIVARIATE, K=K, n_eigen=M, tot_mcmc_iters=T)
    smp = bf.Sampler(cfg, sim["Y"])
    push_state(smp, ch)
    assert smp.dims()["A"] == 36
    O.run_sweeps(model, O.make_hyper(K), ch, O.SWEEP_WARM, seed=5)
    smp.run(bf.sampler.SWEEP_WARM, T, seed=5)
    for nm in WARM_NAMES:
        err = rel_err(smp.get_chain(nm), getattr(ch, ORC_FIELD.get(nm, nm)))
        assert err < 1e-6, (nm, err)
    smp.close()


def test_config5_shape_nu_z_batch_of_eight_chains():
    """BASELINE configs[4] except n_funct: the 8 chains of BFMMM_Nu_Z_multiple_try (n_try = 7) as ONE chain batch at K = 3, P = 30
    -- the grouped k_pair_gram instantiation (one row tile per chain, chains staged and contracted in groups), the lean trailing
    k_curve_z, two half-batches on two streams -- every chain against the oracle run with that chain's RNG id (BFMMM.h:1073-1113)"""
    import bayesfmmm_amd as bf
    S = bf.sampler
    T, n, K, M, NCH = 4, 64, 3, 6, 8
    sim = simulate(n, K, M, 3, 26, seed=406, n_pts=100)
    model = O.Model(sim["y"], sim["B"], K, M)
    cfg = bf.default_config(model=bf.MODEL_FUNCTIONAL, K=K, n_eigen=M, basis_degree=3, tot_mcmc_iters=T)
    batch = bf.Sampler(cfg, sim["y"], sim["t"], sim["internal_knots"], sim["boundary_knots"], n_chains=NCH)
    chains = []
    for q in range(NCH):
        ch = O.Chain(model, T)
        generic_state(ch, sim, np.random.default_rng(500 + q))
        ch.chi[:] = 0.0
        ch.Phi[:] = 0.0
        chains.append(ch)
        batch.select_chain(q)
        push_state(batch, ch)
    batch.run(S.SWEEP_NU_Z, T, seed=9, chain=0, phi_chi_zero=True)
    h = O.make_hyper(K)
    for q in range(NCH):
        O.run_sweeps(model, h, chains[q], O.SWEEP_NU_Z, n_iter=T, seed=9, chain_id=q)
        batch.select_chain(q)
        for nm in ["nu", "Z", "pi", "alpha_3", "tau", "sigma_sq", "loglik"]:
            err = rel_err(batch.get_chain(nm), getattr(chains[q], ORC_FIELD.get(nm, nm)))
            assert err < 1e-6, (q, nm, err)
    batch.close()


def test_config1_shape_theta_sweep():
    """BASELINE configs[0] at its stated shape: n = 40 curves of n_i = 50 points t = 0, 20, ..., 980, cubic splines
    with internal knots {200, 400, 600, 800} on (0, 1000) (P = 8), K = 2, M = 3; truth as in src/test-Nu.cpp:28-38; the Theta sweep
    (BFMMM_Theta order, BFMMM.h:1250-1300) with Z and nu held at the truth, seed 1: the first 20 of its 200 iterations against the
    oracle, the rest through the reference's own recovery criterion on sigma^2."""
    import bayesfmmm_amd as bf
    from simdata import NU_TABLE
    S = bf.sampler
    n, n_i, K, M, T, T_cmp = 40, 50, 2, 3, 200, 20
    rng = np.random.default_rng(1)
    t = np.arange(0.0, 1000.0, 20.0)
    ik = np.array([200.0, 400.0, 600.0, 800.0])
    bk = np.array([0.0, 1000.0])
    B = O.bspline_basis(t, ik, 3, bk)
    P = B.shape[1]
    assert P == 8 and len(t) == n_i
    nu = NU_TABLE[:K].copy()
    Phi = np.stack([(M - m) * 0.1 * rng.uniform(size=(K, P)) for m in range(M)], axis=2)
    chi = rng.standard_normal((n, M))
    Z = rng.dirichlet(np.full(K, 10.0), size=n)
    ys = []
    for i in range(n):
        c = Z[i] @ (nu + np.einsum("m,kpm->kp", chi[i], Phi))
        ys.append(B @ c + 0.1 * rng.standard_normal(n_i))
    sim = dict(n=n, K=K, M=M, P=P, nu=nu, Phi=Phi, chi=chi, Z=Z)
    model = O.Model(ys, [B] * n, K, M)
    ch = O.Chain(model, T_cmp)
    generic_state(ch, sim, np.random.default_rng(2))
    ch.nu[:] = nu[:, :, None]                 # Z and nu fixed at the truth in every slot (the Theta sweep copies them forward)
    ch.Z[:] = Z[:, :, None]
    ch.sigma[0] = 0.01
    cfg = bf.default_config(model=bf.MODEL_FUNCTIONAL, K=K, n_eigen=M, basis_degree=3, tot_mcmc_iters=T)
    smp = bf.Sampler(cfg, ys, [t] * n, ik, bk)
    push_state(smp, ch)
    O.run_sweeps(model, O.make_hyper(K), ch, O.SWEEP_THETA, n_iter=T_cmp, seed=1)
    smp.run(S.SWEEP_THETA, T, seed=1)
    for nm in ["Phi", "chi", "delta", "A", "gamma", "tau", "sigma_sq", "loglik", "nu", "Z"]:
        err = rel_err(smp.get_chain(nm, T_cmp), getattr(ch, ORC_FIELD.get(nm, nm)))
        assert err < 1e-5, (nm, err)          # 20 sweeps: MCMC amplifies the 1e-15 reassociation differences further than 6 sweeps do
    s2 = smp.get_chain("sigma_sq")[T // 2:T]
    assert abs(np.median(s2) - 0.01) < 0.005          # src/test-Sigma.cpp's recovery check on the rest of the run
    smp.close()
