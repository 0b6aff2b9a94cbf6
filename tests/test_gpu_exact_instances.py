"""The exact-shape kernel instances (K, M, D compile-time: k_curve_chi, k_curve_z, k_cov_group) against the
general instances of the same kernels: same inputs, same seeds, whole trajectories, every chain output.  The two are the same
source with constants in place of kernel arguments, so the draws must agree to rounding (1e-12 relative; the oracle parity of the
exact instances is tests/test_gpu_baseline_shapes.py, that of the general ones tests/test_gpu_shapes.py).

Reference loops as there: BFMMM.h:1502-1553 (warm-start sweep), :4809-4894 (Mean_CovAdj), :2597-2650 (multivariate), :1073-1113 (Nu_Z)."""
import numpy as np
import pytest

from gpu_parity import rel_err
from test_gpu_shapes import simulate

pytestmark = pytest.mark.gpu

WARM = ["nu", "Phi", "chi", "Z", "pi", "alpha_3", "delta", "A", "gamma", "tau", "sigma_sq", "loglik"]
COV = ["eta", "xi", "tau_eta", "gamma_xi", "delta_xi", "A_xi"]


def both_ways(build, names):
    """build() -> a sampler that has been run; returns nothing, asserts exact == general"""
    from bayesfmmm_amd import _lib
    lib = _lib.load()
    out = {}
    try:
        for exact in (1, 0):
            lib.bfmmm_set_exact_instances(exact)
            smp = build()
            out[exact] = {nm: np.array(smp.get_chain(nm), copy=True) for nm in names}
            smp.close()
    finally:
        lib.bfmmm_set_exact_instances(1)
    for nm in names:
        assert np.all(np.isfinite(out[1][nm])), nm
        err = rel_err(out[1][nm], out[0][nm])
        assert err < 1e-12, (nm, err)


@pytest.mark.parametrize("K,M,nknots", [(3, 6, 26), (2, 3, 10), (4, 8, 40), (2, 1, 6)])
def test_functional_warm_sweeps(K, M, nknots):
    """k_curve_chi<3, L, false, true, K, M> with the fused Z update against k_curve_chi<3, L, false, true> (P <= 32 and P > 32)"""
    import bayesfmmm_amd as bf
    T = 6
    sim = simulate(80, K, M, 3, nknots, seed=500 + K * 10 + M, n_pts=90)

    def build():
        cfg = bf.default_config(model=bf.MODEL_FUNCTIONAL, K=K, n_eigen=M, basis_degree=3, tot_mcmc_iters=T)
        smp = bf.Sampler(cfg, sim["y"], sim["t"], sim["internal_knots"], sim["boundary_knots"])
        smp.init_state(1, 11, chain=0)
        smp.run(bf.sampler.SWEEP_WARM, T, seed=21)
        return smp
    both_ways(build, WARM)


def test_functional_nu_z_batch():
    """the lean trailing k_curve_z with K exact, four chains in one batch"""
    import bayesfmmm_amd as bf
    S = bf.sampler
    T, K, M = 8, 3, 4
    sim = simulate(96, K, M, 3, 20, seed=531, n_pts=80)

    def build():
        cfg = bf.default_config(model=bf.MODEL_FUNCTIONAL, K=K, n_eigen=M, basis_degree=3, tot_mcmc_iters=T)
        smp = bf.Sampler(cfg, sim["y"], sim["t"], sim["internal_knots"], sim["boundary_knots"], n_chains=4)
        for q in range(4):
            smp.select_chain(q)
            smp.init_state(0, 5, chain=q)
        smp.run(S.SWEEP_NU_Z, T, seed=9, chain=0, phi_chi_zero=True)
        smp.select_chain(2)
        return smp
    both_ways(build, ["nu", "Z", "pi", "alpha_3", "tau", "sigma_sq", "loglik"])


def test_multivariate_warm_sweeps():
    """BW = 0 instances: K = 4, M = 8, dim = 50 (64-lane groups)"""
    import bayesfmmm_amd as bf
    from test_gpu_multivariate import simulate_mv
    T, n, P, K, M = 5, 96, 50, 4, 8
    sim = simulate_mv(n, P, K, M, 0.05, seed=541)

    def build():
        cfg = bf.default_config(model=bf.MODEL_MULTIVARIATE, K=K, n_eigen=M, tot_mcmc_iters=T)
        smp = bf.Sampler(cfg, sim["Y"])
        smp.init_state(1, 3, chain=0)
        smp.run(bf.sampler.SWEEP_WARM, T, seed=5)
        return smp
    both_ways(build, WARM)


@pytest.mark.parametrize("D", [1, 2, 5])
def test_covariate_adjusted_sweeps(D):
    """k_cov_group<3, 32, D>, k_curve_z<.., true, K, false, true>, k_curve_chi<.., true, true, K, M> against the general instances"""
    import bayesfmmm_amd as bf
    S = bf.sampler
    T, n, K, M = 4, 72, 3, 4
    sim = simulate(n, K, M, 3, 20, seed=551 + D, n_pts=80)
    X = np.random.default_rng(61 + D).standard_normal((n, D))

    def build():
        cfg = bf.default_config(model=bf.MODEL_FUNCTIONAL, K=K, n_eigen=M, basis_degree=3, tot_mcmc_iters=T)
        smp = bf.Sampler(cfg, sim["y"], sim["t"], sim["internal_knots"], sim["boundary_knots"])
        smp.set_covariates(X, True)
        smp.init_state(1, 7, chain=0)
        smp.run(S.SWEEP_WARM | S.COV_MEAN | S.COV_XI, T, seed=3)
        return smp
    both_ways(build, WARM + COV)
