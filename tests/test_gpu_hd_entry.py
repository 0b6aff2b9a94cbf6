"""The high-dimensional functional model through its entry points: the documented example of
BHDFMMM_Nu_Z_multiple_try / BHDFMMM_Theta_est / BHDFMMM_warm_start (R/RcppExports.R:2334-2350; src/UserFunctions.cpp:2519,
:3030, :3676) -- the package's own HDSim_data.RDS / HDtime.RDS, 20 surfaces on a 12 x 12 grid, K = 2, quadratic splines with
knots 250/500/750 on (0, 990) in both dimensions (6 x 6 = 36 tensor basis functions), n_eigen = 2, 150 iterations."""
import os

import numpy as np
import pytest

from rds_reader import read_rds

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def example():
    Y = [np.asarray(v).reshape(-1) for v in read_rds(os.path.join(GOLD, "HDSim_data.RDS"))]
    t = [np.asarray(v) for v in read_rds(os.path.join(GOLD, "HDtime.RDS"))]
    return dict(Y=Y, time=t, n_funct=20, K=2, basis_degree=[2, 2], n_eigen=2, boundary_knots=[[0.0, 990.0], [0.0, 990.0]],
                internal_knots=[[250.0, 500.0, 750.0]] * 2)


def test_hd_three_stage_pipeline(example, tmp_path):
    from bayesfmmm_amd import api
    e = example
    T, n, K, P, M = 150, 20, 2, 36, 2
    common = (e["K"], e["Y"], e["time"], e["n_funct"], e["basis_degree"], e["n_eigen"], e["boundary_knots"], e["internal_knots"])
    est1 = api.BHDFMMM_Nu_Z_multiple_try(T, 1, *common, seed=3)
    assert est1["nu"].shape == (K, P, T) and est1["Z"].shape == (n, K, T)
    assert len(est1["B"]) == n and est1["B"][0].shape == (144, P)
    # the basis rows the run used are the tensor-product rows of BSplines.h:18-88
    np.testing.assert_allclose(est1["B"][3], api.TensorBSpline(e["time"][3], e["basis_degree"], e["boundary_knots"],
                                                               e["internal_knots"]), rtol=0, atol=1e-15)
    assert np.allclose(est1["Z"].sum(axis=1), 1.0)
    assert np.isfinite(est1["loglik"]).all() and est1["loglik"][-50:].mean() > est1["loglik"][:5].mean()
    est2 = api.BHDFMMM_Theta_est(T, 1, *common, est1, seed=4)
    assert est2["Phi"].shape == (K, P, M, T) and est2["chi"].shape == (n, M, T)
    assert np.abs(est2["Z"] - est2["Z"][:, :, :1]).max() == 0.0
    burn = int(round(T * 0.8))
    np.testing.assert_allclose(est2["nu"][:, :, 0], np.median(est1["nu"][:, :, burn:], axis=2), rtol=1e-13)
    mcmc = api.BHDFMMM_warm_start(T, *common, est1, est2, seed=5)
    assert mcmc["nu"].shape == (K, P, T + 1) and mcmc["chi"].shape == (n, M, T + 1)
    assert np.isfinite(mcmc["loglik"][:T]).all()
    ysd = np.concatenate(e["Y"]).var()
    assert np.median(mcmc["sigma_sq"][T // 2:T]) < 0.3 * ysd
    # posterior-mean surfaces against the data
    s = slice(T // 2, T)
    nu, Phi, chi, Z = (mcmc[k][..., s] for k in ("nu", "Phi", "chi", "Z"))
    res = []
    for i in range(n):
        coef = np.einsum("kt,kpt->pt", Z[i], nu) + np.einsum("kt,mt,kpmt->pt", Z[i], chi[i], Phi)
        res.append(e["Y"][i] - est1["B"][i] @ coef.mean(axis=1))
    assert np.concatenate(res).var() < 0.3 * ysd
    # on-disk batches of the same run
    d = tmp_path / "trace"
    d.mkdir()
    disk = api.BHDFMMM_warm_start(T, *common, est1, est2, seed=5, dir=str(d) + "/", r_stored_iters=50, thinning_num=1)
    assert disk is not None
    nu2 = api.ReadCube(str(d / "Nu2.txt"))
    assert nu2.shape == (K, P, 50)
    np.testing.assert_allclose(nu2[:, :, 49], mcmc["nu"][:, :, 148], rtol=1e-15)


def test_hd_sampler_state_matches_entry_trajectory(example):
    """The entry point is the sampler over the tensor basis: the same seed, chain and start give the same draws."""
    import bayesfmmm_amd as bf
    from bayesfmmm_amd import api
    e = example
    T = 100
    common = (e["K"], e["Y"], e["time"], e["n_funct"], e["basis_degree"], e["n_eigen"], e["boundary_knots"], e["internal_knots"])
    a = api.BHDFMMM_Nu_Z_multiple_try(T, 1, *common, seed=11, chain_offset=0, chain_stride=100)
    b = api.BHDFMMM_Nu_Z_multiple_try(T, 1, *common, seed=11, chain_offset=0, chain_stride=100)
    np.testing.assert_array_equal(a["nu"], b["nu"])          # keyed variates: bit-reproducible
    np.testing.assert_array_equal(a["Z"], b["Z"])


def test_hd_argument_checks(example):
    from bayesfmmm_amd import _lib, api
    e = example
    base = (e["Y"], e["time"], e["n_funct"])
    with pytest.raises(_lib.BfmmmError, match="number of elemnts in 'basis_degree' does not match"):
        api.BHDFMMM_Nu_Z_multiple_try(150, 1, 2, *base, [2, 2, 2], 2, [[0, 990]] * 3, [[250.0]] * 3)
    with pytest.raises(_lib.BfmmmError, match="'tot_mcmc_iters' must be an integer greater than or equal to 100"):
        api.BHDFMMM_Nu_Z_multiple_try(50, 1, 2, *base, [2, 2], 2, e["boundary_knots"], e["internal_knots"])
    with pytest.raises(_lib.BfmmmError, match="boundary knot"):
        api.BHDFMMM_Nu_Z_multiple_try(150, 1, 2, *base, [2, 2], 2, e["boundary_knots"], [[250.0, 1500.0], [250.0]])


@pytest.mark.parametrize("covariance_adj", [False, True])
def test_hd_three_stage_pipeline_covariate_adjusted(example, covariance_adj):
    """Second half of the documented examples (R/RcppExports.R:2352-2375): X = matrix(rnorm(20), 20, 1); drivers
    BFMMM.h:6750, :6913, :7137 (mean adjusted) and :7655 (mean and covariance adjusted)."""
    from bayesfmmm_amd import api
    e = example
    T, n, K, P, M, D = 150, 20, 2, 36, 2, 1
    X = np.random.default_rng(2).standard_normal((n, D))
    common = (e["Y"], e["time"], e["n_funct"], e["basis_degree"], e["n_eigen"], e["boundary_knots"], e["internal_knots"])
    est1 = api.BHDFMMM_Nu_Z_multiple_try(T, 1, K, *common, X=X, seed=3)
    assert est1["eta"].shape == (P, D, K, T) and est1["tau_eta"].shape == (K, D, T)
    assert np.abs(est1["eta"][..., -1]).max() > 0
    est2 = api.BHDFMMM_Theta_est(T, 1, K, *common, est1, X=X, covariance_adj=covariance_adj, seed=4)
    assert est2["xi"].shape == (P, D, M, K, T)
    assert (np.abs(est2["xi"]).max() > 0) == covariance_adj
    burn = int(round(T * 0.8))
    np.testing.assert_allclose(est2["eta"][..., 0], np.median(est1["eta"][..., burn:], axis=3), rtol=1e-13)
    mcmc = api.BHDFMMM_warm_start(T, K, *common, est1, est2, X=X, covariance_adj=covariance_adj, seed=5)
    assert mcmc["eta"].shape == (P, D, K, T + 1) and np.isfinite(mcmc["loglik"][:T]).all()
    assert ("xi" in mcmc) == covariance_adj
    ysd = np.concatenate(e["Y"]).var()
    assert np.median(mcmc["sigma_sq"][T // 2:T]) < 0.3 * ysd
    tt = api.BHDFMMM_warm_start(T, K, *common, est1, est2, X=X, covariance_adj=covariance_adj, seed=5,
                                n_temp_trans=40, N_t=2, beta_N_t=0.8)
    assert tt["tt_blocks"] == 3 and np.isfinite(tt["loglik"][:T]).all()
    np.testing.assert_array_equal(tt["eta"][..., :40], mcmc["eta"][..., :40])
