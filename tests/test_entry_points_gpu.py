"""The three-stage pipeline of the reference's documentation examples (man/BFMMM_warm_start.Rd:
K=2, cubic splines with knots 250/500/750 on (0,1000), n_eigen=3, 150 iterations, n_try=1, the
package's own Sim_data.RDS / time.RDS) run through the entry points of include/bfmmm_entry.h,
plus the reference's argument checks."""
import os

import numpy as np
import pytest

from rds_reader import read_rds

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def example():
    Y = [np.asarray(v).reshape(-1) for v in read_rds(os.path.join(GOLD, "Sim_data.RDS"))]
    t = [np.asarray(v).reshape(-1) for v in read_rds(os.path.join(GOLD, "time.RDS"))]
    return dict(Y=Y, time=t, n_funct=40, K=2, basis_degree=3, n_eigen=3, boundary_knots=[0.0, 1000.0],
                internal_knots=[250.0, 500.0, 750.0])


def test_three_stage_pipeline(example):
    from bayesfmmm_amd import api
    e = example
    T, n, K, P, M = 150, 40, 2, 7, 3
    est1 = api.BFMMM_Nu_Z_multiple_try(T, 1, e["K"], e["Y"], e["time"], e["n_funct"], e["basis_degree"], e["n_eigen"],
                                       e["boundary_knots"], e["internal_knots"], seed=3)
    assert set(est1) >= {"B", "nu", "pi", "alpha_3", "A", "delta", "sigma_sq", "tau", "Z", "loglik"}
    assert est1["nu"].shape == (K, P, T) and est1["Z"].shape == (n, K, T) and est1["tau"].shape == (T, K)
    assert len(est1["B"]) == n and est1["B"][0].shape == (100, P)
    assert np.allclose(est1["Z"].sum(axis=1), 1.0)
    assert np.isfinite(est1["loglik"]).all() and est1["loglik"][-50:].mean() > est1["loglik"][:5].mean()
    assert est1["best_chain"] in (0.0, 1.0)
    est2 = api.BFMMM_Theta_est(T, 1, e["K"], e["Y"], e["time"], e["n_funct"], e["basis_degree"], e["n_eigen"],
                               e["boundary_knots"], e["internal_knots"], est1, seed=4)
    assert est2["Phi"].shape == (K, P, M, T) and est2["chi"].shape == (n, M, T) and est2["gamma"].shape == (K, P, M, T)
    # Z and nu are pinned to the stage-1 medians in every slot (BFMMM.h:1244-1250)
    assert np.abs(est2["Z"] - est2["Z"][:, :, :1]).max() == 0.0
    assert np.abs(est2["nu"] - est2["nu"][:, :, :1]).max() == 0.0
    burn = int(round(T * 0.8))
    zmed = np.median(est1["Z"][:, :, burn:], axis=2)
    zmed /= zmed.sum(axis=1, keepdims=True)
    np.testing.assert_allclose(est2["Z"][:, :, 0], zmed, rtol=1e-13)
    np.testing.assert_allclose(est2["nu"][:, :, 0], np.median(est1["nu"][:, :, burn:], axis=2), rtol=1e-13)
    mcmc = api.BFMMM_warm_start(T, e["K"], e["Y"], e["time"], e["n_funct"], e["basis_degree"], e["n_eigen"],
                                e["boundary_knots"], e["internal_knots"], est1, est2, seed=5)
    assert set(mcmc) >= {"B_obs", "Z", "nu", "chi", "pi", "alpha_3", "A", "delta", "sigma_sq", "tau", "gamma", "Phi", "loglik"}
    assert mcmc["nu"].shape == (K, P, T + 1) and mcmc["chi"].shape == (n, M, T + 1)      # r_stored_iters = T + 1
    np.testing.assert_array_equal(mcmc["nu"][:, :, T], mcmc["nu"][:, :, T - 1])
    assert np.isfinite(mcmc["loglik"][:T]).all()
    # the fitted curves explain the data: residual variance well below the data variance
    ysd = np.concatenate(e["Y"]).var()
    assert np.median(mcmc["sigma_sq"][T // 2:T]) < 0.2 * ysd


def test_warm_start_with_tempered_transitions(example):
    """BFMMM_warm_start(n_temp_trans = 25, N_t = 3, beta_N_t = 0.7): the documented tempered-transition options
    (UserFunctions.cpp:1151-1153) run through the entry point; a run with the option off is unchanged by its presence."""
    from bayesfmmm_amd import _lib, api
    e = example
    T = 150
    common = (e["K"], e["Y"], e["time"], e["n_funct"], e["basis_degree"], e["n_eigen"], e["boundary_knots"], e["internal_knots"])
    est1 = api.BFMMM_Nu_Z_multiple_try(T, 1, *common, seed=3)
    est2 = api.BFMMM_Theta_est(T, 1, *common, est1, seed=4)
    plain = api.BFMMM_warm_start(T, *common, est1, est2, seed=5)
    tt = api.BFMMM_warm_start(T, *common, est1, est2, seed=5, n_temp_trans=25, N_t=3, beta_N_t=0.7)
    assert tt["tt_blocks"] == 5 and 0 <= tt["tt_accepted"] <= 5          # iterations 25, 50, 75, 100, 125
    assert np.isfinite(tt["loglik"][:T]).all() and np.allclose(tt["Z"][:, :, :T].sum(axis=1), 1.0)
    # identical up to and including iteration 24 (same keyed variates), the first block acts on slot 25
    np.testing.assert_array_equal(tt["nu"][:, :, :25], plain["nu"][:, :, :25])
    assert "tt_blocks" not in plain
    ysd = np.concatenate(e["Y"]).var()
    assert np.median(tt["sigma_sq"][T // 2:T]) < 0.2 * ysd
    with pytest.raises(_lib.BfmmmError, match="'beta_N_t' must be between 0 and 1"):
        api.BFMMM_warm_start(T, *common, est1, est2, n_temp_trans=10, N_t=2, beta_N_t=1.5)
    with pytest.raises(_lib.BfmmmError, match="'N_t' must be a positive integer"):
        api.BFMMM_warm_start(T, *common, est1, est2, n_temp_trans=10, N_t=0)


def test_reference_argument_checks(example):
    from bayesfmmm_amd import _lib, api
    e = example
    common = (e["Y"], e["time"], e["n_funct"], e["basis_degree"], e["n_eigen"], e["boundary_knots"], e["internal_knots"])
    with pytest.raises(_lib.BfmmmError, match="'tot_mcmc_iters' must be an integer greater than or equal to 100"):
        api.BFMMM_Nu_Z_multiple_try(50, 1, 2, *common)
    with pytest.raises(_lib.BfmmmError, match="'n_try' must be an integer greater than or equal to 1"):
        api.BFMMM_Nu_Z_multiple_try(150, 0, 2, *common)
    with pytest.raises(_lib.BfmmmError, match="'K' must be an integer greater than or equal to 2"):
        api.BFMMM_Nu_Z_multiple_try(150, 1, 1, *common)
    with pytest.raises(_lib.BfmmmError, match="'a_Z_PM' must be positive"):
        api.BFMMM_Nu_Z_multiple_try(150, 1, 2, *common, a_Z_PM=-1.0)
    with pytest.raises(_lib.BfmmmError, match="all elements of 'c' must be positive"):
        api.BFMMM_Nu_Z_multiple_try(150, 1, 2, *common, c=[1.0, -1.0])
    with pytest.raises(_lib.BfmmmError, match="number of elements of the vector 'c' must be equal to K"):
        api.BFMMM_Nu_Z_multiple_try(150, 1, 2, *common, c=[1.0, 1.0, 1.0])
    with pytest.raises(_lib.BfmmmError, match="more than or equal to second boundary knot"):
        api.BFMMM_Nu_Z_multiple_try(150, 1, 2, e["Y"], e["time"], 40, 3, 3, [0.0, 1000.0], [250.0, 1500.0])


def test_multi_try_keeps_best_chain(example):
    from bayesfmmm_amd import api
    e = example
    common = (e["Y"], e["time"], e["n_funct"], e["basis_degree"], e["n_eigen"], e["boundary_knots"], e["internal_knots"])
    T = 120
    multi = api.BFMMM_Nu_Z_multiple_try(T, 3, 2, *common, seed=9, max_concurrent=4)
    scores = []
    for c in range(4):      # the same four chains one by one
        r = api.BFMMM_Nu_Z_multiple_try(T, 3, 2, *common, seed=9, chain_offset=c, chain_stride=100)
        assert r["best_chain"] == c
        scores.append(r["best_score"])
    assert multi["best_chain"] == int(np.argmax(scores))
    assert multi["best_score"] == max(scores)
    assert abs(multi["best_score"] - multi["loglik"][T - 99:].mean()) < 1e-9 * abs(multi["best_score"])


@pytest.mark.parametrize("covariance_adj", [False, True])
def test_three_stage_pipeline_covariate_adjusted(example, covariance_adj):
    # second half of the reference's examples (man/BFMMM_warm_start.Rd "Covariate Adj"): X = matrix(rnorm(40), 40, 1)
    from bayesfmmm_amd import api
    e = example
    T, n, K, P, M, D = 150, 40, 2, 7, 3, 1
    X = np.random.default_rng(1).standard_normal((n, D))
    common = (e["Y"], e["time"], e["n_funct"], e["basis_degree"], e["n_eigen"], e["boundary_knots"], e["internal_knots"])
    est1 = api.BFMMM_Nu_Z_multiple_try(T, 1, K, *common, X=X, seed=3)
    assert est1["eta"].shape == (P, D, K, T) and est1["tau_eta"].shape == (K, D, T)
    assert np.abs(est1["eta"][..., -1]).max() > 0
    est2 = api.BFMMM_Theta_est(T, 1, K, *common, est1, X=X, covariance_adj=covariance_adj, seed=4)
    assert est2["xi"].shape == (P, D, M, K, T) and est2["delta_xi"].shape == (K, M, D, T)
    assert (np.abs(est2["xi"]).max() > 0) == covariance_adj          # xi stays 0 under mean-only adjustment
    burn = int(round(T * 0.8))
    np.testing.assert_allclose(est2["eta"][..., 0], np.median(est1["eta"][..., burn:], axis=3), rtol=1e-13)
    mcmc = api.BFMMM_warm_start(T, K, *common, est1, est2, X=X, covariance_adj=covariance_adj, seed=5)
    assert mcmc["eta"].shape == (P, D, K, T + 1) and np.isfinite(mcmc["loglik"][:T]).all()
    assert ("xi" in mcmc) == covariance_adj
    # tempered transitions of the covariate-adjusted drivers (BFMMM.h:4313-4440, :4897-5084)
    tt = api.BFMMM_warm_start(T, K, *common, est1, est2, X=X, covariance_adj=covariance_adj, seed=5,
                              n_temp_trans=40, N_t=2, beta_N_t=0.8)
    assert tt["tt_blocks"] == 3 and np.isfinite(tt["loglik"][:T]).all()
    np.testing.assert_array_equal(tt["eta"][..., :40], mcmc["eta"][..., :40])
    with pytest.raises(Exception, match="'X' must be have 'n_funct' number of rows"):
        api.BFMMM_Nu_Z_multiple_try(T, 1, K, *common, X=X[:10])


def test_warm_start_progress_callback(example):
    """The ABI's counterpart of Rcpp::checkUserInterrupt() / the Rcout progress lines: called between device batches with the
    last iteration and its log-likelihood; it changes nothing about the draws; a non-zero return aborts the run."""
    from bayesfmmm_amd import _lib, api
    e = example
    T = 120
    common = (e["K"], e["Y"], e["time"], e["n_funct"], e["basis_degree"], e["n_eigen"], e["boundary_knots"], e["internal_knots"])
    est1 = api.BFMMM_Nu_Z_multiple_try(T, 1, *common, seed=3)
    est2 = api.BFMMM_Theta_est(T, 1, *common, est1, seed=4)
    plain = api.BFMMM_warm_start(T, *common, est1, est2, seed=5)
    seen = []
    rep = api.BFMMM_warm_start(T, *common, est1, est2, seed=5, progress=(50, lambda it, ll: seen.append((it, ll)) or False))
    assert [it for it, _ in seen] == [49, 99, 119]
    assert all(ll == plain["loglik"][it] for it, ll in seen)
    np.testing.assert_array_equal(rep["nu"], plain["nu"])
    with pytest.raises(_lib.BfmmmError, match="interrupted by the progress callback"):
        api.BFMMM_warm_start(T, *common, est1, est2, seed=5, progress=(40, lambda it, ll: it >= 79))
