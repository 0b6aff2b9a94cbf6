"""Likelihood-based post-processing on the device (include/bfmmm_post.h): FLLik, FDIC, FAIC, FBIC
(src/PostProcessing.cpp:4892, :3660, :4041, :4458) over the on-disk batches a warm-start run writes, against the oracle's
restatement of the reference's loops (oracle/post.c) evaluated on the same draws."""
import numpy as np
import pytest

import oracle_lib as O
from simdata import simulate_functional

pytestmark = pytest.mark.gpu


def _run_and_save(tmp_path, D, cov_adj, T=120, seed=5):
    from bayesfmmm_amd import api
    sim = simulate_functional(n=23, M=2, sigma_sq=0.01, seed=seed, ragged=True, D=max(D, 1))
    X = sim["X"][:, :D] if D else None
    common = (sim["K"], sim["y"], sim["t"], sim["n"], 3, sim["M"], sim["boundary_knots"], sim["internal_knots"])
    kw = dict(X=X) if D else {}
    est1 = api.BFMMM_Nu_Z_multiple_try(T, 1, *common, seed=1, **kw)
    kw2 = dict(kw, covariance_adj=cov_adj) if D else {}
    est2 = api.BFMMM_Theta_est(T, 1, *common, est1, seed=2, **kw2)
    d = tmp_path / "trace"
    d.mkdir()
    api.BFMMM_warm_start(T, *common, est1, est2, seed=3, dir=str(d) + "/", r_stored_iters=40, thinning_num=1, **kw2)
    return sim, X, str(d) + "/"


def _oracle_chain(sim, X, dirn, n_files, cov_adj):
    """the saved draws as an oracle chain (slot = draw)"""
    from bayesfmmm_amd import api
    cat = lambda name, rd: np.concatenate([rd(f"{dirn}{name}{q}.txt") for q in range(n_files)], axis=-1)
    nu, Z, chi = cat("Nu", api.ReadCube), cat("Z", api.ReadCube), cat("Chi", api.ReadCube)
    sigma = np.concatenate([api.ReadVec(f"{dirn}Sigma{q}.txt") for q in range(n_files)])
    T = nu.shape[2]
    K, P, M = nu.shape[0], nu.shape[1], chi.shape[1]
    Phi = np.zeros((K, P, M, T))
    for q in range(n_files):
        f = api.ReadFieldCube(f"{dirn}Phi{q}.txt")
        per = T // n_files
        for l in range(per):
            Phi[..., q * per + l] = f[l, 0]
    B = [np.ascontiguousarray(api.TensorBSpline(np.asarray(t).reshape(-1, 1), [3], [sim["boundary_knots"]], [sim["internal_knots"]]))
         for t in sim["t"]]
    model = O.Model(sim["y"], B, K, M, X=X)
    ch = O.Chain(model, T)
    ch.nu[:], ch.Z[:], ch.chi[:], ch.sigma[:], ch.Phi[:] = nu, Z, chi, sigma, Phi
    if X is not None:
        D = X.shape[1]
        per = T // n_files
        for q in range(n_files):
            f = api.ReadFieldCube(f"{dirn}Eta{q}.txt")
            for l in range(per):
                ch.eta[..., q * per + l] = f[l, 0]
            if cov_adj:
                fx = api.ReadFieldCube(f"{dirn}Xi{q}.txt")
                for l in range(per):
                    for k in range(K):
                        ch.xi[..., k, q * per + l] = fx[l, k]
    return model, ch, B


@pytest.mark.parametrize("D,cov_adj", [(0, False), (1, False), (2, True)])
def test_llik_dic_aic_bic_match_oracle(tmp_path, D, cov_adj):
    from bayesfmmm_amd import api
    sim, X, dirn = _run_and_save(tmp_path, D, cov_adj)
    n_files = 3
    model, ch, B = _oracle_chain(sim, X, dirn, n_files, cov_adj)
    args = (dirn, n_files, 3, sim["boundary_knots"], sim["internal_knots"], sim["t"], sim["y"])
    kw = dict(X=X, cov_adj=cov_adj) if D else {}
    ll = api.FLLik(*args, **kw)
    ll_ref = O.post_llik(model, ch)
    assert ll.shape == (120,)
    np.testing.assert_allclose(ll, ll_ref, rtol=1e-10)
    for burn in (0.1, 0.5):
        dic = api.FDIC(*args, burnin_prop=burn, **kw)
        assert abs(dic - O.post_dic(model, ch, burn)) < 1e-9 * abs(dic)
        aic_ref, bic_ref = O.post_aic_bic(model, ch, burn, D > 0, cov_adj)
        aic, bic = api.FAIC(*args, burnin_prop=burn, **kw), api.FBIC(*args, burnin_prop=burn, **kw)
        assert abs(aic - aic_ref) < 1e-10 * abs(aic_ref) and abs(bic - bic_ref) < 1e-10 * abs(bic_ref)
    # the in-memory form gives the same per-draw values, and its per-observation means are the oracle's
    first = 120 - int(round(0.9 * 120))
    ll2, pdf, fit = api.post_pointwise(sim["y"], B, ch.nu, ch.Phi, ch.Z, ch.chi, ch.sigma, first_kept=first, X=X,
                                       eta=ch.eta if D else None, xi=ch.xi if cov_adj else None)
    np.testing.assert_array_equal(ll2, ll)
    i = 7
    fit_ref = np.array([np.mean([O.lib().orc_fitted(O.C.byref(model.data), O.C.byref(ch.c), t, i, l) for t in range(first, 120)])
                        for l in range(len(sim["y"][i]))])
    np.testing.assert_allclose(fit[i], fit_ref, rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize("D,cov_adj", [(0, False), (1, False), (2, True)])
def test_conditional_predictive_ordinates_match_oracle(tmp_path, D, cov_adj):
    """ConditionalPredictiveOrdinates (src/PostProcessing.cpp:6339, calcLikelihoodCPO): the device evaluates the marginal
    density through the rank-M form, the oracle through the reference's dense n_i x n_i covariance (log det and inverse by
    Cholesky): the two agree to rounding of the dense factorisation."""
    from bayesfmmm_amd import api
    sim, X, dirn = _run_and_save(tmp_path, D, cov_adj)
    model, ch, B = _oracle_chain(sim, X, dirn, 3, cov_adj)
    args = (dirn, 3, 3, sim["boundary_knots"], sim["internal_knots"], sim["t"], sim["y"])
    kw = dict(X=X, cov_adj=cov_adj) if D else {}
    for burn in (0.1, 0.55):
        cpo = api.ConditionalPredictiveOrdinates(*args, burnin_prop=burn, **kw)
        ref = O.post_cpo(model, ch, burn)
        assert cpo.shape == (sim["n"],)
        np.testing.assert_allclose(cpo, ref, rtol=1e-8, atol=1e-8)
    np.testing.assert_allclose(api.ConditionalPredictiveOrdinates(*args, log_CPO=False, **kw), np.exp(O.post_cpo(model, ch, 0.1)), rtol=1e-7)


def test_post_argument_checks(tmp_path):
    from bayesfmmm_amd import _lib, api
    sim, X, dirn = _run_and_save(tmp_path, 0, False)
    a = (3, sim["boundary_knots"], sim["internal_knots"], sim["t"], sim["y"])
    with pytest.raises(_lib.BfmmmError, match="'n_files' must be greater than 0"):
        api.FLLik(dirn, 0, *a)
    with pytest.raises(_lib.BfmmmError, match="'burnin_prop' must be between 0 and 1"):
        api.FDIC(dirn, 3, *a, burnin_prop=1.0)
    with pytest.raises(_lib.BfmmmError, match="'basis_degree' must be an integer greater than or equal to 1"):
        api.FAIC(dirn, 3, 0, *a[1:])
    with pytest.raises(_lib.BfmmmError, match="cannot open"):
        api.FBIC(dirn + "nowhere/", 3, *a)


def test_post_many_observations_and_single_observation_curves():
    """Geometry edges of k_post_pointwise: a curve longer than a workgroup (several observations per thread, basis rows
    read from global memory), curves of one observation (32 draws per tile), more draws than one chunk."""
    from bayesfmmm_amd import api
    rng = np.random.default_rng(3)
    K, P, M, T = 3, 9, 2, 301
    lens = [700, 1, 2, 33, 1, 257, 64]
    n = len(lens)
    bk, ik = [0.0, 1.0], np.linspace(0, 1, 7)[1:-1]
    ts = [np.sort(rng.uniform(0, 1, size=m)) for m in lens]
    B = [np.ascontiguousarray(api.TensorBSpline(t.reshape(-1, 1), [3], [bk], [ik])) for t in ts]
    Y = [rng.standard_normal(m) for m in lens]
    model = O.Model(Y, B, K, M)
    ch = O.Chain(model, T)
    ch.nu[:] = rng.standard_normal((K, P, T))
    ch.Phi[:] = 0.3 * rng.standard_normal((K, P, M, T))
    ch.chi[:] = rng.standard_normal((n, M, T))
    Z = rng.dirichlet(np.full(K, 0.7), size=(n, T)).transpose(0, 2, 1)
    Z[2, 1, :] = 0.0                                  # exact zeros: the reference's skip
    ch.Z[:] = Z
    ch.sigma[:] = rng.gamma(3.0, 0.2, size=T)
    ll, pdf, fit = api.post_pointwise(Y, B, ch.nu, ch.Phi, ch.Z, ch.chi, ch.sigma, first_kept=30)
    np.testing.assert_allclose(ll, O.post_llik(model, ch), rtol=1e-11)
    dic = 2 * np.sum(np.log(np.concatenate(pdf))) - 4 * ll[30:].mean()
    assert abs(dic - O.post_dic(model, ch, 1 - 271 / 301)) < 1e-9 * abs(dic)


# ---- multivariate model: the reference's own shipped trace (inst/test-data/Multivariate_trace, MVSim_data.RDS) -------
def _mv_trace_chain(X, cov_adj):
    import os
    from bayesfmmm_amd import api
    from rds_reader import read_rds
    gold = os.path.join(os.path.dirname(__file__), "golden")
    dirn = os.path.join(gold, "Multivariate_trace") + "/"
    Y = np.asarray(read_rds(os.path.join(gold, "MVSim_data.RDS")), dtype=np.float64)
    n, P = Y.shape
    nu, Z, chi = api.ReadCube(dirn + "Nu0.txt"), api.ReadCube(dirn + "Z0.txt"), api.ReadCube(dirn + "Chi0.txt")
    sigma = api.ReadVec(dirn + "Sigma0.txt")
    K, T, M = nu.shape[0], nu.shape[2], chi.shape[1]
    model = O.Model([Y[i] for i in range(n)], [np.eye(P)] * n, K, M, X=X, mv=True)
    ch = O.Chain(model, T)
    ch.nu[:], ch.Z[:], ch.chi[:], ch.sigma[:] = nu, Z, chi, sigma
    phi = api.ReadFieldCube(dirn + "Phi0.txt")
    for l in range(T):
        ch.Phi[..., l] = phi[l, 0]
    if X is not None:
        eta = api.ReadFieldCube(dirn + "Eta0.txt")
        xi = api.ReadFieldCube(dirn + "Xi0.txt")
        for l in range(T):
            ch.eta[..., l] = eta[l, 0]
            if cov_adj:
                for k in range(K):
                    ch.xi[..., k, l] = xi[l, k]
    return dirn, Y, model, ch


@pytest.mark.parametrize("with_x,cov_adj", [(False, False), (True, False), (True, True)])
def test_mv_criteria_on_the_reference_trace(with_x, cov_adj):
    """The documented examples of MVLLik / MVDIC / MVAIC / MVBIC (src/PostProcessing.cpp:5116, :5452, :5789, :6099:
    `dir <- .../Multivariate_trace/`, n_files = 1, Y = MVSim_data.RDS, X = matrix(rnorm(20), 20, 1)) on the trace files
    the package ships, against the oracle's restatement on the same draws."""
    from bayesfmmm_amd import api
    X = np.random.default_rng(4).standard_normal((20, 1)) if with_x else None
    dirn, Y, model, ch = _mv_trace_chain(X, cov_adj)
    kw = dict(X=X, cov_adj=cov_adj) if with_x else {}
    ll = api.MVLLik(dirn, 1, Y, **kw)
    assert ll.shape == (150,)
    np.testing.assert_allclose(ll, O.post_llik(model, ch), rtol=1e-10)
    for burn in (0.1, 0.6):
        dic = api.MVDIC(dirn, 1, Y, burnin_prop=burn, **kw)
        assert abs(dic - O.post_dic(model, ch, burn)) < 1e-9 * abs(dic)
        aic_ref, bic_ref = O.post_aic_bic(model, ch, burn, with_x, cov_adj)
        assert abs(api.MVAIC(dirn, 1, Y, burnin_prop=burn, **kw) - aic_ref) < 1e-10 * abs(aic_ref)
        assert abs(api.MVBIC(dirn, 1, Y, burnin_prop=burn, **kw) - bic_ref) < 1e-10 * abs(bic_ref)


def test_mv_odd_dimension_integer_division_quirk():
    """calcLikelihoodMV charges (P / 2) log(2 pi sigma) with integer division (CalculateLikelihood.h:155): odd P."""
    from bayesfmmm_amd import api
    rng = np.random.default_rng(8)
    n, P, K, M, T = 13, 7, 2, 2, 40
    Y = rng.standard_normal((n, P))
    model = O.Model([Y[i] for i in range(n)], [np.eye(P)] * n, K, M, mv=True)
    ch = O.Chain(model, T)
    ch.nu[:] = rng.standard_normal((K, P, T))
    ch.Phi[:] = 0.3 * rng.standard_normal((K, P, M, T))
    ch.chi[:] = rng.standard_normal((n, M, T))
    ch.Z[:] = rng.dirichlet(np.ones(K), size=(n, T)).transpose(0, 2, 1)
    ch.sigma[:] = rng.gamma(3.0, 0.3, size=T)
    import tempfile, os
    with tempfile.TemporaryDirectory() as d:
        dirn = d + "/"
        api.write_arma_ascii(dirn + "Nu0.txt", ch.nu)
        api.write_arma_ascii(dirn + "Z0.txt", ch.Z)
        api.write_arma_ascii(dirn + "Chi0.txt", ch.chi)
        api.write_arma_ascii(dirn + "Sigma0.txt", ch.sigma.reshape(-1, 1))
        fld = np.empty((T, 1), dtype=object)
        for l in range(T):
            fld[l, 0] = np.asfortranarray(ch.Phi[..., l])
        api.write_arma_field(dirn + "Phi0.txt", fld)
        ll = api.MVLLik(dirn, 1, Y)
        np.testing.assert_allclose(ll, O.post_llik(model, ch), rtol=1e-11)
        dic = api.MVDIC(dirn, 1, Y, burnin_prop=0.25)
        assert abs(dic - O.post_dic(model, ch, 0.25)) < 1e-9 * abs(dic)


def test_post_pass_takes_what_the_sampler_can_produce(tmp_path):
    """The post-processing pass at the sampler's own limits (round 4: K + M <= 24 in the pointwise pass, n_eigen <= 16 in the
    conditional predictive ordinates; 18 and 8 before): (a) FLLik and the per-observation means with K = 4, M = 16 on random draws
    (in-memory form), (b) ConditionalPredictiveOrdinates over the batches of a warm-start run with n_eigen = 10 -- the M x M system
    of a draw then lives in LDS (kernels_post.hip, M > 8) -- against the oracle's dense n_i x n_i form
    (src/PostProcessing.cpp:4892, :6339; CalculateLikelihood.h:344-389)."""
    from bayesfmmm_amd import api
    # (a)
    rng = np.random.default_rng(12)
    n, K, M, T = 9, 4, 16, 12
    t = [np.sort(rng.uniform(0, 1000, size=rng.integers(5, 40))) for _ in range(n)]
    ik, bk = np.array([250.0, 500.0, 750.0]), np.array([0.0, 1000.0])
    B = [O.bspline_basis(ti, ik, 3, bk) for ti in t]
    P = B[0].shape[1]
    ys = [rng.standard_normal(len(ti)) for ti in t]
    model = O.Model(ys, B, K, M)
    ch = O.Chain(model, T)
    ch.nu[:] = rng.standard_normal(ch.nu.shape)
    ch.Phi[:] = 0.3 * rng.standard_normal(ch.Phi.shape)
    ch.Z[:] = rng.dirichlet(np.ones(K), size=(n, T)).transpose(0, 2, 1)
    ch.chi[:] = rng.standard_normal(ch.chi.shape)
    ch.sigma[:] = rng.uniform(0.5, 1.5, size=T)
    ll, pdf, fit = api.post_pointwise(ys, B, ch.nu, ch.Phi, ch.Z, ch.chi, ch.sigma, first_kept=2)
    np.testing.assert_allclose(ll, O.post_llik(model, ch), rtol=1e-10)
    i = 3
    fit_ref = np.array([np.mean([O.lib().orc_fitted(O.C.byref(model.data), O.C.byref(ch.c), tt, i, l) for tt in range(2, T)])
                        for l in range(len(ys[i]))])
    np.testing.assert_allclose(fit[i], fit_ref, rtol=1e-10, atol=1e-12)
    # (b)
    Tw = 120          # (the entry points require tot_mcmc_iters >= 100: UserFunctions.cpp:198-286)
    sim = simulate_functional(n=12, M=10, sigma_sq=0.01, seed=31, ragged=True)
    common = (sim["K"], sim["y"], sim["t"], sim["n"], 3, sim["M"], sim["boundary_knots"], sim["internal_knots"])
    est1 = api.BFMMM_Nu_Z_multiple_try(Tw, 1, *common, seed=1)
    est2 = api.BFMMM_Theta_est(Tw, 1, *common, est1, seed=2)
    d = tmp_path / "trace10"
    d.mkdir()
    api.BFMMM_warm_start(Tw, *common, est1, est2, seed=3, dir=str(d) + "/", r_stored_iters=40, thinning_num=1)
    dirn = str(d) + "/"
    model2, ch2, _ = _oracle_chain(sim, None, dirn, 3, False)
    assert ch2.chi.shape[1] == 10
    args = (dirn, 3, 3, sim["boundary_knots"], sim["internal_knots"], sim["t"], sim["y"])
    cpo = api.ConditionalPredictiveOrdinates(*args, burnin_prop=0.2)
    np.testing.assert_allclose(cpo, O.post_cpo(model2, ch2, 0.2), rtol=1e-8, atol=1e-8)
