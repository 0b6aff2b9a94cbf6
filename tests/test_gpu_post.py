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
