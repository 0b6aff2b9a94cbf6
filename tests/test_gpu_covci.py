"""FCovCI with covariates, HDFCovCI and MVCovCI (src/PostProcessing.cpp:1781 second branch, :2468, :3097) on the chains the
reference ships (inst/test-data/{Functional,HDFunctional,Multivariate}_trace: the documented examples' directories, all three
written by the covariance-adjusted sampler with one covariate), against the numpy restatement oracle/post_ci.py."""
import os
import sys

import numpy as np
import pytest

from rds_reader import read_rds

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import post_ci as R      # noqa: E402

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
BK, IK = [0.0, 1000.0], [250.0, 500.0, 750.0]


def _load(api, dirn):
    f = api.ReadFieldCube(dirn + "Phi0.txt")
    Phi = np.stack([f[l, 0] for l in range(f.shape[0])], axis=-1)                     # K, P, M, T
    fx = api.ReadFieldCube(dirn + "Xi0.txt")
    xi = np.stack([np.stack([fx[l, k] for k in range(fx.shape[1])], axis=-1) for l in range(fx.shape[0])], axis=-1)   # P, D, M, K, T
    return Phi, xi, api.ReadCube(dirn + "Z0.txt")


def _check(got, ref, names=("CI_Upper", "CI_50", "CI_Lower", "cov_trace")):
    for nm in names:
        assert got[nm].shape == np.asarray(ref[nm]).shape, (nm, got[nm].shape, np.asarray(ref[nm]).shape)
        np.testing.assert_allclose(got[nm], ref[nm], rtol=1e-10, atol=1e-13, err_msg=nm)


@pytest.mark.parametrize("simultaneous", [False, True])
@pytest.mark.parametrize("rescale", [True, False])
def test_fcovci_with_covariates_on_the_shipped_chain(simultaneous, rescale):
    """the documented call: X <- matrix(seq(-2, 2, 0.2), ncol = 1) (PostProcessing.cpp:1773-1777)"""
    from bayesfmmm_amd import api
    dirn = os.path.join(GOLD, "Functional_trace") + "/"
    Phi, xi, Z = _load(api, dirn)
    X = np.arange(-2.0, 2.0001, 0.8).reshape(-1, 1)
    t1, t2 = np.arange(0.0, 1000.0, 100.0), np.arange(0.0, 1000.0, 90.0)
    B = lambda t: np.ascontiguousarray(api.TensorBSpline(t.reshape(-1, 1), [3], [BK], [IK]))
    for l, m in ((1, 1), (2, 1)):
        got = api.FCovCI(dirn, 1, t1, t2, 3, BK, IK, l, m, rescale=rescale, simultaneous=simultaneous, burnin_prop=0.2, X=X)
        ref = R.f_cov_ci_x(Phi, xi, X, B(t1), B(t2), l, m, 0.05, rescale, simultaneous, 0.2, Z=Z)
        assert got["CI_Upper"].shape == (len(t1), len(t2), len(X)) and got["cov_trace"].shape == (len(t1), len(t2), 120, len(X))
        _check(got, ref)
    # at x = 0 the covariate branch is the plain one
    got0 = api.FCovCI(dirn, 1, t1, t2, 3, BK, IK, 1, 2, rescale=rescale, simultaneous=simultaneous, X=np.zeros((1, 1)))
    plain = api.FCovCI(dirn, 1, t1, t2, 3, BK, IK, 1, 2, rescale=rescale, simultaneous=simultaneous)
    np.testing.assert_allclose(got0["CI_50"][..., 0], plain["CI_50"], rtol=1e-12, atol=1e-14)


@pytest.mark.parametrize("with_x", [False, True])
@pytest.mark.parametrize("simultaneous", [False, True])
def test_hdfcovci_on_the_shipped_chain(with_x, simultaneous):
    """HDFCovCI's documented example (HDFunctional_trace, time1 = time2 = HDtime.RDS[[1]], quadratic splines with knots
    250/500/750 on (0, 990) in both dimensions).  The reference builds BOTH bases from time1 (PostProcessing.cpp:2570)."""
    from bayesfmmm_amd import _lib, api
    dirn = os.path.join(GOLD, "HDFunctional_trace") + "/"
    time = np.asarray(read_rds(os.path.join(GOLD, "HDtime.RDS"))[0], dtype=np.float64)[:30]
    degs, bks, iks = [2, 2], [[0.0, 990.0], [0.0, 990.0]], [[250.0, 500.0, 750.0]] * 2
    Phi, xi, Z = _load(api, dirn)
    B1 = np.ascontiguousarray(api.TensorBSpline(time, degs, bks, iks))
    X = np.array([[-1.0], [0.5]]) if with_x else None
    time2 = time[::-1].copy()                      # any other points: they are never used by the reference
    got = api.HDFCovCI(dirn, 1, time, time2, degs, bks, iks, 1, 2, rescale=True, simultaneous=simultaneous, burnin_prop=0.3, X=X)
    if with_x:
        ref = R.f_cov_ci_x(Phi, xi, X, B1, B1, 1, 2, 0.05, True, simultaneous, 0.3, Z=Z)
    else:
        ref = R.f_cov_ci(Phi, B1, B1, 1, 2, 0.05, True, simultaneous, 0.3, Z=Z)
    _check(got, ref)
    with pytest.raises(_lib.BfmmmError, match="same number of rows"):
        api.HDFCovCI(dirn, 1, time, time[:7], degs, bks, iks, 1, 2)


@pytest.mark.parametrize("with_x", [False, True])
@pytest.mark.parametrize("rescale", [True, False])
def test_mvcovci_on_the_shipped_chain(with_x, rescale):
    from bayesfmmm_amd import _lib, api
    dirn = os.path.join(GOLD, "Multivariate_trace") + "/"
    Phi, xi, Z = _load(api, dirn)
    P = Phi.shape[1]
    I = np.eye(P)
    X = np.array([[-0.7], [0.0], [1.3]]) if with_x else None
    for l, m, alpha in ((1, 2, 0.05), (2, 2, 0.2)):
        got = api.MVCovCI(dirn, 1, l, m, alpha=alpha, rescale=rescale, burnin_prop=0.1, X=X)
        if with_x:
            ref = R.f_cov_ci_x(Phi, xi, X, I, I, l, m, alpha, rescale, False, 0.1, Z=Z)
        else:
            ref = R.f_cov_ci(Phi, I, I, l, m, alpha, rescale, False, 0.1, Z=Z)
        assert got["CI_50"].shape[:2] == (P, P)
        _check(got, ref)
    with pytest.raises(_lib.BfmmmError, match="'m' must be less than or equal to the number of clusters"):
        api.MVCovCI(dirn, 1, 1, 3)
    if with_x:
        with pytest.raises(_lib.BfmmmError, match="number of columns in 'X'"):
            api.MVCovCI(dirn, 1, 1, 1, X=np.zeros((2, 2)))
