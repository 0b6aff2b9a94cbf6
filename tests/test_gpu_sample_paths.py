"""FSamplePaths (src/PostProcessing.cpp:6599-6864) on the functional chain the reference ships, its documented example
(man/FSamplePaths.Rd: Functional_trace, time.RDS; no X / X / X with cov_adj): fitted values from the oracle's restatement
(orc_fitted: the Z != 0 skip, the covariate terms), the predictive noise from the shared keyed generator, bands by the
numpy restatement of the reference's quantile / simultaneous rules (oracle/post_ci.py)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

import oracle_lib as O
from rds_reader import read_rds
from test_gpu_post import _oracle_chain

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import post_ci as R      # noqa: E402

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
TRACE = os.path.join(GOLD, "Functional_trace") + "/"
BK, IK = [0.0, 1000.0], [250.0, 500.0, 750.0]
UPD_SAMPLE_PATH = 41


def _oracle_paths(model, ch, B, X, first, seed):
    L = O.lib()
    L.orc_rnorm.restype = C.c_double
    n, K, T = model.n, model.K, ch.T
    paths, mo = [], []
    for i in range(n):
        ni = B[i].shape[0]
        p_i, m_i = np.zeros((T - first, ni)), np.zeros((T - first, ni))
        for t in range(first, T):
            r = O.OrcRng(seed, 0, t, 0)
            for l in range(ni):
                mean = L.orc_fitted(C.byref(model.data), C.byref(ch.c), t, i, l)
                mean_only = 0.0
                for k in range(K):
                    z = ch.Z[i, k, t]
                    if z != 0:
                        coef = ch.nu[k, :, t] + (ch.eta[:, :, k, t] @ X[i] if X is not None else 0.0)
                        mean_only += z * float(coef @ B[i][l])
                m_i[t - first, l] = mean_only
                p_i[t - first, l] = mean + np.sqrt(ch.sigma[t]) * L.orc_rnorm(C.byref(r), UPD_SAMPLE_PATH, int(model.off[i]) + l)
        paths.append(p_i)
        mo.append(m_i)
    return paths, mo


@pytest.mark.parametrize("with_x,cov_adj", [(False, False), (True, False), (True, True)])
@pytest.mark.parametrize("simultaneous", [False, True])
def test_sample_paths_on_the_shipped_chain(with_x, cov_adj, simultaneous):
    from bayesfmmm_amd import api
    t = [np.asarray(v).reshape(-1)[::4] for v in read_rds(os.path.join(GOLD, "time.RDS"))]         # 25 points per curve
    t[3] = t[3][:7]                                                                                  # a ragged one
    e = dict(y=[np.zeros(len(v)) for v in t], t=t, boundary_knots=BK, internal_knots=IK, n=40)
    X = np.random.default_rng(11).standard_normal((40, 1)) if with_x else None
    model, ch, B = _oracle_chain(e, X, TRACE, 1, cov_adj)
    burn, alpha, seed = 0.6, 0.1, 5
    kept = R.kept_count(150, burn)
    got = api.FSamplePaths(TRACE, 1, 3, BK, IK, t, alpha=alpha, burnin_prop=burn, simultaneous=simultaneous, X=X, cov_adj=cov_adj,
                           seed=seed)
    paths, mo = _oracle_paths(model, ch, B, X, 150 - kept, seed)
    assert len(got["Path_trace"]) == 40
    for i in range(40):
        assert got["Path_trace"][i].shape == (kept, len(t[i]))
        np.testing.assert_allclose(got["Mean_only_Path_trace"][i], mo[i], rtol=1e-11, atol=1e-12, err_msg=f"mean only, curve {i}")
        np.testing.assert_allclose(got["Path_trace"][i], paths[i], rtol=1e-10, atol=1e-11, err_msg=f"paths, curve {i}")
        up, md, lo = R.bands(paths[i], alpha, simultaneous)
        for nm, ref in (("CI_Upper", up), ("CI_50", md), ("CI_Lower", lo)):
            np.testing.assert_allclose(got[nm][i], ref, rtol=1e-9, atol=1e-10, err_msg=f"{nm}, curve {i}")
    # the noise has the draw's variance: standardised residuals of the paths are N(0, 1)
    zs = np.concatenate([((got["Path_trace"][i] - np.array([[O.lib().orc_fitted(C.byref(model.data), C.byref(ch.c), 150 - kept + tt, i, l)
                                                                for l in range(len(t[i]))] for tt in range(kept)])) /
                          np.sqrt(ch.sigma[150 - kept:, None])).ravel() for i in (0, 17)])
    assert abs(zs.mean()) < 0.1 and abs(zs.std() - 1) < 0.1
    # a different seed gives different noise, the same means
    again = api.FSamplePaths(TRACE, 1, 3, BK, IK, t, alpha=alpha, burnin_prop=burn, simultaneous=simultaneous, X=X, cov_adj=cov_adj, seed=6)
    assert np.abs(again["Path_trace"][0] - got["Path_trace"][0]).max() > 1e-3
    np.testing.assert_array_equal(again["Mean_only_Path_trace"][0], got["Mean_only_Path_trace"][0])


def test_sample_paths_argument_checks():
    from bayesfmmm_amd import _lib, api
    t = [np.asarray(v).reshape(-1) for v in read_rds(os.path.join(GOLD, "time.RDS"))]
    with pytest.raises(_lib.BfmmmError, match="'alpha' must be between 0 and 1"):
        api.FSamplePaths(TRACE, 1, 3, BK, IK, t, alpha=1.0)
    with pytest.raises(_lib.BfmmmError, match="number of columns in 'X'"):
        api.FSamplePaths(TRACE, 1, 3, BK, IK, t, X=np.zeros((40, 2)))
    with pytest.raises(_lib.BfmmmError, match="number of functions"):
        api.FSamplePaths(TRACE, 1, 3, BK, IK, t[:39])
