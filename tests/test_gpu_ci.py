"""Credible intervals over saved draws on the device (include/bfmmm_post.h): SigmaCI, ZCI, FMeanCI
(src/PostProcessing.cpp:3435, :3505, :99) against the numpy restatement oracle/post_ci.py -- on the trace the package ships
(inst/test-data/Functional_trace: the documented examples' directory) and on batches written by this library's own
warm-start run of the package's K = 2 example.  (The rescaled paths on the shipped trace's own Z0.txt / Chi0.txt:
tests/test_gpu_reference_functional_trace.py.)"""
import os
import sys

import numpy as np
import pytest

from rds_reader import read_rds

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import post_ci as R      # noqa: E402

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
TRACE = os.path.join(GOLD, "Functional_trace") + "/"
BK, IK = [0.0, 1000.0], [250.0, 500.0, 750.0]


def _basis(time):
    from bayesfmmm_amd import api
    return np.ascontiguousarray(api.TensorBSpline(np.asarray(time, dtype=np.float64).reshape(-1, 1), [3], [BK], [IK]))


@pytest.mark.parametrize("simultaneous", [False, True])
@pytest.mark.parametrize("with_x", [False, True])
def test_fmeanci_on_the_reference_trace(simultaneous, with_x):
    from bayesfmmm_amd import api
    time = np.arange(0.0, 1000.0, 10.0)                       # seq(0, 990, 10)
    X = np.arange(-2.0, 2.0001, 0.2).reshape(-1, 1) if with_x else None
    nu = api.ReadCube(TRACE + "Nu0.txt")
    eta = None
    if with_x:
        f = api.ReadFieldCube(TRACE + "Eta0.txt")
        eta = np.stack([f[l, 0] for l in range(nu.shape[2])], axis=-1)
    for k, alpha, burn in ((2, 0.05, 0.1), (1, 0.2, 0.5)):
        got = api.FMeanCI(TRACE, 1, time, 3, BK, IK, k, alpha=alpha, rescale=False, simultaneous=simultaneous, burnin_prop=burn, X=X)
        ref = R.f_mean_ci(nu, _basis(time), k, alpha, False, simultaneous, burn, X=X, eta=eta)
        for nm in ("CI_Upper", "CI_50", "CI_Lower", "mean_trace"):
            assert got[nm].shape == np.asarray(ref[nm]).shape, (nm, got[nm].shape, np.asarray(ref[nm]).shape)
            np.testing.assert_allclose(got[nm], ref[nm], rtol=1e-11, atol=1e-12, err_msg=nm)


def test_sigmaci_on_the_reference_trace():
    from bayesfmmm_amd import api
    sig = api.ReadVec(TRACE + "Sigma0.txt")
    for alpha, burn in ((0.05, 0.1), (0.3, 0.0), (0.01, 0.73)):
        got = api.SigmaCI(TRACE, 1, alpha=alpha, burnin_prop=burn)
        ref = R.sigma_ci(sig, alpha, burn)
        for nm in ("CI_Upper", "CI_50", "CI_Lower"):
            assert got[nm] == pytest.approx(ref[nm], rel=1e-14), nm
    assert got["CI_Lower"] == got["CI_50"]                    # the reference's CI_Lower is its median (PostProcessing.cpp:3496)


@pytest.fixture(scope="module")
def k2_batches(tmp_path_factory):
    from bayesfmmm_amd import api
    Y = [np.asarray(v).reshape(-1) for v in read_rds(os.path.join(GOLD, "Sim_data.RDS"))]
    t = [np.asarray(v).reshape(-1) for v in read_rds(os.path.join(GOLD, "time.RDS"))]
    X = np.random.default_rng(5).standard_normal((40, 1))
    common = (2, Y, t, 40, 3, 3, BK, IK)
    T = 150
    est1 = api.BFMMM_Nu_Z_multiple_try(T, 1, *common, X=X, seed=1)
    est2 = api.BFMMM_Theta_est(T, 1, *common, est1, X=X, seed=2)
    d = tmp_path_factory.mktemp("k2trace")
    api.BFMMM_warm_start(T, *common, est1, est2, X=X, seed=3, dir=str(d) + "/", r_stored_iters=50, thinning_num=1)
    return str(d) + "/", 3


def _cat(dirn, n_files, name, rd):
    return np.concatenate([rd(f"{dirn}{name}{q}.txt") for q in range(n_files)], axis=-1)


@pytest.mark.parametrize("rescale", [True, False])
def test_zci_matches_oracle(k2_batches, rescale):
    from bayesfmmm_amd import api
    dirn, n_files = k2_batches
    Z = _cat(dirn, n_files, "Z", api.ReadCube)
    for alpha, burn in ((0.05, 0.1), (0.2, 0.4)):
        got = api.ZCI(dirn, n_files, alpha=alpha, rescale=rescale, burnin_prop=burn)
        ref = R.z_ci(Z, alpha, rescale, burn)
        for nm in ("CI_Upper", "CI_50", "CI_Lower", "Z_trace"):
            assert got[nm].shape == ref[nm].shape, nm
            np.testing.assert_allclose(got[nm], ref[nm], rtol=1e-10, atol=1e-12, err_msg=nm)


@pytest.mark.parametrize("mode", ["rescale", "trans_mats"])
@pytest.mark.parametrize("with_x", [False, True])
def test_fmeanci_rescaled(k2_batches, mode, with_x):
    from bayesfmmm_amd import api
    dirn, n_files = k2_batches
    nu, Z = _cat(dirn, n_files, "Nu", api.ReadCube), _cat(dirn, n_files, "Z", api.ReadCube)
    T = nu.shape[2]
    eta = np.zeros((7, 1, 2, T))
    for q in range(n_files):
        f = api.ReadFieldCube(f"{dirn}Eta{q}.txt")
        for l in range(T // n_files):
            eta[..., q * (T // n_files) + l] = f[l, 0]
    time = np.linspace(5.0, 995.0, 37)
    X = np.array([[-1.0], [0.3], [1.7]]) if with_x else None
    burn = 0.2
    kept = R.kept_count(T, burn)
    tm = None
    if mode == "trans_mats":
        rng = np.random.default_rng(2)
        tm = np.concatenate([np.eye(2) + 0.1 * rng.standard_normal((2, 2)) for _ in range(kept)], axis=0)
    for simultaneous in (False, True):
        got = api.FMeanCI(dirn, n_files, time, 3, BK, IK, 1, rescale=(mode == "rescale"), simultaneous=simultaneous,
                          burnin_prop=burn, X=X, trans_mats=tm)
        ref = R.f_mean_ci(nu, _basis(time), 1, 0.05, mode == "rescale", simultaneous, burn, Z=Z, X=X,
                          eta=eta if with_x else None, trans_mats=tm)
        for nm in ("CI_Upper", "CI_50", "CI_Lower", "mean_trace"):
            np.testing.assert_allclose(got[nm], ref[nm], rtol=1e-10, atol=1e-12, err_msg=nm)


@pytest.mark.parametrize("simultaneous", [False, True])
def test_fcovci_on_the_reference_trace(simultaneous):
    """FCovCI's documented example (R/RcppExports.R: time1 = time2 = seq(0, 990, 10), l = m = 1 on Functional_trace), rescale
    off here; the default rescale = TRUE on the shipped Z0.txt is in tests/test_gpu_reference_functional_trace.py."""
    from bayesfmmm_amd import api
    f = api.ReadFieldCube(TRACE + "Phi0.txt")
    Phi = np.stack([f[l, 0] for l in range(f.shape[0])], axis=-1)
    t1, t2 = np.arange(0.0, 1000.0, 40.0), np.arange(0.0, 1000.0, 30.0)
    for l, m, burn in ((1, 1, 0.1), (1, 2, 0.4)):
        got = api.FCovCI(TRACE, 1, t1, t2, 3, BK, IK, l, m, rescale=False, simultaneous=simultaneous, burnin_prop=burn)
        ref = R.f_cov_ci(Phi, _basis(t1), _basis(t2), l, m, 0.05, False, simultaneous, burn)
        for nm in ("CI_Upper", "CI_50", "CI_Lower", "cov_trace"):
            assert got[nm].shape == ref[nm].shape, nm
            np.testing.assert_allclose(got[nm], ref[nm], rtol=1e-10, atol=1e-13, err_msg=nm)


def test_fcovci_rescaled_and_transformed(k2_batches):
    from bayesfmmm_amd import api
    dirn, n_files = k2_batches
    Z = _cat(dirn, n_files, "Z", api.ReadCube)
    T = Z.shape[2]
    Phi = np.zeros((2, 7, 3, T))
    for q in range(n_files):
        f = api.ReadFieldCube(f"{dirn}Phi{q}.txt")
        for l in range(T // n_files):
            Phi[..., q * (T // n_files) + l] = f[l, 0]
    t1 = np.linspace(10.0, 990.0, 15)
    kept = R.kept_count(T, 0.3)
    rng = np.random.default_rng(9)
    tm = np.concatenate([np.eye(2) + 0.1 * rng.standard_normal((2, 2)) for _ in range(kept)], axis=0)
    for rescale, tmats, simultaneous in ((True, None, False), (True, tm, True), (False, tm, False)):
        got = api.FCovCI(dirn, n_files, t1, t1, 3, BK, IK, 2, 1, alpha=0.1, rescale=rescale, simultaneous=simultaneous, burnin_prop=0.3,
                         trans_mats=tmats)
        ref = R.f_cov_ci(Phi, _basis(t1), _basis(t1), 2, 1, 0.1, rescale, simultaneous, 0.3, Z=Z, trans_mats=tmats)
        for nm in ("CI_Upper", "CI_50", "CI_Lower", "cov_trace"):
            np.testing.assert_allclose(got[nm], ref[nm], rtol=1e-10, atol=1e-13, err_msg=nm)


@pytest.mark.parametrize("rescale", [True, False])
@pytest.mark.parametrize("with_x", [False, True])
def test_mvmeanci_on_the_reference_trace(rescale, with_x):
    """MVMeanCI's documented examples on the trace the package ships (inst/test-data/Multivariate_trace: K = 2, so the
    default rescale path runs on the reference's own Z draws)."""
    from bayesfmmm_amd import api
    dirn = os.path.join(GOLD, "Multivariate_trace") + "/"
    nu, Z = api.ReadCube(dirn + "Nu0.txt"), api.ReadCube(dirn + "Z0.txt")
    X = np.array([[-1.0], [0.0], [2.5]]) if with_x else None
    eta = None
    if with_x:
        f = api.ReadFieldCube(dirn + "Eta0.txt")
        eta = np.stack([f[l, 0] for l in range(nu.shape[2])], axis=-1)
    for alpha, burn in ((0.05, 0.1), (0.3, 0.5)):
        got = api.MVMeanCI(dirn, 1, alpha=alpha, rescale=rescale, burnin_prop=burn, X=X)
        ref = R.mv_mean_ci(nu, alpha, rescale, burn, Z=Z, X=X, eta=eta)
        for nm in ("CI_Upper", "CI_50", "CI_Lower", "mean_trace"):
            assert got[nm].shape == ref[nm].shape, (nm, got[nm].shape, ref[nm].shape)
            np.testing.assert_allclose(got[nm], ref[nm], rtol=1e-11, atol=1e-13, err_msg=nm)


@pytest.mark.parametrize("rescale,simultaneous,with_x", [(True, False, False), (True, True, True), (False, False, True)])
def test_hdfmeanci_on_the_reference_trace(rescale, simultaneous, with_x):
    """HDFMeanCI's documented example (R/RcppExports.R:147-159: HDFunctional_trace, time = HDtime.RDS[[1]], quadratic splines with
    knots 250/500/750 on (0, 990) in both dimensions, K = 2) on the trace the package ships, default rescale path included."""
    from bayesfmmm_amd import api
    dirn = os.path.join(GOLD, "HDFunctional_trace") + "/"
    time = np.asarray(read_rds(os.path.join(GOLD, "HDtime.RDS"))[0], dtype=np.float64)
    degs, bks, iks = [2, 2], [[0.0, 990.0], [0.0, 990.0]], [[250.0, 500.0, 750.0]] * 2
    nu, Z = api.ReadCube(dirn + "Nu0.txt"), api.ReadCube(dirn + "Z0.txt")
    B = np.ascontiguousarray(api.TensorBSpline(time, degs, bks, iks))
    X = np.array([[-0.5], [1.0]]) if with_x else None
    eta = None
    if with_x:
        f = api.ReadFieldCube(dirn + "Eta0.txt")
        eta = np.stack([f[l, 0] for l in range(nu.shape[2])], axis=-1)
    got = api.HDFMeanCI(dirn, 1, time, degs, bks, iks, 2, rescale=rescale, simultaneous=simultaneous, burnin_prop=0.2, X=X)
    ref = R.f_mean_ci(nu, B, 2, 0.05, rescale, simultaneous, 0.2, Z=Z, X=X, eta=eta)
    for nm in ("CI_Upper", "CI_50", "CI_Lower", "mean_trace"):
        assert got[nm].shape == np.asarray(ref[nm]).shape, nm
        np.testing.assert_allclose(got[nm], ref[nm], rtol=1e-10, atol=1e-12, err_msg=nm)


def test_ci_argument_checks_and_quantile_edges():
    from bayesfmmm_amd import _lib, api
    time = np.arange(0.0, 1000.0, 10.0)
    with pytest.raises(_lib.BfmmmError, match="'alpha' must be between 0 and 1"):
        api.SigmaCI(TRACE, 1, alpha=1.0)
    with pytest.raises(_lib.BfmmmError, match="'n_files' must be greater than 0"):
        api.FMeanCI(TRACE, 0, time, 3, BK, IK, 1)
    with pytest.raises(_lib.BfmmmError, match="'k' must be less than or equal to the number of clusters in the model"):
        api.FMeanCI(TRACE, 1, time, 3, BK, IK, 3, rescale=False)
    with pytest.raises(_lib.BfmmmError, match="'burnin_prop' must be between 0 and 1"):
        api.FMeanCI(TRACE, 1, time, 3, BK, IK, 1, burnin_prop=-0.1)
    # the quantile primitive: ties, extremes beyond (N - 0.5) / N, a non-power-of-two and a one-element column
    lib = api._lib_entry()
    rng = np.random.default_rng(0)
    # (8192 is the longest column sorted in LDS; 8193, 20000 and 40001 take the global-memory network: two, three and four
    #  chunks / global exchange distances)
    for T in (1, 2, 7, 150, 1000, 4097, 8192, 8193, 20000, 40001):
        V = np.asfortranarray(np.round(rng.standard_normal((T, 5)), 1))
        probs = np.array([0.0, 0.004, 0.025, 0.5, 0.75, 0.999, 1.0])
        out = np.zeros((len(probs), 5), order="F")
        assert lib.bfmmm_post_col_quantiles(V.ctypes.data_as(api.c_double_p), T, 5, probs.ctypes.data_as(api.c_double_p), len(probs), 0,
                                            out.ctypes.data_as(api.c_double_p)) == 0
        for c in range(5):
            np.testing.assert_allclose(out[:, c], R.arma_quantile(V[:, c], probs), rtol=1e-15)


def test_bands_over_more_than_8192_draws():
    """the reference has no limit on the number of kept draws (arma::quantile sorts any length): pointwise and simultaneous bands
    of a 12000-draw table against the numpy oracle"""
    from bayesfmmm_amd import api
    lib = api._lib_entry()
    rng = np.random.default_rng(5)
    T, ncol = 12000, 7
    V = np.asfortranarray(rng.standard_normal((T, ncol)) * np.arange(1, ncol + 1) + np.arange(ncol))
    for simultaneous in (0, 1):
        up, mid, lo = (np.zeros(ncol) for _ in range(3))
        assert lib.bfmmm_post_table_bands(V.ctypes.data_as(api.c_double_p), T, ncol, 0.05, simultaneous, 0, up.ctypes.data_as(api.c_double_p),
                                          mid.ctypes.data_as(api.c_double_p), lo.ctypes.data_as(api.c_double_p)) == 0
        if simultaneous:
            m, sd = V.mean(axis=0), V.std(axis=0, ddof=1)
            qc = R.arma_quantile(np.abs((V - m) / sd).max(axis=1), np.array([0.95]))[0]
            ref_up, ref_mid, ref_lo = m + qc * sd, m, m - qc * sd
        else:
            q = np.stack([R.arma_quantile(V[:, c], np.array([0.025, 0.5, 0.975])) for c in range(ncol)])
            ref_lo, ref_mid, ref_up = q[:, 0], q[:, 1], q[:, 2]
        np.testing.assert_allclose(up, ref_up, rtol=1e-12)
        np.testing.assert_allclose(mid, ref_mid, rtol=1e-12)
        np.testing.assert_allclose(lo, ref_lo, rtol=1e-12)
