"""On-disk chain batches of BFMMM_warm_start (dir / r_stored_iters / thinning_num; BFMMM.h:1680-1746, UserFunctions.cpp:
1508-1541): a batched run draws exactly what the in-memory run draws (the RNG is keyed by the iteration, not by the
slot), so every saved file is checked against the in-memory chain of the same seed -- draw 0 of batch q is iteration
q*r, draw p > 0 is iteration q*r + thinning*p - 1 -- including the reference's quirks (alpha_3's first entry, the
r-entry containers of the covariate blocks).  The files are read back with the ReadVec / ReadMat / ReadCube /
ReadFieldCube counterparts."""
import os

import numpy as np
import pytest

from rds_reader import read_rds

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def stages():
    from bayesfmmm_amd import api
    Y = [np.asarray(v).reshape(-1) for v in read_rds(os.path.join(GOLD, "Sim_data.RDS"))]
    t = [np.asarray(v).reshape(-1) for v in read_rds(os.path.join(GOLD, "time.RDS"))]
    common = (2, Y, t, 40, 3, 3, [0.0, 1000.0], [250.0, 500.0, 750.0])
    T = 150
    est1 = api.BFMMM_Nu_Z_multiple_try(T, 1, *common, seed=3)
    est2 = api.BFMMM_Theta_est(T, 1, *common, est1, seed=4)
    return dict(common=common, est1=est1, est2=est2, T=T)


def _iters(q, r, thin, cnt):
    return [q * r + (0 if p == 0 else thin * p - 1) for p in range(cnt)]


@pytest.mark.parametrize("r,thin,tt", [(50, 1, 0), (40, 4, 0), (50, 5, 30)])
def test_batches_match_the_in_memory_chain(stages, tmp_path, r, thin, tt):
    from bayesfmmm_amd import api
    s = stages
    T = s["T"]
    kw = dict(seed=5)
    if tt:
        kw.update(n_temp_trans=tt, N_t=2, beta_N_t=0.8)
    full = api.BFMMM_warm_start(T, *s["common"], s["est1"], s["est2"], **kw)
    d = str(tmp_path) + "/"
    bat = api.BFMMM_warm_start(T, *s["common"], s["est1"], s["est2"], dir=d, r_stored_iters=r, thinning_num=thin, **kw)
    n_batches = T // r
    cnt = r // thin
    names = ["Nu", "Chi", "Pi", "alpha_3", "A", "Delta", "Sigma", "Tau", "Gamma", "Phi", "Z"]
    assert sorted(os.listdir(d)) == sorted(f"{nm}{q}.txt" for nm in names for q in range(n_batches))
    for q in range(n_batches):
        it = _iters(q, r, thin, cnt)
        np.testing.assert_array_equal(api.ReadCube(f"{d}Nu{q}.txt"), full["nu"][:, :, it])
        np.testing.assert_array_equal(api.ReadCube(f"{d}Chi{q}.txt"), full["chi"][:, :, it])
        np.testing.assert_array_equal(api.ReadCube(f"{d}Z{q}.txt"), full["Z"][:, :, it])
        np.testing.assert_array_equal(api.ReadCube(f"{d}A{q}.txt"), full["A"][:, :, it])
        np.testing.assert_array_equal(api.ReadCube(f"{d}Delta{q}.txt"), full["delta"][:, :, it])
        np.testing.assert_array_equal(api.ReadMat(f"{d}Pi{q}.txt"), full["pi"][:, it])
        np.testing.assert_array_equal(api.ReadMat(f"{d}Tau{q}.txt"), full["tau"][it, :])
        np.testing.assert_array_equal(api.ReadVec(f"{d}Sigma{q}.txt"), full["sigma_sq"][it])
        a3 = api.ReadVec(f"{d}alpha_3{q}.txt")
        assert a3[0] == 0.0                                              # alpha_31(0) is never assigned
        np.testing.assert_array_equal(a3[1:], full["alpha_3"][it[1:]])
        phi, gam = api.ReadFieldCube(f"{d}Phi{q}.txt"), api.ReadFieldCube(f"{d}Gamma{q}.txt")
        assert phi.shape == (cnt, 1)
        for p, i in enumerate(it):
            np.testing.assert_array_equal(phi[p, 0], full["Phi"][..., i])
            np.testing.assert_array_equal(gam[p, 0], full["gamma"][..., i])
    # what stays in memory: the r slots of the last batch (slot 0 = the final state when the run ends on a full batch)
    assert bat["nu"].shape == (2, 7, r) and bat["loglik"].shape == (r,)
    last0 = (T // r) * r if T % r else T - r
    if T % r == 0:
        np.testing.assert_array_equal(bat["nu"][:, :, 1:], full["nu"][:, :, last0 + 1:T])
        np.testing.assert_array_equal(bat["nu"][:, :, 0], full["nu"][:, :, T - 1])
        np.testing.assert_array_equal(bat["loglik"], full["loglik"][last0:T])
    else:
        m = T - last0
        np.testing.assert_array_equal(bat["chi"][:, :, :m], full["chi"][:, :, last0:T])
        np.testing.assert_array_equal(bat["chi"][:, :, m:], full["chi"][:, :, last0 - r + m:last0])   # stale slots of the batch before
    if tt:
        assert bat["tt_blocks"] == full["tt_blocks"] and bat["tt_accepted"] == full["tt_accepted"]


def test_covariate_blocks_are_saved_in_r_entry_containers(stages, tmp_path):
    from bayesfmmm_amd import api
    s = stages
    T, r, thin = s["T"], 50, 5
    K, Y, t = s["common"][0], s["common"][1], s["common"][2]
    X = np.random.default_rng(1).standard_normal((40, 1))
    rest = s["common"][3:]
    est1 = api.BFMMM_Nu_Z_multiple_try(T, 1, K, Y, t, *rest, X=X, seed=3)
    est2 = api.BFMMM_Theta_est(T, 1, K, Y, t, *rest, est1, X=X, covariance_adj=True, seed=4)
    full = api.BFMMM_warm_start(T, K, Y, t, *rest, est1, est2, X=X, covariance_adj=True, seed=5)
    d = str(tmp_path) + "/"
    api.BFMMM_warm_start(T, K, Y, t, *rest, est1, est2, X=X, covariance_adj=True, seed=5, dir=d, r_stored_iters=r, thinning_num=thin)
    cnt = r // thin
    for q in range(T // r):
        it = _iters(q, r, thin, cnt)
        eta = api.ReadFieldCube(f"{d}Eta{q}.txt")
        xi = api.ReadFieldCube(f"{d}Xi{q}.txt")
        dxi = api.ReadFieldCube(f"{d}Delta_Xi{q}.txt")
        axi = api.ReadFieldCube(f"{d}A_Xi{q}.txt")
        gxi = api.ReadFieldCube(f"{d}Gamma_Xi{q}.txt")
        assert eta.shape == (r, 1) and xi.shape == (r, K) and gxi.shape == (r, K)          # BFMMM.h:5102-5107
        for p, i in enumerate(it):
            np.testing.assert_array_equal(eta[p, 0], full["eta"][..., i])
            np.testing.assert_array_equal(dxi[p, 0], full["delta_xi"][..., i])
            np.testing.assert_array_equal(axi[p, 0], full["A_xi"][..., i])
            for k in range(K):
                np.testing.assert_array_equal(xi[p, k], full["xi"][..., k, i])
                np.testing.assert_array_equal(gxi[p, k], full["gamma_xi"][..., k, i])
        assert all(eta[p, 0].size == 0 for p in range(cnt, r)) and xi[r - 1, K - 1].size == 0
        te = api.ReadCube(f"{d}Tau_Eta{q}.txt")
        assert te.shape == (K, 1, r)
        np.testing.assert_array_equal(te[:, :, :cnt], full["tau_eta"][:, :, it])
        assert np.all(te[:, :, cnt:] == 1.0)


def test_argument_checks(stages, tmp_path):
    from bayesfmmm_amd import _lib, api
    s = stages
    with pytest.raises(_lib.BfmmmError, match="'r_stored_iters' <= 'tot_mcmc_iters' with no 'dir' specified"):
        api.BFMMM_warm_start(s["T"], *s["common"], s["est1"], s["est2"], r_stored_iters=50)
    with pytest.raises(_lib.BfmmmError, match="'thinning_num' must be a positive integer"):
        api.BFMMM_warm_start(s["T"], *s["common"], s["est1"], s["est2"], thinning_num=0)
    with pytest.raises(_lib.BfmmmError, match="'r_stored_iters' must be a non-negative integer"):
        api.BFMMM_warm_start(s["T"], *s["common"], s["est1"], s["est2"], r_stored_iters=-1)
    # a directory with r_stored_iters = 0: everything stays in memory, nothing is written (UserFunctions.cpp:1510-1512)
    d = str(tmp_path) + "/"
    out = api.BFMMM_warm_start(s["T"], *s["common"], s["est1"], s["est2"], dir=d, seed=5)
    assert out["nu"].shape[2] == s["T"] + 1 and os.listdir(d) == []
