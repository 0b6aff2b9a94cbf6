"""BASELINE.json config 5 / SURVEY.md 8(d): BFMMM_Nu_Z_multiple_try with n_try = 7 -- eight independent chains of the reduced
sweep (Z, pi, alpha_3, nu, tau, sigma^2, log-likelihood; Phi = chi = 0, BFMMM.h:1073-1113) -- on the config-2 data
(n_funct = 4096 curves of 100 points, K = 3, P = 30), through the entry point of include/bfmmm_entry.h.

The entry point runs the eight chains as ONE sampler batch (chain index = grid dimension); the reference runs them one
after the other and keeps the chain with the largest mean log-likelihood over its last 99 draws
(src/UserFunctions.cpp:302-325).  Checked at full size through size-independent properties: every chain run on its own
(chain_offset / chain_stride) reproduces the batched run bit for bit, Z rows lie on the simplex, the log-likelihood is
recomputed on the host from the saved draws, the winner is the argmax of the 99-value tail means with the earlier chain on
ties, and the multi-GPU code path (device list + RCCL gather, bfmmm_gather_best) returns the same result."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

T = 110


@pytest.fixture(scope="module")
def w():
    from bench import make_config2
    return make_config2()


def _common(w):
    return (w["y"], w["t"], w["n"], w["degree"], w["M"], w["boundary_knots"], w["internal_knots"])


def host_loglik(w, nu, Z, s2):
    B = w["B"][0]
    Y = np.stack(w["y"])
    rss = ((Y - (Z @ nu) @ B.T) ** 2).sum()
    return -Y.size * (0.9189385332046727 + 0.5 * np.log(s2)) - rss / (2 * s2)


@pytest.fixture(scope="module")
def batched(w):
    from bayesfmmm_amd import api
    return api.BFMMM_Nu_Z_multiple_try(T, 7, w["K"], *_common(w), seed=11)


def test_eight_chains_one_by_one_equal_the_batched_run(w, batched):
    from bayesfmmm_amd import api
    scores = []
    for c in range(8):
        r = api.BFMMM_Nu_Z_multiple_try(T, 7, w["K"], *_common(w), seed=11, chain_offset=c, chain_stride=100)
        assert r["best_chain"] == c
        Z, nu, s2, ll = r["Z"], r["nu"], r["sigma_sq"], r["loglik"]
        assert Z.shape == (4096, 3, T) and nu.shape == (3, 30, T)
        assert np.abs(Z.sum(axis=1) - 1).max() < 1e-12 and (Z > 0).all()       # rows on the simplex
        assert np.isfinite(ll).all() and (s2 > 0).all()
        for t in (0, T // 2, T - 1):                                             # log-likelihood recomputed on the host
            ref = host_loglik(w, nu[:, :, t], Z[:, :, t], s2[t])
            assert abs(ll[t] - ref) < 1e-8 * abs(ref), (c, t, ll[t], ref)
        assert abs(r["best_score"] - ll[T - 99:].mean()) < 1e-9 * abs(r["best_score"])
        scores.append(r["best_score"])
        if c == int(batched["best_chain"]):
            for nm in ("nu", "Z", "pi", "alpha_3", "tau", "sigma_sq", "loglik", "A", "delta"):
                np.testing.assert_array_equal(batched[nm], r[nm], err_msg=nm)
    # winner = argmax of the tail means, the earlier chain on ties (UserFunctions.cpp:320)
    assert int(batched["best_chain"]) == int(np.argmax(scores))
    assert batched["best_score"] == max(scores)
    assert len(set(scores)) == 8                                                 # eight different chains
    # the chains moved towards the data: the winner's late log-likelihood beats its first draws
    assert batched["loglik"][-20:].mean() > batched["loglik"][:3].mean()


def test_smaller_batches_give_the_same_winner(w, batched):
    from bayesfmmm_amd import api
    r = api.BFMMM_Nu_Z_multiple_try(T, 7, w["K"], *_common(w), seed=11, max_concurrent=3)      # batches of 3, 3, 2 chains
    assert r["best_chain"] == batched["best_chain"] and r["best_score"] == batched["best_score"]
    np.testing.assert_array_equal(r["Z"], batched["Z"])


def test_device_list_and_rccl_gather_give_the_same_result(w, batched):
    """devices = [0]: the multi-GPU path (one host thread + one sampler batch per device, RCCL all-gather of the scores,
    winner's chain sent to devices[0]) with a single rank; with two or more GPUs the chains are dealt over them and the
    result must still be the batched one, bit for bit."""
    import torch
    from bayesfmmm_amd import api
    r = api.BFMMM_Nu_Z_multiple_try(T, 7, w["K"], *_common(w), seed=11, devices=[0])
    assert r["best_chain"] == batched["best_chain"] and r["best_score"] == batched["best_score"]
    for nm in ("nu", "Z", "pi", "alpha_3", "tau", "sigma_sq", "loglik"):
        np.testing.assert_array_equal(r[nm], batched[nm], err_msg=nm)
    if torch.cuda.device_count() >= 2:
        r2 = api.BFMMM_Nu_Z_multiple_try(T, 7, w["K"], *_common(w), seed=11, devices=[0, 1])
        assert r2["best_chain"] == batched["best_chain"] and r2["best_score"] == batched["best_score"]
        for nm in ("nu", "Z", "pi", "alpha_3", "tau", "sigma_sq", "loglik"):
            np.testing.assert_array_equal(r2[nm], batched[nm], err_msg=nm)


def test_gather_best_picks_largest_score_and_lowest_chain_on_ties():
    """bfmmm_gather_best on one rank: NaN scores never win, the call reports rank 0."""
    import ctypes as C
    import bayesfmmm_amd as bf
    from bayesfmmm_amd import _lib
    from simdata import simulate_functional
    sim = simulate_functional(n=24, M=2, sigma_sq=0.01, seed=1)
    cfg = bf.default_config(model=bf.MODEL_FUNCTIONAL, K=sim["K"], n_eigen=sim["M"], basis_degree=3, tot_mcmc_iters=4)
    s = bf.Sampler(cfg, sim["y"], sim["t"], sim["internal_knots"], sim["boundary_knots"], n_chains=2)
    lib = _lib.load()
    hs = (C.c_void_p * 1)(s.h)
    sc = (C.c_double * 1)(-3.5)
    ids = (C.c_int32 * 1)(4)
    win = C.c_int(-1)
    _lib.check(lib.bfmmm_gather_best(hs, 1, sc, ids, C.byref(win)))
    assert win.value == 0
    sc[0] = float("nan")
    with pytest.raises(_lib.BfmmmError, match="no rank holds a valid chain"):
        _lib.check(lib.bfmmm_gather_best(hs, 1, sc, ids, C.byref(win)))
    s.close()
