"""GPU parity tests: the HIP sampler (through the C ABI) against the CPU oracle on the same
inputs, state and RNG keys.  Tolerances: fp64 reassociation only (Gram-form sums on the GPU vs
per-observation loops in the oracle) -> 1e-8 relative on single updates, 1e-6 on short
trajectories (MCMC amplifies rounding differences), as stated in DESIGN.md."""
import numpy as np
import pytest

import oracle_lib as O
from gpu_parity import (gram_reference, make_sampler, oracle_slot, push_state, random_state, rel_err)
from simdata import simulate_functional, truth_chain

pytestmark = pytest.mark.gpu

TOL1 = 1e-8
TOLT = 1e-6


@pytest.fixture(scope="module")
def bf():
    import bayesfmmm_amd
    return bayesfmmm_amd


def setup(seed=1, n=37, M=3, T=8, sigma_sq=0.01, ragged=True):
    sim = simulate_functional(n=n, M=M, sigma_sq=sigma_sq, seed=seed, ragged=ragged)
    model, ch = truth_chain(sim, T)
    random_state(sim, ch, seed + 100)
    smp = make_sampler(sim, T)
    push_state(smp, ch)
    return sim, model, ch, smp


def test_basis_and_statistics(bf):
    sim, model, ch, smp = setup()
    B = smp.get_basis()
    for Bg, Bo in zip(B, sim["B"]):
        np.testing.assert_allclose(Bg, Bo, atol=1e-14)
    d = smp.dims()
    ref = gram_reference(sim, sim["Z"], sim["chi"], sim["M"] + 1)
    rec = smp.debug("rec").reshape(d["n"], d["LREC"])
    P, LG = d["P"], d["LG"]
    assert d["BW"] == 3 and LG == 4 * P
    np.testing.assert_allclose(rec[:, :LG].reshape(d["n"], 4, P), ref["band"], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(rec[:, LG:LG + P], ref["s"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(rec[:, LG + P], ref["yy"], rtol=1e-12)
    assert d["n_obs_total"] == sum(len(y) for y in sim["y"])
    assert d["half_sum"] == sum(len(y) // 2 for y in sim["y"])
    assert abs(d["YY"] - ref["yy"].sum()) <= 1e-10 * ref["yy"].sum()


def test_pair_gram_mfma(bf):
    # exercises the v_mfma_f64_16x16x4_f64 operand / accumulator lane maps with asymmetric data
    sim, model, ch, smp = setup(n=70)
    smp.run(bf.sampler.U_SIGMA, 1, seed=3)          # runs pair-Gram + reduce; sigma draw is irrelevant here
    d = smp.dims()
    K, MD, P, LG = d["K"], d["MD"], d["P"], d["LG"]
    ref = gram_reference(sim, ch.Z[:, :, 0], ch.chi[:, :, 0], MD)
    H = smp.debug("H").reshape(d["R"], 4, P)
    tv = smp.debug("tvec").reshape(d["A"], P)
    np.testing.assert_allclose(tv, ref["tvec"], rtol=1e-11, atol=1e-11)

    def tri(nn, a, b):
        a, b = min(a, b), max(a, b)
        return a * nn - a * (a - 1) // 2 + (b - a)
    ncc = MD * (MD + 1) // 2
    worst = 0.0
    for a in range(K * MD):
        for b in range(K * MD):
            ja, ma, jb, mb = a // MD, a % MD, b // MD, b % MD
            row = tri(K, ja, jb) * ncc + tri(MD, ma, mb)
            Hd = np.zeros((P, P))
            for dd in range(4):
                for p in range(P - dd):
                    Hd[p, p + dd] = Hd[p + dd, p] = H[row, dd, p]
            worst = max(worst, np.abs(Hd - ref["H"][(a, b)]).max() / np.abs(ref["H"][(a, b)]).max())
    assert worst < 1e-11


@pytest.mark.parametrize("which", ["Nu", "Phi", "Chi", "Z", "Sigma", "Tau", "Delta", "A", "Gamma", "Pi", "Alpha3"])
def test_single_update_matches_oracle(bf, which):
    S = bf.sampler
    sim, model, ch, smp = setup(seed=2)
    h = O.make_hyper(sim["K"])
    K, M = sim["K"], sim["M"]
    it, seed = 0, 77
    tilde_tau = np.cumprod(ch.delta[:, :, 0], axis=1)
    calls = {
        "Nu": (S.U_NU, lambda: O.updateNu(model, ch, it, seed=seed), ["nu"]),
        "Phi": (S.U_PHI, lambda: O.updatePhi(model, ch, it, tilde_tau, seed=seed), ["Phi"]),
        "Chi": (S.U_CHI, lambda: O.updateChi(model, ch, it, seed=seed), ["chi"]),
        "Z": (S.U_Z, lambda: O.updateZ_PM(model, ch, it, h.a_Z_PM, seed=seed), ["Z"]),
        "Sigma": (S.U_SIGMA, lambda: O.updateSigma(model, ch, it, h.alpha_0, h.beta_0, seed=seed), ["sigma_sq"]),
        "Tau": (S.U_TAU, lambda: O.updateTau(model, ch, it, h.alpha_nu, h.beta_nu, seed=seed), ["tau"]),
        "Delta": (S.U_DELTA, lambda: O.updateDelta(model, ch, it, seed=seed), ["delta"]),
        "A": (S.U_A, lambda: O.updateA(model, ch, it, h, seed=seed), ["A"]),
        "Gamma": (S.U_GAMMA, lambda: O.updateGamma(model, ch, it, h.nu_1, seed=seed), ["gamma"]),
        "Pi": (S.U_PI, lambda: O.updatePi_PM(model, ch, it, np.full(K, 10.0), h.a_pi_PM, seed=seed), ["pi"]),
        "Alpha3": (S.U_ALPHA3, lambda: O.updateAlpha3(model, ch, it, h.b, h.var_alpha3, seed=seed), ["alpha_3"]),
    }
    mask, orc_call, names = calls[which]
    before = {nm: oracle_slot(ch, nm, 0) for nm in names}
    orc_call()
    smp.run(mask, 1, first_iter=0, seed=seed)
    for nm in names:
        got = smp.get_state(nm).reshape(-1)
        ref = oracle_slot(ch, nm, 0).reshape(-1)
        assert rel_err(got, ref) < TOL1, (which, nm, rel_err(got, ref))
        if which not in ("Pi", "Alpha3", "A", "Z"):       # MH updates may legitimately reject
            assert rel_err(before[nm].reshape(-1), ref) > 1e-6


@pytest.mark.parametrize("sweep", ["nu_z", "theta", "warm"])
def test_short_trajectory_matches_oracle(bf, sweep):
    S = bf.sampler
    T = 6
    sim, model, ch, smp = setup(seed=5, T=T, n=41)
    h = O.make_hyper(sim["K"])
    if sweep == "nu_z":
        ch.chi[:] = 0.0
        ch.Phi[:] = 0.0
        push_state(smp, ch)
        O.run_sweeps(model, h, ch, O.SWEEP_NU_Z, seed=9)
        smp.run(S.SWEEP_NU_Z, T, seed=9, phi_chi_zero=True)
        names = ["nu", "Z", "pi", "alpha_3", "tau", "sigma_sq", "loglik"]
    elif sweep == "theta":
        O.run_sweeps(model, h, ch, O.SWEEP_THETA, seed=9)
        smp.run(S.SWEEP_THETA, T, seed=9)
        names = ["Phi", "chi", "delta", "A", "gamma", "tau", "sigma_sq", "loglik", "nu", "Z"]
    else:
        O.run_sweeps(model, h, ch, O.SWEEP_WARM, seed=9)
        smp.run(S.SWEEP_WARM, T, seed=9)
        names = ["nu", "Phi", "chi", "Z", "pi", "alpha_3", "delta", "A", "gamma", "tau", "sigma_sq", "loglik"]
    from gpu_parity import ORC_FIELD
    for nm in names:
        got = smp.get_chain(nm)
        ref = getattr(ch, ORC_FIELD.get(nm, nm))
        assert got.shape == ref.shape, nm
        assert rel_err(got, ref) < TOLT, (sweep, nm, rel_err(got, ref))


def test_reproducible_and_seed_sensitive(bf):
    S = bf.sampler
    sim, model, ch, smp = setup(seed=6, T=5)
    smp.run(S.SWEEP_WARM, 5, seed=4)
    a = smp.get_chain("nu")
    push_state(smp, ch)
    smp.run(S.SWEEP_WARM, 5, seed=4)
    b = smp.get_chain("nu")
    np.testing.assert_array_equal(a, b)           # bitwise reproducible (fixed-order reductions)
    push_state(smp, ch)
    smp.run(S.SWEEP_WARM, 5, seed=5)
    assert not np.allclose(a, smp.get_chain("nu"))


def test_create_error_messages(bf):
    sim = simulate_functional(n=5, M=2, sigma_sq=0.01, seed=1)
    with pytest.raises(bf._lib.BfmmmError, match="'K' must be an integer greater than or equal to 2"):
        cfg = bf.default_config(model=0, K=1, n_eigen=2, basis_degree=3, tot_mcmc_iters=10)
        bf.Sampler(cfg, sim["y"], sim["t"], sim["internal_knots"], sim["boundary_knots"])
    with pytest.raises(bf._lib.BfmmmError, match="less than or equal to first boundary knot"):
        cfg = bf.default_config(model=0, K=3, n_eigen=2, basis_degree=3, tot_mcmc_iters=10)
        bf.Sampler(cfg, sim["y"], sim["t"], [-5.0, 300.0], sim["boundary_knots"])


def test_committed_oracle_fixture():
    """The device against COMMITTED numbers: inputs, start state and expected chains of tests/golden/oracle_warm_sweep.npz
    (n = 8 ragged curves, K = 2, P = 8, M = 3, two covariates, mean and covariance adjusted, 5 warm-start iterations, seed 17)."""
    import os
    import bayesfmmm_amd as bf
    S = bf.sampler
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "oracle_warm_sweep.npz"))
    n, K, P, M, D, T, seed = (int(x) for x in g["dims"])
    off = g["offsets"]
    y = [g["y"][off[i]:off[i + 1]] for i in range(n)]
    t = [g["t"][off[i]:off[i + 1]] for i in range(n)]
    cfg = bf.default_config(model=bf.MODEL_FUNCTIONAL, K=K, n_eigen=M, basis_degree=3, tot_mcmc_iters=T)
    smp = bf.Sampler(cfg, y, t, g["internal_knots"], g["boundary_knots"])
    smp.set_covariates(g["X"], True)
    names = {"nu": "nu", "Phi": "Phi", "chi": "chi", "Z": "Z", "pi": "pi", "alpha_3": "alpha3", "delta": "delta", "A": "A",
             "gamma": "gamma", "tau": "tau", "sigma_sq": "sigma", "eta": "eta", "xi": "xi", "tau_eta": "tau_eta",
             "gamma_xi": "gamma_xi", "delta_xi": "delta_xi", "A_xi": "A_xi"}
    smp.set_state(**{k: g["init_" + v] for k, v in names.items()})
    smp.run(S.SWEEP_WARM | S.COV_MEAN | S.COV_XI, T, seed=seed)
    for k, v in dict(names, loglik="loglik").items():
        err = rel_err(smp.get_chain(k), g["chain_" + v])
        assert err < 2e-6, (k, err)
