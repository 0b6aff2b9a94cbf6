"""Oracle parity AT THE FULL BENCHMARK SIZES (BASELINE.json configs 2-5): the HIP path against the reference-structure restatement
on the geometry the headline numbers are measured on -- n_funct = 4096 (25 k-slices x 10 column groups of k_pair_gram, k_pg_reduce's
8-at-a-time slab loads with the 24 + 1 tail, 512 curve workgroups over 8 XCDs, the `8 + b` placement of k_curve_chi), N = 8192 rows
of dimension 50 for the multivariate model, D = 5 covariates at n_funct = 4096, and the 8-chain Nu_Z batch.  The smaller-n tests
(tests/test_gpu_baseline_shapes.py) run the same kernel instantiations on 4 k-slices only.

The oracle runs with its row-window loops (oracle/updates.c::orc_set_row_window: bit-identical to the dense loops,
tests/test_oracle_row_window.py), which bring a full-size reference-structure sweep from 5-70 s down to 1-4 s.

Reference loops: BFMMM.h:1502-1553 (warm start), :4809-4894 (Mean_CovAdj), :2597-2650 (multivariate warm start), :1073-1113 (Nu_Z);
UpdateMixedMembership.h:131-185, UpdatePhi.h:23-89, UpdateNu.h:24-74, UpdateChi.h:19-64, UpdateSigma.h:22-58.
Tolerances as everywhere: 1e-8 single updates, 1e-6 short trajectories."""
import numpy as np
import pytest

import oracle_lib as O
from gpu_parity import ORC_FIELD, oracle_slot, push_state, rel_err
from test_gpu_baseline_shapes import COV_NAMES, WARM_NAMES, generic_state

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _row_window():
    O.set_row_window(True)
    yield
    O.set_row_window(False)


def config2_full(T, seed=1):
    import bayesfmmm_amd as bf
    from bench import make_config2
    w = make_config2(seed=seed)
    assert w["n"] == 4096 and w["P"] == 30 and w["K"] == 3 and w["M"] == 6 and w["n_i"] == 100
    sim = dict(n=w["n"], K=w["K"], M=w["M"], P=w["P"], nu=w["state"]["nu"], Phi=w["state"]["Phi"], chi=w["state"]["chi"])
    model = O.Model(w["y"], w["B"], w["K"], w["M"])
    cfg = bf.default_config(model=bf.MODEL_FUNCTIONAL, K=w["K"], n_eigen=w["M"], basis_degree=3, tot_mcmc_iters=T)
    return w, sim, model, cfg


def make_sampler(bf, cfg, w, **kw):
    return bf.Sampler(cfg, w["y"], w["t"], w["internal_knots"], w["boundary_knots"], **kw)


@pytest.mark.parametrize("which", ["Z", "Phi", "Nu", "Chi", "Sigma"])
def test_config2_full_size_single_updates(which):
    """every data-dependent block by itself at n_funct = 4096 from a generic (not the generating) state"""
    import bayesfmmm_amd as bf
    S = bf.sampler
    w, sim, model, cfg = config2_full(2)
    ch = O.Chain(model, 2)
    generic_state(ch, sim, np.random.default_rng(21))
    smp = make_sampler(bf, cfg, w)
    push_state(smp, ch)
    h = O.make_hyper(3)
    it, seed = 0, 78
    tilde_tau = np.cumprod(ch.delta[:, :, 0], axis=1)
    calls = {
        "Nu": (S.U_NU, lambda: O.updateNu(model, ch, it, seed=seed), "nu"),
        "Phi": (S.U_PHI, lambda: O.updatePhi(model, ch, it, tilde_tau, seed=seed), "Phi"),
        "Chi": (S.U_CHI, lambda: O.updateChi(model, ch, it, seed=seed), "chi"),
        "Z": (S.U_Z, lambda: O.updateZ_PM(model, ch, it, h.a_Z_PM, seed=seed), "Z"),
        "Sigma": (S.U_SIGMA, lambda: O.updateSigma(model, ch, it, h.alpha_0, h.beta_0, seed=seed), "sigma_sq"),
    }
    mask, orc_call, nm = calls[which]
    orc_call()
    smp.run(mask, 1, first_iter=0, seed=seed)
    got, ref = smp.get_state(nm).reshape(-1), oracle_slot(ch, nm, 0).reshape(-1)
    err = rel_err(got, ref)
    assert err < 1e-8, (which, err)
    if which == "Z":          # the accept / reject decisions themselves: no row may differ by a whole proposal
        assert np.abs(got - ref).max() < 1e-10
    smp.close()


def test_config2_full_size_warm_trajectory_reference_structure():
    """3 warm-start sweeps at n_funct = 4096 against the reference-structure loops (BFMMM.h:1502-1553), from a generic state"""
    import bayesfmmm_amd as bf
    T = 3
    w, sim, model, cfg = config2_full(T)
    ch = O.Chain(model, T)
    generic_state(ch, sim, np.random.default_rng(22))
    smp = make_sampler(bf, cfg, w)
    d = smp.dims()
    assert d["n"] == 4096 and d["A"] * d["P"] == 630
    push_state(smp, ch)
    O.run_sweeps(model, O.make_hyper(3), ch, O.SWEEP_WARM, seed=4)
    smp.run(bf.sampler.SWEEP_WARM, T, seed=4)
    for nm in WARM_NAMES:
        err = rel_err(smp.get_chain(nm), getattr(ch, ORC_FIELD.get(nm, nm)))
        assert err < 1e-6, (nm, err)
    smp.close()


def test_config2_full_size_warm_trajectory_gram_form_from_the_benchmark_state():
    """6 sweeps started exactly where bench.py starts (the generating values) against oracle/gram.c, which
    tests/test_oracle_gram.py ties to the reference-structure loops"""
    import bayesfmmm_amd as bf
    T = 6
    w, sim, model, cfg = config2_full(T)
    ch = O.Chain(model, T)
    names = {"alpha_3": "alpha3", "sigma_sq": "sigma"}
    for nm, v in w["state"].items():
        arr = getattr(ch, names.get(nm, nm))
        if nm == "tau":
            arr[0, :] = v
        elif arr.ndim == 1:
            arr[0] = np.asarray(v).reshape(-1)[0]
        else:
            arr[..., 0] = v
    smp = make_sampler(bf, cfg, w)
    smp.set_state(**w["state"])
    O.run_warm_gram(model, O.make_hyper(3), ch, seed=1)
    smp.run(bf.sampler.SWEEP_WARM, T, seed=1)
    for nm in WARM_NAMES:
        err = rel_err(smp.get_chain(nm), getattr(ch, ORC_FIELD.get(nm, nm)))
        assert err < 2e-6, (nm, err)
    smp.close()


def test_config4_full_size_multivariate_warm_trajectory():
    """BASELINE configs[3]: N = 8192 rows of dimension 50, K = 4, M = 8 (BFMMM.h:2597-2650): 2 sweeps against the oracle"""
    import bayesfmmm_amd as bf
    from test_gpu_multivariate import simulate_mv
    T, n, P, K, M = 2, 8192, 50, 4, 8
    sim = simulate_mv(n, P, K, M, 0.05, seed=414)
    model = O.Model([sim["Y"][i] for i in range(n)], [np.eye(P)] * n, K, M, mv=True)
    ch = O.Chain(model, T)
    generic_state(ch, sim, np.random.default_rng(415))
    ch.sigma[0] = 0.07
    cfg = bf.default_config(model=bf.MODEL_MULTIVARIATE, K=K, n_eigen=M, tot_mcmc_iters=T)
    smp = bf.Sampler(cfg, sim["Y"])
    push_state(smp, ch)
    assert smp.dims()["A"] == 36 and smp.dims()["n"] == 8192
    O.run_sweeps(model, O.make_hyper(K), ch, O.SWEEP_WARM, seed=5)
    smp.run(bf.sampler.SWEEP_WARM, T, seed=5)
    for nm in WARM_NAMES:
        err = rel_err(smp.get_chain(nm), getattr(ch, ORC_FIELD.get(nm, nm)))
        assert err < 1e-6, (nm, err)
    smp.close()


def test_config3_full_size_covariate_adjusted_trajectory():
    """BASELINE configs[2]: config 2 + D = 5 covariates with covariance adjustment, the 19-update Mean_CovAdj sweep
    (BFMMM.h:4809-4894), at n_funct = 4096: 2 sweeps against the oracle"""
    import bayesfmmm_amd as bf
    S = bf.sampler
    T, D = 2, 5
    w, sim, model0, cfg = config2_full(T)
    n, K, M, P = w["n"], w["K"], w["M"], w["P"]
    rng = np.random.default_rng(33)
    X = rng.standard_normal((n, D))
    eta = 0.5 * rng.standard_normal((P, D, K))
    xi = np.stack([0.1 * (M - m) / M * rng.standard_normal((P, D, K)) for m in range(M)], axis=2)      # P x D x M x K
    st = w["state"]
    coef = np.zeros((n, P))
    for k in range(K):
        u = X @ eta[:, :, k].T
        for m in range(M):
            u = u + st["chi"][:, m:m + 1] * (X @ xi[:, :, m, k].T)
        coef += st["Z"][:, k:k + 1] * u
    B = w["B"][0]
    ys = [w["y"][i] + B @ coef[i] for i in range(n)]
    model = O.Model(ys, w["B"], K, M, X=X)
    ch = O.Chain(model, T)
    generic_state(ch, sim, rng)
    ch.eta[..., 0] = eta + 0.1 * rng.standard_normal((P, D, K))
    ch.xi[..., 0] = xi + 0.05 * rng.standard_normal((P, D, M, K))
    ch.tau_eta[..., 0] = rng.gamma(3.0, 0.5, size=(K, D))
    ch.gamma_xi[..., 0] = rng.gamma(2.0, 0.7, size=(P, D, M, K))
    ch.delta_xi[..., 0] = rng.gamma(2.0, 1.0, size=(K, M, D))
    ch.A_xi[..., 0] = rng.gamma(2.0, 1.0, size=(K, 2, D))
    smp = bf.Sampler(cfg, ys, w["t"], w["internal_knots"], w["boundary_knots"])
    smp.set_covariates(X, True)
    push_state(smp, ch)
    smp.set_state(**{nm: oracle_slot(ch, nm, 0) for nm in COV_NAMES})
    O.run_sweeps(model, O.make_hyper(K), ch, O.SWEEP_WARM, seed=3, covariance_adj=True)
    smp.run(S.SWEEP_WARM | S.COV_MEAN | S.COV_XI, T, seed=3)
    for nm in WARM_NAMES + COV_NAMES:
        err = rel_err(smp.get_chain(nm), getattr(ch, ORC_FIELD.get(nm, nm)))
        assert err < 1e-6, (nm, err)
    smp.close()


def test_config5_full_size_nu_z_batch_of_eight_chains():
    """BASELINE configs[4]: the 8 chains of BFMMM_Nu_Z_multiple_try as ONE batch at n_funct = 4096, every chain against the
    reference-structure Nu_Z sweep run with that chain's RNG id (BFMMM.h:1073-1113): 3 sweeps; chains 0 and 7 also 6 sweeps
    against the sufficient-statistics form"""
    import bayesfmmm_amd as bf
    S = bf.sampler
    T, NCH = 6, 8
    w, sim, model, cfg = config2_full(T)
    batch = make_sampler(bf, cfg, w, n_chains=NCH)
    chains = []
    for q in range(NCH):
        ch = O.Chain(model, T)
        generic_state(ch, sim, np.random.default_rng(600 + q))
        ch.chi[:] = 0.0
        ch.Phi[:] = 0.0
        chains.append(ch)
        batch.select_chain(q)
        push_state(batch, ch)
    batch.run(S.SWEEP_NU_Z, T, seed=9, chain=0, phi_chi_zero=True)
    h = O.make_hyper(3)
    g = O.gram_prepare(model)
    try:
        for q in range(NCH):
            n_cmp = T if q in (0, NCH - 1) else 3
            if q in (0, NCH - 1):
                O.run_warm_gram(model, h, chains[q], n_iter=T, seed=9, chain_id=q, sweep=O.SWEEP_NU_Z, prepared=g)
            else:
                O.run_sweeps(model, h, chains[q], O.SWEEP_NU_Z, n_iter=n_cmp, seed=9, chain_id=q)
            batch.select_chain(q)
            for nm in ["nu", "Z", "pi", "alpha_3", "tau", "sigma_sq", "loglik"]:
                got = np.asarray(batch.get_chain(nm))
                ref = getattr(chains[q], ORC_FIELD.get(nm, nm))
                if nm == "tau":
                    got, ref = got[:n_cmp], ref[:n_cmp]
                else:
                    got, ref = got[..., :n_cmp], ref[..., :n_cmp]
                err = rel_err(got, ref)
                assert err < 2e-6, (q, nm, err)
    finally:
        O.gram_free(g)
    batch.close()


def test_config2_full_size_warm_batch_of_eight_chains_gram_form():
    """8 warm-start chains as one batch (bench.py `multi_chain`) at n_funct = 4096: the batched k_pair_gram (records staged once,
    chains walked inside the workgroup) and the two half-batches on two streams, every chain 3 sweeps against oracle/gram.c"""
    import bayesfmmm_amd as bf
    T, NCH = 3, 8
    w, sim, model, cfg = config2_full(T)
    batch = make_sampler(bf, cfg, w, n_chains=NCH)
    chains = []
    for q in range(NCH):
        ch = O.Chain(model, T)
        generic_state(ch, sim, np.random.default_rng(700 + q))
        chains.append(ch)
        batch.select_chain(q)
        push_state(batch, ch)
    batch.run(bf.sampler.SWEEP_WARM, T, seed=12, chain=0)
    h = O.make_hyper(3)
    g = O.gram_prepare(model)
    try:
        for q in range(NCH):
            O.run_warm_gram(model, h, chains[q], seed=12, chain_id=q, prepared=g)
            batch.select_chain(q)
            for nm in WARM_NAMES:
                err = rel_err(batch.get_chain(nm), getattr(chains[q], ORC_FIELD.get(nm, nm)))
                assert err < 1e-6, (q, nm, err)
    finally:
        O.gram_free(g)
    batch.close()


def test_long_curve_set_through_the_chunked_contraction():
    """n_funct = 32768 (beyond the cache-resident sizes: bfmmm_capi.hip routes the pair-Gram contraction of n > 16384 through
    k_pair_gram_pack -- about 128 k-slices of any length walked in 16-curve chunks with persistent accumulators): 3 warm-start
    sweeps against the sufficient-statistics form of the oracle (BFMMM.h:1502-1553), then a batch of four chains against the same
    chains run alone, bit for bit (the slices are chosen from n alone)"""
    import bayesfmmm_amd as bf
    from bench import make_config2
    T = 3
    w = make_config2(n=32768, n_i=24, seed=3)
    sim = dict(n=w["n"], K=w["K"], M=w["M"], P=w["P"], nu=w["state"]["nu"], Phi=w["state"]["Phi"], chi=w["state"]["chi"])
    model = O.Model(w["y"], w["B"], w["K"], w["M"])
    cfg = bf.default_config(model=bf.MODEL_FUNCTIONAL, K=w["K"], n_eigen=w["M"], basis_degree=3, tot_mcmc_iters=T)
    ch = O.Chain(model, T)
    generic_state(ch, sim, np.random.default_rng(41))
    smp = make_sampler(bf, cfg, w)
    push_state(smp, ch)
    O.run_warm_gram(model, O.make_hyper(3), ch, seed=6)
    smp.run(bf.sampler.SWEEP_WARM, T, seed=6)
    for nm in WARM_NAMES:
        err = rel_err(smp.get_chain(nm), getattr(ch, ORC_FIELD.get(nm, nm)))
        assert err < 1e-6, (nm, err)
    solo_nu, solo_chi = smp.get_chain("nu"), smp.get_chain("chi")
    smp.close()
    NCH = 4
    batch = make_sampler(bf, cfg, w, n_chains=NCH)
    chains = []
    for q in range(NCH):
        chq = O.Chain(model, T)
        generic_state(chq, sim, np.random.default_rng(41 + q))
        chains.append(chq)
        batch.select_chain(q)
        push_state(batch, chq)
    batch.run(bf.sampler.SWEEP_WARM, T, seed=6, chain=0)
    batch.select_chain(0)
    np.testing.assert_array_equal(batch.get_chain("nu"), solo_nu)
    np.testing.assert_array_equal(batch.get_chain("chi"), solo_chi)
    O.run_warm_gram(model, O.make_hyper(3), chains[3], seed=6, chain_id=3)
    batch.select_chain(3)
    for nm in WARM_NAMES:
        err = rel_err(batch.get_chain(nm), getattr(chains[3], ORC_FIELD.get(nm, nm)))
        assert err < 1e-6, (3, nm, err)
    batch.close()
