"""oracle/updates.c's optional row-window mode (orc_set_row_window: loops over a basis row run over its non-zero window only) gives
BIT-IDENTICAL chains to the dense loops -- it is what makes the reference-structure oracle affordable at the full benchmark sizes
(tests/test_gpu_fullsize_oracle.py); bench.py's cpu_baseline times the dense loops."""
import numpy as np

import oracle_lib as O
from simdata import simulate_functional, truth_chain

NAMES = ["nu", "Phi", "chi", "Z", "pi", "alpha3", "delta", "A", "gamma", "tau", "sigma", "loglik"]


def _run(window, build):
    model, ch, kw = build()
    O.set_row_window(window)
    try:
        O.run_sweeps(model, O.make_hyper(model.K), ch, O.SWEEP_WARM, seed=3, **kw)
    finally:
        O.set_row_window(False)
    return ch


def _functional(cov):
    def build():
        sim = simulate_functional(n=17, M=2, sigma_sq=0.01, seed=8, ragged=True)
        rng = np.random.default_rng(2)
        X = rng.standard_normal((sim["n"], 2)) if cov else None
        model = O.Model(sim["y"], sim["B"], sim["K"], sim["M"], X=X)
        _, ch0 = truth_chain(sim, 3)
        ch = O.Chain(model, 3)
        for nm in NAMES[:-1]:
            getattr(ch, nm)[...] = getattr(ch0, nm)
        if cov:
            P, K, M = sim["P"], sim["K"], sim["M"]
            ch.eta[..., 0] = 0.3 * rng.standard_normal((P, 2, K))
            ch.xi[..., 0] = 0.1 * rng.standard_normal((P, 2, M, K))
            ch.tau_eta[..., 0] = 1.0
            ch.gamma_xi[..., 0] = 1.0
            ch.delta_xi[..., 0] = 1.0
            ch.A_xi[..., 0] = 1.0
        return model, ch, dict(covariance_adj=cov)
    return build


def _multivariate():
    rng = np.random.default_rng(5)
    n, P, K, M = 21, 7, 2, 2
    Y = rng.standard_normal((n, P))
    model = O.Model([Y[i] for i in range(n)], [np.eye(P)] * n, K, M, mv=True)
    ch = O.Chain(model, 3)
    ch.nu[:, :, 0] = rng.standard_normal((K, P))
    ch.Phi[..., 0] = 0.3 * rng.standard_normal((K, P, M))
    ch.chi[:, :, 0] = rng.standard_normal((n, M))
    ch.Z[:, :, 0] = rng.dirichlet(np.ones(K), size=n)
    ch.pi[:, 0] = 0.5
    ch.alpha3[0] = 5.0
    ch.delta[:, :, 0] = 1.0
    ch.A[:, :, 0] = 1.0
    ch.gamma[..., 0] = 1.0
    ch.tau[0, :] = 1.0
    ch.sigma[0] = 0.5
    return model, ch, {}


def test_row_window_mode_is_bit_identical():
    for build in (_functional(False), _functional(True), _multivariate):
        a = _run(False, build)
        b = _run(True, build)
        for nm in NAMES + (["eta", "xi", "tau_eta", "gamma_xi", "delta_xi", "A_xi"] if hasattr(a, "eta") else []):
            np.testing.assert_array_equal(getattr(a, nm), getattr(b, nm), err_msg=nm)
        assert np.isfinite(a.loglik).all() and a.loglik[0] != 0.0
