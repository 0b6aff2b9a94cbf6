#!/usr/bin/env python3
"""Generates tests/golden/oracle_warm_sweep.npz: inputs and the expected chains of a small warm-start run (n = 8 ragged
curves, K = 2, P = 8, M = 3, T = 5; with two covariates, mean and covariance adjusted) from the CPU oracle (oracle/, the
line-faithful restatement of the reference) under the keyed generator with seed 17.  The reference itself cannot be run in
this pipeline, so these vectors pin the ORACLE's behaviour (tests/test_oracle_fixture.py, CPU) and give the GPU
parity tests committed numbers to meet (tests/test_gpu_parity.py::test_committed_oracle_fixture).

    python tests/golden/make_oracle_fixtures.py        # rewrites the .npz next to this script
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O                                   # noqa: E402
from simdata import simulate_functional, truth_chain     # noqa: E402

NAMES = ["nu", "Phi", "chi", "Z", "pi", "alpha3", "delta", "A", "gamma", "tau", "sigma", "loglik",
         "eta", "xi", "tau_eta", "gamma_xi", "delta_xi", "A_xi"]


def build(T=5, seed=17):
    sim = simulate_functional(n=8, M=3, sigma_sq=0.01, seed=4, K=2, D=2, ragged=True)
    model, ch = truth_chain(sim, T)
    rng = np.random.default_rng(21)
    K, P, M, n, D = sim["K"], sim["P"], sim["M"], sim["n"], 2
    ch.nu[:, :, 0] += 0.2 * rng.standard_normal((K, P))
    ch.Phi[..., 0] += 0.05 * rng.standard_normal((K, P, M))
    ch.chi[:, :, 0] += 0.2 * rng.standard_normal((n, M))
    ch.Z[:, :, 0] = rng.dirichlet(np.full(K, 2.0), size=n)
    ch.eta[..., 0] += 0.1 * rng.standard_normal((P, D, K))
    ch.xi[..., 0] += 0.02 * rng.standard_normal((P, D, M, K))
    init = {nm: np.array(getattr(ch, nm)[..., 0] if nm != "tau" else ch.tau[0, :], copy=True)
            for nm in NAMES if nm not in ("loglik", "sigma", "alpha3")}
    init["sigma"], init["alpha3"] = np.array([ch.sigma[0]]), np.array([ch.alpha3[0]])
    h = O.make_hyper(K)
    O.run_sweeps(model, h, ch, O.SWEEP_WARM, seed=seed, covariance_adj=True)
    out = {"y": np.concatenate(sim["y"]), "t": np.concatenate(sim["t"]), "offsets": model.off.copy(), "X": np.asarray(sim["X"]),
           "internal_knots": np.asarray(sim["internal_knots"], dtype=np.float64), "boundary_knots": np.asarray(sim["boundary_knots"], dtype=np.float64),
           "dims": np.array([n, K, P, M, D, T, seed])}
    for nm, v in init.items():
        out["init_" + nm] = v
    for nm in NAMES:
        out["chain_" + nm] = np.array(getattr(ch, nm), copy=True)
    return sim, model, ch, out


if __name__ == "__main__":
    _, _, _, out = build()
    np.savez_compressed(os.path.join(HERE, "oracle_warm_sweep.npz"), **out)
    print("wrote", os.path.join(HERE, "oracle_warm_sweep.npz"), {k: v.shape for k, v in out.items() if k.startswith("chain_")})
