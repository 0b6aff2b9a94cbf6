#!/usr/bin/env python3
"""Times BASELINE.json configs[2] (config 2 + D = 5 covariates, mean and covariance adjustment: the 19-update
Mean_CovAdj sweep, BFMMM.h:4809-4894) on one MI355X.  Not the bench line (bench.py measures configs[1]); this is
the measurement behind the covariate rows of DESIGN.md and runs under rocprofv3 as well:

  python tools/bench_config3.py [--steps 200] [--warmup 20]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make_config3(seed=8, D=5):
    """config 2 data + covariates: eta_k ~ 0.3 N(0,1) P x D, xi_km ~ 0.1 (M-m)/M N(0,1) (after src/test-Eta.cpp:41-53)."""
    from bench import make_config2
    w = make_config2()
    rng = np.random.default_rng(seed)
    n, K, P, M = w["n"], w["K"], w["P"], w["M"]
    X = rng.standard_normal((n, D))
    eta = 0.3 * rng.standard_normal((P, D, K))
    xi = np.stack([0.1 * (M - m) / M * rng.standard_normal((P, D, K)) for m in range(M)], axis=2)   # P x D x M x K
    B = w["B"][0]
    st = w["state"]
    coef = np.zeros((n, P))
    for k in range(K):
        u = st["nu"][k][None, :] + X @ eta[:, :, k].T
        for m in range(M):
            u = u + st["chi"][:, m:m + 1] * (st["Phi"][k, :, m][None, :] + X @ xi[:, :, m, k].T)
        coef += st["Z"][:, k:k + 1] * u
    Y = coef @ B.T + 0.1 * rng.standard_normal((n, B.shape[0]))
    w = dict(w)
    w.update(y=[Y[i] for i in range(n)], Y=Y, X=X, eta=eta, xi=xi, D=D)
    return w


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-graph", action="store_true",
                    help="direct launches, synchronised per iteration (for rocprofv3: per-kernel durations)")
    a = ap.parse_args()
    import bayesfmmm_amd as bf
    S = bf.sampler
    w = make_config3()
    T = a.steps + a.warmup
    cfg = bf.default_config(model=bf.MODEL_FUNCTIONAL, K=w["K"], n_eigen=w["M"], basis_degree=3, tot_mcmc_iters=T)
    smp = bf.Sampler(cfg, w["y"], w["t"], w["internal_knots"], w["boundary_knots"])
    smp.set_covariates(w["X"], True)
    smp.set_state(**w["state"])
    smp.set_state(eta=w["eta"], xi=w["xi"])
    if a.no_graph:
        smp.set_profile(True)
    mask = S.SWEEP_WARM | S.COV_MEAN | S.COV_XI
    smp.run(mask, a.warmup, seed=2)
    t0 = time.perf_counter()
    smp.run(mask, a.steps, first_iter=a.warmup, seed=2)
    dt = (time.perf_counter() - t0) / a.steps
    if os.environ.get("COV_STAMPS"):
        st = smp.get_state("stamps")[40:46]
        print("k_cov_group phase clocks:", [int(st[i + 1] - st[i]) for i in range(5)], file=sys.stderr)
        s2 = smp.get_state("stamps")
        print("  partial phase: issue", int(s2[50] - s2[41]), "stage", int(s2[51] - s2[50]), "adds(wait)", int(s2[52] - s2[51]), "reduce", int(s2[42] - s2[52]), file=sys.stderr)
    n, P, M, K, D = w["n"], w["P"], w["M"], w["K"], w["D"]
    b_alg = 7 * n * 8 * (P * P + P + 1) + 8 * n * (2 * M + 2 * K) + 8 * n * D     # 7 blocks + X
    print(json.dumps({"workload": "config 3: n_funct=4096, D=5, K=3, P=30, M=6, Mean_CovAdj sweep (19 updates)",
                      "steps": a.steps, "ms_per_sweep": dt * 1e3, "iterations_per_s": 1.0 / dt,
                      "algorithmic_GBps": b_alg / dt / 1e9, "hbm_frac": b_alg / dt / 8e12,
                      "sigma_sq_last": float(smp.get_chain("sigma_sq")[T - 1])}))


if __name__ == "__main__":
    main()
