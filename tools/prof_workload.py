#!/usr/bin/env python3
"""One workload, graph replay, nothing else: the command rocprofv3 wraps for per-kernel statistics.

  python tools/prof_workload.py --workload warm|nu_z|config3|config4 [--chains C] [--steps N] [--warmup W]
Prints one JSON line (ms per step, chain-iterations/s)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="warm")
    ap.add_argument("--chains", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--n", type=int, default=4096)
    ap.add_argument("--M", type=int, default=6, help="eigenfunctions of the warm / nu_z workloads (config 2: 6)")
    ap.add_argument("--eager", action="store_true", help="event-bracketed eager launches instead of graph replay")
    a = ap.parse_args()
    import bayesfmmm_amd as bf
    from bench import make_config2
    S = bf.sampler
    T = a.steps + a.warmup
    pcz = False
    if a.workload in ("warm", "nu_z", "theta"):
        w = make_config2(n=a.n, n_i=(100 if a.n <= 4096 else 24), M=a.M)
        cfg = bf.default_config(model=bf.MODEL_FUNCTIONAL, K=w["K"], n_eigen=w["M"], basis_degree=3, tot_mcmc_iters=T)
        smp = bf.Sampler(cfg, w["y"], w["t"], w["internal_knots"], w["boundary_knots"], n_chains=a.chains)
        mask = {"warm": S.SWEEP_WARM, "nu_z": S.SWEEP_NU_Z, "theta": S.SWEEP_THETA}[a.workload]
        pcz = a.workload == "nu_z"
        for q in range(a.chains):
            smp.select_chain(q)
            if pcz:
                smp.init_state(0, 1, chain=q)
            else:
                smp.set_state(**w["state"])
    elif a.workload == "config3":
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from bench_config3 import make_config3
        w = make_config3()
        cfg = bf.default_config(model=bf.MODEL_FUNCTIONAL, K=w["K"], n_eigen=w["M"], basis_degree=3, tot_mcmc_iters=T)
        smp = bf.Sampler(cfg, w["y"], w["t"], w["internal_knots"], w["boundary_knots"], n_chains=a.chains)
        smp.set_covariates(w["X"], True)
        for q in range(a.chains):
            smp.select_chain(q)
            smp.set_state(**w["state"])
            smp.set_state(eta=w["eta"], xi=w["xi"])
        mask = S.SWEEP_WARM | S.COV_MEAN | S.COV_XI
    else:
        rng = np.random.default_rng(4)
        n, P, K, M = 8192, 50, 4, 8
        nu = rng.standard_normal((K, P)) * 2
        Phi = np.stack([(M - m) / M * 0.5 * rng.standard_normal((K, P)) for m in range(M)], axis=2)
        chi = rng.standard_normal((n, M))
        Z = rng.dirichlet(np.ones(K), size=n)
        Z = np.clip(Z, 1e-10, None)
        Z /= Z.sum(axis=1, keepdims=True)
        Y = Z @ nu + np.einsum("ik,im,kpm->ip", Z, chi, Phi) + np.sqrt(0.001) * rng.standard_normal((n, P))
        cfg = bf.default_config(model=bf.MODEL_MULTIVARIATE, K=K, n_eigen=M, tot_mcmc_iters=T)
        smp = bf.Sampler(cfg, Y, n_chains=a.chains)
        for q in range(a.chains):
            smp.select_chain(q)
            smp.set_state(nu=nu, Phi=Phi, chi=chi, Z=Z, pi=np.full(K, 1.0 / K), alpha_3=[10.0], delta=np.ones((K, M)),
                          A=np.ones((K, 2)), gamma=np.ones((K, P, M)), tau=np.ones(K), sigma_sq=[0.001])
        mask = S.SWEEP_WARM
    if a.eager:
        smp.set_profile(True)
    smp.run(mask, a.warmup, seed=2, phi_chi_zero=pcz)
    smp.prepare_run(mask, a.steps, first_iter=a.warmup, seed=2, phi_chi_zero=pcz)
    t0 = time.perf_counter()
    smp.run(mask, a.steps, first_iter=a.warmup, seed=2, phi_chi_zero=pcz)
    dt = time.perf_counter() - t0
    out = dict(workload=a.workload, chains=a.chains, steps=a.steps, ms_per_step=dt / a.steps * 1e3,
               chain_iterations_per_s=a.chains * a.steps / dt)
    if a.eager:
        out["event_ms_per_launch"] = {nm: smp.timing(nm)[0] / max(smp.timing(nm)[1], 1)
                                      for nm in ["curve_z", "pair_gram", "pg_reduce", "factor", "sweep", "curve_chi", "loglik"]}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
