#!/usr/bin/env python3
"""Times BASELINE.json configs[3] (multivariate: n = 8192, dim = 50, K = 4, M = 8, warm-start sweep of
BFMMM_MTT_warm_startMV, BFMMM.h:2597-2650) and configs[4] (BFMMM_Nu_Z_multiple_try semantics: 8 independent chains of
the reduced Nu_Z sweep -- Z, pi, alpha_3, nu, tau, sigma^2, loglik; Phi = chi = 0; BFMMM.h:1073-1113 -- on the config-2
data, all on ONE GPU here) on one MI355X.  Not the bench line (bench.py measures configs[1]).

  python tests/perf/bench_configs.py --config 4|5|6 [--steps N] [--no-graph]      (6: the high-dimensional model at scale)
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def config4(args):
    import bayesfmmm_amd as bf
    S = bf.sampler
    rng = np.random.default_rng(4)
    n, P, K, M = 8192, 50, 4, 8
    nu = rng.standard_normal((K, P)) * 2
    Phi = np.stack([(M - m) / M * 0.5 * rng.standard_normal((K, P)) for m in range(M)], axis=2)
    chi = rng.standard_normal((n, M))
    Z = rng.dirichlet(np.ones(K), size=n)
    Z = np.clip(Z, 1e-10, None); Z /= Z.sum(axis=1, keepdims=True)
    Y = Z @ nu + np.einsum("ik,im,kpm->ip", Z, chi, Phi) + np.sqrt(0.001) * rng.standard_normal((n, P))
    T = args.steps + args.warmup
    cfg = bf.default_config(model=bf.MODEL_MULTIVARIATE, K=K, n_eigen=M, tot_mcmc_iters=T)
    smp = bf.Sampler(cfg, Y)
    smp.set_state(nu=nu, Phi=Phi, chi=chi, Z=Z, pi=np.full(K, 1.0 / K), alpha_3=[10.0], delta=np.ones((K, M)),
                  A=np.ones((K, 2)), gamma=np.ones((K, P, M)), tau=np.ones(K), sigma_sq=[0.001])
    if args.no_graph:
        smp.set_profile(True)
    smp.run(S.SWEEP_WARM, args.warmup, seed=2)
    t0 = time.perf_counter()
    smp.run(S.SWEEP_WARM, args.steps, first_iter=args.warmup, seed=2)
    dt = (time.perf_counter() - t0) / args.steps
    b_alg = 5 * n * P * 8 + 8 * n * (2 * M + 2 * K)       # SURVEY.md 8(d), config 4 (G = I: the per-row data is y_i)
    return {"workload": "config 4: BMVMMM warm-start sweep, N=8192, dim=50, K=4, M=8", "steps": args.steps,
            "ms_per_sweep": dt * 1e3, "iterations_per_s": 1.0 / dt, "algorithmic_GBps": b_alg / dt / 1e9,
            "hbm_frac": b_alg / dt / 8e12, "sigma_sq_last": float(smp.get_chain("sigma_sq")[T - 1])}


def config5(args):
    import bayesfmmm_amd as bf
    from bench import make_config2
    S = bf.sampler
    w = make_config2()
    C = 8
    T = args.steps + args.warmup
    cfg = bf.default_config(model=bf.MODEL_FUNCTIONAL, K=w["K"], n_eigen=w["M"], basis_degree=3, tot_mcmc_iters=T)
    smps = [bf.Sampler(cfg, w["y"], w["t"], w["internal_knots"], w["boundary_knots"]) for _ in range(C)]
    for q, s_ in enumerate(smps):
        s_.init_state(0, 1, chain=q)                        # BFMMM.h:1039-1071
        if args.no_graph:
            s_.set_profile(True)
        s_.run(S.SWEEP_NU_Z, args.warmup, seed=1, chain=q, phi_chi_zero=True)
    t0 = time.perf_counter()
    smps[0].run(S.SWEEP_NU_Z, args.steps, first_iter=args.warmup, seed=1, chain=0, phi_chi_zero=True)
    dt1 = (time.perf_counter() - t0) / args.steps

    def work(q):
        smps[q].run(S.SWEEP_NU_Z, args.steps, first_iter=args.warmup, seed=1, chain=q, phi_chi_zero=True)
    ths = [threading.Thread(target=work, args=(q,)) for q in range(1, C)]
    t1 = time.perf_counter()
    [t_.start() for t_ in ths]
    [t_.join() for t_ in ths]
    dt7 = time.perf_counter() - t1
    cpu = cpu_chains(w, args.cpu_iters) if args.cpu_iters > 0 else None
    n, P = w["n"], w["P"]
    b_alg = 3 * n * 8 * (P * P + P + 1) + 8 * n * 2 * w["K"]      # 3 blocks per chain-iteration
    return {"workload": "config 5: Nu_Z sweep (BFMMM_Nu_Z_multiple_try chains) on the config-2 data, one GPU",
            "steps": args.steps, "single_chain_ms_per_sweep": dt1 * 1e3, "single_chain_iterations_per_s": 1.0 / dt1,
            "seven_concurrent_chains_iterations_per_s": 7 * args.steps / dt7,
            "single_chain_hbm_frac": b_alg / dt1 / 8e12, "cpu_all_cores": cpu}


def cpu_chains(w, iters):
    """the CPU restatement of the same chains on ALL host cores -- one independent Nu_Z chain per core
    (the reference itself is single-threaded; independent multi-try chains are the only parallelism it offers)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    cores = len(os.sched_getaffinity(0))
    C = min(8, cores)
    h = O.make_hyper(w["K"])
    models, chains = [], []
    for q in range(C):
        model = O.Model(w["y"], w["B"], w["K"], w["M"])
        ch = O.Chain(model, iters)
        st = w["state"]
        ch.nu[:, :, 0] = st["nu"]; ch.Z[:, :, 0] = st["Z"]; ch.pi[:, 0] = st["pi"]; ch.alpha3[0] = st["alpha_3"][0]
        ch.tau[0, :] = st["tau"]; ch.sigma[0] = st["sigma_sq"][0]
        ch.delta[:, :, 0] = 1.0; ch.A[:, :, 0] = 1.0; ch.gamma[..., 0] = 1.0
        models.append(model); chains.append(ch)

    def work(q):
        O.run_sweeps(models[q], h, chains[q], O.SWEEP_NU_Z, n_iter=iters, seed=1, chain_id=q)
    t0 = time.perf_counter()
    work(0)
    dt1 = time.perf_counter() - t0
    ths = [threading.Thread(target=work, args=(q,)) for q in range(C)]
    t0 = time.perf_counter()
    [t_.start() for t_ in ths]
    [t_.join() for t_ in ths]
    dtc = time.perf_counter() - t0
    return {"threads": C, "host_cores": cores, "one_thread_iterations_per_s": iters / dt1,
            "all_threads_chain_iterations_per_s": C * iters / dtc,
            "sample": f"{iters} Nu_Z sweeps per chain of the reference-structure C restatement, gcc -O2"}


def config_hd(args):
    """High-dimensional functional model at scale (not a BASELINE config; the shape of the package's HD example --
    12 x 12 grid, quadratic splines, 6 x 6 = 36 tensor basis functions, band half-width 14 -- with n = 4096 surfaces,
    K = 3, M = 4): the wide-band kernel set (BW = 31 instantiations, dense factorisation, general sweep)."""
    import bayesfmmm_amd as bf
    from bayesfmmm_amd import api
    S = bf.sampler
    rng = np.random.default_rng(6)
    n, K, M, degs = 4096, 3, 4, [2, 2]
    iks = [[250.0, 500.0, 750.0]] * 2
    bks = [[0.0, 990.0]] * 2
    g = np.arange(12) * 90.0
    tt = np.stack(np.meshgrid(g, g, indexing="ij"), axis=-1).reshape(-1, 2)
    B1 = np.ascontiguousarray(api.TensorBSpline(tt, degs, bks, iks))
    P = B1.shape[1]
    nu = rng.standard_normal((K, P)) * 2
    Phi = np.stack([(M - m) / M * 0.5 * rng.standard_normal((K, P)) for m in range(M)], axis=2)
    chi = rng.standard_normal((n, M))
    Z = rng.dirichlet(np.ones(K), size=n)
    coef = Z @ nu + np.einsum("ik,im,kpm->ip", Z, chi, Phi)
    Y = [B1 @ coef[i] + 0.1 * rng.standard_normal(len(tt)) for i in range(n)]
    T = args.steps + args.warmup
    cfg = bf.default_config(model=bf.MODEL_FUNCTIONAL, K=K, n_eigen=M, basis_degree=2, tot_mcmc_iters=T)
    smp = bf.Sampler(cfg, Y, basis=[B1] * n, band=2 * 6 + 2, penalty=api.GetP(degs, [3, 3]), penalty_band=6)
    smp.set_state(nu=nu, Phi=Phi, chi=chi, Z=Z, pi=np.full(K, 1.0 / K), alpha_3=[10.0], delta=np.ones((K, M)),
                  A=np.ones((K, 2)), gamma=np.ones((K, P, M)), tau=np.ones(K), sigma_sq=[0.01])
    if args.no_graph:
        smp.set_profile(True)
    smp.run(S.SWEEP_WARM, args.warmup, seed=2)
    t0 = time.perf_counter()
    smp.run(S.SWEEP_WARM, args.steps, first_iter=args.warmup, seed=2)
    dt = (time.perf_counter() - t0) / args.steps
    return {"workload": f"HD functional model: n={n}, 144 points per surface, P={P} (6x6 tensor basis, band 14), K={K}, M={M}",
            "steps": args.steps, "ms_per_sweep": dt * 1e3, "iterations_per_s": 1.0 / dt,
            "sigma_sq_last": float(smp.get_chain("sigma_sq")[T - 1])}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, required=True)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=0, help="config 5: also time the CPU restatement, this many sweeps per chain")
    a = ap.parse_args()
    print(json.dumps(config4(a) if a.config == 4 else config_hd(a) if a.config == 6 else config5(a)))


if __name__ == "__main__":
    main()
