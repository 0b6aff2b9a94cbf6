#!/usr/bin/env python3
"""Diagnostic: is the slow first run of the side records in bench.py an after-effect of destroying a large sampler just before?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import bayesfmmm_amd as bf
from bench import make_config2

def config3(label, steps=200, settle=0.0):
    w = make_config2()
    n = w["n"]
    T = steps + 20
    cfg = bf.default_config(model=bf.MODEL_FUNCTIONAL, K=w["K"], n_eigen=w["M"], basis_degree=3, tot_mcmc_iters=T)
    smp = bf.Sampler(cfg, w["y"], w["t"], w["internal_knots"], w["boundary_knots"])
    X = np.random.default_rng(8).standard_normal((n, 5))
    smp.set_covariates(X, True)
    mask = bf.sampler.SWEEP_WARM | bf.sampler.COV_MEAN | bf.sampler.COV_XI
    smp.set_state(**w["state"])
    if settle: time.sleep(settle)
    smp.run(mask, 20, first_iter=0, seed=2)
    smp.prepare_run(mask, steps, first_iter=20, seed=2)
    out = []
    for r in range(3):
        t0 = time.perf_counter(); smp.run(mask, steps, first_iter=20, seed=2); out.append((time.perf_counter() - t0) / steps * 1e6)
    print(label, [round(x, 1) for x in out], flush=True)
    smp.close()

def big(nch):
    w = make_config2()
    cfg = bf.default_config(model=bf.MODEL_FUNCTIONAL, K=w["K"], n_eigen=w["M"], basis_degree=3, tot_mcmc_iters=120)
    s = bf.Sampler(cfg, w["y"], w["t"], w["internal_knots"], w["boundary_knots"], n_chains=nch)
    for q in range(nch):
        s.select_chain(q); s.set_state(**w["state"])
    s.run(bf.SWEEP_WARM, 120, seed=1)
    s.close()

config3("config3 alone          ")
big(32); config3("after a 32-chain batch ")
big(32); config3("after it + 2 s settle  ", settle=2.0)
big(8); config3("after an 8-chain batch ")
config3("again alone            ")
