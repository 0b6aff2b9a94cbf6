#!/usr/bin/env python3
"""Diagnostic: wall time of consecutive run() calls after prepare_run (the 'first run after a capture is slow' effect)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import bayesfmmm_amd as bf
from bench import make_config2

def go(n, steps, label, cov=False):
    w = make_config2(n=n, n_i=(100 if n <= 4096 else 24))
    T = steps + 10
    cfg = bf.default_config(model=bf.MODEL_FUNCTIONAL, K=w["K"], n_eigen=w["M"], basis_degree=3, tot_mcmc_iters=T)
    smp = bf.Sampler(cfg, w["y"], w["t"], w["internal_knots"], w["boundary_knots"])
    mask = bf.SWEEP_WARM
    if cov:
        rng = np.random.default_rng(8)
        X = rng.standard_normal((n, 5))
        smp.set_covariates(X, True)
        mask = bf.sampler.SWEEP_WARM | bf.sampler.COV_MEAN | bf.sampler.COV_XI
    smp.set_state(**w["state"])
    smp.run(mask, 10, first_iter=0, seed=1)
    t0 = time.perf_counter(); smp.prepare_run(mask, steps, first_iter=10, seed=1); tp = time.perf_counter() - t0
    out = []
    for r in range(4):
        t0 = time.perf_counter(); smp.run(mask, steps, first_iter=10, seed=1); out.append((time.perf_counter() - t0) / steps * 1e6)
    dev = smp.timing("total")[0] / steps * 1e3
    print(label, "prepare %.1f ms" % (tp * 1e3), "us/step per call:", [round(x, 1) for x in out], "device us/step (last):", round(dev, 1), flush=True)
    smp.close()

go(4096, 60, "n=4096 warm")
go(16384, 60, "n=16384 warm")
go(16384, 60, "n=16384 warm again")
go(65536, 60, "n=65536 warm")
go(4096, 200, "config3", cov=True)
go(4096, 200, "config3 again", cov=True)
