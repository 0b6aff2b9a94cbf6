#!/usr/bin/env python3
"""Diagnostic: ms per sweep of config 3 (bench.py's data and starting state) over consecutive 50-sweep runs of one chain."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import bayesfmmm_amd as bf
from bench_config3 import make_config3
S = bf.sampler
w = make_config3()
T = 520
cfg = bf.default_config(model=bf.MODEL_FUNCTIONAL, K=w["K"], n_eigen=w["M"], basis_degree=3, tot_mcmc_iters=T)
smp = bf.Sampler(cfg, w["y"], w["t"], w["internal_knots"], w["boundary_knots"])
smp.set_covariates(w["X"], True)
smp.set_state(**w["state"]); smp.set_state(eta=w["eta"], xi=w["xi"])
mask = S.SWEEP_WARM | S.COV_MEAN | S.COV_XI
smp.run(mask, 20, seed=2)
pos = 20
for r in range(10):
    smp.prepare_run(mask, 50, first_iter=pos, seed=2)
    t0 = time.perf_counter(); smp.run(mask, 50, first_iter=pos, seed=2); dt = (time.perf_counter() - t0) / 50
    s2 = smp.get_chain("sigma_sq", pos + 50)[-1]
    print(f"iterations {pos:3d}..{pos+49:3d}: {dt*1e3:.4f} ms per sweep, sigma^2 {s2:.5f}", flush=True)
    pos += 50
