#!/usr/bin/env python3
"""Fixed cost of one bfmmm_run call on BASELINE config 2: wall time of runs of several lengths (graphs prepared), a + b n fit."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bayesfmmm_amd as bf
from bench import make_config2

w = make_config2()
T = 1200
cfg = bf.default_config(model=bf.MODEL_FUNCTIONAL, K=w["K"], n_eigen=w["M"], basis_degree=3, tot_mcmc_iters=T)
smp = bf.Sampler(cfg, w["y"], w["t"], w["internal_knots"], w["boundary_knots"])
smp.set_state(**w["state"])
smp.run(bf.SWEEP_WARM, 5, first_iter=0, seed=1, chain=0)
pos = 5
rows = []
for n in [1, 2, 5, 10, 20, 20, 40, 80, 160, 320]:
    smp.prepare_run(bf.SWEEP_WARM, n, first_iter=pos, seed=1, chain=0)
    t0 = time.perf_counter()
    smp.run(bf.SWEEP_WARM, n, first_iter=pos, seed=1, chain=0)
    dt = time.perf_counter() - t0
    dev_ms, _ = smp.timing("total")
    rows.append((n, dt * 1e6, dev_ms * 1e3))
    pos += n
    print(f"n={n:4d}  wall {dt*1e6:9.1f} us  ({dt*1e6/n:7.2f} us/iter)   device {dev_ms*1e3:9.1f} us ({dev_ms*1e3/n:7.2f} us/iter)   host overhead {dt*1e6 - dev_ms*1e3:7.1f} us")
a = np.array(rows)
A = np.vstack([np.ones(len(a)), a[:, 0]]).T
for nm, col in (("wall", 1), ("device", 2)):
    c, *_ = np.linalg.lstsq(A, a[:, col], rcond=None)
    print(f"{nm}: fixed {c[0]:.1f} us + {c[1]:.2f} us per iteration")
smp.close()
