#!/usr/bin/env python3
"""Diagnostic: which libamdhip64 a process ends up with (torch bundles its own), and what a 20-step run costs on the host under each."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
order = sys.argv[1]
if order == "torch_first":
    import torch; torch.cuda.set_device(0)
import bayesfmmm_amd as bf
from bayesfmmm_amd import _lib
_lib.load()
if order == "lib_first":
    import torch; torch.cuda.set_device(0)
maps = sorted({l.split()[-1] for l in open("/proc/self/maps") if "amdhip64" in l or "hsa-runtime" in l})
print(order, maps, flush=True)
from bench import make_config2
w = make_config2()
cfg = bf.default_config(model=bf.MODEL_FUNCTIONAL, K=w["K"], n_eigen=w["M"], basis_degree=3, tot_mcmc_iters=60)
smp = bf.Sampler(cfg, w["y"], w["t"], w["internal_knots"], w["boundary_knots"])
smp.set_state(**w["state"])
smp.run(bf.SWEEP_WARM, 5, first_iter=0, seed=1)
for rep in range(4):
    smp.prepare_run(bf.SWEEP_WARM, 20, first_iter=5, seed=1)
    if order != "none": torch.cuda.synchronize()
    t0 = time.perf_counter(); smp.run(bf.SWEEP_WARM, 20, first_iter=5, seed=1); dt = time.perf_counter() - t0
    print(f"  rep {rep}: wall {dt*1e6:.1f} us, device {smp.timing('total')[0]*1e3:.1f} us", flush=True)
