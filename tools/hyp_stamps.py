import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(sys.path[0], "tools"))
import numpy as np
import bayesfmmm_amd as bf
from bench_config3 import make_config3
S = bf.sampler
w = make_config3()
T = 30
cfg = bf.default_config(model=bf.MODEL_FUNCTIONAL, K=w["K"], n_eigen=w["M"], basis_degree=3, tot_mcmc_iters=T)
smp = bf.Sampler(cfg, w["y"], w["t"], w["internal_knots"], w["boundary_knots"])
smp.set_covariates(w["X"], True)
smp.set_state(**w["state"]); smp.set_state(eta=w["eta"], xi=w["xi"])
smp.run(S.SWEEP_WARM | S.COV_MEAN | S.COV_XI, T, seed=2)
st = smp.get_state("stamps")
v = st[40:48]
print("k_cov_hyper phase clocks:", [int(v[i+1]-v[i]) for i in range(6)], "total", int(v[6]-v[0]))
