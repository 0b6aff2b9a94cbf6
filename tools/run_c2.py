import sys; sys.path.insert(0,'.')
import numpy as np, bayesfmmm_amd as bf
from bench import make_config2
w = make_config2()
cfg = bf.default_config(model=0, K=3, n_eigen=6, basis_degree=3, tot_mcmc_iters=12)
smp = bf.Sampler(cfg, w["y"], w["t"], w["internal_knots"], w["boundary_knots"])
smp.set_state(**w["state"])
smp.run(bf.SWEEP_WARM, 12, seed=7)
print("ok", smp.get_chain("sigma_sq"))
