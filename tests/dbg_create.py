import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, bayesfmmm_amd as bf
from simdata import simulate_functional
sim = simulate_functional(n=37, M=3, sigma_sq=0.01, seed=1, ragged=True)
cfg = bf.default_config(model=0, K=3, n_eigen=3, basis_degree=3, tot_mcmc_iters=8)
smp = bf.Sampler(cfg, sim["y"], sim["t"], sim["internal_knots"], sim["boundary_knots"])
print("created")
