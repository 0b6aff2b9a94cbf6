import sys; sys.path.insert(0,'.')
import numpy as np, bayesfmmm_amd as bf
from bench import make_config2
w = make_config2()
cfg = bf.default_config(model=0, K=3, n_eigen=6, basis_degree=3, tot_mcmc_iters=100)
smp = bf.Sampler(cfg, w["y"], w["t"], w["internal_knots"], w["boundary_knots"])
smp.set_state(**w["state"])
smp.run(bf.SWEEP_WARM, 50)
st = smp.get_state("stamps")
names = ["start","Skm + tau rows","tau, delta rec, slog","A terms + gamma scale","accept + slots"]
print("job_hyper stamps (us):")
for i in range(1, 5): print(f"  {names[i]:32s} {(st[i]-st[i-1])*0.01:8.2f}")
print("  total", st[4]*0.01)
