"""Shader-clock stamps of k_curve_chi's scalar-job workgroup and of three curve workgroups on its XCD (tools/abl build)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import bayesfmmm_amd as bf
from bench import make_config2
S = bf.sampler
w = make_config2()
T = 60
cfg = bf.default_config(model=bf.MODEL_FUNCTIONAL, K=w["K"], n_eigen=w["M"], basis_degree=3, tot_mcmc_iters=T)
smp = bf.Sampler(cfg, w["y"], w["t"], w["internal_knots"], w["boundary_knots"], n_chains=1)
smp.set_state(**w["state"])
smp.run(S.SWEEP_WARM, T, seed=2)
st = smp.get_state("stamps")
j = st[40:46]; t0 = j[0]
print("job: loads+stage %.0f | recursion %.0f | tau,log,gamma %.0f | A terms %.0f | accept,stores %.0f | total %.0f clocks" %
      (j[1] - j[0], j[2] - j[1], j[3] - j[2], j[4] - j[3], j[5] - j[4], j[5] - j[0]))
for nm, a in (("block 8", 46), ("block 264", 48), ("block 512", 50)):
    print("curve workgroup %s: start %+.0f end %+.0f (duration %.0f) relative to the job's start; job end %+.0f" % (nm, st[a] - t0, st[a + 1] - t0, st[a + 1] - st[a], j[5] - t0))
