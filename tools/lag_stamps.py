import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import bayesfmmm_amd as bf
from bench import make_config2
w = make_config2()
cfg = bf.default_config(model=bf.MODEL_FUNCTIONAL, K=w["K"], n_eigen=w["M"], basis_degree=3, tot_mcmc_iters=40)
smp = bf.Sampler(cfg, w["y"], w["t"], w["internal_knots"], w["boundary_knots"])
smp.set_state(**w["state"])
smp.run(bf.SWEEP_WARM, 30, seed=1)
st = smp.get_state("stamps")
for base, nm in ((40, "chain wave 0"), (48, "row wave (tid 640)")):
    v = st[base:base + 5]
    print(nm, "clocks: touch", int(v[1] - v[0]), "prologue", int(v[2] - v[1]), "loop", int(v[3] - v[2]), "tail", int(v[4] - v[3]))
