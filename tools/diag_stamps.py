#!/usr/bin/env python3
"""Phase clocks of k_sweep_diag on BASELINE config 4 (library built with -DDG_STAMPS: tools/timeline.py build --sub var_dg --extra=-DDG_STAMPS)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bayesfmmm_amd as bf
S = bf.sampler
rng = np.random.default_rng(4)
n, P, K, M, T = 8192, 50, 4, 8, 40
Y = rng.standard_normal((n, P))
smp = bf.Sampler(bf.default_config(model=bf.MODEL_MULTIVARIATE, K=K, n_eigen=M, tot_mcmc_iters=T), Y)
smp.init_state(1, 3, chain=0)
smp.run(S.SWEEP_WARM, T, seed=2)
st = smp.get_state("stamps")[40:45]
print("k_sweep_diag phases (clk of the 100 MHz... shader clock):", np.diff(st), "total", st[-1] - st[0])
smp.close()
