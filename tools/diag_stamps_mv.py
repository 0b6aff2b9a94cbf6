import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import bayesfmmm_amd as bf
S = bf.sampler
rng = np.random.default_rng(4)
n, P, K, M = 8192, 50, 4, 8
nu = rng.standard_normal((K, P)) * 2
Phi = np.stack([(M - m) / M * 0.5 * rng.standard_normal((K, P)) for m in range(M)], axis=2)
chi = rng.standard_normal((n, M))
Z = rng.dirichlet(np.ones(K), size=n); Z = np.clip(Z, 1e-10, None); Z /= Z.sum(axis=1, keepdims=True)
Y = Z @ nu + np.einsum("ik,im,kpm->ip", Z, chi, Phi) + np.sqrt(0.001) * rng.standard_normal((n, P))
cfg = bf.default_config(model=bf.MODEL_MULTIVARIATE, K=K, n_eigen=M, tot_mcmc_iters=30)
smp = bf.Sampler(cfg, Y)
smp.set_state(nu=nu, Phi=Phi, chi=chi, Z=Z, pi=np.full(K, 1.0 / K), alpha_3=[10.0], delta=np.ones((K, M)), A=np.ones((K, 2)), gamma=np.ones((K, P, M)), tau=np.ones(K), sigma_sq=[0.001])
smp.run(S.SWEEP_WARM, 25, seed=2)
v = smp.get_state("stamps")[40:45]
print("k_sweep_diag clocks: setup", int(v[1] - v[0]), "touch", int(v[2] - v[1]), "loop", int(v[3] - v[2]), "sigma", int(v[4] - v[3]))
