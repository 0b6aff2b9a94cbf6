import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
torch.cuda.set_device(0)
def tsync(label):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): torch.cuda.synchronize()
    print(label, "torch.cuda.synchronize: %.1f us" % ((time.perf_counter() - t0) / 200 * 1e6), flush=True)
tsync("no sampler")
import bayesfmmm_amd as bf
from bench import make_config2
w = make_config2()
cfg = bf.default_config(model=bf.MODEL_FUNCTIONAL, K=w["K"], n_eigen=w["M"], basis_degree=3, tot_mcmc_iters=100)
smp = bf.Sampler(cfg, w["y"], w["t"], w["internal_knots"], w["boundary_knots"])
tsync("sampler created")
smp.set_state(**w["state"])
smp.run(bf.SWEEP_WARM, 25, first_iter=0, seed=1, chain=0)
tsync("after a run")
for rep in range(3):
    smp.prepare_run(bf.SWEEP_WARM, 20, first_iter=25, seed=1, chain=0)
    torch.cuda.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    smp.run(bf.SWEEP_WARM, 20, first_iter=25, seed=1, chain=0)
    t1 = time.perf_counter()
    torch.cuda.synchronize(); torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("run %.1f us, + two synchronizes %.1f us" % ((t1 - t0) * 1e6, (t2 - t1) * 1e6), flush=True)
