import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bayesfmmm_amd as bf
from bench import make_config2
w = make_config2()
T = 200
cfg = bf.default_config(model=bf.MODEL_FUNCTIONAL, K=w["K"], n_eigen=w["M"], basis_degree=3, tot_mcmc_iters=T)
smp = bf.Sampler(cfg, w["y"], w["t"], w["internal_knots"], w["boundary_knots"])
smp.set_state(**w["state"])
smp.run(bf.SWEEP_WARM, 5, first_iter=0, seed=1, chain=0)
pos = 5
for rep in range(4):
    smp.prepare_run(bf.SWEEP_WARM, 20, first_iter=pos, seed=1, chain=0)
    t0 = time.perf_counter()
    smp.run(bf.SWEEP_WARM, 20, first_iter=pos, seed=1, chain=0)
    dt = time.perf_counter() - t0
    dev_ms, _ = smp.timing("total")
    print(f"rep {rep}: wall {dt*1e6:.1f} us, device {dev_ms*1e3:.1f} us", file=sys.stderr, flush=True)
    pos += 20
smp.close()
# a run whose remainder graph is new (rem 2 instead of 9), the unrolled one reused
smp = bf.Sampler(cfg, w["y"], w["t"], w["internal_knots"], w["boundary_knots"])
smp.set_state(**w["state"])
smp.run(bf.SWEEP_WARM, 5, first_iter=0, seed=1, chain=0)
import time as _t
for steps in (20, 20, 23, 23, 20, 41, 41):
    smp.prepare_run(bf.SWEEP_WARM, steps, first_iter=5, seed=1, chain=0)
    t0 = _t.perf_counter(); smp.run(bf.SWEEP_WARM, steps, first_iter=5, seed=1, chain=0); dt = _t.perf_counter() - t0
    dev_ms, _ = smp.timing("total")
    print(f"steps {steps}: wall {dt*1e6:.1f} us ({dt*1e6/steps:.2f}/step), device {dev_ms*1e3:.1f} us", file=sys.stderr, flush=True)
