#!/usr/bin/env python3
"""The warm-start sweep of BASELINE configs[1] (K = 3, M = 6, P = 30, cubic splines; BFMMM.h:1502-1553) as the number of curves
grows past what the caches hold: n_funct = 4096 (the benchmark: 5 MB of band-packed records, resident in L2 / Infinity Cache,
latency-bound kernels) up to 524288 (640 MB: every pass over the records comes from HBM).  Prints, per size, the iteration rate,
the band-packed algorithmic bytes per iteration (bench.py's accounting: 5 data-touching blocks x n x 8 (5 P + 1) bytes + the
per-curve state) and the fraction of the 8 TB/s HBM peak they amount to -- the roofline position of the path at scale.
Not the bench line (bench.py measures n_funct = 4096); one chain, graph replay.

  python tests/perf/scale_n.py [--sizes 4096,16384,65536,262144] [--n-i 40] [--steps 60]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="4096,16384,65536,262144")
    ap.add_argument("--n-i", type=int, default=40, help="observations per curve (set-up only: the sweep works on the per-curve statistics)")
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=10)
    a = ap.parse_args()
    import bayesfmmm_amd as bf
    from bench import algorithmic_bytes_per_iteration, make_config2
    out = []
    for n in [int(x) for x in a.sizes.split(",")]:
        w = make_config2(n=n, n_i=a.n_i)
        T = a.steps + a.warmup
        cfg = bf.default_config(model=bf.MODEL_FUNCTIONAL, K=w["K"], n_eigen=w["M"], basis_degree=3, tot_mcmc_iters=T)
        smp = bf.Sampler(cfg, w["y"], w["t"], w["internal_knots"], w["boundary_knots"])
        smp.set_state(**w["state"])
        smp.run(bf.SWEEP_WARM, a.warmup, first_iter=0, seed=1)
        smp.prepare_run(bf.SWEEP_WARM, a.steps, first_iter=a.warmup, seed=1)
        t0 = time.perf_counter()
        smp.run(bf.SWEEP_WARM, a.steps, first_iter=a.warmup, seed=1)
        dt = (time.perf_counter() - t0) / a.steps
        sig = float(smp.get_chain("sigma_sq")[T - 1])
        smp.close()
        b = algorithmic_bytes_per_iteration(n, w["P"], w["M"], w["K"], dense=False)
        rec = {"n_funct": n, "record_MB": n * 8 * (5 * w["P"] + 2) / 1e6, "us_per_iteration": dt * 1e6, "iterations_per_s": 1.0 / dt,
               "curve_updates_per_s": n / dt, "algorithmic_bytes": b, "achieved_GBps": b / dt / 1e9, "hbm_frac": b / dt / 8e12,
               "sigma_sq_last": sig}
        out.append(rec)
        print(json.dumps(rec), flush=True)
    return out


if __name__ == "__main__":
    main()
