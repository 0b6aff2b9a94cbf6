#!/usr/bin/env python3
"""Times the post-processing pass (bfmmm_post_pointwise: FLLik / FDIC / FAIC / FBIC's common evaluation) on one MI355X at
the shape of BASELINE.json configs[1] (n_funct = 4096, n_i = 100, K = 3, P = 30, M = 6) over --draws saved draws, and the
oracle's restatement (oracle/post.c, one host core) on a bounded sample of the draws.  Not the bench line.

  python tests/perf/bench_post.py [--draws 500] [--cpu-draws 2]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--draws", type=int, default=500)
    ap.add_argument("--cpu-draws", type=int, default=2)
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    from bayesfmmm_amd import api
    from bench import make_config2
    w = make_config2()
    n, K, M, T = len(w["y"]), w["K"], w["M"], args.draws
    B = [np.ascontiguousarray(api.TensorBSpline(np.asarray(t).reshape(-1, 1), [3], [w["boundary_knots"]], [w["internal_knots"]]))
         for t in w["t"]]
    P = B[0].shape[1]
    rng = np.random.default_rng(1)
    nu = np.asfortranarray(rng.standard_normal((K, P, T)))
    Phi = np.asfortranarray(0.3 * rng.standard_normal((K, P, M, T)))
    chi = np.asfortranarray(rng.standard_normal((n, M, T)))
    Z = np.asfortranarray(rng.dirichlet(np.ones(K), size=(n, T)).transpose(0, 2, 1))
    sigma = rng.gamma(3.0, 0.2, size=T)
    best_ms, best_wall = 1e30, 1e30
    lib = api._lib_entry()
    for _ in range(args.reps):
        t0 = time.perf_counter()
        ll, pdf, fit = api.post_pointwise(w["y"], B, nu, Phi, Z, chi, sigma, first_kept=T // 10)
        best_wall = min(best_wall, time.perf_counter() - t0)
        best_ms = min(best_ms, lib.bfmmm_post_last_kernel_ms())
    n_obs = sum(len(v) for v in w["y"])
    alg = T * 8 * (n * (K + M) + K * (M + 1) * P + 1 + n)             # bytes per draw
    flops = T * 2.0 * (n * K * (M + 1) * P + n_obs * P)
    out = {"workload": f"post-processing pass, config-2 shape: n={n}, n_obs={n_obs}, K={K}, P={P}, M={M}, draws={T}",
           "kernel_ms": best_ms, "draws_per_s_kernel": T / best_ms * 1e3, "call_wall_s_incl_upload": best_wall,
           "algorithmic_GBps": alg / best_ms / 1e6, "hbm_frac": alg / best_ms / 1e6 / 8000.0,
           "fp64_TFLOPs": flops / best_ms / 1e9}
    if args.cpu_draws > 0:
        import oracle_lib as O
        Tc = args.cpu_draws
        model = O.Model(w["y"], B, K, M)
        ch = O.Chain(model, Tc)
        ch.nu[:], ch.Phi[:], ch.Z[:], ch.chi[:], ch.sigma[:] = nu[..., :Tc], Phi[..., :Tc], Z[..., :Tc], chi[..., :Tc], sigma[:Tc]
        t0 = time.perf_counter()
        ref = O.post_llik(model, ch)
        dt = time.perf_counter() - t0
        out["cpu_port_draws_per_s"] = Tc / dt
        out["cpu_sample"] = f"{Tc} draws, FLLik only, 1 core"
        out["llik_rel_err_vs_port"] = float(np.max(np.abs(ll[:Tc] - ref) / np.abs(ref)))
    # conditional predictive ordinates over the same draws (kept: the last 90 %)
    inp, keep = api._post_input(w["y"], B, nu, Phi, Z, chi, sigma)
    cpo = np.zeros(n)
    t0 = time.perf_counter()
    api._check(lib.bfmmm_post_cpo(api.C.byref(inp), T // 10, cpo.ctypes.data_as(api.c_double_p)))
    out["cpo_call_wall_s"] = time.perf_counter() - t0
    out["cpo_kernel_ms"] = lib.bfmmm_post_last_kernel_ms()
    out["cpo_curve_draws_per_s_kernel"] = n * (T - T // 10) / out["cpo_kernel_ms"] * 1e3
    # credible bands: 4096 kept draws x 1000 time points (column sort in LDS), wall time including the upload of the draws
    Tb, nt = 4096, 1000
    coef = np.ascontiguousarray(rng.standard_normal((Tb, P)))
    tt = np.linspace(w["boundary_knots"][0], w["boundary_knots"][1], nt)
    Bt = np.ascontiguousarray(api.TensorBSpline(tt.reshape(-1, 1), [3], [w["boundary_knots"]], [w["internal_knots"]]))
    up, md, lo = np.zeros(nt), np.zeros(nt), np.zeros(nt)
    for sim in (0, 1):
        t0 = time.perf_counter()
        api._check(lib.bfmmm_post_bands(coef.ctypes.data_as(api.c_double_p), Tb, P, Bt.ctypes.data_as(api.c_double_p), nt, 0.05, sim, 0,
                                        up.ctypes.data_as(api.c_double_p), md.ctypes.data_as(api.c_double_p), lo.ctypes.data_as(api.c_double_p), None))
        out["bands_simultaneous_wall_s" if sim else "bands_pointwise_wall_s"] = time.perf_counter() - t0
    print(json.dumps(out))


if __name__ == "__main__":
    main()
