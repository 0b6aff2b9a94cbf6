#!/usr/bin/env python3
"""bench.py -- Gibbs iterations/sec of the BFMMM warm-start sweep on MI355X.

Workload (BASELINE.json configs[1], SURVEY.md 8(d) "Config 2"): n_funct=4096 curves, n_i=100
(t = 0:10:990), cubic B-splines with 26 equispaced internal knots (P=30), K=3, M=6, fp64,
full warm-start sweep of BFMMM_MTT_warm_start (Z, pi, alpha_3, Phi, delta, A, gamma, nu, tau,
sigma^2, chi + log-likelihood; BFMMM.h:1502-1553,1670), chain started at the generating values,
reference default hyper-parameters.  A "step" is one Gibbs iteration of one chain.

  python bench.py --gpus N --steps K --warmup W
N > 1: launched by torch.distributed.run, one rank per GPU; every rank runs an independent chain
of the same workload (multi-try chains are independent) -- weak scaling, no
data-path collective; RCCL is used for the barrier and the final gather only.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)


def make_config2(n=4096, n_i=100, K=3, M=6, n_internal=26, seed=1):
    """Synthetic data of SURVEY.md 8(d) Config 2 (numpy + scipy only)."""
    from scipy.interpolate import BSpline
    rng = np.random.default_rng(seed)
    degree = 3
    t = np.arange(n_i) * 10.0
    b0, b1 = 0.0, float(t[-1])
    ik = np.linspace(b0, b1, n_internal + 2)[1:-1]
    knots = np.concatenate([[b0] * (degree + 1), ik, [b1] * (degree + 1)])
    P = n_internal + degree + 1
    B = BSpline.design_matrix(t, knots, degree).toarray()
    nu = np.cumsum(rng.standard_normal((K, P)), axis=1) / np.sqrt(P)
    Phi = np.stack([(M - m) / M * 0.5 * rng.standard_normal((K, P)) for m in range(M)], axis=2)
    chi = rng.standard_normal((n, M))
    Z = rng.dirichlet(np.ones(K), size=n)
    Z = np.clip(Z, 1e-10, None)
    Z /= Z.sum(axis=1, keepdims=True)
    sigma_sq = 0.01
    coef = Z @ nu + np.einsum("ik,im,kpm->ip", Z, chi, Phi)
    Y = coef @ B.T + np.sqrt(sigma_sq) * rng.standard_normal((n, n_i))
    state = dict(nu=nu, Phi=Phi, chi=chi, Z=Z, pi=np.full(K, 1.0 / K), alpha_3=np.array([10.0]),
                 delta=np.ones((K, M)), A=np.ones((K, 2)), gamma=np.ones((K, P, M)), tau=np.ones(K),
                 sigma_sq=np.array([sigma_sq]))
    return dict(y=[Y[i] for i in range(n)], t=[t] * n, B=[B] * n, internal_knots=ik,
                boundary_knots=np.array([b0, b1]), K=K, M=M, P=P, n=n, n_i=n_i, degree=degree, state=state)


def algorithmic_bytes_per_iteration(n, P, M, K, n_blocks=5):
    """B_alg = N_blocks * n * 8 * (P^2 + P + 1) + 8 * n * (2M + 2K)."""
    return n_blocks * n * 8 * (P * P + P + 1) + 8 * n * (2 * M + 2 * K)


def cpu_baseline(w, iters=2):
    """Times the CPU oracle (reference-structure C restatement of the Armadillo sweep, 1 thread, the
    reference is single-threaded) on a bounded sample of the SAME workload: `iters` full-size
    warm-start sweeps (about 12 s each on a 2-3 GHz x86 core)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    model = O.Model(w["y"], w["B"], w["K"], w["M"])
    ch = O.Chain(model, iters)
    names = {"alpha_3": "alpha3", "sigma_sq": "sigma"}
    for nm, v in w["state"].items():
        arr = getattr(ch, names.get(nm, nm))
        if nm == "tau":
            arr[0, :] = v
        elif arr.ndim == 1:
            arr[0] = v[0]
        else:
            arr[..., 0] = v
    h = O.make_hyper(w["K"])
    t0 = time.perf_counter()
    O.run_sweeps(model, h, ch, O.SWEEP_WARM, n_iter=iters, seed=1)
    dt = time.perf_counter() - t0
    # the same sweep in sufficient-statistics ("Gram") form on the same core (oracle/gram.c): splits the GPU speed-up
    # into its algorithmic part (this / the reference-structure loops) and its hardware part (GPU / this)
    g_iters = 20
    ch2 = O.Chain(model, g_iters)
    for nm, v in w["state"].items():
        arr = getattr(ch2, names.get(nm, nm))
        if nm == "tau":
            arr[0, :] = v
        elif arr.ndim == 1:
            arr[0] = v[0]
        else:
            arr[..., 0] = v
    dtg = O.run_warm_gram(model, h, ch2, n_iter=g_iters, seed=1)
    return dict(value=iters / dt, unit="Gibbs iterations/sec", cores=1, kind="port",
                sample=f"{iters} full-size warm-start sweeps (n_funct={w['n']}) of the reference-structure C "
                       f"restatement of the Armadillo path, {dt:.1f} s, gcc -O2, 1 thread",
                gram_form=dict(value=g_iters / dtg, unit="Gibbs iterations/sec", cores=1,
                               sample=f"{g_iters} sweeps of the same restatement in sufficient-statistics form "
                                      f"(oracle/gram.c), {dtg:.1f} s; G_i, s_i, yy_i prepared once, not timed"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--n", type=int, default=4096)
    ap.add_argument("--profile-steps", type=int, default=200)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--concurrent-chains", type=int, default=8,
                    help="extra (untimed-region) measurement: this many independent chains at once on GPU 0")
    ap.add_argument("--concurrent-steps", type=int, default=400)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(local_rank)

    import bayesfmmm_amd as bf
    w = make_config2(n=args.n, seed=1)
    T = args.warmup + args.steps
    cfg = bf.default_config(model=bf.MODEL_FUNCTIONAL, K=w["K"], n_eigen=w["M"], basis_degree=w["degree"],
                            tot_mcmc_iters=max(T, args.profile_steps))
    smp = bf.Sampler(cfg, w["y"], w["t"], w["internal_knots"], w["boundary_knots"], device=local_rank)
    smp.set_state(**w["state"])

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    chain_id = rank                                      # independent chain per rank
    smp.run(bf.SWEEP_WARM, args.warmup, first_iter=0, seed=1, chain=chain_id)
    smp.prepare_run(bf.SWEEP_WARM, args.steps, first_iter=args.warmup, seed=1, chain=chain_id)   # graph capture is set-up
    barrier()
    t0 = time.perf_counter()
    smp.run(bf.SWEEP_WARM, args.steps, first_iter=args.warmup, seed=1, chain=chain_id)
    barrier()
    dt = time.perf_counter() - t0
    dev_ms, _ = smp.timing("total")
    if dist is not None:
        tt = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    loglik = smp.get_chain("loglik", T)
    assert np.isfinite(loglik).all()

    # per-kernel HIP-event pass (same sweep, eager launches bracketed by events on the sampler's stream)
    fams = {}
    if rank == 0 and args.profile_steps > 0:
        smp.set_state(**w["state"])
        smp.set_profile(True)
        smp.run(bf.SWEEP_WARM, args.profile_steps, first_iter=0, seed=1, chain=chain_id)
        for nm in ["curve_z", "pair_gram", "factor", "sweep", "curve_chi", "loglik"]:
            ms, cnt = smp.timing(nm)
            fams[nm] = dict(ms_per_launch=ms / max(cnt, 1), launches=cnt)
        smp.set_profile(False)

    # extra: aggregate throughput of C independent chains sharing GPU 0 (multi-try chains are independent;
    # the single-chain sweep is latency-bound and leaves most CUs idle)
    multi = None
    if rank == 0 and args.concurrent_chains > 1 and world == 1:
        import threading
        C_ = args.concurrent_chains
        cfg2 = bf.default_config(model=bf.MODEL_FUNCTIONAL, K=w["K"], n_eigen=w["M"], basis_degree=w["degree"],
                                 tot_mcmc_iters=args.concurrent_steps + 10)
        smps = [bf.Sampler(cfg2, w["y"], w["t"], w["internal_knots"], w["boundary_knots"], device=local_rank)
                for _ in range(C_)]
        for q, s_ in enumerate(smps):
            s_.set_state(**w["state"])
            s_.run(bf.SWEEP_WARM, 10, first_iter=0, seed=1, chain=100 + q)

        def work(q):
            smps[q].run(bf.SWEEP_WARM, args.concurrent_steps, first_iter=10, seed=1, chain=100 + q)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        ths = [threading.Thread(target=work, args=(q,)) for q in range(C_)]
        [t_.start() for t_ in ths]
        [t_.join() for t_ in ths]
        torch.cuda.synchronize()
        dt2 = time.perf_counter() - t1
        multi = dict(chains=C_, steps_per_chain=args.concurrent_steps, value=C_ * args.concurrent_steps / dt2,
                     unit="Gibbs iterations/sec (all chains)")
        for s_ in smps:
            s_.close()

    if rank == 0:
        n, P, M, K = w["n"], w["P"], w["M"], w["K"]
        value = world * args.steps / dt
        b_alg = algorithmic_bytes_per_iteration(n, P, M, K)
        # dominant kernel = the family with the largest time per iteration; its algorithmic bytes are the
        # data-touching update blocks it implements (n*8*(P^2+P+1) per block)
        # (k_curve_chi also carries the next iteration's Z update since the fusion: two blocks; "curve_z" is the single
        #  stand-alone launch at the start of a run)
        blocks = {"curve_z": 1, "pair_gram": 2, "sweep": 1, "curve_chi": 2, "factor": 0, "loglik": 0}
        roofline = None
        if fams:
            dom = max((k for k in fams if blocks[k] > 0), key=lambda k: fams[k]["ms_per_launch"])
            ms = fams[dom]["ms_per_launch"]
            bytes_dom = blocks[dom] * n * 8 * (P * P + P + 1)
            ach = bytes_dom / (ms * 1e-3) / 1e9
            # measured HBM-side bytes per launch of that kernel: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same
            # command (tools/profile_round.sh), condensed by tools/summarize_profile.py into profiles/ (read is corrected
            # x2 as MI355X_MICROARCH.md prescribes for gfx950); null when no summary is committed
            traffic = None
            try:
                import glob
                summ = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*_pmc_summary.json")))
                if summ:
                    pm = json.load(open(summ[-1]))
                    kname = {"sweep": ["k_sweep_fast", "k_sweep"], "curve_z": ["k_curve_z"], "curve_chi": ["k_curve_chi"],
                             "pair_gram": ["k_pair_gram", "k_pg_reduce"]}[dom]
                    tb = 0.0
                    for kn in kname:
                        if kn in pm and pm[kn].get("hbm_read_bytes_per_launch") is not None:
                            tb += pm[kn]["hbm_read_bytes_per_launch"] + (pm[kn].get("hbm_write_bytes_per_launch") or 0.0)
                    traffic = tb if tb > 0 else None
            except Exception:
                traffic = None
            roofline = dict(bound="hbm", kernel=dom, achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s",
                            frac=ach / HBM_PEAK_GBS, traffic=traffic,
                            iteration_achieved=b_alg * (args.steps / dt) / 1e9,
                            iteration_frac=b_alg * (args.steps / dt) / 1e9 / HBM_PEAK_GBS,
                            per_kernel_ms={k: round(v["ms_per_launch"], 6) for k, v in fams.items()})
        out = {
            "metric": "Gibbs iterations/sec (whole node) at n_funct=4096, K=3, P=30",
            "value": value, "unit": "Gibbs iterations/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"BFMMM_warm_start sweep, n_funct={n}, n_i={w['n_i']}, K={K}, P={P}, M={M}, "
                                   f"fp64, one independent chain per GPU", "chains": world,
                       "device_ms_per_step": dev_ms / args.steps},
            "roofline": roofline,
            "multi_chain": multi,
        }
        if not args.no_cpu_baseline and world == 1:      # reported at N = 1 only (rank 0's host cores)
            out["cpu_baseline"] = cpu_baseline(w)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
