#!/usr/bin/env python3
"""bench.py -- Gibbs iterations/sec of the BayesFMMM sweep on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

A "step" is one Gibbs iteration of every chain in flight.  Data: SURVEY.md 8(d) "Config 2" (n_funct = 4096 curves of
n_i = 100 points, cubic B-splines with 26 internal knots => P = 30, K = 3, M = 6, fp64), synthetic.

ONE series for every N (`scaling` = "weak"): `value` = the headline of BASELINE.json configs[1] -- the full warm-start sweep of
BFMMM_MTT_warm_start (Z, pi, alpha_3, Phi, delta, A, gamma, nu, tau, sigma^2, chi + log-likelihood; BFMMM.h:1502-1553, :1670),
started at the generating values, inputs and chain resident in HBM -- with ONE independent chain per GPU: a chain does not shard
("replicas only" inside a chain), so N GPUs run N replicas with RNG chain ids 0 .. N - 1 and `value` is the
whole-node rate, N x steps / (time of the slowest rank).  No collective inside the timed steps.

The same line carries, with the SAME keys at every N, `config5`: BASELINE.json configs[4] / SURVEY 8(d) "Config 5", the 8 chains
of BFMMM_Nu_Z_multiple_try (n_try = 7; reduced sweep Z, pi, alpha_3, nu, tau, sigma^2, log-likelihood, BFMMM.h:1073-1113) on
the config-2 data, dealt round-robin over the N ranks, each rank running its chains as ONE sampler batch (strong scaling:
8 chains whatever N; one GPU already overlaps the 8 chains, so N = 8 can give about 2x over N = 1, not 8x -- DESIGN.md 7), then
the reference's final selection (all-gather of one score per rank + broadcast of the winning chain, RCCL; `gather_s`,
`time_to_best_chain_s` = steps + gather).

N = 1 only (rank 0's GPU and host cores): `roofline` (per-kernel HIP-event times of the same sweep; primary figures = the
band-packed record bytes this build stores and the counter-measured bytes, the SURVEY 8(d) dense-record figure kept beside
them as `accounting_8d`), `chains_sweep` (1 / 8 / 32 warm-start chains as one batch), `other_configs` (configs[2], [3]) and
`cpu_baseline` (the oracle on one host core).
Prints ONE JSON line on rank 0.
"""
import argparse
import glob
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md, HBM section)
FP64_MFMA_PEAK_TF = 78.6    # dense fp64 matrix peak (the same as the vector fp64 peak)
N_CHAINS_CONFIG5 = 8        # 1 + n_try, n_try = 7


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


def algorithmic_bytes_per_iteration(n, P, M, K, n_blocks=5, dense=True, bw=3):
    """B_alg = N_blocks * n * 8 * (P^2 + P + 1) + 8 * n * (2M + 2K).
    dense=False: the same count with the band-packed record this build keeps ((bw + 1) P + P + 1 doubles per curve)."""
    rec = (P * P + P + 1) if dense else ((bw + 1) * P + P + 1)
    return n_blocks * n * 8 * rec + 8 * n * (2 * M + 2 * K)


def host_cpu():
    """model name and logical core count of the host (SURVEY.md 8(d) asks for lscpu's)"""
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return model, os.cpu_count()


def cpu_baseline(w, iters=2):
    """Times the CPU oracle (reference-structure C restatement of the Armadillo sweep, 1 thread: the reference is
    single-threaded, src/Makevars has OpenMP flags but the sources no pragma) on a bounded sample of the SAME workload:
    `iters` full-size warm-start sweeps (about 6 s each on a 2-3 GHz x86 core)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    model = O.Model(w["y"], w["B"], w["K"], w["M"])
    names = {"alpha_3": "alpha3", "sigma_sq": "sigma"}

    def start_chain(T):
        ch = O.Chain(model, T)
        for nm, v in w["state"].items():
            arr = getattr(ch, names.get(nm, nm))
            if nm == "tau":
                arr[0, :] = v
            elif arr.ndim == 1:
                arr[0] = v[0]
            else:
                arr[..., 0] = v
        return ch
    h = O.make_hyper(w["K"])
    ch = start_chain(iters)
    t0 = time.perf_counter()
    O.run_sweeps(model, h, ch, O.SWEEP_WARM, n_iter=iters, seed=1)
    dt = time.perf_counter() - t0
    # the same sweep in sufficient-statistics ("Gram") form on the same core (oracle/gram.c): splits the GPU speed-up
    # into its algorithmic part (this / the reference-structure loops) and its hardware part (GPU / this)
    g_iters = 20
    dtg = O.run_warm_gram(model, h, start_chain(g_iters), n_iter=g_iters, seed=1)
    cpu_model, ncores = host_cpu()
    return dict(value=iters / dt, unit="Gibbs iterations/sec", cores=1, kind="port", host_cpu=cpu_model,
                host_logical_cores=ncores,
                sample=f"{iters} full-size warm-start sweeps (n_funct={w['n']}) of the reference-structure C "
                       f"restatement of the Armadillo path, {dt:.1f} s, gcc -O2, 1 thread (the reference is single-threaded)",
                gram_form=dict(value=g_iters / dtg, unit="Gibbs iterations/sec", cores=1,
                               sample=f"{g_iters} sweeps of the same restatement in sufficient-statistics form "
                                      f"(oracle/gram.c), {dtg:.1f} s; G_i, s_i, yy_i prepared once, not timed"))


def pmc_summary():
    """rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command (tools/profile_round.sh), condensed by
    tools/summarize_profile.py into profiles/*_pmc_summary.json (reads corrected x2 as MI355X_MICROARCH.md prescribes
    for gfx950).  The newest committed summary, or None."""
    summ = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_final_pmc_summary.json")))
    if not summ:
        return None, None
    try:
        return json.load(open(summ[-1])), os.path.basename(summ[-1])
    except Exception:
        return None, None


KERNELS_OF = {"sweep": ["k_sweep_chain", "k_sweep_fast", "k_sweep", "k_sweep_diag"], "curve_z": ["k_curve_z"], "curve_chi": ["k_curve_chi"],
              "pair_gram": ["k_pair_gram"], "pg_reduce": ["k_pg_reduce"], "factor": ["k_factor"]}


def measured_bytes(pm, fam):
    if not pm:
        return None
    tb, hit = 0.0, False
    for kn in KERNELS_OF.get(fam, []):
        if kn in pm and pm[kn].get("hbm_read_bytes_per_launch") is not None:
            tb += pm[kn]["hbm_read_bytes_per_launch"] + (pm[kn].get("hbm_write_bytes_per_launch") or 0.0)
            hit = True
    return tb if hit else None


def run_config5(bf, w, device, chain_ids, steps, warmup, seed=1):
    """This rank's share of the 8 multi-try chains as ONE sampler batch, warmed up and with its graphs captured;
    returns (sampler, number of chain slots)."""
    T = warmup + steps
    cfg = bf.default_config(model=bf.MODEL_FUNCTIONAL, K=w["K"], n_eigen=w["M"], basis_degree=w["degree"], tot_mcmc_iters=T)
    smp = bf.Sampler(cfg, w["y"], w["t"], w["internal_knots"], w["boundary_knots"], device=device, n_chains=len(chain_ids))
    stride = (chain_ids[1] - chain_ids[0]) if len(chain_ids) > 1 else 1
    smp.set_chain_id_stride(stride)
    for q, cid in enumerate(chain_ids):
        smp.select_chain(q)
        smp.init_state(0, seed, chain=cid)          # BFMMM_Nu_Z's own starting state (BFMMM.h:1039-1071)
    smp.run(bf.SWEEP_NU_Z, warmup, first_iter=0, seed=seed, chain=chain_ids[0], phi_chi_zero=True)
    smp.prepare_run(bf.SWEEP_NU_Z, steps, first_iter=warmup, seed=seed, chain=chain_ids[0], phi_chi_zero=True)
    return smp, T


def config5_scores(smp, chain_ids, T):
    """per-chain score of the reference's selection: mean of the last 99 log-likelihood values (UserFunctions.cpp:309)"""
    out = []
    for q in range(len(chain_ids)):
        smp.select_chain(q)
        ll = smp.get_chain("loglik", T)
        assert np.isfinite(ll).all()
        out.append(float(ll[max(T - 99, 0):].mean()))
    return out


def other_configs(bf, steps=200, warmup=20):
    """BASELINE.json configs[2] (covariate-adjusted, D = 5) and configs[3] (multivariate) on this GPU: ms per sweep and the
    SURVEY 8(d) byte fraction, so that driver-run numbers exist for them."""
    S = bf.sampler
    out = {}
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from bench_config3 import make_config3
    w = make_config3()
    T = steps + warmup
    cfg = bf.default_config(model=bf.MODEL_FUNCTIONAL, K=w["K"], n_eigen=w["M"], basis_degree=3, tot_mcmc_iters=T)
    smp = bf.Sampler(cfg, w["y"], w["t"], w["internal_knots"], w["boundary_knots"])
    smp.set_covariates(w["X"], True)
    smp.set_state(**w["state"])
    smp.set_state(eta=w["eta"], xi=w["xi"])
    mask = S.SWEEP_WARM | S.COV_MEAN | S.COV_XI
    def timed(smp_, mask_):
        """the same `steps` sweeps three times over (same iterations, same seed: the chain slots are simply rewritten); the
        MEDIAN is reported as ms_per_sweep and all three are kept in ms_per_sweep_runs (the first is the first call after
        prepare_run, what a user who calls once pays)"""
        smp_.run(mask_, warmup, seed=2)
        smp_.prepare_run(mask_, steps, first_iter=warmup, seed=2)
        runs, dev = [], []
        for _ in range(3):
            t0 = time.perf_counter()
            smp_.run(mask_, steps, first_iter=warmup, seed=2)
            runs.append((time.perf_counter() - t0) / steps)
            dev.append(smp_.timing("total")[0] / steps)      # the same run between its first and last device event
        return sorted(runs)[1], [r * 1e3 for r in runs], dev

    dt, runs3, dev3 = timed(smp, mask)
    n, P, M, K, D = w["n"], w["P"], w["M"], w["K"], w["D"]
    b_alg = 7 * n * 8 * (P * P + P + 1) + 8 * n * (2 * M + 2 * K) + 8 * n * D
    out["config3"] = dict(workload="covariate-adjusted Mean_CovAdj sweep (19 updates, BFMMM.h:4809-4894), n_funct=4096, D=5, K=3, P=30, M=6",
                          steps=steps, ms_per_sweep=dt * 1e3, ms_per_sweep_runs=runs3, device_ms_per_sweep_runs=dev3, iterations_per_s=1.0 / dt, algorithmic_bytes=b_alg,
                          hbm_frac_banded_records=(7 * n * 8 * (5 * P + 1) + 8 * n * (2 * M + 2 * K) + 8 * n * D) / dt / 1e9 / HBM_PEAK_GBS,
                          accounting_8d_ratio_to_peak=b_alg / dt / 1e9 / HBM_PEAK_GBS)
    smp.close()
    rng = np.random.default_rng(4)
    n, P, K, M = 8192, 50, 4, 8
    nu = rng.standard_normal((K, P)) * 2
    Phi = np.stack([(M - m) / M * 0.5 * rng.standard_normal((K, P)) for m in range(M)], axis=2)
    chi = rng.standard_normal((n, M))
    Z = rng.dirichlet(np.ones(K), size=n)
    Z = np.clip(Z, 1e-10, None)
    Z /= Z.sum(axis=1, keepdims=True)
    Y = Z @ nu + np.einsum("ik,im,kpm->ip", Z, chi, Phi) + np.sqrt(0.001) * rng.standard_normal((n, P))
    cfg = bf.default_config(model=bf.MODEL_MULTIVARIATE, K=K, n_eigen=M, tot_mcmc_iters=T)
    smp = bf.Sampler(cfg, Y)
    smp.set_state(nu=nu, Phi=Phi, chi=chi, Z=Z, pi=np.full(K, 1.0 / K), alpha_3=[10.0], delta=np.ones((K, M)),
                  A=np.ones((K, 2)), gamma=np.ones((K, P, M)), tau=np.ones(K), sigma_sq=[0.001])
    dt, runs4, dev4 = timed(smp, S.SWEEP_WARM)
    b_alg = 5 * n * P * 8 + 8 * n * (2 * M + 2 * K)
    out["config4"] = dict(workload="BMVMMM warm-start sweep (BFMMM.h:2597-2650), N=8192, dim=50, K=4, M=8", steps=steps,
                          ms_per_sweep=dt * 1e3, ms_per_sweep_runs=runs4, device_ms_per_sweep_runs=dev4, iterations_per_s=1.0 / dt, algorithmic_bytes=b_alg,
                          hbm_frac=b_alg / dt / 1e9 / HBM_PEAK_GBS)      # (G_i = I: the 8(d) figure IS what is stored, y_i only)
    smp.close()
    return out


def config5_record(bf, w, torch, dist, backend, rank, world, local_rank, steps, warmup, barrier):
    """BASELINE configs[4] with identical keys at every N: the 8 multi-try chains dealt over the ranks, timed between barriers
    (max over ranks), then the reference's selection (score all-gather + winner broadcast; local at N = 1)."""
    n, P, K = w["n"], w["P"], w["K"]
    chain_ids = list(range(rank, N_CHAINS_CONFIG5, world))
    smp, T = None, warmup + steps
    if chain_ids:
        smp, T = run_config5(bf, w, local_rank, chain_ids, steps, warmup)
    barrier()
    t0 = time.perf_counter()
    if smp is not None:
        smp.run(bf.SWEEP_NU_Z, steps, first_iter=warmup, seed=1, chain=chain_ids[0], phi_chi_zero=True)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], device="cuda" if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    # final selection of BFMMM_Nu_Z_multiple_try (UserFunctions.cpp:302-325)
    t1 = time.perf_counter()
    local = None
    if smp is not None:
        scores = config5_scores(smp, chain_ids, T)
        b = int(np.argmax(scores))
        smp.select_chain(b)
        local = dict(best_score=scores[b], best_chain=float(chain_ids[b]))
        for nm in ("nu", "Z", "pi", "alpha_3", "tau", "sigma_sq", "loglik", "A", "delta"):
            local[nm] = smp.get_chain(nm, T)
    if dist is not None:
        from bayesfmmm_amd import parallel
        best = parallel.gather_select_broadcast(local, None)
    else:
        best = local
    barrier()
    gather_s = time.perf_counter() - t1
    assert np.isfinite(best["loglik"]).all() and best["Z"].shape == (n, K, T)
    if smp is not None:
        smp.close()
    total = N_CHAINS_CONFIG5 * steps
    b_band5 = 3 * n * 8 * (4 * P + P + 1) + 8 * n * 2 * K          # band-packed records, per chain-iteration (every chain charged its own record passes)
    cpg = -(-N_CHAINS_CONFIG5 // world)                            # chains of one GPU's batch: they share ONE copy of the records
    b_shared_step = world * (3 * n * 8 * (4 * P + P + 1)) + N_CHAINS_CONFIG5 * 8 * n * 2 * K      # bytes of one step of all 8 chains: one record pass per batch and block + every chain's Z
    b_alg5 = 3 * n * 8 * (P * P + P + 1) + 8 * n * 2 * K           # SURVEY 8(d) config 5 (dense records), per chain-iteration
    pm5 = None
    try:
        f5 = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_nu_z_8_pmc_summary.json")))
        pm5 = json.load(open(f5[-1])) if f5 else None
    except Exception:
        pm5 = None
    traffic = None
    if pm5:      # measured bytes of one step of the 8-chain batch on ONE GPU (FETCH_SIZE x 2 + WRITE_SIZE): every kernel runs once
        # per half-batch and step, i.e. twice per step of the 8 chains (the profiled run: 120 steps)
        traffic = sum(2.0 * ((v.get("hbm_read_bytes_per_launch") or 0.0) + (v.get("hbm_write_bytes_per_launch") or 0.0))
                      for k, v in pm5.items() if k.startswith("k_") and v.get("calls", 0) >= 200)
    return dict(workload=f"BFMMM_Nu_Z_multiple_try chains (n_try=7: 8 chains of the Nu_Z sweep) on the config-2 data, dealt "
                         f"round-robin over {world} GPU(s), {-(-N_CHAINS_CONFIG5 // world)} per GPU as one sampler batch; STRONG scaling of 8 "
                         "latency-bound chains: one GPU already overlaps its 8 chains, so the expected ceiling at N = 8 (one chain "
                         "per GPU) is about 8 x the one-chain Nu_Z rate, i.e. roughly 1.2-1.5 x the N = 1 value, not 8 x (DESIGN.md 7)",
                scaling="strong", n_gpus=world, chains=N_CHAINS_CONFIG5, chains_per_gpu=-(-N_CHAINS_CONFIG5 // world), steps=steps,
                value=total / dt, unit="Gibbs iterations/sec (all chains)", ms_per_step=dt / steps * 1e3,
                gather_s=gather_s, time_to_best_chain_s=dt + gather_s, best_chain=int(best["best_chain"]),
                best_score=float(best["best_score"]),
                roofline=dict(bound="hbm", unit="GB/s", peak=HBM_PEAK_GBS * world,
                              achieved=b_shared_step * steps / dt / 1e9, frac=b_shared_step * steps / dt / 1e9 / (HBM_PEAK_GBS * world),
                              bytes=f"shared-copy algorithmic bytes of one step of the 8 chains: the {cpg} chain(s) of a GPU's batch read ONE "
                                    "band-packed copy of the records, 3 blocks x n x 8 (5 P + 1) per batch, + 8 n 2K of Z per chain",
                              algorithmic_bytes_per_step=b_shared_step,
                              traffic_per_step_one_gpu=traffic,
                              measured_achieved=None if (traffic is None or world != 1) else traffic * steps / dt / 1e9,
                              measured_frac=None if (traffic is None or world != 1) else traffic * steps / dt / 1e9 / HBM_PEAK_GBS,
                              per_chain_record_passes=dict(achieved=b_band5 * total / dt / 1e9, ratio_to_peak=b_band5 * total / dt / 1e9 / (HBM_PEAK_GBS * world),
                                                           note="every chain-iteration charged its own three record passes (the round-3 figure): "
                                                                "bytes a batch does not move, kept for comparison only"),
                              accounting_8d=dict(bytes_per_chain_iteration=b_alg5, ratio_to_peak=b_alg5 * total / dt / 1e9 / (HBM_PEAK_GBS * world),
                                                 note="SURVEY 8(d) charges every chain-iteration three passes over DENSE P x P "
                                                      "records; a batch shares ONE band-packed copy among its chains, so this ratio "
                                                      "can exceed 1 -- it is bookkeeping, not a bandwidth")))


def chains_sweep(bf, w, device, torch, steps=100, warmup=20, counts=(1, 8, 32)):
    """warm-start chains as ONE batch on this GPU: where each kernel stops being latency-bound"""
    out = []
    for nc in counts:
        cfg = bf.default_config(model=bf.MODEL_FUNCTIONAL, K=w["K"], n_eigen=w["M"], basis_degree=w["degree"], tot_mcmc_iters=warmup + steps)
        s = bf.Sampler(cfg, w["y"], w["t"], w["internal_knots"], w["boundary_knots"], device=device, n_chains=nc)
        for q in range(nc):
            s.select_chain(q)
            s.set_state(**w["state"])
        s.run(bf.SWEEP_WARM, warmup, first_iter=0, seed=1, chain=100)
        s.prepare_run(bf.SWEEP_WARM, steps, first_iter=warmup, seed=1, chain=100)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s.run(bf.SWEEP_WARM, steps, first_iter=warmup, seed=1, chain=100)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        s.close()
        out.append(dict(chains=nc, steps_per_chain=steps, value=nc * steps / dt, unit="Gibbs iterations/sec (all chains)",
                        ms_per_step=dt / steps * 1e3, us_per_chain_iteration=dt / steps / nc * 1e6))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--n", type=int, default=4096)
    ap.add_argument("--profile-steps", type=int, default=200)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip config5 / chains_sweep / other_configs (profiling runs)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    # rehearsal on a box with fewer GPUs than ranks (BFMMM_BENCH_BACKEND=gloo): the ranks share the visible devices and
    # the final gather runs over gloo; the driver's runs use one GPU per rank and RCCL ("nccl")
    backend = os.environ.get("BFMMM_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    import bayesfmmm_amd as bf
    w = make_config2(n=args.n, seed=1)
    n, P, M, K = w["n"], w["P"], w["M"], w["K"]

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---------------- the headline (BASELINE.json configs[1]): one warm-start chain per GPU ----------------
    T = args.warmup + args.steps
    cfg = bf.default_config(model=bf.MODEL_FUNCTIONAL, K=K, n_eigen=M, basis_degree=w["degree"],
                            tot_mcmc_iters=max(T, args.profile_steps))
    smp = bf.Sampler(cfg, w["y"], w["t"], w["internal_knots"], w["boundary_knots"], device=local_rank)
    smp.set_state(**w["state"])
    # W warm-up steps, then the set-up of the timed run (bfmmm_prepare_run: graph capture, upload and ONE dry launch of every
    # freshly instantiated graph between a snapshot and a restore of the chain's state -- the first launch of a graph costs the
    # device 13 - 20 us more than later ones; measured on one box, us per step of the 20-step form: 63.8 with the dry launch, 65.2
    # without, 65.6 with the warm-up steps moved behind the set-up: tools/gpu/ab_order.sh)
    if os.environ.get("BFMMM_BENCH_PREPARE_FIRST"):      # (A/B of the two orders)
        smp.prepare_run(bf.SWEEP_WARM, args.steps, first_iter=args.warmup, seed=1, chain=rank)
        smp.run(bf.SWEEP_WARM, args.warmup, first_iter=0, seed=1, chain=rank)
    else:
        smp.run(bf.SWEEP_WARM, args.warmup, first_iter=0, seed=1, chain=rank)
        smp.prepare_run(bf.SWEEP_WARM, args.steps, first_iter=args.warmup, seed=1, chain=rank)
    barrier()
    t0 = time.perf_counter()
    smp.run(bf.SWEEP_WARM, args.steps, first_iter=args.warmup, seed=1, chain=rank)
    t_call = time.perf_counter() - t0
    barrier()
    dt = time.perf_counter() - t0
    if os.environ.get("BFMMM_BENCH_TRACE"):
        print(f"[bench] timed region {dt * 1e6:.1f} us: the run call {t_call * 1e6:.1f} us, the closing barrier {(dt - t_call) * 1e6:.1f} us", file=sys.stderr)
    if dist is not None:
        tt = torch.tensor([dt], device="cuda" if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    dev_ms, _ = smp.timing("total")
    assert np.isfinite(smp.get_chain("loglik", T)).all()
    value = world * args.steps / dt

    # per-kernel pass (N = 1): the same sweep, eager launches bracketed by HIP events on the sampler's stream (a replayed graph
    # cannot carry per-launch events; the brackets add about 3 us per launch, so these are upper bounds -- the rocprofv3
    # averages of the same command are committed under profiles/)
    fams = {}
    if args.profile_steps > 0 and world == 1:
        smp.set_state(**w["state"])
        smp.set_profile(True)
        smp.run(bf.SWEEP_WARM, args.profile_steps, first_iter=0, seed=1, chain=0)
        for nm in ["curve_z", "pair_gram", "pg_reduce", "factor", "sweep", "curve_chi", "loglik"]:
            ms, cnt = smp.timing(nm)
            fams[nm] = dict(ms_per_launch=ms / max(cnt, 1), launches=cnt, ms_per_iteration=ms / args.profile_steps)
        smp.set_profile(False)
    smp.close()

    b_alg = algorithmic_bytes_per_iteration(n, P, M, K)
    b_band = algorithmic_bytes_per_iteration(n, P, M, K, dense=False)
    # SURVEY 8(d) data-touching blocks each kernel family implements (k_curve_chi carries chi + the next iteration's Z)
    # (the sigma^2 block is not charged to k_sweep_chain: that kernel is ONE workgroup working from the 0.45 MB of sufficient
    #  statistics -- it never reads a record, its residual sums come from k_curve_chi's pass -- and a fraction of the HBM roofline
    #  on bytes it does not move says nothing; it stays in per_kernel_ms, where it is as long as k_pair_gram)
    blocks = {"curve_z": 1, "pair_gram": 2, "pg_reduce": 0, "sweep": 0, "curve_chi": 2, "factor": 0, "loglik": 0}
    pm, pm_file = pmc_summary()
    roofline = None
    it_rate = value / world            # iterations/s of one chain
    if fams:
        # dominant kernel = largest time PER ITERATION among the families that run every iteration AND stream the per-curve records
        # (definition FIXED from round 4 on, so that the series stays comparable: `roofline` is k_pair_gram, the longest kernel
        #  that streams the per-curve records; k_sweep_chain -- as long, but ONE workgroup on 0.45 MB -- is printed beside it)
        dom = "pair_gram"
        # Each bracketed launch carries the event pair and the gap of an eager launch (about 3 us); the timed region replays
        # the same kernels as a graph without either.  The families' bracketed times per iteration are therefore brought down
        # by one common per-launch overhead, chosen so that they sum to the measured graph-replay time of an iteration: these
        # are the durations rocprofv3 reports for the same command (profiles/r03_final_kernel_stats.csv).
        it_ms = dt / args.steps * 1e3
        n_launch = sum(v["launches"] for v in fams.values()) / args.profile_steps
        over = max(0.0, (sum(v["ms_per_iteration"] for v in fams.values()) - it_ms) / max(n_launch, 1.0))
        for v in fams.values():
            v["ms_per_launch_in_graph"] = max(v["ms_per_launch"] - over, 0.0) if v["launches"] else 0.0
        ms_live = fams[dom]["ms_per_launch_in_graph"]
        prof_us = None      # the committed rocprofv3 average of the same command (profiles/r*_final_pmc_summary.json), if it is this build's
        if pm and "k_pair_gram" in pm and pm["k_pair_gram"].get("calls", 0) >= 100:
            prof_us = float(pm["k_pair_gram"]["avg_us"])
        # `frac` follows from the committed profile when it agrees with the live measurement (within 20 %: a profile of an older
        # build must not stand in for this one); the live figure is always printed beside it
        use_prof = prof_us is not None and abs(prof_us * 1e-3 - ms_live) <= 0.2 * ms_live
        ms = prof_us * 1e-3 if use_prof else ms_live
        rec_band, rec_dense = 8 * (5 * P + 1), 8 * (P * P + P + 1)          # bytes of one curve's record: stored / SURVEY 8(d)
        bytes_dom = blocks[dom] * n * rec_band
        ach = bytes_dom / (ms * 1e-3) / 1e9
        it_bytes = None
        if pm:
            parts = [measured_bytes(pm, f) for f in ("pair_gram", "pg_reduce", "factor", "sweep", "curve_chi")]
            it_bytes = sum(p for p in parts if p is not None) if any(p is not None for p in parts) else None
        # pair-Gram contraction on the fp64 matrix cores: [R pair rows x n] . [n x (BW+1) P + P columns]
        R_pairs, A_dirs = (K * (K + 1) // 2) * ((M + 1) * (M + 2) // 2), K * (M + 1)
        pg_flop = 2.0 * n * (R_pairs * 4 * P + A_dirs * P)
        roofline = dict(
            bound="hbm", kernel=dom, achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=ach / HBM_PEAK_GBS,
            kernel_definition="k_pair_gram: the longest kernel of the iteration that streams the per-curve records (fixed definition)",
            duration_us=ms * 1e3, duration_source=("rocprofv3 average, profiles/" + pm_file) if use_prof else "HIP events in this run",
            duration_us_live=ms_live * 1e3, duration_us_profile=prof_us,
            achieved_live=bytes_dom / (ms_live * 1e-3) / 1e9,
            traffic=measured_bytes(pm, dom), traffic_source=pm_file,
            traffic_frac=None if measured_bytes(pm, dom) is None else measured_bytes(pm, dom) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            sweep_chain=dict(kernel="k_sweep_chain", note="one workgroup per chain working from the sufficient statistics (no record pass): a latency chain, on no roofline",
                             duration_us=fams["sweep"]["ms_per_launch_in_graph"] * 1e3,
                             measured_bytes_per_launch=measured_bytes(pm, "sweep"),
                             achieved=None if measured_bytes(pm, "sweep") is None else measured_bytes(pm, "sweep") / (fams["sweep"]["ms_per_launch_in_graph"] * 1e-3) / 1e9,
                             frac=None if measured_bytes(pm, "sweep") is None else measured_bytes(pm, "sweep") / (fams["sweep"]["ms_per_launch_in_graph"] * 1e-3) / 1e9 / HBM_PEAK_GBS),
            algorithmic_bytes_per_launch=bytes_dom,
            bytes="band-packed records this build stores, 8 (5 P + 1) bytes per curve and data-touching block of the kernel",
            iteration=dict(algorithmic_bytes=b_band, achieved=b_band * it_rate / 1e9, frac=b_band * it_rate / 1e9 / HBM_PEAK_GBS,
                           measured_bytes=it_bytes,
                           measured_achieved=None if it_bytes is None else it_bytes * it_rate / 1e9,
                           measured_frac=None if it_bytes is None else it_bytes * it_rate / 1e9 / HBM_PEAK_GBS),
            accounting_8d=dict(kernel_bytes_per_launch=blocks[dom] * n * rec_dense,
                               kernel_ratio_to_peak=blocks[dom] * n * rec_dense / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                               iteration_bytes=b_alg, iteration_ratio_to_peak=b_alg * it_rate / 1e9 / HBM_PEAK_GBS,
                               note="SURVEY 8(d): dense P x P records, one pass per data-touching block; this build stores "
                                    "band-packed records (6x fewer bytes) that stay in L2 / Infinity Cache, so this is bookkeeping, "
                                    "not a bandwidth"),
            per_kernel_ms={k: round(v["ms_per_launch_in_graph"], 6) for k, v in fams.items()},
            per_kernel_ms_event_bracketed={k: round(v["ms_per_launch"], 6) for k, v in fams.items()},
            event_overhead_ms_per_launch=round(over, 6),
            rocprofv3_avg_us=None if not pm else {kn: round(pm[kn]["avg_us"], 3) for f_ in ("sweep", "curve_chi", "pair_gram", "pg_reduce", "factor")
                                                  for kn in KERNELS_OF[f_] if kn in pm and pm[kn].get("calls", 0) >= 100},
            per_kernel_ms_per_iteration={k: round(v["ms_per_iteration"], 6) for k, v in fams.items()},
            mfma=dict(kernel="k_pair_gram", flop_per_launch=pg_flop,
                      achieved_tflops=pg_flop / (ms * 1e-3) / 1e12,
                      peak_tflops=FP64_MFMA_PEAK_TF,
                      frac=pg_flop / (ms * 1e-3) / 1e12 / FP64_MFMA_PEAK_TF,
                      note="same duration as `duration_us`; counter evidence: profiles/*_mfma_pmc.json"),
            note="the iteration is five dependent kernels, each bound by dependent-step latency at this size (the records, 5 MB, "
                 "live in L2 / Infinity Cache): `frac` is small by construction; `traffic` (rocprofv3 FETCH_SIZE x 2 + "
                 "WRITE_SIZE per launch) and iteration.measured_* are what actually moves")
    if roofline is None:      # N > 1 (the per-kernel figures are on the N = 1 line): the iteration's bytes against N x 8 TB/s
        roofline = dict(bound="hbm", kernel="iteration (all kernels of a chain-iteration)", unit="GB/s", peak=HBM_PEAK_GBS * world,
                        achieved=b_band * value / 1e9, frac=b_band * value / 1e9 / (HBM_PEAK_GBS * world), traffic=None,
                        bytes="band-packed records this build stores, 8 (5 P + 1) bytes per curve and data-touching block",
                        accounting_8d=dict(iteration_bytes=b_alg, iteration_ratio_to_peak=b_alg * value / 1e9 / (HBM_PEAK_GBS * world)))
    out = None
    if rank == 0:
        out = {
            "metric": "Gibbs iterations/sec (whole node) at n_funct=4096, K=3, P=30",
            "value": value, "unit": "Gibbs iterations/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"BFMMM_warm_start sweep, n_funct={n}, n_i={w['n_i']}, K={K}, P={P}, M={M}, fp64, one "
                                   f"independent chain per GPU ({world} chain(s): a chain does not shard, replicas only)",
                       "chains": world, "chains_per_gpu": 1, "device_ms_per_step": dev_ms / args.steps},
            "roofline": roofline,
        }
    if not args.no_extras:
        c5 = config5_record(bf, w, torch, dist, backend, rank, world, local_rank, max(args.steps, 100), max(args.warmup, 20), barrier)
        if rank == 0:
            out["config5"] = c5
    if rank == 0 and world == 1 and not args.no_extras:
        cs = chains_sweep(bf, w, local_rank, torch)
        out["chains_sweep"] = cs
        out["multi_chain"] = dict(next(c for c in cs if c["chains"] == 8),
                                  workload="8 independent chains of the warm-start sweep as one sampler batch on this GPU")
        # the batch's pair-Gram contraction on the matrix cores (k_pair_gram_pack), from the committed counter pass of the same
        # workload on one stream (tools/profile_round.sh: profiles/r*_warm_8_mfma_pmc.json)
        try:
            fm = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_warm_8_mfma_pmc.json")))
            mp = json.load(open(fm[-1]))["k_pair_gram_pack"] if fm else None
            if mp:
                R_pairs, A_dirs = (K * (K + 1) // 2) * ((M + 1) * (M + 2) // 2), K * (M + 1)
                useful = 8 * 2.0 * n * (R_pairs * 4 * P + A_dirs * P)
                issued = mp["SQ_INSTS_VALU_MFMA_MOPS_F64"] * 512.0
                out["multi_chain"]["pair_gram_mfma"] = dict(
                    kernel="k_pair_gram_pack", source=os.path.basename(fm[-1]), duration_us=mp["avg_us"], useful_flop_per_launch=useful,
                    achieved_tflops=useful / (mp["avg_us"] * 1e-6) / 1e12, peak_tflops=FP64_MFMA_PEAK_TF,
                    frac=useful / (mp["avg_us"] * 1e-6) / 1e12 / FP64_MFMA_PEAK_TF, mfma_pipe_busy_frac=mp["mfma_pipe_busy_frac"],
                    issued_flop_per_launch=issued, tile_padding_frac=1.0 - useful / issued)
        except Exception:
            pass
        out["other_configs"] = other_configs(bf)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:      # rank 0's host cores, N = 1 only
        out["cpu_baseline"] = cpu_baseline(w)
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
