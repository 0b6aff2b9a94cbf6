#!/usr/bin/env python3
"""Device-side timeline of one Gibbs iteration (diagnostic build with -DBFMMM_TIMELINE; 100 MHz wall clock stamps in Dyn).

  python tools/timeline.py build                 # on the CPU box: builds tools/tl/libbfmmm_hip.so (travels with gpurun)
  BFMMM_LIB_PATH=tools/tl/libbfmmm_hip.so python tools/timeline.py run [--workload warm|nu_z] [--chains C]
"""
import argparse
import os
import shutil
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build(sub="tl", extra="-DBFMMM_TIMELINE"):
    src = os.path.join(ROOT, "bayesfmmm_amd", "csrc")
    dst = os.path.join(ROOT, "tools", sub, "csrc")
    os.makedirs(dst, exist_ok=True)
    for f in os.listdir(src):
        if f.endswith((".hip", ".hpp", ".cpp")) or f == "Makefile":
            shutil.copy(os.path.join(src, f), os.path.join(dst, f))
    # the copies include ../../include: keep the relative layout
    inc = os.path.join(ROOT, "tools", "include")
    if not os.path.exists(inc):
        os.symlink(os.path.join(ROOT, "include"), inc)
    subprocess.check_call(["make", "-s", "-j8", "-C", dst, "EXTRA=" + extra])
    print("built", os.path.join(ROOT, "tools", sub, "libbfmmm_hip.so"))


def chi_report(zt_all, zp_all, chains, n):
    """k_curve_chi, last fused launch: per-workgroup start / end (100 MHz) and phase clocks, by XCC."""
    gx = n // 8 + 8
    tot = gx * chains
    zt, zp = zt_all[:tot], zp_all[:tot]
    ok = zt[:, 0] > 0
    bx = np.arange(tot) % gx
    cur = ok & (bx >= 8)
    t00 = zt[ok, 0].min()
    st = (zt[:, 0] - t00) / 100.0
    en = (zt[:, 2] - t00) / 100.0
    dur = en - st
    xcc = (zt[:, 1] // 2**32).astype(int)
    print("  k_curve_chi (last fused launch): %d curve workgroups, kernel span %.2f us; duration us min %.2f median %.2f p90 %.2f max %.2f; start max %.2f" %
          (cur.sum(), en[ok].max(), dur[cur].min(), np.median(dur[cur]), np.percentile(dur[cur], 90), dur[cur].max(), st[cur].max()))
    sj = ok & (bx == 0)
    print("    scalar-job workgroups: duration median %.2f max %.2f, end max %.2f" % (np.median(dur[sj]), dur[sj].max(), en[sj].max()))
    names = ["dyn head", "theta", "barrier", "u_m c0 (+record)", "G u, dots", "GS rss", "fused Z"]
    print("    phase clocks (median per XCC): " + " | ".join(names))
    for x in range(8):
        m = cur & (xcc == x)
        if m.any():
            med = np.median(zp[m][:, :7], axis=0)
            print("    XCC %d: %4d wgs, dur median %.2f max %.2f, start median %.2f, end max %.2f | " % (x, m.sum(), np.median(dur[m]), dur[m].max(), np.median(st[m]), en[m].max()) +
                  " ".join("%6d" % v for v in med))
    # occupancy over time: workgroups resident, sampled every 0.5 us
    ts = np.arange(0, en[ok].max(), 0.5)
    occ = [(int(((st <= t) & (en > t) & cur).sum())) for t in ts]
    print("    resident curve workgroups every 0.5 us:", occ)
    # starts per microsecond
    print("    starts per us:", np.histogram(st[cur], bins=np.arange(0, en[ok].max() + 1, 1.0))[0].tolist())


def run(a):
    import bayesfmmm_amd as bf
    from bench import make_config2
    S = bf.sampler
    w = make_config2()
    T = 40
    cfg = bf.default_config(model=bf.MODEL_FUNCTIONAL, K=w["K"], n_eigen=w["M"], basis_degree=3, tot_mcmc_iters=T)
    smp = bf.Sampler(cfg, w["y"], w["t"], w["internal_knots"], w["boundary_knots"], n_chains=a.chains)
    pcz = a.workload == "nu_z"
    mask = S.SWEEP_NU_Z if pcz else S.SWEEP_WARM
    for q in range(a.chains):
        smp.select_chain(q)
        if pcz:
            smp.init_state(0, 1, chain=q)
        else:
            smp.set_state(**w["state"])
    smp.run(mask, T, seed=2, phi_chi_zero=pcz)
    names = ["curve_z", "pair_gram", "pg_reduce", "factor", "sweep", "curve_chi"]
    for q in sorted({0, a.chains - 1}):
        smp.select_chain(q)
        st = smp.get_state("stamps")
        t0 = min(st[2 * k] for k in range(6) if st[2 * k] > 0)
        print(f"chain {q}: last iteration, microseconds from the first kernel start  [start, latest workgroup start, end]")
        order = sorted(range(6), key=lambda k: st[2 * k])
        for k in order:
            if st[2 * k] == 0:
                continue
            print(f"  {names[k]:10s} {(st[2*k]-t0)/100:8.2f} {(st[16+k]-t0)/100:8.2f} {(st[2*k+1]-t0)/100:8.2f}   span {(st[2*k+1]-st[2*k])/100:7.2f}")
        if st[15] > 0:
            print(f"  k_curve_chi: the scalar-job workgroup (delta, A, gamma, tau) ends {(st[15] - st[10]) / 100:.2f} us into the kernel; the kernel ends at {(st[11] - st[10]) / 100:.2f} us")
        f = st[48:55]
        print("  factor workgroup 1 phases (us):", " ".join(f"{(f[i+1]-f[i])/100:.2f}" for i in range(6)), " total", (f[6] - f[0]) / 100)
        try:
            fc = smp.get_state("fct")
            print("  factor_core of workgroup 1 (us): cholesky", (fc[0] - f[5]) / 100, " inverse", (fc[1] - fc[0]) / 100, " barrier", (fc[2] - fc[1]) / 100,
                  " C = X'X (MFMA)", (fc[3] - fc[2]) / 100, " L store", (f[6] - fc[3]) / 100)
        except Exception as e:
            print("  (no fct stamps:", e, ")")
        for kind, nm in enumerate(["hyper_draws", "z_prepare", "pi_prepare", "chi_normals"]):
            s0, s1 = st[56 + 2 * kind], st[57 + 2 * kind]
            if s0 > 0:
                print(f"  spare job {nm:12s}: first workgroup {(s1 - s0) / 100:.2f} us (starts {(s0 - st[6]) / 100:.2f} us into k_factor)")
        if st[25] > st[24] > 0:
            print(f"  k_sweep_chain: chain loop {(st[25] - st[24]) / 100:.2f} us, pick spins of the chain wave {int(st[26])}; loop starts {(st[24] - st[8]) / 100:.2f} us into the kernel")
            rel = [(st[i] - st[8]) / 100 for i in (28, 24, 25, 30)] + [(st[9] - st[8]) / 100]
            print("  k_sweep_chain stamps (us from kernel start): loads back, LDS set up %.2f | loop start %.2f | chain done %.2f | all rows done %.2f | end %.2f" % tuple(rel))
        print("  k_curve_chi workgroup 10 (another XCD) first four phase clocks:", [int(st[q]) for q in (27, 29, 31, 37)])
        print("  k_curve_chi workgroup 8 phase clocks (load+stage | u_m, c0 | G u_m, dots | Gauss-Seidel, rss | fused Z):", [int(x) for x in st[32:37]])
        try:
            zt_all = smp.get_state("ztrace").reshape(-1, 3)
            zp_all = smp.get_state("zphase").reshape(-1, 8)
            if os.path.isdir(os.path.join(ROOT, "gpurun_out")) and q == 0:
                np.save(os.path.join(ROOT, "gpurun_out", f"chi_ztrace_{a.chains}.npy"), zt_all)
                np.save(os.path.join(ROOT, "gpurun_out", f"chi_zphase_{a.chains}.npy"), zp_all)
            if q == 0:
                chi_report(zt_all, zp_all, a.chains, len(w["y"]))
        except Exception as e:
            print("  (no ztrace:", e, ")")
        pg = st[40:49]
        print("  pair_gram wg0 stamps rel:", [round((x - st[2]) / 100, 2) for x in pg if x > 0])


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("cmd")
    ap.add_argument("--sub", default="tl", help="build: tools/<sub>/ receives the variant library")
    ap.add_argument("--extra", default="-DBFMMM_TIMELINE", help="build: extra compiler flags of the variant")
    ap.add_argument("--workload", default="warm")
    ap.add_argument("--chains", type=int, default=1)
    a = ap.parse_args()
    build(a.sub, a.extra) if a.cmd == "build" else run(a)
