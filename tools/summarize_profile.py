#!/usr/bin/env python3
"""Condenses the rocprofv3 databases written by tools/profile_round.sh into the small CSV / JSON summaries that are
committed under profiles/ (per-kernel launch statistics of the kernel-trace pass; FETCH_SIZE / WRITE_SIZE per launch
of the two --pmc passes, with the gfx950 correction of MI355X_MICROARCH.md: FETCH_SIZE is doubled for wide
coalesced reads -- both the raw and the corrected figure are kept)."""
import json, re, sqlite3, sys
from pathlib import Path


def short(name):
    m = re.search(r"(k_[a-z_0-9]+)", name)
    return m.group(1) if m else name[:40]


def kernel_stats(db):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name").fetchall()
    out = {}
    for name, cnt, tot, avg, mn, mx in rows:
        k = short(name)
        o = out.setdefault(k, dict(calls=0, total_ns=0, min_ns=mn, max_ns=mx))
        o["calls"] += cnt; o["total_ns"] += tot; o["min_ns"] = min(o["min_ns"], mn); o["max_ns"] = max(o["max_ns"], mx)
    for o in out.values():
        o["avg_ns"] = o["total_ns"] / o["calls"]
    return out


def pmc_per_launch(db, counter):
    cur = sqlite3.connect(db).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    name_col = "kernel_name" if "kernel_name" in cols else "name"
    rows = cur.execute(f"select {name_col}, counter_name, sum(value), count(distinct dispatch_id) from counters_collection group by {name_col}, counter_name").fetchall()
    out = {}
    for name, cname, val, nd in rows:
        if cname != counter:
            continue
        k = short(name)
        o = out.setdefault(k, dict(total=0.0, launches=0))
        o["total"] += val; o["launches"] += nd
    return {k: v["total"] / max(v["launches"], 1) for k, v in out.items()}


def main():
    src = Path(sys.argv[1]); tag = sys.argv[2]; dst = Path(sys.argv[3])
    dst.mkdir(parents=True, exist_ok=True)
    ks = kernel_stats(src / "trace" / "run_results.db")
    tot = sum(o["total_ns"] for o in ks.values())
    with open(dst / f"{tag}_kernel_stats.csv", "w") as f:
        f.write("kernel,calls,total_ns,avg_ns,min_ns,max_ns,percent\n")
        for k, o in sorted(ks.items(), key=lambda kv: -kv[1]["total_ns"]):
            f.write(f"{k},{o['calls']},{o['total_ns']},{o['avg_ns']:.1f},{o['min_ns']},{o['max_ns']},{100.0 * o['total_ns'] / tot:.2f}\n")
    pmc = {}
    for sub, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        p = src / sub / "run_results.db"
        if p.exists():
            pmc[counter] = pmc_per_launch(p, counter)
    summary = {}
    for k in ks:
        f_kb = pmc.get("FETCH_SIZE", {}).get(k); w_kb = pmc.get("WRITE_SIZE", {}).get(k)
        summary[k] = dict(avg_us=ks[k]["avg_ns"] / 1e3, calls=ks[k]["calls"],
                          fetch_size_kb_raw=f_kb, write_size_kb_raw=w_kb,
                          # FETCH_SIZE / WRITE_SIZE are reported in KB; gfx950: FETCH_SIZE counts 128-B requests at 64 B
                          hbm_read_bytes_per_launch=None if f_kb is None else 2.0 * f_kb * 1024.0,
                          hbm_write_bytes_per_launch=None if w_kb is None else w_kb * 1024.0)
    json.dump(summary, open(dst / f"{tag}_pmc_summary.json", "w"), indent=1, sort_keys=True)
    # matrix-core counters (third pass): per launch, and the share of the kernel's busy cycles the MFMA pipe was busy
    pm = src / "pmc_mfma" / "run_results.db"
    if pm.exists():
        names = ["SQ_INSTS_VALU_MFMA_MOPS_F64", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVES"]
        per = {nm: pmc_per_launch(pm, nm) for nm in names}
        mf = {}
        for k in ks:
            if not k.startswith("k_"):
                continue
            row = {nm: per[nm].get(k) for nm in names}
            if all(v is None for v in row.values()):
                continue
            busy, mbusy = row.get("SQ_BUSY_CYCLES"), row.get("SQ_VALU_MFMA_BUSY_CYCLES")
            # SQ_BUSY_CYCLES is summed over the 32 shader engines (8 XCDs x 4), SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs
            # (256 CUs x 4): the share of SIMD-cycles during which a matrix-core instruction was executing
            row["mfma_pipe_busy_frac"] = (mbusy / (busy / 32.0 * 1024.0)) if busy and mbusy is not None else None
            row["avg_us"] = ks[k]["avg_ns"] / 1e3
            mf[k] = row
        json.dump(mf, open(dst / f"{tag}_mfma_pmc.json", "w"), indent=1, sort_keys=True)
        for k, v in mf.items():
            print("mfma", k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items()})
    # FETCH_SIZE / WRITE_SIZE of the 8-chain Nu_Z batch (per launch of each kernel; a step of the batch = one launch of each
    # kernel per half-batch, calls_per_step below)
    nzf, nzw = src / "nu_z_8_pmc_fetch" / "run_results.db", src / "nu_z_8_pmc_write" / "run_results.db"
    if nzf.exists() and nzw.exists():
        ksn = kernel_stats(nzf)
        fz, wz = pmc_per_launch(nzf, "FETCH_SIZE"), pmc_per_launch(nzw, "WRITE_SIZE")
        sm = {}
        for k in ksn:
            if not k.startswith("k_"):
                continue
            f_kb, w_kb = fz.get(k), wz.get(k)
            sm[k] = dict(calls=ksn[k]["calls"], avg_us=ksn[k]["avg_ns"] / 1e3, fetch_size_kb_raw=f_kb, write_size_kb_raw=w_kb,
                         hbm_read_bytes_per_launch=None if f_kb is None else 2.0 * f_kb * 1024.0,
                         hbm_write_bytes_per_launch=None if w_kb is None else w_kb * 1024.0)
        json.dump(sm, open(dst / f"{tag.replace('_final', '')}_nu_z_8_pmc_summary.json", "w"), indent=1, sort_keys=True)
    # the same three passes for the 8-chain warm-start batch on one stream
    wf, ww, wm = (src / f"warm_8_pmc_{x}" / "run_results.db" for x in ("fetch", "write", "mfma"))
    if wf.exists() and ww.exists():
        ksn = kernel_stats(wf)
        fz, wz = pmc_per_launch(wf, "FETCH_SIZE"), pmc_per_launch(ww, "WRITE_SIZE")
        sm = {}
        for k in ksn:
            if not k.startswith("k_"):
                continue
            f_kb, w_kb = fz.get(k), wz.get(k)
            sm[k] = dict(calls=ksn[k]["calls"], avg_us=ksn[k]["avg_ns"] / 1e3, fetch_size_kb_raw=f_kb, write_size_kb_raw=w_kb,
                         hbm_read_bytes_per_launch=None if f_kb is None else 2.0 * f_kb * 1024.0,
                         hbm_write_bytes_per_launch=None if w_kb is None else w_kb * 1024.0)
        json.dump(sm, open(dst / f"{tag.replace('_final', '')}_warm_8_pmc_summary.json", "w"), indent=1, sort_keys=True)
    if wm.exists():
        ksn = kernel_stats(wm)
        names = ["SQ_INSTS_VALU_MFMA_MOPS_F64", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVES"]
        per = {nm: pmc_per_launch(wm, nm) for nm in names}
        mf = {}
        for k in ksn:
            if not k.startswith("k_"):
                continue
            row = {nm: per[nm].get(k) for nm in names}
            busy, mbusy = row.get("SQ_BUSY_CYCLES"), row.get("SQ_VALU_MFMA_BUSY_CYCLES")
            row["mfma_pipe_busy_frac"] = (mbusy / (busy / 32.0 * 1024.0)) if busy and mbusy is not None else None
            row["avg_us"] = ksn[k]["avg_ns"] / 1e3
            mf[k] = row
        json.dump(mf, open(dst / f"{tag.replace('_final', '')}_warm_8_mfma_pmc.json", "w"), indent=1, sort_keys=True)
    # the other workloads of the round: per-kernel statistics
    skip = ("trace", "pmc_fetch", "pmc_write", "pmc_mfma", "nu_z_8_pmc_fetch", "nu_z_8_pmc_write", "warm_8_pmc_fetch", "warm_8_pmc_write", "warm_8_pmc_mfma")
    for sub in sorted(p for p in src.iterdir() if p.is_dir() and (p / "run_results.db").exists() and p.name not in skip):
        ks2 = kernel_stats(sub / "run_results.db")
        tot2 = sum(o["total_ns"] for o in ks2.values())
        with open(dst / f"{tag.replace('_final', '')}_{sub.name}_kernel_stats.csv", "w") as f:
            f.write("kernel,calls,total_ns,avg_ns,min_ns,max_ns,percent\n")
            for k, o in sorted(ks2.items(), key=lambda kv: -kv[1]["total_ns"]):
                f.write(f"{k},{o['calls']},{o['total_ns']},{o['avg_ns']:.1f},{o['min_ns']},{o['max_ns']},{100.0 * o['total_ns'] / tot2:.2f}\n")
    for k, v in sorted(summary.items(), key=lambda kv: -kv[1]["avg_us"] * kv[1]["calls"])[:12]:
        print(k, {a: (round(b, 1) if isinstance(b, float) else b) for a, b in v.items()})


if __name__ == "__main__":
    main()
