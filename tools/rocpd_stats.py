#!/usr/bin/env python3
"""Per-kernel launch statistics from a rocprofv3 (rocpd) results database:  python tools/rocpd_stats.py run_results.db [csv_out]"""
import re
import sqlite3
import sys


def main():
    con = sqlite3.connect(sys.argv[1])
    names = [n for (n,) in con.execute("select name from sqlite_master where type='table'")]
    kd = [n for n in names if n.startswith("rocpd_kernel_dispatch")][0]
    ks = [n for n in names if n.startswith("rocpd_info_kernel_symbol")][0]
    rows = con.execute(f"select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), "
                       f"max(d.end-d.start) from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    lines = ["kernel,calls,total_ns,avg_ns,min_ns,max_ns,percent"]
    for name, cnt, t, avg, mn, mx in rows:
        m = re.search(r"(k_[a-z_0-9]+)", name)
        lines.append(f"{m.group(1) if m else name[:40]},{cnt},{t},{avg:.1f},{mn},{mx},{100.0 * t / tot:.2f}")
    out = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out)
    sys.stdout.write(out)


if __name__ == "__main__":
    main()
