#!/bin/bash
# the high-dimensional (tensor-product basis) model at scale: sweep time + per-kernel averages
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4; mkdir -p $O
python tests/perf/bench_configs.py --config 6 --steps 200 > $O/hd_bench.json 2> $O/hd_bench.err; cat $O/hd_bench.json | cut -c1-600
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/hd && rocprofv3 --kernel-trace --stats -d /tmp/hd -o w -f csv -- python3 $R/tests/perf/bench_configs.py --config 6 --steps 200 > /dev/null 2>&1
cp /tmp/hd/w_kernel_stats.csv $O/hd_kernel_stats.csv 2>/dev/null
python3 - <<PY
import csv
for r in csv.DictReader(open("/tmp/hd/w_kernel_stats.csv")):
    if "bfmmm::k_" in r["Name"] and int(r["Calls"]) >= 100:
        print("  %-40s %8.1f us x %s" % (r["Name"].split("bfmmm::")[1].split("(")[0][:40], float(r["AverageNs"]) / 1e3, r["Calls"]))
PY
