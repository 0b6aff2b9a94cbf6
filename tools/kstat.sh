#!/bin/bash
# usage: tools/kstat.sh LIB [workload] [chains]  -- per-kernel average ns of a profiled run with the given library (diagnostic)
R=${GRAFT_REPO_ROOT:-/root/repo}; W=${2:-warm}; CH=${3:-1}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ks && BFMMM_LIB_PATH=$1 rocprofv3 --kernel-trace --stats -d /tmp/ks -o w -f csv -- python3 $R/tools/prof_workload.py --workload $W --chains $CH --steps ${4:-200} ${KSTAT_ARGS:-} > /dev/null 2>&1
python3 - <<PY
import csv
for r in csv.DictReader(open("/tmp/ks/w_kernel_stats.csv")):
    if "bfmmm::k_" in r["Name"] and int(r["Calls"]) >= 100:
        print("  %-28s %8.1f" % (r["Name"].split("bfmmm::")[1].split("(")[0][:28], float(r["AverageNs"])))
PY
