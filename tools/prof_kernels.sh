#!/bin/bash
# rocprofv3 --kernel-trace --stats of tools/prof_workload.py for a list of "workload:chains" pairs; per-kernel CSVs land
# in gpurun_out/prof_<tag>/ (tools/rocpd_stats.py).   usage: tools/prof_kernels.sh <tag> warm:1 warm:8 nu_z:8 ...
set -euo pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  wl=${spec%%:*}; ch=${spec##*:}
  d="$OUT/${wl}_${ch}"
  rocprofv3 --kernel-trace --stats -d "$d" -o run -- python3 "$ROOT/tools/prof_workload.py" --workload "$wl" --chains "$ch" --steps 300 > "$OUT/${wl}_${ch}.json" 2> "$OUT/${wl}_${ch}.err" || { tail -5 "$OUT/${wl}_${ch}.err"; exit 1; }
  db=$(find "$d" -name "*.db" | head -1)
  python3 "$ROOT/tools/rocpd_stats.py" "$db" "$OUT/${wl}_${ch}_kernel_stats.csv" > /dev/null
  echo "== $spec  $(cat "$OUT/${wl}_${ch}.json")"
  head -8 "$OUT/${wl}_${ch}_kernel_stats.csv"
done
