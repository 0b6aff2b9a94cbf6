#!/bin/bash
# Collects the rocprofv3 evidence for one round on the GPU box (run through gpurun from the repo root):
#   1. --kernel-trace --stats of the default bench command (graph replay, no per-kernel event pass)
#   2. two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) on a short run, as MI355X_MICROARCH.md prescribes
# Outputs land in gpurun_out/prof_<tag>/; tools/summarize_profile.py condenses them into profiles/.
set -euo pipefail
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o run -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-extras --profile-steps 0 > "$OUT/bench_trace.log" 2>&1
echo "trace done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT/pmc_fetch" -o run -- python3 "$ROOT/bench.py" --steps 100 --warmup 20 --no-cpu-baseline --no-extras --profile-steps 0 > "$OUT/bench_fetch.log" 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$OUT/pmc_write" -o run -- python3 "$ROOT/bench.py" --steps 100 --warmup 20 --no-cpu-baseline --no-extras --profile-steps 0 > "$OUT/bench_write.log" 2>&1
echo "write done"
find "$OUT" -name "*.csv" | head -20
