set -uo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_r02
cd /tmp && export TMPDIR=/tmp
export BFMMM_BATCH_SPLIT=1
mkdir -p "$OUT"
for spec in warm:8 nu_z:8; do
  wl=${spec%%:*}; ch=${spec##*:}
  rm -rf "$OUT/${wl}_${ch}"
  rocprofv3 --kernel-trace --stats -d "$OUT/${wl}_${ch}" -o run -- python3 "$ROOT/tools/prof_workload.py" --workload "$wl" --chains "$ch" --steps 300 > "$OUT/${wl}_${ch}.json" 2> "$OUT/${wl}_${ch}.err" && echo "$spec done: $(cat "$OUT/${wl}_${ch}.json")"
done
